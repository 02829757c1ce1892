"""Does a memory-bound kernel overlap with a matrix kernel on this chip?  conv3_2-shaped implicit GEMM
(stream A) and a block-2 max-pool backward (stream B) on independent tensors: serial vs concurrent wall
time.  The matrix kernels are power-limited (docs/ROUND_NOTES.md), so overlap could only pay through
the power head-room of the memory-bound kernel.   python bench/bench_overlap_mem_compute.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as K  # noqa: E402

BF16 = torch.bfloat16
N = int(os.environ.get("N", "256"))
x = torch.randn(N, 56, 56, 256, device="cuda").to(BF16)
w = (torch.randn(256, 9 * 256, device="cuda") * 0.02).to(BF16)
b = torch.zeros(256, device="cuda")
px = torch.randn(N, 112, 112, 128, device="cuda").to(BF16)
py, idx = K.maxpool2x2_fwd(px, want_idx=True)
pdy = torch.randn_like(py)
del px
K.fuse_colsum = False
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
REP_C, REP_M = 8, 32          # ~8 ms of convolutions, ~6.7 ms of pool backward


def conv():
    for _ in range(REP_C):
        K.conv3x3_fprop(x, w, b, True)


def pool():
    for _ in range(REP_M):
        K.maxpool2x2_bwd(None, pdy, relu_mask=True, idx=idx, in_shape=(N, 112, 112, 128))


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


def serial():
    with torch.cuda.stream(sa):
        conv()
        pool()


def concurrent():
    with torch.cuda.stream(sa):
        conv()
    with torch.cuda.stream(sb):
        pool()


def only(fn):
    def run():
        with torch.cuda.stream(sa):
            fn()
    return run


for f in (serial, concurrent):
    wall(f)
res = dict(batch=N)
for name, f in (("conv_only_ms", only(conv)), ("pool_only_ms", only(pool)), ("serial_ms", serial),
                ("concurrent_ms", concurrent), ("serial_ms_2", serial), ("concurrent_ms_2", concurrent)):
    res[name] = round(min(wall(f) for _ in range(3)), 3)
print(json.dumps(res))
