"""conv2_1 data gradient (128 reduction channels -> 64 outputs at 112x112): generic implicit GEMM vs
two conv_c64 passes (second accumulating in place)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402
from bench_igemm17 import timeit  # noqa: E402

BF16 = torch.bfloat16
N, HW, C, Co = int(os.environ.get("B", "256")), 112, 64, 128
dy = (torch.randn(N, HW, HW, Co, device="cuda") * 0.1).to(BF16)
w = (torch.randn(Co, 9 * C, device="cuda") * (2.0 / (9 * C)) ** 0.5).to(BF16)
wd = k.conv_weight_to_dgrad(w, Co, C)
halves = k.split_dgrad_weight(wd, C, Co)
flops = 2.0 * N * HW * HW * C * Co * 9
rec = dict(layer="conv2_1 dgrad", batch=N, gflop=flops / 1e9)
for name, flag in (("igemm", False), ("two_pass", True), ("igemm", False), ("two_pass", True)):
    k.two_pass_c128 = flag
    ms = timeit(lambda: k.conv3x3_dgrad(dy, wd, wd_halves=halves))
    rec.setdefault(name + "_ms", []).append(round(ms, 4))
    rec[name + "_tf"] = round(flops / ms / 1e9, 1)
print(json.dumps(rec))
open("gpurun_out/dgrad_c128.jsonl", "w").write(json.dumps(rec) + "\n")
