"""Fused flat SGD alone on the GPU over a VGG16-sized arena (134 M parameters, bf16 gradients) and, second,
next to a stream of small kernels on another stream: how long does a 10 us kernel take while the update runs?
DTP_OPT_GRID_CAP selects the optimizer grid (512 = 2 workgroups per CU, the default; 4096 = the old grid)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd import _lib  # noqa: E402


def main():
    n = 134_272_832 // 256 * 256
    dev = torch.device("cuda")
    p = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev).to(torch.bfloat16)
    m = torch.zeros(n, device=dev)
    lp = torch.empty(n, device=dev, dtype=torch.bfloat16)
    hip = _lib.hip()
    vp = _lib.c_void_p

    def sgd(stream):
        rc = hip.dtp_sgd_flat(vp(p.data_ptr()), vp(g.data_ptr()), 1, vp(m.data_ptr()), vp(lp.data_ptr()), n,
                              0.001, 0.9, 0.0, 1e-4, 1.0, 0, 0, vp(stream.cuda_stream))
        assert rc == 0
    s0 = torch.cuda.current_stream()
    for _ in range(3):
        sgd(s0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        sgd(s0)
    e1.record()
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1) / 20
    gb = n * (4 + 2 + 4 + 4 + 4 + 2) / 1e9
    # small kernels next to it
    side = torch.cuda.Stream()
    a = torch.randn(256, 1024, device=dev)
    small = []
    for _ in range(5):
        side.wait_stream(s0)
        sgd(s0)
        with torch.cuda.stream(side):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            evs[0].record()
            for k in range(20):
                a.mul_(1.0001)
                evs[k + 1].record()
        torch.cuda.synchronize()
        small.append(max(evs[k].elapsed_time(evs[k + 1]) for k in range(20)))
    print(json.dumps({"grid_cap": os.environ.get("DTP_OPT_GRID_CAP", "512 (default)"), "sgd_ms": round(alone, 4),
                      "TB_per_s": round(gb / alone, 3),
                      "worst_small_kernel_ms_next_to_sgd": round(max(small), 4)}))


if __name__ == "__main__":
    main()
