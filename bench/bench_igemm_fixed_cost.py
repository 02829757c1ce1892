"""Fixed cost per tile of the 8-phase implicit GEMM: time at constant (M, N) and varying reduction
depth (C = 64..512 -> 9..72 K-tiles); T = rounds * (a + b * ktiles)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402
from bench_igemm17 import timeit  # noqa: E402

BF16 = torch.bfloat16
N_IMG, HW, Co = int(os.environ.get("B", "256")), int(os.environ.get("HW", "56")), int(os.environ.get("CO", "256"))
rows = []
for C in (64, 128, 256, 512):
    x = torch.randn(N_IMG, HW, HW, C, device="cuda").to(BF16)
    w = (torch.randn(Co, 9 * C, device="cuda") * (2.0 / (9 * C)) ** 0.5).to(BF16)
    b = torch.zeros(Co, device="cuda")
    k._overrides["igemm"] = lambda M, Nc, ks: (16, 1)
    try:
        ms = min(timeit(lambda: k.conv3x3_fprop(x, w, b, True), iters=10) for _ in range(3))
    finally:
        k._overrides.pop("igemm")
    M = N_IMG * HW * HW
    tiles = -(-M // 256) * -(-Co // 256)
    rounds = -(-tiles // 256)
    rec = dict(C=C, ktiles=9 * C // 64, ms=round(ms, 4), tiles=tiles, rounds=rounds, us_per_round=round(1e3 * ms / rounds, 2),
               tf=round(2.0 * M * Co * 9 * C / ms / 1e9, 1))
    rows.append(rec)
    print(json.dumps(rec), flush=True)
    del x, w
(k0, t0), (k1, t1) = [(r["ktiles"], r["us_per_round"]) for r in (rows[1], rows[3])]
bb = (t1 - t0) / (k1 - k0)
print(json.dumps(dict(us_per_ktile=round(bb, 3), fixed_us_per_tile=round(t0 - bb * k0, 2))))
