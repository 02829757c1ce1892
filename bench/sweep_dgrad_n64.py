"""conv2_1's data gradient (128 reduction channels -> 64 outputs at 112x112): the 64-wide implicit-GEMM
tile configurations against each other (the picker's choice is cfg 2)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402
from bench_igemm17 import timeit  # noqa: E402

BF16 = torch.bfloat16
N, HW, C, Co = int(os.environ.get("B", "1024")), 112, 64, 128
dy = (torch.randn(N, HW, HW, Co, device="cuda") * 0.1).to(BF16)
w = (torch.randn(Co, 9 * C, device="cuda") * (2.0 / (9 * C)) ** 0.5).to(BF16)
wd = k.conv_weight_to_dgrad(w, Co, C)
flops = 2.0 * N * HW * HW * C * Co * 9
ref = None
for rnd in range(2):
    for cfg in (2, 4, 10, 12):
        k._overrides["igemm"] = lambda M, Nc, ks, cfg=cfg: (cfg, 1)
        try:
            ms = timeit(lambda: k.conv3x3_dgrad(dy, wd), iters=10)
            out = k.conv3x3_dgrad(dy, wd)
        finally:
            k._overrides.pop("igemm")
        ref = out if ref is None else ref
        print(json.dumps(dict(cfg=cfg, tile=k.IGEMM_TILES[cfg], ms=round(ms, 4), tf=round(flops / ms / 1e9, 1),
                              bit_equal=bool(torch.equal(out, ref)))), flush=True)
