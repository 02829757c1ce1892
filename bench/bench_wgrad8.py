"""2-stage (cfg 0) vs 8-phase (cfg 5) 256x256 weight-gradient kernel on the VGG16 layers that use
it, at the split factor pick_wgrad chooses.  TF/s on random data."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
BF16 = torch.bfloat16


def timeit(fn, iters=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


for (C, Co, HW) in [(256, 256, 56), (256, 512, 28), (512, 512, 28), (512, 512, 14), (128, 128, 112), (128, 256, 56)]:
    x = torch.randn(N, HW, HW, C, device="cuda").to(BF16)
    dy = (torch.randn(N, HW, HW, Co, device="cuda") * 0.1).to(BF16)
    gw = torch.empty(Co, 9 * C, device="cuda", dtype=BF16)
    flops = 2.0 * N * HW * HW * C * Co * 9
    ksteps = (N * HW * HW + 63) // 64
    _, nsplit = k.pick_wgrad(Co, C, 9, ksteps)
    rec = dict(shape="%dx%d@%d" % (C, Co, HW), batch=N, nsplit=nsplit)
    for cfg in ((0, 6) if C >= 256 else (1, 7)):
        for sp in (nsplit,):
            k._overrides["wgrad"] = lambda M, Cc, taps, ks, cfg=cfg, sp=sp: (cfg, sp)
            try:
                rec["cfg%d_s%d_tf" % (cfg, sp)] = round(flops / timeit(lambda: k.conv3x3_wgrad(dy, x, gw)) / 1e9)
            finally:
                k._overrides.pop("wgrad")
    print(json.dumps(rec), flush=True)
