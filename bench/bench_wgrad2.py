"""A/B of the weight-gradient kernels on the VGG16 shapes (batch 256 by default): cfg 6 / 7
(wgrad.hip, builtin transposed reads) vs cfg 8 / 9 (wgrad2.hip), interleaved rounds in one process,
random data (cdna_hip_programming.md 5.4 rules 24, 25)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402

SHAPES = [(128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28), (512, 512, 28), (512, 512, 14)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--cfgs", default="6,8")
    args = ap.parse_args()
    cfgs = [int(c) for c in args.cfgs.split(",")]
    N = args.batch
    for (C, Co, HW) in SHAPES:
        x = torch.randn(N, HW, HW, C, device="cuda").to(torch.bfloat16)
        dy = (torch.randn(N, HW, HW, Co, device="cuda") * 0.1).to(torch.bfloat16)
        gw = torch.empty(Co, 9 * C, device="cuda")
        flops = 2.0 * N * HW * HW * C * Co * 9
        ks = (N * HW * HW + 63) // 64
        res = {c: [] for c in cfgs}
        splits = {}
        _, nsplit = k.pick_wgrad(Co, C, 9, ks)          # production split for this tile size
        for c in cfgs:
            use = c
            if C % 256 and c in (6, 8):
                use = {6: 7, 8: 9}[c]
            splits[c] = (use, nsplit)
        for r in range(args.rounds):
            for c in cfgs:
                use, nsplit = splits[c]
                k._overrides["wgrad"] = lambda M, Cc, t, kk, u=use, n=nsplit: (u, n)
                try:
                    for _ in range(2):
                        k.conv3x3_wgrad(dy, x, gw)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        k.conv3x3_wgrad(dy, x, gw)
                    e1.record()
                    torch.cuda.synchronize()
                    res[c].append(flops / (e0.elapsed_time(e1) / args.iters) / 1e9)
                finally:
                    k._overrides.pop("wgrad")
        print(json.dumps(dict(shape="%dx%d@%d" % (C, Co, HW), batch=N,
                              **{"cfg%d_tf_median" % c: sorted(v)[len(v) // 2] for c, v in res.items()},
                              **{"cfg%d_used" % c: list(splits[c]) for c in cfgs})), flush=True)


if __name__ == "__main__":
    main()
