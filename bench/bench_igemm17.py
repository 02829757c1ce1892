"""128x128 two-stage (cfg 3) vs 512x128 8-phase (cfg 17) implicit GEMM on the VGG16 layers whose output
has <= 128 channels (conv2_2 forward / data gradient, conv3_1 data gradient)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, iters=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--out", default="gpurun_out/igemm17.jsonl")
    ap.add_argument("--cfgs", default="3,17", help="two igemm cfg ids to alternate (18: 3-stage ring, two workgroups per CU)")
    args = ap.parse_args()
    N = args.batch
    ca, cb = [int(c) for c in args.cfgs.split(",")]
    out = open(args.out, "w")
    for name, C, Co, HW, kind in [("conv2_2 fprop", 128, 128, 112, "f"), ("conv2_2 dgrad", 128, 128, 112, "d"),
                                  ("conv3_1 dgrad", 128, 256, 56, "d")]:
        x = torch.randn(N, HW, HW, C, device="cuda").to(BF16)
        dy = (torch.randn(N, HW, HW, Co, device="cuda") * 0.1).to(BF16)
        w = (torch.randn(Co, 9 * C, device="cuda") * (2.0 / (9 * C)) ** 0.5).to(BF16)
        b = torch.zeros(Co, device="cuda")
        wd = k.conv_weight_to_dgrad(w, Co, C)
        flops = 2.0 * N * HW * HW * C * Co * 9
        fn = (lambda: k.conv3x3_fprop(x, w, b, True)) if kind == "f" else (lambda: k.conv3x3_dgrad(dy, wd, mask_src=x))
        rec = dict(layer=name, batch=N, gflop=flops / 1e9)
        outs = {}
        for cfg in (ca, cb, ca, cb, ca, cb):
            k._overrides["igemm"] = lambda M, Nc, ks, cfg=cfg: (cfg, 1)
            try:
                ms = timeit(fn)
                outs[cfg] = fn()
            finally:
                k._overrides.pop("igemm")
            rec.setdefault("cfg%d_ms" % cfg, []).append(round(ms, 4))
            rec["cfg%d_tf" % cfg] = round(flops / ms / 1e9, 1)
        rec["bit_equal"] = bool(torch.equal(outs[ca], outs[cb]))
        rec["max_abs_diff"] = float((outs[ca].float() - outs[cb].float()).abs().max())
        rec["ref_abs_max"] = float(outs[ca].float().abs().max())
        print(json.dumps(rec), flush=True)
        out.write(json.dumps(rec) + "\n")
        del x, dy, w, wd, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
