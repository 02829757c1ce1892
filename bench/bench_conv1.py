"""First-layer kernels alone on the GPU (conv1_fprop, conv1_wgrad with self-masking): ms and
effective HBM bandwidth (the 64-channel full-resolution activation dominates the traffic).
  python bench/bench_conv1.py [--batch 1024] [--in-dtype f32|u8]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--in-dtype", default="f32", choices=["f32", "u8"])
args = ap.parse_args()
N, HW = args.batch, 224
BF16 = torch.bfloat16
if args.in_dtype == "u8":
    x = torch.randint(0, 256, (N, 3, HW, HW), device="cuda", dtype=torch.uint8)
    kw = dict(scale=1.0 / 255, mean=torch.tensor([0.485, 0.456, 0.406], device="cuda"),
              inv_std=1.0 / torch.tensor([0.229, 0.224, 0.225], device="cuda"))
else:
    x = torch.randn(N, 3, HW, HW, device="cuda")
    kw = {}
w = (torch.randn(64, 27, device="cuda") * 0.27).to(BF16)
b = torch.randn(64, device="cuda") * 0.1
dy = torch.randn(N, HW, HW, 64, device="cuda").to(BF16)
gw, gb = torch.empty(64, 27, device="cuda"), torch.empty(64, device="cuda")


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters


act = dy.numel() * 2
ms = timed(lambda: K.conv1_fprop(x, w, b, relu=True, **kw))
print(json.dumps({"kernel": "conv1_fprop", "batch": N, "in": args.in_dtype, "ms": round(ms, 4),
                  "TB_s": round((act + x.numel() * x.element_size()) / ms / 1e9, 3),
                  "grid_cap": os.environ.get("DTP_CONV1_GRID", "default")}), flush=True)
ms = timed(lambda: K.conv1_wgrad(x, dy, gw, gb, mask_w=w, mask_bias=b, **kw))
print(json.dumps({"kernel": "conv1_wgrad (self-mask)", "batch": N, "in": args.in_dtype, "ms": round(ms, 4),
                  "TB_s": round((act + x.numel() * x.element_size()) / ms / 1e9, 3)}), flush=True)
