"""Max-pool backward (codes form) alone on the GPU: effective HBM bandwidth per VGG block.
A/B: DTP_POOL_BWD_ROWS=0 (window-major kernel) vs 1 (row-pair kernel, default).  One JSON line per shape.
  python bench/bench_pool.py [--batch 256]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
BF16 = torch.bfloat16
for name, HW, C in [("block2", 112, 128), ("block3", 56, 256), ("block4", 28, 512), ("block5", 14, 512)]:
    N = args.batch
    x = torch.randn(N, HW, HW, C, device="cuda").to(BF16)
    y, idx = K.maxpool2x2_fwd(x, want_idx=True)
    dy = torch.randn_like(y)
    ref = K.maxpool2x2_bwd(x, dy, relu_mask=True)             # window-major kernel on the pool input
    K.colsum_slot.take(ref, C)
    got = K.maxpool2x2_bwd(None, dy, relu_mask=True, idx=idx, in_shape=tuple(x.shape))
    K.colsum_slot.take(got, C)
    same = bool(torch.equal(ref, got))
    del x, ref
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        got = K.maxpool2x2_bwd(None, dy, relu_mask=True, idx=idx, in_shape=(N, HW, HW, C))
    ev0.record()
    for _ in range(args.iters):
        got = K.maxpool2x2_bwd(None, dy, relu_mask=True, idx=idx, in_shape=(N, HW, HW, C))
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / args.iters
    nbytes = got.numel() * 2 + dy.numel() * 2 + idx.numel() * 4
    print(json.dumps({"shape": name, "batch": N, "rows_form": os.environ.get("DTP_POOL_BWD_ROWS", "1"),
                      "ms": round(ms, 4), "TB_s": round(nbytes / ms / 1e9, 3), "matches_window_major": same}),
          flush=True)
    del got, dy, idx, y
    torch.cuda.empty_cache()
