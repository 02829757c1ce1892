"""Classifier GEMMs of VGG16 at a given batch: linear1 / linear2 forward under different implicit-GEMM
configurations (tile cfg, split-K) through the tuning hook of ops/kernels.py.
  python bench/bench_fc.py [--batch 1024]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as K  # noqa: E402


def timed(fn, iters=30):
    for _ in range(3):
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
    ap.add_argument("--batch", type=int, default=1024)
    a = ap.parse_args()
    B = a.batch
    for name, Kin, Nout in (("linear1", 25088, 4096), ("linear2", 4096, 4096)):
        x = torch.randn(B, Kin, device="cuda").to(torch.bfloat16)
        w = (torch.randn(Nout, Kin, device="cuda") * 0.01).to(torch.bfloat16)
        b = torch.zeros(Nout, device="cuda")
        ref = None
        for cfg, ns in ((None, None), (3, 1), (3, 2), (3, 4), (16, 1), (16, 2), (16, 4), (16, 8), (0, 4)):
            if cfg is None:
                K._overrides.pop("igemm", None)
            else:
                K._overrides["igemm"] = lambda M, N, ks, cfg=cfg, ns=ns: (cfg, ns)
            try:
                y = K.linear_fwd(x, w, b, relu=True)
                torch.cuda.synchronize()
                if ref is None:
                    ref = y.float()
                err = (y.float() - ref).abs().max().item()
                ms = timed(lambda: K.linear_fwd(x, w, b, relu=True))
                print(json.dumps({"layer": name, "batch": B, "cfg": cfg, "nsplit": ns, "ms": round(ms, 4),
                                  "tf": round(2.0 * B * Kin * Nout / ms / 1e9, 1), "max_abs_diff_vs_default": err}))
            except Exception as e:      # a configuration the launcher refuses
                print(json.dumps({"layer": name, "cfg": cfg, "nsplit": ns, "error": str(e)[:80]}))
        K._overrides.pop("igemm", None)


if __name__ == "__main__":
    main()
