"""Inference (eval.py's path: ``model.eval()`` + ``torch.no_grad``) throughput on one MI355X, this
framework vs stock PyTorch-ROCm (bf16 autocast, channels_last, MIOpen / hipBLASLt): VGG16 and
VGG16-BN (here: BatchNorm folded into the convolution's scale / bias at eval time) on device-resident
synthetic 224x224 batches.  Not a BASELINE.json metric — the serving-side counterpart of bench.py.

  python bench/bench_infer.py [--batch 256] [--steps 20] [--batch-norm] [--stock]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-norm", action="store_true")
    ap.add_argument("--fp8-classifier", action="store_true")
    ap.add_argument("--stock", action="store_true")
    ap.add_argument("--uint8", action="store_true", help="uint8 pixels, normalisation fused into the first convolution")
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    if args.stock:
        from bench_stock import make_vgg16
        model = make_vgg16(batch_norm=args.batch_norm).to(dev).to(memory_format=torch.channels_last).eval()
        x = torch.randn(args.batch, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)

        def fwd():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return model(x)
    else:
        from distributed_training_pytorch_amd.models.vgg16 import VGG16
        model = VGG16(3, 3, True, batch_norm=args.batch_norm, fp8_classifier=args.fp8_classifier).to(dev).eval()
        if args.uint8:
            model.set_input_normalization(1.0 / 255.0, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
            x = torch.randint(0, 256, (args.batch, 3, 224, 224), device=dev, dtype=torch.uint8)
        else:
            x = torch.randn(args.batch, 3, 224, 224, device=dev)
        fwd = lambda: model(x)                                         # noqa: E731
    with torch.no_grad():
        for _ in range(args.warmup):
            out = fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fwd()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert torch.isfinite(out.float()).all()
    rec = dict(impl="stock-pytorch-rocm" if args.stock else "this", mode="inference", batch=args.batch,
               batch_norm=args.batch_norm, fp8_classifier=args.fp8_classifier, uint8=args.uint8,
               img_s=round(args.batch * args.steps / dt, 1), ms_per_batch=round(dt / args.steps * 1e3, 3),
               mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
