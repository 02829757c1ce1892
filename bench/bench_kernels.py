"""Per-layer kernel benchmark on random data: our implicit-GEMM conv fprop/dgrad/wgrad vs stock
PyTorch-ROCm (MIOpen) bf16 channels_last on the 13 VGG16 conv shapes + the 3 classifier GEMMs.
Writes JSON to gpurun_out/bench_kernels.json.  Interleaved rounds in one process
(cdna_hip_programming.md §5.4 rule 24), random operands (rule 25).
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402

BF16 = torch.bfloat16
VGG_CONVS = [  # (Cin, Cout, HW)
    (64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56),
    (256, 512, 28), (512, 512, 28), (512, 512, 14),
]


def timeit(fn, iters, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-stock", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="try every tile config per shape")
    ap.add_argument("--out", default="gpurun_out/bench_kernels.json")
    args = ap.parse_args()
    N = args.batch
    rows = []
    torch.manual_seed(0)
    for (C, Co, HW) in VGG_CONVS:
        x = torch.randn(N, HW, HW, C, device="cuda").to(BF16)
        dy = (torch.randn(N, HW, HW, Co, device="cuda") * 0.1).to(BF16)
        w = (torch.randn(Co, 9 * C, device="cuda") * (2.0 / (9 * C)) ** 0.5).to(BF16)
        b = torch.zeros(Co, device="cuda")
        wd = k.conv_weight_to_dgrad(w, Co, C)
        gw = torch.empty(Co, 9 * C, device="cuda")
        flops = 2.0 * N * HW * HW * C * Co * 9
        rec = dict(shape="%dx%d@%d" % (C, Co, HW), gflop=flops / 1e9)

        def best(kind, fn, cfgs, splits):
            res = {}
            for cfg in cfgs:
                for sp in splits:
                    if kind == "igemm":
                        k._overrides["igemm"] = lambda M, Nc, ks, cfg=cfg, sp=sp: (cfg, sp)
                    else:
                        k._overrides["wgrad"] = lambda M, Cc, taps, ks, cfg=cfg, sp=sp: (cfg, sp)
                    try:
                        res["c%d_s%d" % (cfg, sp)] = flops / timeit(fn, args.iters) / 1e9
                    except Exception as e:  # unsupported config for this shape
                        res["c%d_s%d" % (cfg, sp)] = "err:%s" % str(e)[:40]
                    finally:
                        k._overrides.pop(kind, None)
            return res

        rec["fprop_tf"] = flops / timeit(lambda: k.conv3x3_fprop(x, w, b, True), args.iters) / 1e9
        # the training step masks data gradients with 1-bit masks packed by the producer's epilogue
        xbits = ((x.view(-1, 8) > 0).to(torch.int32) * (2 ** torch.arange(8, device="cuda", dtype=torch.int32))).sum(1).to(torch.uint8)
        rec["dgrad_tf"] = flops / timeit(lambda: k.conv3x3_dgrad(dy, wd, mask_src=x, mask_bits=xbits), args.iters) / 1e9
        rec["wgrad_tf"] = flops / timeit(lambda: k.conv3x3_wgrad(dy, x, gw), args.iters) / 1e9
        if args.sweep:
            rec["fprop_sweep"] = best("igemm", lambda: k.conv3x3_fprop(x, w, b, True), [0, 3, 8, 16], [1])
            rec["dgrad_sweep"] = best("igemm", lambda: k.conv3x3_dgrad(dy, wd, mask_src=x), [0, 3, 8, 16], [1])
            rec["wgrad_sweep"] = best("wgrad", lambda: k.conv3x3_wgrad(dy, x, gw), [2, 3, 8, 9], [4, 16, 64])
        if not args.no_stock:
            xs = x.permute(0, 3, 1, 2)          # NCHW view of NHWC memory == channels_last
            ws = w.view(Co, 3, 3, C).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
            bs = b.to(BF16)
            xs.requires_grad_(True)
            ws.requires_grad_(True)
            rec["stock_fprop_tf"] = flops / timeit(lambda: F.relu(F.conv2d(xs, ws, bs, padding=1)), args.iters) / 1e9
            ys = F.conv2d(xs, ws, bs, padding=1)
            dys = dy.permute(0, 3, 1, 2)
            rec["stock_dgrad_tf"] = flops / timeit(
                lambda: torch.autograd.grad(ys, xs, dys, retain_graph=True), args.iters) / 1e9
            rec["stock_wgrad_tf"] = flops / timeit(
                lambda: torch.autograd.grad(ys, ws, dys, retain_graph=True), args.iters) / 1e9
            del ys
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        del x, dy, w, wd, gw
        torch.cuda.empty_cache()
    for (M, K, Nout) in [(N, 25088, 4096), (N, 4096, 4096)]:
        x = torch.randn(M, K, device="cuda").to(BF16)
        w = (torch.randn(Nout, K, device="cuda") * K ** -0.5).to(BF16)
        dy = (torch.randn(M, Nout, device="cuda") * 0.1).to(BF16)
        wt = k.linear_weight_transpose(w)
        gw = torch.empty(Nout, K, device="cuda")
        flops = 2.0 * M * K * Nout
        rec = dict(shape="fc %dx%dx%d" % (M, K, Nout), gflop=flops / 1e9)
        rec["fwd_ms"] = timeit(lambda: k.linear_fwd(x, w, None, relu=True), args.iters)
        rec["dgrad_ms"] = timeit(lambda: k.linear_dgrad(dy, wt), args.iters)
        rec["wgrad_ms"] = timeit(lambda: k.linear_wgrad(dy, x, gw), args.iters)
        rec["transpose_ms"] = timeit(lambda: k.linear_weight_transpose(w, out=wt), args.iters)
        if not args.no_stock:
            rec["stock_fwd_ms"] = timeit(lambda: F.linear(x, w), args.iters)
            rec["stock_wgrad_ms"] = timeit(lambda: dy.t() @ x, args.iters)
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(dict(batch=N, rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
