"""Repeat the conv_c64 kernels on fixed inputs and compare every output bitwise with the first one
(optionally with unrelated kernels running on a second stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402

BF16 = torch.bfloat16
torch.manual_seed(5)
REPS = int(os.environ.get("REPS", "3000"))
side = torch.cuda.Stream()
noise_src = torch.randn(512, 4608, device="cuda").to(BF16)


def run(name, fn, concurrent):
    ref = None
    bad = 0
    for i in range(REPS):
        if concurrent and i % 4 == 0:
            with torch.cuda.stream(side):
                k.conv_weight_to_dgrad(noise_src, 512, 512)
        out = fn()
        outs = out if isinstance(out, tuple) else (out,)
        if ref is None:
            ref = [o.clone() for o in outs]
        else:
            for o, r in zip(outs, ref):
                if not torch.equal(o, r):
                    bad += 1
                    if bad <= 3:
                        d = (o.float() - r.float()).abs()
                        idx = (d > 0).nonzero()
                        print("  %s rep %d: %d elements differ, max %.3g, first idx %s last idx %s" % (
                            name, i, idx.shape[0], float(d.max()), idx[0].tolist(), idx[-1].tolist()), flush=True)
                    break
    torch.cuda.synchronize()
    print("%s concurrent=%s: %d / %d repetitions differ" % (name, concurrent, bad, REPS - 1), flush=True)


for (N, H, W) in [(4, 32, 32), (4, 16, 16), (8, 64, 64)]:
    x = torch.randn(N, H, W, 64, device="cuda").to(BF16)
    w64 = (torch.randn(64, 576, device="cuda") * 0.06).to(BF16)
    w128 = (torch.randn(128, 576, device="cuda") * 0.06).to(BF16)
    b64, b128 = torch.randn(64, device="cuda"), torch.randn(128, device="cuda")
    for conc in (False, True):
        run("pool64 %s" % ((N, H, W),), lambda: k.conv3x3_relu_pool_fprop(x, w64, b64), conc)
        run("pool128 %s" % ((N, H, W),), lambda: k.conv3x3_relu_pool_fprop(x, w128, b128), conc)
        run("plain64 %s" % ((N, H, W),), lambda: k.conv3x3_fprop(x, w64, b64, True), conc)
        run("plain128 %s" % ((N, H, W),), lambda: k.conv3x3_fprop(x, w128, b128, True, want_bits=True), conc)
