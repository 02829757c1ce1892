"""The 8-phase 256x256 weight-gradient kernel (wgrad2.hip cfg 10) against the 2-stage one (cfg 8): same sums in the
same order, so the results must be bit-identical — conv shapes incl. K tails, odd K-tile counts and split-K, and
the linear (taps == 1) form incl. M / N tails."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as K  # noqa: E402

BF16 = torch.bfloat16
torch.manual_seed(0)
ok = True
for (N, H, W, C, Co, ns) in [(2, 28, 28, 256, 256, 1), (3, 14, 14, 256, 512, 1), (5, 14, 14, 512, 256, 3), (1, 8, 8, 256, 256, 1),
                             (4, 56, 56, 256, 256, 7), (2, 10, 6, 256, 304, 2)]:
    x = torch.randn(N, H, W, C, device="cuda").to(BF16)
    dy = (torch.randn(N, H, W, Co, device="cuda") * 0.1).to(BF16)
    outs = []
    for cfg in (8, 10):
        K._overrides["wgrad"] = lambda M, C_, taps, ks, cfg=cfg, ns=ns: (cfg, ns)
        gw = torch.full((Co, 9 * C), 3.0, device="cuda")
        K.conv3x3_wgrad(dy, x, gw)
        torch.cuda.synchronize()
        outs.append(gw.clone())
    same = torch.equal(outs[0], outs[1])
    print("conv N%d %dx%d C%d Co%d split %d: %s (max |diff| %.3g)" % (N, H, W, C, Co, ns, "bit-identical" if same else "DIFFERENT",
                                                                     (outs[0] - outs[1]).abs().max().item()), flush=True)
    ok &= same
for (B, Nout, Kin, ns) in [(1024, 4096, 4096, 1), (200, 300, 1000, 1), (64, 256, 25088, 2)]:
    x = torch.randn(B, Kin, device="cuda").to(BF16)
    ld = (Nout + 63) // 64 * 64
    dy = torch.zeros(B, ld, device="cuda", dtype=BF16)
    dy[:, :Nout] = (torch.randn(B, Nout, device="cuda") * 0.1).to(BF16)
    outs = []
    for cfg in (8, 10):
        K._overrides["wgrad"] = lambda M, C_, taps, ks, cfg=cfg, ns=ns: (cfg, ns)
        gw = torch.full((Nout, Kin), 3.0, device="cuda")
        K.linear_wgrad(dy, x, gw, n_valid=Nout)
        torch.cuda.synchronize()
        outs.append(gw.clone())
    same = torch.equal(outs[0], outs[1])
    print("linear B%d Nout%d Kin%d split %d: %s (max |diff| %.3g)" % (B, Nout, Kin, ns, "bit-identical" if same else "DIFFERENT",
                                                                     (outs[0] - outs[1]).abs().max().item()), flush=True)
    ok &= same
K._overrides.pop("wgrad", None)
sys.exit(0 if ok else 1)
