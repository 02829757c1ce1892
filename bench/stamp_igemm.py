"""DEBUG: in-kernel time stamps of the 8-phase implicit GEMM (entry / loop start / loop end / stores
issued) per workgroup, grouped by CU to see the gap between consecutive workgroups on one CU."""
import os
import sys
import collections

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BF16 = torch.bfloat16
N_IMG, HW, C, Co = int(os.environ.get("N", "256")), int(os.environ.get("HW", "56")), int(os.environ.get("C", "256")), int(os.environ.get("CO", "256"))
CFG = int(os.environ.get("CFG", "16"))
BM, BN = {17: (512, 128), 19: (384, 128)}.get(CFG, (256, 256))
M = N_IMG * HW * HW
tiles = (M + BM - 1) // BM * ((Co + BN - 1) // BN)
stamps = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
os.environ["DTP_IGEMM_STAMPS"] = hex(stamps.data_ptr())
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402

x = torch.randn(N_IMG, HW, HW, C, device="cuda").to(BF16)
w = (torch.randn(Co, 9 * C, device="cuda") * (2.0 / (9 * C)) ** 0.5).to(BF16)
b = torch.zeros(Co, device="cuda")
k._overrides["igemm"] = lambda M_, Nc, ks: (CFG, 1)
MODE = os.environ.get("MODE", "fprop")
if MODE == "fprop":
    for _ in range(3):
        k.conv3x3_fprop(x, w, b, True)
else:
    # data gradient of a Co -> C convolution: output has C channels... use C == Co shapes
    dy = (torch.randn(N_IMG, HW, HW, Co, device="cuda") * 0.1).to(BF16)
    wd = k.conv_weight_to_dgrad(w, Co, C)
    ymask, bits = k.conv3x3_fprop(x, (torch.randn(C, 9 * C, device="cuda") * 0.03).to(BF16), None, True, want_bits=True) if C == Co else (None, None)
    for _ in range(3):
        k.conv3x3_dgrad(dy, wd, mask_src=ymask, mask_bits=bits if MODE == "dgrad_bits" else None)
torch.cuda.synchronize()
st = stamps.view(tiles, 8).cpu()
hw = st[:, 0]
t = st[:, 1:5].double()
t0 = t[:, 0].min()
t = (t - t0)          # s_memtime ticks (100 MHz "REFCLK" on gfx9: 10 ns per tick)
tick_us = float(os.environ.get("TICK_US", "0.01"))
pro, loop, epi = (t[:, 1] - t[:, 0]) * tick_us, (t[:, 2] - t[:, 1]) * tick_us, (t[:, 3] - t[:, 2]) * tick_us
print("workgroups %d  kernel span %.1f us" % (tiles, float(t[:, 3].max()) * tick_us))
print("prologue  mean %.2f us  (p10 %.2f p90 %.2f)" % (pro.mean(), pro.quantile(0.1), pro.quantile(0.9)))
print("main loop mean %.2f us  (p10 %.2f p90 %.2f)" % (loop.mean(), loop.quantile(0.1), loop.quantile(0.9)))
print("epilogue  mean %.2f us  (p10 %.2f p90 %.2f)" % (epi.mean(), epi.quantile(0.1), epi.quantile(0.9)))
t5, t6 = (st[:, 5].double() - t0) , (st[:, 6].double() - t0)
print("  loop end -> drained + barrier %.2f us ; C image written + barrier %.2f us ; store loop %.2f us" % (
    float(((t5 - t[:, 2]) * tick_us).mean()), float(((t6 - t5) * tick_us).mean()), float(((t[:, 3] - t6) * tick_us).mean())))
by_cu = collections.defaultdict(list)
for i in range(tiles):
    h = int(hw[i])
    key = (h >> 32, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15)      # xcc, se, sh, cu
    by_cu[key].append((float(t[i, 0]), float(t[i, 3])))
gaps = []
for key, lst in by_cu.items():
    lst.sort()
    for (a0, a1), (b0, b1) in zip(lst, lst[1:]):
        gaps.append((b0 - a1) * tick_us)
g = torch.tensor(gaps)
print("distinct CUs %d ; gap between a workgroup's last stamp and the next workgroup's entry on the same CU: mean %.2f us (p10 %.2f p90 %.2f)"
      % (len(by_cu), g.mean(), g.quantile(0.1), g.quantile(0.9)))
