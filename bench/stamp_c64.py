"""DEBUG: in-kernel time stamps of conv_c64 (register-resident-filter 3x3 convolution): where a row
step's time goes.  Stamps per workgroup and row step (first 16 steps): 0 loop top, 1 after the
first 12 MFMAs, 2 after the barrier, 3 after the 36 MFMAs (deferred epilogue of the previous row included),
4 row rounded and parked in LDS."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BF16 = torch.bfloat16
N, H, W, O = int(os.environ.get("N", "64")), 224, 224, int(os.environ.get("O", "64"))
MODE = os.environ.get("MODE", "dgrad")
S = (W + 31) // 32
units = N * S
rc = max(1, min(H // 16, (16 * 256 + units - 1) // units))
rpc = (H + rc - 1) // rc
if MODE == "pool" and rpc & 1:
    rpc += 1
RC = (H + rpc - 1) // rpc
grid = units * RC
stamps = torch.zeros(grid * 16 * 6, dtype=torch.int64, device="cuda")
os.environ["DTP_C64_STAMPS"] = hex(stamps.data_ptr())
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402

x = torch.randn(N, H, W, 64, device="cuda").to(BF16)
w = (torch.randn(O, 9 * 64, device="cuda") * 0.05).to(BF16)
b = torch.zeros(O, device="cuda")
for _ in range(3):
    if MODE == "dgrad":
        k.conv3x3_dgrad(x, w)
    elif MODE == "pool":
        k.conv3x3_relu_pool_fprop(x, w, b)
    else:
        k.conv3x3_fprop(x, w, b, True)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5):
    if MODE == "dgrad":
        k.conv3x3_dgrad(x, w)
    elif MODE == "pool":
        k.conv3x3_relu_pool_fprop(x, w, b)
    else:
        k.conv3x3_fprop(x, w, b, True)
t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / 5
print("%s N=%d O=%d grid=%d rows/chunk=%d  %.3f ms  %.0f TF/s" % (MODE, N, O, grid, rpc, ms, 2.0 * N * H * W * O * 576 / ms / 1e9))
st = stamps.view(grid, 16, 6).cpu().double()
ok = st[:, 4:12, :]          # steady-state steps
names = ["12 MFMAs + reads (to the barrier)", "vmcnt(0) + barrier", "LDS-DMA issue + 24 MFMAs + deferred epilogue", "round / ReLU / park"]
step = (ok[:, 1:, 0] - ok[:, :-1, 0])
print("cycles per row step: mean %.0f p10 %.0f p90 %.0f" % (step.mean(), step.quantile(0.1), step.quantile(0.9)))
for i in range(4):
    d = ok[:, :, i + 1] - ok[:, :, i]
    print("  %-46s mean %.0f  p10 %.0f  p90 %.0f" % (names[i], d.mean(), d.quantile(0.1), d.quantile(0.9)))
d = ok[:, 1:, 0] - ok[:, :-1, 4]
print("  %-46s mean %.0f" % ("end -> next top", d.mean()))
