"""DEBUG: in-kernel time stamps of wgrad_tt2 (a -DDTP_WG2_STAMPS build of the kernel library, see
bench/r3_stamp_wg.sh): per workgroup the ticks spent in the K loop, and of those the ticks in the loop's
`s_waitcnt vmcnt(0)` (own LDS-DMA of the next tile not landed yet) and in its barrier (waiting for the other
waves).  s_memtime ticks are 10 ns."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BF16 = torch.bfloat16
N_IMG, HW, C, Co = int(os.environ.get("N", "256")), int(os.environ.get("HW", "56")), int(os.environ.get("C", "256")), int(os.environ.get("CO", "256"))
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
os.environ["DTP_WG2_STAMPS"] = hex(stamps.data_ptr())
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402

x = torch.randn(N_IMG, HW, HW, C, device="cuda").to(BF16)
dy = (torch.randn(N_IMG, HW, HW, Co, device="cuda") * 0.1).to(BF16)
gw = torch.empty(Co, 9 * C, device="cuda")
for _ in range(3):
    k.conv3x3_wgrad(dy, x, gw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
k.conv3x3_wgrad(dy, x, gw)
e1.record()
torch.cuda.synchronize()
st = stamps.view(-1, 8).cpu()
st = st[st[:, 3] > 0].double()
tick = 0.01
loop, vm, bar, ks = st[:, 0] * tick, st[:, 1] * tick, st[:, 2] * tick, st[:, 3]
print("%dx%d@%d batch %d: kernel + reduce %.1f us, %d workgroups, %.0f K-steps each" % (C, Co, HW, N_IMG, e0.elapsed_time(e1) * 1e3, st.shape[0], ks.mean()))
print("K loop            mean %.1f us (p10 %.1f, p90 %.1f) = %.3f us per K-step" % (loop.mean(), loop.quantile(0.1), loop.quantile(0.9), (loop / ks).mean()))
print("  in vmcnt(0)     mean %.1f us = %.1f %% of the loop (p90 %.1f %%)" % (vm.mean(), 100 * (vm / loop).mean(), 100 * (vm / loop).quantile(0.9)))
print("  in the barrier  mean %.1f us = %.1f %% of the loop (p90 %.1f %%)" % (bar.mean(), 100 * (bar / loop).mean(), 100 * (bar / loop).quantile(0.9)))
span = (st[:, 4].max() - st[:, 4].min()) * tick
print("entry-time spread of the workgroups %.1f us; prologue %.2f us" % (span, ((st[:, 5] - st[:, 4]) * tick).mean()))
