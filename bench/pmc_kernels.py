"""Tiny driver for rocprofv3 --pmc passes: launches each hot kernel a few times on VGG16 shapes
(random data) so that counters can be read per kernel.  Run as
  rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d gpurun_out/pmcN -- python3 bench/pmc_kernels.py
(one counter set per run; see profiles/README.md for the sets used)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402

BF16 = torch.bfloat16
N = int(os.environ.get("PMC_BATCH", "64"))
POOL = os.environ.get("PMC_POOL", "1") == "1"
torch.manual_seed(0)


def conv_set(C, Co, HW):
    x = torch.randn(N, HW, HW, C, device="cuda").to(BF16)
    dy = (torch.randn(N, HW, HW, Co, device="cuda") * 0.1).to(BF16)
    w = (torch.randn(Co, 9 * C, device="cuda") * (2.0 / (9 * C)) ** 0.5).to(BF16)
    b = torch.zeros(Co, device="cuda")
    wd = k.conv_weight_to_dgrad(w, Co, C)
    gw = torch.empty(Co, 9 * C, device="cuda")
    for _ in range(3):
        k.conv3x3_fprop(x, w, b, True)
        k.conv3x3_dgrad(dy, wd, mask_src=x)
        k.conv3x3_wgrad(dy, x, gw)
        if POOL and k.conv_pool_fusable(C, Co, HW, HW, N=N):     # round 3: conv + ReLU + max-pool in one kernel
            k.conv3x3_relu_pool_fprop(x, w, b)
    torch.cuda.synchronize()


conv_set(256, 256, 56)    # 8-phase igemm (fprop/dgrad) + 256x256 TT wgrad
conv_set(64, 64, 224)     # 256x64 igemm + column-strip wgrad_c64
conv_set(128, 128, 112)   # 256x128 pipelined igemm + 128x128 TT wgrad
conv_set(64, 128, 112)    # conv_c64<128> fprop, 256x64 igemm dgrad, wgrad_c64<128>
conv_set(512, 512, 28)    # 8-phase igemm + 256x256 TT wgrad, deep K
x = torch.randn(N, 3, 224, 224, device="cuda")
w1 = (torch.randn(64, 27, device="cuda") * 0.27).to(BF16)
y = k.conv1_fprop(x, w1, torch.zeros(64, device="cuda"))
k.maxpool2x2_bwd(y, k.maxpool2x2_fwd(y), relu_mask=True)
# the kernels the training step really uses at the first layer / the pools (round 2)
dy1 = (torch.randn(N, 224, 224, 64, device="cuda") * 0.1).to(BF16)
gw1, gb1 = torch.empty(64, 27, device="cuda"), torch.empty(64, device="cuda")
k.conv1_wgrad(x, dy1, gw1, gb1, mask_w=w1, mask_bias=torch.zeros(64, device="cuda"))
yp, idx = k.maxpool2x2_fwd(y, want_idx=True)
k.maxpool2x2_bwd(None, torch.randn_like(yp), relu_mask=True, idx=idx, in_shape=tuple(y.shape))
# VGG16-BN block end (config 5): BatchNorm + ReLU + pool as one kernel pair
g, b = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
rm, rv = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
zp, codes, stats = k.bn_fwd_train_pool(y, g, b, rm, rv)
k.bn_bwd_pool(y, torch.randn_like(zp), codes, g, stats, torch.empty(64, device="cuda"), torch.empty(64, device="cuda"))
torch.cuda.synchronize()
print("pmc driver done")
