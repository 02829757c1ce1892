import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k
import torch.nn.functional as F
BF16 = torch.bfloat16
torch.manual_seed(0)
N, H, W, C, Co = 5, 7, 9, 64, 512
x = torch.randn(N, H, W, C, device="cuda").to(BF16)
w = (torch.randn(Co, 9 * C, device="cuda") * (2.0 / (9 * C)) ** 0.5).to(BF16)
b = torch.randn(Co, device="cuda")
for relu, bias in ((True, b), (False, None), (False, b), (True, None)):
    y = k.conv3x3_fprop(x, w, bias, relu=relu).float()
    w4 = w.float().view(Co, 3, 3, C).permute(0, 3, 1, 2)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w4, bias, padding=1).permute(0, 2, 3, 1)
    if relu:
        ref = ref.relu()
    d = (y - ref).abs() > 0.05 + 0.02 * ref.abs()
    idx = d.view(-1, Co).nonzero()
    print("relu", relu, "bias", bias is not None, "bad", int(d.sum()), "of", d.numel(),
          "rows", sorted(set((idx[:, 0] // 1).tolist()))[:6], "chan mod 16", sorted(set((idx[:, 1] % 16).tolist())),
          "chan range", (int(idx[:, 1].min()), int(idx[:, 1].max())) if idx.numel() else None)
    if idx.numel():
        r, c = idx[0].tolist()
        print("   sample y", y.view(-1, Co)[r, c].item(), "ref", ref.reshape(-1, Co)[r, c].item(), "bias", b[c].item())
