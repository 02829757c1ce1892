"""Per-step loss / gradient health of the bench configuration (debug aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.data.synthetic import DeviceBatchSource  # noqa: E402
from distributed_training_pytorch_amd.models.vgg16 import VGG16  # noqa: E402
from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy  # noqa: E402
from distributed_training_pytorch_amd.ops.optim import FusedSGD  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
gd = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = VGG16(3, 3, True, grad_dtype=gd).to(dev).train()
opt = FusedSGD(model, lr=lr, momentum=0.9, weight_decay=1e-4)
crit = SoftmaxCrossEntropy()
data = DeviceBatchSource(B, 3, 224, 224, device=dev, n_distinct=2, seed=0)
a = model.param_arena
for it in range(14):
    x, y = next(data)
    opt.zero_grad()
    out = model(x)
    loss = crit(out, y)
    loss.backward()
    torch.cuda.synchronize()
    g = a.flat_grad.float()
    rows = []
    for i, n in enumerate(a.names):
        s = slice(a.offsets[i], a.offsets[i] + a.params[i].numel())
        gi = g[s]
        if not torch.isfinite(gi).all() or it < 2 or gi.abs().max() > 1e3:
            rows.append("%s:%.2e%s" % (n.replace("backbone.", "b").replace(".conv", ""), gi.abs().max().item(),
                                       "" if torch.isfinite(gi).all() else "(NONFINITE %d)" % int((~torch.isfinite(gi)).sum())))
    print("it %d loss %.4f logits absmax %.3e gradnorm %.3e wmax %.3e | %s" % (
        it, float(loss), out.float().abs().max().item(), g.norm().item(), a.flat_master.abs().max().item(),
        " ".join(rows[:40])), flush=True)
    opt.step()
