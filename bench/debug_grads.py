"""Per-parameter gradient agreement of the GPU (HIP, bf16) path with the CPU fp32 reference."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.models.vgg16 import VGG16
from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy

torch.manual_seed(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cpu = VGG16(3, 3, True, dropout=0.0)
gpu = VGG16(3, 3, True, dropout=0.0)
gpu.load_state_dict(cpu.state_dict())
gpu = gpu.cuda()
torch.manual_seed(1)
x = torch.randn(8, 3, size, size)
y = torch.randint(0, 3, (8,))
crit = SoftmaxCrossEntropy()
oc = cpu(x); lc = crit(oc, y); lc.backward()
og = gpu(x.cuda()); lg = crit(og, y.cuda()); lg.backward()
torch.cuda.synchronize()
print("loss cpu %.6f gpu %.6f ; logits max err %.4g (scale %.4g)" % (lc.item(), lg.item(), (og.cpu() - oc).abs().max().item(), oc.abs().max().item()))
a = gpu.param_arena
gp = dict(gpu.named_parameters())
for name, pc in cpu.named_parameters():
    pg = gp[name]
    gg = a._logical_view(a.flat_grad, a.index[id(pg)]).float().cpu()
    gc = pc.grad
    cos = torch.nn.functional.cosine_similarity(gg.flatten(), gc.flatten(), dim=0).item()
    print("%-28s max|ref| %.3e  relmax %.4f  cos %.5f  norm ratio %.4f" % (
        name, gc.abs().max().item(), (gg - gc).abs().max().item() / (gc.abs().max().item() + 1e-20), cos,
        gg.norm().item() / (gc.norm().item() + 1e-20)))
