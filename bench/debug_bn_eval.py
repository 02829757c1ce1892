import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.models.vgg16 import vgg16_bn
from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
from distributed_training_pytorch_amd.ops.optim import FusedSGD
from distributed_training_pytorch_amd.data.synthetic import SyntheticImageDataset
torch.manual_seed(5)
fp8 = os.environ.get("FP8", "1") == "1"
m = vgg16_bn(3, 3, fp8_classifier=fp8, grad_dtype=torch.bfloat16).cuda()
opt = FusedSGD(m, lr=float(os.environ.get("LR", "0.02")), momentum=0.9, weight_decay=1e-4)
crit = SoftmaxCrossEntropy()
tr = SyntheticImageDataset(192, 3, 64, 64, seed=0)
va = SyntheticImageDataset(96, 3, 64, 64, seed=2)
X = torch.stack([tr[i][0] for i in range(192)]).cuda(); Y = torch.tensor([tr[i][1] for i in range(192)]).cuda()
XV = torch.stack([va[i][0] for i in range(96)]).cuda(); YV = torch.tensor([va[i][1] for i in range(96)]).cuda()
for ep in range(int(os.environ.get("EPOCHS", "8"))):
    m.train()
    perm = torch.randperm(192, device="cuda")
    for s in range(0, 192, 32):
        idx = perm[s:s + 32]
        opt.zero_grad(); loss = crit(m(X[idx]), Y[idx]); loss.backward(); opt.step()
    m.eval()
    with torch.no_grad():
        acc_e = (m(XV).argmax(-1) == YV).float().mean().item()
        acc_tr_e = (m(X).argmax(-1) == Y).float().mean().item()
    acc_b = float("nan")
    if os.environ.get("NOVALTRAIN", "0") != "1":
        m.train()
        with torch.no_grad():
            acc_b = (m(XV).argmax(-1) == YV).float().mean().item()   # batch statistics (also moves running stats)
    print("epoch %d loss %.4f  eval-mode val acc %.3f train acc %.3f | batch-stat val acc %.3f" % (ep, float(loss), acc_e, acc_tr_e, acc_b), flush=True)
