"""Host-side enqueue time per training step vs GPU step time (is the step launch-bound?)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.data.synthetic import DeviceBatchSource  # noqa: E402
from distributed_training_pytorch_amd.models.vgg16 import VGG16  # noqa: E402
from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy  # noqa: E402
from distributed_training_pytorch_amd.ops.optim import FusedSGD  # noqa: E402
from distributed_training_pytorch_amd.parallel.data_parallel import DataParallel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = VGG16(3, 3, True, grad_dtype=torch.bfloat16).to("cuda")
opt = FusedSGD(model, lr=0.001, momentum=0.9, weight_decay=1e-4)
crit = SoftmaxCrossEntropy()
dp = DataParallel(model).train()
data = DeviceBatchSource(B, 3, 224, 224, device=torch.device("cuda"), n_distinct=2, seed=0)


def step():
    x, y = next(data)
    opt.zero_grad()
    loss = crit(dp(x), y)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(20):
    h0 = time.perf_counter()
    step()
    host.append(time.perf_counter() - h0)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 20
host.sort()
print("batch %d: wall %.2f ms/step, host enqueue median %.2f ms (min %.2f, max %.2f)" % (
    B, 1e3 * total, 1e3 * host[len(host) // 2], 1e3 * host[0], 1e3 * host[-1]))
