"""conv_c64 kernels in isolation at batch 256: conv1_2 forward(+pool) / data gradient, conv2_1 forward."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.ops import kernels as k  # noqa: E402
from bench_igemm17 import timeit  # noqa: E402
BF16 = torch.bfloat16
N = int(os.environ.get("B", "256"))
x = torch.randn(N, 224, 224, 64, device="cuda").to(BF16)
w64 = (torch.randn(64, 576, device="cuda") * 0.06).to(BF16)
b64 = torch.randn(64, device="cuda")
fl = 2.0 * N * 224 * 224 * 64 * 64 * 9
for name, fn in (("conv1_2 fwd+pool", lambda: k.conv3x3_relu_pool_fprop(x, w64, b64)),
                 ("conv1_2 fwd", lambda: k.conv3x3_fprop(x, w64, b64, True)),
                 ("conv1_2 dgrad", lambda: k.conv3x3_dgrad(x, w64))):
    ms = min(timeit(fn, iters=6) for _ in range(3))
    print(json.dumps(dict(kernel=name, ms=round(ms, 4), tf=round(fl / ms / 1e9, 1))), flush=True)
del x
x2 = torch.randn(N, 112, 112, 64, device="cuda").to(BF16)
w128 = (torch.randn(128, 576, device="cuda") * 0.06).to(BF16)
b128 = torch.randn(128, device="cuda")
fl2 = 2.0 * N * 112 * 112 * 64 * 128 * 9
ms = min(timeit(lambda: k.conv3x3_fprop(x2, w128, b128, True, want_bits=True), iters=6) for _ in range(3))
print(json.dumps(dict(kernel="conv2_1 fwd", ms=round(ms, 4), tf=round(fl2 / ms / 1e9, 1))), flush=True)
