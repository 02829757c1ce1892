"""Per-layer gradient parity of the fused GPU VGG16 against the bf16-rounding-point emulated fp32
reference (tests/test_gpu_model.py::_emulated_reference): prints cosine and norm ratio per parameter
for both gradient dtypes and input sizes — the measured values behind the per-layer thresholds of
test_forward_backward_matches_reference."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_model as T  # noqa: E402
from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy  # noqa: E402

cos = torch.nn.functional.cosine_similarity
worst = {}
for grad_dtype in (torch.float32, torch.bfloat16):
    for size in (64, 96):
        for seed in (0, 1, 2):
            cpu, gpu = T._models(grad_dtype=grad_dtype, seed=seed)
            torch.manual_seed(1 + seed)
            x = torch.randn(8, 3, size, size)
            y = torch.randint(0, 3, (8,))
            crit = SoftmaxCrossEntropy()
            gpu.train()
            crit(gpu(x.cuda()), y.cuda()).backward()
            torch.cuda.synchronize()
            _, _, ref = T._emulated_reference(cpu.state_dict(), x.cuda(), y.cuda())
            a = gpu.param_arena
            gp = dict(gpu.named_parameters())
            for name in gp:
                gg = a._logical_view(a.flat_grad, a.index[id(gp[name])]).float()
                ge = ref[name]
                c = cos(gg.flatten(), ge.flatten(), dim=0).item()
                r = gg.norm().item() / (ge.norm().item() + 1e-20)
                w = worst.setdefault(name, dict(cos=1.0, rlo=9.0, rhi=0.0))
                w["cos"], w["rlo"], w["rhi"] = min(w["cos"], c), min(w["rlo"], r), max(w["rhi"], r)
for name, w in worst.items():
    print(json.dumps(dict(param=name, min_cos=round(w["cos"], 5), ratio_lo=round(w["rlo"], 4), ratio_hi=round(w["rhi"], 4))))
