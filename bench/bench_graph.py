"""Eager vs hipGraph-captured train step at small per-GPU batches (launch-bound regime; the
reference's default is 4 images per GPU, main.py:15 / run.sh:10)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.engine.graph import GraphedTrainStep  # noqa: E402
from distributed_training_pytorch_amd.models.vgg16 import VGG16  # noqa: E402
from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy  # noqa: E402
from distributed_training_pytorch_amd.ops.optim import FusedSGD  # noqa: E402

out = []
for B in [int(b) for b in (sys.argv[1:] or ["4", "8", "16", "32", "64"])]:
    for graphed in (False, True):
        torch.manual_seed(0)
        model = VGG16(3, 3, True, grad_dtype=torch.bfloat16).to("cuda").train()
        opt = FusedSGD(model, lr=0.001, momentum=0.9, weight_decay=1e-4)
        step = GraphedTrainStep(model, SoftmaxCrossEntropy(), opt, warmup=2, enabled=graphed)
        x = torch.randn(B, 3, 224, 224, device="cuda")
        y = torch.randint(0, 3, (B,), device="cuda")
        for _ in range(5):
            step(x, y)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step(x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        rec = dict(batch=B, hip_graph=graphed, ms_per_step=1e3 * dt, images_per_sec=B / dt, loss=float(loss))
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del model, opt, step
