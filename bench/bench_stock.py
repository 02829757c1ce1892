"""Stock PyTorch-ROCm VGG16 training throughput on one MI355X — the only honest comparison point
(the reference publishes no numbers).  Plain torch.nn modules
(MIOpen convs, hipBLASLt GEMMs, torch.optim.SGD), fp32 or bf16-autocast + channels_last,
synthetic data resident on the device, full step (zero_grad, fwd, loss, bwd, optimizer step).
"""
import argparse
import json
import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


def make_vgg16(num_classes=3, batch_norm=False):
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    layers, c = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(c, v, 3, padding=1)] + ([nn.BatchNorm2d(v)] if batch_norm else []) + [nn.ReLU(True)]
            c = v
    return nn.Sequential(*layers, nn.AdaptiveAvgPool2d((7, 7)), nn.Flatten(),
                         nn.Linear(25088, 4096), nn.ReLU(True), nn.Dropout(0.3),
                         nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(0.3),
                         nn.Linear(4096, num_classes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch-norm", action="store_true", help="VGG16-BN (BASELINE.json config 5 comparison)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = make_vgg16(batch_norm=args.batch_norm).to(dev)
    if args.dtype == "bf16":
        model = model.to(memory_format=torch.channels_last)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    x = torch.randn(args.batch, 3, 224, 224, device=dev)
    if args.dtype == "bf16":
        x = x.contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 3, (args.batch,), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            out = model(x)
            loss = F.cross_entropy(out.float(), y)
        loss.backward()
        opt.step()
        return loss

    # MIOpen's first-call kernel search at a new batch size can take minutes: keep a heartbeat
    import sys
    import threading
    done = threading.Event()

    def beat():
        t = time.time()
        while not done.wait(30):
            print("[bench_stock] warming up (MIOpen find) %.0fs" % (time.time() - t), file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    done.set()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = dict(impl="stock-pytorch-rocm", dtype=args.dtype, batch=args.batch, steps=args.steps, batch_norm=args.batch_norm,
               ms_per_step=1e3 * dt / args.steps, images_per_sec=args.batch * args.steps / dt,
               loss=float(loss))
    print(json.dumps(rec), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "a") as f:
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
