"""Uninitialised-read screen: before every step the caching allocator's free blocks are filled with
NaN patterns (large and small pools), so any kernel that consumes memory it (or its producer) never
wrote turns the loss / gradients into NaN.  Reports the first module whose output contains NaN."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.models.vgg16 import VGG16  # noqa: E402
from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy  # noqa: E402
from distributed_training_pytorch_amd.ops.optim import FusedSGD  # noqa: E402
from distributed_training_pytorch_amd import _lib  # noqa: E402


def poison(mb_large=2048, n_small=4000):
    big = torch.full((mb_large * 1024 * 1024 // 4,), float("nan"), device="cuda")
    mids = [torch.full((s // 4,), float("nan"), device="cuda") for s in (1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20) for _ in range(8)]
    small = [torch.full((16 * 1024,), float("nan"), device="cuda") for _ in range(n_small)]
    torch.cuda.synchronize()
    del big, mids, small


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--poison-workspaces", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(22)
    m = VGG16(3, 3, True, dropout=0.0).to("cuda").train()
    o = FusedSGD(m, lr=0.002, momentum=0.9, weight_decay=1e-4)
    c = SoftmaxCrossEntropy()
    a = m.param_arena
    first_nan = []

    def hook(mod, inp, out):
        if isinstance(out, torch.Tensor) and out.is_floating_point() and not first_nan:
            if bool(torch.isnan(out.float()).any()):
                first_nan.append(type(mod).__name__ + ":" + str(tuple(out.shape)))

    for mod in m.modules():
        mod.register_forward_hook(hook)
    torch.manual_seed(21)
    for s in range(args.steps):
        x = torch.randn(args.batch, 3, args.size, args.size, device="cuda")
        y = torch.randint(0, 3, (args.batch,), device="cuda")
        poison()
        if args.poison_workspaces:
            for k, t in _lib.workspace._bufs.items():
                if k[0] != "zero_page" and t.is_floating_point():
                    t.fill_(float("nan"))
        o.zero_grad()
        loss = c(m(x), y)
        loss.backward()
        torch.cuda.synchronize()
        bad = [n for i, (n, p) in enumerate(zip(a.names, a.params))
               if bool(torch.isnan(a.flat_grad[a.offsets[i]:a.offsets[i] + p.numel()].float()).any())]
        print("step %d loss %.6f first-NaN-forward %s NaN-grads %s" % (s, float(loss), first_nan[:1], bad[:6]), flush=True)
        o.step()
        if bad or first_nan or loss != loss:
            break
    print("poison probe done", flush=True)


if __name__ == "__main__":
    main()
