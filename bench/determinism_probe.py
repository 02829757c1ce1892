"""Run the same tiny VGG16 training sequence many times (eager, and optionally through the hipGraph
step) and report any run whose gradients/parameters differ bitwise from the first one, with the
first step and the first parameter tensor that diverged."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_training_pytorch_amd.models.vgg16 import VGG16  # noqa: E402
from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy  # noqa: E402
from distributed_training_pytorch_amd.ops.optim import FusedSGD  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=30)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--bn", action="store_true")
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--adamw", action="store_true")
    ap.add_argument("--grad-bf16", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.0)
    args = ap.parse_args()
    torch.manual_seed(21)
    xs = [torch.randn(args.batch, 3, args.size, args.size, device="cuda") for _ in range(args.steps)]
    ys = [torch.randint(0, 3, (args.batch,), device="cuda") for _ in range(args.steps)]
    ref = None
    bad = 0
    for t in range(args.trials):
        torch.manual_seed(22)
        m = VGG16(3, 3, True, dropout=args.dropout, batch_norm=args.bn, fp8_classifier=args.fp8,
                  grad_dtype=torch.bfloat16 if args.grad_bf16 else torch.float32).to("cuda").train()
        if args.adamw:
            from distributed_training_pytorch_amd.ops.optim import FusedAdamW
            o = FusedAdamW(m, lr=1e-4, weight_decay=1e-2)
        else:
            o = FusedSGD(m, lr=0.002, momentum=0.9, weight_decay=1e-4)
        c = SoftmaxCrossEntropy()
        a = m.param_arena
        g = None
        if args.graph:
            from distributed_training_pytorch_amd.engine.graph import GraphedTrainStep
            g = GraphedTrainStep(m, c, o, warmup=2)
        trace = []
        acts = []
        names = {mod: n for n, mod in m.named_modules()}

        def hook(mod, inp, out):
            if isinstance(out, torch.Tensor):
                acts[-1].append((names[mod], out.detach().clone()))

        if g is None:
            for mod in m.modules():
                if mod is not m:
                    mod.register_forward_hook(hook)     # leaves AND blocks (a fused block skips its leaves' hooks)
        for x, y in zip(xs, ys):
            acts.append([])
            if g is not None:
                g(x, y)
            else:
                o.zero_grad()
                c(m(x), y).backward()
                o.step()
            torch.cuda.synchronize()
            trace.append((a.flat_grad.clone(), a.flat_master.clone()))
        if t % 25 == 0:
            print("trial %d ..." % t, flush=True)
        if ref is None:
            ref, ref_acts = trace, acts
            continue
        for s, ((g0, p0), (g1, p1)) in enumerate(zip(ref, trace)):
            if not torch.equal(g0, g1) or not torch.equal(p0, p1):
                bad += 1
                for (n0, t0), (n1, t1) in zip(ref_acts[s], acts[s]):
                    if not torch.equal(t0, t1):
                        d = (t0.float() - t1.float()).abs()
                        print("  first differing forward output: %s %s n=%d max=%.3g" % (n0, tuple(t0.shape), int((d > 0).sum()),
                                                                                 float(d.max())), flush=True)
                        if d.dim() == 4:
                            idx = (d > 0).nonzero()
                            print("    images %s rows %s cols %s channels %d..%d ; ref-vs-new sample %s / %s" % (
                                sorted(set(idx[:, 0].tolist())), sorted(set(idx[:, 1].tolist())),
                                sorted(set(idx[:, 2].tolist())), int(idx[:, 3].min()), int(idx[:, 3].max()),
                                t0[tuple(idx[0].tolist())].item(), t1[tuple(idx[0].tolist())].item()), flush=True)
                        break
                else:
                    print("  forward outputs identical: divergence starts in backward", flush=True)
                which = []
                for i, (n, p) in enumerate(zip(a.names, a.params)):
                    sl = slice(a.offsets[i], a.offsets[i] + p.numel())
                    if not torch.equal(g0[sl], g1[sl]):
                        d = (g0[sl].float() - g1[sl].float()).abs()
                        which.append("%s(n=%d,max=%.3g)" % (n, int((d > 0).sum()), float(d.max())))
                print("trial %d diverged at step %d: grads differ in %s" % (t, s, ", ".join(which[:8]) or "none (params only)"),
                      flush=True)
                break
    print("determinism probe: %d of %d trials diverged (graph=%s)" % (bad, args.trials - 1, args.graph), flush=True)


if __name__ == "__main__":
    main()
