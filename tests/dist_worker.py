"""Worker for the multi-process CPU/gloo tests (launched by tests/test_dist_gloo.py through
``python -m torch.distributed.run``).  Each scenario asserts inside the ranks and writes
``ok.<rank>`` files; a failing rank exits non-zero, which fails the launcher."""
import json
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributed_training_pytorch_amd.parallel import comm, env as penv  # noqa: E402
from distributed_training_pytorch_amd.parallel.data_parallel import DataParallel  # noqa: E402


def _allclose_across_ranks(t, atol=0.0):
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float((hi - lo).abs().max()) <= atol


def scenario_gpu_shared(out):
    """Two ranks SHARE cuda:0 (one-GPU box) over gloo: the multi-process GPU path — parameter
    broadcast, arena-range buckets, kernel-side ready notifications, all-reduce overlapped with the
    HIP backward kernels, fused optimizer — on a small user-defined fused model, several steps."""
    from distributed_training_pytorch_amd.models import FusedModel
    from distributed_training_pytorch_amd.ops.layers import (AvgPoolFlatten, Conv3x3, ConvFirst, Linear,
                                                             MaxPool2x2, SoftmaxCrossEntropy)
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    rank, world = dist.get_rank(), dist.get_world_size()
    # one-GPU box: every rank on cuda:0 over gloo; DTP_TEST_MULTI_GPU=1 (tests/test_multi_gpu.py, boxes with
    # >= world GPUs): one GPU per rank, RCCL process group, native RCCL gradient transport
    multi = os.environ.get("DTP_TEST_MULTI_GPU") == "1"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if multi else 0)
    torch.cuda.set_device(dev)

    class TinyNet(FusedModel):
        def __init__(self):
            super().__init__()
            self.c1 = ConvFirst(3, 64, relu=True, grad_premasked=True)
            self.c2 = Conv3x3(64, 128, relu=True, input_is_relu_out=True, grad_premasked=True)
            self.pool = MaxPool2x2(relu_mask=True)
            self.flat = AvgPoolFlatten((2, 2))
            self.fc1 = Linear(128 * 4, 256, relu=True, grad_premasked=True)
            self.fc2 = Linear(256, 4, out_fp32=True, input_is_relu_out=True)
            self.finalize()

        def forward(self, x):
            x = self.prepare_input(x)
            return self.fc2(self.fc1(self.flat(self.pool(self.c2(self.c1(x))))))

    torch.manual_seed(100 + rank)                 # different init per rank: broadcast must fix it
    net = TinyNet().to(dev)
    dp = DataParallel(net, bucket_cap_mb=0.05, first_bucket_mb=0.01)
    assert len(dp.reducer.buckets) >= 3, dp.reducer.bucket_summary()
    if multi:
        assert type(dp.reducer.transport).__name__ == "NativeRcclTransport", dp.reducer.transport.info
    torch.cuda.synchronize()
    assert _allclose_across_ranks(net.param_arena.flat_master.cpu()), "broadcast of initial parameters failed"
    opt = FusedSGD(net, lr=0.05, momentum=0.9)
    crit = SoftmaxCrossEntropy()
    g = torch.Generator().manual_seed(7)
    X = torch.randn(16 * world, 3, 32, 32, generator=g)
    Y = torch.randint(0, 4, (16 * world,), generator=g)
    xs, ys = X[rank::world].to(dev), Y[rank::world].to(dev)
    a = net.param_arena
    losses = []
    for it in range(6):
        opt.zero_grad()
        loss = crit(dp(xs), ys)
        loss.backward()
        torch.cuda.synchronize()
        gsum = a.flat_grad.float().cpu()
        assert _allclose_across_ranks(gsum), "step %d: averaged gradients differ across ranks" % it
        opt.step()
        torch.cuda.synchronize()
        assert _allclose_across_ranks(a.flat_master.cpu()), "step %d: parameters diverged across ranks" % it
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    assert all(l == l for l in losses), losses
    assert _allclose_across_ranks(a.flat_master.cpu()), "parameters diverged across ranks"
    if not multi:
        assert dp.reducer.stats["launched"] == 6 * len(dp.reducer.buckets)
    # large-batch reference on this rank (fresh model, rank 0's initial weights are gone: compare the
    # FINAL parameters across ranks above, and the loss trajectory against a single-process run below)
    dp.close()
    open(os.path.join(out, "losses.%d" % rank), "w").write(json.dumps(dict(losses=losses)))


def scenario_reducer(out):
    """Bucketed reducer == single-process large-batch gradient; no_sync accumulates; params stay
    identical across ranks."""
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(1234)                      # same init everywhere, then perturb non-zero ranks
    net = nn.Sequential(nn.Linear(16, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 5))
    ref = nn.Sequential(nn.Linear(16, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 5))
    ref.load_state_dict(net.state_dict())
    if rank != 0:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)                      # DataParallel must overwrite this with rank 0's values
    dp = DataParallel(net, bucket_cap_mb=0.004, first_bucket_mb=0.001)
    assert len(dp.reducer.buckets) >= 3, dp.reducer.bucket_summary()
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.equal(p, q), "broadcast of initial parameters failed"
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8 * world, 16, generator=g)
    Y = torch.randint(0, 5, (8 * world,), generator=g)
    xs, ys = X[rank::world], Y[rank::world]
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9)
    for it in range(3):
        opt.zero_grad(set_to_none=(it % 2 == 0))
        nn.functional.cross_entropy(dp(xs), ys).backward()
        ropt.zero_grad()
        nn.functional.cross_entropy(ref(X), Y).backward()     # mean over the global batch
        for p, q in zip(net.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, q.grad, atol=1e-6), "averaged gradient != large-batch gradient"
        opt.step()
        ropt.step()
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(p, q, atol=1e-6)
        assert _allclose_across_ranks(p.detach())
    # gradient accumulation: two micro-batches, sync only on the second
    opt.zero_grad(set_to_none=False)
    launched = dp.reducer.stats["launched"]
    with dp.no_sync():
        (nn.functional.cross_entropy(dp(xs[:4]), ys[:4]) / 2).backward()
    assert dp.reducer.stats["launched"] == launched, "no_sync must not communicate"
    (nn.functional.cross_entropy(dp(xs[4:]), ys[4:]) / 2).backward()
    ropt.zero_grad()
    Xs = torch.cat([X[r::world][:4] for r in range(world)] + [X[r::world][4:] for r in range(world)])
    Ys = torch.cat([Y[r::world][:4] for r in range(world)] + [Y[r::world][4:] for r in range(world)])
    nn.functional.cross_entropy(ref(Xs), Ys).backward()
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-6), "accumulated gradient mismatch"


def scenario_trainer(out):
    """The real ExampleTrainer on VGG16 with synthetic 32x32x3 data (BASELINE.json config 1)."""
    from distributed_training_pytorch_amd.trainer.example_trainer import ExampleTrainer
    from distributed_training_pytorch_amd.utils.logger import Logger
    rank = dist.get_rank()
    save = os.path.join(out, "runs")
    kw = dict(train_path="synthetic:32", val_path="synthetic:10", labels=["a", "b", "c"], height=32, width=32,
              batch_size=8, pin_memory=True, have_validate=True, save_best_for=("accuracy", "geq"),
              save_period=1, save_folder=save, progress=False, lr=0.01, milestones=(1, 5))
    logger = Logger("T", file=os.path.join(save, "logfile.log"))
    tr = ExampleTrainer(max_epoch=2, snapshot_path=None, logger=logger, **kw)
    assert tr.local_batch_size == 4 and tr.world_size == 2
    tr.train()
    flat = tr.raw_model.param_arena.flat_master
    assert _allclose_across_ranks(flat.detach().clone()), "parameters diverged across ranks"
    comm.barrier()
    files = sorted(os.listdir(os.path.join(save, "weights")))
    assert files == ["best.pth", "last.pth"], files
    from distributed_training_pytorch_amd.utils.checkpoint import load_snapshot
    snap = load_snapshot(os.path.join(save, "weights", "last.pth"))      # the safe (weights_only) loader
    assert snap["epoch"] == 2 and "best_fitness" in snap
    assert abs(tr.scheduler.get_last_lr()[0] - 0.001) < 1e-12          # one milestone passed
    # resume: continues at epoch 2 with the decayed LR and the stored best fitness (quirks #4, #7)
    tr2 = ExampleTrainer(max_epoch=3, snapshot_path=os.path.join(save, "weights", "last.pth"), logger=logger, **kw)
    assert tr2.cur_epoch == 2
    assert abs(tr2.scheduler.get_last_lr()[0] - 0.001) < 1e-12
    assert tr2.best_fitness["epoch"] is not None
    assert torch.equal(tr2.raw_model.param_arena.flat_master, flat)
    tr2.train()
    if rank == 0:
        log = open(os.path.join(save, "logfile.log")).read()
        for needle in ("Epoch 3/3", "VALIDATE RESULTS", "The BEST model is at EPOCH", "THE NEXT LEARNING RATE VALUE IS",
                       "TOTAL LOCAL TRAINING LOSS", "Saved model at epoch 3!", "Finished!"):
            assert needle in log, needle
    assert not os.path.exists(os.path.join(save, "logfile.log.rank1"))


def scenario_sharded(out):
    """Sharded mode (reduce-scatter -> optimizer on the owned slice -> all-gather, per bucket) must
    train exactly like all-reduce + replicated optimizer: same parameters bit for bit after several
    steps (gloo's reduce-scatter stand-in is its all-reduce, so the sums are the same sums), with
    gradient accumulation, for SGD and AdamW; consolidate() completes state on every rank."""
    from distributed_training_pytorch_amd.models import FusedModel
    from distributed_training_pytorch_amd.ops.layers import Linear
    from distributed_training_pytorch_amd.ops.optim import FusedAdamW, FusedSGD
    rank, world = dist.get_rank(), dist.get_world_size()

    class Net(FusedModel):
        def __init__(self):
            super().__init__()
            self.a = Linear(16, 96, relu=True)
            self.b = Linear(96, 136, relu=True)
            self.b2 = Linear(136, 120, relu=True)
            self.c = Linear(120, 5, out_fp32=True)
            self.finalize()

        def forward(self, x):
            return self.c(self.b2(self.b(self.a(self.prepare_input(x)))))

    g = torch.Generator().manual_seed(7)
    X = torch.randn(6, 8 * world, 16, generator=g)
    Y = torch.randint(0, 5, (6, 8 * world), generator=g)

    def run(opt_cls, sharded, per_bucket=False, align=None, **okw):
        torch.manual_seed(4321)
        net = Net()
        opt = opt_cls(net, **okw)
        # the reference run uses the SAME bucket ranges (snapped to world * 64 elements): gloo's ring
        # all-reduce sums an element in an order that depends on its position in the message, so
        # only equal messages give equal bits at world 4
        dp = DataParallel(net, bucket_cap_mb=0.02, first_bucket_mb=0.004,
                          shard_optimizer=opt if sharded else None,
                          bucket_optimizer=opt if per_bucket else None,
                          bucket_align_elems=None if (sharded or per_bucket) else (align or world * 64))
        assert dp.reducer.sharded == sharded
        assert dp.reducer.fused_update == (sharded or per_bucket)
        assert len(dp.reducer.buckets) >= 3, dp.reducer.bucket_summary()
        if sharded:
            for b in dp.reducer.buckets:
                assert (b.hi - b.lo) % (world * 64) == 0, (b.lo, b.hi)
            cover = sorted((b.lo, b.hi) for b in dp.reducer.buckets)
            assert cover[0][0] == 0 and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
            assert cover[-1][1] >= net.param_arena.total
        for it in range(6):
            xs, ys = X[it, rank::world], Y[it, rank::world]
            opt.zero_grad()
            if it % 3 == 2:                                   # a two-micro-step accumulation window
                with dp.no_sync():
                    (nn.functional.cross_entropy(dp(xs[:4]), ys[:4]) / 2).backward()
                (nn.functional.cross_entropy(dp(xs[4:]), ys[4:]) / 2).backward()
            else:
                nn.functional.cross_entropy(dp(xs), ys).backward()
            opt.step()
        dp.consolidate()
        sd = opt.state_dict()
        state = torch.cat([v.reshape(-1) for st in sd["state"].values() for v in st.values() if torch.is_tensor(v)])
        return net.param_arena.flat_master[:net.param_arena.total].clone(), state, dp

    for opt_cls, okw in ((FusedSGD, dict(lr=0.05, momentum=0.9, weight_decay=1e-4)),
                         (FusedAdamW, dict(lr=1e-3, weight_decay=1e-2))):
        ref_w, ref_s, _ = run(opt_cls, False, **okw)
        sh_w, sh_s, dp = run(opt_cls, True, **okw)
        assert torch.equal(ref_w, sh_w), "%s: sharded parameters differ (max %g)" % (
            opt_cls.__name__, float((ref_w - sh_w).abs().max()))
        assert torch.equal(ref_s, sh_s), "%s: consolidated optimizer state differs" % opt_cls.__name__
        assert _allclose_across_ranks(sh_w) and _allclose_across_ranks(sh_s)
        try:                                                  # step() without a synchronised backward: loud
            dp.reducer.opt.step()
            raise AssertionError("sharded step() without a backward pass must raise")
        except RuntimeError:
            pass
        # the default data-parallel path: all-reduce per bucket + the optimizer on each bucket's range
        # right behind it (replicated) == all-reduce + one optimizer pass, bit for bit
        ref_w, ref_s, _ = run(opt_cls, False, align=64, **okw)
        pb_w, pb_s, dpb = run(opt_cls, False, per_bucket=True, **okw)
        assert not dpb.reducer.sharded and len(dpb.reducer.buckets) >= 3
        assert torch.equal(ref_w, pb_w), "%s: per-bucket optimizer parameters differ (max %g)" % (
            opt_cls.__name__, float((ref_w - pb_w).abs().max()))
        assert torch.equal(ref_s, pb_s), "%s: per-bucket optimizer state differs" % opt_cls.__name__
        assert _allclose_across_ranks(pb_w) and _allclose_across_ranks(pb_s)


def scenario_gpu_modes(out):
    """VGG16-BN (biases + BatchNorm affine pairs + running statistics, dropout 0.3) on GPUs, 5 steps,
    in every gradient-synchronisation mode: all-reduce + replicated step, per-bucket optimizer (the
    default), sharded optimizer, and — on the native RCCL transport — the hipGraph-captured step.
    Asserts: parameters (incl. the fp32 biases / BN affine values the kernels read from the master)
    agree across ranks after EVERY step in every mode; buffers agree after eval(); per-bucket ==
    all-reduce bit for bit; sharded and hipGraph land on the all-reduce result.
    One-GPU box: ranks share cuda:0 over gloo (Python reducer path); DTP_TEST_MULTI_GPU=1: one GPU
    per rank, RCCL, native transport (C++ scheduler)."""
    from distributed_training_pytorch_amd.engine.graph import GraphedTrainStep
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    rank, world = dist.get_rank(), dist.get_world_size()
    multi = os.environ.get("DTP_TEST_MULTI_GPU") == "1"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if multi else 0)
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(11)
    X = torch.randn(5, 8 * world, 3, 32, 32, generator=g)
    Y = torch.randint(0, 3, (5, 8 * world), generator=g)

    def run(mode):
        torch.manual_seed(55)
        net = VGG16(3, 3, True, batch_norm=True, dropout=0.3, grad_dtype=torch.bfloat16).to(dev).train()
        net.dropout_rng.seed = 77
        opt = FusedSGD(net, lr=0.01, momentum=0.9, weight_decay=1e-4)
        crit = SoftmaxCrossEntropy()
        dp = DataParallel(net, shard_optimizer=opt if mode == "sharded" else None,
                          bucket_optimizer=opt if mode == "bucket" else None,
                          bucket_align_elems=64 if mode in ("allreduce", "graph") else None,
                          bucket_cap_mb=4.0, first_bucket_mb=1.0)
        if multi:
            assert type(dp.reducer.transport).__name__ == "NativeRcclTransport", dp.reducer.transport.info
        a = net.param_arena
        small = torch.cat([torch.arange(a.offsets[i], a.offsets[i] + p.numel()) for i, p in enumerate(a.params)
                           if p.dim() <= 1]).to(dev)
        graphed = GraphedTrainStep(dp, crit, opt, warmup=1) if mode == "graph" else None
        if graphed is not None:
            assert graphed.enabled and graphed.distributed
        losses = []
        for it in range(5):
            xs, ys = X[it, rank::world].to(dev), Y[it, rank::world].to(dev)
            if graphed is not None:
                loss = graphed(xs, ys)
            else:
                opt.zero_grad()
                loss = crit(dp(xs), ys)
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            losses.append(float(loss.detach()))
            # what the NEXT forward reads: bf16 working copy everywhere, fp32 master for biases / BN affine
            assert _allclose_across_ranks(a.flat_lp.float().cpu()), "%s step %d: working weights diverged" % (mode, it)
            assert _allclose_across_ranks(a.flat_master[small].cpu()), \
                "%s step %d: fp32-read parameters (biases / BatchNorm affine) differ across ranks" % (mode, it)
        if graphed is not None:
            assert graphed.captures == 1 and graphed.replays == 4
        dp.consolidate()
        dp.eval()                                              # collective: running statistics follow rank 0
        torch.cuda.synchronize()
        w = a.flat_master[:a.total].detach().cpu().clone()
        bufs = torch.cat([b.detach().float().reshape(-1).cpu() for b in net.buffers() if b.is_floating_point()])
        assert _allclose_across_ranks(w), "%s: parameters diverged across ranks" % mode
        assert _allclose_across_ranks(bufs), "%s: buffers differ across ranks after eval()" % mode
        assert all(l == l for l in losses), (mode, losses)
        dp.close()
        return w, losses

    ref_w, ref_l = run("allreduce")
    pb_w, pb_l = run("bucket")
    assert torch.equal(ref_w, pb_w) and ref_l == pb_l, "per-bucket optimizer != all-reduce + replicated step (max %g)" % (
        float((ref_w - pb_w).abs().max()))
    sh_w, sh_l = run("sharded")
    # different bucket cuts -> a different summation order inside the collective at world > 2; bf16 gradients
    scale = float(ref_w.abs().max())
    assert float((ref_w - sh_w).abs().max()) <= 2e-3 * scale, "sharded drifted from all-reduce: %g" % (
        float((ref_w - sh_w).abs().max()))
    if world == 2 and not multi:
        assert torch.equal(ref_w, sh_w), "world 2 over gloo: sharded must equal all-reduce bit for bit"
    if multi:
        gr_w, gr_l = run("graph")
        assert torch.equal(ref_w, gr_w) and ref_l == gr_l, "hipGraph data-parallel step != eager (max %g)" % (
            float((ref_w - gr_w).abs().max()))
    open(os.path.join(out, "modes.%d" % rank), "w").write(json.dumps(dict(allreduce=ref_l, bucket=pb_l, sharded=sh_l)))


def scenario_small_val(out):
    """len(val) < world: a rank with an EMPTY validation shard must still take part in the metric
    reduction with a vector of the right length (was: size-mismatched collective)."""
    from distributed_training_pytorch_amd.trainer.example_trainer import ExampleTrainer
    world = dist.get_world_size()
    tr = ExampleTrainer(train_path="synthetic:%d" % (4 * world), val_path="synthetic:1", labels=["a", "b", "c"],
                        height=32, width=32, max_epoch=1, batch_size=2 * world, pin_memory=False, have_validate=True,
                        save_best_for=("accuracy", "geq"), save_period=1, save_folder=os.path.join(out, "runs"),
                        progress=False, lr=0.01, max_steps_per_epoch=1)
    tr.model.eval()
    m = tr.validate()
    assert set(m) == {"accuracy"} and m["accuracy"] in (0.0, 1.0), m      # one sample in total
    vals = [None] * world
    dist.all_gather_object(vals, m)
    assert all(v == vals[0] for v in vals), vals
    tr.train()                                                             # the epoch loop runs through it too


def scenario_fault(out):
    """Kill-a-rank: rank 1 dies at global step 1; rank 0 must fail fast (timeout), not hang."""
    from distributed_training_pytorch_amd.trainer.example_trainer import ExampleTrainer
    os.environ["FAULT_INJECT_STEP"] = "1"
    os.environ["FAULT_INJECT_RANK"] = "1"
    tr = ExampleTrainer(train_path="synthetic:32", val_path="synthetic:8", labels=["a", "b", "c"], height=32,
                        width=32, max_epoch=1, batch_size=8, pin_memory=False, have_validate=False,
                        save_period=1, save_folder=os.path.join(out, "runs"), progress=False, lr=0.01)
    tr.train()


if __name__ == "__main__":
    scenario, out = sys.argv[1], sys.argv[2]
    penv.ddp_setup("nccl" if os.environ.get("DTP_TEST_MULTI_GPU") == "1" else "gloo",
                   timeout_s=float(os.environ.get("DTP_PG_TIMEOUT_S", "60")))
    {"reducer": scenario_reducer, "trainer": scenario_trainer, "fault": scenario_fault,
     "gpu_shared": scenario_gpu_shared, "sharded": scenario_sharded, "small_val": scenario_small_val,
     "gpu_modes": scenario_gpu_modes}[scenario](out)
    comm.barrier()
    open(os.path.join(out, "ok.%d" % dist.get_rank()), "w").write("ok")
    penv.destroy_process()
