"""Headline benchmark: VGG16 bf16 224x224 training images/sec (whole job), synthetic data,
random-init weights, on N MI355X of one node (BASELINE.json "metric").

  python bench.py                                  # 1 GPU
  python bench.py --gpus 8                         # self-launch: 8 fresh worker processes, one per GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W          # the driver's launch

One timed step = zero_grad + ingest (NCHW f32 -> NHWC bf16, or pinned uint8 host batch -> H2D copy
-> fused normalisation with ``--data host-u8``) + forward + fused softmax-xent + backward
(weight/bias gradients for all 32 tensors, bucketed RCCL all-reduce overlapped with backward when
N > 1) + fused SGD-momentum step on fp32 master weights.  Weak scaling: the per-GPU batch is fixed.
Rank 0 prints ONE JSON line.

N > 1 on GPUs: before a rank touches its GPU it rehearses the run (3 steps at batch 8, native RCCL transport
forced) in a throw-away child process on another rendezvous port; if that fails, hangs or crashes on any
rank, the real run uses the torch.distributed process-group transport and reports it (``preflight``,
``native_transport_fell_back``).  DTP_PREFLIGHT=0 skips the rehearsal.

Without WORLD_SIZE in the environment and with --gpus N > 1 this script is its own launcher
(reference run.sh:9-14 starts the ranks with torchrun): the parent process never touches the GPU
— it does not import torch — and starts N brand-new workers (distributed_training_pytorch_amd/launch.py),
relays their output and exits non-zero if any of them does.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Default per-GPU batch 1024 (50 GB of the 288 GB HBM; BASELINE.json config 4): ~10,000 img/s on one
# MI355X vs 9,500 at 256 and 3,995 for stock PyTorch-ROCm at 1024.
# stock PyTorch-ROCm (MIOpen / hipBLASLt, autocast bf16, channels_last), 1 MI355X, same step, by
# per-GPU batch (profiles/00_*, 07_*: bench/bench_stock.py); the comparison uses the entry measured
# at the nearest batch not above the one benchmarked
STOCK_1GPU_BF16_IMG_S = {4: 739.0, 16: 2002.0, 64: 3352.0, 128: 3759.0, 256: 3939.0, 1024: 3995.0}


def stock_img_s(batch):
    keys = [k for k in sorted(STOCK_1GPU_BF16_IMG_S) if k <= batch] or [min(STOCK_1GPU_BF16_IMG_S)]
    return STOCK_1GPU_BF16_IMG_S[keys[-1]], keys[-1]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=int(os.environ.get("DTP_BENCH_BATCH", "1024")))
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--grad-dtype", default=os.environ.get("DTP_BENCH_GRAD_DTYPE", "bf16"), choices=["fp32", "bf16"])
    ap.add_argument("--lr", type=float, default=0.001)
    ap.add_argument("--backend", default=None, help="process-group backend (nccl = RCCL; gloo for rehearsals / CPU)")
    ap.add_argument("--device", default="auto", choices=["auto", "cuda", "cpu"],
                    help="cpu: plumbing rehearsal on the PyTorch reference ops (not a benchmark)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (multi-process plumbing on a one-GPU box)")
    ap.add_argument("--grad-accum", type=int, default=1,
                    help="micro-batches per optimizer step (BASELINE config 4): the per-GPU batch is split "
                         "into this many micro-steps, all-reduce only on the last one (no_sync)")
    ap.add_argument("--data", default="device", choices=["device", "host-u8"],
                    help="device: pre-generated device-resident f32 batches; host-u8: uint8 batches in a pinned "
                         "host ring, H2D copy on the copy stream inside the timed region, normalisation fused "
                         "into the first convolution (the reference's pinned loader + per-step H2D, main.py:16)")
    ap.add_argument("--batch-norm", action="store_true", help="VGG16-BN (BASELINE.json config 5)")
    ap.add_argument("--fp8-classifier", action="store_true", help="linear1/linear2 GEMMs in OCP e4m3 (config 5)")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="reduce-scatter + per-bucket sharded SGD + all-gather instead of all-reduce + replicated SGD")
    ap.add_argument("--fresh-data", action="store_true",
                    help="every warm-up and timed step sees a batch it has not seen before (n_distinct = warmup + "
                         "steps device-resident batches) instead of cycling through 2: the net cannot memorise, "
                         "so activation / gradient sparsity — which the power-limited matrix kernels are "
                         "sensitive to — stays that of a freshly initialised net")
    ap.add_argument("--hip-graph", default="auto", choices=["auto", "on", "off"],
                    help="capture zero_grad + forward + loss + backward + optimizer into ONE hipGraph launch "
                         "(engine/graph.py).  auto: single-GPU runs at per-GPU batch <= 128 — measured +3.8 % at 32, "
                         "+2.8 % at 64, +1.5 % at 128, +0.4 % at 256 and -1.8 % at 1024 (there the per-bucket optimizer "
                         "slices under backward are worth more than the launches); the reference's own regime is 4 "
                         "images per GPU (main.py:15 / run.sh:10)")
    ap.add_argument("--no-bucket-optimizer", action="store_true",
                    help="N > 1: run the fused SGD as one pass after the last gradient bucket instead of per "
                         "bucket behind its all-reduce (the default; parallel/reducer.py)")
    ap.add_argument("--preflight-child", action="store_true",
                    help="internal: this process is a pre-flight rehearsal of an N > 1 run (see preflight_native)")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """Parent side of ``python bench.py --gpus N``: no torch import, no GPU call."""
    from distributed_training_pytorch_amd.launch import spawn_workers
    return spawn_workers(os.path.abspath(__file__), argv, args.gpus)


def preflight_native(args, world):
    """N > 1 on GPUs with the transport left to ``auto``: before this rank touches its GPU, rehearse the
    run — 3 steps at per-GPU batch 8 with the native RCCL transport FORCED, same model (same gradient
    buckets), same optimizer placement, bench.py's own cross-rank self-checks — in one throw-away child per
    rank (launch.preflight).  If the rehearsal fails, hangs or crashes on any rank, the real run takes the
    torch.distributed process-group transport and says so in its JSON line: a failure that only more than
    one rank can produce costs a fallback, not the measurement.  (The native path has run on RCCL at world
    size 1 only from this repository's one-GPU boxes; reference trainer/trainer.py:52 is the DDP it replaces.)
    DTP_PREFLIGHT=0 skips it, DTP_PREFLIGHT=force runs it for CPU / gloo rehearsals too."""
    mode = os.environ.get("DTP_PREFLIGHT", "1")
    if world <= 1 or mode == "0" or args.preflight_child:
        return None
    if os.environ.get("DTP_TRANSPORT", "auto").lower() != "auto":
        return None
    if mode != "force":
        if args.device == "cpu" or (args.backend or "nccl") != "nccl" or args.share_device:
            return None
        import torch
        if torch.cuda.device_count() < 1:       # (counting devices does not initialise HIP)
            return None
    from distributed_training_pytorch_amd.launch import preflight
    child = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch-per-gpu", "8", "--preflight-child",
             "--grad-dtype", args.grad_dtype, "--hip-graph", "off", "--device", args.device]
    if args.backend:
        child += ["--backend", args.backend]
    for flag, on in (("--batch-norm", args.batch_norm), ("--fp8-classifier", args.fp8_classifier),
                     ("--shard-optimizer", args.shard_optimizer), ("--no-bucket-optimizer", args.no_bucket_optimizer),
                     ("--share-device", args.share_device)):
        if on:
            child.append(flag)
    if args.image_size != 224:
        child += ["--image-size", str(args.image_size)]
    env = {"DTP_NATIVE_INIT_TIMEOUT_S": os.environ.get("DTP_NATIVE_INIT_TIMEOUT_S", "60"),
           "DTP_PG_TIMEOUT_S": os.environ.get("DTP_PG_TIMEOUT_S", "180")}
    if mode != "force":
        env["DTP_TRANSPORT"] = "native"
    t0 = time.perf_counter()
    ok, reason = preflight(os.path.abspath(__file__), child,
                           timeout_s=float(os.environ.get("DTP_PREFLIGHT_TIMEOUT_S", "270")), extra_env=env)
    return dict(ok=bool(ok), reason=reason, seconds=round(time.perf_counter() - t0, 1))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.share_device and args.gpus > 1 and (args.backend or "nccl") == "nccl" and args.device != "cpu":
        # RCCL refuses two ranks on one device ("duplicate GPU"); say so before anything is spawned
        raise SystemExit("--share-device rehearses the multi-process plumbing on ONE GPU and needs --backend gloo "
                         "(RCCL cannot place two ranks on the same device)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, argv))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it here
    if args.preflight_child and os.environ.get("DTP_PREFLIGHT_INJECT"):      # tests: a rehearsal that fails / hangs
        if os.environ["DTP_PREFLIGHT_INJECT"] == "hang":
            time.sleep(3600)
        sys.exit(3)
    pre = preflight_native(args, int(os.environ.get("WORLD_SIZE", "1")))
    # stdout carries ONE line, the JSON result of rank 0.  Libraries below us write there too (RCCL's
    # version banner and WARN lines, gloo's "Rank 0 is connected to ..." notices), so file descriptor 1
    # points at stderr for the duration of the run and is restored for the result line.
    sys.stdout.flush()
    try:
        real_stdout = os.dup(1)
        os.dup2(2, 1)
    except OSError:                 # no usable stdout / stderr descriptor: leave the streams as they are
        real_stdout = None
    if os.environ.get("DTP_BENCH_DUMP_AFTER"):      # debugging aid: dump every thread's stack and exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["DTP_BENCH_DUMP_AFTER"]), exit=True)

    import torch
    import torch.distributed as dist
    from distributed_training_pytorch_amd.parallel import env as penv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    on_gpu = args.device == "cuda" or (args.device == "auto" and torch.cuda.is_available())
    backend = args.backend or ("nccl" if on_gpu else "gloo")
    applied_env = penv.apply_rccl_env_defaults()
    if on_gpu:
        if args.share_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    else:
        dev = torch.device("cpu")
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(world, 1)))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        kw = dict(timeout=datetime.timedelta(seconds=float(os.environ.get("DTP_PG_TIMEOUT_S", "600"))))
        if backend == "nccl":
            kw["device_id"] = dev
        dist.init_process_group(backend, **kw)
    if pre is not None and world > 1:
        # every rank takes the same transport: the rehearsal counts only if it passed everywhere
        from distributed_training_pytorch_amd.parallel import comm
        all_ok = comm.all_reduce_scalars([1.0 if pre["ok"] else 0.0], op="min")[0] >= 1.0
        if not all_ok:
            os.environ["DTP_TRANSPORT"] = "torch"
            os.environ["DTP_NATIVE_FELL_BACK"] = "1"
            os.environ["DTP_NATIVE_FALLBACK_REASON"] = "pre-flight rehearsal of the native transport failed: %s" % (
                pre["reason"] or "on another rank")
            if rank == 0:
                print("[bench] %s; this run uses the torch.distributed process-group transport"
                      % os.environ["DTP_NATIVE_FALLBACK_REASON"], file=sys.stderr, flush=True)
        pre["all_ranks_ok"] = bool(all_ok)

    from distributed_training_pytorch_amd.data.synthetic import DeviceBatchSource
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    from distributed_training_pytorch_amd.parallel.data_parallel import DataParallel

    torch.manual_seed(0)
    gdt = torch.bfloat16 if (args.grad_dtype == "bf16" and on_gpu) else torch.float32
    model = VGG16(3, args.classes, True, grad_dtype=gdt, batch_norm=args.batch_norm,
                  fp8_classifier=args.fp8_classifier).to(dev)
    opt = FusedSGD(model, lr=args.lr, momentum=0.9, weight_decay=1e-4)
    crit = SoftmaxCrossEntropy()
    per_bucket = not args.shard_optimizer and not args.no_bucket_optimizer
    dp = DataParallel(model, shard_optimizer=opt if args.shard_optimizer else None,
                      bucket_optimizer=opt if per_bucket else None)
    dp.train()
    if args.grad_accum < 1 or args.batch_per_gpu % args.grad_accum:
        raise SystemExit("--batch-per-gpu must be a multiple of --grad-accum")
    micro = args.batch_per_gpu // args.grad_accum
    if args.data == "host-u8":
        from distributed_training_pytorch_amd.data.prefetch import PinnedRingSource
        data = PinnedRingSource(micro, args.classes, args.image_size, args.image_size, device=dev, seed=rank)
        model.set_input_normalization(*data.normalization)
        data_desc = ("synthetic (uint8 NCHW batches in a pinned host ring, H2D copy on the copy stream inside the "
                     "timed region, normalisation fused into conv1; random-init weights)")
    else:
        n_distinct = (args.warmup + args.steps) * args.grad_accum if args.fresh_data else 2
        data = DeviceBatchSource(micro, args.classes, args.image_size, args.image_size,
                                 device=dev, n_distinct=n_distinct, seed=rank)
        data_desc = "synthetic (device-resident f32 NCHW batches, %s, random-init weights)" % (
            "a fresh batch every step" if args.fresh_data else "2 distinct batches cycled")

    use_graph = args.hip_graph == "on" or (args.hip_graph == "auto" and world == 1 and args.batch_per_gpu <= 128)
    use_graph = use_graph and on_gpu and args.grad_accum == 1 and args.data == "device"
    graphed = None
    if use_graph:
        from distributed_training_pytorch_amd.engine.graph import GraphedTrainStep
        graphed = GraphedTrainStep(dp, crit, opt, warmup=2)
        if not graphed.enabled:
            graphed = None

    def step():
        if graphed is not None:                 # the same work as below, replayed as one graph launch
            x, y = next(data)
            return graphed(x, y)
        opt.zero_grad()
        for k in range(args.grad_accum):
            x, y = next(data)
            if k + 1 < args.grad_accum:
                with dp.no_sync():
                    (crit(dp(x), y) / args.grad_accum).backward()
            else:
                loss = crit(dp(x), y)
                (loss / args.grad_accum if args.grad_accum > 1 else loss).backward()
        opt.step()
        return loss

    def fence():
        if on_gpu:
            torch.cuda.synchronize()          # drain compute + gradient-comm streams before the barrier
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    verbose = os.environ.get("DTP_BENCH_VERBOSE") == "1"
    for i in range(args.warmup):
        loss = step()
        if verbose:
            if on_gpu:
                torch.cuda.synchronize()
            print("[bench rank %d] warm-up step %d done, loss %.4f" % (rank, i, float(loss.detach())),
                  file=sys.stderr, flush=True)
    fence()
    exposed0 = dp.reducer.exposed_ms_total()
    wait0 = getattr(data, "wait_ms_total", lambda: 0.0)()
    starved0 = getattr(data, "starved_ms_total", lambda: 0.0)()
    # HIP events around the first and the last five timed steps (recorded on the compute stream, read
    # after the closing fence: no extra synchronisation): shows whether throughput drifts while the
    # net memorises its batches (activation sparsity moves; the matrix kernels are power-limited)
    marks = {}
    edge = min(5, args.steps)

    def mark(name):
        if on_gpu:
            marks[name] = torch.cuda.Event(enable_timing=True)
            marks[name].record()

    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == 0:
            mark("first_begin")
        if i == args.steps - edge:
            mark("last_begin")
        loss = step()
        if i == edge - 1:
            mark("first_end")
    mark("last_end")
    fence()
    dt = time.perf_counter() - t0
    ms_first = marks["first_begin"].elapsed_time(marks["first_end"]) / edge if on_gpu and edge else None
    ms_last = marks["last_begin"].elapsed_time(marks["last_end"]) / edge if on_gpu and edge else None
    exposed = (dp.reducer.exposed_ms_total() - exposed0) / max(1, args.steps)
    data_wait = (getattr(data, "wait_ms_total", lambda: 0.0)() - wait0) / max(1, args.steps)
    data_starved = (getattr(data, "starved_ms_total", lambda: 0.0)() - starved0) / max(1, args.steps)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([dt], device=cdev, dtype=torch.float64)
    ones = torch.ones(1, device=cdev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)          # how many ranks really took part
    dt = float(t.item())
    ranks_seen = int(round(float(ones.item())))
    final_loss = float(loss.detach().float().item())
    # self-check of the gradient transport itself: average of (rank + 1) must be (world + 1) / 2
    tr = dp.reducer.transport
    probe = torch.full((256,), float(rank + 1), device=dev)
    tr.wait(tr.launch(probe), probe)
    if on_gpu:
        torch.cuda.synchronize()
    transport_ok = bool(((probe - (world + 1) / 2.0).abs() < 1e-5).all())
    # after all those steps every rank must hold the same model: checksum of the fp32 master weights,
    # min and max over ranks (a diverged replica — stale slices, a lost bucket — shows up here)
    params_agree = True
    if world > 1:
        dp.consolidate()                                  # sharded mode: complete the master on every rank
        a = model.param_arena
        cs = a.flat_master[:a.total].double().abs().sum().reshape(1).to(cdev)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        params_agree = bool((hi - lo).abs().item() <= 1e-9 * max(1.0, abs(hi.item())))
    if rank == 0:
        global_batch = args.batch_per_gpu * world
        value = global_batch * args.steps / dt
        info = getattr(tr, "info", None) or {}
        plain = not (args.batch_norm or args.fp8_classifier)
        sys.stdout.flush()
        if real_stdout is not None:
            os.dup2(real_stdout, 1)
        print(json.dumps({
            "metric": "train images/sec (whole node) VGG16 bf16 224px",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "ms_first5": ms_first, "ms_last5": ms_last,      # GPU time per step over the first / last 5 timed steps (rank 0)
            "scaling": "weak",
            "vs_baseline": (value / (stock_img_s(args.batch_per_gpu)[0] * world)) if (on_gpu and plain) else None,
            "baseline": "stock PyTorch-ROCm bf16 channels_last on 1 MI355X (%.0f img/s at per-GPU batch %d) x "
                        "n_gpus; the reference publishes no numbers" % stock_img_s(args.batch_per_gpu),
            "dtype": "bf16" if on_gpu else "fp32 (CPU rehearsal on the reference ops, not a benchmark)",
            "data": data_desc,
            "final_loss": final_loss,
            "hip_graph": graphed is not None and graphed.replays > 0,
            "max_memory_reserved_gb": (torch.cuda.max_memory_reserved(dev) / 2 ** 30) if on_gpu else 0.0,
            "allreduce_exposed_ms_per_step": exposed,     # compute stream idle waiting for gradient buckets (rank 0)
            "data_wait_ms_per_step": data_wait,           # host blocked on the input ring (rank 0)
            "data_starved_ms_per_step": data_starved,     # compute stream idle waiting for its batch's H2D copy
            "ranks_seen": ranks_seen,
            "grad_transport": type(tr).__name__,
            "grad_transport_selfcheck_ok": transport_ok,
            "params_agree_across_ranks": params_agree,
            "native_transport_fell_back": bool(info.get("fell_back", False)),
            "native_transport_fallback_reason": info.get("reason"),
            "bucket_mbytes": [round(b["mbytes"], 3) for b in dp.reducer.bucket_summary()],
            "rccl_env_applied_by_bench": applied_env,
            "preflight": pre,       # N > 1: rehearsal of the native transport in throw-away children (None: not run)
            "config": {"model": "VGG16 (%s, %d classes, %dM params)" % (
                           "BN" if args.batch_norm else "no BN", args.classes,
                           round(sum(p.numel() for p in model.parameters()) / 1e6)),
                       "global_batch": global_batch, "image_size": args.image_size,
                       "seq_len": None, "parallelism": "dp%d" % world,
                       "optimizer": "fused SGD momentum 0.9 wd 1e-4, fp32 master weights%s" % (
                           ", sharded per bucket (reduce-scatter + all-gather)" if dp.reducer.sharded else
                           ", per bucket behind its all-reduce" if dp.reducer.fused_update else ""),
                       "grad_dtype": args.grad_dtype if on_gpu else "fp32", "grad_accum": args.grad_accum,
                       "fp8_classifier": bool(args.fp8_classifier), "backend": backend,
                       "grad_transport": type(tr).__name__}}), flush=True)
        if real_stdout is not None:
            os.dup2(2, 1)                   # teardown chatter (communicator destruction) stays off stdout
    if real_stdout is not None:
        os.close(real_stdout)
    if on_gpu:
        torch.cuda.synchronize()
    if hasattr(data, "close"):
        data.close()
    dp.close()
    if world > 1:
        dist.destroy_process_group()
    if not transport_ok or ranks_seen != world or not params_agree:
        raise SystemExit("gradient transport self-check failed (ok=%s, ranks seen %d of %d, parameters agree: %s)"
                         % (transport_ok, ranks_seen, world, params_agree))


if __name__ == "__main__":
    main()
