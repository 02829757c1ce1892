"""Template-method distributed training runtime.

Parity target: reference ``trainer/trainer.py:14-253`` — same constructor arguments and defaults
(``:15-24``), same static ``ddp_setup(backend)`` / ``destroy_process()`` (``:74-82``), same nine
overridable hooks (``:220-253``), same epoch loop (``:104-181``): periodic validation + best-model
tracking (``geq``/``leq``), ``sampler.set_epoch``, per-batch ``train_step`` dicts, per-epoch
``scheduler.step()`` + LR log, rank-0 snapshots ``best`` / ``last`` / ``checkpoint_epoch_N`` fenced
by barriers, and the same log strings.

MI355X-first re-design (every item of SURVEY.md §2.6 is fixed, none is copied):
  * gradient sync is ``parallel.DataParallel`` (bucketed RCCL all-reduce over flat gradient
    arenas) instead of ``torch.nn.parallel.DistributedDataParallel`` (``:52``);
  * validation is sharded over all ranks and metrics are all-reduced (quirks #8/#9) instead of
    rank 0 validating while the others sit in a barrier;
  * batches arrive through an async pinned prefetcher on a copy stream (quirk #16) and step
    metrics may stay on the device — they are read back every ``log_every`` steps, not every step
    (SURVEY.md §2.5 K18);
  * model is placed on its device *before* optimizer / resume (quirk #4); config is validated up
    front (quirks #5, #11); snapshot writes are atomic and carry ``best_fitness`` (quirk #7);
    means are sample-weighted (quirk #20); works without torchrun and on CPU/gloo.
"""
import copy
import json
import os
import time

import torch

from ..utils.profiler import trace_range
from torch.utils.data import DataLoader

from ..data.device_cache import DeviceCachedLoader
from ..data.prefetch import DevicePrefetcher
from ..parallel import comm, env as penv
from ..parallel.data_parallel import DataParallel
from ..parallel.sampler import ShardedSampler
from ..utils import checkpoint as ckpt
from ..utils.meter import StepTimer, ThroughputMeter

try:
    from tqdm import tqdm as _tqdm
except Exception:  # tqdm is optional
    _tqdm = None

RULE = 100 * "="


def _to_float(v):
    if torch.is_tensor(v):
        return float(v.detach().float().mean().item())
    return float(v)


class Trainer(object):
    def __init__(self,
                 max_epoch,
                 batch_size,
                 pin_memory,
                 have_validate=False,
                 save_best_for=None,
                 save_period=None,
                 save_folder='.',
                 snapshot_path=None,
                 logger=None,
                 *,
                 num_workers=0,
                 grad_accum_steps=1,
                 device=None,
                 log_every=10,
                 progress=True,
                 max_steps_per_epoch=None,
                 bucket_cap_mb=None,
                 reduce_dtype=None,
                 metrics_file=None,
                 seed=0,
                 use_hip_graph=False,
                 cache_dataset_on_device=False,
                 shard_optimizer=False,
                 bucket_optimizer=True,
                 step_timing=None):
        # T1 logging shim (plain method, keyword-friendly; reference :26)
        self.logger = logger
        # T5 rank discovery (reference :48-50) — defaults make `python main.py` work
        denv = penv.DistEnv.from_env()
        self.local_rank, self.world_rank, self.world_size = denv.local_rank, denv.rank, denv.world_size
        if comm.is_dist():
            self.world_rank, self.world_size = comm.rank(), comm.world_size()
        self.device = torch.device(device) if device is not None else penv.default_device(self.local_rank)

        # config validation (quirks #5, #11)
        if save_period is None:
            save_period = 1
        if not isinstance(save_period, int) or save_period < 1:
            raise ValueError("save_period must be a positive int, got %r" % (save_period,))
        if have_validate:
            if (not save_best_for or len(save_best_for) != 2 or save_best_for[1] not in ("geq", "leq")):
                raise ValueError("have_validate=True needs save_best_for=(metric_name, 'geq'|'leq')")
        if batch_size < self.world_size or batch_size % self.world_size != 0:
            raise ValueError("global batch_size %d must be a positive multiple of world size %d"
                             % (batch_size, self.world_size))
        if grad_accum_steps < 1:
            raise ValueError("grad_accum_steps must be >= 1")

        # T2 save dir (quirk #18)
        self.save_folder = save_folder
        self.save_weight_folder = os.path.join(self.save_folder, "weights")
        os.makedirs(self.save_weight_folder, exist_ok=True)

        self.save_best_for = save_best_for
        self.have_validate = have_validate
        self.save_period = save_period
        self.cur_epoch = 0
        self.max_epoch = max_epoch
        self.grad_accum_steps = int(grad_accum_steps)
        self.log_every = max(1, int(log_every))
        self.progress = progress
        self.max_steps_per_epoch = max_steps_per_epoch
        self.seed = seed
        self.global_step = 0
        self.best_fitness = dict(epoch=None, value=None, metrics=None)
        self.metrics_file = metrics_file
        self.meter = ThroughputMeter()
        # per-phase GPU timings (forward / backward / optimizer) from HIP events, resolved once per
        # epoch; opt-in (DTP_STEP_TIMING=1) because every event costs a launch-sized host call
        if step_timing is None:
            step_timing = os.environ.get("DTP_STEP_TIMING", "0") == "1"
        self.step_timer = StepTimer(enabled=bool(step_timing))
        self.steps_this_epoch = None
        # capture the whole train step into one hipGraph (engine/graph.py); hooks opt in through
        # self.graphed_step — runs without gradient accumulation are captured; data-parallel ones only
        # on the native RCCL transport (the reduction is then part of the graph)
        # keep whole (small enough) datasets resident on the GPU and gather batches there
        # (data/device_cache.py); falls back to the DataLoader when the dataset does not fit
        self.cache_dataset_on_device = bool(cache_dataset_on_device)
        self.use_hip_graph = bool(use_hip_graph)
        self.graphed_step = None

        # RNG is part of the run state: ``seed`` seeds torch (weight init, CPU dropout, augmentation
        # draws of datasets that use the global generator) and keys the GPU dropout masks together
        # with (rank, step, layer) — ops/rng.py.  The reference never seeds (model/vgg16.py:40).
        if seed is not None:
            torch.manual_seed(int(seed))

        # T3/T6 model on its device first, then criterion / optimizer / scheduler
        self.model = self.build_model().to(self.device)
        rng = getattr(self.model, "dropout_rng", None)
        if rng is not None and seed is not None:
            rng.seed = int(seed)
        self.criterion = self.build_criterion()
        self.optimizer = self.build_optimizer()
        self.scheduler = self.build_scheduler()

        # T4 resume (state lands on the training device; quirk #4)
        if snapshot_path is not None:
            self._load_snapshot(snapshot_path)

        # gradient synchronisation (replaces DDP; broadcasts rank 0's state, X2)
        # shard_optimizer: reduce-scatter + per-bucket optimizer on 1/world of the arena + all-gather
        # (parallel/reducer.py) instead of all-reduce + replicated optimizer
        # bucket_optimizer (default): all-reduce as usual, but the fused optimizer updates each bucket's
        # range right behind its collective (native transport: on the communication stream) — the
        # 0.5 ms pass over 134 M parameters no longer runs un-overlapped after the last bucket.  Needs
        # a fused flat-arena optimizer; a captured hipGraph step keeps the replicated step (the
        # per-pass optimizer constants live on the host).
        sharded = bool(shard_optimizer) and self.world_size > 1
        # (world size 1 on a GPU: the same placement without communication — the slices run on a side
        # stream under the rest of backward; that form is capturable, the communicating ones are not)
        per_bucket = (bool(bucket_optimizer) and not sharded
                      and ((self.world_size > 1 and not use_hip_graph) or (self.world_size == 1 and self.device.type == "cuda"))
                      and hasattr(self.optimizer, "update_range")
                      and getattr(self.optimizer, "arena", None) is getattr(self.model, "param_arena", None)
                      and getattr(self.model, "param_arena", None) is not None)
        self.model = DataParallel(self.model, bucket_cap_mb=bucket_cap_mb, reduce_dtype=reduce_dtype,
                                  shard_optimizer=self.optimizer if sharded else None,
                                  bucket_optimizer=self.optimizer if per_bucket else None)
        if self.use_hip_graph and self.device.type == "cuda" and self.grad_accum_steps == 1:
            from ..engine.graph import GraphedTrainStep
            self.graphed_step = GraphedTrainStep(self.model, self.criterion, self.optimizer)
            if not self.graphed_step.enabled:
                self.log("use_hip_graph: not capturable here (world size > 1 without the native RCCL transport, "
                         "sharded optimizer, no parameter arena or an optimizer with per-step host arguments) "
                         "- running eagerly", "warning")
                self.graphed_step = None

        # T7/T8 data
        self.batch_size = batch_size
        self.local_batch_size = batch_size // self.world_size
        self.pin_memory = bool(pin_memory) and self.device.type == "cuda"
        self.num_workers = num_workers
        train_dataset = self.build_train_dataset()
        self.train_dataloader = self.build_dataloader(
            train_dataset, self.local_batch_size, self.pin_memory,
            collate_fn=train_dataset.collate_fn if callable(getattr(train_dataset, "collate_fn", None)) else None,
            phase="train")
        if self.have_validate:
            val_dataset = self.build_val_dataset()
            self.val_dataloader = self.build_dataloader(
                val_dataset, self.local_batch_size, self.pin_memory,
                collate_fn=val_dataset.collate_fn if callable(getattr(val_dataset, "collate_fn", None)) else None,
                phase="val")

    # ------------------------------------------------------------------ logging
    def log(self, msg, log_type="info"):
        if self.logger is not None:
            self.logger.log(msg, log_type)
        elif self.world_rank == 0:
            print("%s: %s" % (log_type.upper(), msg), flush=True)

    # ------------------------------------------------------------------ process group (T9)
    @staticmethod
    def ddp_setup(backend="nccl"):
        return penv.ddp_setup(backend)

    @staticmethod
    def destroy_process():
        penv.destroy_process()

    # ------------------------------------------------------------------ snapshots (T10, T4)
    @property
    def raw_model(self):
        return self.model.module if hasattr(self.model, "module") else self.model

    def _save_snapshot(self, epoch, name="last"):
        path = os.path.join(self.save_weight_folder, "%s.pth" % name)
        extra = dict(best_fitness=copy.deepcopy(self.best_fitness), global_step=self.global_step,
                     rng_state=self._rng_state(), framework="distributed_training_pytorch_amd")
        ckpt.save_snapshot(path, epoch, self.raw_model, self.optimizer, self.scheduler, extra)
        self.log("Saved model at epoch %d!" % epoch, log_type="info")

    def _load_snapshot(self, path):
        snap = ckpt.load_snapshot(path)
        self.cur_epoch = int(snap["epoch"])
        self.raw_model.load_state_dict(snap["model_state_dict"])
        if snap.get("optimizer_state_dict"):
            self.optimizer.load_state_dict(snap["optimizer_state_dict"])
        if snap.get("scheduler_state_dict") and self.scheduler is not None:
            self.scheduler.load_state_dict(snap["scheduler_state_dict"])
        bf = snap.get("best_fitness")
        if isinstance(bf, dict) and "epoch" in bf:
            self.best_fitness = bf
        self.global_step = int(snap.get("global_step", 0))
        self._load_rng_state(snap.get("rng_state"))
        self.log("Resumed from %s at epoch %d" % (path, self.cur_epoch), log_type="info")

    def _rng_state(self):
        """What makes a resumed run draw the numbers the uninterrupted run would have drawn: the
        dropout call counter of the model (ops/rng.py; rank 0's — all ranks advance in lock step) and
        torch's CPU generator.  Extra snapshot key: old readers ignore it."""
        state = dict(torch_cpu=torch.get_rng_state())
        rng = getattr(self.raw_model, "dropout_rng", None)
        if rng is not None:
            state.update(rng.state_dict())
        return state

    def _load_rng_state(self, state):
        if not isinstance(state, dict):
            return
        rng = getattr(self.raw_model, "dropout_rng", None)
        if rng is not None:
            rng.load_state_dict(state)
        t = state.get("torch_cpu")
        if torch.is_tensor(t) and t.dtype == torch.uint8:
            torch.set_rng_state(t.cpu())

    # ------------------------------------------------------------------ helpers for hooks
    def should_sync(self, micro_step):
        """True when this micro-step ends a gradient-accumulation window: every
        ``grad_accum_steps`` micro-steps, and at the LAST step of the epoch — a trailing short
        window is flushed (synchronised + stepped), never dropped."""
        if (micro_step + 1) % self.grad_accum_steps == 0:
            return True
        return self.steps_this_epoch is not None and micro_step + 1 >= self.steps_this_epoch

    def accum_window(self, micro_step):
        """Length of the gradient-accumulation window this micro-step belongs to (the loss of each
        of its micro-steps is divided by it): ``grad_accum_steps``, or what is left of the epoch."""
        A = self.grad_accum_steps
        if self.steps_this_epoch is None:
            return A
        start = micro_step - micro_step % A
        return max(1, min(A, self.steps_this_epoch - start))

    def _progress(self, iterable, desc=None):
        if self.progress and _tqdm is not None and self.world_rank == 0:
            return _tqdm(iterable, desc=desc, leave=False)
        return iterable

    def _maybe_inject_fault(self):
        """Kill-a-rank test hook: FAULT_INJECT_STEP / FAULT_INJECT_RANK."""
        s = os.environ.get("FAULT_INJECT_STEP")
        restarted = max(int(os.environ.get("TORCHELASTIC_RESTART_COUNT", "0") or 0),
                        int(os.environ.get("DTP_RESTART_COUNT", "0") or 0)) > 0
        if s is not None and not restarted and int(s) == self.global_step and \
                int(os.environ.get("FAULT_INJECT_RANK", "0")) == self.world_rank:
            self.log("fault injection: rank %d exits at step %d" % (self.world_rank, self.global_step), "error")
            os._exit(17)

    @staticmethod
    def _batch_len(batch):
        for e in batch if isinstance(batch, (list, tuple)) else [batch]:
            if torch.is_tensor(e) and e.dim() > 0:
                return int(e.shape[0])
        return 1

    def _reduce_means(self, sums, count):
        """Sample-weighted means over all ranks: one small all-reduce of (sums..., count).

        The metric names are agreed on first (union over ranks, fixed order): a rank whose shard
        was empty (``len(val) < world``) has no names of its own and must still send a vector of
        the same length as its peers — it contributes zeros and a zero count."""
        keys = list(sums.keys())
        if self.world_size > 1:
            seen = []
            for ks in comm.all_gather_object(keys):
                for k in ks:
                    if k not in seen:
                        seen.append(k)
            keys = seen
        vals = comm.all_reduce_scalars([_to_float(sums[k]) if k in sums else 0.0 for k in keys] + [float(count)])
        total = max(vals[-1], 1.0)
        return {k: v / total for k, v in zip(keys, vals[:-1])}

    # ------------------------------------------------------------------ training pipeline (T11)
    def train(self):
        for epoch in range(self.cur_epoch, self.max_epoch):
            self.cur_epoch = epoch

            # validation gate + best-model tracking (T12/T13): every rank validates its shard
            if self.have_validate and (epoch % self.save_period == 0):
                self.model.eval()
                metrics = self.validate()
                key, mode = self.save_best_for
                if key not in metrics:
                    raise KeyError("save_best_for metric %r not produced by validate_step (have %s)"
                                   % (key, sorted(metrics)))
                bf = self.best_fitness
                better = bf["epoch"] is None or (metrics[key] >= bf["value"] if mode == "geq"
                                                 else metrics[key] <= bf["value"])
                if better:
                    bf["epoch"], bf["value"], bf["metrics"] = epoch, metrics[key], copy.deepcopy(metrics)
                    self.model.consolidate()     # every rank takes this branch: metrics are all-reduced
                    if self.world_rank == 0:
                        self._save_snapshot(epoch, name="best")
                self.log(RULE)
                msg = "The BEST model is at EPOCH %s and has " % bf["epoch"]
                for k, v in bf["metrics"].items():
                    msg += " | %s = %s | " % (k.upper(), v)
                self.log(msg)
                comm.barrier()

            # train one epoch
            self.model.train()
            sampler = getattr(self.train_dataloader, "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            ds = getattr(self.train_dataloader, "dataset", None)
            if hasattr(ds, "set_epoch"):         # per-epoch augmentation streams (data/image_folder.py)
                ds.set_epoch(epoch)
            self.log(RULE)
            self.log("[GPU%d] Epoch %d/%d" % (self.world_rank, epoch + 1, self.max_epoch))
            sums, count = None, 0
            self.meter.start()
            try:
                self.steps_this_epoch = len(self.train_dataloader)
                if self.max_steps_per_epoch is not None:
                    self.steps_this_epoch = min(self.steps_this_epoch, self.max_steps_per_epoch)
            except TypeError:       # a loader without a length: windows cannot be flushed at the end
                self.steps_this_epoch = None
            prefetcher = DevicePrefetcher(self.train_dataloader, self.device)
            starved0 = prefetcher.starved_ms_total()
            loop = self._progress(prefetcher)
            it, i = iter(loop), -1
            data_wait_s = step_host_s = 0.0
            reducer = getattr(self.model, "reducer", None)
            exposed0 = reducer.exposed_ms_total() if reducer is not None else 0.0
            while True:
                t0 = time.perf_counter()
                with trace_range("data_wait"):
                    batch = next(it, None)
                if batch is None:
                    break
                i += 1
                if self.max_steps_per_epoch is not None and i >= self.max_steps_per_epoch:
                    break
                t1 = time.perf_counter()
                data_wait_s += t1 - t0
                self._maybe_inject_fault()
                self.micro_step = i
                with trace_range("train_step"):
                    loss = self.train_step(batch)
                step_host_s += time.perf_counter() - t1
                n = self._batch_len(batch)
                if sums is None:
                    sums = {k: (v.detach().float() if torch.is_tensor(v) else float(v)) * n for k, v in loss.items()}
                else:
                    for k, v in loss.items():
                        sums[k] = sums[k] + (v.detach().float() if torch.is_tensor(v) else float(v)) * n
                count += n
                self.meter.add(n * self.world_size)
                self.global_step += 1
                if hasattr(loop, "set_postfix") and (i % self.log_every == 0):
                    loop.set_postfix({k: _to_float(v) for k, v in loss.items()})
            steps_done = max(1, i + 1 if batch is None else i)

            # LR schedule (reference :159-160)
            if self.scheduler is not None:
                self.scheduler.step()
                self.log("THE NEXT LEARNING RATE VALUE IS %s" % self.scheduler.get_last_lr()[0])

            # snapshots (reference :163-172): rank 0 writes, everyone waits
            if self.have_validate or epoch % self.save_period == 0:
                self.model.consolidate()         # collective: sharded optimizer state -> complete on every rank
            if self.world_rank == 0:
                if self.have_validate:
                    self._save_snapshot(epoch + 1, name="last")
                elif epoch % self.save_period == 0:
                    self._save_snapshot(epoch + 1, name="checkpoint_epoch_%d" % (epoch + 1))
            comm.barrier()

            # epoch summary: local (reference string) and global sample-weighted means
            if sums is not None:
                local = {k: _to_float(v) / max(count, 1) for k, v in sums.items()}
                msg = "TOTAL LOCAL TRAINING LOSS: "
                for k, v in local.items():
                    msg += " | %s = %s | " % (k, v)
                self.log(msg)
                glob = self._reduce_means(sums, count)
                ips = self.meter.stop(sync_device=self.device)
                if self.world_size > 1:
                    msg = "TOTAL GLOBAL TRAINING LOSS: "
                    for k, v in glob.items():
                        msg += " | %s = %s | " % (k, v)
                    self.log(msg)
                self.log("THROUGHPUT: %.1f images/sec (%d images, %.2fs)" % (ips, self.meter.items, self.meter.elapsed))
                exposed = (reducer.exposed_ms_total() - exposed0) if reducer is not None else 0.0
                phases = self.step_timer.summary()
                self.step_timer.reset()
                self._write_metrics(dict(
                    epoch=epoch + 1, images_per_sec=ips, steps=steps_done,
                    **{"gpu_%s_ms_per_step" % k: v["total_ms"] / steps_done for k, v in phases.items()},
                    # host blocked in next(loader): starvation OR back-pressure (a host that runs ahead
                    # of the GPU waits here for a free slot of the copy pipeline most of every step)
                    data_wait_ms_per_step=1e3 * data_wait_s / steps_done,
                    # GPU-side: the compute stream idle because its batch's copy had not landed
                    data_starved_ms_per_step=(prefetcher.starved_ms_total() - starved0) / steps_done,
                    host_step_ms_per_step=1e3 * step_host_s / steps_done,      # host time inside train_step
                    allreduce_exposed_ms_per_step=exposed / steps_done,        # GPU idle waiting for buckets
                    **{"train_" + k: v for k, v in glob.items()}))

        self.log("Finished!")

    def _write_metrics(self, rec):
        if self.metrics_file and self.world_rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(self.metrics_file)), exist_ok=True)
            rec = dict(rec, time=time.time(), world_size=self.world_size)
            with open(self.metrics_file, "a") as f:
                f.write(json.dumps(rec) + "\n")

    # ------------------------------------------------------------------ validation (T12)
    @torch.no_grad()
    def validate(self):
        sums, count = None, 0
        loop = self._progress(DevicePrefetcher(self.val_dataloader, self.device), desc="val")
        for batch in loop:
            batch_metrics = self.validate_step(batch)
            n = self._batch_len(batch)
            if sums is None:
                sums = {k: (v.detach().float() if torch.is_tensor(v) else float(v)) * n for k, v in batch_metrics.items()}
            else:
                for k, v in batch_metrics.items():
                    sums[k] = sums[k] + (v.detach().float() if torch.is_tensor(v) else float(v)) * n
            count += n
        if sums is None:
            sums = {}
        avg_metrics = self._reduce_means(sums, count) if (sums or self.world_size > 1) else {}
        msg = "VALIDATE RESULTS: "
        for k, v in avg_metrics.items():
            msg += " | %s = %s | " % (k, v)
        self.log(msg)
        self._write_metrics(dict(epoch=self.cur_epoch, **{"val_" + k: v for k, v in avg_metrics.items()}))
        return avg_metrics

    # ------------------------------------------------------------------ dataloader factory (T8)
    def build_dataloader(self, dataset, batch_size, pin_memory, collate_fn=None, phase="train"):
        if phase == "train":
            sampler = ShardedSampler(len(dataset), self.world_rank, self.world_size, shuffle=True, seed=self.seed)
        else:
            sampler = ShardedSampler(len(dataset), self.world_rank, self.world_size, shuffle=False, pad=False)
        if (self.cache_dataset_on_device and self.device.type == "cuda" and collate_fn is None
                and DeviceCachedLoader.fits(dataset, self.device)):
            try:
                # un-augmented samples in HBM; training-phase augmentation runs on the device per batch
                return DeviceCachedLoader(dataset, batch_size, sampler, self.device)
            except ValueError as e:      # a CPU-augmenting dataset without raw access: never freeze its draws
                self.log("cache_dataset_on_device refused: %s - using the DataLoader" % e, "warning")
        return DataLoader(dataset, batch_size=batch_size, pin_memory=pin_memory, shuffle=False,
                          sampler=sampler, collate_fn=collate_fn, num_workers=self.num_workers,
                          # datasets with per-epoch state (set_epoch) get fresh workers every epoch so
                          # that the workers' copies see it
                          persistent_workers=self.num_workers > 0 and not hasattr(dataset, "set_epoch"),
                          drop_last=False)

    # ------------------------------------------------------------------ hooks (T14)
    def build_train_dataset(self):
        raise NotImplementedError("Please implement the build_train_dataset method before calling")

    def build_val_dataset(self):
        raise NotImplementedError("Please implement the build_val_dataset method before calling")

    def build_model(self):
        raise NotImplementedError("Please implement the build_model method before calling")

    def build_criterion(self):
        raise NotImplementedError("Please implement the build_criterion method before calling")

    def build_optimizer(self):
        raise NotImplementedError("Please implement the build_optimizer method before calling")

    def build_scheduler(self):
        raise NotImplementedError("Please implement the build_scheduler method before calling")

    def preprocess_batch(self, batch):
        raise NotImplementedError("Please implement the preprocess_batch method before calling")

    def train_step(self, batch):
        raise NotImplementedError("Please implement the train_step method before calling")

    def validate_step(self, batch):
        raise NotImplementedError("Please implement the validate_step method before calling")
