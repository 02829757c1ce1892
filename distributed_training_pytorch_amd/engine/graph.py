"""Whole-step hipGraph capture: ``zero_grad + forward + loss + backward + optimizer.step`` of a
fused model becomes ONE ``hipGraphLaunch``.

Why: at the reference's default per-GPU batch (4, ``main.py:15`` / ``run.sh:10``) a VGG16 step is
~150 kernel launches of a few microseconds each — the GPU idles between them while Python,
autograd and ctypes issue the next one (the reference never captures anything).
All kernels of this framework take explicit HIP streams and static workspaces, so the whole step
is capturable with ``torch.cuda.graph`` (PyTorch-ROCm records it into a hipGraph).

What is frozen at capture and how it is kept correct on replay:
  * kernel arguments (learning rate, momentum, beta flags, tile choices): the step is re-captured
    when the optimizer's hyper-parameters change (e.g. a MultiStepLR milestone);
  * the dropout seed: read from device memory (``Linear.device_seed``) and advanced by a captured
    add, so every replay draws a fresh mask;
  * input batches: copied into static device buffers before each replay;
  * the bf16 working copy of the weights: written by the captured fused optimizer; a
    ``load_state_dict`` / external parameter edit is detected through the parameters' version
    counters and triggers a re-capture.
Data-parallel runs: the step is captured INCLUDING the gradient all-reduce when the reducer runs on
the native RCCL transport (``BucketedReducer.graph_capturable``): its C++ scheduler forks the
communication stream into the capture with an event record / wait pair per bucket and joins it at the
end of backward, RCCL's collectives are recorded as kernel nodes, and a replay is one launch per rank.
All ranks capture at the same step (same warm-up count, same hyper-parameter key), so the collective
sequence stays matched.  Capture uses ``capture_error_mode="thread_local"``: RCCL's proxy thread makes
HIP calls of its own while this thread records.  With the torch process-group transport, the sharded
optimizer or gloo the step stays eager.  Verified on one GPU (world-size-1 RCCL communicator: eager ==
graph bit for bit, tests/test_gpu_model.py); multi-GPU replays are exercised by the driver's runs only
when ``--hip-graph`` is passed — it is opt-in.
"""
import torch

from ..ops import rng as _rng
from ..ops.layers import Linear


class GraphedTrainStep(object):
    _SEED_STEP = _rng.STEP

    def __init__(self, model, criterion, optimizer, warmup=2, enabled=True):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.raw = getattr(model, "module", model)
        self.arena = getattr(self.raw, "param_arena", None)
        self.warmup = int(warmup)
        self.graph = None
        self._key = None
        self.replays = 0
        self.captures = 0
        reducer = getattr(model, "reducer", None) or getattr(self.arena, "reducer", None)
        world = getattr(reducer, "world", 1)
        self.distributed = reducer is not None and bool(getattr(reducer, "graph_capturable", False))
        self.enabled = (bool(enabled) and torch.cuda.is_available() and self.arena is not None
                        and (world == 1 or self.distributed) and getattr(optimizer, "graph_safe", False))
        self._seed = None
        self._seed_calls = 0
        self.rng = getattr(self.raw, "dropout_rng", None) or _rng.DropoutRng()
        self._eager_left = 0

    # ------------------------------------------------------------------ helpers
    def _hyper_key(self, x, y):
        groups = tuple(tuple(sorted((k, repr(v)) for k, v in g.items() if k != "params"))
                       for g in self.optimizer.param_groups)
        return (tuple(x.shape), x.dtype, tuple(y.shape), y.dtype, groups, self.raw.training)

    def _eager(self, x, y):
        self.optimizer.zero_grad()
        loss = self.criterion(self.model(x), y)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _attach_seed(self, device):
        if self._seed is None:
            # device copy of the model's dropout state (ops/rng.py): base + calls * STEP.  From here on
            # the captured add advances it; the host counter follows in __call__ so that snapshots and
            # the eager path see the same number
            self._seed = torch.zeros(1, dtype=torch.int64, device=device)
            self._seed.fill_(self.rng.device_value())
            self._seed_calls = self.rng.calls
        elif self._seed_calls != self.rng.calls:         # a snapshot was loaded / someone ran the model eagerly
            self._seed.fill_(self.rng.device_value())
            self._seed_calls = self.rng.calls
        for m in self.raw.modules():
            if isinstance(m, Linear):
                m.device_seed = self._seed

    def _capture(self, x, y):
        self.x_static = torch.empty_like(x)
        self.y_static = torch.empty_like(y)
        self.x_static.copy_(x)
        self.y_static.copy_(y)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        steps_before = getattr(self.optimizer, "_steps", None)
        kw = dict(capture_error_mode="thread_local") if self.distributed else {}
        with torch.cuda.graph(g, **kw):
            self._seed.add_(self._SEED_STEP)
            self.loss_static = self._eager(self.x_static, self.y_static)
        if steps_before is not None:       # capture ran the Python of step() once without executing it
            self.optimizer._steps = steps_before
        self.graph = g
        self.captures += 1
        self._versions = self.arena._versions()

    # ------------------------------------------------------------------ the step
    def __call__(self, x, y):
        """One optimizer step on batch (x, y); returns the (device-resident) loss.  The first
        ``warmup`` calls after a (re)configuration run eagerly — they size every workspace, build
        the derived weight layouts and take the optimizer past its state-initialising step — the
        next call captures, and every later one is a single graph launch."""
        if not self.enabled or not x.is_cuda:
            return self._eager(x, y)
        key = self._hyper_key(x, y)
        if key != self._key or (self.graph is not None and self.arena._versions() != self._versions):
            self.graph, self._key, self._eager_left = None, key, max(1, self.warmup)
            self._seed_calls = -1                      # re-derive the device copy of the dropout state
        self._attach_seed(x.device)
        if self.graph is None:
            if self._eager_left > 0:
                self._eager_left -= 1
                self._seed.add_(self._SEED_STEP)
                loss = self._eager(x, y)               # (the model's forward advanced rng.calls)
                self._seed_calls = self.rng.calls
                return loss
            self.arena.refresh_lp()
            self._capture(x, y)                        # ran the forward's Python once: rng.calls advanced
        else:
            self.x_static.copy_(x, non_blocking=True)
            self.y_static.copy_(y, non_blocking=True)
            if self.raw.training:
                self.rng.advance()                     # the replayed add advances the device copy
        self._seed_calls = self.rng.calls
        self.graph.replay()
        self.replays += 1
        if self.raw.training and hasattr(self.model, "mark_buffers_dirty"):
            # a replay bypasses DataParallel.forward: BatchNorm running statistics moved all the same,
            # the lazy buffer broadcast (eval() / consolidate()) must not skip them
            self.model.mark_buffers_dirty()
        # host-side bookkeeping that the eager path does per step
        if hasattr(self.optimizer, "_steps"):
            self.optimizer._steps += 1
        self.arena.lp_epoch += 1
        return self.loss_static
