"""``DataParallel`` — the module wrapper the Trainer uses instead of stock DDP.

Parity target: reference ``trainer/trainer.py:52`` (``DDP(model, device_ids=[local_rank], ...)``),
``:88`` (``self.model.module.state_dict()``) and the hook code that calls ``self.model(img)``
(``example_trainer.py:82,97``): the wrapper exposes ``.module``, is callable, follows
``train()/eval()`` and offers ``no_sync()``.

What happens at construction: rank 0's parameters and buffers are broadcast
as a few flat messages; then a :class:`BucketedReducer` takes over gradient synchronisation.
Buffers (BatchNorm running stats of VGG16-BN) follow rank 0 like under DDP, but lazily: they are
broadcast when the wrapper switches to eval mode and before a snapshot (``sync_buffers``), not at
every training forward — and never from inside an eval forward, which removes the
rank-0-only-validation deadlock of SURVEY.md §2.6 quirk #9.
"""
import torch
import torch.nn as nn

from . import comm
from .reducer import BucketedReducer, TorchPGTransport


class DataParallel(nn.Module):
    def __init__(self, module, bucket_cap_mb=None, first_bucket_mb=None, split_large_mb=None,
                 reduce_dtype=None, broadcast_buffers=True, transport=None, sync_init=True,
                 shard_optimizer=None, bucket_align_elems=None, bucket_optimizer=None):
        super().__init__()
        self.module = module
        self.broadcast_buffers = broadcast_buffers
        arena = getattr(module, "param_arena", None)
        self._arena = arena
        self._buffers_dirty = False
        params = [p for p in module.parameters() if p.requires_grad]
        if transport is None:
            transport = self._pick_transport(params[0].device if params else torch.device("cpu"))
        kw = dict(bucket_cap_mb=bucket_cap_mb, first_bucket_mb=first_bucket_mb,
                  split_large_mb=split_large_mb, reduce_dtype=reduce_dtype, transport=transport)
        if (shard_optimizer is not None or bucket_optimizer is not None) and not (arena is not None and arena.is_flat):
            raise ValueError("shard_optimizer / bucket_optimizer need a model with a flat ParamArena")
        if arena is not None and arena.is_flat:
            self.reducer = BucketedReducer(arena.params, flat_grad=arena.flat_grad,
                                           flat_offsets=arena.offsets, shard_optimizer=shard_optimizer,
                                           bucket_align_elems=bucket_align_elems,
                                           bucket_optimizer=bucket_optimizer, **kw)
            arena.reducer = self.reducer
            self.reducer.pre_ready = arena.sync_autograd_grad
        else:
            self.reducer = BucketedReducer(params, **kw)
        if sync_init and self.reducer.world > 1:
            self.sync_module_states()

    @staticmethod
    def _pick_transport(device):
        """Native RCCL transport (own communicator + comm stream, C++) on GPUs, the
        ``torch.distributed`` process group otherwise.  ``DTP_TRANSPORT=torch|native`` forces one
        (default ``auto``: native when it comes up on every rank).  All ranks end up with the same
        kind: ``rccl_native.bring_up`` runs on the calling thread and agrees on the outcome through
        the process group before anyone uses the transport."""
        import os
        want = os.environ.get("DTP_TRANSPORT", "auto").lower()
        info = dict(requested=want, kind="torch", fell_back=os.environ.get("DTP_NATIVE_FELL_BACK") == "1",
                    reason=os.environ.get("DTP_NATIVE_FALLBACK_REASON",
                                          "relaunched by the self-launcher after a native bring-up timeout")
                    if os.environ.get("DTP_NATIVE_FELL_BACK") == "1" else None)
        if device.type != "cuda" or not comm.is_dist() or comm.world_size() == 1 or want == "torch":
            t = TorchPGTransport()
            t.info = info
            return t
        from .rccl_native import bring_up
        native, info = bring_up(device, want)
        t = native if native is not None else TorchPGTransport()
        t.info = info
        return t

    def sync_module_states(self):
        arena = getattr(self.module, "param_arena", None)
        if arena is not None and arena.is_flat:
            self.reducer.broadcast_parameters([arena.flat_master])
            arena.mark_master_updated()
        else:
            self.reducer.broadcast_parameters([p.data for p in self.module.parameters()])
        bufs = [b for b in self.module.buffers() if b.numel() > 0 and b.is_floating_point()]
        if bufs:
            self.reducer.broadcast_parameters(bufs)

    def consolidate(self):
        """Collective (all ranks): complete the fp32 master weights and the optimizer state before
        ``state_dict()`` when the optimizer is sharded; a no-op otherwise."""
        self.reducer.consolidate()
        if self.broadcast_buffers:
            self.sync_buffers()

    def sync_buffers(self):
        """Collective: rank 0's floating-point buffers (BatchNorm running statistics) to every rank.
        Stock DDP re-broadcasts them at the start of EVERY training forward (reference
        ``trainer/trainer.py:52``); nothing reads the other ranks' copies during
        training, so here they are synchronised lazily — when the wrapper switches to eval mode and
        before a snapshot — with the same values at every point where they are observable."""
        if self.reducer.world > 1 and self._buffers_dirty:
            bufs = [b for b in self.module.buffers() if b.numel() > 0 and b.is_floating_point()
                    and getattr(b, "_dtp_sync", True)]
            if bufs:
                self.reducer.broadcast_parameters(bufs)
                for m in self.module.modules():          # caches keyed on the statistics (folded conv+BN weights)
                    if hasattr(m, "stats_epoch"):
                        m.stats_epoch += 1
        self._buffers_dirty = False

    def train(self, mode=True):
        """NOTE: ``eval()`` / ``train(False)`` is a COLLECTIVE when the module has floating-point
        buffers (BatchNorm running statistics) and world size > 1: every rank must call it (the
        Trainer does).  ``module.eval()`` on the wrapped module switches one rank without
        communication — at the price of that rank reading its own, unsynchronised statistics."""
        if not mode and self.broadcast_buffers and getattr(self, "reducer", None) is not None:
            self.sync_buffers()
        return super().train(mode)

    def mark_buffers_dirty(self):
        """Engines that run the training forward without going through :meth:`forward` (hipGraph
        replays, engine/graph.py) report here that running statistics moved."""
        self._buffers_dirty = True

    def forward(self, *args, **kwargs):
        if getattr(self.module, "param_arena", None) is not self._arena:
            raise RuntimeError("the model's ParamArena was rebuilt after DataParallel wrapped it (model.to() / "
                               ".float() on the wrapped module): the reducer's buckets point at the old "
                               "gradient arena — wrap the model after it is on its final device")
        if self.training:
            self._buffers_dirty = True       # running statistics move: re-sync before anyone reads them
        return self.module(*args, **kwargs)

    def no_sync(self):
        return self.reducer.no_sync()

    def close(self):
        """Release the gradient transport (native RCCL communicator + stream); call before
        ``destroy_process_group`` at the end of a run."""
        self.reducer.remove_hooks()
        tr = self.reducer.transport
        if hasattr(tr, "close"):
            tr.close()

    def state_dict(self, *args, **kwargs):
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        return self.module.load_state_dict(*args, **kwargs)
