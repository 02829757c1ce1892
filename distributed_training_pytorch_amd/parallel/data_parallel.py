"""``DataParallel`` — the module wrapper the Trainer uses instead of stock DDP.

Parity target: reference ``trainer/trainer.py:52`` (``DDP(model, device_ids=[local_rank], ...)``),
``:88`` (``self.model.module.state_dict()``) and the hook code that calls ``self.model(img)``
(``example_trainer.py:82,97``): the wrapper exposes ``.module``, is callable, follows
``train()/eval()`` and offers ``no_sync()``.

What happens at construction: rank 0's parameters and buffers are broadcast
as a few flat messages; then a :class:`BucketedReducer` takes over gradient synchronisation.
Buffers (BatchNorm running stats of VGG16-BN) are re-broadcast from rank 0 at the start of a
*training* forward only — never in eval — which removes the rank-0-only-validation deadlock of
SURVEY.md §2.6 quirk #9.
"""
import torch
import torch.nn as nn

from . import comm
from .reducer import BucketedReducer, TorchPGTransport


class DataParallel(nn.Module):
    def __init__(self, module, bucket_cap_mb=None, first_bucket_mb=None, split_large_mb=None,
                 reduce_dtype=None, broadcast_buffers=True, transport=None, sync_init=True):
        super().__init__()
        self.module = module
        self.broadcast_buffers = broadcast_buffers
        arena = getattr(module, "param_arena", None)
        params = [p for p in module.parameters() if p.requires_grad]
        if transport is None:
            transport = self._pick_transport(params[0].device if params else torch.device("cpu"))
        kw = dict(bucket_cap_mb=bucket_cap_mb, first_bucket_mb=first_bucket_mb,
                  split_large_mb=split_large_mb, reduce_dtype=reduce_dtype, transport=transport)
        if arena is not None and arena.is_flat:
            self.reducer = BucketedReducer(arena.params, flat_grad=arena.flat_grad,
                                           flat_offsets=arena.offsets, **kw)
            arena.reducer = self.reducer
            self.reducer.pre_ready = arena.sync_autograd_grad
        else:
            self.reducer = BucketedReducer(params, **kw)
        if sync_init and self.reducer.world > 1:
            self.sync_module_states()

    @staticmethod
    def _pick_transport(device):
        """Native RCCL transport (own communicator + comm stream, C++) on GPUs, the
        ``torch.distributed`` process group otherwise.  ``DTP_TRANSPORT=torch|native`` forces one (default: native when it comes up everywhere).
        All ranks must end up with the same kind: the outcome of the native set-up (communicator
        creation plus one self-checked all-reduce) is agreed on through the process group before
        anyone uses it."""
        import os
        want = os.environ.get("DTP_TRANSPORT", "auto").lower()
        if device.type != "cuda" or not comm.is_dist() or comm.world_size() == 1 or want == "torch":
            return TorchPGTransport()
        # The set-up runs in a watchdog thread: a communicator that never comes up (a rank stuck in
        # ncclCommInitRank or in the probe collective) must degrade to the process-group transport
        # after DTP_NATIVE_INIT_TIMEOUT_S seconds instead of hanging the job.
        import threading
        box = dict(native=None, ok=0.0)

        def _bring_up():
            try:
                torch.cuda.set_device(device)                 # current device is per thread
                from .rccl_native import NativeRcclTransport
                box["native"] = NativeRcclTransport(device=device.index if device.index is not None
                                                    else torch.cuda.current_device())
                probe = torch.full((256,), float(comm.rank() + 1), device=device)
                box["native"].wait(box["native"].launch(probe), probe)
                torch.cuda.current_stream().synchronize()
                expect = (comm.world_size() + 1) / 2.0        # mean of 1..world
                box["ok"] = 1.0 if bool(((probe - expect).abs() < 1e-5).all()) else 0.0
            except Exception:
                box["ok"] = 0.0

        th = threading.Thread(target=_bring_up, name="dtp-native-rccl-init", daemon=True)
        th.start()
        th.join(float(os.environ.get("DTP_NATIVE_INIT_TIMEOUT_S", "120")))
        timed_out = th.is_alive()
        native, ok = (None, 0.0) if timed_out else (box["native"], box["ok"])
        agreed = comm.all_reduce_scalars([ok], op="min")[0]
        if agreed >= 1.0:
            return native
        if native is not None:
            native.close()
        if comm.rank() == 0:
            import warnings
            warnings.warn("native RCCL transport unavailable (%s on at least one rank); gradients go through the "
                          "torch.distributed process group" % ("bring-up timed out" if timed_out else
                                                               "bring-up or self-check failed" if ok < 1.0 else
                                                               "another rank failed"))
        if want == "native":
            raise RuntimeError("DTP_TRANSPORT=native but the RCCL transport did not come up on every rank")
        return TorchPGTransport()

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

    def forward(self, *args, **kwargs):
        if (self.broadcast_buffers and self.training and self.reducer.world > 1
                and self.reducer.require_sync):
            bufs = [b for b in self.module.buffers() if b.numel() > 0 and b.is_floating_point()
                    and getattr(b, "_dtp_sync", True)]
            if bufs:
                self.reducer.broadcast_parameters(bufs)
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
