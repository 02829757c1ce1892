"""Base class / helper that turns ANY ``nn.Module`` built from the fused layers of ``ops.layers``
(``ConvFirst``, ``Conv3x3``, ``MaxPool2x2``, ``AvgPoolFlatten``, ``Linear`` ...) into a model the
kernels, the fused optimizers, the reducer and the hipGraph step can run.

The reference's ``Trainer`` is model-agnostic — ``build_model()`` is a hook
(``trainer/trainer.py:220-253``) and VGG16 is only the worked example — so the machinery VGG16
needs (flat parameter arena, re-homing after ``.to(device)``, bf16 working copy refresh,
contiguous ``state_dict`` entries) lives here and is reusable:

    class TinyNet(FusedModel):
        def __init__(self):
            super().__init__()
            self.c1 = ConvFirst(3, 64, relu=True, grad_premasked=True)
            self.pool = MaxPool2x2(relu_mask=True)
            self.flat = AvgPoolFlatten((7, 7))
            self.fc = Linear(64 * 49, 10, out_fp32=True)
            self.finalize()                      # builds the arena; call last in __init__

        def forward(self, x):
            x = self.prepare_input(x)            # refreshes the bf16 weight copy when needed
            return self.fc(self.flat(self.pool(self.c1(x))))

``attach_arena(module)`` does the same for an already-constructed plain ``nn.Module`` (one shot:
call it again after moving the module to another device).
"""
import os

import torch
import torch.nn as nn

from ..ops.layers import FusedLayer
from ..ops.rng import DropoutRng
from .param_arena import ParamArena


def attach_arena(module, grad_dtype=torch.float32):
    """Re-home ``module``'s parameters in a fresh :class:`ParamArena` and bind every fused layer
    to it.  Returns the arena (also stored as ``module.param_arena``)."""
    arena = ParamArena(module, grad_dtype=grad_dtype)
    object.__setattr__(module, "param_arena", arena)
    on_gpu = arena.device.type == "cuda"
    rng = getattr(module, "dropout_rng", None)
    if rng is None:
        rng = DropoutRng()
        object.__setattr__(module, "dropout_rng", rng)
    for index, m in enumerate(module.modules()):
        if isinstance(m, FusedLayer):
            object.__setattr__(m, "_arena", arena)
            if hasattr(m, "_seed_salt"):          # dropout layers: key = f(run rng, position in the model)
                m._rng, m._seed_salt = rng, DropoutRng.salt_for(index)
            if on_gpu:
                for p in m.parameters(recurse=False):
                    arena.kernel_written.add(id(p))
            m._derived, m._derived_epoch = {}, -1
    return arena


def _contiguous_state_dict_hook(module, state_dict, prefix, local_metadata):
    """Emit standalone, contiguous fp32 tensors (not views of the arena)."""
    for k in list(state_dict.keys()):
        v = state_dict[k]
        if torch.is_tensor(v) and k.startswith(prefix):
            state_dict[k] = v.detach().clone(memory_format=torch.contiguous_format)
    return state_dict


class FusedModel(nn.Module):
    def __init__(self, grad_dtype=torch.float32):
        super().__init__()
        self.grad_dtype = grad_dtype
        self.param_arena = None
        self._finalized = False
        # dropout masks are a function of (seed, rank, training-forward number, layer): ops/rng.py;
        # the Trainer seeds it, snapshots carry it
        self.dropout_rng = DropoutRng()

    def finalize(self):
        """Call at the end of ``__init__``: all parameters exist, build the arena."""
        self._register_state_dict_hook(_contiguous_state_dict_hook)
        self._finalized = True
        self._attach_arena()
        return self

    def _attach_arena(self):
        attach_arena(self, self.grad_dtype)

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if self._finalized and not self._still_in_arena():
            self._attach_arena()     # parameters were re-created/moved: re-home them in a fresh arena
            self._after_rehome()
        return out

    def _still_in_arena(self):
        """True when ``_apply`` was a no-op for the parameters (``.to()`` of the device/dtype they
        already have): optimizers, reducers and captured graphs hold views of THIS arena, so it
        must not be rebuilt behind their back."""
        a = self.param_arena
        if a is None:
            return False
        base = a.flat_master.untyped_storage().data_ptr()
        for p in a.params:
            if (p.device != a.device or p.dtype != torch.float32
                    or p.data.untyped_storage().data_ptr() != base):
                return False
        return len(a.params) == sum(1 for q in self.parameters() if q.requires_grad)

    def _after_rehome(self):
        """Hook for subclasses that keep device-side constants (e.g. input normalisation)."""

    def set_grad_dtype(self, dtype):
        self.grad_dtype = dtype
        self._attach_arena()

    def prepare_input(self, x):
        if self.training:
            self.dropout_rng.advance()       # one call number per training forward (micro-step)
        if x.is_cuda and self.param_arena is not None:
            self.param_arena.refresh_lp()
            from ..ops import layers as L
            L._pending_join.clear()          # a backward pass that raised must not block future joins
            del L._keepalive[:]
            if self.training and torch.is_grad_enabled():
                self._prefetch_derived(x.device)
        return x

    def _prefetch_derived(self, device):
        """Build the per-step derived weight layouts (dgrad-order conv filters, W^T of the linear
        layers: ~15 small transposes) on a side stream while the forward pass runs, instead of
        lazily inside backward on the critical path."""
        arena = self.param_arena
        if getattr(self, "_derived_epoch", None) == arena.lp_epoch:
            return
        self._derived_epoch = arena.lp_epoch
        from ..ops import layers as L
        if not L.OVERLAP_WGRAD or os.environ.get("DTP_PREFETCH_DERIVED", "1") == "0":
            return
        side = L._side_stream(device)
        side.wait_stream(torch.cuda.current_stream())
        first = True
        with torch.cuda.stream(side):
            plan = self._derived_transpose_plan(arena)
            if plan is not None:                      # every plain transpose of the model: ONE launch
                plan.run()
                for m, key, out in plan.outputs:
                    m._set_derived(key, out)
            for m in self.modules():
                if isinstance(m, FusedLayer):
                    for key, builder in m.derived_builders(needs_input_grad=not first):
                        m._cached(key, builder)
                    first = False if any(True for _ in m.parameters(recurse=False)) else first
            ev = torch.cuda.Event()
            ev.record(side)
        arena.derived_event = ev       # the first backward function that needs a layout waits on it

    def _derived_transpose_plan(self, arena):
        """Job table over the derived layouts that are plain transposes (``FusedLayer.derived_transposes``),
        built once per arena: persistent destinations, one ``transpose_multi`` launch per optimizer step."""
        if os.environ.get("DTP_MULTI_TRANSPOSE", "1") == "0":
            return None
        key = (id(arena), arena.flat_lp.data_ptr())
        cached = self.__dict__.get("_transpose_plan")
        if cached is not None and cached[0] == key:
            return cached[1]
        from ..ops import kernels as K
        jobs, outputs, first = [], [], True
        for m in self.modules():
            if isinstance(m, FusedLayer):
                for k, out, job in m.derived_transposes(needs_input_grad=not first):
                    jobs.append(job)
                    outputs.append((m, k, out))
                first = False if any(True for _ in m.parameters(recurse=False)) else first
        plan = None
        if jobs:
            plan = K.TransposePlan(jobs, arena.device)
            plan.outputs = outputs
        self.__dict__["_transpose_plan"] = (key, plan)
        return plan

    def finish_forward(self, y):
        """Optional: join the side stream of ``_prefetch_derived`` at the end of forward (the
        backward functions wait on its event themselves, see ``ParamArena.wait_derived``)."""
        if y.is_cuda and self.param_arena is not None:
            self.param_arena.wait_derived()
        return y
