"""Fused flat-arena optimizers: ONE kernel launch per step for the whole model.

Parity target: reference ``example_trainer.py:62`` builds ``optim.SGD(model.parameters(), lr=0.1,
momentum=0.9, weight_decay=1e-4)`` and ``:88`` calls ``optimizer.step()`` (torch's foreach path,
32 tensors); ``:78`` calls ``optimizer.zero_grad()``.

``FusedSGD`` / ``FusedAdamW`` subclass ``torch.optim.Optimizer`` so LR schedulers
(``MultiStepLR``, reference ``example_trainer.py:66``) and ``state_dict()`` /
``load_state_dict()`` work unchanged: per-parameter state entries (``momentum_buffer`` — torch
SGD's schema; ``exp_avg``/``exp_avg_sq``/``step`` — torch AdamW's) are views of
flat state buffers and are serialised as standalone contiguous tensors.

GPU: ``csrc/optim.hip`` (fp32 master update + bf16 working-copy emission fused).  CPU: the same
update rule on the flat buffers with a handful of torch ops (tested equal to torch.optim).
"""
import torch

from . import kernels as K


def _arena_from(model_or_arena):
    a = getattr(model_or_arena, "param_arena", None)
    if a is None and hasattr(model_or_arena, "module"):
        a = getattr(model_or_arena.module, "param_arena", None)
    if a is None and hasattr(model_or_arena, "flat_master"):
        a = model_or_arena
    if a is None:
        raise ValueError("fused optimizers need a model with a ParamArena (e.g. models.VGG16)")
    return a


class _FusedBase(torch.optim.Optimizer):
    _state_keys = ()
    graph_safe = False     # True: step() passes nothing that changes from step to step by value

    def __init__(self, model, defaults):
        self.arena = _arena_from(model)
        import weakref
        owner = model if getattr(model, "param_arena", None) is not None else getattr(model, "module", None)
        self._model_ref = weakref.ref(owner) if owner is not None and hasattr(owner, "param_arena") else None
        super().__init__(self.arena.params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("fused optimizers support a single parameter group")
        self._flat_state = {k: torch.zeros_like(self.arena.flat_master) for k in self._state_keys}
        self._steps = 0
        # sharded mode (parallel/reducer.py, ``shard_optimizer``): the gradient reducer runs
        # ``update_range`` on this rank's slice of every bucket between the bucket's reduce-scatter
        # and its all-gather; ``step()`` then only closes the pass
        self._sharded_by = None
        self._external_pass_done = False
        self._bind_state_views()

    def _check_arena(self):
        """The model may have been re-homed (``.to()`` builds a fresh arena): stepping the orphaned
        buffers would silently train nothing."""
        live = getattr(self, "_model_ref", None)
        live = live() if live is not None else None
        if live is not None and getattr(live, "param_arena", None) is not self.arena:
            raise RuntimeError("the model's ParamArena was rebuilt after this optimizer was created "
                               "(model.to()/float() after build_optimizer): create the optimizer after "
                               "the model is on its final device")

    def flat_state_tensors(self):
        """Flat optimizer-state buffers (same layout as the arena), for sharded-state consolidation."""
        return list(self._flat_state.values())

    def begin_external_pass(self):
        """Called by the sharded reducer before it starts updating slices in a backward pass:
        returns the per-pass constants the slice updates need (e.g. the step number)."""
        raise NotImplementedError

    def step_closed_by_reducer(self):
        self._external_pass_done = True

    def _bind_state_views(self):
        a = self.arena
        for i, p in enumerate(a.params):
            st = self.state[p]
            for k, flat in self._flat_state.items():
                st[k] = a._logical_view(flat, i)

    def zero_grad(self, set_to_none=True):
        # grads stay views of the arena; kernels overwrite on first write
        self.arena.zero_grad()

    def state_dict(self):
        sd = super().state_dict()
        for st in sd["state"].values():
            for k, v in list(st.items()):
                if torch.is_tensor(v) and v.dim() > 0:
                    st[k] = v.detach().clone(memory_format=torch.contiguous_format)
        sd["fused_steps"] = self._steps
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        steps = state_dict.pop("fused_steps", None)
        super().load_state_dict(state_dict)
        a = self.arena
        loaded_any = False
        with torch.no_grad():
            for i, p in enumerate(a.params):
                st = self.state.get(p, {})
                for k, flat in self._flat_state.items():
                    view = a._logical_view(flat, i)
                    if k in st and torch.is_tensor(st[k]):
                        if st[k].data_ptr() != view.data_ptr():
                            view.copy_(st[k])
                        loaded_any = True
                    st[k] = view
                self.state[p] = st
        if steps is not None:
            self._steps = int(steps)
        elif loaded_any:
            self._steps = max(self._steps, 1)   # a torch.optim checkpoint with live buffers


class FusedSGD(_FusedBase):
    _state_keys = ("momentum_buffer",)
    graph_safe = True      # (after the first, state-initialising step; lr changes force a re-capture)

    def __init__(self, model, lr=0.1, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        # maximize/foreach/differentiable/fused: inert keys of torch.optim.SGD's group schema, kept so
        # that a checkpoint written here resumes a stock torch.optim.SGD and vice versa
        super().__init__(model, dict(lr=lr, momentum=momentum, dampening=dampening,
                                     weight_decay=weight_decay, nesterov=nesterov, maximize=False,
                                     foreach=None, differentiable=False, fused=None))

    def begin_external_pass(self):
        return dict(first=self._steps == 0)

    @torch.no_grad()
    def update_range(self, lo, hi, grad_scale=1.0, first=None):
        """SGD-momentum update of arena elements [lo, hi) only (whole arena: ``step``)."""
        a, g = self.arena, self.param_groups[0]
        first = (self._steps == 0) if first is None else first
        mom = self._flat_state["momentum_buffer"]
        if a.flat_master.is_cuda:
            K.sgd_flat(a.flat_master[lo:hi], a.flat_grad[lo:hi], mom[lo:hi], a.flat_lp[lo:hi], g["lr"], g["momentum"],
                       g["dampening"], g["weight_decay"], grad_scale, g["nesterov"], first)
            return
        m, p = mom[lo:hi], a.flat_master[lo:hi]
        d = a.flat_grad[lo:hi].float() * grad_scale
        if g["weight_decay"] != 0:
            d = d.add(p, alpha=g["weight_decay"])
        if g["momentum"] != 0:
            if first:
                m.copy_(d)
            else:
                m.mul_(g["momentum"]).add_(d, alpha=1 - g["dampening"])
            d = d.add(m, alpha=g["momentum"]) if g["nesterov"] else m
        p.add_(d, alpha=-g["lr"])

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        a = self.arena
        self._check_arena()
        if self._sharded_by is not None:
            if not self._external_pass_done:
                raise RuntimeError("sharded optimizer: step() needs a synchronised backward pass first "
                                   "(the reducer updates the slices; do not call step() inside no_sync())")
            self._external_pass_done = False
            if a.flat_master.is_cuda:
                a.lp_written_by_optimizer()
            self._steps += 1
            return loss
        if a.flat_master.is_cuda:
            a.collect_autograd_grads()
        else:
            a.gather_autograd_grads()
        self.update_range(0, a.flat_master.numel(), grad_scale)
        if a.flat_master.is_cuda:
            a.lp_written_by_optimizer()
        self._steps += 1
        return loss


class FusedAdamW(_FusedBase):
    _state_keys = ("exp_avg", "exp_avg_sq")

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(model, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def begin_external_pass(self):
        return dict(step=self._steps + 1)

    @torch.no_grad()
    def update_range(self, lo, hi, grad_scale=1.0, step=None):
        a, g = self.arena, self.param_groups[0]
        b1, b2 = g["betas"]
        step = (self._steps + 1) if step is None else step
        m, v = self._flat_state["exp_avg"][lo:hi], self._flat_state["exp_avg_sq"][lo:hi]
        if a.flat_master.is_cuda:
            K.adamw_flat(a.flat_master[lo:hi], a.flat_grad[lo:hi], m, v, a.flat_lp[lo:hi], g["lr"], b1, b2, g["eps"],
                         g["weight_decay"], grad_scale, step)
            return
        p = a.flat_master[lo:hi]
        d = a.flat_grad[lo:hi].float() * grad_scale
        p.mul_(1 - g["lr"] * g["weight_decay"])
        m.mul_(b1).add_(d, alpha=1 - b1)
        v.mul_(b2).addcmul_(d, d, value=1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(g["eps"])
        p.addcdiv_(m, denom, value=-g["lr"] / bc1)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        a = self.arena
        self._check_arena()
        if self._sharded_by is not None:
            if not self._external_pass_done:
                raise RuntimeError("sharded optimizer: step() needs a synchronised backward pass first")
            self._external_pass_done = False
            if a.flat_master.is_cuda:
                a.lp_written_by_optimizer()
            self._steps += 1
            return loss
        if a.flat_master.is_cuda:
            a.collect_autograd_grads()
        else:
            a.gather_autograd_grads()
        self.update_range(0, a.flat_master.numel(), grad_scale)
        if a.flat_master.is_cuda:
            a.lp_written_by_optimizer()
        self._steps += 1
        return loss
