"""Flat parameter / gradient / low-precision arenas.

All trainable parameters of a module are re-homed as *views* of one fp32 master buffer; their
gradients are views of one gradient buffer (fp32, or bf16 to halve all-reduce bytes); on GPU a
bf16 working copy of the whole arena is what the MFMA kernels read.  Consequences:

  * the optimizer is ONE kernel launch over the arena (``ops.optim.FusedSGD/FusedAdamW``) that
    also re-emits the bf16 copy;
  * the gradient reducer's buckets are contiguous ranges of the gradient arena (no gather copies);
  * ``zero_grad`` is a flag flip: the weight-gradient kernels overwrite (beta = 0) on first write
    and accumulate (beta = 1) afterwards;
  * ``state_dict()`` still yields per-tensor fp32 entries with the reference's key names and
    logical NCHW shapes (conv weights are stored tap-major/channel-minor, i.e. torch's
    channels_last, which is exactly the ``[Cout][3][3][Cin]`` layout the implicit-GEMM kernels
    want; checkpoints are written contiguous by models.vgg16's state-dict hook).
"""
import torch

ALIGN = 64  # elements: 256 B of fp32, 128 B of bf16 — keeps every tensor 16-byte aligned


def _round_up(n, a=ALIGN):
    return (n + a - 1) // a * a


class ParamArena(object):
    def __init__(self, module, grad_dtype=torch.float32, lp_dtype=torch.bfloat16):
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("module has no trainable parameters")
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.device = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != self.device:
                raise ValueError("ParamArena needs fp32 parameters on one device")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += _round_up(p.numel())
        self.total = off
        # zero tail padding: lets the sharded reducer round its last bucket up to a multiple of
        # world * 64 elements (any world size <= 64) without leaving the arena
        self.capacity = (off + 4095) // 4096 * 4096
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.grad_dtype = grad_dtype
        self.flat_master = torch.zeros(self.capacity, dtype=torch.float32, device=self.device)
        self.flat_grad = torch.zeros(self.capacity, dtype=grad_dtype, device=self.device)
        self.flat_lp = (torch.zeros(self.capacity, dtype=lp_dtype, device=self.device)
                        if self.device.type == "cuda" else None)
        self.fresh = [True] * len(self.params)
        self.kernel_written = set()   # id(p) of params whose grads are written by fused kernels
        self.reducer = None
        self.lp_epoch = 0            # bumped whenever flat_lp content changes
        self._lp_versions = None     # parameter versions flat_lp was derived from
        self.is_flat = True
        with torch.no_grad():
            for i, p in enumerate(self.params):
                view = self._logical_view(self.flat_master, i)
                view.copy_(p.data)
                p.data = view
                if grad_dtype == p.dtype:
                    p.grad = self._logical_view(self.flat_grad, i)
                else:
                    p.grad = None

    # ------------------------------------------------------------------ views
    def _mem_shape(self, i):
        """Shape of parameter i in memory order (conv: [Cout, kh, kw, Cin])."""
        s = tuple(self.params[i].shape)
        if len(s) == 4:
            return (s[0], s[2], s[3], s[1])
        return s

    def _logical_view(self, flat, i):
        p = self.params[i]
        v = flat[self.offsets[i]:self.offsets[i] + p.numel()].view(self._mem_shape(i))
        if p.dim() == 4:
            v = v.permute(0, 3, 1, 2)
        return v

    def flat_slice(self, flat, p):
        i = self.index[id(p)]
        return flat[self.offsets[i]:self.offsets[i] + p.numel()]

    def mem2d(self, flat, p):
        """Memory-order 2-D view [shape[0], rest] of parameter p inside ``flat``."""
        return self.flat_slice(flat, p).view(p.shape[0], -1) if p.dim() > 1 else self.flat_slice(flat, p)

    def lp(self, p):
        return self.mem2d(self.flat_lp, p)

    def grad(self, p):
        return self.mem2d(self.flat_grad, p)

    # ------------------------------------------------------------------ low-precision copy
    def _versions(self):
        return tuple(p._version for p in self.params)

    def refresh_lp(self, force=False):
        """Re-derive the bf16 working copy if any master tensor was modified behind our back
        (load_state_dict, a stock torch optimizer, a broadcast)."""
        if self.flat_lp is None:
            return
        v = self._versions()
        if force or v != self._lp_versions:
            from ..ops import kernels
            kernels.cast_f32_bf16(self.flat_master, self.flat_lp)
            self._lp_versions = v
            self.lp_epoch += 1

    def mark_master_updated(self):
        self._lp_versions = None

    def lp_written_by_optimizer(self):
        """The fused optimizer updated master and lp together (no version bump happens)."""
        self._lp_versions = self._versions()
        self.lp_epoch += 1

    # ------------------------------------------------------------------ gradients
    def zero_grad(self):
        """Flag flip for kernel-written gradients; a real memset only for gradients that
        torch.autograd accumulates into (CPU path, BatchNorm affine parameters)."""
        self.fresh = [True] * len(self.params)
        if not self.kernel_written:
            self.flat_grad.zero_()
            return
        for i, p in enumerate(self.params):
            if id(p) not in self.kernel_written:
                self.flat_grad[self.offsets[i]:self.offsets[i] + p.numel()].zero_()

    def sync_autograd_grad(self, p):
        """Bring an autograd-produced ``p.grad`` (fresh tensor or other dtype: BatchNorm affine
        parameters, generic fallbacks) into the arena: overwrite on the first write after
        zero_grad, accumulate afterwards (gradient accumulation)."""
        if id(p) in self.kernel_written or p.grad is None:
            return
        i = self.index[id(p)]
        view = self._logical_view(self.flat_grad, i)
        if p.grad.data_ptr() == view.data_ptr():
            self.fresh[i] = False
            return
        with torch.no_grad():
            if self.fresh[i]:
                view.copy_(p.grad)
            else:
                view.add_(p.grad.to(view.dtype))
            self.fresh[i] = False
            p.grad = view if self.grad_dtype == p.dtype else None

    def collect_autograd_grads(self):
        """GPU optimizer path: parameters whose gradients come from torch.autograd rather than
        from the fused kernels (none in plain VGG16; the BatchNorm affine pairs of VGG16-BN)."""
        if len(self.kernel_written) == len(self.params):
            return
        for p in self.params:
            if id(p) not in self.kernel_written:
                self.sync_autograd_grad(p)

    def grad_beta(self, p):
        """0.0 on the first gradient write after zero_grad (overwrite), 1.0 afterwards."""
        i = self.index[id(p)]
        if self.fresh[i] or (self.grad_dtype == p.dtype and p.grad is None):
            self.fresh[i] = False
            if self.grad_dtype == p.dtype and p.grad is None:
                p.grad = self._logical_view(self.flat_grad, i)
            return 0.0
        return 1.0

    def notify_grad_ready(self, p):
        if self.reducer is not None:
            self.reducer._on_grad_ready(p)

    def gather_autograd_grads(self):
        """CPU / generic path: autograd may have re-allocated .grad (zero_grad(set_to_none));
        copy such gradients back into the arena and re-point them.  Fresh-but-untouched
        gradients are zeroed so the optimizer sees zeros, like torch would skip them."""
        if self.grad_dtype != torch.float32:
            return
        with torch.no_grad():
            for i, p in enumerate(self.params):
                view = self._logical_view(self.flat_grad, i)
                if id(p) in self.kernel_written:
                    if p.grad is None:
                        p.grad = view      # the kernels wrote straight into the arena
                    continue
                if p.grad is None:
                    view.zero_()
                    p.grad = view
                elif p.grad.data_ptr() != view.data_ptr():
                    view.copy_(p.grad)
                    p.grad = view

    def wait_derived(self):
        """Backward functions call this before touching a derived weight layout: if FusedModel
        queued their construction on a side stream during forward, make the current stream wait
        for it (no-op otherwise, and after the first call of a pass)."""
        ev = getattr(self, "derived_event", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self.derived_event = None

    def memory_bytes(self):
        n = self.flat_master.numel() * 4 + self.flat_grad.numel() * self.flat_grad.element_size()
        if self.flat_lp is not None:
            n += self.flat_lp.numel() * 2
        return n
