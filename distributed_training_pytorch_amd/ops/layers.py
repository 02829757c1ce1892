"""nn.Modules + autograd Functions over the HIP kernels (GPU) with plain-PyTorch reference
implementations on CPU (selected by ``device.type == 'cpu'`` only — on a GPU the HIP path is the
only path and a missing native library is an error).

Internal GPU activation layout is NHWC bf16 ``[N, H, W, C]``; CPU reference layout is NCHW fp32.

Fusion contracts:
  * ``Conv3x3`` computes conv + bias + ReLU in one kernel;
  * ReLU-backward is never a kernel of its own: the consumer of an activation applies
    ``grad * (act > 0)`` in the epilogue of its data-gradient kernel (``input_is_relu_out``) or
    in the max-pool backward (``relu_mask``); a layer whose incoming gradient is already masked is
    built with ``grad_premasked=True``;
  * dropout is an epilogue of the classifier GEMM (counter-based RNG, no mask tensor: the
    backward mask is ``output != 0``);
  * weight/bias gradients are written by the kernels straight into the gradient arena
    (overwrite-on-first-write), the reducer is notified per parameter.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as K
from .rng import DropoutRng

BF16 = torch.bfloat16
# dropout layers used outside a FusedModel (tests, ad-hoc modules) share this state; models own theirs
_default_rng = DropoutRng()

# run each conv layer's weight gradient on a side stream, concurrently with its data gradient
OVERLAP_WGRAD = os.environ.get("DTP_OVERLAP_WGRAD", "1") == "1"
# unmasked linear data gradient without the per-step W^T copy (A/B switch)
LINEAR_DGRAD_TT = os.environ.get("DTP_LINEAR_DGRAD_TT", "1") == "1"
FP8_DGRAD = os.environ.get("DTP_FP8_DGRAD", "1") == "1"     # A/B switch: fp8 layers run their data gradient in e5m2 x e4m3
# Inside a hipGraph capture the weight-gradient kernel of a layer is RECORDED after the layer's data gradient
# (its dependencies are unchanged: the side stream waited for the fork point before either was queued).  The
# order of a node's out-edges decides which child the ROCm graph executor keeps on the parent's stream: with
# the data gradient recorded first the backward chain stays on one hardware queue and the side work hops.
GRAPH_MAIN_FIRST = os.environ.get("DTP_GRAPH_MAIN_FIRST", "1") == "1"
# opt-in: the bias gradient of a convolution issued behind its weight gradient on the side stream instead of in
# front of the data gradient.  Measured -0.3 to -0.6 % at batch 1024 (three same-box pairs): the two tiny kernels
# cost less in the gap in front of the data gradient than as a tail on the weight-gradient chain
BIAS_GRAD_ON_SIDE = os.environ.get("DTP_BIAS_GRAD_ON_SIDE", "0") == "1"
_side_streams = {}


def _record_late():
    return GRAPH_MAIN_FIRST and torch.cuda.is_current_stream_capturing()
# Deferred join (join the side stream only at the end of backward) is OFF by default: it needs
# record_stream() on every activation the weight-gradient kernels read, which parks those blocks in
# the caching allocator until the side stream's events retire — with the host running steps ahead
# the reserved memory grew from 14 GB to 140-235 GB at batch 256 and one run in ~12 hit the limit
# and ran 5x slower (allocator free/sync storm).  The per-layer join needs no record_stream.
DEFER_JOIN = os.environ.get("DTP_DEFER_JOIN", "0") in ("1", "2")
# "2": the same deferred join WITHOUT record_stream — the tensors the weight-gradient kernels read are simply kept
# alive (a Python list) until the join at the end of backward, and released there, after the main stream has waited
# for the side stream: no allocator events, at the price of every layer's dY staying allocated until backward ends
DEFER_KEEPALIVE = os.environ.get("DTP_DEFER_JOIN", "0") == "2"
_keepalive = []
# ... but inside a hipGraph capture it is free: the captured step owns a private memory pool, nothing runs
# ahead, and the data-gradient chain (the critical path of a launch-bound small-batch step) no longer waits
# for each layer's weight gradient.  The per-bucket optimizer waits for both streams (parallel/reducer.py).
GRAPH_DEFER_JOIN = os.environ.get("DTP_GRAPH_DEFER_JOIN", "0") == "1"


_pending_join = {}


def _fork_wgrad(arena, device, *tensors):
    """Side stream for a weight-gradient kernel, ordered after everything queued so far on the
    current stream.  Without a gradient reducer the join is deferred to the END of backward (an
    autograd-engine callback): the chain of weight gradients then runs as its own pipeline next to
    the data-gradient / pool-backward chain.  The tensors the kernel reads are handed to the
    caching allocator with ``record_stream`` so that they are not recycled under it."""
    side = _side_stream(device)
    side.wait_stream(torch.cuda.current_stream())
    deferred = ((DEFER_JOIN or (GRAPH_DEFER_JOIN and torch.cuda.is_current_stream_capturing()))
                and (arena.reducer is None or getattr(arena.reducer, "world", 1) == 1))
    if deferred:
        if DEFER_KEEPALIVE and not torch.cuda.is_current_stream_capturing():
            _keepalive.extend(tensors)
        else:
            for t in tensors:
                t.record_stream(side)
        key = id(side)
        if key not in _pending_join:
            _pending_join[key] = side

            def _join(side=side, key=key):
                _pending_join.pop(key, None)
                torch.cuda.current_stream().wait_stream(side)
                del _keepalive[:]
            torch.autograd.Variable._execution_engine.queue_callback(_join)
    return side, deferred


def _side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=device)
    return s


def _bias_grad(layer, arena, dy2d, bias, ws_key="colsum"):
    """Bias gradient of a convolution = column sums of its dY.  When a training-mode BatchNorm
    follows the convolution, dY comes out of the BatchNorm backward and its column sums are zero by
    construction (sum_rows dY = g * (S1 - M*c1 - c2 * sum_rows xhat) with c1 = S1/M and
    sum_rows xhat = 0): the bias cannot influence a batch-normalised output, so the 1x read of dY by
    the column-sum kernel is skipped and an exact zero is written (or nothing, when accumulating)."""
    beta = arena.grad_beta(bias)
    if getattr(layer, "bn_follows", False) and layer.training:
        if beta == 0.0:
            arena.grad(bias).zero_()
        return
    K.bias_grad(dy2d, arena.grad(bias), beta=beta, ws_key=ws_key)


def _arena_of(layer):
    a = getattr(layer, "_arena", None)
    if a is None:
        raise RuntimeError(
            "%s runs on the GPU only inside a model that owns a ParamArena "
            "(subclass models.FusedModel, or call models.attach_arena(model) after moving it to the device)" % type(layer).__name__)
    return a


class _PoolLink(object):
    """Hand-over between a 2x2 max-pool and the convolution that consumes its output.

    Forward: the pool hangs one of these on its output tensor (``_dtp_pool_link``) and the next
    Conv3x3 keeps it.  Backward: that convolution's data-gradient kernel routes every value to its
    arg-max position itself (csrc/igemm.hip, un-pooling epilogue) and parks the FULL-RESOLUTION
    gradient here; what it returns to autograd is a stride-0 placeholder of the pooled shape.  The
    pool's backward takes the parked tensor instead of running the pool-backward kernel — after
    checking that the gradient it was handed is that very placeholder (a second consumer of the
    pooled activation would make autograd sum into a fresh tensor, which is refused loudly)."""

    __slots__ = ("codes", "in_shape", "relu_mask", "full", "token")

    def __init__(self, codes, in_shape, relu_mask):
        self.codes, self.in_shape, self.relu_mask = codes, tuple(in_shape), bool(relu_mask)
        self.full = None
        self.token = None

    def park(self, full, pooled_shape):
        self.full = full
        self.token = full.as_strided(tuple(pooled_shape), (0,) * len(pooled_shape))
        return self.token

    def take(self, dyp):
        """The parked full-resolution gradient if ``dyp`` is this link's placeholder, else None."""
        full, token = self.full, self.token
        self.full = self.token = None
        if full is None:
            return None
        if dyp.data_ptr() != token.data_ptr() or dyp.stride() != token.stride() or dyp.shape != token.shape:
            raise RuntimeError("un-pooled gradient hand-over: the pooled activation has more than one consumer "
                               "(or a hook rewrote its gradient); run with DTP_FUSE_UNPOOL=0")
        return full


def _dgrad_through_pool(link, x, dy, wd):
    """Data gradient of a convolution whose input ``x`` is ``link``'s pool output, un-pooled in the
    same kernel; None when there is no fused path for this shape."""
    if link is None or not K.fuse_unpool:
        return None
    N, H, W, C = x.shape
    if link.in_shape != (N, 2 * H, 2 * W, C):
        return None
    full = K.conv3x3_dgrad(dy, wd, unpool=(link.codes, link.relu_mask))
    if full is None:
        return None
    return link.park(full, x.shape)


class FusedLayer(nn.Module):
    """Base: access to the arena's bf16 weight copy and lazily derived dgrad layouts."""

    def __init__(self):
        super().__init__()
        self._arena = None
        self._derived = {}
        self._derived_epoch = -1

    def bind_arena(self, arena):
        object.__setattr__(self, "_arena", arena)
        for p in self.parameters(recurse=False):
            arena.kernel_written.add(id(p))

    def derived_builders(self, needs_input_grad=True):
        """(key, builder) pairs of the weight layouts this layer derives once per optimizer step
        for its backward pass; FusedModel builds them on a side stream during forward."""
        return []

    def derived_transposes(self, needs_input_grad=True):
        """(key, persistent output tensor, K.TransposePlan row) for the derived layouts that are plain
        transposes of the bf16 weights: FusedModel runs ALL of a model's in one launch per step."""
        return []

    def _set_derived(self, key, value):
        arena = _arena_of(self)
        if self._derived_epoch != arena.lp_epoch:
            self._derived = {}
            self._derived_epoch = arena.lp_epoch
        self._derived[key] = value

    def _cached(self, key, builder):
        arena = _arena_of(self)
        if self._derived_epoch != arena.lp_epoch:
            self._derived = {}
            self._derived_epoch = arena.lp_epoch
        v = self._derived.get(key)
        if v is None:
            v = builder()
            self._derived[key] = v
        return v


# =============================================================================== conv 3x3
class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, layer):
        arena = _arena_of(layer)
        # 1-bit ReLU masks: this layer packs (y > 0) for the data gradient of the NEXT convolution and
        # picks up the bits its producer left on x (tensor attribute set in Conv3x3.forward)
        want_bits = K.relu_bitmask and layer.relu and layer.emit_relu_bits and not layer.__dict__.get("_infer", False)
        # a training-mode BatchNorm follows: the epilogue leaves its batch statistics behind
        want_stats = bool(getattr(layer, "bn_follows", False)) and layer.training and not layer.relu
        y = K.conv3x3_fprop(x, arena.lp(weight), bias.detach() if bias is not None else None, relu=layer.relu,
                            want_bits=want_bits, want_bn_stats=want_stats)
        if want_bits:
            y, layer._bits_tmp = y
        ctx.layer = layer
        ctx.has_bias = bias is not None
        ctx.mask_bits = getattr(x, "_dtp_relu_bits", None) if (K.relu_bitmask and layer.input_is_relu_out) else None
        # x is the output of a training-mode BatchNorm(+ReLU): (its input y, its statistics, relu) — this
        # layer's data gradient then leaves that BatchNorm's backward reductions behind
        ctx.bn_src = getattr(x, "_dtp_bn_src", None)
        # x is the output of a 2x2 max-pool: the data gradient un-pools in its epilogue (see _PoolLink)
        ctx.pool_link = getattr(x, "_dtp_pool_link", None) if not layer.input_is_relu_out else None
        ctx.save_for_backward(x, y, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer = ctx.layer
        arena = _arena_of(layer)
        x, y, weight, bias = ctx.saved_tensors
        dy = dy.contiguous()
        if layer.relu and not layer.grad_premasked:
            dy = K.relu_mask(dy, y)
        Co = weight.shape[0]
        side, deferred = (_fork_wgrad(arena, dy.device, dy, x) if (OVERLAP_WGRAD and ctx.needs_input_grad[0])
                          else (None, False))
        late = side is not None and _record_late()
        bias_done = False
        if side is not None:
            # weight gradient on a second stream, concurrent with the data gradient below: both
            # read dY, neither reads the other's output, and the tail of one kernel's last round of
            # tiles (conv5: 196 tiles on 256 CUs) is filled by the other's workgroups
            beta_w = arena.grad_beta(weight)
            if not late:
                with torch.cuda.stream(side):
                    K.conv3x3_wgrad(dy, x, arena.grad(weight), beta=beta_w)
                    if ctx.has_bias and BIAS_GRAD_ON_SIDE:
                        # the bias gradient (a reduction of the partial rows dY's producer left) rides behind the
                        # weight gradient: two tiny kernels less per layer on the data-gradient chain
                        _bias_grad(layer, arena, dy.view(-1, Co), bias, ws_key="colsum_side")
                        bias_done = True
                if K.wgrad_is_persistent(x.shape[3], Co) and dy.numel() >= K.WGRAD_HEAD_START_MIN_ELEMS:
                    K.stream_delay(K.wgrad_head_start_us)     # the persistent grid must be resident first
        else:
            K.conv3x3_wgrad(dy, x, arena.grad(weight), beta=arena.grad_beta(weight))
        if ctx.has_bias and not bias_done:
            _bias_grad(layer, arena, dy.view(-1, Co), bias)
        dx = None
        if ctx.needs_input_grad[0]:
            Ci = weight.shape[1]
            arena.wait_derived()
            wd = layer._cached("wd", lambda: K.conv_weight_to_dgrad(arena.lp(weight), Co, Ci))
            dx = _dgrad_through_pool(ctx.pool_link, x, dy, wd) if ctx.bn_src is None else None
            if dx is None:
                dx = K.conv3x3_dgrad(dy, wd, mask_src=x if layer.input_is_relu_out else None, mask_bits=ctx.mask_bits,
                                     wd_halves=layer._wd_halves(wd), bn_bwd_src=ctx.bn_src)
        if late:
            with torch.cuda.stream(side):
                K.conv3x3_wgrad(dy, x, arena.grad(weight), beta=beta_w)
        if side is not None and not deferred:
            torch.cuda.current_stream().wait_stream(side)    # join before the reducer ships the gradient
        if ctx.has_bias:
            arena.notify_grad_ready(bias)
        arena.notify_grad_ready(weight)
        return dx, None, None, None


class _Conv3x3ReluPoolFn(torch.autograd.Function):
    """conv3x3 + bias + ReLU + 2x2 max-pool as one kernel (64-reduction-channel layers): the
    pre-pool activation never reaches HBM.  Backward: pool backward from the 4-bit codes (ReLU
    mask included, bias-gradient column sums left behind), then the usual wgrad / dgrad."""

    @staticmethod
    def forward(ctx, x, weight, bias, layer):
        arena = _arena_of(layer)
        y, codes = K.conv3x3_relu_pool_fprop(x, arena.lp(weight), bias.detach())
        ctx.layer = layer
        ctx.mask_bits = getattr(x, "_dtp_relu_bits", None) if (K.relu_bitmask and layer.input_is_relu_out) else None
        ctx.pool_link = _PoolLink(codes, (x.shape[0], x.shape[1], x.shape[2], weight.shape[0]), True)
        layer.__dict__["_link_tmp"] = ctx.pool_link
        ctx.save_for_backward(x, codes, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dyp):
        layer = ctx.layer
        arena = _arena_of(layer)
        x, codes, weight, bias = ctx.saved_tensors
        N, H, W, _ = x.shape
        Co, Ci = weight.shape[0], weight.shape[1]
        dy = ctx.pool_link.take(dyp)        # the consumer's data gradient may have un-pooled already
        if dy is None:
            dy = K.maxpool2x2_bwd(None, dyp.contiguous(), relu_mask=True, idx=codes, in_shape=(N, H, W, Co))
        side, deferred = (_fork_wgrad(arena, dy.device, dy, x) if (OVERLAP_WGRAD and ctx.needs_input_grad[0])
                          else (None, False))
        late = side is not None and _record_late()
        bias_done = False
        if side is not None:
            beta_w = arena.grad_beta(weight)
            if not late:
                with torch.cuda.stream(side):
                    K.conv3x3_wgrad(dy, x, arena.grad(weight), beta=beta_w)
                    if BIAS_GRAD_ON_SIDE:
                        K.bias_grad(dy.view(-1, Co), arena.grad(bias), beta=arena.grad_beta(bias), ws_key="colsum_side")
                        bias_done = True
                if K.wgrad_is_persistent(Ci, Co) and dy.numel() >= K.WGRAD_HEAD_START_MIN_ELEMS:
                    K.stream_delay(K.wgrad_head_start_us)
        else:
            K.conv3x3_wgrad(dy, x, arena.grad(weight), beta=arena.grad_beta(weight))
        if not bias_done:
            K.bias_grad(dy.view(-1, Co), arena.grad(bias), beta=arena.grad_beta(bias))
        dx = None
        if ctx.needs_input_grad[0]:
            arena.wait_derived()
            wd = layer._cached("wd", lambda: K.conv_weight_to_dgrad(arena.lp(weight), Co, Ci))
            dx = K.conv3x3_dgrad(dy, wd, mask_src=x if layer.input_is_relu_out else None, mask_bits=ctx.mask_bits,
                                 wd_halves=layer._wd_halves(wd))
        if late:
            with torch.cuda.stream(side):
                K.conv3x3_wgrad(dy, x, arena.grad(weight), beta=beta_w)
        if side is not None and not deferred:
            torch.cuda.current_stream().wait_stream(side)
        arena.notify_grad_ready(bias)
        arena.notify_grad_ready(weight)
        return dx, None, None, None


def conv_relu_pool(conv, x):
    """Fused forward of ``conv`` (a :class:`Conv3x3` with ReLU) followed by MaxPool2x2, or None when
    this layer/shape is not eligible (caller then runs the two modules)."""
    if (not x.is_cuda or not isinstance(conv, Conv3x3) or not conv.relu or not conv.grad_premasked
            or conv.bias is None or x.dim() != 4):
        return None
    if not K.conv_pool_fusable(x.shape[3], conv.out_channels, x.shape[1], x.shape[2], N=x.shape[0]):
        return None
    y = _Conv3x3ReluPoolFn.apply(x, conv.weight, conv.bias, conv)
    link = conv.__dict__.pop("_link_tmp", None)
    if link is not None and torch.is_grad_enabled():
        y._dtp_pool_link = link          # travels with the pooled activation to the next Conv3x3
    return y


class _Conv3x3Derived(object):
    def derived_builders(self, needs_input_grad=True):
        if not needs_input_grad:
            return []
        arena = _arena_of(self)
        Co, Ci = self.out_channels, self.in_channels
        out = [("wd", lambda: K.conv_weight_to_dgrad(arena.lp(self.weight), Co, Ci))]
        if Co == 128 and Ci == 64 and K.two_pass_c128:
            out.append(("wd_halves", lambda: K.split_dgrad_weight(self._cached("wd", out[0][1]), Ci, Co)))
        return out

    def derived_transposes(self, needs_input_grad=True):
        if not needs_input_grad:
            return []
        arena = _arena_of(self)
        Co, Ci = self.out_channels, self.in_channels
        out = torch.empty((Ci, 9 * Co), dtype=BF16, device=arena.device)
        return [("wd", out, K.conv_weight_to_dgrad_job(arena.lp(self.weight), Co, Ci, out))]

    def _wd_halves(self, wd):
        """Reduction halves of the dgrad weight for the two-pass 128 -> 64 data gradient (else None)."""
        Co, Ci = self.out_channels, self.in_channels
        if not (Co == 128 and Ci == 64 and K.two_pass_c128):
            return None
        return self._cached("wd_halves", lambda: K.split_dgrad_weight(wd, Ci, Co))


class Conv3x3(_Conv3x3Derived, FusedLayer):
    """3x3, stride 1, pad 1 convolution (+bias, +ReLU fused).  ``weight`` keeps torch's logical
    ``[Cout, Cin, 3, 3]`` shape (checkpoint compatibility with reference model/vgg16.py:9,12)."""

    def __init__(self, in_channels, out_channels, relu=True, input_is_relu_out=False, grad_premasked=False,
                 emit_relu_bits=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.relu = relu
        self.input_is_relu_out = input_is_relu_out
        self.grad_premasked = grad_premasked
        self.emit_relu_bits = emit_relu_bits       # False: the consumer is not a masking Conv3x3 (e.g. a pool)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        if not x.is_cuda:
            y = F.conv2d(x, self.weight, self.bias, padding=1)
            return F.relu(y) if self.relu else y
        self.__dict__["_infer"] = not torch.is_grad_enabled()     # no backward: no ReLU masks to leave behind
        y = _Conv3x3Fn.apply(x, self.weight, self.bias, self)
        bits = self.__dict__.pop("_bits_tmp", None)
        if bits is not None:
            y._dtp_relu_bits = bits          # travels with the activation to the next Conv3x3
        return y

    def extra_repr(self):
        return "%d, %d, relu=%s" % (self.in_channels, self.out_channels, self.relu)


# =============================================================================== first conv (Cin = 3)
class _ConvFirstFn(torch.autograd.Function):
    """Image ingest + first convolution (csrc/conv_first.hip): the NCHW f32/bf16/u8 batch is
    normalised and im2col'ed inside the kernel; the weight AND bias gradient come out of one
    persistent kernel.  Inputs with Cin != 3 take the generic kernels on a channel-padded copy."""

    @staticmethod
    def forward(ctx, x, weight, bias, layer):
        arena = _arena_of(layer)
        x = x.contiguous()
        ctx.layer = layer
        ctx.fast = x.shape[1] == 3
        if ctx.fast:
            want_stats = bool(getattr(layer, "bn_follows", False)) and layer.training and not layer.relu
            y = K.conv1_fprop(x, arena.lp(weight), bias.detach(), layer.relu, layer.in_scale, layer.in_mean,
                              layer.in_inv_std, want_bn_stats=want_stats)
            ctx.save_for_backward(x, y, weight, bias)
            return y
        xp = K.nchw_to_nhwc_bf16(x, layer.c_pad, layer.in_scale, layer.in_mean, layer.in_inv_std)
        wp = layer._cached("w_pad", lambda: layer._pad_weight(arena.lp(weight)))
        y = K.conv3x3_fprop(xp, wp, bias.detach(), relu=layer.relu)
        ctx.save_for_backward(xp, y, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer = ctx.layer
        arena = _arena_of(layer)
        xp, y, weight, bias = ctx.saved_tensors
        dy = dy.contiguous()
        in_kernel_mask = ctx.fast and layer.relu and layer.self_mask and not layer.grad_premasked
        if layer.relu and not layer.grad_premasked and not in_kernel_mask:
            dy = K.relu_mask(dy, y)
        Co, Ci = weight.shape[0], weight.shape[1]
        if ctx.fast:
            # self_mask: the weight-gradient kernel recomputes sign(conv1(x)+b) per tile and masks dy
            # itself, so the layer above skipped the 64-channel mask read in its data-gradient
            K.conv1_wgrad(xp, dy, arena.grad(weight), arena.grad(bias), arena.grad_beta(weight),
                          arena.grad_beta(bias), layer.in_scale, layer.in_mean, layer.in_inv_std,
                          mask_w=arena.lp(weight) if in_kernel_mask else None,
                          mask_bias=bias.detach() if in_kernel_mask else None)
        else:
            gpad = K.workspace.get("convfirst_gw", Co * 9 * layer.c_pad, torch.float32, dy.device)
            gpad = gpad[:Co * 9 * layer.c_pad].view(Co, 9 * layer.c_pad)
            K.conv3x3_wgrad(dy, xp, gpad, beta=0.0)
            g = gpad.view(Co, 9, layer.c_pad)[:, :, :Ci]
            dst = arena.grad(weight).view(Co, 9, Ci)
            if arena.grad_beta(weight) == 0.0:
                dst.copy_(g)
            else:
                dst.add_(g)
            _bias_grad(layer, arena, dy.view(-1, Co), bias)
        arena.notify_grad_ready(bias)
        arena.notify_grad_ready(weight)
        return None, None, None, None


class ConvFirst(FusedLayer):
    def __init__(self, in_channels, out_channels, relu=True, grad_premasked=False, c_pad=64, self_mask=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.relu = relu
        self.grad_premasked = grad_premasked
        self.self_mask = bool(self_mask)      # ReLU backward applied inside this layer's own wgrad kernel
        self.c_pad = c_pad
        self.in_scale, self.in_mean, self.in_inv_std = 1.0, None, None
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")

    def set_input_normalization(self, scale=1.0, mean=None, std=None):
        """Fuse (x*scale - mean)/std into the ingest kernel (uint8 images over PCIe, K17)."""
        dev = self.weight.device
        self.in_scale = float(scale)
        self.in_mean = None if mean is None else torch.tensor(mean, dtype=torch.float32, device=dev)
        self.in_inv_std = None if std is None else 1.0 / torch.tensor(std, dtype=torch.float32, device=dev)

    def _pad_weight(self, w_lp):
        Co, Ci = self.out_channels, self.in_channels
        wp = torch.zeros((Co, 9, self.c_pad), dtype=BF16, device=w_lp.device)
        wp[:, :, :Ci] = w_lp.view(Co, 9, Ci)
        return wp.view(Co, 9 * self.c_pad)

    def forward(self, x):
        if not x.is_cuda:
            x = x.float()
            if self.in_mean is not None or self.in_scale != 1.0:
                x = x * self.in_scale
                if self.in_mean is not None:
                    x = (x - self.in_mean.view(1, -1, 1, 1).to(x.device)) * self.in_inv_std.view(1, -1, 1, 1).to(x.device)
            y = F.conv2d(x, self.weight, self.bias, padding=1)
            return F.relu(y) if self.relu else y
        return _ConvFirstFn.apply(x, self.weight, self.bias, self)


# =============================================================================== pooling
class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, relu_mask):
        # the forward pass leaves 4-bit (argmax, positive) codes: backward reads 1/16 of the bytes
        # of the full-resolution input it would otherwise re-scan
        y, idx = K.maxpool2x2_fwd(x, want_idx=True)
        ctx.relu_mask = relu_mask
        ctx.in_shape = tuple(x.shape)
        ctx.pool_link = _PoolLink(idx, ctx.in_shape, relu_mask) if (x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0) else None
        _MaxPoolFn.last_link = ctx.pool_link
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        full = ctx.pool_link.take(dy) if ctx.pool_link is not None else None
        if full is not None:
            return full, None
        return K.maxpool2x2_bwd(None, dy.contiguous(), relu_mask=ctx.relu_mask, idx=idx,
                                in_shape=ctx.in_shape), None


class MaxPool2x2(nn.Module):
    """2x2/s2 max-pool; ``relu_mask=True`` folds the preceding ReLU's backward into the pool
    backward (then the producer conv is built with ``grad_premasked=True``)."""

    def __init__(self, relu_mask=False):
        super().__init__()
        self.relu_mask = relu_mask

    def forward(self, x):
        if not x.is_cuda:
            return F.max_pool2d(x, 2, 2)
        y = _MaxPoolFn.apply(x, self.relu_mask)
        link, _MaxPoolFn.last_link = getattr(_MaxPoolFn, "last_link", None), None
        if link is not None and torch.is_grad_enabled() and y.requires_grad:
            y._dtp_pool_link = link
        return y


class _AvgPoolFlattenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        ctx.shape = tuple(x.shape)
        ctx.o = (oh, ow)
        return K.avgpool_flatten_fwd(x, oh, ow)

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C = ctx.shape
        return K.avgpool_flatten_bwd(dy.contiguous(), N, H, W, C, *ctx.o), None, None


class AvgPoolFlatten(nn.Module):
    """AdaptiveAvgPool2d((7,7)) + flatten in the reference's C,H,W order (model/vgg16.py:34,67-70):
    on GPU one kernel turns NHWC features into the [N, C*7*7] classifier input."""

    def __init__(self, output_size=(7, 7)):
        super().__init__()
        self.output_size = tuple(output_size)

    def forward(self, x):
        if not x.is_cuda:
            return torch.flatten(F.adaptive_avg_pool2d(x, self.output_size), 1)
        return _AvgPoolFlattenFn.apply(x, *self.output_size)


# =============================================================================== linear
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, layer, drop_p, seed, seed_ptr=None):
        arena = _arena_of(layer)
        if layer.fp8 and x.shape[1] % 128 == 0:
            # fp8 (OCP e4m3) forward GEMM, per-tensor scales computed on the device; the weight copy
            # is re-quantised once per optimizer step (cache keyed on the arena's lp epoch).
            # Backward: the data gradient runs on e5m2 gradients x e4m3 transposed weights (below),
            # the weight gradient stays on the bf16 operands (reduction over the batch axis: both
            # operands would need a transposing 8-bit read)
            wq, w_scale = layer._cached("w_fp8", lambda: K.quant_fp8(arena.lp(weight)))
            xq, x_scale = K.quant_fp8(x)
            y = K.linear_fwd_fp8(xq, x_scale, wq, w_scale, bias.detach(), relu=layer.relu, drop_p=drop_p,
                                 seed=seed, out_fp32=layer.out_fp32, seed_ptr=seed_ptr)
        else:
            y = K.linear_fwd(x, arena.lp(weight), bias.detach(), relu=layer.relu, drop_p=drop_p, seed=seed,
                             out_fp32=layer.out_fp32, seed_ptr=seed_ptr)
        ctx.layer = layer
        ctx.in_scale = layer._input_mask_scale()
        ctx.save_for_backward(x, y, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer = ctx.layer
        arena = _arena_of(layer)
        x, y, weight, bias = ctx.saved_tensors
        Nout, Kin = weight.shape
        ld = (Nout + 63) // 64 * 64
        if dy.dtype != BF16 or dy.shape[1] != ld or not dy.is_contiguous():
            d16 = torch.zeros((dy.shape[0], ld), dtype=BF16, device=dy.device) if ld != Nout else \
                torch.empty((dy.shape[0], ld), dtype=BF16, device=dy.device)
            d16[:, :Nout] = dy
            dy = d16
        if (layer.relu or layer.drop_p > 0) and not layer.grad_premasked:
            scale = 1.0 / (1.0 - layer.drop_p) if (layer.drop_p > 0 and layer.training) else 1.0
            dy = K.relu_mask(dy, y, scale)
        side, deferred = (_fork_wgrad(arena, dy.device, dy, x) if (OVERLAP_WGRAD and ctx.needs_input_grad[0])
                          else (None, False))
        late = side is not None and _record_late()
        if side is not None:
            beta_w = arena.grad_beta(weight)
            if not late:
                with torch.cuda.stream(side):
                    K.linear_wgrad(dy, x, arena.grad(weight), n_valid=Nout, beta=beta_w)
        else:
            K.linear_wgrad(dy, x, arena.grad(weight), n_valid=Nout, beta=arena.grad_beta(weight))
        K.bias_grad(dy, arena.grad(bias), n_valid=Nout, beta=arena.grad_beta(bias))
        dx = None
        if ctx.needs_input_grad[0] and layer.fp8 and FP8_DGRAD and ld % 128 == 0:
            # fp8 data gradient: dY quantised to e5m2 (per-tensor scale, on the device), W^T to e4m3
            # once per optimizer step; the ReLU/dropout mask of the layer below rides in the epilogue
            arena.wait_derived()
            wt = layer._cached("wt", lambda: K.linear_weight_transpose(arena.lp(weight)))
            wtq, wt_scale = layer._cached("wt_fp8", lambda: K.quant_fp8(wt))
            dyq, dy_scale = K.quant_fp8(dy, e5m2=True)
            dx = K.linear_dgrad_fp8(dyq, dy_scale, wtq, wt_scale, mask_src=x if layer.input_is_relu_out else None,
                                    mask_scale=ctx.in_scale)
        elif ctx.needs_input_grad[0] and layer._dgrad_without_wt(dy.shape[0]):
            # dX = dY @ W as a "TT" GEMM (reduction axis slow in BOTH operands, csrc/wgrad.hip) over the
            # weight in its native [N_out, K_in] layout: only the small dY is transposed, the 205 MB
            # W^T copy of linear1 is neither built (0.17 ms of HBM traffic per step) nor kept
            dyT = K.linear_weight_transpose(dy, pad_to=8)
            dx = torch.empty((dy.shape[0], Kin), dtype=BF16, device=dy.device)
            K.linear_wgrad(dyT, arena.lp(weight), dx, n_valid=dy.shape[0])
        elif ctx.needs_input_grad[0]:
            arena.wait_derived()
            wt = layer._cached("wt", lambda: K.linear_weight_transpose(arena.lp(weight)))
            dx = K.linear_dgrad(dy, wt, mask_src=x if layer.input_is_relu_out else None, mask_scale=ctx.in_scale)
        if late:
            with torch.cuda.stream(side):
                K.linear_wgrad(dy, x, arena.grad(weight), n_valid=Nout, beta=beta_w)
        if side is not None and not deferred:
            torch.cuda.current_stream().wait_stream(side)
        arena.notify_grad_ready(bias)
        arena.notify_grad_ready(weight)
        return dx, None, None, None, None, None, None


class _LinearDerived(object):
    def derived_builders(self, needs_input_grad=True):
        if not needs_input_grad or (self._dgrad_without_wt(None) and not (self.fp8 and FP8_DGRAD)):
            return []
        arena = _arena_of(self)
        return [("wt", lambda: K.linear_weight_transpose(arena.lp(self.weight)))]

    def derived_transposes(self, needs_input_grad=True):
        if not self.derived_builders(needs_input_grad):
            return []
        arena = _arena_of(self)
        ld = (self.out_features + 63) // 64 * 64
        out = (torch.zeros if ld != self.out_features else torch.empty)((self.in_features, ld), dtype=BF16,
                                                                        device=arena.device)
        return [("wt", out, K.linear_weight_transpose_job(arena.lp(self.weight), out))]

    def _dgrad_without_wt(self, batch):
        """Unmasked data gradients of wide layers (linear1) run on the transposing-read GEMM over
        the native weight layout; ``batch=None`` asks whether the layer ever needs its W^T."""
        ok = (LINEAR_DGRAD_TT and not self.input_is_relu_out and self.out_features % 64 == 0
              and self.in_features % 8 == 0 and self.in_features * self.out_features >= (1 << 24))
        return ok       # any batch: dY^T is zero-padded to 8 columns and only ``batch`` output rows are stored


class Linear(_LinearDerived, FusedLayer):
    """Linear (+bias) with optional fused ReLU and dropout epilogue (order: linear, relu, dropout —
    reference model/vgg16.py:71-76).  ``out_fp32`` for the logits layer."""

    def __init__(self, in_features, out_features, relu=False, drop_p=0.0, out_fp32=False,
                 input_is_relu_out=False, input_drop_p=0.0, grad_premasked=False, fp8=False):
        super().__init__()
        self.fp8 = bool(fp8)        # forward GEMM on fp8 MFMA (GPU only; CPU path ignores it)
        self.in_features, self.out_features = in_features, out_features
        self.relu, self.drop_p, self.out_fp32 = relu, float(drop_p), out_fp32
        self.input_is_relu_out = input_is_relu_out
        self.input_drop_p = float(input_drop_p)
        self.grad_premasked = grad_premasked
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features))
        nn.init.normal_(self.weight, 0, 0.01)
        # hipGraph replay: kernel arguments are frozen at capture, so the dropout seed must come
        # from device memory (engine.graph.GraphedTrainStep advances it inside the graph)
        self.device_seed = None
        # mask key = f(seed, rank, forward-call number of the model, this salt): ops/rng.py.  The salt is
        # the layer's position inside its model (models.fused_model.attach_arena re-assigns it), so it
        # is the same in every process and after every re-creation of the model
        self._seed_salt = DropoutRng.salt_for(0)
        self._rng = None

    def _input_mask_scale(self):
        if self.input_is_relu_out and self.input_drop_p > 0 and self.training:
            return 1.0 / (1.0 - self.input_drop_p)
        return 1.0

    def forward(self, x):
        if not x.is_cuda:
            y = F.linear(x, self.weight, self.bias)
            if self.relu:
                y = F.relu(y)
            if self.drop_p > 0:
                y = F.dropout(y, self.drop_p, self.training)
            return y
        p = self.drop_p if self.training else 0.0
        if p > 0 and self.device_seed is not None:
            return _LinearFn.apply(x, self.weight, self.bias, self, p, self._seed_salt, self.device_seed)
        seed = 0
        if p > 0:
            rng = self._rng
            if rng is None:                      # stand-alone layer: every forward is a new call
                rng = _default_rng
                rng.advance()
            seed = rng.key(self._seed_salt)
        return _LinearFn.apply(x, self.weight, self.bias, self, p, seed)


# =============================================================================== batch norm (+ ReLU)
class _BatchNormReLUFn(torch.autograd.Function):
    """Training-mode BatchNorm + ReLU on NHWC bf16 (csrc/bn.hip): batch statistics, running-stat
    update, normalise + affine + ReLU in the forward; the backward recomputes the ReLU mask from
    the saved convolution output, reduces d(gamma) / d(beta) straight into the gradient arena and
    emits the convolution's incoming gradient."""

    @staticmethod
    def forward(ctx, y, weight, bias, layer):
        arena = _arena_of(layer)
        y = y.contiguous()
        z, stats = K.bn_fwd_train(y, arena.flat_slice(arena.flat_master, weight), arena.flat_slice(arena.flat_master, bias),
                                  layer.running_mean, layer.running_var, layer.eps, layer.momentum, layer.relu)
        ctx.layer = layer
        ctx.save_for_backward(y, stats, weight, bias)
        layer._bn_src_tmp = (y, stats, bool(layer.relu))     # travels with z to the consuming convolution
        return z

    @staticmethod
    def backward(ctx, dz):
        layer = ctx.layer
        arena = _arena_of(layer)
        y, stats, weight, bias = ctx.saved_tensors
        dy = K.bn_bwd(y, dz.contiguous(), arena.flat_slice(arena.flat_master, weight), stats, arena.grad(weight),
                      arena.grad(bias), arena.grad_beta(weight), arena.grad_beta(bias), relu=layer.relu)
        arena.notify_grad_ready(bias)
        arena.notify_grad_ready(weight)
        return dy, None, None, None


class _BatchNormReLUPoolFn(torch.autograd.Function):
    """Block end of VGG16-BN: BatchNorm + ReLU + 2x2 max-pool.  Forward writes only the pooled
    activation and the pool codes; backward goes from the pooled gradient straight to the
    convolution's incoming gradient (csrc/bn.hip, "block ends": 5.75 instead of 10.5
    activation-sized HBM passes for the layer)."""

    @staticmethod
    def forward(ctx, y, weight, bias, layer):
        arena = _arena_of(layer)
        y = y.contiguous()
        zp, codes, stats = K.bn_fwd_train_pool(y, arena.flat_slice(arena.flat_master, weight),
                                               arena.flat_slice(arena.flat_master, bias), layer.running_mean,
                                               layer.running_var, layer.eps, layer.momentum)
        ctx.layer = layer
        ctx.save_for_backward(y, codes, stats, weight, bias)
        ctx.mark_non_differentiable(codes)
        return zp

    @staticmethod
    def backward(ctx, dzp):
        layer = ctx.layer
        arena = _arena_of(layer)
        y, codes, stats, weight, bias = ctx.saved_tensors
        dy = K.bn_bwd_pool(y, dzp.contiguous(), codes, arena.flat_slice(arena.flat_master, weight), stats,
                           arena.grad(weight), arena.grad(bias), arena.grad_beta(weight), arena.grad_beta(bias))
        arena.notify_grad_ready(bias)
        arena.notify_grad_ready(weight)
        return dy, None, None, None


FUSE_BN_POOL = os.environ.get("DTP_FUSE_BN_POOL", "1") != "0"      # A/B switch


def bn_relu_pool(bn, y):
    """``maxpool2x2(relu(bn(y)))`` in training mode as the fused kernel pair, or None when the shape
    (or the mode) does not allow it and the caller runs the three modules one after the other."""
    if not (FUSE_BN_POOL and bn.training and bn.relu and y.is_cuda and K.bn_pool_fusable(y)):
        return None
    bn.num_batches_tracked += 1
    bn.stats_epoch += 1             # the kernels move the running statistics through raw pointers
    return _BatchNormReLUPoolFn.apply(y, bn.weight, bn.bias, bn)


class BatchNormReLU(FusedLayer):
    """BatchNorm2d over the channel axis of the internal NHWC bf16 layout with the following ReLU
    fused (VGG16-BN, BASELINE.json config 5).  Parameter and buffer names are ``nn.BatchNorm2d``'s
    (``weight, bias, running_mean, running_var, num_batches_tracked``), so checkpoints interchange
    with a stock ``torch.nn`` VGG16-BN.  GPU: hand-written kernels in training AND eval (eval-mode
    normalisation is folded into the preceding convolution by ``ConvBlock`` when no gradient is
    needed); CPU: ``F.batch_norm`` reference."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, relu=True):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.relu = num_features, float(eps), float(momentum), bool(relu)
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        # bumped whenever the running statistics change behind torch's back (the kernels and the native
        # RCCL broadcast write them through raw pointers: tensor._version does not move) — part of the
        # key of the folded-weights cache of conv_bn_relu_folded
        self.stats_epoch = 0

    def eval_coeffs(self):
        """(scale, shift) of the eval-mode affine map, from the fp32 master parameters."""
        arena = _arena_of(self)
        return K.bn_eval_coeffs(arena.flat_slice(arena.flat_master, self.weight),
                                arena.flat_slice(arena.flat_master, self.bias), self.running_mean, self.running_var,
                                self.eps)

    def forward(self, x):
        if not x.is_cuda:
            y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training,
                             self.momentum, self.eps)
            if self.training:
                self.num_batches_tracked += 1
            return F.relu(y) if self.relu else y
        if self.training:
            self.num_batches_tracked += 1
            self.stats_epoch += 1
            z = _BatchNormReLUFn.apply(x, self.weight, self.bias, self)
            src = self.__dict__.pop("_bn_src_tmp", None)
            if src is not None:
                z._dtp_bn_src = src
            return z
        arena = _arena_of(self)
        if torch.is_grad_enabled() and x.requires_grad:
            raise RuntimeError("eval-mode BatchNormReLU on the GPU is inference-only (no autograd)")
        return K.bn_fwd_eval(x.contiguous(), arena.flat_slice(arena.flat_master, self.weight),
                             arena.flat_slice(arena.flat_master, self.bias), self.running_mean, self.running_var,
                             self.eps, self.relu)

    def extra_repr(self):
        return "%d, eps=%g, momentum=%g, relu=%s" % (self.num_features, self.eps, self.momentum, self.relu)


def conv_bn_relu_folded(conv, bn, x, pool=False):
    """Inference: ``relu(bn(conv(x)))`` as ONE convolution kernel — the eval-mode scale is folded
    into the bf16 filter copy and scale/shift into the bias (cached until the weights or the
    running statistics change), so BatchNorm costs nothing at evaluation time.
    ``pool``: the block's 2x2 max-pool follows; where the shapes allow it (64 input channels) it
    joins the same kernel and ``(y, True)`` is returned, else ``(y, False)`` with y un-pooled."""
    arena = _arena_of(conv)
    key = (arena.lp_epoch, bn.stats_epoch, bn.running_mean._version, bn.running_var._version)
    cached = conv.__dict__.get("_bnfold")
    if cached is None or cached[0] != key:
        scale, shift = bn.eval_coeffs()
        w = arena.lp(conv.weight)
        wf = (w.float() * scale.view(-1, 1)).to(BF16).contiguous()
        bf = (conv.bias.detach().float() * scale + shift).contiguous()
        cached = (key, wf, bf)
        conv.__dict__["_bnfold"] = cached
    _, wf, bf = cached
    if isinstance(conv, ConvFirst):
        x = x.contiguous()
        if x.shape[1] != 3:
            return None
        y = K.conv1_fprop(x, wf, bf, bn.relu, conv.in_scale, conv.in_mean, conv.in_inv_std)
        return (y, False) if pool else y
    if pool:
        if bn.relu and x.dim() == 4 and K.conv_pool_fusable(x.shape[3], conv.out_channels, x.shape[1], x.shape[2],
                                                            N=x.shape[0]):
            return K.conv3x3_relu_pool_fprop(x, wf, bf)[0], True
        return K.conv3x3_fprop(x, wf, bf, relu=bn.relu), False
    return K.conv3x3_fprop(x, wf, bf, relu=bn.relu)


class ReLUMarker(nn.Module):
    """Parameter-less placeholder that keeps the reference's nn.Sequential indices (a ReLU sits at
    every odd index of ConvBlock.conv, reference model/vgg16.py:10,13) — the ReLU itself is fused
    into the preceding kernel.  ``active=True`` makes it a real ReLU (unfused producer)."""

    def __init__(self, active=False):
        super().__init__()
        self.active = active

    def forward(self, x):
        return F.relu(x) if self.active else x


# =============================================================================== loss
class _SoftmaxXentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, holder):
        out2, dl32, _ = K.softmax_xent(logits.contiguous(), labels.contiguous(), want_bf16_grad=False)
        ctx.save_for_backward(dl32)
        holder.last_stats = out2
        return out2[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dl32,) = ctx.saved_tensors
        return dl32 * g, None, None


class SoftmaxCrossEntropy(nn.Module):
    """Mean softmax cross-entropy; on GPU one fused kernel yields loss, d(logits) and the number of
    top-1 hits (``last_stats`` = device tensor [loss, hits], no host sync)."""

    def __init__(self):
        super().__init__()
        self.last_stats = None

    def forward(self, logits, labels):
        if not logits.is_cuda:
            return F.cross_entropy(logits.float(), labels, reduction="mean")
        return _SoftmaxXentFn.apply(logits.float(), labels, self)


def top1_hits(logits, labels):
    """Number of correct top-1 predictions as a device scalar (argmax on logits: softmax is
    monotone)."""
    if not logits.is_cuda:
        return (logits.argmax(dim=-1) == labels).sum().float()
    out2, _, _ = K.softmax_xent(logits.float().contiguous(), labels.contiguous(), want_bf16_grad=False)
    return out2[1]
