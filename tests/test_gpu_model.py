"""End-to-end single-GPU checks of the fused VGG16 path (SURVEY.md §4.2 'End-to-end single GPU'):
GPU (HIP kernels, bf16) forward/backward against the CPU fp32 reference path of the *same*
modules and weights; loss decreases on a learnable synthetic set; checkpoint round trip."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(size=32, classes=3, seed=0, grad_dtype=torch.float32):
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    torch.manual_seed(seed)
    cpu = VGG16(3, classes, True, dropout=0.0)
    gpu = VGG16(3, classes, True, dropout=0.0, grad_dtype=grad_dtype)
    gpu.load_state_dict(cpu.state_dict())
    return cpu, gpu.to("cuda")


def test_native_library_is_loaded():
    from distributed_training_pytorch_amd import _lib
    _lib.hip()
    maps = open("/proc/self/maps").read()
    assert "libdtp_hip.so" in maps


class _RoundBF16(torch.autograd.Function):
    """Straight-through bf16 rounding of activations (forward) and their gradients (backward):
    the rounding points of the fused kernels, so ReLU masks agree with the GPU path."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _emulated_reference(sd, x, y):
    """fp32 torch ops on bf16-rounded weights/activations; returns (logits, loss, grads by name)."""
    import torch.nn.functional as F
    rb = _RoundBF16.apply
    params = {}

    def P(name, lowp=True):
        t = sd[name].detach().cuda().float()
        if lowp:
            t = t.bfloat16().float()
        t.requires_grad_(True)
        params[name] = t
        return t

    h = rb(x)
    for b, n in enumerate([2, 2, 3, 3, 3]):
        for i in range(n):
            w, bias = P("backbone.%d.conv.%d.weight" % (b, 2 * i)), P("backbone.%d.conv.%d.bias" % (b, 2 * i), False)
            h = rb(F.relu(F.conv2d(h, w, bias, padding=1)))
        h = F.max_pool2d(h, 2, 2)
    h = rb(torch.flatten(F.adaptive_avg_pool2d(h, (7, 7)), 1))
    h = rb(F.relu(F.linear(h, P("linear1.weight"), P("linear1.bias", False))))
    h = rb(F.relu(F.linear(h, P("linear2.weight"), P("linear2.bias", False))))
    logits = F.linear(h, P("linear3.weight"), P("linear3.bias", False))
    loss = F.cross_entropy(logits, y)
    loss.backward()
    return logits.detach(), loss.detach(), {k: v.grad for k, v in params.items()}


@pytest.mark.parametrize("grad_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("size", [64, 96])
def test_forward_backward_matches_reference(grad_dtype, size):
    """GPU (HIP, bf16) vs (a) an fp32 torch reference with the same bf16 rounding points — tight —
    and (b) the pure fp32 CPU path of the same modules — loose: in a ReLU net ~1% of the masks flip
    under bf16 noise, which decorrelates deep-layer gradients (cos ~0.93 at conv1_1, measured)."""
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    cpu, gpu = _models(grad_dtype=grad_dtype)
    torch.manual_seed(1)
    x = torch.randn(8, 3, size, size)
    y = torch.randint(0, 3, (8,))
    crit = SoftmaxCrossEntropy()
    cpu.train(); gpu.train()
    out_c = cpu(x)
    loss_c = crit(out_c, y)
    loss_c.backward()
    out_g = gpu(x.cuda())
    loss_g = crit(out_g, y.cuda())
    loss_g.backward()
    torch.cuda.synchronize()
    assert out_g.dtype == torch.float32 and out_g.shape == out_c.shape
    scale = out_c.abs().max().item()
    assert (out_g.cpu() - out_c).abs().max().item() < 0.05 * scale + 1e-3
    assert abs(loss_g.item() - loss_c.item()) < 2e-2
    ref_logits, ref_loss, ref_grads = _emulated_reference(cpu.state_dict(), x.cuda(), y.cuda())
    assert (out_g - ref_logits).abs().max().item() < 0.01 * scale + 1e-4
    a = gpu.param_arena
    gp = dict(gpu.named_parameters())
    cos = torch.nn.functional.cosine_similarity
    for name, pc in cpu.named_parameters():
        gg = a._logical_view(a.flat_grad, a.index[id(gp[name])]).float()
        ge = ref_grads[name]
        c_emu = cos(gg.flatten(), ge.flatten(), dim=0).item()
        ratio = gg.norm().item() / (ge.norm().item() + 1e-20)
        # Per-layer thresholds from the measured minima over 12 configurations (bench/measure_grad_parity.py,
        # profiles/68_grad_parity_per_layer.jsonl): linear3's gradient has no ReLU mask of its own (0.9998);
        # every other layer's contains masks that flip where a pre-activation sits within bf16 noise of 0
        # (the net is randomly initialised: linear outputs ~1e-3) — linear2 0.986, linear1 / block 5 0.975-0.982,
        # blocks 3-4 0.969-0.975, blocks 1-2 0.961-0.974; bias gradients are 64..4096-element sums with heavy
        # cancellation and move more (0.946 at conv1_2).  The tight, mask-independent check is
        # test_layerwise_teacher_forced_gradients_are_tight below.
        if name.startswith("linear3"):
            lo, hi, cmin = 0.995, 1.005, 0.999
        elif name.startswith("linear"):
            lo, hi, cmin = 0.98, 1.02, 0.975
        elif name.startswith(("backbone.3", "backbone.4")):
            lo, hi, cmin = 0.97, 1.03, 0.96
        elif name.endswith("bias"):
            lo, hi, cmin = 0.90, 1.10, 0.93
        else:
            lo, hi, cmin = 0.95, 1.05, 0.95
        assert c_emu > cmin and lo < ratio < hi, "%s: vs emulated ref cos %.4f norm ratio %.4f" % (name, c_emu, ratio)
        c_cpu = cos(gg.flatten().cpu(), pc.grad.flatten(), dim=0).item()
        assert c_cpu > 0.85, "%s: vs fp32 CPU cos %.4f" % (name, c_cpu)


@pytest.mark.parametrize("grad_dtype", [torch.float32, torch.bfloat16])
def test_layerwise_teacher_forced_gradients_are_tight(grad_dtype):
    """The model-level check above compares against a reference that re-runs the WHOLE net: ~1 % of the
    ReLU masks flip under bf16 rounding and decorrelate deep-layer gradients (measured minimum cosine
    0.946-0.987, bench/measure_grad_parity.py -> profiles/68_*), so its thresholds cannot catch a 5 %
    wiring error.  This one can.  One real forward/backward of the fused VGG16 (conv+pool fusion off so
    that every layer runs through its module); hooks capture each layer's GPU input, output and incoming
    gradient; every layer is then recomputed on its OWN captured tensors with plain fp32 torch ops —
    same operand bits, same masks (the GPU activation's own sign), only the summation order differs —
    and its weight, bias and data gradient must agree to cosine >= 0.9999 (fp32 gradient arena; 0.999
    with a bf16 arena: one rounding per element) and a norm ratio within 0.5 %.  Checks the contracts
    between layers: which tensor is saved, who applies whose ReLU mask (grad_premasked /
    input_is_relu_out / conv1_1's self-mask), bias gradients from producer epilogues, flatten order."""
    import torch.nn.functional as F
    from distributed_training_pytorch_amd.ops import kernels as K
    from distributed_training_pytorch_amd.ops.layers import Conv3x3, ConvFirst, Linear, SoftmaxCrossEntropy
    _, gpu = _models(size=64, grad_dtype=grad_dtype, seed=3)
    torch.manual_seed(4)
    x = torch.randn(8, 3, 64, 64, device="cuda")
    y = torch.randint(0, 3, (8,), device="cuda")
    cap = {}

    def hook(name):
        def fn(mod, inp, out):
            rec = cap.setdefault(name, {})
            rec["x"], rec["y"] = inp[0].detach(), out.detach()
            out.register_hook(lambda g, rec=rec: rec.__setitem__("gy", g.detach()))
            if inp[0].requires_grad:
                inp[0].register_hook(lambda g, rec=rec: rec.__setitem__("gx", g.detach()))
        return fn

    layers = [(n, m) for n, m in gpu.named_modules() if isinstance(m, (Conv3x3, ConvFirst, Linear))]
    handles = [m.register_forward_hook(hook(n)) for n, m in layers]
    K.fuse_conv_pool = False
    try:
        gpu.train()
        SoftmaxCrossEntropy()(gpu(x), y).backward()
        torch.cuda.synchronize()
    finally:
        K.fuse_conv_pool = True
        for h in handles:
            h.remove()
    a = gpu.param_arena
    cos = lambda u, v: F.cosine_similarity(u.flatten().double(), v.flatten().double(), dim=0).item()   # noqa: E731
    cmin = 0.9999 if grad_dtype == torch.float32 else 0.999
    checked = 0
    for name, m in layers:
        rec = cap[name]
        w = m.weight.detach().bfloat16().float()
        b = m.bias.detach().float()
        gy = rec["gy"].float()
        yv = rec["y"].float()
        if isinstance(m, Linear):
            xin = rec["x"].float()
            g = gy if (m.grad_premasked or not m.relu) else gy * (yv > 0)
            pre = xin @ w.t() + b
            ref_y = F.relu(pre) if m.relu else pre
            gw, gb, gx = g.t() @ xin, g.sum(0), g @ w
            if m.input_is_relu_out:
                gx = gx * (xin > 0)
        else:
            if isinstance(m, ConvFirst):
                xin = rec["x"].float().bfloat16().float()                    # NCHW image, rounded by the ingest
                y_nchw, g_nchw = yv.permute(0, 3, 1, 2), gy.permute(0, 3, 1, 2)
            else:
                xin = rec["x"].float().permute(0, 3, 1, 2)                   # NHWC bf16 activation
                y_nchw, g_nchw = yv.permute(0, 3, 1, 2), gy.permute(0, 3, 1, 2)
            g = g_nchw if (m.grad_premasked or not m.relu) else g_nchw * (y_nchw > 0)
            pre = F.conv2d(xin, w, b, padding=1)
            ref_y = (F.relu(pre) if m.relu else pre)
            gw = torch.nn.grad.conv2d_weight(xin, w.shape, g.contiguous(), padding=1)
            gb = g.sum((0, 2, 3))
            gx = F.conv_transpose2d(g.contiguous(), w, padding=1)
            if getattr(m, "input_is_relu_out", False):
                gx = gx * (xin > 0)
            ref_y, gx = ref_y.permute(0, 2, 3, 1), gx.permute(0, 2, 3, 1)
        # forward: one bf16 rounding (fp32 logits: accumulation order only)
        err = (yv - ref_y).abs().max().item()
        assert err <= 0.01 * ref_y.abs().max().item() + 1e-5, "%s forward: max err %g" % (name, err)
        got_w = a._logical_view(a.flat_grad, a.index[id(m.weight)]).float()
        got_b = a._logical_view(a.flat_grad, a.index[id(m.bias)]).float()
        for what, got, ref in (("weight", got_w, gw), ("bias", got_b, gb)):
            c, r = cos(got, ref), got.norm().item() / (ref.norm().item() + 1e-30)
            assert c >= cmin and abs(r - 1) < 5e-3, "%s %s gradient: cos %.6f norm ratio %.5f" % (name, what, c, r)
        if "gx" in rec:
            c, r = cos(rec["gx"].float(), gx), rec["gx"].float().norm().item() / (gx.norm().item() + 1e-30)
            assert c >= 0.9999 and abs(r - 1) < 5e-3, "%s data gradient: cos %.6f norm ratio %.5f" % (name, c, r)
        checked += 1
    assert checked == 16


def test_training_reduces_loss_and_checkpoint_roundtrip(tmp_path):
    from distributed_training_pytorch_amd.data.synthetic import SyntheticImageDataset
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    from distributed_training_pytorch_amd.utils import checkpoint as ckpt
    torch.manual_seed(0)
    ds = SyntheticImageDataset(64, 3, 32, 32, seed=0, signal=2.0)
    xs = torch.stack([ds[i][0] for i in range(64)]).cuda()
    ys = torch.tensor([ds[i][1] for i in range(64)]).cuda()
    model = VGG16(3, 3, True, dropout=0.0).cuda().train()
    opt = FusedSGD(model, lr=0.004, momentum=0.9, weight_decay=1e-4)
    crit = SoftmaxCrossEntropy()
    losses = []
    for it in range(60):
        opt.zero_grad()
        loss = crit(model(xs), ys)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    # full-batch momentum SGD on 64 learnable samples: the loss must come down and stay down
    assert min(losses[20:]) < 0.6 * losses[0] and sum(losses[-5:]) / 5 < 0.85 * losses[0], losses[::5]
    # checkpoint in the reference schema, reload into a fresh model: identical logits in eval
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [50, 100, 200], gamma=0.1)
    path = str(tmp_path / "weights" / "last.pth")
    ckpt.save_snapshot(path, 3, model, opt, sched)
    snap = ckpt.load_snapshot(path)
    assert set(ckpt.REQUIRED_KEYS) <= set(snap.keys())
    assert snap["model_state_dict"]["backbone.0.conv.0.weight"].shape == (64, 3, 3, 3)
    assert snap["model_state_dict"]["backbone.0.conv.0.weight"].is_contiguous()
    m2 = VGG16(3, 3, True)
    m2.load_state_dict(snap["model_state_dict"])
    m2 = m2.cuda().eval()
    model.eval()
    with torch.no_grad():
        assert torch.equal(model(xs), m2(xs))
    opt2 = FusedSGD(m2, lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt2.load_state_dict(snap["optimizer_state_dict"])
    assert torch.equal(opt2._flat_state["momentum_buffer"], opt._flat_state["momentum_buffer"])


def test_grad_accumulation_equals_big_batch():
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    _, gpu = _models(seed=3)
    crit = SoftmaxCrossEntropy()
    torch.manual_seed(4)
    x = torch.randn(16, 3, 32, 32, device="cuda")
    y = torch.randint(0, 3, (16,), device="cuda")
    a = gpu.param_arena
    a.zero_grad()
    crit(gpu(x), y).backward()
    ref = a.flat_grad.clone()
    a.zero_grad()
    (crit(gpu(x[:8]), y[:8]) / 2).backward()
    (crit(gpu(x[8:]), y[8:]) / 2).backward()
    torch.cuda.synchronize()
    err = (a.flat_grad - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
    assert err < 0.03, err


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_native_rccl_transport_world1_and_reducer_over_it():
    """World-size-1 RCCL self all-reduce on the one GPU: exercises communicator creation, the
    dedicated comm stream and the event hand-offs of csrc/runtime/rccl_transport.cpp (multi-GPU
    runs are the driver's; the bucket logic itself is covered over gloo on CPU)."""
    from distributed_training_pytorch_amd.parallel.rccl_native import NativeRcclTransport
    from distributed_training_pytorch_amd.parallel.reducer import BucketedReducer
    t = NativeRcclTransport(rank=0, world=1)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(1 << 20, device="cuda").to(dt)
        ref = x.clone()
        h = t.launch(x)
        t.wait(h, x)
        torch.cuda.synchronize()
        assert torch.equal(x, ref)                      # average over one rank
    assert t.stats()["launched"] == 2
    assert "libdtp_runtime.so" in open("/proc/self/maps").read()
    _, gpu = _models(seed=5)
    a = gpu.param_arena
    red = BucketedReducer(a.params, flat_grad=a.flat_grad, flat_offsets=a.offsets, transport=t,
                          bucket_cap_mb=64, first_bucket_mb=8, split_large_mb=128)
    red.world = 1
    a.reducer = red
    assert len(red.buckets) >= 5                        # linear1 split + conv buckets
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    crit = SoftmaxCrossEntropy()
    x = torch.randn(8, 3, 32, 32, device="cuda")
    y = torch.randint(0, 3, (8,), device="cuda")
    # order of events: every ready notification must come before its bucket's launch, once per
    # parameter (autograd's AccumulateGrad fires the hook a second time for kernel-written params)
    trace, _mr, _la = [], red.mark_ready, red._launch
    red.mark_ready = lambda p: (trace.append(("ready", id(p))), _mr(p))
    red._launch = lambda b: (trace.append(("launch", b.index)), _la(b))

    def run(poison):
        a.zero_grad()
        if poison:                                      # a bucket reduced (read-modify-write on the
            a.flat_grad.fill_(7.0)                      # comm stream) before its last gradient is
            torch.cuda.synchronize()                    # written would carry the poison through
        crit(gpu(x), y).backward()
        torch.cuda.synchronize()
        return a.flat_grad.clone()

    g1 = run(True)
    assert red.stats["launched"] == len(red.buckets) and red.stats["finalized"] == 1
    assert red._core is not None                    # bucket counting / launching ran in the C++ scheduler
    cs = red._core.stats()
    assert cs["passes"] == 1 and cs["launched"] == len(red.buckets)
    seen = set()
    for kind, v in trace:
        if kind == "ready":
            seen.add(v)
        else:
            assert red.buckets[v].param_ids <= seen, "bucket %d launched before its gradients" % v
    g2 = run(True)
    a.reducer = None
    ref = run(False)
    for i, p in enumerate(a.params):                    # padding between tensors keeps the poison
        s = slice(a.offsets[i], a.offsets[i] + p.numel())
        assert torch.equal(g1[s], ref[s]) and torch.equal(g2[s], ref[s]), a.names[i]
    t.close()


def test_fused_bias_grads_match_standalone_colsum():
    """Whole model: bias gradients assembled from producer-epilogue partial sums (dgrad / pool
    backward) equal the ones from the stand-alone column-sum kernel; weight grads are untouched."""
    from distributed_training_pytorch_amd.ops import kernels as K
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    _, gpu = _models(seed=7)
    a = gpu.param_arena
    crit = SoftmaxCrossEntropy()
    torch.manual_seed(8)
    x = torch.randn(8, 3, 64, 64, device="cuda")
    y = torch.randint(0, 3, (8,), device="cuda")

    def run():
        a.zero_grad()
        crit(gpu(x), y).backward()
        torch.cuda.synchronize()
        return a.flat_grad.clone()

    fused = run()
    K.fuse_colsum = False
    try:
        plain = run()
    finally:
        K.fuse_colsum = True
    for i, (n, p) in enumerate(zip(a.names, a.params)):
        s = slice(a.offsets[i], a.offsets[i] + p.numel())
        if n.endswith("weight"):
            assert torch.equal(fused[s], plain[s]), n
        else:
            ref = plain[s].abs().max().item()
            assert (fused[s] - plain[s]).abs().max().item() <= 1e-4 * ref + 1e-7, n


def test_relu_bitmask_model_gradients_bit_identical():
    """Whole model: data gradients masked from the 1-bit (y > 0) masks the forward kernels pack are
    bit-identical to the ones masked by re-reading the activation (DTP_RELU_BITMASK A/B switch),
    and the masked convolutions really received the bits."""
    from distributed_training_pytorch_amd.ops import kernels as K
    from distributed_training_pytorch_amd.ops.layers import Conv3x3, SoftmaxCrossEntropy
    _, gpu = _models(seed=9)
    gpu.eval()                                   # dropout off: both runs see the same masks
    a = gpu.param_arena
    crit = SoftmaxCrossEntropy()
    torch.manual_seed(10)
    x = torch.randn(8, 3, 64, 64, device="cuda")
    y = torch.randint(0, 3, (8,), device="cuda")
    seen = []
    orig = K.conv3x3_dgrad

    def spy(*args, **kw):
        seen.append(kw.get("mask_bits") is not None)
        return orig(*args, **kw)

    def run():
        a.zero_grad()
        crit(gpu(x), y).backward()
        torch.cuda.synchronize()
        return a.flat_grad.clone()

    K.conv3x3_dgrad = spy
    n_masked = sum(1 for m in gpu.modules() if isinstance(m, Conv3x3) and m.input_is_relu_out)
    assert n_masked >= 7
    try:
        # (a) the shipped tile picker (small grids: split-K producers emit no bits, conv_c64 does);
        # (b) single-pass 128x128 tiles everywhere: every masked convolution must receive bits
        for override, expect in ((None, 1), ((3, 1), n_masked)):
            if override is not None:
                K._overrides["igemm"] = lambda M, Nc, ks, o=override: o
            assert K.relu_bitmask
            del seen[:]
            with_bits = run()
            assert sum(seen) >= expect, (override, seen)
            K.relu_bitmask = False
            del seen[:]
            plain = run()
            K.relu_bitmask = True
            assert not any(seen)
            assert torch.equal(with_bits, plain), override
    finally:
        K.relu_bitmask = True
        K.conv3x3_dgrad = orig
        K._overrides.pop("igemm", None)


def test_unpooling_data_gradient_model_gradients_bit_identical():
    """Whole model: with the pool backward folded into the data gradient of the convolution above it
    (DTP_FUSE_UNPOOL A/B switch) every gradient is bit-identical to the separate pool-backward
    kernel, the four pools between convolutions really took the fused path, and a second consumer
    of a pooled activation is refused instead of producing a wrong sum."""
    from distributed_training_pytorch_amd.ops import kernels as K
    from distributed_training_pytorch_amd.ops import layers as L
    _, gpu = _models(seed=12)
    gpu.eval()                                   # dropout off: both runs see the same masks
    a = gpu.param_arena
    crit = L.SoftmaxCrossEntropy()
    torch.manual_seed(13)
    x = torch.randn(8, 3, 64, 64, device="cuda")
    y = torch.randint(0, 3, (8,), device="cuda")
    pool_bwd_calls = []
    orig = K.maxpool2x2_bwd

    def spy(*args, **kw):
        pool_bwd_calls.append(tuple(kw.get("in_shape") or ()))
        return orig(*args, **kw)

    def run():
        a.zero_grad()
        crit(gpu(x), y).backward()
        torch.cuda.synchronize()
        return {n: a.grad(p).clone() for n, p in gpu.named_parameters()}

    def same(g1, g2):
        # weight gradients bit for bit; a bias gradient is a column sum whose f32 partial rows now come from
        # the GEMM's tiles instead of the pool kernel's blocks: same addends, another order
        for n in g1:
            if n.endswith("bias"):
                torch.testing.assert_close(g1[n], g2[n], rtol=1e-4, atol=1e-5 * max(1.0, g2[n].abs().max().item()), msg=n)
            else:
                assert torch.equal(g1[n], g2[n]), n

    K.maxpool2x2_bwd = spy
    K._overrides["igemm"] = lambda M, Nc, ks: (3, 1)      # single-pass tiles (tiny grids would take split-K)
    try:
        assert K.fuse_unpool
        fused = run()
        n_fused = len(pool_bwd_calls)
        K.fuse_unpool = False
        del pool_bwd_calls[:]
        plain = run()
        n_plain = len(pool_bwd_calls)
        K.fuse_unpool = True
        assert n_plain == 5 and n_fused == 1, (n_plain, n_fused)      # only the pool under the classifier stays a kernel
        same(fused, plain)
        # the shipped picker on this tiny grid: whatever mix of fused / separate it chooses, same bits
        K._overrides.pop("igemm")
        shipped = run()
        K.fuse_unpool = False
        same(shipped, run())
    finally:
        K.fuse_unpool = True
        K.maxpool2x2_bwd = orig
        K._overrides.pop("igemm", None)

    # a pooled activation with two consumers: autograd sums the placeholder with the other gradient
    from distributed_training_pytorch_amd.models.fused_model import attach_arena
    conv = L.Conv3x3(64, 128).cuda()
    attach_arena(conv)
    pool = L.MaxPool2x2(relu_mask=True)
    xin = torch.randn(2, 16, 16, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    K._overrides["igemm"] = lambda M, Nc, ks: (3, 1)
    try:
        p = pool(xin)
        out = conv(p).float().sum() + p.float().sum()
        with pytest.raises(RuntimeError, match="more than one consumer"):
            out.backward()
    finally:
        K._overrides.pop("igemm", None)


def test_no_uninitialised_reads_with_poisoned_allocator():
    """Every free block of the caching allocator (large and small pools) and every persistent
    workspace is filled with NaN before each step: a kernel that consumes memory neither it nor its
    producer wrote (a split-K slab of an empty split, a partial-sum row past the grid, a padded
    column) would turn the loss or a gradient into NaN."""
    from distributed_training_pytorch_amd import _lib
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD

    def poison():
        big = torch.full((256 * 1024 * 1024 // 4,), float("nan"), device="cuda")
        mids = [torch.full((sz // 4,), float("nan"), device="cuda") for sz in (1 << 20, 4 << 20, 16 << 20) for _ in range(6)]
        small = [torch.full((16 * 1024,), float("nan"), device="cuda") for _ in range(1500)]
        for key, t in _lib.workspace._bufs.items():
            if key[0] != "zero_page" and t.is_floating_point():
                t.fill_(float("nan"))
        torch.cuda.synchronize()
        del big, mids, small

    for size, batch in ((32, 4), (64, 8)):
        torch.manual_seed(23)
        m = VGG16(3, 3, True, dropout=0.3).to("cuda").train()
        o = FusedSGD(m, lr=0.002, momentum=0.9, weight_decay=1e-4)
        c = SoftmaxCrossEntropy()
        a = m.param_arena
        for step in range(3):
            x = torch.randn(batch, 3, size, size, device="cuda")
            y = torch.randint(0, 3, (batch,), device="cuda")
            poison()
            o.zero_grad()
            loss = c(m(x), y)
            loss.backward()
            torch.cuda.synchronize()
            assert bool(torch.isfinite(loss.detach())), (size, step)
            assert bool(torch.isfinite(a.flat_grad.float()).all()), (size, step)
            o.step()
        assert bool(torch.isfinite(a.flat_master).all())


def test_hip_graph_step_matches_eager_and_redraws_dropout():
    """GraphedTrainStep: (a) with dropout off, N graph replays == N eager steps bit for bit;
    (b) with dropout on, replays draw different masks (device-resident seed) and still train;
    (c) an LR change re-captures."""
    from distributed_training_pytorch_amd.engine.graph import GraphedTrainStep
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    torch.manual_seed(21)
    xs = [torch.randn(4, 3, 32, 32, device="cuda") for _ in range(6)]
    ys = [torch.randint(0, 3, (4,), device="cuda") for _ in range(6)]

    def make(dropout):
        torch.manual_seed(22)
        m = VGG16(3, 3, True, dropout=dropout).to("cuda").train()
        return m, FusedSGD(m, lr=0.002, momentum=0.9, weight_decay=1e-4), SoftmaxCrossEntropy()

    m1, o1, c1 = make(0.0)
    m2, o2, c2 = make(0.0)
    g = GraphedTrainStep(m2, c2, o2, warmup=2)
    assert g.enabled
    for i, (x, y) in enumerate(zip(xs, ys)):
        o1.zero_grad()
        l1 = c1(m1(x), y)
        l1.backward()
        o1.step()
        l2 = g(x, y)
        assert torch.equal(l1.detach(), l2), "step %d: eager %r graph %r" % (i, float(l1), float(l2))
    torch.cuda.synchronize()
    assert g.captures == 1 and g.replays == 4
    assert torch.equal(m1.param_arena.flat_master, m2.param_arena.flat_master)
    assert o2._steps == o1._steps == 6
    for grp in o2.param_groups:                       # MultiStepLR milestone
        grp["lr"] = 0.001
    for grp in o1.param_groups:
        grp["lr"] = 0.001
    for x, y in zip(xs[:4], ys[:4]):
        o1.zero_grad()
        c1(m1(x), y).backward()
        o1.step()
        g(x, y)
    torch.cuda.synchronize()
    assert g.captures == 2
    assert torch.equal(m1.param_arena.flat_master, m2.param_arena.flat_master)
    # eval after graph steps sees the updated weights
    m1.eval(); m2.eval()
    with torch.no_grad():
        assert torch.equal(m1(xs[0]), m2(xs[0]))

    m3, o3, c3 = make(0.3)
    for grp in o3.param_groups:                       # frozen weights: only the masks can move the loss
        grp["lr"], grp["weight_decay"] = 0.0, 0.0
    g3 = GraphedTrainStep(m3, c3, o3, warmup=1)
    losses = [float(g3(xs[0], ys[0])) for _ in range(8)]
    assert g3.captures == 1 and g3.replays == 7
    assert all(l == l for l in losses)
    assert len(set(losses[1:])) >= 6                  # every replay drew a fresh dropout mask


def test_dropout_rng_is_run_state_resume_is_bit_exact_and_ranks_differ(tmp_path):
    """Dropout key = (seed, rank, forward-call number, layer) — ops/rng.py, reference model/vgg16.py:40,73-76:
    (a) train 4 steps == train 2 + snapshot + fresh model/optimizer + resume + 2, bit for bit, dropout 0.3;
    (b) the hipGraph step draws the SAME masks as the eager path (device copy of the same state);
    (c) two ranks draw different masks, two models with the same (seed, rank) the same ones."""
    from distributed_training_pytorch_amd.engine.graph import GraphedTrainStep
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    from distributed_training_pytorch_amd.utils import checkpoint as ckpt
    torch.manual_seed(31)
    xs = [torch.randn(4, 3, 32, 32, device="cuda") for _ in range(6)]
    ys = [torch.randint(0, 3, (4,), device="cuda") for _ in range(6)]

    def make(rank=0):
        torch.manual_seed(32)
        m = VGG16(3, 3, True, dropout=0.3).to("cuda").train()
        m.dropout_rng.seed, m.dropout_rng.rank = 1234, rank
        return m, FusedSGD(m, lr=0.002, momentum=0.9, weight_decay=1e-4), SoftmaxCrossEntropy()

    def steps(m, o, c, lo, hi):
        out = []
        for x, y in zip(xs[lo:hi], ys[lo:hi]):
            o.zero_grad()
            loss = c(m(x), y)
            loss.backward()
            o.step()
            out.append(float(loss.detach()))
        return out

    m1, o1, c1 = make()
    ref = steps(m1, o1, c1, 0, 4)
    # (a) resume
    m2, o2, c2 = make()
    got = steps(m2, o2, c2, 0, 2)
    path = str(tmp_path / "mid.pth")
    ckpt.save_snapshot(path, 0, m2, o2, None, extra=dict(rng_state=m2.dropout_rng.state_dict()))
    snap = ckpt.load_snapshot(path)
    m3, o3, c3 = make()
    m3.dropout_rng.seed = 999                        # must come back from the snapshot
    m3.load_state_dict(snap["model_state_dict"])
    o3.load_state_dict(snap["optimizer_state_dict"])
    m3.dropout_rng.load_state_dict(snap["rng_state"])
    got += steps(m3, o3, c3, 2, 4)
    assert got == ref, (got, ref)
    assert torch.equal(m1.param_arena.flat_master, m3.param_arena.flat_master)
    # (b) hipGraph == eager with dropout on
    m4, o4, c4 = make()
    g = GraphedTrainStep(m4, c4, o4, warmup=1)
    lg = [float(g(x, y)) for x, y in zip(xs[:4], ys[:4])]
    assert g.captures == 1 and g.replays == 3
    assert lg == ref, (lg, ref)
    assert torch.equal(m1.param_arena.flat_master, m4.param_arena.flat_master)
    assert m4.dropout_rng.calls == m1.dropout_rng.calls == 4
    # (c) ranks
    m5, o5, c5 = make(rank=1)
    other = steps(m5, o5, c5, 0, 1)
    assert other[0] != ref[0], "rank 1 drew rank 0's dropout mask"


def test_hip_graph_step_matches_eager_vgg16_bn():
    """VGG16-BN under GraphedTrainStep == eager, bit for bit, running statistics included: the
    BatchNorm statistic hand-overs (producer epilogue -> bn.hip, host-side slots) are decided at
    capture time and must replay exactly; and the same model with the hand-overs switched off
    lands within rounding of it."""
    from distributed_training_pytorch_amd.engine.graph import GraphedTrainStep
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops import kernels as K
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    torch.manual_seed(31)
    xs = [torch.randn(8, 3, 64, 64, device="cuda") for _ in range(5)]
    ys = [torch.randint(0, 3, (8,), device="cuda") for _ in range(5)]

    def make():
        torch.manual_seed(32)
        m = VGG16(3, 3, True, batch_norm=True, dropout=0.0).to("cuda").train()
        return m, FusedSGD(m, lr=0.002, momentum=0.9, weight_decay=1e-4), SoftmaxCrossEntropy()

    def eager(m, o, c):
        out = []
        for x, y in zip(xs, ys):
            o.zero_grad()
            l = c(m(x), y)
            l.backward()
            o.step()
            out.append(l.detach().clone())
        return out

    m1, o1, c1 = make()
    l1 = eager(m1, o1, c1)
    m2, o2, c2 = make()
    g = GraphedTrainStep(m2, c2, o2, warmup=2)
    assert g.enabled
    for i, (x, y) in enumerate(zip(xs, ys)):
        l2 = g(x, y)
        assert torch.equal(l1[i], l2), "step %d: eager %r graph %r" % (i, float(l1[i]), float(l2))
    torch.cuda.synchronize()
    assert g.captures == 1 and g.replays == 3
    assert torch.equal(m1.param_arena.flat_master, m2.param_arena.flat_master)
    for (n1, b1), (n2, b2) in zip(m1.named_buffers(), m2.named_buffers()):
        assert torch.equal(b1, b2), n1
    # hand-overs off: the stand-alone reductions, same training up to f32 summation order
    K.fuse_bn_stats = False
    try:
        m3, o3, c3 = make()
        l3 = eager(m3, o3, c3)
    finally:
        K.fuse_bn_stats = True
    # (a different f32 summation order flips bf16 roundings of the normalised activations; on random data
    # at batch 8 that moves the first loss by ~0.2 % and the fifth by ~2 %: 1.0687 / 1.0668, 1.5245 / 1.5544)
    assert all(abs(float(a) - float(b)) < 5e-2 * max(1.0, abs(float(a))) for a, b in zip(l1, l3))
    assert abs(float(l1[0]) - float(l3[0])) < 1e-2
    # (the statistics themselves are compared at 2e-5 in tests/test_gpu_kernels.py; five chaotic steps apart
    # the running means differ in the second decimal)


@pytest.mark.parametrize("grad_dtype", [torch.float32, torch.bfloat16])
def test_vgg16_bn_trains_on_gpu(grad_dtype):
    """VGG16-BN (BASELINE config 5 model): fused convs/linears + the hand-written BatchNorm+ReLU
    kernels (csrc/bn.hip); BN affine gradients are written into the flat arena by the kernels in
    both gradient dtypes; the eval-mode forward (conv + BN + ReLU folded into one convolution)
    agrees with the fp32 CPU path of the same weights."""
    from distributed_training_pytorch_amd.models.vgg16 import vgg16_bn
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    torch.manual_seed(31)
    cpu = vgg16_bn(3, 3, dropout=0.0)
    gpu = vgg16_bn(3, 3, dropout=0.0, grad_dtype=grad_dtype)
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.to("cuda")
    assert "backbone.0.conv.1.running_mean" in gpu.state_dict()
    x = torch.randn(16, 3, 32, 32)
    y = torch.randint(0, 3, (16,))
    cpu.eval(); gpu.eval()
    with torch.no_grad():
        oc, og = cpu(x), gpu(x.cuda())
    assert (og.cpu() - oc).abs().max().item() < 0.05 * oc.abs().max().item() + 2e-3
    gpu.train()
    opt = FusedSGD(gpu, lr=0.01, momentum=0.9)
    crit = SoftmaxCrossEntropy()
    bn_w = gpu.backbone[2].conv[1].weight
    conv_w = gpu.backbone[2].conv[0].weight
    bn0, cw0 = bn_w.detach().clone(), conv_w.detach().clone()
    xg, yg = x.cuda(), y.cuda()
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = crit(gpu(xg), yg)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses)
    assert not torch.equal(bn_w.detach(), bn0), "BatchNorm affine parameters did not train"
    assert not torch.equal(conv_w.detach(), cw0)
    assert min(losses[10:]) < 0.5 * losses[0], losses
    assert float(gpu.backbone[0].conv[1].num_batches_tracked) == 30
    # the trained model: folded eval forward == unfolded eval forward == CPU reference of its weights
    gpu.eval()
    with torch.no_grad():
        og = gpu(xg)
    cpu2 = vgg16_bn(3, 3, dropout=0.0)
    cpu2.load_state_dict(gpu.state_dict())
    cpu2.eval()
    with torch.no_grad():
        oc2 = cpu2(x)
    assert (og.cpu() - oc2).abs().max().item() < 0.05 * oc2.abs().max().item() + 5e-3
    from distributed_training_pytorch_amd.ops.layers import BatchNormReLU
    assert sum(isinstance(m, BatchNormReLU) for m in gpu.modules()) == 13
    assert all(id(p) in gpu.param_arena.kernel_written for p in gpu.parameters())   # nothing left to autograd


def test_config5_bn_fp8_train_snapshot_eval(tmp_path, capsys):
    """BASELINE.json config 5 end to end on the GPU: train VGG16-BN with the fp8 (OCP e4m3)
    classifier through the Trainer -> snapshot in the reference schema -> eval.py --batch-norm
    --fp8-classifier reports top-1 on the synthetic test split (same class prototypes, fresh noise).
    lr 0.005: at 0.02 the first steps overshoot (loss 1.8) and the eval-mode model lags the
    batch-statistics model by ~40 steps until the running statistics (momentum 0.1) have caught up
    with the weights (bench/debug_bn_eval.py); at 0.005 both agree after the first epoch."""
    import os
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    from distributed_training_pytorch_amd.trainer.example_trainer import ExampleTrainer
    torch.manual_seed(5)
    save = str(tmp_path / "runs")
    tr = ExampleTrainer(train_path="synthetic:192", val_path="synthetic:48", labels=["a", "b", "c"], height=64, width=64,
                        max_epoch=8, batch_size=32, pin_memory=True, have_validate=True,
                        save_best_for=("accuracy", "geq"), save_period=1, save_folder=save, progress=False, lr=0.005,
                        batch_norm=True, fp8_classifier=True, grad_dtype=torch.bfloat16)
    tr.train()
    assert tr.best_fitness["value"] is not None
    import eval as eval_script
    acc1, acck, n = eval_script.main(["--synthetic", "96", "--model", os.path.join(save, "weights", "last.pth"),
                                      "--image-size", "64", "--labels", "a,b,c", "--batch-size", "32", "--batch-norm",
                                      "--fp8-classifier", "--device", "cuda:0"])
    assert n == 96 and acc1 > 0.9 and acck >= acc1, (acc1, acck)
    assert "EVALUATION ACCURACY RESULTS: TOP-1=" in capsys.readouterr().out


def test_fp8_classifier_forward_and_training():
    """BASELINE config 5: VGG16-BN with fp8 classifier GEMMs (forward) + bf16 convs; logits stay
    close to the bf16 classifier's, training converges, eval top-1 runs."""
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy, top1_hits
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    torch.manual_seed(41)
    ref = VGG16(3, 3, True, batch_norm=True, dropout=0.0)
    m8 = VGG16(3, 3, True, batch_norm=True, dropout=0.0, fp8_classifier=True)
    m8.load_state_dict(ref.state_dict())
    ref, m8 = ref.to("cuda").eval(), m8.to("cuda").eval()
    x = torch.randn(16, 3, 64, 64, device="cuda")
    y = torch.randint(0, 3, (16,), device="cuda")
    with torch.no_grad():
        lo, l8 = ref(x), m8(x)
    rel = (lo - l8).norm().item() / (lo.norm().item() + 1e-12)
    assert 0 < rel < 0.25, rel                       # fp8 really ran (not bit-equal); 3 mantissa bits, two chained GEMMs
    m8.train()
    opt = FusedSGD(m8, lr=0.01, momentum=0.9)
    crit = SoftmaxCrossEntropy()
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = crit(m8(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses) and min(losses[10:]) < 0.5 * losses[0], losses
    with torch.no_grad():                            # batch statistics (train mode): the fit itself
        assert float(top1_hits(m8(x), y)) / 16 > 0.8
        m8.eval()                                    # eval top-1 pass runs on the fp8 path (running
        acc = float(top1_hits(m8(x), y)) / 16        # stats after 30 steps are still far from the batch's)
    assert 0.0 <= acc <= 1.0


def test_uint8_ingest_matches_float_input_on_gpu():
    """uint8 NCHW batch + (x*scale - mean)/std fused into the conv1 ingest kernel == feeding the
    already-normalised float batch; backward runs through the same kernel (uint8 source)."""
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    _, gpu = _models(seed=9)
    torch.manual_seed(10)
    xu = torch.randint(0, 256, (8, 3, 64, 64), dtype=torch.uint8, device="cuda")
    y = torch.randint(0, 3, (8,), device="cuda")
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    xf = (xu.float() / 255.0 - torch.tensor(mean, device="cuda").view(1, 3, 1, 1)) / torch.tensor(std, device="cuda").view(1, 3, 1, 1)
    crit = SoftmaxCrossEntropy()
    a = gpu.param_arena
    a.zero_grad()
    out_f = gpu(xf)
    crit(out_f, y).backward()
    torch.cuda.synchronize()
    g_f = a.flat_grad.clone()
    gpu.set_input_normalization(1.0 / 255.0, mean, std)
    a.zero_grad()
    out_u = gpu(xu)
    crit(out_u, y).backward()
    torch.cuda.synchronize()
    scale = out_f.abs().max().item()
    assert (out_u - out_f).abs().max().item() < 0.02 * scale + 1e-4
    cos = torch.nn.functional.cosine_similarity(a.flat_grad.flatten(), g_f.flatten(), dim=0).item()
    assert cos > 0.99, cos


@pytest.mark.parametrize("transport,world", [("torch", 2), ("auto", 2), ("torch", 4)])
def test_two_ranks_sharing_the_gpu_over_gloo(tmp_path, transport, world):
    """Multi-process GPU path on a one-GPU box: two ranks share cuda:0, process group over gloo,
    a small user-defined FusedModel, six optimizer steps; gradients and parameters must be
    identical across ranks after every all-reduce (tests/dist_worker.py::scenario_gpu_shared).
    ``auto``: RCCL cannot build a communicator over a duplicated GPU, so the native bring-up must
    fail or run into its watchdog deadline and every rank must agree on the process-group
    transport instead of hanging."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DTP_TRANSPORT=transport, DTP_NATIVE_INIT_TIMEOUT_S="15", OMP_NUM_THREADS="2",
               DTP_PG_TIMEOUT_S="60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(29750 + os.getpid() % 100 + world),
           os.path.join(root, "tests", "dist_worker.py"), "gpu_shared", str(tmp_path)]
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=150)
    assert r.returncode == 0, r.stdout[-4000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ok.")) == ["ok.%d" % i for i in range(world)]


def test_bench_preflight_rehearsal_on_a_shared_gpu_over_gloo():
    """bench.py under torch.distributed.run with two ranks on cuda:0 (gloo) and the pre-flight rehearsal
    forced: each rank first runs a throw-away child (3 steps at batch 8 on the HIP kernels, its own
    rendezvous port, the native bring-up failing over to the process group on the duplicated GPU), then
    the real run; ONE JSON line, both ranks seen, parameters agree, rehearsal recorded as passed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DTP_PREFLIGHT="force", DTP_NATIVE_INIT_TIMEOUT_S="15", OMP_NUM_THREADS="2",
               DTP_PG_TIMEOUT_S="60")
    env.pop("DTP_TRANSPORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29650 + os.getpid() % 100),
           os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo", "--batch-per-gpu", "4",
           "--image-size", "64", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-3000:]
    j = json.loads(lines[0])
    assert j["preflight"]["ok"] is True and j["preflight"]["all_ranks_ok"] is True
    assert j["ranks_seen"] == 2 and j["params_agree_across_ranks"] is True and j["dtype"] == "bf16"


def test_every_sync_mode_on_a_shared_gpu_over_gloo(tmp_path):
    """tests/dist_worker.py::scenario_gpu_modes with two ranks sharing cuda:0 over gloo: VGG16-BN with
    dropout through all-reduce, the per-bucket optimizer (default) and the sharded optimizer on the GPU
    kernels.  After every step the bf16 working copy AND the fp32 biases / BatchNorm affine values the
    kernels read from the master must agree across ranks (round 2's sharded mode left the non-owned
    fp32 slices stale: ADVICE r2, reducer.py), per-bucket == all-reduce bit for bit, sharded == all-reduce."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DTP_TRANSPORT="torch", OMP_NUM_THREADS="2", DTP_PG_TIMEOUT_S="120")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(29860 + os.getpid() % 100),
           os.path.join(root, "tests", "dist_worker.py"), "gpu_modes", str(tmp_path)]
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-6000:]
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("ok.")) == ["ok.0", "ok.1"]


def test_per_bucket_optimizer_in_the_native_scheduler_world1(monkeypatch):
    """The default data-parallel path's optimizer placement, on the real thing as far as one GPU goes:
    a world-size-1 RCCL communicator, the C++ bucket scheduler issuing ncclAllReduce AND the fused SGD
    kernel of each bucket on the communication stream (function pointer into libdtp_hip.so, real HIP
    events both ways).  Must equal the plain step bit for bit, incl. momentum, over 5 steps with a
    gradient-accumulation window."""
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    from distributed_training_pytorch_amd.parallel.data_parallel import DataParallel
    from distributed_training_pytorch_amd.parallel.rccl_native import NativeRcclTransport
    monkeypatch.setenv("DTP_BUCKET_OPT_FORCE", "1")
    torch.manual_seed(41)
    xs = [torch.randn(8, 3, 32, 32, device="cuda") for _ in range(5)]
    ys = [torch.randint(0, 3, (8,), device="cuda") for _ in range(5)]

    def run(per_bucket):
        torch.manual_seed(42)
        m = VGG16(3, 3, True, dropout=0.3, grad_dtype=torch.bfloat16).to("cuda").train()
        m.dropout_rng.seed = 5
        o = FusedSGD(m, lr=0.01, momentum=0.9, weight_decay=1e-4)
        c = SoftmaxCrossEntropy()
        tr = NativeRcclTransport(rank=0, world=1)
        dp = DataParallel(m, transport=tr, bucket_optimizer=o if per_bucket else None,
                          bucket_align_elems=None if per_bucket else 64, bucket_cap_mb=4.0, first_bucket_mb=1.0)
        assert dp.reducer._core is not None and dp.reducer.fused_update == per_bucket
        assert len(dp.reducer.buckets) >= 4
        losses = []
        for i, (x, y) in enumerate(zip(xs, ys)):
            o.zero_grad()
            if i == 2:
                with dp.no_sync():
                    (c(dp(x[:4]), y[:4]) / 2).backward()
                loss = c(dp(x[4:]), y[4:]) / 2
            else:
                loss = c(dp(x), y)
            loss.backward()
            o.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        w = m.param_arena.flat_master.clone()
        mom = o._flat_state["momentum_buffer"].clone()
        lp = m.param_arena.flat_lp.clone()
        st = dp.reducer._core.stats()
        dp.close()
        return w, mom, lp, losses, st

    w0, m0, l0, loss0, _ = run(False)
    w1, m1, l1, loss1, st = run(True)
    assert st["passes"] == 5 and st["launched"] >= 20
    assert loss0 == loss1, (loss0, loss1)
    assert torch.equal(w0, w1) and torch.equal(m0, m1) and torch.equal(l0, l1)


def test_pinned_ring_input_pipeline():
    """The native pinned ring: (a) PinnedRingSource hands out exactly the bytes its producer wrote
    (uint8 images + int64 labels), one step behind the copies it queues, without the host ever
    waiting; (b) DevicePrefetcher stages UNPINNED loader batches through ring slots (no per-batch
    pin_memory) and delivers bit-identical tensors, pinned ones bypass the ring; (c) a model trains
    from the uint8 ring with the normalisation fused into conv1."""
    from distributed_training_pytorch_amd.data.prefetch import DevicePrefetcher, PinnedRingSource
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    dev = torch.device("cuda", 0)
    src = PinnedRingSource(8, 3, 32, 32, device=dev, seed=3, n_slots=3, threads=2)
    seen = []
    for k in range(7):
        x, y = next(src)
        torch.cuda.synchronize()
        slot = k % 3
        hx = src.ring.host_view(slot, 0, 8 * 3 * 32 * 32, torch.uint8, (8, 3, 32, 32))
        hy = src.ring.host_view(slot, src.label_off, 8 * 8, torch.int64, (8,))
        assert x.dtype == torch.uint8 and torch.equal(x.cpu(), hx) and torch.equal(y.cpu(), hy)
        assert int(y.min()) >= 0 and int(y.max()) < 3
        seen.append(x.cpu().clone())
    assert not torch.equal(seen[0], seen[1]) and torch.equal(seen[0], seen[3])     # 3 distinct slots, cycled
    assert 100 < float(seen[0].float().mean()) < 160                                # ~N(128, .) pixels
    assert src.wait_ms_total() == 0.0 and src.ring.stats()["copies"] == 2 * 8
    torch.cuda.synchronize()
    stall = src.starved_ms_total()          # GPU-side input stall: HIP events around the stream's wait for the copy
    assert 0.0 <= stall < 1e3 and not src.stall._pairs, stall       # every pair was read after the sync
    src.close()

    # (b) generic loader path
    data = [(torch.randn(4, 3, 16, 16), torch.arange(4) + 10 * i) for i in range(5)]
    pf = DevicePrefetcher(data, dev)
    got = list(pf)
    assert pf.ring is not None and pf.ring.stats()["copies"] == 10                  # every tensor went through a slot
    torch.cuda.synchronize()
    assert 0.0 <= pf.starved_ms_total() < 1e3
    for (a, b), (ga, gb) in zip(data, got):
        assert ga.is_cuda and torch.equal(ga.cpu(), a) and torch.equal(gb.cpu(), b)
    pinned = [(a.pin_memory(), b.pin_memory()) for a, b in data]
    pf2 = DevicePrefetcher(pinned, dev)
    got2 = list(pf2)
    assert pf2.ring is None and all(torch.equal(g[0].cpu(), d[0]) for g, d in zip(got2, data))

    # (c) train from the uint8 ring
    torch.manual_seed(0)
    src = PinnedRingSource(16, 3, 32, 32, device=dev, seed=1, n_slots=4, threads=2)
    model = VGG16(3, 3, True, dropout=0.0).to(dev)
    model.set_input_normalization(*src.normalization)
    opt = FusedSGD(model, lr=0.01, momentum=0.9)
    crit = SoftmaxCrossEntropy()
    losses = []
    for _ in range(40):
        x, y = next(src)
        opt.zero_grad()
        loss = crit(model(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(l == l for l in losses) and min(losses[20:]) < 0.7 * losses[0], losses
    src.close()


@pytest.mark.gpu
def test_hip_graph_captures_native_rccl_reduction():
    """The whole step INCLUDING the bucketed gradient all-reduce in one hipGraph: a reducer on the
    native RCCL transport (world-size-1 communicator on the one GPU — multi-GPU runs are the
    driver's) is forked into the capture by the C++ scheduler's event hand-offs; replays == eager
    steps bit for bit, the collectives are in the graph (RCCL launch counter stands still during
    replays while the gradients keep being reduced), and nothing ran on a fallback."""
    from distributed_training_pytorch_amd.engine.graph import GraphedTrainStep
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD
    from distributed_training_pytorch_amd.parallel.rccl_native import NativeRcclTransport
    from distributed_training_pytorch_amd.parallel.reducer import BucketedReducer
    torch.manual_seed(41)
    xs = [torch.randn(4, 3, 32, 32, device="cuda") for _ in range(6)]
    ys = [torch.randint(0, 3, (4,), device="cuda") for _ in range(6)]

    def make():
        torch.manual_seed(42)
        m = VGG16(3, 3, True, dropout=0.0).to("cuda").train()
        return m, FusedSGD(m, lr=0.002, momentum=0.9, weight_decay=1e-4), SoftmaxCrossEntropy()

    m1, o1, c1 = make()                                 # eager, no reducer
    m2, o2, c2 = make()                                 # graph + native reducer
    t = NativeRcclTransport(rank=0, world=1)
    a = m2.param_arena
    red = BucketedReducer(a.params, flat_grad=a.flat_grad, flat_offsets=a.offsets, transport=t,
                          bucket_cap_mb=64, first_bucket_mb=8, split_large_mb=128)
    red.world = 1
    a.reducer = red
    assert red.graph_capturable
    g = GraphedTrainStep(m2, c2, o2, warmup=2)
    assert g.enabled and g.distributed
    launched = []
    for i, (x, y) in enumerate(zip(xs, ys)):
        o1.zero_grad()
        l1 = c1(m1(x), y)
        l1.backward()
        o1.step()
        l2 = g(x, y)
        torch.cuda.synchronize()
        launched.append(t.stats()["launched"])
        assert torch.equal(l1.detach(), l2), "step %d: eager %r graph %r" % (i, float(l1), float(l2))
    assert g.captures == 1 and g.replays == 4
    assert torch.equal(m1.param_arena.flat_master, m2.param_arena.flat_master)
    nb = len(red.buckets)
    # 2 eager warm-up passes + the capture pass call into RCCL from the host; replays do not
    assert launched[2] == 3 * nb and launched[-1] == launched[2], launched
    a.reducer = None
    t.close()
