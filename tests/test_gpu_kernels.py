"""Numerics of every hand-written HIP kernel against a plain PyTorch fp32 reference of the same op
(SURVEY.md §4.2 'Kernel numerics' tier).  Inputs are bf16-rounded random data; references are
computed in fp32 from the same bf16 values, tolerances are those of bf16 storage + f32 accumulate.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


def _k():
    from distributed_training_pytorch_amd.ops import kernels
    return kernels


def _rand(*shape, scale=1.0, dev="cuda"):
    return (torch.randn(*shape, device=dev) * scale).to(BF16)


def _close(a, b, rtol, atol, name):
    a = a.float()
    b = b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol).float().mean().item()
    assert bad < 1e-4, "%s: %.5f%% elements out of tolerance, max err %.4g (ref max %.4g)" % (
        name, bad * 100, err.max().item(), b.abs().max().item())
    # hard cap: the few elements allowed past the tolerance (bf16 rounding at a boundary) may miss it
    # by a small factor only — a stale LDS row or a dropped K-tile is wrong by O(1) and must fail
    worst = (err / tol).max().item()
    assert worst < 4.0, "%s: an element is %.1fx over its tolerance (max err %.4g, ref max %.4g)" % (
        name, worst, err.max().item(), b.abs().max().item())


def _ref_conv(x_nhwc, w_flat, bias, relu):
    N, H, W, C = x_nhwc.shape
    Co = w_flat.shape[0]
    w = w_flat.float().view(Co, 3, 3, C).permute(0, 3, 1, 2).contiguous()
    y = F.conv2d(x_nhwc.float().permute(0, 3, 1, 2), w, bias, padding=1)
    if relu:
        y = y.relu()
    return y.permute(0, 2, 3, 1).contiguous()


CONV_SHAPES = [  # (N, H, W, Cin, Cout)
    (2, 16, 16, 64, 64),
    (3, 14, 14, 128, 256),
    (2, 28, 28, 64, 128),
    (1, 56, 56, 256, 64),
    (5, 7, 9, 64, 512),      # odd sizes: M tail + W != H
    (2, 14, 14, 512, 512),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_fprop(shape):
    k = _k()
    N, H, W, C, Co = shape
    torch.manual_seed(0)
    x = _rand(N, H, W, C)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * C)) ** 0.5)
    b = torch.randn(Co, device="cuda")
    y = k.conv3x3_fprop(x, w, b, relu=True)
    ref = _ref_conv(x, w, b, True)
    _close(y, ref, 2e-2, 2e-2, "fprop%s" % (shape,))
    y2 = k.conv3x3_fprop(x, w, None, relu=False)
    _close(y2, _ref_conv(x, w, None, False), 2e-2, 2e-2, "fprop-nobias%s" % (shape,))


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 16, 17])
def test_conv3x3_fprop_all_tile_configs(cfg):
    k = _k()
    torch.manual_seed(1)
    N, H, W, C, Co = 2, 20, 12, 128, 256
    x = _rand(N, H, W, C)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * C)) ** 0.5)
    b = torch.randn(Co, device="cuda")
    for nsplit in (1, 3):
        k._overrides["igemm"] = lambda M, Nc, ks: (cfg, nsplit)
        try:
            y = k.conv3x3_fprop(x, w, b, relu=True)
        finally:
            k._overrides.pop("igemm")
        _close(y, _ref_conv(x, w, b, True), 2e-2, 2e-2, "cfg%d split%d" % (cfg, nsplit))


def test_conv3x3_asymmetric_layout():
    """A = shifted delta, asymmetric B: catches row/col swaps and tap mirroring."""
    k = _k()
    N, H, W, C, Co = 1, 8, 8, 64, 64
    x = torch.zeros(N, H, W, C, device="cuda")
    x[0, 3, 4, 5] = 1.0
    w = torch.arange(Co * 9 * C, device="cuda", dtype=torch.float32).view(Co, 9 * C) % 251 / 251.0
    xb, wb = x.to(BF16), w.to(BF16)
    y = k.conv3x3_fprop(xb, wb, None, relu=False)
    _close(y, _ref_conv(xb, wb, None, False), 1e-2, 1e-3, "delta")


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_dgrad(shape):
    k = _k()
    N, H, W, C, Co = shape
    torch.manual_seed(2)
    dy = _rand(N, H, W, Co)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * Co)) ** 0.5)
    xmask = _rand(N, H, W, C)
    wd = k.conv_weight_to_dgrad(w, Co, C)
    # reference for the weight transform
    ref_wd = w.view(Co, 9, C).flip(1).permute(2, 1, 0).reshape(C, 9 * Co)
    assert torch.equal(wd, ref_wd)
    dx = k.conv3x3_dgrad(dy, wd, mask_src=xmask, mask_scale=1.0)
    w4 = w.float().view(Co, 3, 3, C).permute(0, 3, 1, 2).contiguous()
    ref = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w4, padding=1).permute(0, 2, 3, 1)
    ref = ref * (xmask.float() > 0)
    _close(dx, ref, 2e-2, 2e-2, "dgrad%s" % (shape,))


@pytest.mark.parametrize("shape", CONV_SHAPES)
@pytest.mark.parametrize("out_dtype", [torch.float32, BF16])
def test_conv3x3_wgrad(shape, out_dtype):
    k = _k()
    N, H, W, C, Co = shape
    torch.manual_seed(3)
    x = _rand(N, H, W, C)
    dy = _rand(N, H, W, Co, scale=0.1)
    out = torch.full((Co, 9 * C), 7.0, device="cuda", dtype=out_dtype)
    k.conv3x3_wgrad(dy, x, out, beta=0.0)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    wref = torch.zeros(Co, C, 3, 3, device="cuda", requires_grad=True)
    yr = F.conv2d(xr, wref, padding=1)
    (gw,) = torch.autograd.grad(yr, wref, dy.float().permute(0, 3, 1, 2))
    ref = gw.permute(0, 2, 3, 1).reshape(Co, 9 * C)
    scale = ref.abs().max().item()
    _close(out, ref, 2e-2, 2e-2 * scale, "wgrad%s" % (shape,))
    # accumulate (beta = 1)
    k.conv3x3_wgrad(dy, x, out, beta=1.0)
    _close(out, 2 * ref, 3e-2, 3e-2 * scale, "wgrad-acc%s" % (shape,))


@pytest.mark.parametrize("cfg", [2, 3, 4, 8, 9])
def test_conv3x3_wgrad_all_tile_configs(cfg):
    k = _k()
    torch.manual_seed(4)
    N, H, W, C, Co = 3, 10, 14, 256, 256
    x = _rand(N, H, W, C)
    dy = _rand(N, H, W, Co, scale=0.1)
    xr = x.float().permute(0, 3, 1, 2)
    wref = torch.zeros(Co, C, 3, 3, device="cuda", requires_grad=True)
    (gw,) = torch.autograd.grad(F.conv2d(xr, wref, padding=1), wref, dy.float().permute(0, 3, 1, 2))
    ref = gw.permute(0, 2, 3, 1).reshape(Co, 9 * C)
    for nsplit in (1, 4, 3):          # 7 K-tiles: odd counts per split exercise the 8-phase zero tile
        k._overrides["wgrad"] = lambda M, Cc, taps, ks: (cfg, nsplit)
        try:
            out = torch.empty((Co, 9 * C), device="cuda")
            k.conv3x3_wgrad(dy, x, out)
        finally:
            k._overrides.pop("wgrad")
        _close(out, ref, 2e-2, 2e-2 * ref.abs().max().item(), "wgrad cfg%d split%d" % (cfg, nsplit))


LIN_SHAPES = [(64, 25088, 4096), (37, 4096, 4096), (16, 512, 1024), (130, 4096, 3)]


@pytest.mark.parametrize("shape", LIN_SHAPES)
def test_linear_fwd_dgrad_wgrad(shape):
    k = _k()
    M, K, Nout = shape
    torch.manual_seed(5)
    x = _rand(M, K)
    w = _rand(Nout, K, scale=K ** -0.5)
    b = torch.randn(Nout, device="cuda")
    fp32_out = Nout % 8 != 0
    y = k.linear_fwd(x, w, b, relu=not fp32_out, out_fp32=fp32_out)
    ref = F.linear(x.float(), w.float(), b)
    if not fp32_out:
        ref = ref.relu()
    _close(y, ref, 2e-2, 2e-2, "linear_fwd%s" % (shape,))
    # dgrad: dx = dy @ W, via the transposed+padded weight copy
    ld = (Nout + 63) // 64 * 64
    dy = torch.zeros(M, ld, device="cuda", dtype=BF16)
    dy[:, :Nout] = _rand(M, Nout, scale=0.1)
    wt = k.linear_weight_transpose(w)
    assert wt.shape == (K, ld) and torch.equal(wt[:, :Nout], w.t())
    mask = _rand(M, K)
    dx = k.linear_dgrad(dy, wt, mask_src=mask, mask_scale=1.0 / 0.7)
    refdx = (dy[:, :Nout].float() @ w.float()) * (mask.float() > 0) / 0.7
    _close(dx, refdx, 2e-2, 2e-2 * max(refdx.abs().max().item(), 1e-3), "linear_dgrad%s" % (shape,))
    # wgrad
    gw = torch.empty(Nout, K, device="cuda")
    k.linear_wgrad(dy, x, gw, n_valid=Nout)
    refgw = dy[:, :Nout].float().t() @ x.float()
    _close(gw, refgw, 2e-2, 2e-2 * refgw.abs().max().item(), "linear_wgrad%s" % (shape,))
    # bias grad
    gb = torch.empty(Nout, device="cuda")
    k.colsum(dy, gb, n_valid=Nout)
    _close(gb, dy[:, :Nout].float().sum(0), 1e-3, 1e-3, "colsum%s" % (shape,))


@pytest.mark.parametrize("rows,C,nv", [(4, 4096, 4096), (4, 1024, 1000), (784, 512, 512), (1024, 72, 70),
                                       (1025, 512, 512), (20000, 64, 64)])
def test_colsum_single_launch_and_two_pass(rows, C, nv):
    """rows <= 1024 take the single-launch kernel (small-batch bias gradients), larger ones the
    two-pass form: both against an fp32 sum, with beta accumulation and a bf16 destination."""
    k = _k()
    torch.manual_seed(rows + C)
    x = _rand(rows, C, scale=0.5)
    ref = x.float().sum(0)[:nv]
    tol = 1e-3 * max(1.0, ref.abs().max().item())
    out = torch.full((nv,), 7.0, device="cuda")
    k.colsum(x, out, n_valid=nv)
    _close(out, ref, 1e-3, tol, "colsum(%d,%d)" % (rows, C))
    k.colsum(x, out, n_valid=nv, beta=1.0)
    _close(out, 2 * ref, 1e-3, 2 * tol, "colsum beta(%d,%d)" % (rows, C))
    o16 = torch.zeros(nv, device="cuda", dtype=BF16)
    k.colsum(x, o16, n_valid=nv)
    _close(o16, ref, 1e-2, 1e-2 * max(1.0, ref.abs().max().item()), "colsum bf16(%d,%d)" % (rows, C))


def test_linear_dropout_statistics_and_determinism():
    k = _k()
    torch.manual_seed(6)
    M, K, Nout = 256, 512, 4096
    x = _rand(M, K)
    w = _rand(Nout, K, scale=K ** -0.5)
    b = torch.ones(Nout, device="cuda") * 3.0      # keep activations positive
    y0 = k.linear_fwd(x, w, b, relu=True, drop_p=0.0)
    y1 = k.linear_fwd(x, w, b, relu=True, drop_p=0.3, seed=123)
    y2 = k.linear_fwd(x, w, b, relu=True, drop_p=0.3, seed=123)
    y3 = k.linear_fwd(x, w, b, relu=True, drop_p=0.3, seed=124)
    assert torch.equal(y1, y2)
    assert not torch.equal(y1, y3)
    pos = y0.float() > 0
    dropped = ((y1.float() == 0) & pos).float().sum() / pos.float().sum()
    assert abs(dropped.item() - 0.3) < 0.01
    kept = (y1.float() != 0) & pos
    _close(y1[kept], y0[kept].float() / 0.7, 2e-2, 2e-2, "dropout scale")


@pytest.mark.parametrize("shape", [(2, 8, 8, 64), (3, 14, 14, 512), (2, 7, 9, 128), (1, 224, 224, 64)])
def test_maxpool(shape):
    k = _k()
    torch.manual_seed(7)
    N, H, W, C = shape
    x = _rand(N, H, W, C)
    # coarse quantisation forces many exact ties (incl. zero ties): tests the first-max rule
    x = ((x.float() * 2).round() / 2).to(BF16)
    y = k.maxpool2x2_fwd(x)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    assert torch.equal(y.float(), yr.permute(0, 2, 3, 1))
    dy = _rand(*y.shape)
    (gx,) = torch.autograd.grad(yr, xr, dy.float().permute(0, 3, 1, 2))
    dx = k.maxpool2x2_bwd(x, dy, relu_mask=False)
    assert torch.equal(dx.float(), gx.permute(0, 2, 3, 1))
    dxm = k.maxpool2x2_bwd(x, dy, relu_mask=True)
    assert torch.equal(dxm.float(), (gx * (xr > 0)).permute(0, 2, 3, 1))
    # index-code path: the forward pass's 4-bit codes replace the input tensor in the backward pass
    y2, idx = k.maxpool2x2_fwd(x, want_idx=True)
    assert torch.equal(y2, y) and idx.dtype == torch.int32
    for rm, ref in ((False, dx), (True, dxm)):
        dxi = k.maxpool2x2_bwd(None, dy, relu_mask=rm, idx=idx, in_shape=tuple(x.shape))
        assert torch.equal(dxi, ref)


@pytest.mark.parametrize("hw", [(7, 7), (1, 1), (14, 14), (5, 3)])
def test_avgpool_flatten(hw):
    k = _k()
    torch.manual_seed(8)
    N, C = 3, 64
    H, W = hw
    x = _rand(N, H, W, C)
    y = k.avgpool_flatten_fwd(x)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = torch.flatten(F.adaptive_avg_pool2d(xr, (7, 7)), 1)
    _close(y, yr, 1e-2, 1e-2, "avgpool fwd")
    dy = _rand(N, C * 49)
    (gx,) = torch.autograd.grad(yr, xr, dy.float())
    dx = k.avgpool_flatten_bwd(dy, N, H, W, C)
    _close(dx, gx.permute(0, 2, 3, 1), 1e-2, 1e-2, "avgpool bwd")


@pytest.mark.parametrize("nc", [(64, 3), (33, 10), (128, 1000)])
def test_softmax_xent(nc):
    k = _k()
    torch.manual_seed(9)
    N, C = nc
    logits = torch.randn(N, C, device="cuda") * 3
    labels = torch.randint(0, C, (N,), device="cuda")
    out2, dl32, dl16 = k.softmax_xent(logits, labels)
    lr = logits.clone().requires_grad_(True)
    loss = F.cross_entropy(lr, labels)
    loss.backward()
    assert abs(out2[0].item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
    hits = (logits.argmax(1) == labels).sum().item()
    assert out2[1].item() == hits
    _close(dl32, lr.grad, 1e-4, 1e-6, "dlogits")
    _close(dl16[:, :C], lr.grad, 1e-2, 1e-5, "dlogits bf16")
    assert torch.all(dl16[:, C:] == 0)


def test_sgd_and_adamw_flat_match_torch():
    k = _k()
    torch.manual_seed(10)
    n = 1 << 16
    for gdt in (torch.float32, BF16):
        p0 = torch.randn(n, device="cuda")
        ref_p = torch.nn.Parameter(p0.clone())
        opt = torch.optim.SGD([ref_p], lr=0.1, momentum=0.9, weight_decay=1e-4)
        p = p0.clone()
        mom = torch.zeros(n, device="cuda")
        lp = torch.empty(n, device="cuda", dtype=BF16)
        for step in range(3):
            g = torch.randn(n, device="cuda").to(gdt)
            ref_p.grad = g.float().clone()
            opt.step()
            k.sgd_flat(p, g, mom, lp, 0.1, 0.9, 0.0, 1e-4, 1.0, False, step == 0)
        _close(p, ref_p.data, 1e-5, 1e-6, "sgd %s" % gdt)
        assert torch.equal(lp, p.to(BF16))
        ref_p = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([ref_p], lr=1e-3, weight_decay=0.01)
        p = p0.clone()
        m = torch.zeros(n, device="cuda")
        v = torch.zeros(n, device="cuda")
        for step in range(3):
            g = torch.randn(n, device="cuda").to(gdt)
            ref_p.grad = g.float().clone()
            opt.step()
            k.adamw_flat(p, g, m, v, lp, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1.0, step + 1)
        _close(p, ref_p.data, 1e-4, 1e-6, "adamw %s" % gdt)


def test_nchw_to_nhwc_and_cast_and_relu_mask():
    k = _k()
    torch.manual_seed(11)
    x = torch.randn(2, 3, 9, 11, device="cuda")
    y = k.nchw_to_nhwc_bf16(x, 64)
    assert y.shape == (2, 9, 11, 64)
    assert torch.equal(y[..., :3], x.permute(0, 2, 3, 1).to(BF16))
    assert torch.all(y[..., 3:] == 0)
    u8 = torch.randint(0, 256, (2, 3, 9, 11), device="cuda", dtype=torch.uint8)
    mean = torch.tensor([0.485, 0.456, 0.406], device="cuda")
    inv_std = 1.0 / torch.tensor([0.229, 0.224, 0.225], device="cuda")
    z = k.nchw_to_nhwc_bf16(u8, 8, scale=1 / 255.0, mean=mean, inv_std=inv_std)
    ref = ((u8.float() / 255.0 - mean.view(1, 3, 1, 1)) * inv_std.view(1, 3, 1, 1)).permute(0, 2, 3, 1)
    _close(z[..., :3], ref, 1e-2, 1e-2, "u8 normalise")
    a = torch.randn(1000 * 8 + 5, device="cuda")
    b = torch.empty_like(a, dtype=BF16)
    k.cast_f32_bf16(a, b)
    assert torch.equal(b, a.to(BF16))
    g = _rand(4096)
    yv = _rand(4096)
    o = k.relu_mask(g, yv, 2.0)
    _close(o, g.float() * (yv.float() > 0) * 2.0, 1e-2, 1e-3, "relu_mask")


def test_kernels_are_deterministic():
    """Race screen: repeated launches must be bit-identical (cdna_hip_programming.md §5 pitfall)."""
    k = _k()
    torch.manual_seed(12)
    x = _rand(4, 28, 28, 128)
    w = _rand(256, 9 * 128, scale=0.03)
    dy = _rand(4, 28, 28, 256, scale=0.1)
    y0 = k.conv3x3_fprop(x, w, None, relu=True)
    g0 = torch.empty(256, 9 * 128, device="cuda")
    k.conv3x3_wgrad(dy, x, g0)
    for _ in range(5):
        assert torch.equal(k.conv3x3_fprop(x, w, None, relu=True), y0)
        g = torch.empty_like(g0)
        k.conv3x3_wgrad(dy, x, g)
        assert torch.equal(g, g0)


@pytest.mark.parametrize("shape", [(2, 16, 16), (3, 9, 13), (1, 224, 224), (5, 7, 7)])
@pytest.mark.parametrize("in_dtype", [torch.float32, torch.uint8])
def test_conv1_fprop_wgrad(shape, in_dtype):
    """Dedicated first-layer kernels (fused ingest/normalise + conv + bias + ReLU; wgrad + bias grad)."""
    k = _k()
    torch.manual_seed(13)
    N, H, W = shape
    mean = torch.tensor([0.485, 0.456, 0.406], device="cuda")
    inv_std = 1.0 / torch.tensor([0.229, 0.224, 0.225], device="cuda")
    if in_dtype == torch.uint8:
        x = torch.randint(0, 256, (N, 3, H, W), device="cuda", dtype=torch.uint8)
        xn = (x.float() / 255.0 - mean.view(1, 3, 1, 1)) * inv_std.view(1, 3, 1, 1)
        kw = dict(scale=1 / 255.0, mean=mean, inv_std=inv_std)
    else:
        x = torch.randn(N, 3, H, W, device="cuda")
        xn = x
        kw = {}
    xn = xn.to(BF16).float()                      # the kernel rounds the normalised input to bf16
    w = _rand(64, 27, scale=(2.0 / 27) ** 0.5)
    b = torch.randn(64, device="cuda")
    y = k.conv1_fprop(x, w, b, relu=True, **kw)
    w4 = w.float().view(64, 3, 3, 3).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    yr = F.conv2d(xn, w4, b, padding=1)
    _close(y, yr.relu().permute(0, 2, 3, 1), 2e-2, 2e-2, "conv1 fprop")
    dy = _rand(N, H, W, 64, scale=0.1)
    for dt in (torch.float32, BF16):
        gw = torch.full((64, 27), 3.0, device="cuda", dtype=dt)
        gb = torch.full((64,), 3.0, device="cuda", dtype=dt)
        k.conv1_wgrad(x, dy, gw, gb, **kw)
        (gref,) = torch.autograd.grad(yr, w4, dy.float().permute(0, 3, 1, 2), retain_graph=True)
        gref = gref.permute(0, 2, 3, 1).reshape(64, 27)
        bref = dy.float().sum((0, 1, 2))
        _close(gw, gref, 2e-2, 2e-2 * gref.abs().max().item(), "conv1 wgrad")
        _close(gb, bref, 2e-2, 2e-2 * bref.abs().max().item(), "conv1 bgrad")
        k.conv1_wgrad(x, dy, gw, gb, beta_w=1.0, beta_b=1.0, **kw)
        _close(gw, 2 * gref, 3e-2, 3e-2 * gref.abs().max().item(), "conv1 wgrad acc")
        _close(gb, 2 * bref, 3e-2, 3e-2 * bref.abs().max().item(), "conv1 bgrad acc")
    # self-masking: the kernel recomputes sign(conv1(x)+b) and masks dy itself == feeding it the
    # pre-masked dy (mask from the forward kernel's own bf16 output), bit for bit
    dym = (dy.float() * (y.float() > 0)).to(BF16)
    g1, b1 = torch.empty(64, 27, device="cuda"), torch.empty(64, device="cuda")
    g2, b2 = torch.empty(64, 27, device="cuda"), torch.empty(64, device="cuda")
    k.conv1_wgrad(x, dym, g1, b1, **kw)
    k.conv1_wgrad(x, dy, g2, b2, mask_w=w, mask_bias=b, **kw)
    assert torch.equal(g1, g2) and torch.equal(b1, b2)
    assert not torch.equal(g1, gw.float()) or (y.float() > 0).all()


@pytest.mark.parametrize("cfg", [16, 17, 18, 19])
@pytest.mark.parametrize("shape", [(4, 28, 28, 128, 256), (2, 14, 14, 512, 512), (3, 9, 11, 64, 256), (2, 56, 56, 256, 256),
                                   (3, 30, 26, 128, 128)])
def test_conv3x3_8phase_schedule(shape, cfg):
    """cfg 16 (256x256, 2x4 waves) / 17 (512x128, 4x2 waves): 8-phase loop with LDS-DMA in flight
    across barriers — even and odd K-tile counts, M/N tails, split-K; repeated to screen for
    DMA/read races."""
    k = _k()
    N, H, W, C, Co = shape
    torch.manual_seed(21)
    x = _rand(N, H, W, C)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * C)) ** 0.5)
    b = torch.randn(Co, device="cuda")
    ref = _ref_conv(x, w, b, True)
    for nsplit in (1, 2, 3):
        k._overrides["igemm"] = lambda M, Nc, ks: (cfg, nsplit)
        try:
            outs = [k.conv3x3_fprop(x, w, b, relu=True) for _ in range(4)]
        finally:
            k._overrides.pop("igemm")
        _close(outs[0], ref, 2e-2, 2e-2, "8phase%s split%d" % (shape, nsplit))
        for o in outs[1:]:
            assert torch.equal(o, outs[0])


@pytest.mark.parametrize("shape", [(4, 28, 28, 128, 256), (2, 14, 14, 512, 512), (2, 56, 56, 256, 256), (3, 30, 26, 128, 128),
                                   (3, 9, 11, 256, 320), (5, 7, 5, 128, 256), (1, 40, 3, 128, 256), (2, 112, 112, 128, 64),
                                   (300, 2, 2, 128, 256), (1, 16, 16, 384, 256), (2, 33, 16, 128, 256)])
def test_conv3x3_shared_activation_image_schedule(shape):
    """cfg 18 (csrc/igemm.hip PIPE 3): the three dx taps of a filter row read ONE staged activation
    image; the dx borders are per-lane selects of a zeroed LDS region, the dy borders per-row zero
    fill at staging.  Same reduction order as cfg 16 -> bit-identical to it, forward and masked data
    gradient (with the 1-bit masks, column sums and BatchNorm statistics of the shared epilogue), on
    widths from 2 to 112 (borders in every lane position, several per 16-pixel fragment, W = 16 where
    every fragment starts at w == 0), M and N tails, images smaller than a tile, and repeated runs."""
    k = _k()
    N, H, W, C, Co = shape
    torch.manual_seed(61)
    x = _rand(N, H, W, C)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * C)) ** 0.5)
    b = torch.randn(Co, device="cuda")
    ref = _ref_conv(x, w, b, True)
    y16, bits16 = _with_cfg(k, 16, lambda: k.conv3x3_fprop(x, w, b, relu=True, want_bits=True))
    outs = [_with_cfg(k, cfg, lambda: k.conv3x3_fprop(x, w, b, relu=True, want_bits=True)) for cfg in (18, 18, 19, 19)]
    _close(outs[0][0], ref, 2e-2, 2e-2, "shared-image fprop%s" % (shape,))
    for y, bits in outs:
        assert torch.equal(y, y16), "cfg 18/19 != cfg 16 %s: %g" % (shape, (y.float() - y16.float()).abs().max().item())
        assert torch.equal(bits, bits16)
    # masked data gradient of a layer with Co inputs (reduction over Co: needs Co % 128 == 0)
    if Co % 128 == 0:
        dy = _rand(N, H, W, Co, scale=0.1)
        wd = _rand(C, 9 * Co, scale=(2.0 / (9 * Co)) ** 0.5)
        d16 = _with_cfg(k, 16, lambda: k.conv3x3_dgrad(dy, wd, mask_src=x))
        cs16 = torch.empty(C, device="cuda")
        k.bias_grad(d16.view(-1, C), cs16)
        for cfg in (18, 19):
            d18 = _with_cfg(k, cfg, lambda: k.conv3x3_dgrad(dy, wd, mask_src=x))
            assert k.colsum_slot.tensor is d18
            cs18 = torch.empty(C, device="cuda")
            k.bias_grad(d18.view(-1, C), cs18)
            assert torch.equal(d16, d18), "cfg %d" % cfg
            assert torch.allclose(cs16, cs18, rtol=1e-5, atol=1e-5 * cs16.abs().max().item())     # per-tile partial rows: other tile heights, other f32 order
        w4 = wd.float().view(C, 3, 3, Co)        # wd[ci][t][co] = w[co][8-t][ci]: reference through the plain conv
        refd = F.conv2d(dy.float().permute(0, 3, 1, 2), w4.permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1) * (x.float() > 0)
        _close(d18, refd, 2e-2, 2e-2 * refd.abs().max().item(), "shared-image dgrad%s" % (shape,))


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 9, 11, 16, 17, 18, 19])
def test_fused_bias_grad_from_dgrad_epilogue(cfg):
    """The masked data-gradient kernel leaves per-tile column sums of its output behind; the
    consumer's bias gradient built from them equals the column sum of the stored bf16 tensor."""
    k = _k()
    torch.manual_seed(11)
    N, H, W, C, Co = 3, 18, 14, 192, 128          # M = 756: row tail in every tile size
    if cfg == 18:
        k.igemm_shared_a = True                    # (Co = 128 reduction channels: the shared-image schedule applies)
    dy = _rand(N, H, W, Co)
    wd = _rand(C, 9 * Co, scale=(2.0 / (9 * Co)) ** 0.5)
    xmask = _rand(N, H, W, C)
    k._overrides["igemm"] = lambda M, Nc, ks: (cfg, 1)
    try:
        dx = k.conv3x3_dgrad(dy, wd, mask_src=xmask)
    finally:
        k._overrides.pop("igemm")
    assert k.colsum_slot.tensor is dx
    parts, nparts = k.colsum_slot.parts, k.colsum_slot.nparts
    for dt in (torch.float32, BF16):
        out = torch.full((C,), 3.0, device="cuda", dtype=dt)
        k.bias_grad(dx.view(-1, C), out, beta=1.0)
        ref = 3.0 + dx.float().view(-1, C).sum(0)
        _close(out, ref, 1e-2 if dt == BF16 else 1e-5, 2e-2 if dt == BF16 else 1e-4, "fused colsum cfg%d" % cfg)
        k.colsum_slot.put(dx, parts, nparts, C)     # re-arm for the second dtype
    k.colsum_slot.take(dx, C)
    # a different tensor must NOT pick the slot up
    k.colsum_slot.put(dx, parts, nparts, C)
    other = dx.clone()
    assert k.colsum_slot.take(other.view(-1, C), C) is None and k.colsum_slot.tensor is None


@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (3, 28, 20, 128), (2, 8, 8, 512), (4, 112, 112, 64)])
def test_fused_bias_grad_from_maxpool_backward(shape):
    k = _k()
    torch.manual_seed(12)
    N, H, W, C = shape
    x = _rand(N, H, W, C)
    dy = _rand(N, H // 2, W // 2, C)
    dx = k.maxpool2x2_bwd(x, dy, relu_mask=True)
    assert k.colsum_slot.tensor is dx
    out = torch.empty(C, device="cuda")
    k.bias_grad(dx.view(-1, C), out)
    assert k.colsum_slot.tensor is None
    _close(out, dx.float().view(-1, C).sum(0), 1e-5, 1e-3, "pool colsum %s" % (shape,))
    k.fuse_colsum = False
    try:
        dx2 = k.maxpool2x2_bwd(x, dy, relu_mask=True)
        out2 = torch.empty(C, device="cuda")
        k.bias_grad(dx2.view(-1, C), out2)
    finally:
        k.fuse_colsum = True
    assert torch.equal(dx, dx2)
    _close(out, out2, 1e-5, 1e-3, "pool colsum vs stand-alone %s" % (shape,))


@pytest.mark.parametrize("shape", [(64, 256, 1024), (200, 4096, 25088 // 4), (256, 64, 128), (8, 4096, 4096)])
def test_fp8_quant_and_linear_forward(shape):
    """fp8 (OCP e4m3) classifier GEMM: device-side per-tensor quantisation matches torch's
    float8_e4m3fn cast, and the MFMA GEMM matches an fp32 matmul of the dequantised operands."""
    k = _k()
    M, N, K = shape
    torch.manual_seed(13)
    x = _rand(M, K).relu()                         # classifier inputs are ReLU outputs
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF16)
    b = torch.randn(N, device="cuda")
    xq, xs = k.quant_fp8(x)
    wq, ws = k.quant_fp8(w)
    assert abs(xs.item() - x.float().abs().max().item() / 448.0) < 1e-6 * xs.item() + 1e-12
    ref_xq = (x.float() / xs).to(torch.float8_e4m3fn)
    same = (xq.view(torch.float8_e4m3fn).float() == ref_xq.float()).float().mean().item()
    assert same > 0.999, "quantisation differs from torch's e4m3fn cast on %.4f%% of elements" % (100 * (1 - same))
    xd = xq.view(torch.float8_e4m3fn).float() * xs
    wd = wq.view(torch.float8_e4m3fn).float() * ws
    ref = torch.relu(xd @ wd.t() + b)
    y = k.linear_fwd_fp8(xq, xs, wq, ws, b, relu=True)
    _close(y, ref, 1e-2, 1e-2 * ref.abs().max().item(), "fp8 linear %s" % (shape,))
    y32 = k.linear_fwd_fp8(xq, xs, wq, ws, b, relu=True, out_fp32=True)
    _close(y32, ref, 2e-3, 2e-3 * ref.abs().max().item(), "fp8 linear f32 out %s" % (shape,))
    # and the quantisation error against the bf16 GEMM is the expected few percent
    full = torch.relu(x.float() @ w.float().t() + b)
    rel = (y32 - full).norm().item() / (full.norm().item() + 1e-12)
    assert rel < 0.06, rel


@pytest.mark.parametrize("shape", [(256, 4096, 4096), (64, 512, 1024), (200, 384, 2048)])
def test_linear_dgrad_fp8_e5m2(shape):
    """fp8 data gradient of the classifier (BASELINE.json config 5): dY quantised to e5m2 on the
    device == torch's float8_e5m2 cast of dY/scale; the e5m2 x e4m3 GEMM (v_mfma_f32_16x16x32_fp8_bf8)
    == an fp32 matmul of the dequantised operands, with the ReLU mask of the layer below in the
    epilogue; and against the unquantised gradient the error is the expected few percent."""
    k = _k()
    M, Nout, Kin = shape
    torch.manual_seed(23)
    dy = _rand(M, Nout, scale=0.02)
    w = _rand(Nout, Kin, scale=0.02)
    xmask = _rand(M, Kin)
    dyq, ds = k.quant_fp8(dy, e5m2=True)
    assert abs(ds.item() - dy.float().abs().max().item() / 57344.0) < 1e-6 * ds.item() + 1e-12
    ref_q = (dy.float() / ds).to(torch.float8_e5m2)
    same = (dyq.view(torch.float8_e5m2).float() == ref_q.float()).float().mean().item()
    assert same > 0.999, "e5m2 quantisation differs from torch's cast on %.4f%% of elements" % (100 * (1 - same))
    wt = k.linear_weight_transpose(w)                      # [Kin, Nout]
    wtq, ws = k.quant_fp8(wt)
    dyd = dyq.view(torch.float8_e5m2).float() * ds
    wtd = wtq.view(torch.float8_e4m3fn).float() * ws
    ref = (dyd @ wtd.t()) * (xmask.float() > 0) * 1.25
    dx = k.linear_dgrad_fp8(dyq, ds, wtq, ws, mask_src=xmask, mask_scale=1.25)
    _close(dx, ref, 1e-2, 1e-2 * ref.abs().max().item(), "fp8 dgrad %s" % (shape,))
    dx0 = k.linear_dgrad_fp8(dyq, ds, wtq, ws)
    full = dy.float() @ w.float()
    rel = (dx0.float() - full).norm().item() / (full.norm().item() + 1e-12)
    assert rel < 0.12, rel


@pytest.mark.parametrize("cfg", [8, 9])
@pytest.mark.parametrize("shape", [(2, 28, 28, 512, 512), (3, 7, 9, 256, 512), (1, 56, 56, 256, 256),
                                   (70, 4, 4, 256, 256), (2, 224, 9, 128, 384)])
def test_wgrad_8phase_schedule(shape, cfg):
    """The hand-scheduled weight-gradient kernels — cfg 8 / 9 (wgrad2.hip: asm transposed reads,
    descriptor staging with hardware zero fill, ballot border masks) — on conv shapes incl. K tails, images smaller than one K-step (4x4: several
    wraps of the row tracker per step), W < 64 < H, bf16 output with beta accumulation, and the
    linear (taps == 1) form with M / N tails."""
    k = _k()
    if shape[3] % 256 and cfg != 9:
        pytest.skip("256-wide tiles need C % 256 == 0")
    N, H, W, C, Co = shape
    torch.manual_seed(14)
    x = _rand(N, H, W, C)
    dy = _rand(N, H, W, Co, scale=0.1)
    xr = x.float().permute(0, 3, 1, 2)
    wref = torch.zeros(Co, C, 3, 3, device="cuda", requires_grad=True)
    (gw,) = torch.autograd.grad(F.conv2d(xr, wref, padding=1), wref, dy.float().permute(0, 3, 1, 2))
    ref = gw.permute(0, 2, 3, 1).reshape(Co, 9 * C)
    for nsplit, dt, beta in ((1, torch.float32, 0.0), (5, torch.float32, 1.0), (2, BF16, 1.0)):
        k._overrides["wgrad"] = lambda M, Cc, taps, ks: (cfg, nsplit)
        try:
            out = torch.full((Co, 9 * C), 0.5, device="cuda", dtype=dt)
            k.conv3x3_wgrad(dy, x, out, beta=beta)
        finally:
            k._overrides.pop("wgrad")
        _close(out, ref + 0.5 * beta, 2e-2, 2e-2 * ref.abs().max().item(), "wgrad8 %s split%d" % (shape, nsplit))
    # linear form: dW[Nout, K] = dy^T x with tails in every dimension
    M, Kin, Nout = 200, 520, 300
    xl = _rand(M, Kin)
    dl = _rand(M, 320, scale=0.1)
    k._overrides["wgrad"] = lambda Mm, Cc, taps, ks: (cfg, 2)
    try:
        gl = torch.empty((Nout, Kin), device="cuda")
        k.linear_wgrad(dl, xl, gl, n_valid=Nout)
    finally:
        k._overrides.pop("wgrad")
    refl = dl.float()[:, :Nout].t() @ xl.float()
    _close(gl, refl, 2e-2, 2e-2 * refl.abs().max().item(), "wgrad8 linear")


@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (3, 28, 36, 128), (1, 64, 70, 64), (2, 224, 224, 64)])
def test_fused_conv_relu_pool_matches_separate_kernels(shape):
    """conv_c64 POOL epilogue == conv3x3_fprop followed by maxpool2x2_fwd, bit for bit (pooled
    activations AND the 4-bit argmax/positive codes), incl. strips with tails (W = 36, 70)."""
    k = _k()
    N, H, W, Co = shape
    torch.manual_seed(15)
    x = _rand(N, H, W, 64)
    w = _rand(Co, 9 * 64, scale=(2.0 / (9 * 64)) ** 0.5)
    b = torch.randn(Co, device="cuda") * 0.1
    y_full = k.conv3x3_fprop(x, w, b, relu=True)
    y_ref, codes_ref = k.maxpool2x2_fwd(y_full, want_idx=True)
    y, codes = k.conv3x3_relu_pool_fprop(x, w, b)
    assert torch.equal(y, y_ref)
    assert torch.equal(codes, codes_ref)
    dyp = _rand(*y.shape)
    dx_ref = k.maxpool2x2_bwd(None, dyp, relu_mask=True, idx=codes_ref, in_shape=tuple(y_full.shape))
    dx = k.maxpool2x2_bwd(None, dyp, relu_mask=True, idx=codes, in_shape=tuple(y_full.shape))
    assert torch.equal(dx, dx_ref)


@pytest.mark.parametrize("shape", [(8, 112, 112, 128, 128), (6, 56, 56, 256, 256), (9, 28, 28, 512, 512), (40, 14, 14, 512, 512),
                                   (3, 30, 26, 128, 256), (5, 16, 14, 128, 72), (2, 18, 44, 256, 320), (70, 4, 14, 128, 128)])
def test_fused_conv_relu_pool_in_the_implicit_gemm(shape):
    """cfg 20 / 21 (csrc/igemm.hip PIPE 3 + POOL): conv3x3 + bias + ReLU + 2x2 max-pool as ONE kernel for
    the block ends of VGG16's blocks 2-5 (and odd shapes: M / N tails, H != W, widths from 14 to 112,
    images smaller than a tile).  Output rows are enumerated (n, h/2, w, h&1) inside the kernel; pooled
    activations AND the 4-bit (first argmax, max > 0) codes must equal conv3x3_fprop + maxpool2x2_fwd
    bit for bit, so the existing pool backward applies unchanged."""
    k = _k()
    N, H, W, C, Co = shape
    torch.manual_seed(71)
    x = _rand(N, H, W, C)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * C)) ** 0.5)
    b = torch.randn(Co, device="cuda") * 0.1
    assert k.conv_pool_fusable(C, Co, H, W)
    y_full = _with_cfg(k, 3, lambda: k.conv3x3_fprop(x, w, b, relu=True))
    y_ref, codes_ref = k.maxpool2x2_fwd(y_full, want_idx=True)
    for _ in range(2):
        y, codes = k.conv3x3_relu_pool_fprop(x, w, b)
        assert tuple(y.shape) == (N, H // 2, W // 2, Co)
        assert torch.equal(y, y_ref), "pooled output: max diff %g" % (y.float() - y_ref.float()).abs().max().item()
        assert torch.equal(codes, codes_ref)
    dyp = _rand(*y.shape)
    dx_ref = k.maxpool2x2_bwd(None, dyp, relu_mask=True, idx=codes_ref, in_shape=tuple(y_full.shape))
    dx = k.maxpool2x2_bwd(None, dyp, relu_mask=True, idx=codes, in_shape=tuple(y_full.shape))
    assert torch.equal(dx, dx_ref)


@pytest.mark.parametrize("shape", [(2, 20, 12, 128, 256), (3, 9, 11, 64, 128), (2, 14, 14, 256, 192), (1, 30, 34, 64, 64)])
def test_relu_bitmask_roundtrip(shape):
    """conv3x3_fprop(want_bits) packs (y > 0) one bit per element (igemm and conv_c64 epilogues); the
    data gradient masked from those bits is bit-identical to the one masked from y itself."""
    k = _k()
    N, H, W, C, Co = shape
    torch.manual_seed(31)
    x = _rand(N, H, W, C)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * C)) ** 0.5)
    b = torch.randn(Co, device="cuda")
    if C != 64:                       # tiny test grids would take split-K (no bf16 epilogue, no bits)
        k._overrides["igemm"] = lambda M, Nc, ks: (3, 1)
    try:
        y, bits = k.conv3x3_fprop(x, w, b, relu=True, want_bits=True)
        y_plain = k.conv3x3_fprop(x, w, b, relu=True)
    finally:
        k._overrides.pop("igemm", None)
    assert bits is not None and bits.numel() * 8 == y.numel()
    assert torch.equal(y, y_plain)
    pos = (y.float() > 0).view(-1, 8).to(torch.int32)
    ref_bits = (pos * (2 ** torch.arange(8, device="cuda", dtype=torch.int32))).sum(1).to(torch.uint8)
    assert torch.equal(bits, ref_bits)
    # consumer: a conv with Co inputs -> its dgrad output has Co channels and is masked by y
    C2 = 128
    dy = _rand(N, H, W, C2)
    wd = _rand(Co, 9 * C2, scale=(2.0 / (9 * C2)) ** 0.5)
    for cfg in (None, 3, 16, 17, 18, 19):
        if cfg is not None:
            k._overrides["igemm"] = lambda M, Nc, ks, cfg=cfg: (cfg, 1)
        try:
            a = k.conv3x3_dgrad(dy, wd, mask_src=y, mask_scale=1.0)
            cs_a = torch.empty(Co, device="cuda")
            k.bias_grad(a.view(-1, Co), cs_a)
            bm = k.conv3x3_dgrad(dy, wd, mask_src=y, mask_scale=1.0, mask_bits=bits)
            cs_b = torch.empty(Co, device="cuda")
            k.bias_grad(bm.view(-1, Co), cs_b)
        finally:
            k._overrides.pop("igemm", None)
        assert torch.equal(a, bm), "cfg %s" % cfg
        assert torch.equal(cs_a, cs_b)
    with pytest.raises(ValueError):
        k.conv3x3_dgrad(dy, wd, mask_src=y, mask_bits=bits[:-1])


@pytest.mark.parametrize("shape", [(2, 30, 34, 64, 128), (1, 112, 112, 64, 128), (3, 17, 9, 64, 128)])
@pytest.mark.parametrize("masked", [False, True])
def test_conv3x3_dgrad_two_pass_c128(shape, masked):
    """128 reduction channels -> 64 outputs (conv2_1's data gradient): two conv_c64 passes over the
    channel halves, the second accumulating in place, against the f32 reference and against the
    generic implicit GEMM (one extra bf16 rounding of the first half's partial sums)."""
    k = _k()
    N, H, W, C, Co = shape
    torch.manual_seed(41)
    dy = _rand(N, H, W, Co)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * Co)) ** 0.5)
    xmask = _rand(N, H, W, C) if masked else None
    wd = k.conv_weight_to_dgrad(w, Co, C)
    k.two_pass_c128 = True            # opt-in path (DTP_DGRAD_TWO_PASS=1)
    try:
        dx = k.conv3x3_dgrad(dy, wd, mask_src=xmask)
        dx2 = k.conv3x3_dgrad(dy, wd, mask_src=xmask, wd_halves=k.split_dgrad_weight(wd, C, Co))
    finally:
        k.two_pass_c128 = False
    assert torch.equal(dx, dx2)
    w4 = w.float().view(Co, 3, 3, C).permute(0, 3, 1, 2).contiguous()
    ref = F.conv_transpose2d(dy.float().permute(0, 3, 1, 2), w4, padding=1).permute(0, 2, 3, 1)
    if masked:
        ref = ref * (xmask.float() > 0)
    _close(dx, ref, 2e-2, 2e-2, "two-pass dgrad%s" % (shape,))
    one = k.conv3x3_dgrad(dy, wd, mask_src=xmask)
    _close(dx, one.float(), 2e-2, 1e-2, "two-pass vs igemm%s" % (shape,))


def _with_cfg(k, cfg, fn):
    k._overrides["igemm"] = lambda M, Nc, ks: (cfg, 1)
    try:
        return fn()
    finally:
        k._overrides.pop("igemm", None)


def test_lds_dma_kernels_are_bitwise_repeatable_under_memory_contention():
    """Regression: conv_c64<64, POOL> had no s_waitcnt vmcnt(0) in front of its row-loop barrier
    (hipcc's __syncthreads() emits it in some instantiations only), so an input row staged by
    LDS-DMA could be read before it landed whenever HBM was busy with another stream's kernels —
    one wrong pooled row in ~5 % of tiny training runs.  All DMA-publishing barriers now go through
    sync_lds_dma(); here the DMA kernels run next to a bandwidth hog on a second stream and every
    repetition must match the first bit for bit."""
    k = _k()
    torch.manual_seed(51)
    x = _rand(4, 32, 32, 64)
    w64 = _rand(64, 576, scale=0.06)
    w128 = _rand(128, 576, scale=0.06)
    b64, b128 = torch.randn(64, device="cuda"), torch.randn(128, device="cuda")
    x2 = _rand(4, 16, 16, 128)
    w2 = _rand(256, 9 * 128, scale=0.04)
    dy = _rand(4, 32, 32, 64)
    gw = torch.empty(64, 576, device="cuda")
    _dy2 = _rand(4, 16, 16, 256)
    _gw2 = torch.empty(256, 9 * 128, device="cuda")
    hog_src = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
    hog_dst = torch.empty_like(hog_src)
    side = torch.cuda.Stream()
    cases = [
        ("conv+pool 64", lambda: k.conv3x3_relu_pool_fprop(x, w64, b64)[0]),
        ("conv+pool 128", lambda: k.conv3x3_relu_pool_fprop(x, w128, b128)[0]),
        ("conv_c64", lambda: k.conv3x3_fprop(x, w64, b64, True)),
        ("igemm", lambda: k.conv3x3_fprop(x2, w2, None, True)),
        ("igemm 8-phase 256x256", lambda: _with_cfg(k, 16, lambda: k.conv3x3_fprop(x2, w2, None, True))),
        ("igemm 8-phase 512x128", lambda: _with_cfg(k, 17, lambda: k.conv3x3_fprop(x2, w2, None, True))),
        ("wgrad_tt", lambda: k.conv3x3_wgrad(_dy2, x2, _gw2).clone()),
        ("wgrad_c64", lambda: k.conv3x3_wgrad(dy, x, gw).clone()),
    ]
    for name, fn in cases:
        ref = fn().clone()
        torch.cuda.synchronize()
        for i in range(150):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    hog_dst.copy_(hog_src)
            out = fn()
            assert torch.equal(out, ref), "%s: repetition %d differs" % (name, i)
        torch.cuda.synchronize()


@pytest.mark.parametrize("shape", [(8, 4096, 25088), (256, 4096, 4096), (24, 512, 1024), (4, 4096, 25088), (13, 512, 1024)])
def test_linear_dgrad_on_native_weight_layout(shape):
    """dX = dY @ W computed by the transposing-read GEMM (wgrad_tt) over W [N_out, K_in] as stored —
    the path Linear takes for unmasked wide layers instead of building W^T every step."""
    k = _k()
    M, Nout, Kin = shape
    torch.manual_seed(61)
    dy = _rand(M, Nout, scale=0.1)
    w = _rand(Nout, Kin, scale=Nout ** -0.5)
    dyT = k.linear_weight_transpose(dy, pad_to=8)
    Mp = (M + 7) // 8 * 8
    assert tuple(dyT.shape) == (Nout, Mp) and torch.equal(dyT[:, :M], dy.t()) and not dyT[:, M:].any()
    dx = torch.empty(M, Kin, dtype=BF16, device="cuda")
    k.linear_wgrad(dyT, w, dx, n_valid=M)
    ref = dy.float() @ w.float()
    _close(dx, ref, 2e-2, 2e-2, "tt dgrad %s" % (shape,))
    via_wt = k.linear_dgrad(dy, k.linear_weight_transpose(w))
    _close(dx, via_wt.float(), 2e-2, 1e-2, "tt vs nt dgrad %s" % (shape,))


@pytest.mark.parametrize("shape", [(4, 14, 14, 64), (3, 9, 7, 128), (2, 28, 28, 256), (5, 7, 7, 512), (64, 56, 56, 64)])
@pytest.mark.parametrize("relu", [True, False])
def test_batchnorm_relu_fwd_bwd_against_torch(shape, relu):
    """csrc/bn.hip against F.batch_norm (+ relu) in fp32 on the same bf16-rounded values: output,
    running statistics (momentum, unbiased variance), dY, d(gamma), d(beta); f32 and bf16 gradient
    outputs with accumulation; eval mode from the running statistics."""
    k = _k()
    N, H, W, C = shape
    torch.manual_seed(21)
    y = (_rand(N, H, W, C).float() * (1.0 + torch.arange(C, device="cuda") % 5) + 0.3 * (torch.arange(C, device="cuda") % 7 - 3)).to(BF16)
    dz = _rand(N, H, W, C, scale=0.1)
    gamma = (1.0 + 0.2 * torch.randn(C, device="cuda")).contiguous()
    beta = (0.3 * torch.randn(C, device="cuda")).contiguous()
    rm, rv = 0.1 * torch.randn(C, device="cuda"), 0.5 + torch.rand(C, device="cuda")
    rm_k, rv_k = rm.clone(), rv.clone()
    z, stats = k.bn_fwd_train(y, gamma, beta, rm_k, rv_k, eps=1e-5, momentum=0.1, relu=relu)
    # fp32 reference on the same inputs
    yr = y.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    g_r, b_r = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    zr = F.batch_norm(yr, rm_r, rv_r, g_r, b_r, True, 0.1, 1e-5)
    if relu:
        zr = zr.relu()
    _close(z, zr.permute(0, 2, 3, 1), 1e-2, 1e-2, "bn fwd %s" % (shape,))
    assert torch.allclose(rm_k, rm_r, atol=1e-4, rtol=1e-4) and torch.allclose(rv_k, rv_r, atol=1e-4, rtol=1e-3)
    zr.backward(dz.float().permute(0, 3, 1, 2))
    # the kernel's ReLU mask comes from ITS forward value; at |z| ~ 0 the fp32 reference may differ by
    # one element's sign: compare against the reference masked with the kernel's own output
    for dt in (torch.float32, BF16):
        dg = torch.full((C,), 0.25, device="cuda", dtype=dt)
        db = torch.full((C,), -0.5, device="cuda", dtype=dt)
        dy = k.bn_bwd(y, dz, gamma, stats, dg, db, beta_g=1.0, beta_b=1.0, relu=relu)
        M = N * H * W
        tol_g = 3e-2 * (g_r.grad.abs().max().item() + 1.0)
        assert (dg.float() - (g_r.grad + 0.25)).abs().max().item() < tol_g, "dgamma %s %s" % (shape, dt)
        assert (db.float() - (b_r.grad - 0.5)).abs().max().item() < 3e-2 * (b_r.grad.abs().max().item() + 1.0), "dbeta"
        _close(dy, yr.grad.permute(0, 2, 3, 1), 3e-2, 3e-2 * yr.grad.abs().max().item() + 1e-5, "bn dy %s" % (shape,))
    # eval mode: running statistics
    ze = k.bn_fwd_eval(y, gamma, beta, rm_k, rv_k, 1e-5, relu)
    zer = F.batch_norm(y.float().permute(0, 3, 1, 2), rm_k, rv_k, gamma, beta, False, 0.1, 1e-5)
    if relu:
        zer = zer.relu()
    _close(ze, zer.permute(0, 2, 3, 1), 1e-2, 1e-2, "bn eval %s" % (shape,))
    sc, sh = k.bn_eval_coeffs(gamma, beta, rm_k, rv_k, 1e-5)
    assert torch.allclose(sc, gamma / torch.sqrt(rv_k + 1e-5), rtol=1e-5, atol=1e-6)
    assert torch.allclose(sh, beta - rm_k * sc, rtol=1e-5, atol=1e-5)
    # determinism (two-level reductions, no atomics)
    z2, stats2 = k.bn_fwd_train(y, gamma, beta, rm.clone(), rv.clone(), relu=relu)
    assert torch.equal(z, z2) and torch.equal(stats, stats2)


@pytest.mark.parametrize("shape", [(4, 14, 14, 64), (3, 8, 6, 128), (2, 28, 28, 256), (5, 14, 14, 512), (16, 112, 112, 64)])
def test_batchnorm_relu_pool_fused_matches_the_three_kernels(shape):
    """Block-end fusion (bn_fwd_train_pool / bn_bwd_pool) against the un-fused chain of this
    library's own kernels, which the test above pins to F.batch_norm: pooled activation, pool codes,
    statistics and running statistics bit for bit; dY / d(gamma) / d(beta) to rounding (the fused
    backward evaluates the same expression in a different association)."""
    k = _k()
    N, H, W, C = shape
    torch.manual_seed(33)
    y = (_rand(N, H, W, C).float() * (1.0 + torch.arange(C, device="cuda") % 5) + 0.3 * (torch.arange(C, device="cuda") % 7 - 3)).to(BF16)
    y[0, :2, :2, :8] = 0.25                      # an exact four-way tie: the FIRST window element must win
    dzp = _rand(N, H // 2, W // 2, C, scale=0.1)
    gamma = (1.0 + 0.2 * torch.randn(C, device="cuda")).contiguous()
    beta = (0.3 * torch.randn(C, device="cuda")).contiguous()
    rm, rv = 0.1 * torch.randn(C, device="cuda"), 0.5 + torch.rand(C, device="cuda")
    rm_a, rv_a, rm_b, rv_b = rm.clone(), rv.clone(), rm.clone(), rv.clone()
    z, stats = k.bn_fwd_train(y, gamma, beta, rm_a, rv_a, relu=True)
    zp_ref, codes_ref = k.maxpool2x2_fwd(z, want_idx=True)
    zp, codes, stats_f = k.bn_fwd_train_pool(y, gamma, beta, rm_b, rv_b)
    assert torch.equal(stats, stats_f) and torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
    assert torch.equal(zp, zp_ref), "pooled activation %s" % (shape,)
    assert torch.equal(codes, codes_ref), "pool codes %s" % (shape,)
    for dt in (torch.float32, BF16):
        dg_r, db_r = torch.full((C,), 0.25, device="cuda", dtype=dt), torch.full((C,), -0.5, device="cuda", dtype=dt)
        dg_f, db_f = dg_r.clone(), db_r.clone()
        k.fuse_colsum = False
        try:
            dz = k.maxpool2x2_bwd(None, dzp, relu_mask=True, idx=codes_ref, in_shape=(N, H, W, C))
        finally:
            k.fuse_colsum = True
        dy_ref = k.bn_bwd(y, dz, gamma, stats, dg_r, db_r, beta_g=1.0, beta_b=1.0, relu=True)
        dy = k.bn_bwd_pool(y, dzp, codes, gamma, stats, dg_f, db_f, beta_g=1.0, beta_b=1.0)
        tol = 2e-2 if dt == BF16 else 2e-3
        assert (dg_f.float() - dg_r.float()).abs().max().item() <= tol * (dg_r.float().abs().max().item() + 1.0)
        assert (db_f.float() - db_r.float()).abs().max().item() <= tol * (db_r.float().abs().max().item() + 1.0)
        _close(dy, dy_ref.float(), 2e-2, 2e-2 * dy_ref.float().abs().max().item() + 1e-6, "fused bn+pool dy %s" % (shape,))
    # determinism
    zp2, codes2, _ = k.bn_fwd_train_pool(y, gamma, beta, rm.clone(), rv.clone())
    assert torch.equal(zp, zp2) and torch.equal(codes, codes2)


VGG_LAYERS = [  # every 3x3 layer of VGG16 after conv1_1: (Cin, Cout, H = W)
    (64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 256, 56),
    (256, 512, 28), (512, 512, 28), (512, 512, 28), (512, 512, 14), (512, 512, 14), (512, 512, 14),
]


@pytest.mark.parametrize("layer", sorted(set(VGG_LAYERS)))
@pytest.mark.parametrize("N", [2, 3])
def test_vgg_conv_shapes_through_the_production_pickers(layer, N):
    """The REAL VGG16 layer shapes at 224x224 (not miniatures) through whatever tile configuration,
    split factor and kernel family the production pickers choose for them — forward, masked data
    gradient (1-bit masks) and weight gradient — against fp32 PyTorch on the same bf16 values."""
    k = _k()
    C, Co, HW = layer
    torch.manual_seed(C + Co + HW + N)
    x = _rand(N, HW, HW, C).relu_()                     # a ReLU output: about half the mask bits are 0
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * C)) ** 0.5)
    b = torch.randn(Co, device="cuda")
    y, bits = k.conv3x3_fprop(x, w, b, relu=True, want_bits=True)
    ref = _ref_conv(x, w, b, True)
    _close(y, ref, 2e-2, 2e-2 * ref.abs().max().item(), "fprop %s" % (layer,))
    if bits is not None:
        got = (bits.view(-1, 1) >> torch.arange(8, device="cuda", dtype=torch.uint8)) & 1
        assert torch.equal(got.view(-1).bool(), (y.view(-1) > 0))
    dy = _rand(N, HW, HW, Co, scale=0.1)
    wd = k.conv_weight_to_dgrad(w, Co, C)
    xbits = ((x.view(-1, 8) > 0).to(torch.int32) * (2 ** torch.arange(8, device="cuda", dtype=torch.int32))).sum(1).to(torch.uint8)
    dx = k.conv3x3_dgrad(dy, wd, mask_src=x, mask_bits=xbits)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.float().view(Co, 3, 3, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    out = F.conv2d(xr, wr, padding=1)
    gx, gw = torch.autograd.grad(out, (xr, wr), dy.float().permute(0, 3, 1, 2))
    refdx = (gx * (xr > 0)).permute(0, 2, 3, 1)
    _close(dx, refdx, 2e-2, 2e-2 * refdx.abs().max().item(), "dgrad %s" % (layer,))
    refgw = gw.permute(0, 2, 3, 1).reshape(Co, 9 * C)
    for dt in (torch.float32, BF16):
        g = torch.empty((Co, 9 * C), device="cuda", dtype=dt)
        k.conv3x3_wgrad(dy, x, g)
        _close(g, refgw, 2e-2, 2e-2 * refgw.abs().max().item(), "wgrad %s %s" % (layer, dt))


@pytest.mark.parametrize("shape", [(2, 25088, 4096), (2, 4096, 4096), (2, 4096, 3), (1024, 25088, 4096)])
def test_vgg_classifier_shapes_through_the_production_pickers(shape):
    k = _k()
    M, K, Nout = shape
    torch.manual_seed(M + Nout)
    x = _rand(M, K)
    w = _rand(Nout, K, scale=K ** -0.5)
    b = torch.randn(Nout, device="cuda")
    y = k.linear_fwd(x, w, b, relu=False, out_fp32=Nout % 8 != 0)
    ref = x.float() @ w.float().t() + b
    _close(y, ref, 2e-2, 2e-2 * ref.abs().max().item(), "fc fwd %s" % (shape,))
    ld = (Nout + 63) // 64 * 64
    dy = torch.zeros(M, ld, device="cuda", dtype=BF16)
    dy[:, :Nout] = _rand(M, Nout, scale=0.1)
    gw = torch.empty(Nout, K, device="cuda")
    k.linear_wgrad(dy, x, gw, n_valid=Nout)
    refgw = dy.float()[:, :Nout].t() @ x.float()
    _close(gw, refgw, 2e-2, 2e-2 * refgw.abs().max().item() + 1e-6, "fc wgrad %s" % (shape,))
    wt = k.linear_weight_transpose(w)
    dx = k.linear_dgrad(dy, wt)
    refdx = dy.float()[:, :Nout] @ w.float()
    _close(dx, refdx, 2e-2, 2e-2 * refdx.abs().max().item(), "fc dgrad %s" % (shape,))


def test_stream_delay_holds_the_stream_for_about_the_requested_time():
    """stream_delay (the head start wgrad_c64's persistent grid gets before the data gradient is
    dispatched): one idle wave, bounded, and it ends by itself."""
    k = _k()
    k.stream_delay(50.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    k.stream_delay(500.0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    assert 0.4 < ms < 5.0, ms
    k.stream_delay(0.0)                     # off: no launch
    with pytest.raises(Exception):
        k.stream_delay(5000.0)              # a head start, not a timer
    assert k.wgrad_is_persistent(64, 64) and k.wgrad_is_persistent(64, 128) and not k.wgrad_is_persistent(128, 128)


@pytest.mark.parametrize("shape", [(2, 28, 28, 128, 256), (3, 14, 14, 256, 512), (2, 56, 56, 128, 128),
                                   (2, 28, 28, 64, 128), (3, 20, 36, 64, 64), (1, 48, 70, 64, 64)])
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("cfg", [3, 16, 17])
def test_batchnorm_statistics_from_the_convolution_epilogue(shape, pool, cfg):
    """The implicit GEMM in front of a training-mode BatchNorm leaves (sum y, sum y^2) per output
    tile; bn_fwd_train / bn_fwd_train_pool take them instead of reading y again.  Same statistics
    as the stand-alone reduction up to f32 summation order, same normalised output up to a bf16
    rounding at a boundary, and the hand-over only happens for the very tensor the conv wrote."""
    k = _k()
    N, H, W, C, Co = shape
    if C == 64 and cfg != 3:
        pytest.skip("64 reduction channels take conv_c64 (its store epilogue leaves the statistics): one case")
    torch.manual_seed(77)
    x = _rand(N, H, W, C)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * C)) ** 0.5)
    b = torch.randn(Co, device="cuda") * 0.5
    gamma = torch.rand(Co, device="cuda") + 0.5
    beta = torch.randn(Co, device="cuda") * 0.1

    def run(fused):
        k.fuse_bn_stats = fused
        if C != 64:
            k._overrides["igemm"] = lambda M, Nc, ks: (cfg, 1)     # tiny test grids would take split-K (no bf16 epilogue)
        try:
            y = k.conv3x3_fprop(x, w, b, relu=False, want_bn_stats=True)
        finally:
            k._overrides.pop("igemm", None)
        handed = k.bn_stats_slot.tensor is not None
        rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
        if pool:
            z, codes, stats = k.bn_fwd_train_pool(y, gamma, beta, rm, rv)
        else:
            z, stats = k.bn_fwd_train(y, gamma, beta, rm, rv)
            codes = None
        assert k.bn_stats_slot.tensor is None          # consumed (or never there)
        return y, z, codes, stats, rm, rv, handed
    try:
        y0, z0, c0, s0, rm0, rv0, h0 = run(False)
        y1, z1, c1, s1, rm1, rv1, h1 = run(True)
    finally:
        k.fuse_bn_stats = True
    assert not h0 and h1
    assert torch.equal(y0, y1)
    yf = y0.float().view(-1, Co)
    assert torch.allclose(s1[:Co], yf.mean(0), rtol=1e-4, atol=1e-5)
    assert torch.allclose(s1[Co:2 * Co], (yf.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4)
    assert torch.allclose(s0, s1, rtol=2e-5, atol=2e-6)
    assert torch.allclose(rm0, rm1, rtol=2e-5, atol=2e-6) and torch.allclose(rv0, rv1, rtol=2e-5, atol=2e-6)
    d = (z0.float() - z1.float()).abs()
    assert (d > 0).float().mean().item() < 1e-3 and (d <= 0.0079 * z0.float().abs() + 1e-6).all()
    # a different tensor of the same shape must not pick the statistics up
    if C != 64:
        k._overrides["igemm"] = lambda M, Nc, ks: (cfg, 1)
    try:
        y2 = k.conv3x3_fprop(x, w, b, relu=False, want_bn_stats=True)
    finally:
        k._overrides.pop("igemm", None)
    other = y2.clone()
    rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
    _, s2 = k.bn_fwd_train(other, gamma, beta, rm, rv)
    assert k.bn_stats_slot.tensor is None and torch.equal(s2, s0)


@pytest.mark.parametrize("shape,dtype", [((3, 3, 40, 36), torch.float32), ((2, 3, 64, 64), torch.uint8),
                                         ((5, 3, 30, 50), torch.bfloat16)])
def test_batchnorm_statistics_from_the_first_convolution(shape, dtype):
    """conv1_fprop's store loop leaves the batch statistics of a following BatchNorm (one partial
    row per persistent workgroup); bn_fwd_train takes them instead of reading y again."""
    k = _k()
    torch.manual_seed(78)
    if dtype == torch.uint8:
        x = torch.randint(0, 256, shape, device="cuda", dtype=torch.uint8)
        scale = 1.0 / 255.0
    else:
        x = torch.randn(shape, device="cuda").to(dtype)
        scale = 1.0
    w = _rand(64, 27, scale=(2.0 / 27) ** 0.5)
    b = torch.randn(64, device="cuda") * 0.3
    gamma = torch.rand(64, device="cuda") + 0.5
    beta = torch.randn(64, device="cuda") * 0.1
    res = []
    try:
        for fused in (False, True):
            k.fuse_bn_stats = fused
            y = k.conv1_fprop(x, w, b, relu=False, scale=scale, want_bn_stats=True)
            assert (k.bn_stats_slot.tensor is not None) == fused
            rm, rv = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
            z, stats = k.bn_fwd_train(y, gamma, beta, rm, rv)
            assert k.bn_stats_slot.tensor is None
            res.append((y, z, stats, rm, rv))
    finally:
        k.fuse_bn_stats = True
    (y0, z0, s0, rm0, rv0), (y1, z1, s1, rm1, rv1) = res
    assert torch.equal(y0, y1)
    yf = y0.float().view(-1, 64)
    assert torch.allclose(s1[:64], yf.mean(0), rtol=1e-4, atol=1e-5)
    assert torch.allclose(s1[64:128], (yf.var(0, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4)
    assert torch.allclose(s0, s1, rtol=2e-5, atol=2e-6)
    assert torch.allclose(rm0, rm1, rtol=2e-5, atol=2e-6) and torch.allclose(rv0, rv1, rtol=2e-5, atol=2e-6)
    d = (z0.float() - z1.float()).abs()
    assert (d > 0).float().mean().item() < 1e-3 and (d <= 0.0079 * z0.float().abs() + 1e-6).all()


@pytest.mark.parametrize("shape", [(2, 28, 28, 128, 256), (3, 14, 14, 256, 512), (2, 56, 56, 128, 128)])
@pytest.mark.parametrize("cfg", [3, 16, 17])
@pytest.mark.parametrize("relu", [True, False])
def test_batchnorm_backward_reductions_from_the_data_gradient_epilogue(shape, cfg, relu):
    """The data gradient that produces dz of a BatchNorm(+ReLU) leaves (sum dyh, sum dyh*y) per tile
    when it is given that BatchNorm's y / scale / shift; bn_bwd takes them instead of reading (dz, y)
    once more.  Same dgamma / dbeta / dy as the stand-alone reduction up to f32 summation order."""
    k = _k()
    N, H, W, C, Co = shape          # the BatchNorm has C channels; the convolution above maps C -> Co
    torch.manual_seed(79)
    y = (_rand(N, H, W, C).float() * 1.5 + 0.3).to(BF16)
    gamma = torch.rand(C, device="cuda") + 0.5
    beta = torch.randn(C, device="cuda") * 0.2
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    z, stats = k.bn_fwd_train(y, gamma, beta, rm, rv, relu=relu)
    dyn = _rand(N, H, W, Co, scale=0.1)
    w = _rand(Co, 9 * C, scale=(2.0 / (9 * C)) ** 0.5)
    wd = k.conv_weight_to_dgrad(w, Co, C)
    res = []
    try:
        for fused in (False, True):
            k.fuse_bn_stats = fused
            k._overrides["igemm"] = lambda M, Nc, ks: (cfg, 1)
            try:
                dz = k.conv3x3_dgrad(dyn, wd, bn_bwd_src=(y, stats, relu))
            finally:
                k._overrides.pop("igemm", None)
            assert (k.bn_bwd_slot.tensor is not None) == fused
            dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
            dy = k.bn_bwd(y, dz, gamma, stats, dg, db, relu=relu)
            assert k.bn_bwd_slot.tensor is None
            res.append((dz, dy, dg, db))
    finally:
        k.fuse_bn_stats = True
    (dz0, dy0, dg0, db0), (dz1, dy1, dg1, db1) = res
    assert torch.equal(dz0, dz1)
    # f32 reference of the two reductions
    yf, dzf = y.float().view(-1, C), dz0.float().view(-1, C)
    mean, rstd, sc, sh = stats[:C], stats[C:2 * C], stats[2 * C:3 * C], stats[3 * C:]
    dyh = dzf * ((yf * sc + sh) > 0) if relu else dzf
    ref_db = dyh.sum(0)
    ref_dg = (dyh * (yf - mean) * rstd).sum(0)
    tol_b = 1e-3 * dyh.abs().sum(0).max().item()
    assert (db1 - ref_db).abs().max().item() < tol_b and (dg1 - ref_dg).abs().max().item() < 2 * tol_b
    assert (db1 - db0).abs().max().item() < 1e-4 * dyh.abs().sum(0).max().item()
    assert (dg1 - dg0).abs().max().item() < 2e-4 * dyh.abs().sum(0).max().item()
    _close(dy1, dy0.float(), 1e-2, 1e-2 * dy0.float().abs().max().item(), "bn_bwd dy from epilogue reductions")


@pytest.mark.parametrize("shape", [(12, 3, 40, 40), (5, 3, 24, 41), (3, 1, 130, 130), (2, 3, 224, 224)])
def test_device_augmentation_matches_numpy(shape):
    """csrc/augment.hip (gather + rot90 / h-flip / v-flip + per-sample brightness-contrast / gamma table)
    against the numpy pipeline of data/device_augment.augment_reference, which uses the very expressions
    of the CPU dataset (data/image_folder.py): bit-identical, for square and non-square images, odd
    widths, several 64-row bands, with and without the gather."""
    import numpy as np
    from distributed_training_pytorch_amd import _lib
    from distributed_training_pytorch_amd._lib import check, ptr, stream
    from distributed_training_pytorch_amd.data import device_augment as da
    N, C, H, W = shape
    rs = np.random.RandomState(7)
    cache = rs.randint(0, 256, size=(N + 9, C, H, W), dtype=np.uint8)
    sel = rs.permutation(N + 9)[:N].astype(np.int64)
    p = da.draw_params(sel, epoch=3, seed=11, prob=0.7, square=(H == W))
    assert H != W or N < 8 or len(set(p["k"].tolist())) > 1          # quarter turns really drawn
    ref = da.augment_reference(cache[sel], p)
    x = torch.from_numpy(cache).cuda()
    out = da.DeviceAugmenter(seed=11, prob=0.7)(x, torch.from_numpy(sel).cuda(), sel, 3)
    torch.cuda.synchronize()
    assert out.dtype == torch.uint8 and tuple(out.shape) == (N, C, H, W)
    assert np.array_equal(out.cpu().numpy(), ref)
    # identity: no geometry, no table, no gather
    out2 = torch.empty_like(x)
    check(_lib.hip().dtp_augment_u8(ptr(x), None, None, None, ptr(out2), N + 9, C, H, W, 1, stream()), "augment_u8")
    assert torch.equal(out2, x)
    # other epoch / other seed -> other draws
    p2 = da.draw_params(sel, epoch=4, seed=11, prob=0.7, square=(H == W))
    assert not np.array_equal(da.geom_codes(p), da.geom_codes(p2)) or N < 4


def test_transpose_plan_matches_single_transposes():
    """All derived weight layouts of a model in ONE launch (K.TransposePlan, csrc/eltwise.hip
    transpose_multi_kernel) == the per-layer conv_weight_to_dgrad / linear_weight_transpose kernels,
    bit for bit, incl. odd sizes, the rotated-tap negative batch stride and zero padding columns."""
    k = _k()
    torch.manual_seed(3)
    convs = [(64, 3), (128, 64), (512, 256), (72, 40)]          # (Cout, Cin)
    lins = [(4096, 512), (3, 4096), (100, 72)]                   # (Nout, Kin)
    jobs, pairs, keep = [], [], []                               # (the plan reads its sources at run())
    for Co, Ci in convs:
        w = _rand(Co, 9 * Ci)
        keep.append(w)
        out = torch.full((Ci, 9 * Co), 7.0, dtype=BF16, device="cuda")
        jobs.append(k.conv_weight_to_dgrad_job(w, Co, Ci, out))
        pairs.append((out, k.conv_weight_to_dgrad(w, Co, Ci)))
    for No, Ki in lins:
        w = _rand(No, Ki)
        keep.append(w)
        ld = (No + 63) // 64 * 64
        out = torch.zeros((Ki, ld), dtype=BF16, device="cuda")
        jobs.append(k.linear_weight_transpose_job(w, out))
        pairs.append((out, k.linear_weight_transpose(w)))
    plan = k.TransposePlan(jobs, torch.device("cuda"))
    plan.run()
    torch.cuda.synchronize()
    for i, (got, ref) in enumerate(pairs):
        assert torch.equal(got, ref), "job %d differs" % i


def test_model_derived_layouts_from_one_launch(monkeypatch):
    """FusedModel builds every conv dgrad filter / classifier W^T with one transpose_multi launch per
    optimizer step: training is bit-identical to the per-layer launches (DTP_MULTI_TRANSPOSE=0)."""
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.ops.layers import SoftmaxCrossEntropy
    from distributed_training_pytorch_amd.ops.optim import FusedSGD

    def run(multi):
        monkeypatch.setenv("DTP_MULTI_TRANSPOSE", "1" if multi else "0")
        torch.manual_seed(0)
        model = VGG16(3, 3, True, grad_dtype=BF16).cuda().train()
        opt = FusedSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
        crit = SoftmaxCrossEntropy()
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(4, 3, 64, 64, device="cuda", generator=g)
        y = torch.randint(0, 3, (4,), device="cuda", generator=g)
        for _ in range(3):
            opt.zero_grad()
            crit(model(x), y).backward()
            opt.step()
        torch.cuda.synchronize()
        assert (model.__dict__.get("_transpose_plan") or (None, None))[1] is not None if multi else True
        return model.param_arena.flat_master.clone()
    a, b = run(True), run(False)
    assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(2, 28, 28, 256, 256, 1), (3, 14, 14, 256, 512, 1), (5, 14, 14, 512, 256, 3),
                                  (1, 8, 8, 256, 256, 1), (2, 10, 6, 256, 304, 2)])
def test_wgrad_8phase_is_bit_identical_to_the_two_stage_kernel(case):
    """wgrad2.hip cfg 10 (256x256 tile on the 8-phase schedule: half-tile slots, counted vmcnt, wave rows staggered
    by one barrier) sums the same products in the same order as cfg 8: bit-identical weight gradients, incl.
    K tails, odd K-tile counts (the all-zero padding tile), M tails and split-K; and both match the f32 reference."""
    k = _k()
    N, H, W, C, Co, ns = case
    torch.manual_seed(5)
    x, dy = _rand(N, H, W, C), _rand(N, H, W, Co, scale=0.1)
    outs = []
    try:
        for cfg in (8, 10):
            k._overrides["wgrad"] = lambda M, C_, taps, ks, cfg=cfg: (cfg, ns)
            gw = torch.full((Co, 9 * C), 3.0, device="cuda")
            k.conv3x3_wgrad(dy, x, gw)
            torch.cuda.synchronize()
            outs.append(gw)
    finally:
        k._overrides.pop("wgrad", None)
    assert torch.equal(outs[0], outs[1])
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    w = torch.zeros(Co, C, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(xr, w, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(Co, 9 * C)
    _close(outs[1], ref, 2e-2, 2e-2 * ref.abs().max().item(), "wgrad 8-phase vs f32 reference")


def test_wgrad_8phase_linear_form_is_bit_identical():
    k = _k()
    torch.manual_seed(6)
    for B, Nout, Kin, ns in ((512, 4096, 1024, 1), (200, 300, 1000, 1), (64, 256, 6272, 2)):
        x = _rand(B, Kin)
        ld = (Nout + 63) // 64 * 64
        dy = torch.zeros(B, ld, device="cuda", dtype=BF16)
        dy[:, :Nout] = _rand(B, Nout, scale=0.1)
        outs = []
        try:
            for cfg in (8, 10):
                k._overrides["wgrad"] = lambda M, C_, taps, ks, cfg=cfg, ns=ns: (cfg, ns)
                gw = torch.full((Nout, Kin), 3.0, device="cuda")
                k.linear_wgrad(dy, x, gw, n_valid=Nout)
                torch.cuda.synchronize()
                outs.append(gw)
        finally:
            k._overrides.pop("wgrad", None)
        assert torch.equal(outs[0], outs[1]), (B, Nout, Kin, ns)
        ref = dy[:, :Nout].float().t() @ x.float()
        _close(outs[1], ref, 2e-2, 2e-2 * ref.abs().max().item(), "linear wgrad 8-phase")


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 9, 11, 16, 17, 18, 19])
@pytest.mark.parametrize("relu_mask", [True, False])
def test_dgrad_unpools_in_its_epilogue_bit_for_bit(cfg, relu_mask):
    """conv3x3_dgrad(unpool=...) == maxpool2x2_bwd(conv3x3_dgrad(...)) bit for bit, for every tile
    shape of the implicit GEMM, with image rows that wrap inside a lane's row stride (W = 10 < G),
    a row tail in every tile size, and the column sums of the routed values left behind."""
    k = _k()
    torch.manual_seed(21)
    N, H, W, C, Co = 3, 18, 10, 192, 128           # pooled map 18x10 (M = 540), pool input 36x20
    if cfg == 18:
        k.igemm_shared_a = True
    xin = _rand(N, 2 * H, 2 * W, C)                 # the pool's input: about half of the maxima are <= 0
    _, codes = k.maxpool2x2_fwd(xin, want_idx=True)
    dy = _rand(N, H, W, Co)
    wd = _rand(C, 9 * Co, scale=(2.0 / (9 * Co)) ** 0.5)
    k._overrides["igemm"] = lambda M, Nc, ks: (cfg, 1)
    try:
        full = k.conv3x3_dgrad(dy, wd, unpool=(codes, relu_mask))
        assert full is not None and tuple(full.shape) == (N, 2 * H, 2 * W, C)
        assert k.colsum_slot.tensor is full
        bias_g = torch.empty(C, device="cuda")
        k.bias_grad(full.view(-1, C), bias_g)
        dxp = k.conv3x3_dgrad(dy, wd)
    finally:
        k._overrides.pop("igemm")
    ref = k.maxpool2x2_bwd(None, dxp, relu_mask=relu_mask, idx=codes, in_shape=(N, 2 * H, 2 * W, C))
    k.colsum_slot.take(ref, C)
    assert torch.equal(full, ref), "cfg %d: %d elements differ" % (cfg, (full != ref).sum().item())
    _close(bias_g, ref.float().view(-1, C).sum(0), 1e-5, 1e-3, "un-pooled colsum cfg%d" % cfg)


def test_dgrad_unpool_declines_what_it_cannot_do():
    k = _k()
    N, H, W, C, Co = 2, 8, 8, 64, 128
    dy, wd = _rand(N, H, W, Co), _rand(C, 9 * Co)
    codes = torch.zeros((N, H, W, C // 8), dtype=torch.int32, device="cuda")
    k._overrides["igemm"] = lambda M, Nc, ks: (3, 2)            # split-K: no bf16 C-image epilogue
    try:
        assert k.conv3x3_dgrad(dy, wd, unpool=(codes, True)) is None
    finally:
        k._overrides.pop("igemm")
    k.fuse_unpool = False
    try:
        assert k.conv3x3_dgrad(dy, wd, unpool=(codes, True)) is None
    finally:
        k.fuse_unpool = True
    with pytest.raises(ValueError):
        k.conv3x3_dgrad(dy, wd, unpool=(codes[:, :4], True))
