"""Raw (non-autograd) Python entry points of the HIP kernels, with host-side shape checks.

Every function validates operand shapes/dtypes/contiguity *before* the launch (a faulting kernel
can reset the GPU), picks a tile configuration / split-K factor for 256 CUs in 8 XCDs, and
launches on the caller's current HIP stream.  Tensors are internal-layout: activations NHWC bf16
``[N, H, W, C]``, conv weights ``[Cout, 9*Cin]`` (tap-major, channel-minor), linear weights
``[out, in]``.
"""
import ctypes as _ctypes
import os as _os

import torch

from .. import _lib
from .._lib import ptr, stream, check, workspace

BF16 = torch.bfloat16
N_CU = 256
wgrad_8phase = _os.environ.get("DTP_WGRAD_8PHASE", "1") == "1"     # 256x256 weight-gradient tile on the 8-phase schedule

# (BM, BN) per igemm cfg id / per wgrad cfg id — must match csrc/igemm.hip, csrc/wgrad.hip
IGEMM_TILES = {0: (256, 256), 1: (256, 128), 2: (256, 64), 3: (128, 128), 4: (128, 64), 16: (256, 256), 17: (512, 128),
               18: (256, 256),   # cfg 16's schedule over ONE staged activation image per tap row (3x3, C % 128 == 0)
               19: (384, 128),   # the same for outputs of <= 128 channels (4x2 waves of 96x64)
               20: (256, 256), 21: (384, 128),   # cfg 18 / 19 + ReLU + 2x2 max-pool in the epilogue (pooled output)
               8: (256, 256), 9: (256, 128), 10: (256, 64), 11: (128, 128), 12: (128, 64)}
WGRAD_TILES = {2: (128, 64), 3: (64, 128), 4: (64, 64),    # wgrad.hip: small-tile fallbacks (narrow outputs)
               8: (256, 256), 9: (128, 128),   # wgrad2.hip: asm transposed reads (DMA stays in flight), SRD staging
               10: (256, 256)}                 # wgrad2.hip: the 256x256 tile on the 8-phase schedule

_overrides = {}   # tuning hooks: {"igemm": fn(M, N, K, taps) -> (cfg, nsplit), "wgrad": ...}


def _need(t, dtype, name):
    if not t.is_cuda:
        raise ValueError("%s must be a CUDA tensor" % name)
    if t.dtype != dtype:
        raise ValueError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if t.data_ptr() % 16 != 0:
        raise ValueError("%s must be 16-byte aligned" % name)


def _cdiv(a, b):
    return (a + b - 1) // b


def pick_igemm(M, N, ksteps):
    """Largest tile that still yields >= ~2 workgroups per CU; split-K when the grid is tiny."""
    if "igemm" in _overrides:
        return _overrides["igemm"](M, N, ksteps)
    # measured on MI355X (bench/bench_kernels.py --sweep, batch 128, random data): the
    # software-pipelined 256x256 / 256x128 variants (cfg 8 / 9) win on wide outputs with many
    # rows; 128x128 at two workgroups per CU (cfg 3) wins on mid-sized grids; 64-wide outputs
    # are LDS-fill bound and prefer the plain 256x64 loop (cfg 2).
    if N > 128:
        order = [16, 3]     # 8-phase counted-vmcnt schedule: +10-18 % over the 2-stage loops
        # long reductions over few output tiles (linear1 at batch 1024: 4 x 16 tiles of 256x256, 392 K-steps):
        # split K so that the 8-phase tiles fill the chip once — 0.344 ms (128x128 tiles, 612 TF/s) -> 0.205 ms
        # (1,027 TF/s), bench/bench_fc.py; linear2 (64 K-steps) stays on the 128x128 tiles (measured faster)
        t256 = _cdiv(M, 256) * _cdiv(N, 256)
        if N_CU // 8 <= t256 < N_CU // 2 and ksteps >= 128:
            ns = max(1, N_CU // t256)
            if ksteps // ns >= 32:
                return 16, ns
    elif N > 64:
        order = [3]         # re-swept after the chunk-major K order: 128x128 beats the pipelined 256x128
        # 512x128 8-phase tile (one workgroup per CU): +11 % on conv2_2 forward, +3 % on its data
        # gradient at batch 256 (bench/bench_igemm17.py), but only while the last round of tiles
        # is nearly full (conv3_1's data gradient, 6.1 rounds: -6 %)
        t512 = _cdiv(M, 512)
        if igemm_512 and t512 >= 8 * N_CU and t512 >= 0.93 * _cdiv(t512, N_CU) * N_CU:
            return 17, 1
    else:
        order = [2, 4]
    cfg = order[-1]
    for c in order:
        bm, bn = IGEMM_TILES[c]
        # 256x256 tiles run one workgroup per CU: even a single partial round (196 tiles for
        # conv5 at batch 128) beats 128x128 tiles by 1.4x (sweep: 1153 vs 832 TF/s)
        need = N_CU // 2 if bm * bn >= 256 * 256 else 2 * N_CU
        if _cdiv(M, bm) * _cdiv(N, bn) >= need:
            cfg = c
            break
    bm, bn = IGEMM_TILES[cfg]
    tiles = _cdiv(M, bm) * _cdiv(N, bn)
    # split-K only when fewer than half the CUs would get a tile: one partial round of 256x256
    # tiles (196 tiles for conv5 at batch 128: 1072 TF/s) beats a 3-way split with f32 slabs (497)
    nsplit = 1
    if tiles < N_CU // 2 and ksteps >= 16:
        nsplit = max(1, min(ksteps // 8, _cdiv(2 * N_CU, tiles)))
    return cfg, nsplit


def pick_wgrad(M, C, taps, ksteps):
    if "wgrad" in _overrides:
        return _overrides["wgrad"](M, C, taps, ksteps)
    if taps == 9:
        # cfg 8 / 9 (wgrad2.hip): the LDS-DMA of the next K-tile stays in flight under the MFMAs
        # (hipcc drains it in front of builtin transposed reads, the round-1 kernels): 1.13-1.18 PF/s vs
        # 0.89-0.95 on the 256/512-channel layers, 0.92-0.97 vs 0.69 on the 128-channel ones
        # (batch 256, bench/bench_wgrad2.py, profiles/23_*)
        if M >= 256 and C % 256 == 0:
            cfg = 8
        elif M >= 128 and C % 128 == 0:
            cfg = 9
        elif M >= 128:
            cfg = 2
        elif C % 128 == 0:
            cfg = 3
        else:
            cfg = 4
    else:
        if M >= 256 and C >= 256:
            cfg = 8
        elif M >= 128:
            cfg = 9 if C >= 128 else 2
        else:
            cfg = 3 if C >= 128 else 4
    bm, bn = WGRAD_TILES[cfg]
    tiles = _cdiv(M, bm) * (taps * (C // bn) if taps == 9 else _cdiv(C, bn))
    # split the reduction so that tiles*nsplit fills whole rounds of the 256 CUs (big tiles run one
    # workgroup per CU); fewer, fuller rounds also mean fewer f32 partial slabs to write and re-read
    per_cu = 1 if bm * bn >= 256 * 256 else 2
    slots = N_CU * per_cu
    max_split = max(1, ksteps // 8) if ksteps >= 16 else 1
    nsplit = 1
    for rounds in (1, 2, 3, 4):
        n = (slots * rounds) // tiles
        if n >= 1:
            nsplit = min(n, max_split)
            if n <= max_split and tiles * n >= 0.85 * slots * rounds:
                break
    nsplit = max(1, nsplit)
    if cfg == 8 and wgrad_8phase and ksteps // nsplit >= 160:
        # the same tile on the 8-phase schedule (wgrad_tt8_kernel): same sums, bit for bit; +1-4 % isolated on
        # long reductions (its longer prologue loses on short ones: 112 K-tiles per split -2.5 %), +1.15 % on
        # the full step where it shares the chip with the data gradient (bench/r3_wg8.sh)
        cfg = 10
    return cfg, nsplit


class _ColsumSlot(object):
    """Column sums that a producer kernel left behind for the NEXT backward function.

    The bias gradient of a layer is the column sum of its dY, and dY is the output of an earlier
    kernel of the same backward pass (the next layer's data-gradient with the ReLU mask fused, or
    the max-pool backward).  Those kernels fold their output tile into per-workgroup f32 partial
    rows while it is still in registers, so the 1.6 GB (conv1_2, batch 256) re-read of dY by a
    separate column-sum kernel disappears.  One slot: the producer overwrites it, the consumer
    takes it only when its dY is *the same tensor* (the slot keeps the producer's output alive, so
    an equal data_ptr cannot be a recycled allocation)."""

    def __init__(self):
        self.tensor = None
        self.parts = None
        self.nparts = 0
        self.C = 0

    def put(self, tensor, parts, nparts, C):
        self.tensor, self.parts, self.nparts, self.C = tensor, parts, nparts, C

    def take(self, dy2d_or_4d, C):
        t = self.tensor
        self.tensor = None
        if (t is None or t.data_ptr() != dy2d_or_4d.data_ptr() or t.numel() != dy2d_or_4d.numel()
                or self.C != C or t.device != dy2d_or_4d.device):
            return None
        return self.parts, self.nparts


colsum_slot = _ColsumSlot()
# the same hand-over for BatchNorm's batch statistics: a convolution followed by a training-mode
# BatchNorm leaves (sum y, sum y^2) per output tile ([tiles][2][C]), bn_fwd_train* takes them when
# its input is that very tensor and skips its own pass over y
bn_stats_slot = _ColsumSlot()
# ... and for its backward: the data gradient that produces dz (the gradient of relu(bn(y))) leaves
# (sum dyh, sum dyh*y) per tile when it is told y / scale / shift of that BatchNorm; bn_bwd takes them
bn_bwd_slot = _ColsumSlot()
fuse_bn_stats = _os.environ.get("DTP_FUSE_BN_STATS", "1") != "0"      # A/B switch
fuse_conv_pool = _os.environ.get("DTP_FUSE_CONV_POOL", "1") != "0"    # A/B switch
fuse_igemm_pool = _os.environ.get("DTP_FUSE_IGEMM_POOL", "1") != "0"  # A/B switch: ... for the implicit-GEMM layers (blocks 2-5)
fuse_unpool = _os.environ.get("DTP_FUSE_UNPOOL", "1") != "0"          # A/B switch: pool backward inside the data gradient above it
fuse_colsum = True      # tests flip this to compare against the stand-alone column-sum kernel
# conv2_1 dgrad as 2 conv_c64 passes: measured 640 vs 662 TF/s for the implicit GEMM at batch 256 and
# no end-to-end difference (bench/bench_dgrad_c128.py, profiles/18_*): opt-in only
two_pass_c128 = _os.environ.get("DTP_DGRAD_TWO_PASS", "0") == "1"
igemm_512 = _os.environ.get("DTP_IGEMM_512", "1") != "0"              # A/B switch: 512x128 8-phase tile
igemm_shared_a = _os.environ.get("DTP_IGEMM_SHARED_A", "1") != "0"    # A/B switch: one staged activation image per tap row (cfg 18)
igemm_shared_a_n128 = _os.environ.get("DTP_IGEMM_SHARED_A_N128", "1") != "0"   # A/B switch: ... for <= 128 output channels (cfg 19)
relu_bitmask = _os.environ.get("DTP_RELU_BITMASK", "1") != "0"        # A/B switch: 1-bit ReLU masks for dgrad


_cs_parts_flip = [0]


def _colsum_parts_workspace(numel, device):
    """Partial-row buffer for the next producer: two buffers, used alternately.  The rows a producer leaves are
    reduced by the bias gradient of the NEXT backward function, which may run on the weight-gradient side stream
    while that function's own data gradient (the next producer) is already writing: it writes the other buffer,
    and the side stream is joined before the function after that starts."""
    _cs_parts_flip[0] ^= 1
    return workspace.get("colsum_parts%d" % _cs_parts_flip[0], numel, torch.float32, device)


def colsum_parts(parts, nparts, C, out, n_valid=None, beta=0.0, ws_key="colsum"):
    """out[:n_valid] = beta*out + sum of the producer's partial rows parts[nparts][C]."""
    nv = C if n_valid is None else n_valid
    if out.numel() != nv or out.dtype not in (torch.float32, BF16) or parts.numel() < nparts * C:
        raise ValueError("colsum_parts: bad shapes")
    ws = workspace.get(ws_key, 1024 * max(C, 64), torch.float32, parts.device)
    rc = _lib.hip().dtp_colsum_parts(ptr(parts), nparts, C, nv, ptr(out), ptr(ws), ws.numel(),
                                     int(out.dtype == torch.float32), float(beta), stream())
    check(rc, "colsum_parts")
    return out


def bias_grad(dy2d, out, n_valid=None, beta=0.0, ws_key="colsum"):
    """Bias gradient = column sums of dY: from the producer's partial rows when the kernel that
    wrote dY left them (see _ColsumSlot), else with the two-pass column-sum kernel.  ``ws_key``: scratch
    buffer name — a caller on a side stream passes its own so that it shares nothing with the main stream."""
    got = colsum_slot.take(dy2d, dy2d.shape[-1]) if fuse_colsum else None
    if got is not None:
        return colsum_parts(got[0], got[1], dy2d.shape[-1], out, n_valid, beta, ws_key=ws_key)
    return colsum(dy2d, out, n_valid, beta, ws_key=ws_key)


def _igemm(A, Bm, bias, mask_src, out, M, Ncols, C, taps, H, W, lda, ldb, ldo, out_fp32, relu,
           mask_scale, drop_p, seed, want_colsum=False, seed_ptr=None, fp8_scales=None, mask_bits=None,
           bits_out=None, fp8_mode=1, want_bn_stats=False, bn_bwd_src=None, pool_codes=None, unpool=None):
    """``unpool`` = (codes, relu_mask) of the 2x2 max-pool whose output these M rows are the gradient of:
    ``out`` is then the full-resolution gradient [4*M][ldo] (csrc/igemm.hip, un-pooling epilogue).
    Returns None instead when the chosen kernel cannot do it (split-K): the caller un-pools separately."""
    fp8 = fp8_scales is not None
    ksteps = taps * (C // (128 if fp8 else 64))
    if pool_codes is not None:
        # conv + ReLU + 2x2 max-pool in one kernel: `out` is the pooled tensor (csrc/igemm.hip POOL)
        cfg = 20 if Ncols > 128 else 21
        rc = _lib.hip().dtp_igemm_nt(ptr(A), ptr(Bm), ptr(bias), None, None, None, ptr(out), None, None,
                                     ptr(workspace.zero_page(A.device)), M, Ncols, C, taps, H, W, lda, ldb, ldo, 0, 1,
                                     1.0, 0.0, 0, None, 0, None, None, 0, None, None, None, ptr(pool_codes), None, 0,
                                     1, cfg, stream())
        check(rc, "igemm_nt (conv + relu + pool)")
        return out
    cfg, nsplit = pick_igemm(M, Ncols, ksteps)
    # cfg 18 = cfg 16 with the three dx taps of a filter row reading one staged activation image
    # (csrc/igemm.hip PIPE 3: a third of the A-operand LDS-DMA / L2 reads; same sums, bit for bit)
    shared_ok = taps == 9 and C % 128 == 0 and nsplit == 1 and not fp8
    if cfg == 16 and shared_ok and igemm_shared_a:
        cfg = 18
    elif cfg == 18 and not shared_ok:
        cfg = 16
    elif (cfg in (17, 3) and 64 < Ncols <= 128 and shared_ok and igemm_shared_a and igemm_shared_a_n128
          and "igemm" not in _overrides and _cdiv(M, 384) >= 4 * N_CU):
        cfg = 19          # 384x128 tiles over the shared image: conv2_2 forward / data gradient, conv3_1's data gradient
    elif cfg == 19 and not shared_ok:
        cfg = 3
    if fp8:      # fp8 operands are instantiated for the 2-stage 256x128 / 128x128 / 128x64 tiles
        cfg = {16: 1, 0: 1, 8: 1, 9: 1, 1: 1, 11: 3, 3: 3, 2: 4, 10: 4, 12: 4, 4: 4}[cfg]
        if IGEMM_TILES[cfg][0] * IGEMM_TILES[cfg][1] < 256 * 256 and nsplit == 1:
            tiles = _cdiv(M, IGEMM_TILES[cfg][0]) * _cdiv(Ncols, IGEMM_TILES[cfg][1])
            if tiles < N_CU and ksteps >= 16:
                nsplit = max(1, min(ksteps // 8, _cdiv(2 * N_CU, tiles)))
    if unpool is not None and (nsplit > 1 or fp8 or out_fp32 or ldo != Ncols or cfg in (0, 8)):
        return None
    partial = None
    if nsplit > 1:
        partial = workspace.get("igemm_partial", nsplit * M * Ncols, torch.float32, A.device)
    cs_part = None
    if want_colsum and fuse_colsum and nsplit == 1 and not out_fp32 and ldo == Ncols:
        nparts = _cdiv(M, IGEMM_TILES[cfg][0])
        cs_part = _colsum_parts_workspace(nparts * Ncols, A.device)
        colsum_slot.put(out, cs_part, nparts, Ncols)
    colsum_sq = 0
    if (want_bn_stats and fuse_bn_stats and cs_part is None and nsplit == 1 and not out_fp32 and ldo == Ncols
            and mask_src is None and mask_bits is None):
        nparts = _cdiv(M, IGEMM_TILES[cfg][0])
        cs_part = workspace.get("bn_tile_stats", nparts * 2 * Ncols, torch.float32, A.device)
        bn_stats_slot.put(out, cs_part, nparts, Ncols)
        colsum_sq = 1
    bn_y = bn_sc = bn_sh = None
    if (bn_bwd_src is not None and fuse_bn_stats and cs_part is None and nsplit == 1 and not out_fp32
            and ldo == Ncols and mask_src is None and mask_bits is None and not relu and bias is None):
        y_src, stats, bn_relu = bn_bwd_src          # stats = [mean | rstd | scale | shift]
        if y_src.numel() == M * Ncols and y_src.dtype == BF16 and y_src.is_contiguous() and stats.numel() == 4 * Ncols:
            nparts = _cdiv(M, IGEMM_TILES[cfg][0])
            cs_part = workspace.get("bn_tile_stats_bwd", nparts * 2 * Ncols, torch.float32, A.device)
            bn_bwd_slot.put(out, cs_part, nparts, Ncols)
            colsum_sq = 1
            bn_y = y_src
            if bn_relu:
                bn_sc, bn_sh = stats[2 * Ncols:3 * Ncols], stats[3 * Ncols:]
    if nsplit > 1 or out_fp32:          # the 1-bit masks live in the bf16 C-image epilogue only
        if bits_out is not None:
            raise ValueError("igemm: bits_out needs the single-pass bf16 epilogue")
        mask_bits = None
    if mask_bits is not None:
        mask_src = None
    rc = _lib.hip().dtp_igemm_nt(ptr(A), ptr(Bm), ptr(bias), ptr(mask_src), ptr(mask_bits), ptr(bits_out), ptr(out),
                                 ptr(partial), ptr(cs_part),
                                 ptr(workspace.zero_page(A.device)), M, Ncols, C, taps, H, W, lda, ldb, ldo,
                                 int(out_fp32), int(relu), float(mask_scale), float(drop_p), int(seed) & (2 ** 64 - 1),
                                 ptr(seed_ptr), int(fp8_mode) if fp8 else 0, ptr(fp8_scales[0]) if fp8 else None,
                                 ptr(fp8_scales[1]) if fp8 else None, colsum_sq, ptr(bn_y), ptr(bn_sc), ptr(bn_sh),
                                 None, ptr(unpool[0]) if unpool is not None else None,
                                 int(bool(unpool[1])) if unpool is not None else 0, nsplit, cfg, stream())
    check(rc, "igemm_nt")
    return out


def _conv_c64(x, w, bias, mask_src, out, N, H, W, O, relu, mask_scale, codes=None, bits_out=None, ldx=64,
              acc_src=None, want_bn_stats=False):
    """64 reduction channels: register-resident filters + LDS row ring (csrc/conv_c64.hip).
    ``ldx``: pixel stride of x in elements (x may be a 64-channel slice of a wider tensor);
    ``acc_src``: bf16 partial sums added before the epilogue (may be ``out`` itself)."""
    part, rows, nparts = None, 0, _ctypes.c_int(0)
    if want_bn_stats and fuse_bn_stats and codes is None and mask_src is None and acc_src is None:
        rows = N * _cdiv(W, 32) * max(1, H // 16)          # one partial row per workgroup: the launcher's upper bound
        part = workspace.get("bn_tile_stats", rows * 2 * O, torch.float32, x.device)
    rc = _lib.hip().dtp_conv_c64(ptr(x), ptr(w), ptr(bias), ptr(mask_src), ptr(out), ptr(codes), ptr(bits_out),
                                 ptr(workspace.zero_page(x.device)), N, H, W, O, int(relu), float(mask_scale),
                                 int(ldx), ptr(acc_src), ptr(part), rows, _ctypes.byref(nparts), stream())
    check(rc, "conv_c64")
    if part is not None:
        bn_stats_slot.put(out, part, nparts.value, O)
    return out


def relu_bits_for(out):
    """Empty 1-bit-per-element mask buffer for a bf16 activation (byte i covers elements 8i..8i+7)."""
    return torch.empty(out.numel() // 8, dtype=torch.uint8, device=out.device)


def conv3x3_fprop(x, w, bias=None, relu=True, out=None, want_bits=False, want_bn_stats=False):
    """y = relu(conv3x3_same(x, w) + bias); x [N,H,W,C] bf16, w [Cout, 9*C] bf16, bias f32 [Cout].
    ``want_bits``: also return (y > 0) packed 1 bit per element — the ReLU-backward mask the data
    gradient of the NEXT convolution applies to its output (16x fewer bytes than re-reading y)."""
    _need(x, BF16, "x")
    _need(w, BF16, "w")
    N, H, W, C = x.shape
    Cout = w.shape[0]
    if w.numel() != Cout * 9 * C or C % 64 or Cout % 8:
        raise ValueError("conv3x3_fprop: bad shapes x=%s w=%s" % (tuple(x.shape), tuple(w.shape)))
    if bias is not None:
        _need(bias, torch.float32, "bias")
        if bias.numel() != Cout:
            raise ValueError("bias size")
    if out is None:
        out = torch.empty((N, H, W, Cout), dtype=BF16, device=x.device)
    else:
        _need(out, BF16, "out")
        if out.numel() != N * H * W * Cout:
            raise ValueError("out size")
    bits = None
    if want_bits:
        M, ks = N * H * W, 9 * (C // 64)
        c64 = C == 64 and Cout in (64, 128) and "igemm" not in _overrides
        if c64 or pick_igemm(M, Cout, ks)[1] == 1:       # split-K passes have no bf16 epilogue to pack from
            bits = relu_bits_for(out)
    if C == 64 and Cout in (64, 128) and "igemm" not in _overrides:
        _conv_c64(x, w, bias, None, out, N, H, W, Cout, relu, 1.0, bits_out=bits, want_bn_stats=want_bn_stats)
    else:
        _igemm(x, w, bias, None, out, N * H * W, Cout, C, 9, H, W, C, 9 * C, Cout, False, relu, 1.0, 0.0, 0,
               bits_out=bits, want_bn_stats=want_bn_stats)
    return (out, bits) if want_bits else out


def conv_pool_fusable(C, Cout, H, W, N=None):
    """Shapes for which conv3x3 + ReLU + 2x2 max-pool run as ONE kernel: 64 reduction channels
    (conv_c64's POOL epilogue: block 1), or the shared-image implicit GEMM with the pooled epilogue
    (cfg 20 / 21: reduction channels a multiple of 128, > 64 outputs, even H, even W >= 14 — the block
    ends of VGG16's blocks 2-5).  ``N`` (batch): the implicit-GEMM form needs enough tiles to fill the
    CUs (small batches keep the split-K capable unfused kernels)."""
    if not fuse_conv_pool or H % 2 or W % 2 or "igemm" in _overrides:
        return False
    if C == 64 and Cout in (64, 128):
        return True
    if not (fuse_igemm_pool and igemm_shared_a and C % 128 == 0 and Cout > 64 and Cout % 8 == 0 and W >= 14):
        return False
    if N is None:
        return True
    bm, bn = IGEMM_TILES[20 if Cout > 128 else 21]
    return _cdiv(N * H * W, bm) * _cdiv(Cout, bn) >= N_CU // 2


def conv3x3_relu_pool_fprop(x, w, bias):
    """(maxpool2x2(relu(conv3x3(x, w) + bias)), codes): the full-resolution activation is never
    written — backward needs only the pooled output's 4-bit (argmax, positive) codes."""
    _need(x, BF16, "x")
    _need(w, BF16, "w")
    N, H, W, C = x.shape
    Cout = w.shape[0]
    if w.numel() != Cout * 9 * C or not conv_pool_fusable(C, Cout, H, W):
        raise ValueError("conv3x3_relu_pool_fprop: unsupported shapes x=%s w=%s" % (tuple(x.shape), tuple(w.shape)))
    if bias is not None:
        _need(bias, torch.float32, "bias")
    y = torch.empty((N, H // 2, W // 2, Cout), dtype=BF16, device=x.device)
    codes = torch.empty((N, H // 2, W // 2, Cout // 8), dtype=torch.int32, device=x.device)
    if C == 64:
        _conv_c64(x, w, bias, None, y, N, H, W, Cout, True, 1.0, codes=codes)
    else:
        _igemm(x, w, bias, None, y, N * H * W, Cout, C, 9, H, W, C, 9 * C, Cout, False, True, 1.0, 0.0, 0,
               pool_codes=codes)
    return y, codes


def split_dgrad_weight(wd, Cin, Co):
    """The dgrad-layout weight [Cin][9][Co] cut into its two 64-channel reduction halves, each
    [Cin][9*64] (operands of the two-pass conv_c64 data gradient)."""
    w3 = wd.view(Cin, 9, Co)
    return (w3[:, :, :64].contiguous().view(Cin, 9 * 64), w3[:, :, 64:128].contiguous().view(Cin, 9 * 64))


def conv3x3_dgrad(dy, wd, mask_src=None, mask_scale=1.0, out=None, mask_bits=None, wd_halves=None, bn_bwd_src=None,
                  unpool=None):
    """dx = conv3x3_same(dy, wd) [* (mask_src > 0)]; wd is the dgrad-layout weight [Cin, 9*Cout].
    ``mask_bits`` (from ``conv3x3_fprop(..., want_bits=True)`` of the layer that produced
    ``mask_src``) replaces the 2-byte-per-element mask read by 1 bit per element.
    ``bn_bwd_src`` = (y, stats, relu) of the training-mode BatchNorm(+ReLU) whose OUTPUT this
    convolution consumed: dx is that BatchNorm's dz, and the kernel leaves the two reductions of
    its backward pass behind (``bn_bwd_slot``).
    ``unpool`` = (codes, relu_mask): the convolution's input was the output of a 2x2 max-pool with
    those forward codes; the result is the gradient of the pool's INPUT, [N, 2H, 2W, Cin], with its
    column sums left behind — or None when this shape has no fused path (caller: plain dgrad, then
    :func:`maxpool2x2_bwd`)."""
    _need(dy, BF16, "dy")
    _need(wd, BF16, "wd")
    N, H, W, Co = dy.shape
    Cin = wd.shape[0]
    if unpool is not None:
        codes, relu_mask = unpool
        if (not fuse_unpool or mask_src is not None or out is not None or (Co == 64 and "igemm" not in _overrides)
                or (Co == 128 and Cin == 64 and two_pass_c128 and "igemm" not in _overrides)
                or wd.numel() != Cin * 9 * Co or Co % 64 or Cin % 8):
            return None
        if codes.dtype != torch.int32 or not codes.is_contiguous() or tuple(codes.shape) != (N, H, W, Cin // 8):
            raise ValueError("conv3x3_dgrad: unpool codes shape %s" % (tuple(codes.shape),))
        full = torch.empty((N, 2 * H, 2 * W, Cin), dtype=BF16, device=dy.device)
        return _igemm(dy, wd, None, None, full, N * H * W, Cin, Co, 9, H, W, Co, 9 * Co, Cin, False, False,
                      1.0, 0.0, 0, want_colsum=True, unpool=(codes, relu_mask))
    if wd.numel() != Cin * 9 * Co or Co % 64 or Cin % 8:
        raise ValueError("conv3x3_dgrad: bad shapes dy=%s wd=%s" % (tuple(dy.shape), tuple(wd.shape)))
    if mask_src is not None:
        _need(mask_src, BF16, "mask_src")
        if mask_src.numel() != N * H * W * Cin:
            raise ValueError("mask_src size")
    if mask_bits is not None:
        if mask_src is None or mask_bits.dtype != torch.uint8 or not mask_bits.is_cuda \
                or mask_bits.numel() * 8 != N * H * W * Cin or not mask_bits.is_contiguous():
            raise ValueError("mask_bits: uint8 CUDA tensor of numel(dx)/8 bytes next to its mask_src")
    if out is None:
        out = torch.empty((N, H, W, Cin), dtype=BF16, device=dy.device)
    if Co == 64 and Cin in (64, 128) and "igemm" not in _overrides:
        return _conv_c64(dy, wd, None, mask_src, out, N, H, W, Cin, False, mask_scale)
    if Co == 128 and Cin == 64 and two_pass_c128 and "igemm" not in _overrides:
        # conv2_1's data gradient: a 64-wide output makes the generic kernel LDS-fill bound
        # (395 TF/s); two register-resident-filter passes over the two halves of the 128 reduction
        # channels, the second accumulating in place, run at the conv_c64 rate instead.  The
        # first pass's partial sums are rounded to bf16 once (one extra rounding of a bf16 result).
        lo, hi = wd_halves if wd_halves is not None else split_dgrad_weight(wd, Cin, Co)
        dy2 = dy.view(-1, Co)
        _conv_c64(dy2, lo, None, None, out, N, H, W, Cin, False, 1.0, ldx=Co)
        return _conv_c64(dy2[:, 64:], hi, None, mask_src, out, N, H, W, Cin, False, mask_scale, ldx=Co, acc_src=out)
    # a masked data-gradient is the dY of the convolution below: leave its column sums behind
    return _igemm(dy, wd, None, mask_src, out, N * H * W, Cin, Co, 9, H, W, Co, 9 * Co, Cin, False, False,
                  mask_scale, 0.0, 0, want_colsum=mask_src is not None, mask_bits=mask_bits,
                  bn_bwd_src=bn_bwd_src if mask_src is None else None)


def linear_fwd(x, w, bias=None, relu=False, drop_p=0.0, seed=0, out_fp32=False, out=None, seed_ptr=None):
    """y = x @ w.T (+bias)(relu)(dropout); x [M,K] bf16, w [Nout,K] bf16."""
    _need(x, BF16, "x")
    _need(w, BF16, "w")
    M, K = x.shape
    Nout = w.shape[0]
    if w.shape[1] != K or K % 64:
        raise ValueError("linear_fwd: bad shapes x=%s w=%s" % (tuple(x.shape), tuple(w.shape)))
    if bias is not None:
        _need(bias, torch.float32, "bias")
    if not out_fp32 and Nout % 8:
        raise ValueError("bf16 output needs out_features % 8 == 0 (use out_fp32)")
    if out is None:
        out = torch.empty((M, Nout), dtype=torch.float32 if out_fp32 else BF16, device=x.device)
    if seed_ptr is not None and (seed_ptr.dtype != torch.int64 or not seed_ptr.is_cuda or seed_ptr.numel() < 1):
        raise ValueError("seed_ptr must be a CUDA int64 tensor")
    return _igemm(x, w, bias, None, out, M, Nout, K, 1, 1, 1, K, K, Nout, out_fp32, relu, 1.0, drop_p, seed,
                  seed_ptr=seed_ptr)


def quant_fp8(x, e5m2=False):
    """Per-tensor 8-bit float quantisation on the device — OCP e4m3 (activations, weights) or
    e5m2 (``e5m2=True``: gradients) — returns (q uint8 tensor shaped like x, scale f32[1]) with
    x ~= q * scale; amax, scale and conversion never touch the host."""
    if not x.is_cuda or not x.is_contiguous() or x.dtype not in (torch.float32, BF16) or x.numel() % 4:
        raise ValueError("quant_fp8: contiguous CUDA f32/bf16 tensor with numel % 4 == 0")
    q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = workspace.get("amax", 1024, torch.float32, x.device)
    check(_lib.hip().dtp_quant_fp8(ptr(x), int(x.dtype == BF16) | (2 if e5m2 else 0), x.numel(), ptr(q), ptr(scale), ptr(ws), stream()),
          "quant_fp8")
    return q, scale


def linear_fwd_fp8(xq, x_scale, wq, w_scale, bias=None, relu=False, drop_p=0.0, seed=0, out_fp32=False,
                   seed_ptr=None):
    """y = (xq @ wq.T) * x_scale * w_scale (+bias)(relu)(dropout) on v_mfma_f32_16x16x32_fp8_fp8;
    xq [M,K] / wq [Nout,K] uint8 holding e4m3, K % 128 == 0, f32 accumulation."""
    for t, name in ((xq, "xq"), (wq, "wq")):
        _need(t, torch.uint8, name)
    M, K = xq.shape
    Nout = wq.shape[0]
    if wq.shape[1] != K or K % 128:
        raise ValueError("linear_fwd_fp8: bad shapes xq=%s wq=%s" % (tuple(xq.shape), tuple(wq.shape)))
    if bias is not None:
        _need(bias, torch.float32, "bias")
    if not out_fp32 and Nout % 8:
        raise ValueError("bf16 output needs out_features % 8 == 0 (use out_fp32)")
    out = torch.empty((M, Nout), dtype=torch.float32 if out_fp32 else BF16, device=xq.device)
    return _igemm(xq, wq, bias, None, out, M, Nout, K, 1, 1, 1, K, K, Nout, out_fp32, relu, 1.0, drop_p, seed,
                  seed_ptr=seed_ptr, fp8_scales=(x_scale, w_scale))


def linear_dgrad_fp8(dyq, dy_scale, wtq, wt_scale, mask_src=None, mask_scale=1.0):
    """dx = (dyq @ wtq.T) * dy_scale * wt_scale [* (mask_src > 0) * mask_scale] on
    v_mfma_f32_16x16x32_fp8_bf8: dyq [M, ld] uint8 holding e5m2 gradients, wtq [K_in, ld] uint8
    holding the e4m3 TRANSPOSED weight (``quant_fp8(W.T)``), ld % 128 == 0, f32 accumulation, bf16 out."""
    for t, name in ((dyq, "dyq"), (wtq, "wtq")):
        _need(t, torch.uint8, name)
    M, ld = dyq.shape
    Kin = wtq.shape[0]
    if wtq.shape[1] != ld or ld % 128 or Kin % 8:
        raise ValueError("linear_dgrad_fp8: bad shapes dyq=%s wtq=%s" % (tuple(dyq.shape), tuple(wtq.shape)))
    if mask_src is not None:
        _need(mask_src, BF16, "mask_src")
        if tuple(mask_src.shape) != (M, Kin):
            raise ValueError("mask_src shape")
    out = torch.empty((M, Kin), dtype=BF16, device=dyq.device)
    return _igemm(dyq, wtq, None, mask_src, out, M, Kin, ld, 1, 1, 1, ld, ld, Kin, False, False, mask_scale, 0.0, 0,
                  fp8_scales=(dy_scale, wt_scale), fp8_mode=2)


def linear_dgrad(dy, wt, ld_dy=None, mask_src=None, mask_scale=1.0, out=None):
    """dx = dy @ W with W given transposed (wt = W.T as [K_in, N_out_padded]); dy [M, ld] bf16."""
    _need(dy, BF16, "dy")
    _need(wt, BF16, "wt")
    M, ld = dy.shape
    Kin, Nred = wt.shape
    if Nred != ld or ld % 64 or Kin % 8:
        raise ValueError("linear_dgrad: bad shapes dy=%s wt=%s" % (tuple(dy.shape), tuple(wt.shape)))
    if mask_src is not None:
        _need(mask_src, BF16, "mask_src")
        if tuple(mask_src.shape) != (M, Kin):
            raise ValueError("mask_src shape")
    if out is None:
        out = torch.empty((M, Kin), dtype=BF16, device=dy.device)
    return _igemm(dy, wt, None, mask_src, out, M, Kin, ld, 1, 1, 1, ld, ld, Kin, False, False, mask_scale, 0.0, 0)


def _wgrad(dy2d, x2d, out, K, M, Ntot, C, taps, H, W, lda, ldb, beta):
    ksteps = _cdiv(K, 64)
    cfg, nsplit = pick_wgrad(M, C if taps == 9 else Ntot, taps, ksteps)
    partial = None
    if nsplit > 1:
        partial = workspace.get("wgrad_partial", nsplit * M * Ntot, torch.float32, dy2d.device)
    out_fp32 = out.dtype == torch.float32
    rc = _lib.hip().dtp_wgrad_tt(ptr(dy2d), ptr(x2d), ptr(out), ptr(partial), ptr(workspace.zero_page(dy2d.device)),
                                 K, M, Ntot, C, taps, H, W, lda, ldb, int(out_fp32), float(beta), nsplit, cfg,
                                 stream())
    check(rc, "wgrad_tt")
    return out


def conv3x3_wgrad(dy, x, out, beta=0.0):
    """out[Cout, 9*C] (f32 or bf16, row-major) = beta*out + sum_pixels dy (x) shifted x."""
    _need(dy, BF16, "dy")
    _need(x, BF16, "x")
    N, H, W, Co = dy.shape
    C = x.shape[3]
    if tuple(x.shape[:3]) != (N, H, W) or C % 64 or Co % 8:
        raise ValueError("conv3x3_wgrad: bad shapes dy=%s x=%s" % (tuple(dy.shape), tuple(x.shape)))
    if out.numel() != Co * 9 * C or not out.is_contiguous() or out.dtype not in (torch.float32, BF16):
        raise ValueError("conv3x3_wgrad: bad out")
    if C == 64 and Co in (64, 128) and "wgrad" not in _overrides:
        # all-nine-taps column-strip kernel: activations are fetched ~once instead of 9x
        max_grid = 2 * N_CU if Co == 64 else N_CU
        partial = workspace.get("wgrad_partial", max_grid * Co * 576, torch.float32, dy.device)
        rc = _lib.hip().dtp_wgrad_c64(ptr(dy), ptr(x), ptr(out), ptr(partial), max_grid,
                                      ptr(workspace.zero_page(dy.device)), N, H, W, Co,
                                      int(out.dtype == torch.float32), float(beta), stream())
        check(rc, "wgrad_c64")
        return out
    return _wgrad(dy, x, out, N * H * W, Co, 9 * C, C, 9, H, W, Co, C, beta)


def linear_wgrad(dy, x, out, n_valid=None, beta=0.0):
    """out[Nout, K] = beta*out + dy[:, :Nout].T @ x ; dy [M, ld>=Nout] bf16, x [M, K] bf16."""
    _need(dy, BF16, "dy")
    _need(x, BF16, "x")
    M, ld = dy.shape
    K = x.shape[1]
    Nout = ld if n_valid is None else n_valid
    if x.shape[0] != M or ld % 8 or K % 8 or Nout > ld:
        raise ValueError("linear_wgrad: bad shapes dy=%s x=%s" % (tuple(dy.shape), tuple(x.shape)))
    if out.numel() != Nout * K or not out.is_contiguous() or out.dtype not in (torch.float32, BF16):
        raise ValueError("linear_wgrad: bad out")
    return _wgrad(dy, x, out, M, Nout, K, K, 1, 1, 1, ld, K, beta)


def colsum(x2d, out, n_valid=None, beta=0.0, ws_key="colsum"):
    """out[:n_valid] = beta*out + column sums of x2d [rows, C] (bias gradient)."""
    _need(x2d, BF16, "x2d")
    rows, C = x2d.shape
    nv = C if n_valid is None else n_valid
    if C % 8 or out.numel() != nv or out.dtype not in (torch.float32, BF16):
        raise ValueError("colsum: bad shapes")
    ws = workspace.get(ws_key, 1024 * max(C, 64), torch.float32, x2d.device)
    rc = _lib.hip().dtp_colsum(ptr(x2d), ptr(out), ptr(ws), ws.numel(), rows, C, nv, C,
                               int(out.dtype == torch.float32), float(beta), stream())
    check(rc, "colsum")
    return out


def maxpool2x2_fwd(x, want_idx=False):
    """y = maxpool 2x2/s2 of NHWC bf16 x.  ``want_idx``: also return the int32 tensor of 4-bit
    (first-argmax, max > 0) codes that lets :func:`maxpool2x2_bwd` run without re-reading x."""
    _need(x, BF16, "x")
    N, H, W, C = x.shape
    if C % 8 or H < 2 or W < 2:
        raise ValueError("maxpool2x2_fwd: need C %% 8 == 0 and H, W >= 2, got %s" % (tuple(x.shape),))
    y = torch.empty((N, H // 2, W // 2, C), dtype=BF16, device=x.device)
    idx = torch.empty((N, H // 2, W // 2, C // 8), dtype=torch.int32, device=x.device) if want_idx else None
    check(_lib.hip().dtp_maxpool2x2_fwd(ptr(x), ptr(y), ptr(idx), N, H, W, C, stream()), "maxpool_fwd")
    return (y, idx) if want_idx else y


def maxpool2x2_bwd(x, dy, relu_mask=False, idx=None, in_shape=None):
    """dx of the 2x2 max-pool (first-max tie rule), optionally times (pool input > 0).  Either the
    pool input ``x`` or the forward pass's ``idx`` codes (+ ``in_shape``) identify the winners."""
    _need(dy, BF16, "dy")
    if idx is None:
        _need(x, BF16, "x")
        N, H, W, C = x.shape
    else:
        N, H, W, C = in_shape
        if idx.dtype != torch.int32 or not idx.is_contiguous() or tuple(idx.shape) != (N, H // 2, W // 2, C // 8):
            raise ValueError("maxpool2x2_bwd: idx shape %s" % (tuple(idx.shape),))
        x = None
    if tuple(dy.shape) != (N, H // 2, W // 2, C):
        raise ValueError("maxpool2x2_bwd: dy shape %s" % (tuple(dy.shape),))
    dx = torch.empty((N, H, W, C), dtype=BF16, device=dy.device)
    cs_part, max_parts = None, 0
    if fuse_colsum and 256 % (C // 8) == 0 and H % 2 == 0 and W % 2 == 0:
        # dx is the dY of the convolution that produced x: leave its column sums behind
        max_parts = 1024
        cs_part = _colsum_parts_workspace(max_parts * C, dy.device)
    nparts = _ctypes.c_int(0)
    check(_lib.hip().dtp_maxpool2x2_bwd(ptr(x), ptr(idx), ptr(dy), ptr(dx), N, H, W, C, int(relu_mask),
                                        ptr(cs_part), max_parts, _ctypes.byref(nparts), stream()), "maxpool_bwd")
    if cs_part is not None and nparts.value > 0:
        colsum_slot.put(dx, cs_part, nparts.value, C)
    return dx


def avgpool_flatten_fwd(x, OH=7, OW=7):
    _need(x, BF16, "x")
    N, H, W, C = x.shape
    y = torch.empty((N, C * OH * OW), dtype=BF16, device=x.device)
    if (H, W) == (OH, OW):       # 224x224 input: the pool is the identity, only the NHWC -> CHW-flatten order changes
        return transpose_bf16(x, y, H * W, C, C, H * W, batch=N, in_bs=H * W * C, out_bs=C * H * W)
    check(_lib.hip().dtp_avgpool_flatten_fwd(ptr(x), ptr(y), N, H, W, C, OH, OW, stream()), "avgpool_fwd")
    return y


def avgpool_flatten_bwd(dy, N, H, W, C, OH=7, OW=7):
    _need(dy, BF16, "dy")
    if dy.numel() != N * C * OH * OW:
        raise ValueError("avgpool_flatten_bwd: dy size")
    dx = torch.empty((N, H, W, C), dtype=BF16, device=dy.device)
    if (H, W) == (OH, OW):       # identity pool: a batched [C][HW] -> [HW][C] transpose (was 104 us of div/mod + 2-byte gathers)
        return transpose_bf16(dy, dx, C, H * W, H * W, C, batch=N, in_bs=C * H * W, out_bs=H * W * C)
    check(_lib.hip().dtp_avgpool_flatten_bwd(ptr(dy), ptr(dx), N, H, W, C, OH, OW, stream()), "avgpool_bwd")
    return dx


def transpose_bf16(inp, out, R, Cc, ld_in, ld_out, batch=1, in_bs=0, out_bs=0, in_off=0, out_off=0):
    """out[b][c][r] = in[b][r][c] on raw strided bf16 storage (element strides/offsets)."""
    rc = _lib.hip().dtp_transpose_bf16(_lib.c_void_p(inp.data_ptr() + 2 * in_off),
                                       _lib.c_void_p(out.data_ptr() + 2 * out_off), R, Cc, ld_in, ld_out, batch,
                                       in_bs, out_bs, stream())
    check(rc, "transpose")
    return out


def conv_weight_to_dgrad(w, Cout, Cin, out=None):
    """wd[ci][t][co] = w[co][8-t][ci]  (rotated, channel-transposed filter for the data gradient)."""
    _need(w, BF16, "w")
    if w.numel() != Cout * 9 * Cin:
        raise ValueError("conv_weight_to_dgrad: size")
    if out is None:
        out = torch.empty((Cin, 9 * Cout), dtype=BF16, device=w.device)
    # batch b = tap t of the input; output tap 8-t  -> negative output batch stride
    return transpose_bf16(w, out, Cout, Cin, 9 * Cin, 9 * Cout, batch=9, in_bs=Cin, out_bs=-Cout,
                          out_off=8 * Cout)


def linear_weight_transpose(w, pad_to=64, out=None):
    """wt [K_in, ceil(N_out, pad_to)] = w.T, zero padded along N_out (reduction axis of dgrad)."""
    _need(w, BF16, "w")
    Nout, K = w.shape
    ld = _cdiv(Nout, pad_to) * pad_to
    if out is None:
        out = (torch.zeros if ld != Nout else torch.empty)((K, ld), dtype=BF16, device=w.device)
    return transpose_bf16(w, out, Nout, K, K, ld)


def _transpose_job(inp, out, R, Cc, ld_in, ld_out, batch=1, in_bs=0, out_bs=0, in_off=0, out_off=0):
    """One row of a :class:`TransposePlan` table (the arguments of :func:`transpose_bf16`)."""
    return [inp.data_ptr() + 2 * in_off, out.data_ptr() + 2 * out_off, ld_in, ld_out, in_bs, out_bs, R, Cc,
            _cdiv(Cc, 64), _cdiv(R, 64), batch, 0]


def conv_weight_to_dgrad_job(w, Cout, Cin, out):
    """:func:`conv_weight_to_dgrad` as a table row: ``out`` [Cin, 9*Cout] is a persistent buffer."""
    _need(w, BF16, "w")
    _need(out, BF16, "out")
    if w.numel() != Cout * 9 * Cin or out.numel() != Cout * 9 * Cin:
        raise ValueError("conv_weight_to_dgrad_job: size")
    return _transpose_job(w, out, Cout, Cin, 9 * Cin, 9 * Cout, batch=9, in_bs=Cin, out_bs=-Cout, out_off=8 * Cout)


def linear_weight_transpose_job(w, out):
    """:func:`linear_weight_transpose` as a table row: ``out`` [K_in, ld >= N_out], padding columns zeroed once."""
    _need(w, BF16, "w")
    _need(out, BF16, "out")
    Nout, Kin = w.shape
    if out.shape[0] != Kin or out.shape[1] < Nout:
        raise ValueError("linear_weight_transpose_job: size")
    return _transpose_job(w, out, Nout, Kin, Kin, out.shape[1])


class TransposePlan(object):
    """Many bf16 transposes as ONE launch (csrc/eltwise.hip ``transpose_multi_kernel``): the job table
    lives in device memory, sources and destinations are persistent buffers, so the launch has no
    per-step arguments (hipGraph-friendly) and replaces ~15 launch-sized kernels per optimizer step."""

    def __init__(self, jobs, device):
        if not jobs:
            raise ValueError("TransposePlan: no jobs")
        first = 0
        rows = []
        for j in jobs:
            j = list(j)
            j[11] = first
            first += j[8] * j[9] * j[10]
            rows.append(j)
        self.n, self.total = len(rows), first
        self.table = torch.tensor(rows, dtype=torch.int64).reshape(-1).to(device)

    def run(self):
        check(_lib.hip().dtp_transpose_multi(ptr(self.table), self.n, self.total, stream()), "transpose_multi")


def nchw_to_nhwc_bf16(x, c_pad, scale=1.0, mean=None, inv_std=None):
    kinds = {torch.float32: 0, BF16: 1, torch.uint8: 2}
    if x.dtype not in kinds or not x.is_cuda or not x.is_contiguous() or x.dim() != 4:
        raise ValueError("nchw_to_nhwc_bf16: need contiguous CUDA NCHW f32/bf16/u8")
    N, Cin, H, W = x.shape
    out = torch.empty((N, H, W, c_pad), dtype=BF16, device=x.device)
    rc = _lib.hip().dtp_nchw_to_nhwc_bf16(ptr(x), kinds[x.dtype], ptr(out), N, Cin, H, W, c_pad, float(scale),
                                          ptr(mean), ptr(inv_std), stream())
    check(rc, "nchw_to_nhwc")
    return out


_IN_KINDS = {torch.float32: 0, BF16: 1, torch.uint8: 2}


def conv1_fprop(x, w, bias, relu=True, scale=1.0, mean=None, inv_std=None, want_bn_stats=False):
    """First layer: NCHW f32/bf16/u8 [N,3,H,W] -> NHWC bf16 [N,H,W,64] = relu(conv3x3(norm(x)) + b)."""
    if x.dtype not in _IN_KINDS or not x.is_cuda or not x.is_contiguous() or x.dim() != 4 or x.shape[1] != 3:
        raise ValueError("conv1_fprop: need contiguous CUDA NCHW f32/bf16/u8 with 3 channels")
    _need(w, BF16, "w")
    if w.numel() != 64 * 27:
        raise ValueError("conv1_fprop: weight must be [64, 27]")
    if bias is not None:
        _need(bias, torch.float32, "bias")
    N, Cin, H, W = x.shape
    y = torch.empty((N, H, W, 64), dtype=BF16, device=x.device)
    part, rows, nparts = None, 0, _ctypes.c_int(0)
    if want_bn_stats and fuse_bn_stats:
        rows = 1024                                    # persistent grid: <= 768 workgroups per launch
        part = workspace.get("bn_tile_stats", rows * 2 * 64, torch.float32, x.device)
    rc = _lib.hip().dtp_conv1_fprop(ptr(x), _IN_KINDS[x.dtype], ptr(w), ptr(bias), ptr(y), N, Cin, H, W, int(relu),
                                    float(scale), ptr(mean), ptr(inv_std), ptr(part), rows, _ctypes.byref(nparts),
                                    stream())
    if rc == -15 and part is not None:                 # more batch chunks than partial rows: no hand-over
        return conv1_fprop(x, w, bias, relu, scale, mean, inv_std)
    check(rc, "conv1_fprop")
    if part is not None:
        bn_stats_slot.put(y, part, nparts.value, 64)
    return y


def conv1_wgrad(x, dy, out_w, out_b, beta_w=0.0, beta_b=0.0, scale=1.0, mean=None, inv_std=None,
                mask_w=None, mask_bias=None):
    """out_w[64, 27] and out_b[64] (same dtype, f32 or bf16) from dy NHWC bf16 [N,H,W,64] and the raw input.
    ``mask_w`` ([64, 27] bf16 forward weights, + ``mask_bias``): dy is NOT yet masked by this layer's
    ReLU; the kernel recomputes the forward sign per tile and masks dy itself."""
    if x.dtype not in _IN_KINDS or not x.is_cuda or not x.is_contiguous() or x.dim() != 4 or x.shape[1] != 3:
        raise ValueError("conv1_wgrad: bad x")
    _need(dy, BF16, "dy")
    N, Cin, H, W = x.shape
    if tuple(dy.shape) != (N, H, W, 64) or out_w.numel() != 64 * 27 or out_b.numel() != 64:
        raise ValueError("conv1_wgrad: bad shapes")
    if out_w.dtype != out_b.dtype or out_w.dtype not in (torch.float32, BF16):
        raise ValueError("conv1_wgrad: bad output dtype")
    if mask_w is not None:
        _need(mask_w, BF16, "mask_w")
        if mask_w.numel() != 64 * 27:
            raise ValueError("conv1_wgrad: mask_w must be the [64, 27] forward weights")
        if mask_bias is not None:
            _need(mask_bias, torch.float32, "mask_bias")
    max_grid = 3 * N_CU
    partial = workspace.get("conv1_partial", max_grid * 2048, torch.float32, x.device)
    rc = _lib.hip().dtp_conv1_wgrad(ptr(x), _IN_KINDS[x.dtype], ptr(dy), ptr(out_w), ptr(out_b), ptr(partial),
                                    max_grid, ptr(workspace.zero_page(x.device)), N, Cin, H, W, float(scale),
                                    ptr(mean), ptr(inv_std), int(out_w.dtype == torch.float32), float(beta_w),
                                    float(beta_b), ptr(mask_w), ptr(mask_bias) if mask_w is not None else None,
                                    stream())
    check(rc, "conv1_wgrad")


def _bn_check(y, C, *vecs):
    _need(y, BF16, "y")
    if y.shape[-1] != C or C < 64 or C > 512 or C & (C - 1):
        raise ValueError("batch norm: channel count must be a power of two in [64, 512], got %s" % (tuple(y.shape),))
    for v in vecs:
        if v is None:
            continue
        if not v.is_cuda or v.dtype != torch.float32 or not v.is_contiguous() or v.numel() != C:
            raise ValueError("batch norm: per-channel vectors must be contiguous CUDA f32 [C]")


def bn_fwd_train(y, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1, relu=True):
    """z = (relu) batch_norm(y) over the rows of NHWC bf16 y [..., C] with the batch statistics;
    running_mean / running_var are updated in place (unbiased variance, like torch).  Returns
    (z, stats) with stats = [mean | rstd | scale | shift] (f32 [4*C]) for the backward pass."""
    C = y.shape[-1]
    _bn_check(y, C, gamma, beta, running_mean, running_var)
    M = y.numel() // C
    z = torch.empty_like(y)
    stats = torch.empty(4 * C, dtype=torch.float32, device=y.device)
    part = workspace.get("bn_partial", 1024 * 2 * 512, torch.float32, y.device)
    got = bn_stats_slot.take(y, C) if fuse_bn_stats else None
    check(_lib.hip().dtp_bn_fwd_train(ptr(y), ptr(z), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                      ptr(stats), ptr(part), M, C, float(eps), float(momentum), int(relu),
                                      ptr(got[0]) if got else None, got[1] if got else 0, stream()),
          "bn_fwd_train")
    return z, stats


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps=1e-5):
    """(scale, shift) of eval-mode batch norm: y*scale + shift (for folding into a convolution)."""
    C = gamma.numel()
    co = torch.empty(2 * C, dtype=torch.float32, device=gamma.device)
    for v in (gamma, beta, running_mean, running_var):
        if not v.is_cuda or v.dtype != torch.float32 or not v.is_contiguous() or v.numel() != C:
            raise ValueError("bn_eval_coeffs: contiguous CUDA f32 [C] vectors")
    check(_lib.hip().dtp_bn_fwd_eval(None, None, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(co),
                                     1, C, float(eps), 0, stream()),
          "bn_eval_coeffs")
    return co[:C], co[C:]


def bn_fwd_eval(y, gamma, beta, running_mean, running_var, eps=1e-5, relu=True):
    C = y.shape[-1]
    _bn_check(y, C, gamma, beta, running_mean, running_var)
    z = torch.empty_like(y)
    co = torch.empty(2 * C, dtype=torch.float32, device=y.device)
    check(_lib.hip().dtp_bn_fwd_eval(ptr(y), ptr(z), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(co),
                                     y.numel() // C, C, float(eps), int(relu), stream()), "bn_fwd_eval")
    return z


def bn_bwd(y, dz, gamma, stats, dgamma, dbeta, beta_g=0.0, beta_b=0.0, relu=True):
    """dy of (relu) batch_norm given the forward's input y and ``stats``; dgamma / dbeta ([C], f32
    or bf16, same dtype) are written (beta_* = 0) or accumulated into (beta_* = 1)."""
    C = y.shape[-1]
    _bn_check(y, C, gamma)
    _need(dz, BF16, "dz")
    if dz.numel() != y.numel() or stats.numel() != 4 * C or stats.dtype != torch.float32:
        raise ValueError("bn_bwd: bad shapes")
    if dgamma.dtype != dbeta.dtype or dgamma.dtype not in (torch.float32, BF16) or dgamma.numel() != C \
            or dbeta.numel() != C or not dgamma.is_contiguous() or not dbeta.is_contiguous():
        raise ValueError("bn_bwd: dgamma / dbeta must be contiguous [C] of one dtype (f32 / bf16)")
    dy = torch.empty_like(y)
    part = workspace.get("bn_partial", 1024 * 2 * 512, torch.float32, y.device)
    work = workspace.get("bn_work", 2 * 512, torch.float32, y.device)
    got = bn_bwd_slot.take(dz, C) if fuse_bn_stats else None
    check(_lib.hip().dtp_bn_bwd(ptr(y), ptr(dz), ptr(dy), ptr(gamma), ptr(stats), ptr(dgamma), ptr(dbeta),
                                int(dgamma.dtype == torch.float32), float(beta_g), float(beta_b), ptr(part), ptr(work),
                                y.numel() // C, C, int(relu), ptr(got[0]) if got else None, got[1] if got else 0,
                                stream()), "bn_bwd")
    return dy


def bn_pool_fusable(y):
    """BatchNorm + ReLU + 2x2 max-pool as one kernel pair: NHWC bf16, even H and W, C a power of two in 64..512."""
    if y.dim() != 4 or not y.is_cuda or y.dtype != BF16:
        return False
    N, H, W, C = y.shape
    return H >= 2 and W >= 2 and H % 2 == 0 and W % 2 == 0 and 64 <= C <= 512 and (C & (C - 1)) == 0


def bn_fwd_train_pool(y, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1):
    """maxpool2x2(relu(batch_norm(y))) with the batch statistics for NHWC bf16 y [N, H, W, C]: the
    normalised full-resolution activation is never written.  Returns (z_pooled, codes, stats);
    codes are :func:`maxpool2x2_fwd`'s 4-bit (first argmax, max > 0) codes."""
    if not bn_pool_fusable(y):
        raise ValueError("bn_fwd_train_pool: shape %s" % (tuple(y.shape),))
    N, H, W, C = y.shape
    _bn_check(y, C, gamma, beta, running_mean, running_var)
    zp = torch.empty((N, H // 2, W // 2, C), dtype=BF16, device=y.device)
    codes = torch.empty((N, H // 2, W // 2, C // 8), dtype=torch.int32, device=y.device)
    stats = torch.empty(4 * C, dtype=torch.float32, device=y.device)
    part = workspace.get("bn_partial", 1024 * 2 * 512, torch.float32, y.device)
    got = bn_stats_slot.take(y, C) if fuse_bn_stats else None
    check(_lib.hip().dtp_bn_fwd_train_pool(ptr(y), ptr(zp), ptr(codes), ptr(gamma), ptr(beta), ptr(running_mean),
                                           ptr(running_var), ptr(stats), ptr(part), N, H, W, C, float(eps),
                                           float(momentum), ptr(got[0]) if got else None, got[1] if got else 0,
                                           stream()), "bn_fwd_train_pool")
    return zp, codes, stats


def bn_bwd_pool(y, dzp, codes, gamma, stats, dgamma, dbeta, beta_g=0.0, beta_b=0.0):
    """dy of :func:`bn_fwd_train_pool` from the POOLED incoming gradient: the full-resolution
    gradient of the normalised activation is never materialised."""
    if not bn_pool_fusable(y):
        raise ValueError("bn_bwd_pool: shape %s" % (tuple(y.shape),))
    N, H, W, C = y.shape
    _bn_check(y, C, gamma)
    _need(dzp, BF16, "dzp")
    if tuple(dzp.shape) != (N, H // 2, W // 2, C) or tuple(codes.shape) != (N, H // 2, W // 2, C // 8) \
            or codes.dtype != torch.int32 or not codes.is_contiguous() or stats.numel() != 4 * C \
            or stats.dtype != torch.float32:
        raise ValueError("bn_bwd_pool: bad shapes")
    if dgamma.dtype != dbeta.dtype or dgamma.dtype not in (torch.float32, BF16) or dgamma.numel() != C \
            or dbeta.numel() != C or not dgamma.is_contiguous() or not dbeta.is_contiguous():
        raise ValueError("bn_bwd_pool: dgamma / dbeta must be contiguous [C] of one dtype (f32 / bf16)")
    dy = torch.empty_like(y)
    part = workspace.get("bn_partial", 1024 * 2 * 512, torch.float32, y.device)
    work = workspace.get("bn_work", 2 * 512, torch.float32, y.device)
    check(_lib.hip().dtp_bn_bwd_pool(ptr(y), ptr(dzp), ptr(codes), ptr(dy), ptr(gamma), ptr(stats), ptr(dgamma),
                                     ptr(dbeta), int(dgamma.dtype == torch.float32), float(beta_g), float(beta_b),
                                     ptr(part), ptr(work), N, H, W, C, stream()), "bn_bwd_pool")
    return dy


def cast_f32_bf16(src, dst):
    _need(src, torch.float32, "src")
    _need(dst, BF16, "dst")
    if src.numel() != dst.numel():
        raise ValueError("cast size mismatch")
    check(_lib.hip().dtp_cast_f32_bf16(ptr(src), ptr(dst), src.numel(), stream()), "cast")
    return dst


# head start (microseconds) for wgrad_c64's persistent grid before the data gradient of the same layer
# is dispatched on the other stream (csrc/eltwise.hip stream_delay_kernel); 0 = off
wgrad_head_start_us = float(_os.environ.get("DTP_WGRAD_HEAD_START_US", "20"))
# ... only where the kernels are long enough for grid residency to matter: at per-GPU batch 4 the two 20 us
# delays were 44 us of a 2.3 ms step (profiles/73_*) for kernels that take 30 us themselves
WGRAD_HEAD_START_MIN_ELEMS = 1 << 25


def wgrad_is_persistent(C, Co):
    """True when ``conv3x3_wgrad`` takes the persistent-grid kernel (wgrad_c64)."""
    return C == 64 and Co in (64, 128) and "wgrad" not in _overrides


def stream_delay(us):
    """Keep the current stream busy with nothing for ``us`` microseconds (one idle wave)."""
    if us > 0:
        check(_lib.hip().dtp_stream_delay(float(us), stream()), "stream_delay")


def relu_mask(g, y, scale=1.0, out=None):
    _need(g, BF16, "g")
    _need(y, BF16, "y")
    if g.numel() != y.numel() or g.numel() % 8:
        raise ValueError("relu_mask: size")
    if out is None:
        out = torch.empty_like(g)
    check(_lib.hip().dtp_relu_mask(ptr(g), ptr(y), ptr(out), g.numel(), float(scale), stream()), "relu_mask")
    return out


def softmax_xent(logits, labels, grad_scale=None, want_bf16_grad=True, bf16_ld=None):
    """Returns (out2 = [mean loss, #hits] f32, dlogits_f32 [N,C], dlogits_bf16 [N,ld] or None)."""
    _need(logits, torch.float32, "logits")
    if labels.dtype != torch.int64 or not labels.is_cuda or not labels.is_contiguous():
        raise ValueError("labels must be contiguous CUDA int64")
    N, C = logits.shape
    if labels.numel() != N:
        raise ValueError("labels size")
    if grad_scale is None:
        grad_scale = 1.0 / N
    ld = bf16_ld if bf16_ld is not None else _cdiv(C, 64) * 64
    row_ws = torch.empty(2 * N, dtype=torch.float32, device=logits.device)
    out2 = torch.empty(2, dtype=torch.float32, device=logits.device)
    dl32 = torch.empty((N, C), dtype=torch.float32, device=logits.device)
    dl16 = torch.empty((N, ld), dtype=BF16, device=logits.device) if want_bf16_grad else None
    rc = _lib.hip().dtp_softmax_xent(ptr(logits), ptr(labels), ptr(row_ws), ptr(out2), ptr(dl32), ptr(dl16), N, C,
                                     logits.stride(0), ld, float(grad_scale), 1.0 / N, stream())
    check(rc, "softmax_xent")
    return out2, dl32, dl16


def sgd_flat(p, g, mom, lp, lr, momentum, dampening, wd, grad_scale, nesterov, first_step):
    n = p.numel()
    rc = _lib.hip().dtp_sgd_flat(ptr(p), ptr(g), int(g.dtype == BF16), ptr(mom), ptr(lp), n, float(lr),
                                 float(momentum), float(dampening), float(wd), float(grad_scale), int(nesterov),
                                 int(first_step), stream())
    check(rc, "sgd_flat")


def adamw_flat(p, g, m, v, lp, lr, beta1, beta2, eps, wd, grad_scale, step):
    n = p.numel()
    rc = _lib.hip().dtp_adamw_flat(ptr(p), ptr(g), int(g.dtype == BF16), ptr(m), ptr(v), ptr(lp), n, float(lr),
                                   float(beta1), float(beta2), float(eps), float(wd), float(grad_scale), int(step),
                                   stream())
    check(rc, "adamw_flat")
