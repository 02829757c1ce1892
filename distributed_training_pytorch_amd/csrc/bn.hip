// BatchNorm (+ fused ReLU) over the channel axis of NHWC bf16 activations [M = N*H*W][C] for gfx950
// (VGG16-BN, BASELINE.json config 5).  Memory-bound streaming kernels, 16 bytes (8 channels) per
// lane, deterministic two-level reductions (no float atomics):
//
//   forward, training:  bn_stats  (sum, sum of squares per channel -> per-block partials)
//                       bn_finalize (mean, biased var, rstd; scale = gamma*rstd, shift = beta - mean*scale;
//                                    running_mean / running_var momentum update, unbiased var)
//                       bn_apply  (z = relu(y*scale + shift), bf16)
//   forward, eval:      bn_apply with scale/shift from the running statistics (or folded into the
//                       convolution's weights by ops/layers.py)
//   backward:           bn_bwd_reduce (dbeta = sum dyh, dgamma = sum dyh*xhat, dyh = dz*(z > 0))
//                       bn_bwd_finalize (gradients into the arena, beta accumulate; c1 = dbeta/M, c2 = dgamma/M)
//                       bn_bwd_apply (dy = gamma*rstd*(dyh - c1 - xhat*c2), bf16)
//
// Reference parity: the reference model has no BatchNorm (model/vgg16.py:5-17); BASELINE.json
// config 5 asks for the VGG16-BN variant, which stock PyTorch runs on MIOpen batch-norm kernels.
#include "dtp_common.h"

namespace {

constexpr int BN_T = 256;          // threads per block
constexpr int BN_MAXBLK = 1024;    // partial rows per reduction

DTP_DEVINL void unpack8(const u32x4_t v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __uint_as_float(v[e] << 16);
    f[2 * e + 1] = __uint_as_float(v[e] & 0xFFFF0000u);
  }
}

// Two per-channel sums over the rows of this block's share of [M][C]:
//   MODE 0: (sum y, sum y^2)
//   MODE 1: (sum dyh, sum dyh*xhat) with dyh = dz * (relu ? y*scale+shift > 0 : 1), xhat = (y-mean)*rstd
// partial layout: [gridDim.x][2][C]
template <int MODE>
__global__ __launch_bounds__(BN_T) void bn_reduce_kernel(const u32x4_t* __restrict__ y, const u32x4_t* __restrict__ dz,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ partial, long M, int C, int relu) {
  __shared__ float red[2][2048];
  const int cg = C >> 3;                       // 16-byte groups per row (8..64)
  const int R = BN_T / cg;                     // rows per block iteration
  const int g = threadIdx.x % cg, tr = threadIdx.x / cg;
  float a0[8], a1[8], sc[8], sh[8], mu[8], rs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a0[e] = a1[e] = 0.f;
    if (MODE == 1) {
      sc[e] = scale[g * 8 + e]; sh[e] = shift[g * 8 + e];
      mu[e] = mean[g * 8 + e]; rs[e] = rstd[g * 8 + e];
    }
  }
  for (long r = (long)blockIdx.x * R + tr; r < M; r += (long)gridDim.x * R) {
    float f[8];
    unpack8(y[r * cg + g], f);
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { a0[e] += f[e]; a1[e] = fmaf(f[e], f[e], a1[e]); }
    } else {
      float d[8];
      unpack8(dz[r * cg + g], d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dyh = (!relu || fmaf(f[e], sc[e], sh[e]) > 0.f) ? d[e] : 0.f;
        a0[e] += dyh;
        a1[e] = fmaf(dyh, (f[e] - mu[e]) * rs[e], a1[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][tr * C + g * 8 + e] = a0[e];
    red[1][tr * C + g * 8 + e] = a1[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += BN_T) {
    const int which = c / C, ch = c - which * C;
    float s = 0.f;
    for (int k = 0; k < R; ++k) s += red[which][k * C + ch];
    partial[((size_t)blockIdx.x * 2 + which) * C + ch] = s;
  }
}

// Fold of the per-block partial rows, 16 channels per 256-thread block: thread (part, ch) sums rows
// part, part + 16, ... in double, the 16 part sums are combined through LDS in a fixed order
// (deterministic).  One thread per channel walking up to 1024 rows serially took 260-275 us per call
// — 7 ms per VGG16-BN step of pure load latency.
constexpr int BN_FC = 16;       // channels per finalize block
DTP_DEVINL bool bn_fold(const float* __restrict__ partial, int nblk, int C, int c, int part, double (*red)[16][BN_FC],
                        double& s, double& q) {
  double a = 0.0, b = 0.0;
  if (c < C) {
    for (int r = part; r < nblk; r += 16) {
      a += (double)partial[((size_t)r * 2) * C + c];
      b += (double)partial[((size_t)r * 2 + 1) * C + c];
    }
  }
  red[0][part][threadIdx.x & (BN_FC - 1)] = a;
  red[1][part][threadIdx.x & (BN_FC - 1)] = b;
  __syncthreads();
  if (part != 0 || c >= C) return false;
  s = 0.0; q = 0.0;
  for (int k = 0; k < 16; ++k) { s += red[0][k][threadIdx.x]; q += red[1][k][threadIdx.x]; }
  return true;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, long M, float eps, float momentum,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ running_mean,
                                   float* __restrict__ running_var) {
  __shared__ double red[2][16][BN_FC];
  const int c = blockIdx.x * BN_FC + (threadIdx.x & (BN_FC - 1));
  double s, q;
  if (!bn_fold(partial, nblk, C, c, threadIdx.x / BN_FC, red, s, q)) return;
  const double m = s / (double)M;
  double var = q / (double)M - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)m;
  rstd[c] = r;
  const float sc = gamma[c] * r;
  scale[c] = sc;
  shift[c] = beta[c] - (float)m * sc;
  if (running_mean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// eval: scale / shift from the running statistics
__global__ void bn_eval_coeffs_kernel(int C, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] * rsqrtf(running_var[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - running_mean[c] * sc;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, long M,
                                       void* __restrict__ dgamma, void* __restrict__ dbeta, int out_fp32,
                                       float beta_g, float beta_b, float* __restrict__ c1, float* __restrict__ c2) {
  __shared__ double red[2][16][BN_FC];
  const int c = blockIdx.x * BN_FC + (threadIdx.x & (BN_FC - 1));
  double s, q;
  if (!bn_fold(partial, nblk, C, c, threadIdx.x / BN_FC, red, s, q)) return;
  c1[c] = (float)(s / (double)M);
  c2[c] = (float)(q / (double)M);
  if (out_fp32) {
    float* dg = (float*)dgamma; float* db = (float*)dbeta;
    dg[c] = (beta_g != 0.f ? beta_g * dg[c] : 0.f) + (float)q;
    db[c] = (beta_b != 0.f ? beta_b * db[c] : 0.f) + (float)s;
  } else {
    bf16_t* dg = (bf16_t*)dgamma; bf16_t* db = (bf16_t*)dbeta;
    dg[c] = f32_to_bf16((beta_g != 0.f ? beta_g * bf16_to_f32(dg[c]) : 0.f) + (float)q);
    db[c] = f32_to_bf16((beta_b != 0.f ? beta_b * bf16_to_f32(db[c]) : 0.f) + (float)s);
  }
}

// MODE 0: z = (relu) y*scale + shift
// MODE 1: dy = g*(dyh - c1 - xhat*c2), g = gamma*rstd (passed as `scale2`), dyh = dz*(z > 0)
template <int MODE>
__global__ __launch_bounds__(BN_T) void bn_apply_kernel(const u32x4_t* __restrict__ y, const u32x4_t* __restrict__ dz,
                                                        u32x4_t* __restrict__ out, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ c1,
                                                        const float* __restrict__ c2, const float* __restrict__ gamma,
                                                        long nvec, int C, int relu) {
  const int cg = C >> 3;
  const int g = threadIdx.x % cg;            // BN_T and the grid stride are multiples of cg: fixed channel group
  float sc[8], sh[8], mu[8], rs[8], k1[8], k2[8], gs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = scale[g * 8 + e]; sh[e] = shift[g * 8 + e];
    if (MODE == 1) {
      mu[e] = mean[g * 8 + e]; rs[e] = rstd[g * 8 + e];
      k1[e] = c1[g * 8 + e]; k2[e] = c2[g * 8 + e];
      gs[e] = gamma[g * 8 + e] * rs[e];
    }
  }
  for (long i = (long)blockIdx.x * BN_T + threadIdx.x; i < nvec; i += (long)gridDim.x * BN_T) {
    float f[8], o[8];
    unpack8(y[i], f);
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = fmaf(f[e], sc[e], sh[e]);
        o[e] = relu ? fmaxf(z, 0.f) : z;
      }
    } else {
      float d[8];
      unpack8(dz[i], d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dyh = (!relu || fmaf(f[e], sc[e], sh[e]) > 0.f) ? d[e] : 0.f;
        const float xh = (f[e] - mu[e]) * rs[e];
        o[e] = gs[e] * (dyh - k1[e] - xh * k2[e]);
      }
    }
    u32x4_t v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
    out[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Block ends (conv -> BN -> ReLU -> 2x2 max-pool): the normalised full-resolution activation z and
// its gradient dz exist only inside these kernels.  Un-fused, a block-end layer costs 10.5
// activation-sized passes over HBM per step (stats R; apply R+W; pool R + W/4; pool-bwd R/4 + W;
// bwd-reduce 2R; bwd-apply 2R+W); fused it costs 5.75 (stats R; apply+pool R + W/4; bwd-reduce
// R + R/4; bwd-apply R + R/4 + W) — every kernel here streams at the HBM rate, so passes are time.
//
// Work mapping (all three kernels): a workgroup owns pairs of pooled rows q = n*OH + oh; thread t
// serves row 2*pair + (t >> 7) and the 16-byte vectors j = (t & 127) + 128*k of that row, so its
// channel group g = j mod (C/8) is fixed (C/8 divides 128) and scale/shift stay in registers.
// The 4-bit codes are maxpool2x2_fwd_kernel's: bits 1:0 the FIRST maximal window position (compared
// on the bf16-rounded values, exactly what the un-fused pool sees), bit 2 "max > 0".
struct PoolGeom { int NOH, OH, H, W, C, cg_shift; };

DTP_DEVINL void bn_round_relu8(const u32x4_t v, const float* sc, const float* sh, float* z) {
  float f[8];
  unpack8(v, f);
#pragma unroll
  for (int e = 0; e < 4; ++e) {          // bf16 rounding, then ReLU: the values the un-fused z tensor holds
    const uint32_t pk = pack_bf16x2(fmaxf(fmaf(f[2 * e], sc[2 * e], sh[2 * e]), 0.f),
                                    fmaxf(fmaf(f[2 * e + 1], sc[2 * e + 1], sh[2 * e + 1]), 0.f));
    z[2 * e] = __uint_as_float(pk << 16);
    z[2 * e + 1] = __uint_as_float(pk & 0xFFFF0000u);
  }
}

__global__ __launch_bounds__(BN_T) void bn_apply_pool_kernel(const u32x4_t* __restrict__ y, u32x4_t* __restrict__ zp,
                                                             uint32_t* __restrict__ codes,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const PoolGeom gm) {
  const int cg = gm.C >> 3, OW = gm.W >> 1, rowlen = OW * cg;
  const int half = threadIdx.x >> 7, t = threadIdx.x & 127, g = t & (cg - 1);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = scale[g * 8 + e]; sh[e] = shift[g * 8 + e]; }
  for (int q = 2 * blockIdx.x + half; q < gm.NOH; q += 2 * gridDim.x) {
    const int n = q / gm.OH, oh = q - n * gm.OH;
    const u32x4_t* r0 = y + ((size_t)n * gm.H + 2 * oh) * gm.W * cg;
    const u32x4_t* r1 = r0 + (size_t)gm.W * cg;
    for (int j = t; j < rowlen; j += 128) {
      const int ow = j >> gm.cg_shift;
      const int i0 = (2 * ow) * cg + g;
      float za[8], zb[8], zc[8], zd[8];
      bn_round_relu8(r0[i0], sc, sh, za);
      bn_round_relu8(r0[i0 + cg], sc, sh, zb);
      bn_round_relu8(r1[i0], sc, sh, zc);
      bn_round_relu8(r1[i0 + cg], sc, sh, zd);
      uint32_t code = 0;
      float mx[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        uint32_t arg = 0; float m = za[e];
        if (zb[e] > m) { m = zb[e]; arg = 1; }
        if (zc[e] > m) { m = zc[e]; arg = 2; }
        if (zd[e] > m) { m = zd[e]; arg = 3; }
        mx[e] = m;
        code |= (arg | (m > 0.f ? 4u : 0u)) << (4 * e);
      }
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e)   // exact: the values are bf16 already
        o[e] = (__float_as_uint(mx[2 * e]) >> 16) | (__float_as_uint(mx[2 * e + 1]) & 0xFFFF0000u);
      const size_t oi = ((size_t)q * OW + ow) * cg + g;
      zp[oi] = o;
      codes[oi] = code;
    }
  }
}

// MODE 1 of bn_reduce_kernel for a pooled dz: only the window's winner carries a gradient, so
// (sum dyh, sum dyh*xhat) needs y at the winner — all four vectors are read (one pass over y).
__global__ __launch_bounds__(BN_T) void bn_reduce_pool_kernel(const u32x4_t* __restrict__ y,
                                                              const u32x4_t* __restrict__ dzp,
                                                              const uint32_t* __restrict__ codes,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd,
                                                              float* __restrict__ partial, const PoolGeom gm) {
  __shared__ float red[2][2048];
  const int cg = gm.C >> 3, OW = gm.W >> 1, rowlen = OW * cg;
  const int half = threadIdx.x >> 7, t = threadIdx.x & 127, g = t & (cg - 1);
  float a0[8], a1[8], mu[8], rs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a0[e] = a1[e] = 0.f; mu[e] = mean[g * 8 + e]; rs[e] = rstd[g * 8 + e]; }
  for (int q = 2 * blockIdx.x + half; q < gm.NOH; q += 2 * gridDim.x) {
    const int n = q / gm.OH, oh = q - n * gm.OH;
    const u32x4_t* r0 = y + ((size_t)n * gm.H + 2 * oh) * gm.W * cg;
    const u32x4_t* r1 = r0 + (size_t)gm.W * cg;
    for (int j = t; j < rowlen; j += 128) {
      const int ow = j >> gm.cg_shift;
      const int i0 = (2 * ow) * cg + g;
      const size_t oi = ((size_t)q * OW + ow) * cg + g;
      float fa[8], fb[8], fc[8], fd[8], d[8];
      unpack8(r0[i0], fa); unpack8(r0[i0 + cg], fb); unpack8(r1[i0], fc); unpack8(r1[i0 + cg], fd);
      unpack8(dzp[oi], d);
      const uint32_t code = codes[oi];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t c4 = (code >> (4 * e)) & 15u, arg = c4 & 3u;
        const float dyh = (c4 & 4u) ? d[e] : 0.f;
        const float yw = arg == 0 ? fa[e] : (arg == 1 ? fb[e] : (arg == 2 ? fc[e] : fd[e]));
        a0[e] += dyh;
        a1[e] = fmaf(dyh, (yw - mu[e]) * rs[e], a1[e]);
      }
    }
  }
  // fold the BN_T / cg threads that share a channel group (thread -> slot threadIdx.x / cg)
  const int slot = (threadIdx.x >> gm.cg_shift), R = BN_T >> gm.cg_shift;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][slot * gm.C + g * 8 + e] = a0[e];
    red[1][slot * gm.C + g * 8 + e] = a1[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * gm.C; c += BN_T) {
    const int which = c / gm.C, ch = c - which * gm.C;
    float s = 0.f;
    for (int k = 0; k < R; ++k) s += red[which][k * gm.C + ch];
    partial[((size_t)blockIdx.x * 2 + which) * gm.C + ch] = s;
  }
}

// MODE 1 of bn_apply_kernel for a pooled dz: dy = g*(dyh - c1 - xhat*c2) at all four window
// positions, dyh = dz_pooled at the winner (when its z > 0), 0 elsewhere.
__global__ __launch_bounds__(BN_T) void bn_bwd_apply_pool_kernel(const u32x4_t* __restrict__ y,
                                                                 const u32x4_t* __restrict__ dzp,
                                                                 const uint32_t* __restrict__ codes,
                                                                 u32x4_t* __restrict__ dy,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd,
                                                                 const float* __restrict__ c1,
                                                                 const float* __restrict__ c2,
                                                                 const float* __restrict__ gamma, const PoolGeom gm) {
  const int cg = gm.C >> 3, OW = gm.W >> 1, rowlen = OW * cg;
  const int half = threadIdx.x >> 7, t = threadIdx.x & 127, g = t & (cg - 1);
  float gs[8], ka[8], kb[8];       // dy = gs*dyh + (ka + y*kb):  ka = -gs*(c1 - mu*rs*c2), kb = -gs*rs*c2
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float mu = mean[g * 8 + e], rs = rstd[g * 8 + e], k1 = c1[g * 8 + e], k2 = c2[g * 8 + e];
    gs[e] = gamma[g * 8 + e] * rs;
    ka[e] = -gs[e] * (k1 - mu * rs * k2);
    kb[e] = -gs[e] * rs * k2;
  }
  for (int q = 2 * blockIdx.x + half; q < gm.NOH; q += 2 * gridDim.x) {
    const int n = q / gm.OH, oh = q - n * gm.OH;
    const size_t row0 = ((size_t)n * gm.H + 2 * oh) * gm.W * cg, row1 = row0 + (size_t)gm.W * cg;
    for (int j = t; j < rowlen; j += 128) {
      const int ow = j >> gm.cg_shift;
      const int i0 = (2 * ow) * cg + g;
      const size_t oi = ((size_t)q * OW + ow) * cg + g;
      float f[4][8], d[8];
      unpack8(y[row0 + i0], f[0]); unpack8(y[row0 + i0 + cg], f[1]);
      unpack8(y[row1 + i0], f[2]); unpack8(y[row1 + i0 + cg], f[3]);
      unpack8(dzp[oi], d);
      const uint32_t code = codes[oi];
      float gd[8];
      uint32_t arg[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t c4 = (code >> (4 * e)) & 15u;
        arg[e] = c4 & 3u;
        gd[e] = (c4 & 4u) ? gs[e] * d[e] : 0.f;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(f[p][e], kb[e], ka[e]) + (arg[e] == (uint32_t)p ? gd[e] : 0.f);
        u32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
        dy[((p & 2) ? row1 : row0) + i0 + ((p & 1) ? cg : 0)] = v;
      }
    }
  }
}

int bn_grid(long M, int C) {
  const int R = BN_T / (C >> 3);
  long blocks = (M + (long)R * 8 - 1) / ((long)R * 8);      // >= 8 rows per thread
  if (blocks > BN_MAXBLK) blocks = BN_MAXBLK;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// Producer-side statistics: the convolution in front of this BatchNorm left (sum y, sum y^2) per
// output tile ([ntiles][2][C], igemm.hip colsum_sq) — fold groups of consecutive tile rows into at
// most BN_MAXBLK partial rows (fixed order: deterministic) for bn_finalize_kernel; the pass over y
// that bn_reduce_kernel<0> makes is not needed then.
__global__ __launch_bounds__(256) void bn_fold_tiles_kernel(const float* __restrict__ tp, int ntiles, int per, int C,
                                                            float* __restrict__ partial) {
  const int r0 = blockIdx.x * per;
  int r1 = r0 + per;
  if (r1 > ntiles) r1 = ntiles;
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += tp[(size_t)r * 2 * C + c];
    partial[(size_t)blockIdx.x * 2 * C + c] = s;
  }
}

// The backward form: the data gradient that produced dz left (sum dyh, sum dyh*y) per tile (igemm.hip
// bn_y); sum dyh*xhat = rstd * (sum dyh*y - mean * sum dyh), applied per folded row.
__global__ __launch_bounds__(256) void bn_fold_tiles_bwd_kernel(const float* __restrict__ tp, int ntiles, int per, int C,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ partial) {
  const int r0 = blockIdx.x * per;
  int r1 = r0 + per;
  if (r1 > ntiles) r1 = ntiles;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int r = r0; r < r1; ++r) {
      a += tp[(size_t)r * 2 * C + c];
      b += tp[((size_t)r * 2 + 1) * C + c];
    }
    partial[(size_t)blockIdx.x * 2 * C + c] = a;
    partial[((size_t)blockIdx.x * 2 + 1) * C + c] = rstd[c] * (b - mean[c] * a);
  }
}

int bn_fold_tiles_bwd(const float* tp, int ntiles, int C, const float* mean, const float* rstd, float* partial,
                      hipStream_t s) {
  const int per = (ntiles + BN_MAXBLK - 1) / BN_MAXBLK;
  const int nblk = (ntiles + per - 1) / per;
  hipLaunchKernelGGL(bn_fold_tiles_bwd_kernel, dim3(nblk), dim3(256), 0, s, tp, ntiles, per, C, mean, rstd, partial);
  return nblk;
}

// -> number of partial rows written
int bn_fold_tiles(const float* tp, int ntiles, int C, float* partial, hipStream_t s) {
  const int per = (ntiles + BN_MAXBLK - 1) / BN_MAXBLK;
  const int nblk = (ntiles + per - 1) / per;
  hipLaunchKernelGGL(bn_fold_tiles_kernel, dim3(nblk), dim3(256), 0, s, tp, ntiles, per, C, partial);
  return nblk;
}

bool bn_shape_ok(long M, int C) { return M > 0 && C >= 64 && C <= 512 && (C & (C - 1)) == 0; }

}  // namespace

// partial: >= 1024 * 2 * C floats.  stats (4 x C floats): mean | rstd | scale | shift.
extern "C" int dtp_bn_fwd_train(const void* y, void* z, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float* stats, float* partial, long M, int C, float eps,
                                float momentum, int relu, const float* tile_partial, int ntiles, hipStream_t s) {
  if (!bn_shape_ok(M, C)) return -10;
  if (tile_partial && ntiles < 1) return -11;
  int nblk = bn_grid(M, C);
  float *mean = stats, *rstd = stats + C, *scale = stats + 2 * C, *shift = stats + 3 * C;
  if (tile_partial)
    nblk = bn_fold_tiles(tile_partial, ntiles, C, partial, s);
  else
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(nblk), dim3(BN_T), 0, s, (const u32x4_t*)y, nullptr, nullptr, nullptr,
                       nullptr, nullptr, partial, M, C, relu);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + BN_FC - 1) / BN_FC), dim3(256), 0, s, partial, nblk, C, M, eps, momentum, gamma,
                     beta, mean, rstd, scale, shift, running_mean, running_var);
  const long nvec = M * (C >> 3);
  long g = (nvec + BN_T * 8 - 1) / (BN_T * 8);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(bn_apply_kernel<0>, dim3((int)g), dim3(BN_T), 0, s, (const u32x4_t*)y, nullptr, (u32x4_t*)z, scale,
                     shift, nullptr, nullptr, nullptr, nullptr, nullptr, nvec, C, relu);
  return DTP_CHECK_LAUNCH();
}

// coeffs (2 x C floats): scale | shift from the running statistics
extern "C" int dtp_bn_fwd_eval(const void* y, void* z, const float* gamma, const float* beta, const float* running_mean,
                               const float* running_var, float* coeffs, long M, int C, float eps, int relu,
                               hipStream_t s) {
  if (!bn_shape_ok(M, C)) return -10;
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, s, C, eps, gamma, beta, running_mean,
                     running_var, coeffs, coeffs + C);
  if (y == nullptr) return DTP_CHECK_LAUNCH();          // coefficients only (conv folding)
  const long nvec = M * (C >> 3);
  long g = (nvec + BN_T * 8 - 1) / (BN_T * 8);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(bn_apply_kernel<0>, dim3((int)g), dim3(BN_T), 0, s, (const u32x4_t*)y, nullptr, (u32x4_t*)z,
                     coeffs, coeffs + C, nullptr, nullptr, nullptr, nullptr, nullptr, nvec, C, relu);
  return DTP_CHECK_LAUNCH();
}

// stats as written by dtp_bn_fwd_train; work: 2 x C floats (c1 | c2)
extern "C" int dtp_bn_bwd(const void* y, const void* dz, void* dy, const float* gamma, const float* stats, void* dgamma,
                          void* dbeta, int grads_fp32, float beta_g, float beta_b, float* partial, float* work, long M,
                          int C, int relu, const float* tile_partial, int ntiles, hipStream_t s) {
  if (!bn_shape_ok(M, C)) return -10;
  if (tile_partial && ntiles < 1) return -11;
  int nblk = bn_grid(M, C);
  const float *mean = stats, *rstd = stats + C, *scale = stats + 2 * C, *shift = stats + 3 * C;
  if (tile_partial)      // the producer of dz left the two reductions per tile
    nblk = bn_fold_tiles_bwd(tile_partial, ntiles, C, mean, rstd, partial, s);
  else
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(nblk), dim3(BN_T), 0, s, (const u32x4_t*)y, (const u32x4_t*)dz, scale,
                       shift, mean, rstd, partial, M, C, relu);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + BN_FC - 1) / BN_FC), dim3(256), 0, s, partial, nblk, C, M, dgamma, dbeta,
                     grads_fp32, beta_g, beta_b, work, work + C);
  const long nvec = M * (C >> 3);
  long g = (nvec + BN_T * 8 - 1) / (BN_T * 8);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(bn_apply_kernel<1>, dim3((int)g), dim3(BN_T), 0, s, (const u32x4_t*)y, (const u32x4_t*)dz,
                     (u32x4_t*)dy, scale, shift, mean, rstd, work, work + C, gamma, nvec, C, relu);
  return DTP_CHECK_LAUNCH();
}

// ---- block-end forms: BN + ReLU + 2x2/s2 max-pool (training).  y: [N][H][W][C] bf16, H and W even.
static bool bn_pool_geom(PoolGeom& gm, int N, int H, int W, int C) {
  if (N <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || !bn_shape_ok((long)N * H * W, C)) return false;
  if ((long)N * (H / 2) > 0x7fffffffL) return false;
  gm.NOH = N * (H / 2); gm.OH = H / 2; gm.H = H; gm.W = W; gm.C = C; gm.cg_shift = __builtin_ctz(C >> 3);
  return true;
}
static int bn_pool_grid(const PoolGeom& gm) {
  long b = ((long)gm.NOH + 1) / 2;
  if (b > BN_MAXBLK) b = BN_MAXBLK;
  return (int)(b < 1 ? 1 : b);
}

// zp: [N][H/2][W/2][C] bf16, codes: [N][H/2][W/2][C/8] uint32; stats / partial as dtp_bn_fwd_train
extern "C" int dtp_bn_fwd_train_pool(const void* y, void* zp, void* codes, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float* stats, float* partial, int N,
                                     int H, int W, int C, float eps, float momentum, const float* tile_partial,
                                     int ntiles, hipStream_t s) {
  PoolGeom gm;
  if (!bn_pool_geom(gm, N, H, W, C)) return -10;
  if (tile_partial && ntiles < 1) return -11;
  const long M = (long)N * H * W;
  int nblk = bn_grid(M, C);
  float *mean = stats, *rstd = stats + C, *scale = stats + 2 * C, *shift = stats + 3 * C;
  if (tile_partial)
    nblk = bn_fold_tiles(tile_partial, ntiles, C, partial, s);
  else
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(nblk), dim3(BN_T), 0, s, (const u32x4_t*)y, nullptr, nullptr, nullptr,
                       nullptr, nullptr, partial, M, C, 1);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + BN_FC - 1) / BN_FC), dim3(256), 0, s, partial, nblk, C, M, eps,
                     momentum, gamma, beta, mean, rstd, scale, shift, running_mean, running_var);
  hipLaunchKernelGGL(bn_apply_pool_kernel, dim3(2 * bn_pool_grid(gm)), dim3(BN_T), 0, s, (const u32x4_t*)y,
                     (u32x4_t*)zp, (uint32_t*)codes, scale, shift, gm);
  return DTP_CHECK_LAUNCH();
}

extern "C" int dtp_bn_bwd_pool(const void* y, const void* dzp, const void* codes, void* dy, const float* gamma,
                               const float* stats, void* dgamma, void* dbeta, int grads_fp32, float beta_g,
                               float beta_b, float* partial, float* work, int N, int H, int W, int C, hipStream_t s) {
  PoolGeom gm;
  if (!bn_pool_geom(gm, N, H, W, C)) return -10;
  const long M = (long)N * H * W;
  const float *mean = stats, *rstd = stats + C;
  const int nblk = bn_pool_grid(gm);
  hipLaunchKernelGGL(bn_reduce_pool_kernel, dim3(nblk), dim3(BN_T), 0, s, (const u32x4_t*)y, (const u32x4_t*)dzp,
                     (const uint32_t*)codes, mean, rstd, partial, gm);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + BN_FC - 1) / BN_FC), dim3(256), 0, s, partial, nblk, C, M,
                     dgamma, dbeta, grads_fp32, beta_g, beta_b, work, work + C);
  hipLaunchKernelGGL(bn_bwd_apply_pool_kernel, dim3(2 * nblk), dim3(BN_T), 0, s, (const u32x4_t*)y,
                     (const u32x4_t*)dzp, (const uint32_t*)codes, (u32x4_t*)dy, mean, rstd, work, work + C, gamma, gm);
  return DTP_CHECK_LAUNCH();
}
