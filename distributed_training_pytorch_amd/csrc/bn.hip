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

int bn_grid(long M, int C) {
  const int R = BN_T / (C >> 3);
  long blocks = (M + (long)R * 8 - 1) / ((long)R * 8);      // >= 8 rows per thread
  if (blocks > BN_MAXBLK) blocks = BN_MAXBLK;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

bool bn_shape_ok(long M, int C) { return M > 0 && C >= 64 && C <= 512 && (C & (C - 1)) == 0; }

}  // namespace

// partial: >= 1024 * 2 * C floats.  stats (4 x C floats): mean | rstd | scale | shift.
extern "C" int dtp_bn_fwd_train(const void* y, void* z, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float* stats, float* partial, long M, int C, float eps,
                                float momentum, int relu, hipStream_t s) {
  if (!bn_shape_ok(M, C)) return -10;
  const int nblk = bn_grid(M, C);
  float *mean = stats, *rstd = stats + C, *scale = stats + 2 * C, *shift = stats + 3 * C;
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
                          int C, int relu, hipStream_t s) {
  if (!bn_shape_ok(M, C)) return -10;
  const int nblk = bn_grid(M, C);
  const float *mean = stats, *rstd = stats + C, *scale = stats + 2 * C, *shift = stats + 3 * C;
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
