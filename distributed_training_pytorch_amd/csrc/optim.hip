// Fused flat-buffer optimizers for gfx950: one launch updates every parameter of the model.
// All parameters live in one f32 master arena; gradients in one arena (f32 or bf16); the kernel
// also emits the bf16 working copy the MFMA kernels read, so no separate cast pass exists.
// Reference parity: torch.optim.SGD(lr, momentum, weight_decay) foreach path used at
// example_trainer.py:62,88 — same update rule, incl. dampening/nesterov;
// AdamW per BASELINE.json north_star.
#include "dtp_common.h"
#include <cstdlib>

template <bool GRAD_BF16>
DTP_DEVINL f32x4_t load_grad4(const void* g, size_t i4) {
  if (GRAD_BF16) {
    const u32x2_t q = *(const u32x2_t*)((const bf16_t*)g + i4 * 4);
    return (f32x4_t){__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xFFFF0000u),
                     __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xFFFF0000u)};
  }
  return *(const f32x4_t*)((const float*)g + i4 * 4);
}

// torch.optim.SGD semantics:  g += wd*p ; buf = first ? g : m*buf + (1-damp)*g ; g = nesterov ? g + m*buf : buf
//                             p -= lr*g
// Both kernels are "polite" streaming kernels: a bounded persistent grid (grid_for: 2 workgroups per CU) whose
// threads keep OPT_UNROLL independent 16-byte groups in flight — enough outstanding loads for the HBM rate
// at 2 waves per SIMD, while the other wave slots stay free.  With the earlier 4096-block grid every wave
// slot of the chip was taken for the length of the kernel, and a kernel launched next to it from another
// stream (the rest of backward, when the update of a gradient bucket runs under it) did not start until the
// update had drained: at per-GPU batch 4 a 10 us split-K reduction took 250 us next to the 103 M-element
// bucket (profiles/76_*).
#define OPT_UNROLL 4

template <bool GRAD_BF16>
__global__ __launch_bounds__(256) void sgd_flat_kernel(float* __restrict__ p, const void* __restrict__ g, float* __restrict__ mom,
                                bf16_t* __restrict__ lp, size_t n4, float lr, float momentum, float dampening,
                                float wd, float grad_scale, int nesterov, int first_step) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const bool use_mom = momentum != 0.f;
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * OPT_UNROLL) {
    f32x4_t pv[OPT_UNROLL], gv[OPT_UNROLL], bv[OPT_UNROLL];
#pragma unroll
    for (int u = 0; u < OPT_UNROLL; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) {
        pv[u] = *(const f32x4_t*)(p + i * 4);
        gv[u] = load_grad4<GRAD_BF16>(g, i);
        if (use_mom && !first_step) bv[u] = *(const f32x4_t*)(mom + i * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < OPT_UNROLL; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) {
        f32x4_t gg = gv[u] * grad_scale;
        gg += pv[u] * wd;
        if (use_mom) {
          f32x4_t b;
          if (first_step) b = gg;
          else b = bv[u] * momentum + gg * (1.f - dampening);
          *(f32x4_t*)(mom + i * 4) = b;
          gg = nesterov ? gg + b * momentum : b;
        }
        const f32x4_t pn = pv[u] - gg * lr;
        *(f32x4_t*)(p + i * 4) = pn;
        if (lp) {
          u32x2_t o;
          o[0] = pack_bf16x2(pn[0], pn[1]);
          o[1] = pack_bf16x2(pn[2], pn[3]);
          *(u32x2_t*)(lp + i * 4) = o;
        }
      }
    }
  }
}

// AdamW (decoupled weight decay): p *= 1 - lr*wd ; m,v EMA ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
template <bool GRAD_BF16>
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const void* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, bf16_t* __restrict__ lp, size_t n4, float lr, float beta1,
                                  float beta2, float eps, float wd, float grad_scale, float bc1, float bc2_sqrt) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float step = lr / bc1;
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * OPT_UNROLL) {
    f32x4_t pv[OPT_UNROLL], gv[OPT_UNROLL], mv[OPT_UNROLL], vv[OPT_UNROLL];
#pragma unroll
    for (int u = 0; u < OPT_UNROLL; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) {
        pv[u] = *(const f32x4_t*)(p + i * 4);
        gv[u] = load_grad4<GRAD_BF16>(g, i);
        mv[u] = *(const f32x4_t*)(m + i * 4);
        vv[u] = *(const f32x4_t*)(v + i * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < OPT_UNROLL; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) {
        const f32x4_t gg = gv[u] * grad_scale;
        f32x4_t pn = pv[u] * (1.f - lr * wd);
        const f32x4_t mn = mv[u] * beta1 + gg * (1.f - beta1);
        const f32x4_t vn = vv[u] * beta2 + gg * gg * (1.f - beta2);
#pragma unroll
        for (int e = 0; e < 4; ++e) pn[e] -= step * mn[e] / (sqrtf(vn[e]) / bc2_sqrt + eps);
        *(f32x4_t*)(p + i * 4) = pn;
        *(f32x4_t*)(m + i * 4) = mn;
        *(f32x4_t*)(v + i * 4) = vn;
        if (lp) {
          u32x2_t o;
          o[0] = pack_bf16x2(pn[0], pn[1]);
          o[1] = pack_bf16x2(pn[2], pn[3]);
          *(u32x2_t*)(lp + i * 4) = o;
        }
      }
    }
  }
}

// 2 workgroups (8 waves) per CU by default; DTP_OPT_GRID_CAP overrides (A/B: 4096 = the old full-chip grid)
static inline int grid_for(size_t work) {
  static int cap = 0;
  if (cap == 0) {
    const char* e = getenv("DTP_OPT_GRID_CAP");
    cap = e ? atoi(e) : 512;
    if (cap < 1) cap = 512;
  }
  size_t g = (work + 256 * OPT_UNROLL - 1) / (256 * OPT_UNROLL);
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// n must be a multiple of 4 (the arena pads); all pointers 16-byte aligned.
extern "C" int dtp_sgd_flat(float* p, const void* g, int grad_bf16, float* mom, void* lp, size_t n, float lr,
                            float momentum, float dampening, float wd, float grad_scale, int nesterov,
                            int first_step, hipStream_t s) {
  if (n == 0 || (n & 3)) return -10;
  if (momentum != 0.f && mom == nullptr) return -11;
  if (grad_bf16)
    hipLaunchKernelGGL(sgd_flat_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, s, p, g, mom, (bf16_t*)lp, n / 4, lr,
                       momentum, dampening, wd, grad_scale, nesterov, first_step);
  else
    hipLaunchKernelGGL(sgd_flat_kernel<false>, dim3(grid_for(n / 4)), dim3(256), 0, s, p, g, mom, (bf16_t*)lp, n / 4,
                       lr, momentum, dampening, wd, grad_scale, nesterov, first_step);
  return DTP_CHECK_LAUNCH();
}

extern "C" int dtp_adamw_flat(float* p, const void* g, int grad_bf16, float* m, float* v, void* lp, size_t n, float lr,
                              float beta1, float beta2, float eps, float wd, float grad_scale, int step,
                              hipStream_t s) {
  if (n == 0 || (n & 3) || step < 1) return -10;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  if (grad_bf16)
    hipLaunchKernelGGL(adamw_flat_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, s, p, g, m, v, (bf16_t*)lp, n / 4,
                       lr, beta1, beta2, eps, wd, grad_scale, bc1, bc2_sqrt);
  else
    hipLaunchKernelGGL(adamw_flat_kernel<false>, dim3(grid_for(n / 4)), dim3(256), 0, s, p, g, m, v, (bf16_t*)lp, n / 4,
                       lr, beta1, beta2, eps, wd, grad_scale, bc1, bc2_sqrt);
  return DTP_CHECK_LAUNCH();
}
