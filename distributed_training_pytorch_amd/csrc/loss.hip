// Fused softmax-cross-entropy (+ top-1 hit) for gfx950.
// One kernel produces, per row, the loss term, the correct-prediction flag and the gradient
// dlogits = (softmax - onehot) * grad_scale in both f32 and bf16 (bf16 copy is row-padded so it can
// feed the weight-gradient GEMM directly).  A second tiny kernel reduces the per-row terms in a
// fixed order (deterministic) into {mean loss, #correct}.
// Reference parity: F.cross_entropy(reduction="mean") + its backward at example_trainer.py:57,84-86
// and softmax/argmax/eq/sum at example_trainer.py:99-101.
#include "dtp_common.h"

// one wave per row
__global__ void softmax_xent_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                    float* __restrict__ row_loss, float* __restrict__ row_hit,
                                    float* __restrict__ dlogits_f32, bf16_t* __restrict__ dlogits_bf16, int N, int C,
                                    int ld_logits, int ld_bf16, float grad_scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* x = logits + (size_t)row * ld_logits;
  float mx = -INFINITY;
  int amax = 0;
  for (int c = lane; c < C; c += 64) {
    const float v = x[c];
    if (v > mx) { mx = v; amax = c; }
  }
  // wave arg-max with lowest-index tie rule
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float om = __shfl_xor(mx, off, 64);
    const int oa = __shfl_xor(amax, off, 64);
    if (om > mx || (om == mx && oa < amax)) { mx = om; amax = oa; }
  }
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += __expf(x[c] - mx);
  se = wave_reduce_sum(se);
  const int lb = (int)labels[row];
  const bool lb_ok = lb >= 0 && lb < C;
  const float lse = __logf(se) + mx;
  if (lane == 0) {
    row_loss[row] = lb_ok ? lse - x[lb] : 0.f;
    row_hit[row] = (lb_ok && amax == lb) ? 1.f : 0.f;
  }
  const float inv = 1.f / se;
  for (int c = lane; c < ld_bf16 || c < C; c += 64) {
    float g = 0.f;
    if (c < C && lb_ok) g = (__expf(x[c] - mx) * inv - (c == lb ? 1.f : 0.f)) * grad_scale;
    if (dlogits_f32 && c < C) dlogits_f32[(size_t)row * C + c] = g;
    if (dlogits_bf16 && c < ld_bf16) dlogits_bf16[(size_t)row * ld_bf16 + c] = f32_to_bf16(g);
  }
}

// out[0] = sum(row_loss) * loss_scale ; out[1] = sum(row_hit)      (single block, fixed order)
__global__ void xent_reduce_kernel(const float* __restrict__ row_loss, const float* __restrict__ row_hit,
                                   float* __restrict__ out, int N, float loss_scale) {
  __shared__ float sl[256], sh[256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) { a += row_loss[i]; b += row_hit[i]; }
  sl[threadIdx.x] = a; sh[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sl[threadIdx.x] += sl[threadIdx.x + s]; sh[threadIdx.x] += sh[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sl[0] * loss_scale; out[1] = sh[0]; }
}

// row_ws: 2*N floats of scratch. out2: {loss, hits}. grad_scale normally 1/N (mean) * 1/world.
extern "C" int dtp_softmax_xent(const float* logits, const long long* labels, float* row_ws, float* out2,
                                float* dlogits_f32, void* dlogits_bf16, int N, int C, int ld_logits, int ld_bf16,
                                float grad_scale, float loss_scale, hipStream_t s) {
  if (N <= 0 || C <= 0 || ld_logits < C) return -10;
  if (dlogits_bf16 && ld_bf16 < C) return -11;
  const int rows_per_block = 4;
  hipLaunchKernelGGL(softmax_xent_kernel, dim3((N + rows_per_block - 1) / rows_per_block), dim3(256), 0, s, logits,
                     labels, row_ws, row_ws + N, dlogits_f32, (bf16_t*)dlogits_bf16, N, C, ld_logits,
                     dlogits_bf16 ? ld_bf16 : 0, grad_scale);
  hipLaunchKernelGGL(xent_reduce_kernel, dim3(1), dim3(256), 0, s, row_ws, row_ws + N, out2, N, loss_scale);
  return DTP_CHECK_LAUNCH();
}
