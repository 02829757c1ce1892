// NHWC bf16 pooling kernels for gfx950: 2x2/s2 max-pool forward/backward (vectorised 16 B per
// lane, scatter-free backward with the ReLU-backward mask fused in) and the
// AdaptiveAvgPool(OHxOW)+flatten pair that hands the feature map to the classifier in the
// reference's CHW-flatten order (reference model/vgg16.py:14,34,67-70).
#include "dtp_common.h"

DTP_DEVINL void unpack8(const u32x4_t v, float* f) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __uint_as_float(v[e] << 16);
    f[2 * e + 1] = __uint_as_float(v[e] & 0xFFFF0000u);
  }
}

// Each thread: one output pixel x 8 channels. Windows never overlap (k=2, s=2); odd trailing
// row/column is dropped like torch's floor mode.
// idx (optional): one uint32 per (output pixel, 8 channels) = eight 4-bit codes
// {bit 2: max > 0, bits 1:0: position of the FIRST maximum in the window}.  With it the backward
// pass no longer re-reads the full-resolution input (1.6 GB for block 1 at batch 256).
__global__ void maxpool2x2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                      uint32_t* __restrict__ idx, int N, int H, int W, int C) {
  const int OH = H >> 1, OW = W >> 1, CV = C >> 3;
  const size_t total = (size_t)N * OH * OW * CV;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    size_t r = i / CV;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    const size_t base = (((size_t)n * H + 2 * oh) * W + 2 * ow) * C + cv * 8;
    const u32x4_t a = *(const u32x4_t*)(x + base);
    const u32x4_t b = *(const u32x4_t*)(x + base + C);
    const u32x4_t c = *(const u32x4_t*)(x + base + (size_t)W * C);
    const u32x4_t d = *(const u32x4_t*)(x + base + (size_t)W * C + C);
    float fa[8], fb[8], fc[8], fd[8];
    unpack8(a, fa); unpack8(b, fb); unpack8(c, fc); unpack8(d, fd);
    u32x4_t o;
    float mx[8];
    uint32_t code = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      uint32_t arg = 0; float m = fa[e];
      if (fb[e] > m) { m = fb[e]; arg = 1; }
      if (fc[e] > m) { m = fc[e]; arg = 2; }
      if (fd[e] > m) { m = fd[e]; arg = 3; }
      mx[e] = m;
      code |= (arg | (m > 0.f ? 4u : 0u)) << (4 * e);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)   // exact: inputs are bf16
      o[e] = (__float_as_uint(mx[2 * e]) >> 16) | (__float_as_uint(mx[2 * e + 1]) & 0xFFFF0000u);
    const size_t oo = ((size_t)n * OH + oh) * OW + ow;
    *(u32x4_t*)(y + oo * C + cv * 8) = o;
    if (idx) idx[oo * CV + cv] = code;
  }
}

// dx = dy routed to the FIRST maximal element of each window (torch's tie rule), optionally
// multiplied by (x > 0): the ReLU that precedes the pool in every VGG block.
// colsum_part (optional, [gridDim.x][C] f32): per-block column sums of dx — the bias gradient of
// the conv that produced x, for free while dx is in registers (needs 256 % (C/8) == 0 so that a
// thread keeps its channel chunk over the grid-stride loop).
__global__ void maxpool2x2_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                      bf16_t* __restrict__ dx, int N, int H, int W, int C, int relu_mask,
                                      float* __restrict__ colsum_part) {
  const int OH = H >> 1, OW = W >> 1, CV = C >> 3;
  const size_t total = (size_t)N * OH * OW * CV;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    size_t r = i / CV;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    const size_t base = (((size_t)n * H + 2 * oh) * W + 2 * ow) * C + cv * 8;
    const size_t o01 = C, o10 = (size_t)W * C, o11 = (size_t)W * C + C;
    float fa[8], fb[8], fc[8], fd[8], g[8];
    unpack8(*(const u32x4_t*)(x + base), fa);
    unpack8(*(const u32x4_t*)(x + base + o01), fb);
    unpack8(*(const u32x4_t*)(x + base + o10), fc);
    unpack8(*(const u32x4_t*)(x + base + o11), fd);
    unpack8(*(const u32x4_t*)(dy + (((size_t)n * OH + oh) * OW + ow) * C + cv * 8), g);
    float ra[8], rb[8], rc[8], rd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int arg = 0; float m = fa[e];
      if (fb[e] > m) { m = fb[e]; arg = 1; }
      if (fc[e] > m) { m = fc[e]; arg = 2; }
      if (fd[e] > m) { m = fd[e]; arg = 3; }
      const float gv = (relu_mask && !(m > 0.f)) ? 0.f : g[e];
      ra[e] = arg == 0 ? gv : 0.f; rb[e] = arg == 1 ? gv : 0.f;
      rc[e] = arg == 2 ? gv : 0.f; rd[e] = arg == 3 ? gv : 0.f;
      cs[e] += gv;                 // exactly one element of the window receives gv
    }
    u32x4_t oa, ob, oc, od;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      oa[e] = (__float_as_uint(ra[2 * e]) >> 16) | (__float_as_uint(ra[2 * e + 1]) & 0xFFFF0000u);
      ob[e] = (__float_as_uint(rb[2 * e]) >> 16) | (__float_as_uint(rb[2 * e + 1]) & 0xFFFF0000u);
      oc[e] = (__float_as_uint(rc[2 * e]) >> 16) | (__float_as_uint(rc[2 * e + 1]) & 0xFFFF0000u);
      od[e] = (__float_as_uint(rd[2 * e]) >> 16) | (__float_as_uint(rd[2 * e + 1]) & 0xFFFF0000u);
    }
    *(u32x4_t*)(dx + base) = oa;
    *(u32x4_t*)(dx + base + o01) = ob;
    *(u32x4_t*)(dx + base + o10) = oc;
    *(u32x4_t*)(dx + base + o11) = od;
  }
  if (colsum_part) {   // kernel-uniform
    __shared__ float red[256 * 8];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = cs[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int cv = c >> 3, e = c & 7;
      float s = 0.f;
      for (int t = cv; t < 256; t += CV) s += red[t * 8 + e];
      colsum_part[(size_t)blockIdx.x * C + c] = s;
    }
  }
}

// Same routing from the forward pass's 4-bit codes instead of the input tensor.
__global__ void maxpool2x2_bwd_idx_kernel(const uint32_t* __restrict__ idx, const bf16_t* __restrict__ dy,
                                          bf16_t* __restrict__ dx, int N, int H, int W, int C, int relu_mask,
                                          float* __restrict__ colsum_part) {
  const int OH = H >> 1, OW = W >> 1, CV = C >> 3;
  const size_t total = (size_t)N * OH * OW * CV;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV);
    size_t r = i / CV;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    const size_t base = (((size_t)n * H + 2 * oh) * W + 2 * ow) * C + cv * 8;
    const size_t o01 = C, o10 = (size_t)W * C, o11 = (size_t)W * C + C;
    const uint32_t code = idx[i];
    float g[8];
    unpack8(*(const u32x4_t*)(dy + (((size_t)n * OH + oh) * OW + ow) * C + cv * 8), g);
    float ra[8], rb[8], rc[8], rd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t c4 = (code >> (4 * e)) & 15u;
      const uint32_t arg = c4 & 3u;
      const float gv = (relu_mask && !(c4 & 4u)) ? 0.f : g[e];
      ra[e] = arg == 0 ? gv : 0.f; rb[e] = arg == 1 ? gv : 0.f;
      rc[e] = arg == 2 ? gv : 0.f; rd[e] = arg == 3 ? gv : 0.f;
      cs[e] += gv;
    }
    u32x4_t oa, ob, oc, od;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      oa[e] = (__float_as_uint(ra[2 * e]) >> 16) | (__float_as_uint(ra[2 * e + 1]) & 0xFFFF0000u);
      ob[e] = (__float_as_uint(rb[2 * e]) >> 16) | (__float_as_uint(rb[2 * e + 1]) & 0xFFFF0000u);
      oc[e] = (__float_as_uint(rc[2 * e]) >> 16) | (__float_as_uint(rc[2 * e + 1]) & 0xFFFF0000u);
      od[e] = (__float_as_uint(rd[2 * e]) >> 16) | (__float_as_uint(rd[2 * e + 1]) & 0xFFFF0000u);
    }
    *(u32x4_t*)(dx + base) = oa;
    *(u32x4_t*)(dx + base + o01) = ob;
    *(u32x4_t*)(dx + base + o10) = oc;
    *(u32x4_t*)(dx + base + o11) = od;
  }
  if (colsum_part) {   // kernel-uniform
    __shared__ float red[256 * 8];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = cs[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int cv = c >> 3, e = c & 7;
      float s = 0.f;
      for (int t = cv; t < 256; t += CV) s += red[t * 8 + e];
      colsum_part[(size_t)blockIdx.x * C + c] = s;
    }
  }
}

// Odd H or W: the dropped last row/column gets zero gradient.
__global__ void zero_tail_kernel(bf16_t* __restrict__ dx, int N, int H, int W, int C) {
  const int OH2 = (H >> 1) * 2, OW2 = (W >> 1) * 2;
  const size_t total = (size_t)N * H * W * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    if (h >= OH2 || w >= OW2) dx[i] = 0;
  }
}

DTP_DEVINL int win_start(int o, int in, int out) { return (o * in) / out; }
DTP_DEVINL int win_end(int o, int in, int out) { return ((o + 1) * in + out - 1) / out; }

// y[n][c*OH*OW + oh*OW + ow] = mean over the adaptive window of x[n][h][w][c]   (NHWC -> CHW-flat)
__global__ void avgpool_flatten_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H,
                                           int W, int C, int OH, int OW) {
  const size_t total = (size_t)N * OH * OW * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);          // c fastest: coalesced NHWC reads
    size_t r = i / C;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    const int h0 = win_start(oh, H, OH), h1 = win_end(oh, H, OH);
    const int w0 = win_start(ow, W, OW), w1 = win_end(ow, W, OW);
    float s = 0.f;
    for (int h = h0; h < h1; ++h)
      for (int w = w0; w < w1; ++w) s += bf16_to_f32(x[(((size_t)n * H + h) * W + w) * C + c]);
    s /= (float)((h1 - h0) * (w1 - w0));
    y[(size_t)n * C * OH * OW + (size_t)c * OH * OW + oh * OW + ow] = f32_to_bf16(s);
  }
}

__global__ void avgpool_flatten_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N, int H,
                                           int W, int C, int OH, int OW) {
  const size_t total = (size_t)N * H * W * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float s = 0.f;
    for (int oh = 0; oh < OH; ++oh) {
      const int h0 = win_start(oh, H, OH), h1 = win_end(oh, H, OH);
      if (h < h0 || h >= h1) continue;
      for (int ow = 0; ow < OW; ++ow) {
        const int w0 = win_start(ow, W, OW), w1 = win_end(ow, W, OW);
        if (w < w0 || w >= w1) continue;
        s += bf16_to_f32(dy[(size_t)n * C * OH * OW + (size_t)c * OH * OW + oh * OW + ow]) /
             (float)((h1 - h0) * (w1 - w0));
      }
    }
    dx[i] = f32_to_bf16(s);
  }
}

static inline int grid_for(size_t work, int block = 256, int cap = 8192) {
  size_t g = (work + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// idx: null, or N*(H/2)*(W/2)*(C/8) uint32 codes for dtp_maxpool2x2_bwd (x == null form)
extern "C" int dtp_maxpool2x2_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, hipStream_t s) {
  if (N <= 0 || H < 2 || W < 2 || (C & 7)) return -10;
  const size_t work = (size_t)N * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3(grid_for(work)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y,
                     (uint32_t*)idx, N, H, W, C);
  return DTP_CHECK_LAUNCH();
}

// colsum_part: null, or room for max_parts*C floats; returns the number of partial rows written
// through *nparts (0 when not requested)
// exactly one of x (pool input) / idx (codes written by the forward pass) is given
extern "C" int dtp_maxpool2x2_bwd(const void* x, const void* idx, const void* dy, void* dx, int N, int H, int W,
                                  int C, int relu_mask, float* colsum_part, int max_parts, int* nparts,
                                  hipStream_t s) {
  if (N <= 0 || H < 2 || W < 2 || (C & 7) || ((x == nullptr) == (idx == nullptr))) return -10;
  if (nparts) *nparts = 0;
  if (colsum_part && (max_parts < 1 || (256 % (C >> 3)) != 0)) return -11;
  if ((H & 1) || (W & 1)) {
    hipLaunchKernelGGL(zero_tail_kernel, dim3(grid_for((size_t)N * H * W * C)), dim3(256), 0, s, (bf16_t*)dx, N, H,
                       W, C);
  }
  const size_t work = (size_t)N * (H / 2) * (W / 2) * (C / 8);
  int grid = grid_for(work);
  if (colsum_part) {
    if (grid > max_parts) grid = max_parts;
    if (nparts) *nparts = grid;
  }
  if (idx)
    hipLaunchKernelGGL(maxpool2x2_bwd_idx_kernel, dim3(grid), dim3(256), 0, s, (const uint32_t*)idx,
                       (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, relu_mask, colsum_part);
  else
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, relu_mask, colsum_part);
  return DTP_CHECK_LAUNCH();
}

extern "C" int dtp_avgpool_flatten_fwd(const void* x, void* y, int N, int H, int W, int C, int OH, int OW,
                                       hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return -10;
  hipLaunchKernelGGL(avgpool_flatten_fwd_kernel, dim3(grid_for((size_t)N * OH * OW * C)), dim3(256), 0, s,
                     (const bf16_t*)x, (bf16_t*)y, N, H, W, C, OH, OW);
  return DTP_CHECK_LAUNCH();
}

extern "C" int dtp_avgpool_flatten_bwd(const void* dy, void* dx, int N, int H, int W, int C, int OH, int OW,
                                       hipStream_t s) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return -10;
  hipLaunchKernelGGL(avgpool_flatten_bwd_kernel, dim3(grid_for((size_t)N * H * W * C)), dim3(256), 0, s,
                     (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, OH, OW);
  return DTP_CHECK_LAUNCH();
}
