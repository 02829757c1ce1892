// First VGG layer (Cin = 3): image ingest + 3x3 conv + bias + ReLU, and its weight/bias gradient.
//
// K = 9*Cin = 27 is far too shallow for the generic 64-deep implicit-GEMM K loop (padding Cin to
// 64 costs 21x the MFMA work and a 128 B/pixel ingest write), so this layer gets dedicated kernels:
//
// fprop: a workgroup takes 256 consecutive output pixels, builds their im2col rows [256][32] bf16
//   (k = tap*3 + c, 27 real + 5 zero) in LDS straight from the NCHW f32/bf16/u8 batch — the
//   (x*scale - mean)/std normalisation and the NCHW->NHWC layout change are fused here, no
//   converted copy of the input is ever written — then ONE K=32 MFMA step per 16x16 tile against
//   the [64][32] weight image, bias+ReLU, and full-row coalesced NHWC bf16 stores.
// wgrad: dW[64][27] = sum_p dY[p][64] (x) im2col[p][27]; reduction over pixels.  dY tiles are
//   staged row-major by LDS-DMA and read transposed (ds_read_b64_tr_b16); the im2col tile is
//   written transposed [k][pixel].  Column k = 27 of the im2col tile is set to 1.0, so the bias
//   gradient (column sum of dY) drops out of the same MFMAs for free.  A persistent grid
//   accumulates in registers over many tiles; per-workgroup partials are summed by a tiny
//   deterministic reduce kernel (no float atomics).
//
// Reference parity: first nn.Conv2d(3, 64) + ReLU of model/vgg16.py:24 (cuDNN fwd / bwd-filter)
// and the blocking H2D + dataset normalisation feeding it.
#include "dtp_common.h"
#include <algorithm>
#include <cstdlib>

#define C1_COUT 64
#define C1_TILE 256

struct Conv1Params {
  const void* x;          // [N][Cin][H][W]
  const bf16_t* w;        // [64][9*Cin] bf16 (tap-major, channel-minor)
  const float* bias;      // [64]
  bf16_t* y;              // [N][H][W][64]
  const float* mean;      // [Cin] or null
  const float* inv_std;   // [Cin] or null
  float scale;
  int N, Cin, H, W, relu;
  long M;                 // N*H*W          (< 2^31, and the batch < 4 GiB: the launchers chunk)
  unsigned hw;            // H*W
  float rcp_hw, rcp_w;    // 1/(H*W), 1/W for the float-estimate divisions
  float* stats_part;      // STATS kernels: [gridDim.x][2][64] f32 (sum y, sum y^2) of the stored outputs — batch
                          // statistics of a BatchNorm that follows (bn.hip partial layout)
};

template <typename T> DTP_DEVINL float ld_f32(const T* p);
template <> DTP_DEVINL float ld_f32<float>(const float* p) { return *p; }
template <> DTP_DEVINL float ld_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <> DTP_DEVINL float ld_f32<uint8_t>(const uint8_t* p) { return (float)*p; }

// n / d and n % d for n / d < 2^22 from a float estimate and one correction step (the compiler's
// exact 32-bit division is ~20 instructions with four quarter-rate multiplies, its 64-bit one ~10x
// that: the two divisions per pixel were a third of the VALU work of these kernels, which are
// VALU-issue-bound, not bandwidth-bound — three workgroups per CU instead of two changed nothing).
DTP_DEVINL unsigned udiv_est(unsigned n, unsigned d, float rcp_d, unsigned& rem) {
  unsigned q = (unsigned)((float)n * rcp_d);
  int r = (int)(n - q * d);
  if (r < 0) { q -= 1; r += (int)d; }
  else if (r >= (int)d) { q += 1; r -= (int)d; }
  rem = (unsigned)r;
  return q;
}

// per-channel affine form of the ingest normalisation: v = raw * a[c] + b[c]
struct Conv1Norm { float a[3], b[3]; };
DTP_DEVINL Conv1Norm load_norm(const Conv1Params& p) {
  Conv1Norm nm;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float mean = p.mean ? p.mean[c] : 0.f, istd = p.mean ? p.inv_std[c] : 1.f;
    nm.a[c] = p.scale * istd;
    nm.b[c] = -mean * istd;
  }
  return nm;
}

// im2col row of one pixel: v[k], k = tap*3 + c (27 values), zero outside the image.
// All 27 loads are UNCONDITIONAL on clamped coordinates and the border / tail zeroing is a select
// afterwards: with a branch around each load the compiler serialised them (one global-load
// latency per tap, ~13 us per workgroup measured); now they issue back to back.
// Two phases so that a persistent kernel can have the next tile's 27 loads in flight while it works
// on the current one: gather_issue() leaves the raw values in registers and returns the pixel's
// packed coordinates (h | w << 16; h = 0x7ff0 for a pixel past the end, which fails every row test),
// gather_finish() normalises (one FMA per value) and zeroes.
// Addresses are a uniform base + 32-bit byte offsets (global_load with an SGPR base).
template <typename T>
DTP_DEVINL uint32_t gather_issue(const Conv1Params& p, long pix, float* raw) {
  const bool inb = pix < p.M;
  const unsigned pc = inb ? (unsigned)pix : 0u;
  unsigned rem, w;
  const unsigned n = udiv_est(pc, p.hw, p.rcp_hw, rem);
  const unsigned h = udiv_est(rem, (unsigned)p.W, p.rcp_w, w);
  const unsigned cs = p.hw * (unsigned)sizeof(T);                 // channel stride, bytes (uniform)
  const unsigned base = n * 3u * cs;
  unsigned roff[3], coff[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int hh = (int)h + d - 1, ww = (int)w + d - 1;
    const int hc = hh < 0 ? 0 : (hh >= p.H ? p.H - 1 : hh);
    const int wc = ww < 0 ? 0 : (ww >= p.W ? p.W - 1 : ww);
    roff[d] = base + (unsigned)hc * (unsigned)p.W * (unsigned)sizeof(T);
    coff[d] = (unsigned)wc * (unsigned)sizeof(T);
  }
  const char* xb = (const char*)p.x;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const unsigned o = roff[t / 3] + coff[t % 3];
#pragma unroll
    for (int c = 0; c < 3; ++c) raw[t * 3 + c] = ld_f32<T>((const T*)(xb + (size_t)(o + (unsigned)c * cs)));
  }
  return inb ? (h | (w << 16)) : 0x7ff0u;
}

DTP_DEVINL void gather_finish(const Conv1Params& p, const Conv1Norm& nm, const float* raw, uint32_t hwp, float* v) {
  const int h = (int)(hwp & 0xFFFFu), w = (int)(hwp >> 16);
  bool rok[3], cok[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    rok[d] = (unsigned)(h + d - 1) < (unsigned)p.H;
    cok[d] = (unsigned)(w + d - 1) < (unsigned)p.W;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const bool ok = rok[t / 3] && cok[t % 3];
#pragma unroll
    for (int c = 0; c < 3; ++c) v[t * 3 + c] = ok ? __builtin_fmaf(raw[t * 3 + c], nm.a[c], nm.b[c]) : 0.f;
  }
}

template <typename T>
DTP_DEVINL void gather_patch(const Conv1Params& p, const Conv1Norm& nm, long pix, float* v) {
  float raw[27];
  const uint32_t hwp = gather_issue<T>(p, pix, raw);
  gather_finish(p, nm, raw, hwp, v);
}

// Persistent: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... with the NEXT tile's 27
// gather loads in flight (registers) while the current tile goes A image -> MFMA -> C image ->
// stores.  The one-shot form (one tile per workgroup, 3 workgroups per CU by LDS) was a serial
// load-latency chain per workgroup: 3.3 TB/s of output at batch 1024.  The gather is issued BEFORE
// the tile's stores so that waiting for it (vmcnt counts in issue order) never drains the stores.
template <typename T, bool STATS = false>
__global__ __launch_bounds__(256) void conv1_fprop_kernel(const Conv1Params p, const int ntiles) {
  // LDS: A image [256][32] bf16 (64-B rows, 16 KiB) | W image [64][32] (4 KiB) | C image [256][64] (32 KiB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lA = smem;
  char* lW = smem + 256 * 64;
  char* lC = lW + 64 * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = 9 * p.Cin;

  const Conv1Norm nm = load_norm(p);
  float raw[27];
  uint32_t hwp = gather_issue<T>(p, (long)blockIdx.x * C1_TILE + tid, raw);

  {  // weights: thread -> (row = tid>>2, 8-wide k chunk = tid&3)
    const int row = tid >> 2, ch = tid & 3;
    u32x4_t q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = ch * 8 + 2 * e;
      const uint32_t lo = k0 < K ? p.w[row * K + k0] : 0;
      const uint32_t hi = k0 + 1 < K ? p.w[row * K + k0 + 1] : 0;
      q[e] = lo | (hi << 16);
    }
    *(u32x4_t*)(lW + row * 64 + ((ch ^ ((row >> 1) & 3)) << 4)) = q;
  }
  const int fr = lane & 15, fk = lane >> 4;
  const int foff = fr * 64 + ((fk ^ ((fr >> 1) & 3)) << 4);
  const int cq = (lane >> 4) * 4;
  // bias as the MFMA's C operand, ReLU after the rounding as a packed signed-16-bit max on the bf16
  // bit patterns (as in igemm.hip's epilogue)
  f32x4_t bvv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bvv[j] = p.bias ? *(const f32x4_t*)(p.bias + j * 16 + cq) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  const bool relu = p.relu != 0;
  // STATS: a thread stores the same 8-channel chunk (tid & 7) of 8 pixels per tile
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    {  // im2col row of this thread's pixel (loads issued one tile ago)
      float v[32];
#pragma unroll
      for (int k = 27; k < 32; ++k) v[k] = 0.f;
      gather_finish(p, nm, raw, hwp, v);
      const int sw = (tid >> 1) & 3;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        u32x4_t q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = pack_bf16x2(v[ch * 8 + 2 * e], v[ch * 8 + 2 * e + 1]);
        *(u32x4_t*)(lA + tid * 64 + ((ch ^ sw) << 4)) = q;
      }
    }
    if (tile + (int)gridDim.x < ntiles)       // next tile's gather: in flight across everything below
      hwp = gather_issue<T>(p, (long)(tile + gridDim.x) * C1_TILE + tid, raw);
    __syncthreads();                          // A (and, first time, W) image complete; previous C image drained

    bf16x8_t xf[4], wf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[i] = *(const bf16x8_t*)(lA + (wave * 64 + i * 16) * 64 + foff);
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8_t*)(lW + (j * 16) * 64 + foff);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4_t acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], bvv[j], 0, 0, 0);
        uint32_t lo = pack_bf16x2(acc[0], acc[1]), hi = pack_bf16x2(acc[2], acc[3]);
        if (relu) {
          lo = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, lo), (s16x2_t){0, 0}));
          hi = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, hi), (s16x2_t){0, 0}));
        }
        const int pl = wave * 64 + i * 16 + fr;       // pixel in tile
        const int nl = j * 16 + cq;                   // first of 4 channels
        const u32x2_t pk = {lo, hi};
        *(u32x2_t*)(lC + pl * 128 + (((nl >> 3) ^ (pl & 7)) << 4) + ((nl >> 2) & 1) * 8) = pk;
      }
    }
    __syncthreads();                          // C image complete; every wave is done with the A image
    const long m0 = (long)tile * C1_TILE;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int id = tid + r * 256;
      const int pl = id >> 3, q = id & 7;
      if (m0 + pl >= p.M) continue;
      const u32x4_t v = *(const u32x4_t*)(lC + pl * 128 + ((q ^ (pl & 7)) << 4));
      *(u32x4_t*)(p.y + (m0 + pl) * 64 + q * 8) = v;
      if constexpr (STATS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xFFFF0000u);
          cs[2 * e] += lo;
          cs[2 * e + 1] += hi;
          sq[2 * e] = fmaf(lo, lo, sq[2 * e]);
          sq[2 * e + 1] = fmaf(hi, hi, sq[2 * e + 1]);
        }
      }
    }
  }
  if constexpr (STATS) {
    // 32 threads (tid >> 3) share a chunk: fold them through LDS in a fixed order (the C image is dead)
    float* red = (float*)lC;                  // [32][64]
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(tid >> 3) * 64 + (tid & 7) * 8 + e] = which ? sq[e] : cs[e];
      __syncthreads();
      if (tid < 64) {
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += red[k * 64 + tid];
        p.stats_part[((size_t)blockIdx.x * 2 + which) * 64 + tid] = t;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ wgrad
struct Conv1WgradParams {
  Conv1Params c;          // x / geometry / normalisation (w, bias, y unused)
  const char* dy;         // [M][64] bf16
  float* partial;         // [gridDim.x][64][32] f32
  const char* zero_page;
  int ntiles;
  // self-masking (optional): recompute sign(conv1(x) + bias) for the tile from the im2col rows the
  // kernel builds anyway (one K=32 MFMA step per 16x16 block) and zero dY where the forward ReLU
  // was inactive.  The consumer's data-gradient kernel then skips its mask read of the 64-channel
  // full-resolution activation (1.6 GB at batch 256).
  const bf16_t* mask_w;     // [64][27] bf16 forward weights, or null
  const float* mask_bias;   // [64] or null
};

// XOR swizzle (in 16-byte chunks) of the weight-gradient kernel's 64-byte-row LDS images.  Found by
// exhaustive search over the GF(2)-linear maps of the low four row bits against the lane groups of
// MI355X_MICROARCH.md's LDS table: conflict-free for (i) the ds_read_b128 fragment reads of the
// self-masking forward, (ii) the ds_write_b128 of the image rows and (iii) the transposed
// ds_read_b64_tr_b16 reads of the gradient MFMAs, whose 16-lane groups sit 8 rows apart — the
// forward kernel's (row >> 1) & 3 satisfies (i) and (ii) only (2-way conflicts on (iii)).
DTP_DEVINL int s64w(int row) { return ((row >> 2) & 1) | ((((row >> 1) ^ (row >> 2) ^ (row >> 3)) & 1) << 1); }

template <typename T>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const Conv1WgradParams p) {
  // LDS: dY image [256 px][64] bf16 (128-B rows, tr-read swizzle) 32 KiB | A image [256 px][32 k] bf16
  //      (64-B rows, the im2col rows as in conv1_fprop) 16 KiB | W image [64][32] bf16 4 KiB = 52 KiB,
  //      three workgroups per CU.  Both MFMA operands of the gradient are read TRANSPOSED
  //      (ds_read_b64_tr_b16: 8 consecutive pixels = the reduction index, per lane) — dY from its
  //      image and the im2col operand from the same A image the self-masking forward uses; a second,
  //      k-major copy of the im2col tile (16.5 KiB, 27 two-byte LDS writes per lane) is not needed.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lD = smem;
  char* lA = smem + 256 * 128;
  char* lW = lA + 256 * 64;
  const bool self_mask = p.mask_w != nullptr;
  const Conv1Norm nm = load_norm(p.c);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  typedef __attribute__((ext_vector_type(4))) short s4;

  // dY staging geometry: 8 lanes per 128-B row, 32 rows per pass, 8 passes
  const int drow = tid >> 3, dpc = tid & 7;
  const int dsw = (((drow >> 1) & 1) | (((drow >> 3) & 1) << 1)) << 1;   // s_of<128>(row); row+32*i keeps it
  const int dlc = dpc ^ dsw;

  // fragment read geometry
  const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int krow = 8 * g4 + q;
  const int fsw = (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1;
  const int sub = pp >> 1, half8 = (pp & 1) * 8;

  f32x4_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (self_mask) {   // forward weight image, same layout as conv1_fprop_kernel; lives for the whole kernel
    const int K = 9 * p.c.Cin;
    const int row = tid >> 2, ch = tid & 3;
    u32x4_t qv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = ch * 8 + 2 * e;
      const uint32_t lo = k0 < K ? p.mask_w[row * K + k0] : 0;
      const uint32_t hi = k0 + 1 < K ? p.mask_w[row * K + k0 + 1] : 0;
      qv[e] = lo | (hi << 16);
    }
    *(u32x4_t*)(lW + row * 64 + ((ch ^ s64w(row)) << 4)) = qv;
  }

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const long m0 = (long)tile * C1_TILE;
    __syncthreads();   // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long m = m0 + drow + i * 32;
      const char* g = m < p.c.M ? p.dy + (m * 64 + dlc * 8) * 2 : p.zero_page;
      glds16(g, lD + wave * 1024 + i * 4096);
    }
    {  // im2col row [px][32]: k < 27 the patch, k = 27 the ones column (bias gradient; the forward
       // weight image is zero there, so the recomputed activation does not see it), 28..31 zero
      float v[32];
      gather_patch<T>(p.c, nm, m0 + tid, v);
      v[27] = (m0 + tid < p.c.M) ? 1.f : 0.f;
#pragma unroll
      for (int k = 28; k < 32; ++k) v[k] = 0.f;
      const int sw = s64w(tid);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        u32x4_t qv;
#pragma unroll
        for (int e = 0; e < 4; ++e) qv[e] = pack_bf16x2(v[ch * 8 + 2 * e], v[ch * 8 + 2 * e + 1]);
        *(u32x4_t*)(lA + tid * 64 + ((ch ^ sw) << 4)) = qv;
      }
    }
    sync_lds_dma();    // dY tile (LDS-DMA) and the im2col image landed everywhere
    if (self_mask) {
      // y = W * im2col + b for this wave's 64 pixels; zero dY (in its LDS image) where y <= 0.
      // The wave masks exactly the rows it reads below, so a wave-level fence is enough.
      const int fr = lane & 15, fk = lane >> 4;
      const int foff = fr * 64 + ((fk ^ s64w(fr)) << 4);
      const int cq = fk * 4;
      bf16x8_t xf[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[i] = *(const bf16x8_t*)(lA + (wave * 64 + i * 16) * 64 + foff);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8_t*)(lW + (j * 16) * 64 + foff);
      // same arithmetic as conv1_fprop_kernel (bias is the MFMA's C operand) so the recomputed sign
      // is bit-identical to the stored activation's; all bias loads in one batch
      f32x4_t bvv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bvv[j] = p.mask_bias ? *(const f32x4_t*)(p.mask_bias + j * 16 + cq) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4_t y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], bvv[j], 0, 0, 0);
          const int pl = wave * 64 + i * 16 + fr;       // pixel (dY image row)
          const int nl = j * 16 + cq;                   // first of 4 channels
          const int rsw = (((pl >> 1) & 1) | (((pl >> 3) & 1) << 1)) << 1;
          u32x2_t* dp = (u32x2_t*)(lD + pl * 128 + ((((nl >> 3)) ^ rsw) << 4) + ((nl >> 2) & 1) * 8);
          u32x2_t d = *dp;
          // the forward kernel stores bf16(relu(y + b)): active <=> that bf16 value is > 0 <=> y rounds
          // (to nearest even) to a positive bf16 <=> its f32 bit pattern, as a signed integer, is above
          // 0x8000 (2^-134: that and everything below rounds to zero; negatives are negative integers)
          const uint32_t k0 = ((int)__float_as_uint(y[0]) > 0x8000 ? 0xFFFFu : 0u) |
                              ((int)__float_as_uint(y[1]) > 0x8000 ? 0xFFFF0000u : 0u);
          const uint32_t k1 = ((int)__float_as_uint(y[2]) > 0x8000 ? 0xFFFFu : 0u) |
                              ((int)__float_as_uint(y[3]) > 0x8000 ? 0xFFFF0000u : 0u);
          d[0] &= k0;
          d[1] &= k1;
          *dp = d;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    // wave: pixels [64*wave, 64*wave+64) -> two 32-deep k-substeps
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int px0 = wave * 64 + ks * 32;
      bf16x8_t af[4], bf[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = ((2 * i) ^ fsw) + sub;
        const char* a0 = lD + (px0 + krow) * 128 + (ch << 4) + half8;
        s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0));
        s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + 4 * 128));
        af[i] = (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {   // lane li: im2col index j*16 + li, pixels px0 + 8*g4 + 0..7
        const int r0 = px0 + krow, r1 = r0 + 4;
        const char* b0 = lA + r0 * 64 + (((2 * j + sub) ^ s64w(r0)) << 4) + half8;
        const char* b1 = lA + r1 * 64 + (((2 * j + sub) ^ s64w(r1)) << 4) + half8;
        s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(b0));
        s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(b1));
        bf[j] = (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  // cross-wave sum through LDS, then one partial [64][32] per workgroup
  __syncthreads();
  float* red = (float*)smem;   // [4 waves][64][32] f32 = 32 KiB
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)   // D row = cout (A operand), col = k (B operand)
        red[wave * 2048 + (i * 16 + g4 * 4 + e) * 32 + j * 16 + li] = acc[i][j][e];
  __syncthreads();
  for (int o = tid; o < 2048; o += 256)
    p.partial[(size_t)blockIdx.x * 2048 + o] = (red[o] + red[2048 + o]) + (red[4096 + o] + red[6144 + o]);
}

// out_w[co][k] (k < K) and out_b[co] (k == 27) = beta*old + sum_g partial[g][co][k]
__global__ void conv1_wgrad_reduce_kernel(const float* __restrict__ partial, int G, int K, void* __restrict__ out_w,
                                          void* __restrict__ out_b, int out_fp32, float beta_w, float beta_b) {
  // 16 outputs x 16 slab groups per workgroup, four independent accumulators per thread, fixed-order
  // fold through LDS (this kernel is the tail of backward: one accumulator walking G/4 slabs per
  // thread was ~80 us of exposed load latency)
  __shared__ float red[16][16];
  const int ol = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + ol;          // 0..2047 = co*32 + k
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int g = grp;
  for (; g + 48 < G; g += 64) {
    s0 += partial[(size_t)g * 2048 + o];
    s1 += partial[(size_t)(g + 16) * 2048 + o];
    s2 += partial[(size_t)(g + 32) * 2048 + o];
    s3 += partial[(size_t)(g + 48) * 2048 + o];
  }
  for (; g < G; g += 16) s0 += partial[(size_t)g * 2048 + o];
  red[grp][ol] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp != 0) return;
  float s = red[0][ol];
#pragma unroll
  for (int k = 1; k < 16; ++k) s += red[k][ol];
  const int co = o >> 5, k = o & 31;
  void* dst; int idx; float beta;
  if (k < K) { dst = out_w; idx = co * K + k; beta = beta_w; }
  else if (k == 27 && out_b) { dst = out_b; idx = co; beta = beta_b; }
  else return;
  if (out_fp32) {
    float* d = (float*)dst;
    d[idx] = (beta != 0.f ? beta * d[idx] : 0.f) + s;
  } else {
    bf16_t* d = (bf16_t*)dst;
    d[idx] = f32_to_bf16((beta != 0.f ? beta * bf16_to_f32(d[idx]) : 0.f) + s);
  }
}

static int fill_params(Conv1Params& c, const void* x, int N, int Cin, int H, int W, float scale, const float* mean,
                       const float* inv_std) {
  if (N <= 0 || Cin != 3 || H <= 0 || W <= 0) return -10;   // k = tap*3 + c is baked in
  c.x = x; c.N = N; c.Cin = Cin; c.H = H; c.W = W; c.scale = scale; c.mean = mean; c.inv_std = inv_std;
  c.M = (long)N * H * W;
  c.hw = (unsigned)H * (unsigned)W;
  c.rcp_hw = 1.0f / (float)c.hw;
  c.rcp_w = 1.0f / (float)W;
  c.w = nullptr; c.bias = nullptr; c.y = nullptr; c.relu = 0;
  return 0;
}

// Largest batch one launch may cover: pixel index < 2^31, byte offsets into x < 2^32, image index
// < 2^22 (udiv_est).  Larger batches are processed in chunks by the launchers below.
static int conv1_max_batch(int H, int W, int esz) {
  const unsigned long long hw = (unsigned long long)H * W;
  unsigned long long n = 0x7fffffffULL / hw;
  n = std::min(n, 0xffffffffULL / (3ULL * hw * (unsigned long long)esz));
  n = std::min(n, (unsigned long long)(1 << 22));
  return (int)n;
}
static int conv1_esz(int in_kind) { return in_kind == 0 ? 4 : (in_kind == 1 ? 2 : 1); }

extern "C" int dtp_conv1_fprop(const void* x, int in_kind, const void* w, const float* bias, void* y, int N, int Cin,
                               int H, int W, int relu, float scale, const float* mean, const float* inv_std,
                               float* stats_part, int stats_rows, int* nparts_out, hipStream_t s) {
  if (in_kind < 0 || in_kind > 2) return -11;
  if (stats_part && !nparts_out) return -14;
  int rows_used = 0;
  if (N <= 0 || Cin != 3 || H <= 0 || W <= 0 || H >= 0x7ff0 || W >= 0x7ff0) return -10;
  const int esz = conv1_esz(in_kind), nmax = conv1_max_batch(H, W, esz);
  if (nmax < 1) return -13;
  constexpr int lds = 256 * 64 + 64 * 64 + 256 * 128;
  // 52 KiB of LDS: 3 workgroups per CU; 768 persistent workgroups fill the 256 CUs
  static const int cap = getenv("DTP_CONV1_GRID") ? atoi(getenv("DTP_CONV1_GRID")) : 768;
  for (int n0 = 0; n0 < N; n0 += nmax) {
    const int nc = std::min(nmax, N - n0);
    Conv1Params c;
    int rc = fill_params(c, (const char*)x + (size_t)n0 * 3 * H * W * esz, nc, Cin, H, W, scale, mean, inv_std);
    if (rc) return rc;
    c.w = (const bf16_t*)w; c.bias = bias; c.y = (bf16_t*)y + (size_t)n0 * H * W * 64; c.relu = relu;
    const int tiles = (int)((c.M + C1_TILE - 1) / C1_TILE);
    const int grid = tiles < cap ? tiles : cap;
    c.stats_part = nullptr;
    if (stats_part) {            // one partial row per workgroup, chunk after chunk
      if (rows_used + grid > stats_rows) return -15;
      c.stats_part = stats_part + (size_t)rows_used * 128;
      rows_used += grid;
      if (in_kind == 0) hipLaunchKernelGGL((conv1_fprop_kernel<float, true>), dim3(grid), dim3(256), lds, s, c, tiles);
      else if (in_kind == 1) hipLaunchKernelGGL((conv1_fprop_kernel<bf16_t, true>), dim3(grid), dim3(256), lds, s, c, tiles);
      else hipLaunchKernelGGL((conv1_fprop_kernel<uint8_t, true>), dim3(grid), dim3(256), lds, s, c, tiles);
      continue;
    }
    if (in_kind == 0) hipLaunchKernelGGL(conv1_fprop_kernel<float>, dim3(grid), dim3(256), lds, s, c, tiles);
    else if (in_kind == 1) hipLaunchKernelGGL(conv1_fprop_kernel<bf16_t>, dim3(grid), dim3(256), lds, s, c, tiles);
    else hipLaunchKernelGGL(conv1_fprop_kernel<uint8_t>, dim3(grid), dim3(256), lds, s, c, tiles);
  }
  if (stats_part) *nparts_out = rows_used;
  return DTP_CHECK_LAUNCH();
}

// partial: >= grid*2048 floats (grid chosen here, <= max_grid)
extern "C" int dtp_conv1_wgrad(const void* x, int in_kind, const void* dy, void* out_w, void* out_b, float* partial,
                               int max_grid, const void* zero_page, int N, int Cin, int H, int W, float scale,
                               const float* mean, const float* inv_std, int out_fp32, float beta_w, float beta_b,
                               const void* mask_w, const float* mask_bias, hipStream_t s) {
  if (in_kind < 0 || in_kind > 2) return -11;
  if (N <= 0 || Cin != 3 || H <= 0 || W <= 0 || H >= 0x7ff0 || W >= 0x7ff0) return -10;
  const int esz = conv1_esz(in_kind), nmax = conv1_max_batch(H, W, esz);
  if (nmax < 1) return -13;
  constexpr int lds = 256 * 128 + 256 * 64 + 64 * 64;
  for (int n0 = 0; n0 < N; n0 += nmax) {     // chunks accumulate into the outputs (beta = 1 after the first)
    const int nc = std::min(nmax, N - n0);
    Conv1WgradParams p;
    p.mask_w = (const bf16_t*)mask_w; p.mask_bias = mask_bias;
    int rc = fill_params(p.c, (const char*)x + (size_t)n0 * 3 * H * W * esz, nc, Cin, H, W, scale, mean, inv_std);
    if (rc) return rc;
    p.dy = (const char*)dy + (size_t)n0 * H * W * 128; p.partial = partial; p.zero_page = (const char*)zero_page;
    p.ntiles = (int)((p.c.M + C1_TILE - 1) / C1_TILE);
    int grid = p.ntiles < max_grid ? p.ntiles : max_grid;
    if (grid < 1) return -12;
    if (in_kind == 0) hipLaunchKernelGGL(conv1_wgrad_kernel<float>, dim3(grid), dim3(256), lds, s, p);
    else if (in_kind == 1) hipLaunchKernelGGL(conv1_wgrad_kernel<bf16_t>, dim3(grid), dim3(256), lds, s, p);
    else hipLaunchKernelGGL(conv1_wgrad_kernel<uint8_t>, dim3(grid), dim3(256), lds, s, p);
    hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(128), dim3(256), 0, s, partial, grid, 9 * Cin, out_w, out_b,
                       out_fp32, n0 ? 1.f : beta_w, n0 ? 1.f : beta_b);
  }
  return DTP_CHECK_LAUNCH();
}
