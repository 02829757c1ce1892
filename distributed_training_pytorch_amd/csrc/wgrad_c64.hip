// Weight gradient of 3x3 convolutions with 64 input channels (VGG conv1_2: 64->64 @224^2,
// conv2_1: 64->128 @112^2) for gfx950.
//
// The generic TT kernel (wgrad.hip) gives one (tap, channel-block) output tile to a workgroup, so a
// 64x64 tile re-reads 16 KB of activations per 32 MFMAs and all nine taps re-read the same input:
// at 64 channels it is bound by the LDS fill rate of a CU (measured 207 TF/s).  This kernel keeps
// ALL NINE TAPS of a [COUT x 9*64] gradient in one workgroup's accumulators and walks down an image
// column strip of 32 pixels: per step it loads ONE new input row segment (34 px incl. halo) into a
// 4-slot LDS ring — rows h-1, h, h+1 are already resident — plus the dY row segment, and issues
// 36 MFMAs per wave (4 output-channel blocks x 9 taps).  Each activation byte is fetched from
// global memory ~1.06x instead of 9x.
//   * operands are staged row-major by LDS-DMA and read transposed (ds_read_b64_tr_b16), with the
//     conflict-free XOR swizzle of wgrad.hip applied on the global source address;
//   * image borders / strip tails are redirected per lane to a zeroed page;
//   * a persistent grid accumulates over many (image, strip, row-chunk) work items; per-workgroup
//     f32 partials are summed by a deterministic reduce kernel (beta / dtype cast fused).
#include "dtp_common.h"

struct WgC64Params {
  const char* dY;        // [N*H*W][COUT] bf16
  const char* X;         // [N*H*W][64]   bf16
  float* partial;        // [gridDim.x][COUT][576] f32
  const char* zero_page;
  int N, H, W;
  int S;                 // 32-pixel strips per image row
  int RC;                // row chunks per strip
  int rows_per_chunk;
  int nchunks;
};

DTP_DEVINL int sw128(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }
DTP_DEVINL int sw256(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }

template <int COUT>
__global__ __launch_bounds__(COUT * 4) void wgrad_c64_kernel(const WgC64Params p) {
  constexpr int T = COUT * 4;
  constexpr int NW = T / 64;              // 4 or 8 waves
  constexpr int RBA = COUT * 2;           // dY LDS row bytes (128 / 256)
  constexpr int DY_BYTES = 32 * RBA;
  constexpr int DY_WI = DY_BYTES / 1024;  // 1-KiB LDS-DMA pieces per dY tile (4 / 8)
  constexpr int X_SLOT = 40 * 128;        // 34 px + 6 pad rows
  constexpr int X_WI = 5;
  typedef __attribute__((ext_vector_type(4))) short s4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lX = smem;                        // 4 ring slots
  char* lD = smem + 4 * X_SLOT;           // 2 dY buffers

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- per-lane staging geometry
  // X piece wi (0..4): row pos = wi*8 + (lane>>3), physical chunk lane&7
  const int xr = lane >> 3, xpc = lane & 7;
  // dY piece di: COUT=64: 8 rows of 8 chunks; COUT=128: 4 rows of 16 chunks
  constexpr int D_LPR = RBA / 16;         // lanes per dY row (8 / 16)
  const int dr = lane / D_LPR, dpc = lane % D_LPR;

  // ---- per-lane fragment geometry
  const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int krow = 8 * g4 + q;
  const int sub = pp >> 1, half8 = (pp & 1) * 8;
  const int sa = (RBA >= 256) ? sw256(krow) : sw128(krow);
  const int a_cw = (wm * 64) >> 3;
  // X fragment byte offsets inside a ring slot for dx = -1, 0, +1 and the two 4-row reads
  int xo[3][2];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int pos = krow + d + 4 * r;   // = krow + dx + 1 (+4)
      xo[d][r] = pos * 128 + ((((2 * wn) ^ sw128(pos)) + sub) << 4) + half8;
    }

  // LDS byte addresses for the asm transposed reads (see dtp_common.h: the builtin form made hipcc
  // drain the LDS-DMA of the next row before the first fragment read of this one)
  const unsigned lds0 = lds_addr(smem);
  unsigned a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = lds0 + 4 * X_SLOT + krow * RBA + ((((a_cw + 2 * i) ^ sa) + sub) << 4) + half8;

  f32x4_t acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int chunk = blockIdx.x; chunk < p.nchunks; chunk += gridDim.x) {
    const int unit = chunk / p.RC, rc = chunk - unit * p.RC;
    const int n = unit / p.S, s = unit - n * p.S;
    const int w0 = s * 32;
    const int h0 = rc * p.rows_per_chunk;
    int h1 = h0 + p.rows_per_chunk;
    if (h1 > p.H) h1 = p.H;
    const size_t img_px = (size_t)n * p.H * p.W;

    auto stage_x = [&](int hrow) {       // input row segment [w0-1, w0+33) of row hrow -> ring slot
      char* slot = lX + ((hrow + 1) & 3) * X_SLOT;
      const bool row_ok = hrow >= 0 && hrow < p.H;
#pragma unroll
      for (int k = 0; k < (X_WI + NW - 1) / NW; ++k) {
        const int wi = wave + k * NW;
        if (wi < X_WI) {
          const int pos = wi * 8 + xr;
          const int w = w0 - 1 + pos;
          const bool ok = row_ok && pos < 34 && w >= 0 && w < p.W;
          const int lc = xpc ^ sw128(pos);
          const char* g = ok ? p.X + ((img_px + (size_t)hrow * p.W + w) * 64 + lc * 8) * 2 : p.zero_page;
          glds16(g, slot + wi * 1024);
        }
      }
    };
    auto stage_dy = [&](int hrow) {      // dY row segment [w0, w0+32) of row hrow -> buffer hrow&1
      char* buf = lD + (hrow & 1) * DY_BYTES;
#pragma unroll
      for (int k = 0; k < (DY_WI + NW - 1) / NW; ++k) {
        const int di = wave + k * NW;
        if (di < DY_WI) {
          const int px = di * (64 / D_LPR) + dr;
          const int lc = dpc ^ ((RBA >= 256) ? sw256(px) : sw128(px));
          const bool ok = (w0 + px) < p.W;
          const char* g = ok ? p.dY + ((img_px + (size_t)hrow * p.W + w0 + px) * COUT + lc * 8) * 2 : p.zero_page;
          glds16(g, buf + di * 1024);
        }
      }
    };

    __syncthreads();                     // previous chunk completely consumed
    stage_x(h0 - 1);
    stage_x(h0);
    stage_x(h0 + 1);
    stage_dy(h0);
    for (int h = h0; h < h1; ++h) {
      sync_lds_dma();                    // loads for step h landed (explicit vmcnt(0) + barrier); step h-1 consumed
      if (h + 1 < h1) {
        stage_x(h + 2);
        stage_dy(h + 1);
      }
      // dY fragments (8 reads) + the first tap's X fragment (2 reads); then per tap: issue the
      // next tap's two reads, wait for this tap's, 4 MFMAs.  The DMA issued above stays in flight.
      const unsigned dsel = (h & 1) * DY_BYTES;
      u32x2_t al[4], ah[4], bl[2], bh[2];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) DTP_TR2(al[i], ah[i], a_off[i] + dsel, 0, RBA);
      {
        const unsigned slot = lds0 + ((h + 0) & 3) * X_SLOT;
        DTP_TR1(bl[0], slot + xo[0][0]);
        DTP_TR1(bh[0], slot + xo[0][1]);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t + 1 < 9) {
          const int dyi = (t + 1) / 3, dxi = (t + 1) % 3;
          const unsigned slot = lds0 + ((h + dyi) & 3) * X_SLOT;     // row h + dyi - 1 lives in slot (row+1)&3
          DTP_TR1(bl[(t + 1) & 1], slot + xo[dxi][0]);
          DTP_TR1(bh[(t + 1) & 1], slot + xo[dxi][1]);
        }
        if (t == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) DTP_LGKM_WAIT2(2, al[i], ah[i]);
        }
        if (t + 1 < 9) DTP_LGKM_WAIT2(2, bl[t & 1], bh[t & 1]);
        else DTP_LGKM_WAIT2(0, bl[t & 1], bh[t & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8_t bf = frag_of(bl[t & 1], bh[t & 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, frag_of(al[i], ah[i]), acc[i][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- partial slab: lane owns dW[cout = .. + li][tap*64 + wn*16 + g4*4 .. +3]
  float* slab = p.partial + (size_t)blockIdx.x * COUT * 576;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = wm * 64 + i * 16 + li;
#pragma unroll
    for (int t = 0; t < 9; ++t) *(f32x4_t*)(slab + (size_t)m * 576 + t * 64 + wn * 16 + g4 * 4) = acc[i][t];
  }
}

// out = beta*out + sum_g partial[g]   (total % 4 == 0)
__global__ __launch_bounds__(256) void wgrad_c64_reduce_kernel(const float* __restrict__ partial, int G, size_t total,
                                                               void* __restrict__ out, int out_fp32, float beta) {
  // 32 float4 columns x 8 slab groups per workgroup: thread (li, gg) sums slabs g = gg, gg+8, ...
  // (four independent accumulators), the 8 group sums are folded through LDS in a fixed order.
  // (One thread per column walking all <= 512 slabs ran 36-72 workgroups of pure load latency:
  // 40 us alone, ~200 us next to the data-gradient stream.)
  __shared__ f32x4_t red[8][32];
  const size_t nvec = total >> 2;
  const int li = threadIdx.x & 31, gg = threadIdx.x >> 5;
  {
    const size_t i = (size_t)blockIdx.x * 32 + li;
    f32x4_t v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
    if (i < nvec) {
      const float* base = partial + (i << 2);
      int g = gg;
      for (; g + 24 < G; g += 32) {
        v0 += *(const f32x4_t*)(base + (size_t)g * total);
        v1 += *(const f32x4_t*)(base + (size_t)(g + 8) * total);
        v2 += *(const f32x4_t*)(base + (size_t)(g + 16) * total);
        v3 += *(const f32x4_t*)(base + (size_t)(g + 24) * total);
      }
      for (; g < G; g += 8) v0 += *(const f32x4_t*)(base + (size_t)g * total);
    }
    red[gg][li] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (gg != 0 || i >= nvec) return;
    f32x4_t v = red[0][li];
#pragma unroll
    for (int k = 1; k < 8; ++k) v += red[k][li];
    if (out_fp32) {
      float* d = (float*)out + (i << 2);
      if (beta != 0.f) v += *(const f32x4_t*)d * beta;
      *(f32x4_t*)d = v;
    } else {
      bf16_t* d = (bf16_t*)out + (i << 2);
      if (beta != 0.f) {
        const u32x2_t old = *(const u32x2_t*)d;
        v[0] += beta * bf16_to_f32((bf16_t)(old[0] & 0xFFFFu));
        v[1] += beta * bf16_to_f32((bf16_t)(old[0] >> 16));
        v[2] += beta * bf16_to_f32((bf16_t)(old[1] & 0xFFFFu));
        v[3] += beta * bf16_to_f32((bf16_t)(old[1] >> 16));
      }
      u32x2_t pk;
      pk[0] = pack_bf16x2(v[0], v[1]);
      pk[1] = pack_bf16x2(v[2], v[3]);
      *(u32x2_t*)d = pk;
    }
  }
}

// dY [N,H,W,Cout] bf16, X [N,H,W,64] bf16 -> out [Cout][9*64].  partial: >= max_grid*Cout*576 floats.
extern "C" int dtp_wgrad_c64(const void* dY, const void* X, void* out, float* partial, int max_grid,
                             const void* zero_page, int N, int H, int W, int Cout, int out_fp32, float beta,
                             hipStream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || (Cout != 64 && Cout != 128) || max_grid < 1) return -10;
  WgC64Params p;
  p.dY = (const char*)dY; p.X = (const char*)X; p.partial = partial; p.zero_page = (const char*)zero_page;
  p.N = N; p.H = H; p.W = W;
  p.S = (W + 31) / 32;
  const int units = N * p.S;
  // enough row chunks for ~4 work items per workgroup, at least ~16 rows each
  int rc = (4 * max_grid + units - 1) / units;
  if (rc > H / 16) rc = H / 16;
  if (rc < 1) rc = 1;
  p.rows_per_chunk = (H + rc - 1) / rc;
  p.RC = (H + p.rows_per_chunk - 1) / p.rows_per_chunk;
  p.nchunks = units * p.RC;
  const int grid = p.nchunks < max_grid ? p.nchunks : max_grid;
  if (Cout == 64) {
    constexpr int lds = 4 * 40 * 128 + 2 * 32 * 128;
    hipLaunchKernelGGL(wgrad_c64_kernel<64>, dim3(grid), dim3(256), lds, stream, p);
  } else {
    constexpr int lds = 4 * 40 * 128 + 2 * 32 * 256;
    hipLaunchKernelGGL(wgrad_c64_kernel<128>, dim3(grid), dim3(512), lds, stream, p);
  }
  if (DTP_CHECK_LAUNCH() != 0) return -1;
  const size_t total = (size_t)Cout * 576;
  int blocks = (int)((total / 4 + 31) / 32);
  hipLaunchKernelGGL(wgrad_c64_reduce_kernel, dim3(blocks), dim3(256), 0, stream, partial, grid, total, out,
                     out_fp32, beta);
  return DTP_CHECK_LAUNCH();
}
