// 3x3 convolution with 64 REDUCTION channels (VGG conv1_2 forward and data-gradient, conv2_1
// forward) for gfx950: weights live in REGISTERS, activations in a 4-row LDS ring.
//
// The generic implicit GEMM (igemm.hip) re-stages a 256-pixel activation tile for each of the nine
// taps; with only 64 output channels per tile that is 40 KB of LDS fill per 2 MFLOP and the kernel
// sits on the per-CU LDS-DMA rate (measured ~420-600 TF/s).  Here:
//   * each wave owns 32 output channels and keeps its whole filter slice, 9 taps x 64 channels =
//     36 A-fragments of v_mfma_f32_32x32x16_bf16 (144 VGPRs), in registers for the kernel's life;
//   * a workgroup walks DOWN a 32-pixel-wide image strip: per step ONE new input row segment
//     (34 px with halo, 4.3 KB) enters a 4-slot LDS ring by LDS-DMA while rows h-1, h, h+1 are
//     resident; 36 ds_read_b128 + 36 MFMAs produce a 32 px x 32 ch output block per wave;
//   * global->LDS traffic drops ~9x (each activation row is staged once, not once per tap), so the
//     loop is MFMA-paced instead of fill-paced;
//   * bias initialises the accumulator; ReLU (forward) or the ReLU-backward mask of the consumer
//     (data-gradient) is applied on the way out; stores are 64-byte row segments via a per-wave
//     LDS shuffle.
// Borders / strip tails are redirected per lane to a zeroed page.  XOR swizzle (chunk ^= (pos>>1)&7)
// on the DMA source address and on the reads keeps the 32-row fragment reads conflict-free.
#include "dtp_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct ConvC64Params {
  const char* X;         // [N*H*W][ldx] bf16: the 64 reduction channels are a 128-byte slice of each pixel row
  const char* acc_src;   // optional [N*H*W][O] bf16 added to the accumulator (may alias out): lets a
                         // 128-channel reduction run as two 64-channel passes, the second in place
  int ldx;               // pixel stride of X in elements (64, or 128 for a half of a 128-channel tensor)
  const char* Wt;        // [O][9*64] bf16
  const float* bias;     // [O] or null
  const char* mask_src;  // [N*H*W][O] bf16 or null
  char* out;             // [N*H*W][O] bf16   (POOL: [N*(H/2)*(W/2)][O], the 2x2-max-pooled output)
  unsigned char* bits_out;   // optional (non-POOL): (output > 0), byte (px*O + c)/8, bit c%8 — ReLU mask for a later dgrad
  uint32_t* codes;       // POOL: [N*(H/2)*(W/2)][O/8] 4-bit (first-argmax, max > 0) codes for the pool backward
  const char* zero_page;
  int N, H, W;
  int S, RC, rows_per_chunk;
  int relu;
  float mask_scale;
};

template <int O, bool POOL>
__global__ __launch_bounds__(O * 2, 2) void conv_c64_kernel(const ConvC64Params p) {
  constexpr int NW = O / 32;             // waves per workgroup (2 or 4)
  constexpr int X_SLOT = 40 * 128;
  constexpr int X_WI = 5;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lX = smem;                       // 4 ring slots
  char* lO = smem + 4 * X_SLOT;          // per-wave 2 KiB output shuffle

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hb = lane >> 5;

  // ---- work item
  const int chunk = blockIdx.x;
  const int unit = chunk / p.RC, rc = chunk - unit * p.RC;
  const int n = unit / p.S, s = unit - n * p.S;
  const int w0 = s * 32;
  const int h0 = rc * p.rows_per_chunk;
  int h1 = h0 + p.rows_per_chunk;
  if (h1 > p.H) h1 = p.H;
  const size_t img_px = (size_t)n * p.H * p.W;

  // ---- this wave's filter slice -> registers: A operand, lane = (row cout = l31, k = 8*hb + j)
  bf16x8_t wf[9][4];
  {
    const char* wbase = p.Wt + ((size_t)(wave * 32 + l31) * 576 + 8 * hb) * 2;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) wf[t][k] = *(const bf16x8_t*)(wbase + (t * 64 + k * 16) * 2);
  }
  // accumulator row (cout within the wave's 32) of register r: (r&3) + 8*(r>>2) + 4*hb.  The bias
  // sits in LDS and is added after the MFMAs: 16 registers less than a register-resident copy, which
  // lets the fragment reads run one tap (4 x ds_read_b128) ahead of the MFMAs.
  float* lBias = (float*)(smem + 4 * X_SLOT + NW * 2048 + O * 2 * 32);
  if (tid < O) lBias[tid] = p.bias ? p.bias[tid] : 0.f;

  // ---- ring staging geometry
  const int xr = lane >> 3, xpc = lane & 7;
  auto stage_x = [&](int hrow) {
    char* slot = lX + ((hrow + 1) & 3) * X_SLOT;
    const bool row_ok = hrow >= 0 && hrow < p.H;
#pragma unroll
    for (int k = 0; k < (X_WI + NW - 1) / NW; ++k) {
      const int wi = wave + k * NW;
      if (wi < X_WI) {
        const int pos = wi * 8 + xr;
        const int w = w0 - 1 + pos;
        const bool ok = row_ok && pos < 34 && w >= 0 && w < p.W;
        const int lc = xpc ^ ((pos >> 1) & 7);
        const char* g = ok ? p.X + ((img_px + (size_t)hrow * p.W + w) * p.ldx + lc * 8) * 2 : p.zero_page;
        glds16(g, slot + wi * 1024);
      }
    }
  };
  // fragment read geometry: B operand, lane = (col px = l31, k = 8*hb + j); ring position px+dx+1
  int xoff[3][4];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int pos = l31 + d;
#pragma unroll
    for (int k = 0; k < 4; ++k) xoff[d][k] = pos * 128 + ((((2 * k + hb) ^ ((pos >> 1) & 7))) << 4);
  }
  char* myO = lO + wave * 2048;
  const bool px_ok = (w0 + l31) < p.W;

  uint32_t top_a = 0;
  char* myT = lO + NW * 2048 + tid * 32;   // POOL: per-lane parking slot for the top row of a window
  stage_x(h0 - 1);
  stage_x(h0);
  stage_x(h0 + 1);
  for (int h = h0; h < h1; ++h) {
    sync_lds_dma();                      // row h+1 landed (explicit vmcnt(0) + barrier); step h-1 done everywhere
    if (h + 1 < h1) stage_x(h + 2);
    // earlier partial sums (two-pass 128-channel reduction): issued now, added after the MFMAs.
    // Register r holds cout 8*(r>>2) + 4*hb + (r&3) of pixel l31 -> four 8-byte loads per lane.
    u32x2_t pv[4] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}};
    if (!POOL && p.acc_src && px_ok) {
      const char* ap = p.acc_src + ((img_px + (size_t)h * p.W + w0 + l31) * O + wave * 32 + 4 * hb) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) pv[q] = *(const u32x2_t*)(ap + q * 16);
    }
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // tap t+1's four fragment reads are issued before tap t's four MFMAs; asm reads with counted
    // waits (hipcc's own placement waited lgkmcnt(0) at every second tap)
    u32x4_t xf[2][4];
    const unsigned lx0 = lds_addr(lX);
#define C64_RD(buf, t)                                                                          \
  {                                                                                             \
    const unsigned s_ = lx0 + ((h + (t) / 3) & 3) * X_SLOT;                                     \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) DTP_DSR128(xf[buf][k], s_ + xoff[(t) % 3][k]); \
  }
    __builtin_amdgcn_sched_barrier(0);
    C64_RD(0, 0)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) {
        C64_RD((t + 1) & 1, t + 1)
        DTP_LGKM_WAIT4(4, xf[t & 1][0], xf[t & 1][1], xf[t & 1][2], xf[t & 1][3]);
      } else {
        DTP_LGKM_WAIT4(0, xf[t & 1][0], xf[t & 1][1], xf[t & 1][2], xf[t & 1][3]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][k], __builtin_bit_cast(bf16x8_t, xf[t & 1][k]), acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef C64_RD
    {
      // bias: register 4q + e holds cout 8q + 4hb + e -> one 16-byte LDS read per q
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4_t b4 = *(const f32x4_t*)(lBias + wave * 32 + 8 * q + 4 * hb);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q * 4 + e] += b4[e];
      }
    }
    if (!POOL && p.acc_src) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q * 4 + 0] += __uint_as_float(pv[q][0] << 16);
        acc[q * 4 + 1] += __uint_as_float(pv[q][0] & 0xFFFF0000u);
        acc[q * 4 + 2] += __uint_as_float(pv[q][1] << 16);
        acc[q * 4 + 3] += __uint_as_float(pv[q][1] & 0xFFFF0000u);
      }
    }
    if constexpr (POOL) {
      // ---- fused ReLU + 2x2/s2 max-pool: the full-resolution output is never written.  Lane =
      // (pixel l31, hb); horizontal partner = lane ^ 1; the top row of each window waits in
      // registers for the bottom row.  Tie rule = torch's (first maximum in (0,0),(0,1),(1,0),(1,1)
      // order), evaluated on the bf16-rounded, ReLU'd values like the stand-alone pool kernel.
      float hm[16];
      uint32_t ha = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = bf16_to_f32(f32_to_bf16(fmaxf(acc[r], 0.f)));
        const float o = __shfl_xor(v, 1);
        const bool right = o > v;                      // meaningful on even-pixel lanes
        hm[r] = right ? o : v;
        ha |= (right ? 1u : 0u) << (2 * r);
      }
      if (((h - h0) & 1) == 0) {
        // park the top row in this lane's private LDS strip (bf16 pairs): keeps the kernel at two
        // waves per SIMD (the filter slice already owns 144 VGPRs)
        u32x4_t t0, t1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          t0[e] = pack_bf16x2(hm[2 * e], hm[2 * e + 1]);
          t1[e] = pack_bf16x2(hm[8 + 2 * e], hm[8 + 2 * e + 1]);
        }
        *(u32x4_t*)(myT) = t0;
        *(u32x4_t*)(myT + 16) = t1;
        top_a = ha;
      } else {
        float top_m[16];
        {
          const u32x4_t t0 = *(const u32x4_t*)(myT), t1 = *(const u32x4_t*)(myT + 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            top_m[2 * e] = __uint_as_float(t0[e] << 16);
            top_m[2 * e + 1] = __uint_as_float(t0[e] & 0xFFFF0000u);
            top_m[8 + 2 * e] = __uint_as_float(t1[e] << 16);
            top_m[8 + 2 * e + 1] = __uint_as_float(t1[e] & 0xFFFF0000u);
          }
        }
        uint32_t hc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float m4[4];
          uint32_t nib = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = q * 4 + e;
            float m = top_m[r];
            uint32_t arg = (top_a >> (2 * r)) & 1u;
            if (hm[r] > m) { m = hm[r]; arg = 2u + ((ha >> (2 * r)) & 1u); }
            m4[e] = m;
            nib |= (arg | (m > 0.f ? 4u : 0u)) << (4 * e);
          }
          hc[q] = nib;                                 // channels 8q + 4hb .. +3 of this wave's 32
          u32x2_t pk;
          pk[0] = pack_bf16x2(m4[0], m4[1]);
          pk[1] = pack_bf16x2(m4[2], m4[3]);
          const int pp = l31 >> 1;                     // pooled pixel within the strip (even lanes)
          const int slot8 = (2 * q + hb) ^ (pp & 7);
          if ((l31 & 1) == 0) *(u32x2_t*)(myO + pp * 64 + slot8 * 8) = pk;
        }
        // codes: lane hb = 0 holds nibbles 0-3 of each 8-channel group, its partner (lane + 32) 4-7
        const uint32_t c01 = hc[0] | (hc[1] << 16), c23 = hc[2] | (hc[3] << 16);
        const uint32_t o01 = __shfl_xor(c01, 32), o23 = __shfl_xor(c23, 32);
        __builtin_amdgcn_wave_barrier();
        const int OHp = p.H >> 1, OWp = p.W >> 1;
        const size_t prow = ((size_t)n * OHp + (h >> 1)) * OWp + (w0 >> 1);
        {
          const int px = lane >> 2, c16 = lane & 3;    // 16 pooled pixels x 4 chunks of 16 B
          const int x7 = px & 7;
          u32x4_t v;
          const u32x2_t lo = *(const u32x2_t*)(myO + px * 64 + (((2 * c16) ^ x7) << 3));
          const u32x2_t hi = *(const u32x2_t*)(myO + px * 64 + (((2 * c16 + 1) ^ x7) << 3));
          v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
          if ((w0 + 2 * px) < p.W) *(u32x4_t*)(p.out + ((prow + px) * O + wave * 32 + c16 * 8) * 2) = v;
        }
        if (hb == 0 && (l31 & 1) == 0 && (w0 + l31) < p.W) {
          u32x4_t cw;
          cw[0] = (c01 & 0xFFFFu) | (o01 << 16);
          cw[1] = (c01 >> 16) | (o01 & 0xFFFF0000u);
          cw[2] = (c23 & 0xFFFFu) | (o23 << 16);
          cw[3] = (c23 >> 16) | (o23 & 0xFFFF0000u);
          *(u32x4_t*)(p.codes + (prow + (l31 >> 1)) * (O / 8) + wave * 4) = cw;
        }
      }
    } else {
      // ---- epilogue: (relu) -> bf16 -> per-wave LDS shuffle -> 16-byte row-segment stores
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[q * 4 + e];
          if (p.relu) v[e] = fmaxf(v[e], 0.f);
        }
        u32x2_t pk;
        pk[0] = pack_bf16x2(v[0], v[1]);
        pk[1] = pack_bf16x2(v[2], v[3]);
        const int slot8 = (2 * q + hb) ^ (l31 & 7);            // 8-byte slot = couts 8q + 4hb .. +3
        *(u32x2_t*)(myO + l31 * 64 + slot8 * 8) = pk;
      }
      __builtin_amdgcn_wave_barrier();
      const size_t row_px = img_px + (size_t)h * p.W + w0;
  #pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int id = lane + 64 * r;                           // 128 chunks of 16 B: (px, c16)
        const int px = id >> 2, c16 = id & 3;
        const int x7 = px & 7;
        u32x4_t v;
        const u32x2_t lo = *(const u32x2_t*)(myO + px * 64 + (((2 * c16) ^ x7) << 3));
        const u32x2_t hi = *(const u32x2_t*)(myO + px * 64 + (((2 * c16 + 1) ^ x7) << 3));
        v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
        if ((w0 + px) < p.W) {
          const size_t goff = ((row_px + px) * O + wave * 32 + c16 * 8) * 2;
          if (p.mask_src) {
            const u32x4_t mk = *(const u32x4_t*)(p.mask_src + goff);
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t mw = mk[e], vw = v[e];
              const bool lo_pos = ((mw & 0x8000u) == 0) && ((mw & 0x7FFFu) != 0);
              const bool hi_pos = ((mw & 0x80000000u) == 0) && ((mw & 0x7FFF0000u) != 0);
              const float a = lo_pos ? bf16_to_f32((bf16_t)(vw & 0xFFFFu)) * p.mask_scale : 0.f;
              const float b = hi_pos ? bf16_to_f32((bf16_t)(vw >> 16)) * p.mask_scale : 0.f;
              v[e] = pack_bf16x2(a, b);
            }
          }
          *(u32x4_t*)(p.out + goff) = v;
          if (p.bits_out) {
            uint32_t b = 0;
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t vw = v[e];
              b |= (uint32_t)(((vw & 0x8000u) == 0) && ((vw & 0x7FFFu) != 0)) << (2 * e);
              b |= (uint32_t)(((vw & 0x80000000u) == 0) && ((vw & 0x7FFF0000u) != 0)) << (2 * e + 1);
            }
            p.bits_out[goff >> 4] = (unsigned char)b;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  (void)px_ok;
}

// X [N,H,W,64] bf16, Wt [O][576] bf16 -> out [N,H,W,O] bf16 (O = 64 or 128)
// codes != null: fused ReLU + 2x2 max-pool epilogue; out is then [N,H/2,W/2,O] and codes [N,H/2,W/2,O/8] uint32
extern "C" int dtp_conv_c64(const void* X, const void* Wt, const float* bias, const void* mask_src, void* out,
                            void* codes, void* bits_out, const void* zero_page, int N, int H, int W, int O, int relu,
                            float mask_scale, int ldx, const void* acc_src, hipStream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || (O != 64 && O != 128)) return -10;
  if (ldx < 64 || (ldx & 7)) return -14;
  if (acc_src && codes) return -15;
  ConvC64Params p;
  p.X = (const char*)X; p.Wt = (const char*)Wt; p.bias = bias; p.mask_src = (const char*)mask_src;
  p.out = (char*)out; p.zero_page = (const char*)zero_page;
  p.ldx = ldx; p.acc_src = (const char*)acc_src;
  p.N = N; p.H = H; p.W = W; p.relu = relu; p.mask_scale = mask_scale;
  p.S = (W + 31) / 32;
  const long units = (long)N * p.S;
  // row chunks: enough workgroups to fill the chip several times over, >= 16 rows each so the
  // 18 KB register filter load and the 3-row ring prologue amortise
  int rc = (int)((16L * 256 + units - 1) / units);
  if (rc > H / 16) rc = H / 16;
  if (rc < 1) rc = 1;
  p.rows_per_chunk = (H + rc - 1) / rc;
  p.RC = (H + p.rows_per_chunk - 1) / p.rows_per_chunk;
  const long grid = units * p.RC;
  if (grid > 0x7FFFFFFFL) return -11;
  p.codes = (uint32_t*)codes;
  p.bits_out = (unsigned char*)bits_out;
  if (codes && bits_out) return -13;     // the pool codes already carry the sign bit
  if (codes) {   // fused ReLU + max-pool: windows must not straddle row chunks or the image edge
    if ((H & 1) || (W & 1) || !relu || mask_src) return -12;
    if (p.rows_per_chunk & 1) {
      p.rows_per_chunk += 1;
      p.RC = (H + p.rows_per_chunk - 1) / p.rows_per_chunk;
    }
  }
  const long grid2 = units * p.RC;
  if (O == 64) {
    constexpr int lds = 4 * 40 * 128 + 2 * 2048 + 128 * 32 + 64 * 4;
    if (codes) hipLaunchKernelGGL((conv_c64_kernel<64, true>), dim3((unsigned)grid2), dim3(128), lds, stream, p);
    else hipLaunchKernelGGL((conv_c64_kernel<64, false>), dim3((unsigned)grid2), dim3(128), lds, stream, p);
  } else {
    constexpr int lds = 4 * 40 * 128 + 4 * 2048 + 256 * 32 + 128 * 4;
    if (codes) hipLaunchKernelGGL((conv_c64_kernel<128, true>), dim3((unsigned)grid2), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((conv_c64_kernel<128, false>), dim3((unsigned)grid2), dim3(256), lds, stream, p);
  }
  return DTP_CHECK_LAUNCH();
}
