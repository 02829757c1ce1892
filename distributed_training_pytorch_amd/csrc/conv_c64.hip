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
#include <cstdlib>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// MFMA shape of the row step.  1 (default): v_mfma_f32_16x16x32_bf16 — a 32 px x 32 cout block is four
// 16x16 accumulators, one B-fragment read (16 px x 32 channels) feeds two MFMAs (cout halves), so LDS
// bytes, matrix cycles and the 144 filter registers are those of the 32x32x16 form (0, kept for A/B:
// -DDTP_C64_MFMA16=0), but the chip holds a higher clock under this shape (MI355X_MICROARCH.md "DVFS
// give-back" item 7; probed here at +3.5-5 % before the conversion).  Only the lane -> (pixel, channel)
// maps of the operands and of the accumulator differ; everything behind the parked row buffer is shared.
#ifndef DTP_C64_MFMA16
#define DTP_C64_MFMA16 1
#endif

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
  float* stats_part;     // STATS kernels: [gridDim.x][2][O] f32 (sum y, sum y^2) of this workgroup's stored outputs —
                         // the batch statistics of a BatchNorm that follows (bn.hip partial layout)
  unsigned long long* stamps;   // DEBUG (DTP_C64_STAMPS): per workgroup, first 16 row steps x 6 stamps (bench/stamp_c64.py)
};

// LDS ops the compiler does not track (counted waits, see dtp_common.h DTP_DSR128)
#define C64_DSW64(addr, data) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(data) : "memory")
#define C64_NAME2(a, b) asm volatile("" : "+v"(a), "+v"(b))
// XOR swizzle (16-byte chunks) of ring position pos: conflict-free for the lane groups of ds_read_b128
// under the fragment geometry in use (bench/lds_bank_sim.py model; the 32-row reads of the 32x32x16
// form want (pos >> 1) & 7, the 16-row reads of the 16x16x32 form pos & 6 — with the former they
// would conflict 2-way at dx = 0, +1)
#if DTP_C64_MFMA16
#define C64_SWZ(pos) ((pos) & 6)
#else
#define C64_SWZ(pos) (((pos) >> 1) & 7)
#endif
#define C64_NAME4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

// Row step = 36 MFMAs on ONE accumulator chain.  In-kernel stamps (bench/stamp_c64.py) showed a wave
// spending 4,100 cycles per step for 1,152 cycles of matrix work — the same at one and at two waves
// per SIMD: not pipe contention but the step's serial chain (DMA wait + barrier 500, LDS-DMA issue
// ~400, exposed first-tap read latency, 4 reads between every 4 MFMAs, then bias / ReLU / pack / LDS
// shuffle 640 and store issue 490 with the matrix pipe idle).  This version keeps the chain short:
//   * fragment reads are issued ONE PER MFMA, four MFMAs ahead, with a counted lgkmcnt in front of
//     each MFMA — no read burst between taps;
//   * the accumulator is initialised with the bias by four LDS reads straight into its registers;
//   * the finished row is only rounded, ReLU'd and parked in a per-wave LDS row buffer (24 VALU +
//     4 ds_write_b64); its shuffle read-back, masking, stores and 1-bit masks — or the 2x2 max-pool
//     of a finished row pair with its argmax codes — run in the gaps of the NEXT row's MFMAs;
//   * the workgroup barrier sits after tap 2 (the last read of ring row h-1): the LDS-DMA of row
//     h+3 that follows it has two row steps to land, and the deferred stores issued right after it
//     are a full step old when the next `vmcnt(0)` comes.
template <int O, bool POOL, bool MASK, bool STATS = false>
__global__ __launch_bounds__(O * 2, 2) void conv_c64_kernel(const ConvC64Params p) {
  static_assert(!STATS || (!POOL && !MASK), "statistics come out of the plain store epilogue");
  constexpr int NW = O / 32;             // waves per workgroup (2 or 4)
  constexpr int PW = NW == 2 ? 3 : 2;    // LDS-DMA pieces (8 ring positions, 1 KiB) per wave and row
  constexpr int X_SLOT = PW * NW * 1024; // 48 / 64 ring positions: 34 used, the rest take hardware zeros
  constexpr int ND = POOL ? 4 : 2;       // deferred LDS reads per step (always issued: the counted waits assume them)
  constexpr unsigned BAD = 0x40000000u;  // half of an out-of-range offset: row part + column part, either or both BAD, stay out of range
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lX = smem;                       // 4 ring slots: row r lives in slot (r + 1) & 3
  char* lO = smem + 4 * X_SLOT;          // per wave: two 2 KiB row buffers [32 px][32 cout] bf16
  float* lBias = (float*)(lO + NW * 4096);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hb = lane >> 5;
  const int l15 = lane & 15, g4 = lane >> 4;      // 16x16x32 operand / accumulator geometry

  // ---- work item
  const int chunk = blockIdx.x;
  const int unit = chunk / p.RC, rc = chunk - unit * p.RC;
  const int n = unit / p.S, s = unit - n * p.S;
  const int w0 = s * 32;
  const int h0 = rc * p.rows_per_chunk;
  int h1 = h0 + p.rows_per_chunk;
  if (h1 > p.H) h1 = p.H;
  const size_t img_px = (size_t)n * p.H * p.W;
  const int OHp = p.H >> 1, OWp = p.W >> 1;

  // ---- this wave's filter slice -> registers: A operand, lane = (row cout = l31, k = 8*hb + j)
  bf16x8_t wf[9][4];
  {
#if DTP_C64_MFMA16
    // wf[t][2*ch + oh]: tap t, channel half ch (32 channels = one K step), cout half oh;
    // lane = (row cout = 16*oh + l15, k = 8*g4 + j)
    const char* wbase = p.Wt + ((size_t)(wave * 32 + l15) * 576 + 8 * g4) * 2;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        wf[t][k] = *(const bf16x8_t*)(wbase + ((size_t)(k & 1) * 16 * 576 + t * 64 + (k >> 1) * 32) * 2);
#else
    const char* wbase = p.Wt + ((size_t)(wave * 32 + l31) * 576 + 8 * hb) * 2;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) wf[t][k] = *(const bf16x8_t*)(wbase + (t * 64 + k * 16) * 2);
#endif
  }
  if (tid < O) lBias[tid] = p.bias ? p.bias[tid] : 0.f;

  // ---- per-image buffer descriptors: every tail (image border, strip tail, rows past the chunk,
  // "nothing parked yet", absent 1-bit-mask / stamp buffers) is an out-of-range offset — loads give
  // zeros (LDS-DMA: zeros written to LDS), stores are dropped — so a row step is ONE basic block and
  // hipcc can spread the deferred epilogue between the MFMAs.
  const long x_img_bytes = ((long)p.H * p.W - 1) * p.ldx * 2 + 128;
  const __amdgpu_buffer_rsrc_t x_rs = make_rsrc(p.X + img_px * p.ldx * 2, x_img_bytes);
  const long o_img_bytes = POOL ? (long)OHp * OWp * O * 2 : (long)p.H * p.W * O * 2;
  const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(p.out + (size_t)n * o_img_bytes, o_img_bytes);
  const __amdgpu_buffer_rsrc_t m_rs = make_rsrc(MASK ? p.mask_src + (size_t)n * o_img_bytes : nullptr, MASK ? o_img_bytes : 0);
  // second output: !POOL 1-bit masks (1 byte per 16-byte chunk), POOL argmax codes (4 bytes per 16-byte chunk)
  const long b_img_bytes = POOL ? o_img_bytes / 4 : o_img_bytes / 16;
  const char* b_base = POOL ? (const char*)p.codes : (const char*)p.bits_out;
  const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(b_base ? b_base + (size_t)n * b_img_bytes : nullptr, b_base ? b_img_bytes : 0);

  // ---- ring staging: piece k of this wave covers ring positions (wave + k*NW)*8 .. +7
  unsigned xcol[PW];
#pragma unroll
  for (int k = 0; k < PW; ++k) {
    const int pos = (wave + k * NW) * 8 + (lane >> 3);
    const int w = w0 - 1 + pos;
    const int lc = (lane & 7) ^ C64_SWZ(pos);
    xcol[k] = (pos < 34 && w >= 0 && w < p.W) ? (unsigned)((w * p.ldx + lc * 8) * 2) : BAD;
  }
  const unsigned x_row_bytes = (unsigned)(p.W * p.ldx * 2);
  auto stage_x = [&](int hrow) {
    char* slot = lX + ((hrow + 1) & 3) * X_SLOT + wave * 1024;
    const unsigned roff = (hrow >= 0 && hrow < p.H && hrow <= h1) ? (unsigned)hrow * x_row_bytes : BAD;
#pragma unroll
    for (int k = 0; k < PW; ++k) glds16_buf(x_rs, xcol[k] + roff, slot + k * NW * 1024);
  };
  // fragment read geometry: B operand, lane = (col px = l31, k = 8*hb + j); ring position px+dx+1
  // byte offset of fragment k at ring position pos = l31 + d: pos*128 + (((2k + hb) ^ ((pos>>1)&7)) << 4)
  //   = xo[d] ^ (k << 5)   (2k only touches chunk bits 1..2): three registers instead of twelve
#if DTP_C64_MFMA16
  // 16x16x32: B operand, lane = (col px = 16*ph + l15, k = 8*g4 + j of channel half ch); fragment
  // (ph, ch) at ring position pos = 16*ph + l15 + d: pos*128 + (((4*ch + g4) ^ swz(pos)) << 4)
  //   = xo[ph][d] ^ (ch << 6)
  int xo[2][3];
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int pos = 16 * ph + l15 + d;
      xo[ph][d] = pos * 128 + ((g4 ^ C64_SWZ(pos)) << 4);
    }
#else
  int xo[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int pos = l31 + d;
    xo[d] = pos * 128 + ((hb ^ C64_SWZ(pos)) << 4);
  }
#endif
  const unsigned lx0 = lds_addr(lX);
  const unsigned myO = lds_addr(lO + wave * 4096);
  // accumulator register 4q + e = cout 8q + 4hb + e of pixel l31: bias as four 16-byte LDS reads
  // row buffer: pixel px = 64 bytes = four 16-byte chunks (8 couts), chunk c stored at c ^ ((px >> 2) & 3)
#if DTP_C64_MFMA16
  // accumulator (oh, ph)[e] = cout 16*oh + 4*g4 + e of pixel 16*ph + l15: bias as two 16-byte LDS reads
  const unsigned bias_a = lds_addr(lBias) + (wave * 32 + 4 * g4) * 4;
  const unsigned wr_a = myO + l15 * 64 + (g4 & 1) * 8;          // + ph * 1024; chunk 2*oh + (g4 >> 1)
  const int wr_sw = (l15 >> 2) & 3;
#else
  const unsigned bias_a = lds_addr(lBias) + (wave * 32 + 4 * hb) * 4;
  const unsigned wr_a = myO + l31 * 64 + hb * 8;
  const int wr_sw = (l31 >> 2) & 3;
#endif
  // deferred read-back geometry
  //   !POOL: lane = (px = lane/4 + 16r, chunk = lane%4): one 16-byte row segment per read / store
  //   POOL:  lane = (pooled px pp = lane/4, chunk = lane%4): pixels 2pp, 2pp+1 of both row buffers
  const int dch = lane & 3;
  unsigned rd_a[2], ocol[2];             // ocol: column part of the output byte offset inside the image
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int px = POOL ? 2 * (lane >> 2) + r : (lane >> 2) + 16 * r;
    rd_a[r] = myO + px * 64 + ((dch ^ ((px >> 2) & 3)) << 4);
    if (POOL) {
      const int pp = lane >> 2;
      ocol[r] = (w0 + 2 * pp) < p.W ? (unsigned)((((w0 >> 1) + pp) * O + wave * 32 + dch * 8) * 2) : BAD;
    } else {
      ocol[r] = (w0 + px) < p.W ? (unsigned)(((w0 + px) * O + wave * 32 + dch * 8) * 2) : BAD;
    }
  }
  const unsigned o_row_bytes = (unsigned)((POOL ? OWp : p.W) * O * 2);

  u32x4_t st[4];                         // read-back of the parked row(s)
  u32x4_t mk[2];                         // MASK: bf16 ReLU-mask source of the parked row
  auto deferred_reads = [&](unsigned row_off) {    // row_off: byte offset of the parked row in the image, or BAD
    DTP_DSR128(st[0], rd_a[0]);
    DTP_DSR128(st[1], rd_a[1]);
    if constexpr (POOL) {
      DTP_DSR128(st[2], rd_a[0] + 2048);
      DTP_DSR128(st[3], rd_a[1] + 2048);
    }
    if constexpr (MASK) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        mk[r] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(m_rs, (int)(ocol[r] + row_off), 0, 0));
    }
  };
  // !POOL: the parked row -> global (mask, 1-bit masks); st[] complete (see the callers)
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // STATS
  auto deferred_store = [&](unsigned row_off) {
    C64_NAME2(st[0], st[1]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      u32x4_t v = st[r];
      const unsigned off = ocol[r] + row_off;
      if constexpr (STATS) {
        // only what is really stored counts: strip tail (ocol BAD) and "nothing parked" (row BAD) are dropped
        const uint32_t keep = (ocol[r] < BAD && row_off < BAD) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float((v[e] << 16) & keep), hi = __uint_as_float(v[e] & 0xFFFF0000u & keep);
          cs[2 * e] += lo;
          cs[2 * e + 1] += hi;
          cq[2 * e] = fmaf(lo, lo, cq[2 * e]);
          cq[2 * e + 1] = fmaf(hi, hi, cq[2 * e + 1]);
        }
      }
      if constexpr (MASK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t mw = mk[r][e], vw = v[e];
          const bool lo_pos = ((mw & 0x8000u) == 0) && ((mw & 0x7FFFu) != 0);
          const bool hi_pos = ((mw & 0x80000000u) == 0) && ((mw & 0x7FFF0000u) != 0);
          const float a = lo_pos ? bf16_to_f32((bf16_t)(vw & 0xFFFFu)) * p.mask_scale : 0.f;
          const float b = hi_pos ? bf16_to_f32((bf16_t)(vw >> 16)) * p.mask_scale : 0.f;
          v[e] = pack_bf16x2(a, b);
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(v, o_rs, (int)off, 0, 0);
      const uint32_t b = bf16x2_pos_bits(v[0]) | (bf16x2_pos_bits(v[1]) << 2) | (bf16x2_pos_bits(v[2]) << 4) | (bf16x2_pos_bits(v[3]) << 6);
      __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b, b_rs, (int)(off >> 4), 0, 0);
    }
  };
  // POOL: 2x2/s2 max over the two parked rows (top = buffer 0, bottom = buffer 1), 8 channels of one
  // pooled pixel per lane.  The values are post-ReLU bf16 (non-negative), so their bit patterns
  // order like the numbers; tie rule = torch's (first maximum in (0,0),(0,1),(1,0),(1,1) order);
  // code nibble = argmax | (max > 0) << 2.
  u32x4_t pool_v;
  uint32_t pool_code;
  auto pool_word = [&](int j) {
    uint32_t outw = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const uint32_t A = (st[0][j] >> (16 * half)) & 0xFFFFu, B = (st[1][j] >> (16 * half)) & 0xFFFFu;
      const uint32_t C = (st[2][j] >> (16 * half)) & 0xFFFFu, D = (st[3][j] >> (16 * half)) & 0xFFFFu;
      const uint32_t ta = B > A ? 1u : 0u, t = B > A ? B : A;
      const uint32_t ua = D > C ? 1u : 0u, u = D > C ? D : C;
      const bool take_u = u > t;
      const uint32_t m = take_u ? u : t;
      const uint32_t nib = (take_u ? 2u + ua : ta) | (m != 0u ? 4u : 0u);
      outw |= m << (16 * half);
      pool_code |= nib << (8 * j + 4 * half);
    }
    pool_v[j] = outw;
  };
  auto pool_store = [&](unsigned row_off) {
    const unsigned off = ocol[0] + row_off;
    __builtin_amdgcn_raw_buffer_store_b128(pool_v, o_rs, (int)off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(pool_code, b_rs, (int)(off >> 2), 0, 0);
  };

#ifdef DTP_C64_STAMPS   // diagnostic build only (DTP_BUILD_DEFS=-DDTP_C64_STAMPS): the production kernel holds no stamp code
#define C64_STAMP(i) if (p.stamps && tid == 0 && h - h0 < 16) p.stamps[((size_t)blockIdx.x * 16 + (h - h0)) * 6 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define C64_STAMP(i)
#endif
  stage_x(h0 - 1);
  stage_x(h0);
  stage_x(h0 + 1);
  stage_x(h0 + 2);
  sync_lds_dma();                        // ring rows h0-1 .. h0+2 and the bias are in LDS
  for (int h = h0; h < h1; ++h) {
    C64_STAMP(0)
    // parked work from earlier rows: its output row offset in the image, BAD when there is none
    const bool pend = POOL ? (h > h0 && ((h - h0) & 1) == 0) : h > h0;
    const unsigned prev_off = pend ? (unsigned)(POOL ? (h - 1) >> 1 : h - 1) * o_row_bytes : BAD;
    unsigned sl[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) sl[d] = lx0 + ((h + d) & 3) * X_SLOT;    // rows h-1, h, h+1
    u32x4_t bq[4], xf[2][4];
    __builtin_amdgcn_sched_barrier(0);
#if DTP_C64_MFMA16
    // read i of a tap (i % 4): pixel half ph = i & 1, channel half ch = (i >> 1) & 1
    DTP_DSR128(bq[0], bias_a);
    DTP_DSR128(bq[1], bias_a + 64);
    DTP_DSR128(bq[2], bias_a);            // (bq[2], bq[3]: the counted waits below assume four bias reads)
    DTP_DSR128(bq[3], bias_a + 64);
#define C64_RD(i) DTP_DSR128(xf[((i) / 4) & 1][(i) % 4], (sl[(i) / 12] + xo[(i) & 1][((i) / 4) % 3]) ^ ((((i) >> 1) & 1) << 6));
    C64_RD(0) C64_RD(1) C64_RD(2) C64_RD(3)
    f32x4_t acc4[2][2];                   // [cout half oh][pixel half ph]
#else
#pragma unroll
    for (int q = 0; q < 4; ++q) DTP_DSR128(bq[q], bias_a + q * 32);
#define C64_RD(i) DTP_DSR128(xf[((i) / 4) & 1][(i) % 4], (sl[(i) / 12] + xo[((i) / 4) % 3]) ^ (((i) % 4) << 5));
    C64_RD(0) C64_RD(1) C64_RD(2) C64_RD(3)
    f32x16_t acc;
#endif
    // (a macro per step, not a `#pragma unroll` loop: hipcc left the loop rolled and sent the
    // filter registers to scratch)
    // LDS ops younger than read i when MFMA i issues: reads i+1 .. i+3, and for i = 2 .. 5 the ND
    // deferred reads issued behind read 5
#define C64_YOUNG(i) ((35 - (i) < 3 ? 35 - (i) : 3) + (((i) >= 2 && (i) <= 5) ? ND : 0))
#ifdef DTP_C64_FENCED
#define C64_SB __builtin_amdgcn_sched_barrier(0);
#else
#define C64_SB
#endif
#if DTP_C64_MFMA16
#define C64_ACC_INIT(q, e) acc4[(q) & 1][(q) >> 1][e]      /* bq[q] = bias quad of cout half q & 1 */
#define C64_MMA(i)                                                                                             \
    acc4[0][(i) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                                \
        wf[(i) / 4][(((i) >> 1) & 1) * 2], __builtin_bit_cast(bf16x8_t, xf[((i) / 4) & 1][(i) % 4]), acc4[0][(i) & 1], 0, 0, 0); \
    acc4[1][(i) & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                                \
        wf[(i) / 4][(((i) >> 1) & 1) * 2 + 1], __builtin_bit_cast(bf16x8_t, xf[((i) / 4) & 1][(i) % 4]), acc4[1][(i) & 1], 0, 0, 0);
#else
#define C64_ACC_INIT(q, e) acc[(q) * 4 + (e)]
#define C64_MMA(i)                                                                                             \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[(i) / 4][(i) % 4],                                        \
                                                  __builtin_bit_cast(bf16x8_t, xf[((i) / 4) & 1][(i) % 4]), acc, 0, 0, 0);
#endif
#define C64_STEP(i)                                                                                          \
  {                                                                                                          \
    if ((i) == 0) {                                                                                          \
      asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(xf[0][0]), "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3])); \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int e = 0; e < 4; ++e)            \
          C64_ACC_INIT(q, e) = __uint_as_float(bq[q][e]);                                                    \
    } else {                                                                                                 \
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(xf[((i) / 4) & 1][(i) % 4]) : "n"(C64_YOUNG(i)));          \
    }                                                                                                        \
    C64_SB                                                                                                   \
    C64_MMA(i)                                                                                               \
    C64_SB                                                                                                   \
    if ((i) + 4 < 36) { C64_RD(((i) + 4 < 36 ? (i) + 4 : 0)) }                                               \
    if ((i) == 1) {                                                                                          \
      deferred_reads(prev_off);                                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    if ((i) == 11) {                                                                                         \
      /* every wave is past its last read of ring row h-1 (reads 0..11 were waited for): its slot takes */   \
      /* row h+3.  vmcnt(0): this wave's earlier DMA (row h+2, a step old) has landed. */                    \
      C64_STAMP(1)                                                                                           \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
      __builtin_amdgcn_s_barrier();                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      C64_STAMP(2)                                                                                           \
      stage_x(h + 3);                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    if ((i) == 12) {                                                                                         \
      /* the parked row's read-back -> stores (POOL: the pool of the parked row pair): straight-line */      \
      /* code behind the LDS-DMA issue; the group barriers below spread it over the MFMA gaps */             \
      if constexpr (!POOL) {                                                                                 \
        deferred_store(prev_off);                                                                            \
      } else {                                                                                               \
        C64_NAME4(st[0], st[1], st[2], st[3]);                                                               \
        pool_code = 0;                                                                                       \
        pool_word(0); pool_word(1); pool_word(2); pool_word(3);                                              \
        pool_store(prev_off);                                                                                \
      }                                                                                                      \
    }                                                                                                        \
  }
    C64_STEP(0) C64_STEP(1) C64_STEP(2) C64_STEP(3) C64_STEP(4) C64_STEP(5) C64_STEP(6) C64_STEP(7) C64_STEP(8)
    C64_STEP(9) C64_STEP(10) C64_STEP(11) C64_STEP(12) C64_STEP(13) C64_STEP(14) C64_STEP(15) C64_STEP(16)
    C64_STEP(17) C64_STEP(18) C64_STEP(19) C64_STEP(20) C64_STEP(21) C64_STEP(22) C64_STEP(23) C64_STEP(24)
    C64_STEP(25) C64_STEP(26) C64_STEP(27) C64_STEP(28) C64_STEP(29) C64_STEP(30) C64_STEP(31) C64_STEP(32)
    C64_STEP(33) C64_STEP(34) C64_STEP(35)
    // scheduling region = barrier .. end of the chain: one MFMA, then a few VALU / VMEM of the
    // deferred epilogue and of the read addresses, 24 times
#pragma unroll
    for (int g = 0; g < 24; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, DTP_C64_MFMA16 ? 2 : 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, POOL ? 8 : (MASK ? 8 : 5), 0);
      __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
    }
#undef C64_STEP
#undef C64_YOUNG
#undef C64_RD
#undef C64_MMA
#undef C64_ACC_INIT
    C64_STAMP(3)
    // ---- the finished row: (+ earlier partial sums) -> bf16 -> ReLU -> parked in the row buffer
    if constexpr (!POOL) {
      // The run-time acc_src branch below ends the basic block right behind the last MFMA, and hipcc
      // (ROCm 7.2) left the first accumulator read behind that branch 2 wait states after an MFMA
      // that needs 11 (8 passes + 3): on the GPU the last tap's product was missing from the first
      // packed pair of every row whose row h+1 is not zero.  The POOL kernels have no branch here
      // and get the compiler's own s_nops.  Cover the hazard by hand where the branch is.
#if DTP_C64_MFMA16
      asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc4[0][0]), "+v"(acc4[0][1]), "+v"(acc4[1][0]), "+v"(acc4[1][1]));
#else
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc));
#endif
    }
#if DTP_C64_MFMA16
    if (!POOL && p.acc_src) {
      // two-pass 128-channel reduction (opt-in): accumulator (oh, ph)[e] = cout 16*oh + 4*g4 + e of pixel 16*ph + l15
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        if ((w0 + 16 * ph + l15) < p.W) {
          const char* ap = p.acc_src + ((img_px + (size_t)h * p.W + w0 + 16 * ph + l15) * O + wave * 32 + 4 * g4) * 2;
#pragma unroll
          for (int oh = 0; oh < 2; ++oh) {
            const u32x2_t pv = *(const u32x2_t*)(ap + oh * 32);
            acc4[oh][ph][0] += __uint_as_float(pv[0] << 16);
            acc4[oh][ph][1] += __uint_as_float(pv[0] & 0xFFFF0000u);
            acc4[oh][ph][2] += __uint_as_float(pv[1] << 16);
            acc4[oh][ph][3] += __uint_as_float(pv[1] & 0xFFFF0000u);
          }
        }
      }
    }
    {
      typedef short s16x2_t __attribute__((ext_vector_type(2)));
      const bool relu = POOL || p.relu;
      const unsigned buf = POOL ? (unsigned)((h - h0) & 1) * 2048u : 0u;
#pragma unroll
      for (int oh = 0; oh < 2; ++oh)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const f32x4_t a = acc4[oh][ph];
          uint32_t lo = pack_bf16x2(a[0], a[1]), hi = pack_bf16x2(a[2], a[3]);
          if (relu) {   // round(relu(x)) == relu(round(x)): packed signed-16-bit max on the bf16 bit patterns
            lo = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, lo), (s16x2_t){0, 0}));
            hi = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, hi), (s16x2_t){0, 0}));
          }
          const u32x2_t pk = {lo, hi};
          C64_DSW64(wr_a + buf + ph * 1024 + (((2 * oh + (g4 >> 1)) ^ wr_sw) << 4), pk);
        }
    }
#else
    if (!POOL && p.acc_src && (w0 + l31) < p.W) {
      // two-pass 128-channel reduction (opt-in): register r = cout 8*(r>>2) + 4*hb + (r&3) of pixel l31
      const char* ap = p.acc_src + ((img_px + (size_t)h * p.W + w0 + l31) * O + wave * 32 + 4 * hb) * 2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x2_t pv = *(const u32x2_t*)(ap + q * 16);
        acc[q * 4 + 0] += __uint_as_float(pv[0] << 16);
        acc[q * 4 + 1] += __uint_as_float(pv[0] & 0xFFFF0000u);
        acc[q * 4 + 2] += __uint_as_float(pv[1] << 16);
        acc[q * 4 + 3] += __uint_as_float(pv[1] & 0xFFFF0000u);
      }
    }
    {
      typedef short s16x2_t __attribute__((ext_vector_type(2)));
      const bool relu = POOL || p.relu;
      const unsigned buf = POOL ? (unsigned)((h - h0) & 1) * 2048u : 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t lo = pack_bf16x2(acc[q * 4 + 0], acc[q * 4 + 1]), hi = pack_bf16x2(acc[q * 4 + 2], acc[q * 4 + 3]);
        if (relu) {   // round(relu(x)) == relu(round(x)): packed signed-16-bit max on the bf16 bit patterns
          lo = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, lo), (s16x2_t){0, 0}));
          hi = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, hi), (s16x2_t){0, 0}));
        }
        const u32x2_t pk = {lo, hi};
        C64_DSW64(wr_a + buf + ((q ^ wr_sw) << 4), pk);
      }
    }
#endif
    C64_STAMP(4)
  }
  // ---- flush the last parked row / row pair
  {
    const unsigned last_off = (unsigned)(POOL ? (h1 - 1) >> 1 : h1 - 1) * o_row_bytes;
    deferred_reads(last_off);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!POOL) {
      deferred_store(last_off);
    } else {
      C64_NAME4(st[0], st[1], st[2], st[3]);
      pool_code = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) pool_word(j);
      pool_store(last_off);
    }
  }
  if constexpr (STATS) {
    // lanes with the same chunk (lane & 3) hold the same 8 couts of different pixels: fold lane bits 2..5
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int m = 4; m < 64; m <<= 1) {
        cs[e] += __shfl_xor(cs[e], m);
        cq[e] += __shfl_xor(cq[e], m);
      }
    }
    if (lane < 4) {
      float* row = p.stats_part + (size_t)blockIdx.x * 2 * O + wave * 32 + lane * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        row[e] = cs[e];
        row[O + e] = cq[e];
      }
    }
  }
}

// X [N,H,W,64] bf16, Wt [O][576] bf16 -> out [N,H,W,O] bf16 (O = 64 or 128)
// codes != null: fused ReLU + 2x2 max-pool epilogue; out is then [N,H/2,W/2,O] and codes [N,H/2,W/2,O/8] uint32
extern "C" int dtp_conv_c64(const void* X, const void* Wt, const float* bias, const void* mask_src, void* out,
                            void* codes, void* bits_out, const void* zero_page, int N, int H, int W, int O, int relu,
                            float mask_scale, int ldx, const void* acc_src, float* stats_part, int stats_rows,
                            int* nparts_out, hipStream_t stream) {
  if (N <= 0 || H <= 0 || W <= 0 || (O != 64 && O != 128)) return -10;
  if (ldx < 64 || (ldx & 7)) return -14;
  if (acc_src && codes) return -15;
  ConvC64Params p;
  p.X = (const char*)X; p.Wt = (const char*)Wt; p.bias = bias; p.mask_src = (const char*)mask_src;
  p.out = (char*)out; p.zero_page = (const char*)zero_page;
  p.ldx = ldx; p.acc_src = (const char*)acc_src;
  p.N = N; p.H = H; p.W = W; p.relu = relu; p.mask_scale = mask_scale;
  p.stamps = nullptr;
  if (const char* e = std::getenv("DTP_C64_STAMPS")) p.stamps = (unsigned long long*)std::strtoull(e, nullptr, 0);
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
  p.stats_part = stats_part;
  if (stats_part) {      // one partial row per workgroup
    if (codes || mask_src || acc_src || grid2 > stats_rows || !nparts_out) return -16;
    *nparts_out = (int)grid2;
  }
  static const int lds_pad = std::getenv("DTP_C64_LDS_PAD") ? std::atoi(std::getenv("DTP_C64_LDS_PAD")) : 0;   // DEBUG: occupancy experiments
  const bool mask = mask_src != nullptr;
  if (O == 64) {
    const int lds = 4 * 6144 + 2 * 4096 + 64 * 4 + lds_pad;
    if (codes) hipLaunchKernelGGL((conv_c64_kernel<64, true, false>), dim3((unsigned)grid2), dim3(128), lds, stream, p);
    else if (mask) hipLaunchKernelGGL((conv_c64_kernel<64, false, true>), dim3((unsigned)grid2), dim3(128), lds, stream, p);
    else if (stats_part) hipLaunchKernelGGL((conv_c64_kernel<64, false, false, true>), dim3((unsigned)grid2), dim3(128), lds, stream, p);
    else hipLaunchKernelGGL((conv_c64_kernel<64, false, false>), dim3((unsigned)grid2), dim3(128), lds, stream, p);
  } else {
    const int lds = 4 * 8192 + 4 * 4096 + 128 * 4 + lds_pad;
    if (codes) hipLaunchKernelGGL((conv_c64_kernel<128, true, false>), dim3((unsigned)grid2), dim3(256), lds, stream, p);
    else if (mask) hipLaunchKernelGGL((conv_c64_kernel<128, false, true>), dim3((unsigned)grid2), dim3(256), lds, stream, p);
    else if (stats_part) hipLaunchKernelGGL((conv_c64_kernel<128, false, false, true>), dim3((unsigned)grid2), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((conv_c64_kernel<128, false, false>), dim3((unsigned)grid2), dim3(256), lds, stream, p);
  }
  return DTP_CHECK_LAUNCH();
}
