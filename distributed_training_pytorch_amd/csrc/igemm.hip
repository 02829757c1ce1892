// Implicit-GEMM 3x3 convolution (forward and data-gradient) and the "NT" linear GEMM, bf16 in,
// f32 accumulate on MFMA (v_mfma_f32_16x16x32_bf16), for gfx950.
//
//   out[m][n] = epilogue( sum_{tap} sum_{c} A[pix(m) + shift(tap)][c] * B[n][tap*C + c] )
//
// * A: NHWC activations viewed as a [M = N*H*W][C] row-major matrix (TAPS9) or a plain [M][lda]
//   matrix (linear, one "tap").  B: [Ncols][ldb] with the reduction axis contiguous.
// * The K loop walks (tap, 64-channel chunk).  A tiles are fetched with global_load_lds (16 B per
//   lane, direct to LDS); image-border taps and M/N tails are redirected, per lane, to a zeroed
//   16-byte page, so there is no materialised im2col and no padding of the tensors.
// * LDS image: [rows][64 bf16] (128-B rows), XOR-swizzled at 16-B granularity
//   (chunk ^= (row>>1)&7), applied on the *global source address* because LDS-DMA writes are
//   lane-linear, and again on the ds_read_b128 address -> conflict-free fragment reads.
// * Two LDS stages, one barrier per K-step: stage(k+1) is issued before compute(k).
// * MFMA orientation: weights are the A operand, pixels the B operand, so each lane ends up with
//   4 consecutive output channels of one pixel -> packed 8-byte LDS writes, then full-row 16-byte
//   coalesced NHWC stores.  Fused epilogues: +bias, ReLU, dropout (counter RNG), or the
//   ReLU-backward mask (grad * (act > 0) * scale) used by data-gradient calls.
// * blockIdx is remapped XCD-aware (bijective) so tiles that share pixels/halos share an L2.
//
// Reference parity: replaces cuDNN conv fwd / bwd-data + ATen relu_/threshold_backward and
// cuBLAS linear that the reference reaches through
// model/vgg16.py:9-13,71-79.
#include "dtp_common.h"
#include <cstdlib>
#include <type_traits>

struct IgemmParams {
  const char* A;
  const char* B;
  const float* bias;
  const char* mask_src;   // bf16 [M][ldo]: output is multiplied by (mask_src > 0) * mask_scale
  const unsigned char* mask_bits;   // the same mask as 1 bit per element (byte (m*ldo+n)/8, bit n%8); wins over mask_src
  unsigned char* bits_out;          // optional: (final bf16 output > 0) in that layout (ReLU mask for a later dgrad)
  char* out;              // bf16 or f32 [M][ldo]
  float* partial;         // split-K slabs [z][M][Ncols] f32 (when gridDim.y > 1)
  float* colsum_part;     // optional [tiles_m][Ncols] f32: column sums of the final bf16 output tile
                          // (bias gradient of the layer that consumes this output as its dY)
  const char* bn_y;       // optional bf16 [M][ldo] (with colsum_sq): this output is dz, the gradient of
  const float* bn_scale;  //   z = relu(y*scale + shift) (a training-mode BatchNorm + ReLU); the partial rows become
  const float* bn_shift;  //   (sum dyh, sum dyh*y) with dyh = dz * (y*scale + shift > 0)   [no mask when bn_scale is null]
  uint32_t* pool_codes;   // POOL kernels: `out` is the 2x2-max-pooled output [M/4][ldo], this the 4-bit codes [M/4][Ncols/8]
                          //   — the two reductions of the BatchNorm backward, which then skips its pass over (dz, y)
  const uint32_t* unpool_codes;   // optional [M][Ncols/8] 4-bit codes of the 2x2 max-pool whose OUTPUT this GEMM's rows are the
                          //   gradient of (a data gradient right above a pool): the epilogue routes every value to its
                          //   arg-max position and `out` is the FULL-RESOLUTION gradient [N][2H][2W][ldo] — the pool
                          //   backward kernel and the pooled gradient's round trip through HBM disappear
  int unpool_relu;        //   ... and clears values whose code says "max <= 0" (the ReLU in front of the pool)
  int colsum_sq;          // colsum_part is [tiles_m][2][Ncols]: sums AND sums of squares — the batch
                          // statistics of a BatchNorm that follows this convolution (bn.hip partial layout)
  const char* zero_page;
  int M, Ncols, C;
  int H, W;
  int lda, ldb, ldo;      // in elements
  int out_fp32;
  int relu;
  float mask_scale;
  float drop_p;
  unsigned long long seed;
  int a_bf8;                            // fp8 kernels: the A rows (x) are e5m2, not e4m3 (gradients)
  const float* dq_a;                    // optional device scalars: accumulator *= *dq_a * *dq_b (fp8 dequant)
  const float* dq_b;
  const unsigned long long* seed_ptr;   // optional device-resident base seed (hipGraph replay): seed += *seed_ptr
  int ksteps_total;
  int ksteps_per_split;
  int tiles_m, tiles_n;
  int pf_rows;            // PIPE 3, not POOL: the image loads past the last super-tile (issued anyway to keep the counted
                          // waits uniform) read super-tile 0 of the tile pf_rows rows further down — the tile this XCD
                          // starts next — so that its prologue finds the rows in L2; 0 = off
  int k_order;            // 3x3 reduction order: 0 = tap-major (tap, channel chunk), 1 = chunk-major (chunk, tap)
  unsigned long long* stamps;   // DEBUG (DTP_IGEMM_STAMPS): per workgroup {hw_id|xcc, t_entry, t_loop, t_loop_end, t_exit}
};

// 16x16x32 MFMA step on one 16-byte fragment pair.  FP8: the 16 bytes are 16 OCP e4m3 values, fed
// to two v_mfma_f32_16x16x32_fp8_fp8 (8 bytes per lane each); A and B use the same lane->k map, so
// the k order inside the 128-element chunk is irrelevant to the dot product.
// MODE 0: bf16; 1: both operands e4m3; 2: x (the activation / gradient rows) in e5m2 ("bf8"), w in e4m3
template <int MODE>
DTP_DEVINL f32x4_t mma16(const bf16x8_t& w, const bf16x8_t& x, f32x4_t c) {
  if constexpr (MODE != 0) {
    typedef long l2_t __attribute__((ext_vector_type(2)));
    const l2_t wl = __builtin_bit_cast(l2_t, w), xl = __builtin_bit_cast(l2_t, x);
    if constexpr (MODE == 1) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl[0], xl[0], c, 0, 0, 0);
      return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl[1], xl[1], c, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(wl[0], xl[0], c, 0, 0, 0);
      return __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(wl[1], xl[1], c, 0, 0, 0);
    }
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, c, 0, 0, 0);
  }
}

template <int BM, int BN, int WGM, int WGN, bool TAPS9, int PIPE, bool FP8 = false, bool POOL = false>
__global__ __launch_bounds__(WGM * WGN * 64) void igemm_nt_kernel(const IgemmParams p) {
  static_assert(!POOL || PIPE == 3, "the fused max-pool rides on the shared-image schedule");
  static_assert(!FP8 || (PIPE == 0 && !TAPS9), "fp8 operands: plain GEMM, 2-stage loop only");
  static_assert(PIPE != 3 || TAPS9, "shared-image schedule is a 3x3 convolution schedule");
  constexpr int ES = FP8 ? 1 : 2;     // operand element size; a K-step is always 128 bytes per row
  constexpr int T = WGM * WGN * 64;
  constexpr int RP = T / 8;          // tile rows covered by one staging pass (8 lanes per 128-B row)
  constexpr int AP = BM / RP;
  constexpr int BP = BN / RP;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int STAGE = (BM + BN) * 128;
  static_assert(AP >= 1 && BP >= 1, "tile too small for the thread count");
  static_assert(BM * BN * 2 <= 2 * STAGE, "C image must fit in the staging buffers");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int kbeg = blockIdx.y * p.ksteps_per_split;
  int kend = kbeg + p.ksteps_per_split;
  if (kend > p.ksteps_total) kend = p.ksteps_total;
#define DTP_STAMP(i) if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime();
  if (p.stamps && tid == 0)
    p.stamps[(size_t)blockIdx.x * 8] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                       ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  DTP_STAMP(1)

  // ---------------- staging set-up (per-lane source pointers, fixed over the K loop)
  const int srow = tid >> 3;
  const int pch = tid & 7;
  const int lch = pch ^ ((srow >> 1) & 7);   // logical 16-B chunk this lane fetches
  const char* aptr[AP];
  unsigned amask[AP];
  if (TAPS9) {
    // (h, w) of this lane's first row by two integer divisions, of its other rows (RP pixels further
    // each) incrementally; the 9-bit tap-validity mask is the outer product of 3 row bits and 3
    // column bits.  (Dividing per row and testing per tap was ~900 VALU instructions per wave:
    // most of the 3.8 us prologue of a 256x256 tile, in-kernel stamps.)
    const int hw = p.H * p.W;
    const int step_h = RP / p.W, step_w = RP - step_h * p.W;       // uniform
    const int mfirst = m0 + srow;
    const int rem = mfirst % hw;
    int h = rem / p.W;
    int w = rem - h * p.W;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = mfirst + i * RP;
      const unsigned colbits = (w > 0 ? 1u : 0u) | 2u | (w < p.W - 1 ? 4u : 0u);
      unsigned mk = colbits << 3;
      if (h > 0) mk |= colbits;
      if (h < p.H - 1) mk |= colbits << 6;
      amask[i] = m < p.M ? mk : 0u;
      aptr[i] = p.A + ((size_t)m * p.C + lch * 8) * 2;
      w += step_w;
      h += step_h;
      if (w >= p.W) { w -= p.W; ++h; }
      while (h >= p.H) h -= p.H;
    }
  } else {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int m = m0 + srow + i * RP;
      amask[i] = m < p.M ? 1u : 0u;
      aptr[i] = p.A + (size_t)m * p.lda * ES + lch * 16;
    }
  }
  const char* bptr[BP];
  bool bok[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    const int n = n0 + srow + i * RP;
    bok[i] = n < p.Ncols;
    bptr[i] = p.B + (size_t)n * p.ldb * ES + lch * 16;
  }
  const int cchunks = p.C >> 6;   // 64-channel chunks per tap

  auto stage = [&](int kt, int buf) {
    int tap = 0, cc = kt;
    if (TAPS9) {
      if (p.k_order) { cc = kt / 9; tap = kt - cc * 9; }
      else { tap = kt / cchunks; cc = kt - tap * cchunks; }
    }
    long koffA, koffB;
    if (TAPS9) {
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      koffA = ((long)(dy * p.W + dx) * p.C + cc * 64) * 2;
      koffB = ((long)tap * p.C + cc * 64) * 2;
    } else {
      koffA = (long)cc * 128;
      koffB = (long)cc * 128;
    }
    char* lA = smem + buf * STAGE + wave * (8 * 128);
    char* lB = lA + BM * 128;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const bool ok = (amask[i] >> tap) & 1u;
      const char* g = ok ? aptr[i] + koffA : p.zero_page;
      glds16(g, lA + i * RP * 128);
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const char* g = bok[i] ? bptr[i] + koffB : p.zero_page;
      glds16(g, lB + i * RP * 128);
    }
  };

  // ---------------- fragment read addresses (per-lane constants)
  const int frow = lane & 15;
  const int fk = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  // byte offset inside a tile image for k-substep ks: row*128 + ((ks*4 + fk) ^ fsw)*16
  const int foff0 = frow * 128 + (((0 * 4 + fk) ^ fsw) << 4);
  const int foff1 = frow * 128 + (((1 * 4 + fk) ^ fsw) << 4);
  const int a_base = (wm * WM) * 128;              // pixel rows of this wave
  const int b_base = BM * 128 + (wn * WN) * 128;   // weight rows of this wave

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  auto compute_m = [&](int buf, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    const char* base = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int fo = ks ? foff1 : foff0;
      bf16x8_t xf[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) xf[i] = *(const bf16x8_t*)(base + a_base + i * 2048 + fo);
#pragma unroll
      for (int j = 0; j < NI; ++j) wf[j] = *(const bf16x8_t*)(base + b_base + j * 2048 + fo);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mma16<MODE>(wf[j], xf[i], acc[i][j]);
    }
  };
  auto compute = [&](int buf) {
    if constexpr (!FP8) compute_m(buf, std::integral_constant<int, 0>{});
    else if (p.a_bf8) compute_m(buf, std::integral_constant<int, 2>{});      // kernel-uniform
    else compute_m(buf, std::integral_constant<int, 1>{});
  };

  // Software-pipelined K-step (PIPE == 1): the wave's tile is cut into sub-steps
  // (k-substep x m-half); fragments of sub-step s+1 are read from LDS while the MFMAs of sub-step s
  // issue, and the LDS-DMA of the NEXT K-step (address selects + global_load_lds) is spread over
  // the sub-steps so its VALU/SALU work hides under the matrix pipe instead of preceding it.
  auto step = [&](int kt, int buf, auto pf_tag) {
    constexpr bool PF = decltype(pf_tag)::value;
    constexpr int MH = MI >= 2 ? 2 : 1;
    constexpr int MIH = MI / MH;
    constexpr int NSUB = 2 * MH;
    constexpr int NP = AP + BP;
    const char* base = smem + buf * STAGE;
    int tap = 0, cc = kt + 1;
    long koffA = 0, koffB = 0;
    char* lA = smem + (buf ^ 1) * STAGE + wave * (8 * 128);
    char* lB = lA + BM * 128;
    if (PF) {
      if (TAPS9) {
        if (p.k_order) { cc = (kt + 1) / 9; tap = (kt + 1) - cc * 9; }
        else { tap = (kt + 1) / cchunks; cc = (kt + 1) - tap * cchunks; }
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        koffA = ((long)(dy * p.W + dx) * p.C + cc * 64) * 2;
        koffB = ((long)tap * p.C + cc * 64) * 2;
      } else {
        koffA = (long)cc * 128;
        koffB = (long)cc * 128;
      }
    }
    bf16x8_t xa[2][MIH], wb[2][NI];
#pragma unroll
    for (int i = 0; i < MIH; ++i) xa[0][i] = *(const bf16x8_t*)(base + a_base + i * 2048 + foff0);
#pragma unroll
    for (int j = 0; j < NI; ++j) wb[0][j] = *(const bf16x8_t*)(base + b_base + j * 2048 + foff0);
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const int ks = s / MH, mh = s % MH;
      if (s + 1 < NSUB) {
        const int ks1 = (s + 1) / MH, mh1 = (s + 1) % MH;
        const int fo1 = ks1 ? foff1 : foff0;
#pragma unroll
        for (int i = 0; i < MIH; ++i)
          xa[(s + 1) & 1][i] = *(const bf16x8_t*)(base + a_base + (mh1 * MIH + i) * 2048 + fo1);
        if (ks1 != ks) {
#pragma unroll
          for (int j = 0; j < NI; ++j) wb[ks1 & 1][j] = *(const bf16x8_t*)(base + b_base + j * 2048 + fo1);
        }
      }
      if (PF) {
#pragma unroll
        for (int pc = (s * NP) / NSUB; pc < ((s + 1) * NP) / NSUB; ++pc) {
          if (pc < AP) {
            const bool ok = (amask[pc] >> tap) & 1u;
            const char* g = ok ? aptr[pc] + koffA : p.zero_page;
            glds16(g, lA + pc * RP * 128);
          } else {
            const char* g = bok[pc - AP] ? bptr[pc - AP] + koffB : p.zero_page;
            glds16(g, lB + (pc - AP) * RP * 128);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < MIH; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[mh * MIH + i][j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ks & 1][j], xa[s & 1][i], acc[mh * MIH + i][j], 0, 0, 0);
    }
  };

  // ======================= PIPE == 2: 8-phase schedule, LDS-DMA in flight across barriers ===========
  // (cdna_hip_programming.md "256^2 8-phase template", re-derived for the implicit GEMM.)
  // LDS = 8 half-tile slots of 16 KiB: [dbuf][A0, A1, B0, B1]; a half-tile holds, for every wave,
  // one half of its rows (A: 64 of its 128 pixels, B: 32 of its 64 channels) x 64 k.
  // A K-tile is consumed in four quadrant phases (A0B0, A0B1, A1B1, A1B0); each phase issues ONE
  // half-tile of LDS-DMA (2 global_load_lds per lane) for a tile 1.5-2 K-tiles ahead, waits with a
  // COUNTED vmcnt(8) (4 half-tiles stay in flight across the barriers; never vmcnt(0) in the loop),
  // and runs 16 MFMAs between two raw s_barriers.  The two wave rows (wm = 0/1) are staggered by
  // one barrier, so one group's ds_reads/address work overlaps the other group's MFMAs.
  // Hazard bookkeeping (derivation in docs/KERNELS.md):
  //   RAW  every slot is read >= 1 phase after the vmcnt that retires its DMA (+ a barrier);
  //   WAR  every slot is re-staged >= 2 phases after its last ds_read (covers the stagger).
  if constexpr (PIPE == 2) {
    // Two instances: 256x256 as 2x4 waves and 512x128 as 4x2 waves (outputs of <= 128 channels);
    // both give every wave a 128x64 tile, the smallest whose LDS fragment traffic (24 KB per 64
    // MFMAs) stays under the matrix pipe's time.
    static_assert(WGM * WGN == 8 && BM == WGM * 128 && BN == WGN * 64, "8-phase: 8 waves of 128x64 each");
    constexpr int NA = WGM;               // global_load_lds per lane for an A half-tile (64 rows a pass)
    constexpr int NB = WGN / 2;           // ... for a B half-tile
    constexpr int SA = WGM * 8192;        // A half-tile slot: 64 rows per wave row
    constexpr int SB = WGN * 4096;        // B half-tile slot: 32 rows per wave column
    constexpr int DB = 2 * SA + 2 * SB;   // one K-tile: [A0, A1, B0, B1]
    constexpr int VMC = 2 * (NA + NB);    // the 4 youngest half-tile issues (2 A + 2 B) may stay in flight
    const int grp = wm * 2 / WGM;         // stagger group: upper / lower half of the wave rows
    // B half-tile staging pointers: LDS row rho = 64*j + srow  <->  tile row ((rho>>5)<<6) + h*32 + (rho&31)
    const char* bp8[2][NB];
    bool bk8[2][NB];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int rho = 64 * j + srow;
        const int n = n0 + ((rho >> 5) << 6) + h * 32 + (rho & 31);
        bk8[h][j] = n < p.Ncols;
        bp8[h][j] = p.B + ((size_t)n * p.ldb + lch * 8) * 2;
      }
    auto koffs = [&](int T, int& tap, long& koffA, long& koffB) {
      int cc = T;
      tap = 0;
      if (TAPS9) {
        if (p.k_order) { cc = T / 9; tap = T - cc * 9; }
        else { tap = T / cchunks; cc = T - tap * cchunks; }
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        koffA = ((long)(dy * p.W + dx) * p.C + cc * 64) * 2;
        koffB = ((long)tap * p.C + cc * 64) * 2;
      } else {
        koffA = (long)cc * 128;
        koffB = (long)cc * 128;
      }
    };
    auto issue_A = [&](auto h_, auto d_, int T) {   // half-tile A_h of K-tile T -> slot [d][h]
      constexpr int h = decltype(h_)::value, d = decltype(d_)::value;
      int tap; long ka, kb;
      koffs(T, tap, ka, kb);
      const bool live = T < kend;
      char* dst = smem + d * DB + h * SA + wave * 1024;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const bool ok = live && ((amask[2 * j + h] >> tap) & 1u);
        const char* g = ok ? aptr[2 * j + h] + ka : p.zero_page;
        glds16(g, dst + j * 8192);
      }
    };
    auto issue_B = [&](auto h_, auto d_, int T) {
      constexpr int h = decltype(h_)::value, d = decltype(d_)::value;
      int tap; long ka, kb;
      koffs(T, tap, ka, kb);
      const bool live = T < kend;
      char* dst = smem + d * DB + 2 * SA + h * SB + wave * 1024;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const char* g = (live && bk8[h][j]) ? bp8[h][j] + kb : p.zero_page;
        glds16(g, dst + j * 8192);
      }
    };
    // fragment registers are plain named arrays used through macros (passing them to lambdas by
    // reference sent them to scratch)
    bf16x8_t xa[2][4], b0f[2][2], b1f[2][2];
#define DTP_READ_A(h, d)                                                                              \
  {                                                                                                   \
    const char* s_ = smem + (d) * DB + (h) * SA + (wm * 64) * 128;                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)    \
        xa[ks][i] = *(const bf16x8_t*)(s_ + i * 2048 + (ks ? foff1 : foff0));                         \
  }
#define DTP_READ_B(h, d, dst)                                                                         \
  {                                                                                                   \
    const char* s_ = smem + (d) * DB + 2 * SA + (h) * SB + (wn * 32) * 128;                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j)    \
        dst[ks][j] = *(const bf16x8_t*)(s_ + j * 2048 + (ks ? foff1 : foff0));                        \
  }
#define DTP_MMA(mh, nh, bf)                                                                           \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[(mh) * 4 + i][(nh) * 2 + j] =                 \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], xa[ks][i], acc[(mh) * 4 + i][(nh) * 2 + j], 0, 0, 0);
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
#define DTP_PHASE_SYNC_IN()                                   \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");  \
  __builtin_amdgcn_s_barrier();                               \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
  __builtin_amdgcn_sched_barrier(0);                          \
  __builtin_amdgcn_s_setprio(1);
#define DTP_PHASE_SYNC_OUT()                                  \
  __builtin_amdgcn_s_setprio(0);                              \
  __builtin_amdgcn_sched_barrier(0);                          \
  __builtin_amdgcn_s_barrier();                               \
  __builtin_amdgcn_sched_barrier(0);

    // prologue: tile kbeg entirely, tile kbeg+1's B0/A0 (issue order = consumption order)
    issue_B(I0{}, I0{}, kbeg);
    issue_A(I0{}, I0{}, kbeg);
    issue_B(I1{}, I0{}, kbeg);
    issue_A(I1{}, I0{}, kbeg);
    issue_B(I0{}, I1{}, kbeg + 1);
    issue_A(I0{}, I1{}, kbeg + 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();    // stagger the second group of wave rows by one barrier
    __builtin_amdgcn_sched_barrier(0);

    DTP_STAMP(2)
    int t = kbeg;
    for (; t < kend; t += 2) {
      // ---- K-tile t (dbuf 0)
      DTP_READ_A(0, 0) DTP_READ_B(0, 0, b0f)           // P1
      issue_B(I1{}, I1{}, t + 1);
      DTP_PHASE_SYNC_IN(); DTP_MMA(0, 0, b0f) DTP_PHASE_SYNC_OUT();
      DTP_READ_B(1, 0, b1f)                                // P2
      issue_A(I1{}, I1{}, t + 1);
      DTP_PHASE_SYNC_IN(); DTP_MMA(0, 1, b1f) DTP_PHASE_SYNC_OUT();
      DTP_READ_A(1, 0)                                     // P3
      issue_B(I0{}, I0{}, t + 2);
      DTP_PHASE_SYNC_IN(); DTP_MMA(1, 1, b1f) DTP_PHASE_SYNC_OUT();
      issue_A(I0{}, I0{}, t + 2);                             // P4
      DTP_PHASE_SYNC_IN(); DTP_MMA(1, 0, b0f) DTP_PHASE_SYNC_OUT();
      // ---- K-tile t+1 (dbuf 1).  With an odd tile count this is an all-zero tile (loads past kend
      // come from the zero page): a mid-loop exit here makes hipcc spill the accumulators.
      DTP_READ_A(0, 1) DTP_READ_B(0, 1, b0f)           // P5
      issue_B(I1{}, I0{}, t + 2);
      DTP_PHASE_SYNC_IN(); DTP_MMA(0, 0, b0f) DTP_PHASE_SYNC_OUT();
      DTP_READ_B(1, 1, b1f)                                // P6
      issue_A(I1{}, I0{}, t + 2);
      DTP_PHASE_SYNC_IN(); DTP_MMA(0, 1, b1f) DTP_PHASE_SYNC_OUT();
      DTP_READ_A(1, 1)                                     // P7
      issue_B(I0{}, I1{}, t + 3);
      DTP_PHASE_SYNC_IN(); DTP_MMA(1, 1, b1f) DTP_PHASE_SYNC_OUT();
      issue_A(I0{}, I1{}, t + 3);                             // P8
      DTP_PHASE_SYNC_IN(); DTP_MMA(1, 0, b0f) DTP_PHASE_SYNC_OUT();
    }
    DTP_STAMP(3)
    if (grp == 0) __builtin_amdgcn_s_barrier();    // pair the stagger barrier of the other group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain (dummy) DMA before the C image reuses LDS
#undef DTP_PHASE_SYNC_IN
#undef DTP_READ_A
#undef DTP_READ_B
#undef DTP_MMA
#undef DTP_PHASE_SYNC_OUT
  }

  // ======================= PIPE == 3: 8-phase schedule over ONE activation image per tap ROW ===========
  // The three dx taps of a filter row read the same 64-channel activation rows shifted by one pixel:
  // tile row R, tap (dy, dx) = flat pixel m0 + R + dy*W + dx.  PIPE 2 stages those rows three times
  // (3 x 32 KB per tap row and channel chunk); here they are staged ONCE as an image of 258 rows
  // (pixels m0-1+dy*W .. m0+256+dy*W) and the fragment reads of tap dx start dx rows further down.
  // A-operand LDS-DMA (and L2 reads) drop from 12 to 5 global_load_lds per lane and 3 K-tiles; the
  // kernels are power-limited (docs/ROUND_NOTES.md), so fewer bytes through L2 and LDS per MFMA is
  // what buys clock.  (Diagnostic upper bound, round 3: same kernel with the dx = +-1 A loads simply
  // skipped: +8 % forward, +10-16 % data gradient on the 256/512-channel layers.)
  //  * dy borders (and M tails) are decided per IMAGE ROW at staging time as before (zero page);
  //  * dx borders cannot be: an image row is a legitimate operand for two of its three consumers.
  //    They are per (output pixel, tap) = per lane and fragment: lanes whose pixel sits at w == 0
  //    (dx = -1) or w == W-1 (dx = +1) read their fragment from a zeroed LDS region instead — ONE
  //    v_cndmask per ds_read_b128 on the address, selected by wave masks kept in SGPR pairs;
  //  * K order: (channel chunk, dy) super-tiles of three K-tiles (dx = -1, 0, +1); B (weights) keeps
  //    PIPE 2's half-tile slots and issue points; the image of super-tile S+1 is issued in the even
  //    phases of super-tile S (chunk 0 + halo, 1, 2, 3: rows 64c .. 64c+63; halo = rows 256, 257);
  //  * the loop body is two super-tiles = 24 phases (image double buffer x B double buffer), every
  //    phase with its own COUNTED vmcnt (derivation: docs/KERNELS.md "shared activation image").
  if constexpr (PIPE == 3) {
    // Two geometries, both 8 waves whose tile is 64 channels wide: 256x256 as 2x4 waves of 128x64 (cfg 18)
    // and 384x128 as 4x2 waves of 96x64 (cfg 19, outputs of <= 128 channels: two 520-row images + the
    // weights of a 512x128 tile do not fit the 160 KB of LDS).
    static_assert(TAPS9 && WGM * WGN == 8 && BN == WGN * 64 && BM == WGM * (BM / WGM) && ((BM / WGM) % 32) == 0 &&
                  BM % 64 == 0, "shared-image schedule: 8 waves, 64 output channels per wave");
    constexpr int WR = BM / WGM;              // pixel rows per wave row (128 / 96)
    constexpr int FH = WR / 32;               // 16-row fragments per half of a wave's rows (4 / 3)
    constexpr int HS = FH * 2048;             // bytes between the two halves' first rows
    constexpr int NCH = BM / 64;              // 64-row staging chunks per image (4 / 6), + the halo chunk
    constexpr int IMG = (BM + 8) * 128;       // one image: BM + 2 rows used, BM + 8 staged (wave-granular DMA)
    constexpr int SB = WGN * 4096;            // B half-tile slot: 32 rows per wave column (as PIPE 2)
    constexpr int BOFF = 2 * IMG;             // LDS: [img 0][img 1][B: dbuf x {B0, B1}][zeros]
    constexpr int ZOFF = BOFF + 4 * SB;
    constexpr int ZSZ = (2 * FH - 1) * 2048 + 512;     // largest fragment-read immediate + a row, rounded
    static_assert(ZOFF + ZSZ <= 160 * 1024 && IMG + ZSZ < 65536 && ZSZ >= 7 * 1024, "LDS budget / ds_read immediates / halo dump");
    constexpr int NB = WGN / 2;
    const int NS = 3 * cchunks;               // super-tiles (host guarantees: even, no split-K)
    const int grp = wm * 2 / WGM;             // stagger group: upper / lower half of the wave rows
    // image chunks per even-phase slot of a super-tile (slot order (j0,q2) (j0,q4) (j1,q2) (j1,q4) (j2,q2)
    // (j2,q4)); the halo rides with chunk 0; the last chunk goes out at (j1,q4): 5 phases before its first read
    constexpr int L0 = NCH == 4 ? 2 : 3, L1 = NCH == 4 ? 1 : 2, L2 = 1, L3 = 1, L4 = 0, L5 = 0;
    static_assert((NCH == 4 || NCH == 6) && L0 + L1 + L2 + L3 + L4 + L5 == NCH + 1, "chunk -> slot plan");
    // counted waits (loads that may stay in flight behind the one the NEXT phase reads; derivation below)
    constexpr int VQ1_0 = L4 + L5 + 2 * NB, VQ1_1 = L0 + L1 + 2 * NB, VQ1_2 = L2 + L3 + 2 * NB;
    constexpr int VQ4_0 = L5 + L0 + L1 + 2 * NB, VQ4_1 = L1 + L2 + L3 + 2 * NB, VQ4_2 = L4 + L5 + 2 * NB;

    // ---- zero region (also the dump target of the halo DMA slot of waves 1..7)
    for (int o = tid * 16; o < ZSZ; o += T * 16) *(u32x4_t*)(smem + ZOFF + o) = (u32x4_t){0u, 0u, 0u, 0u};

    // ---- staging through buffer descriptors (as wgrad2.hip): per-lane 32-bit byte offsets that never
    // change, wave-uniform offsets in the scalar operand, rows outside the tensor / past Ncols zero-filled
    // by the hardware's range check; an invalid image row (dy border, M tail) is an out-of-range offset.
    // Image row r = 64c + srow (c < NCH; c == NCH: halo rows BM, BM + 1 = lanes 0..15 of wave 0) holds the
    // pixel m0 - 1 + r + dy*W; the descriptor starts W + 1 pixels before the tile so that every scalar
    // offset is non-negative (what lies before the tensor is never addressed: those rows are invalid).
    // POOL (conv + ReLU + 2x2 max-pool in one kernel): output rows are enumerated (n, h/2, w, h&1) —
    // m = ((n*(H/2) + h/2)*W + w)*2 + (h&1) — so a pooling window is four CONSECUTIVE tile rows
    // (4k: (h, w), 4k+1: (h+1, w), 4k+2: (h, w+1), 4k+3: (h+1, w+1)) and never straddles a tile, whatever
    // W is, while the dx taps stay a uniform shift of the staged image: TWO rows instead of one.  The
    // pooled pixels of a tile are consecutive rows of the pooled NHWC tensor (m / 4).  The tensors
    // themselves keep their layout: only the row -> pixel map of the loaders and the epilogue changes.
    constexpr int DXR = POOL ? 2 : 1;         // image rows per dx step = halo rows on either side
    // Bank swizzle of the IMAGE (16-byte chunk ^= swz_img(row)): the fragment reads of the three taps start 0,
    // DXR, 2*DXR rows into it, and the swizzle of PIPE 2 ((row >> 1) & 7, conflict-free for ALIGNED 16-row
    // fragments) is 2-way conflicted for every shifted one under gfx950's ds_read_b128 lane groups ({0-3, 12-15,
    // 20-27}, {4-11, 16-19, 28-31}, ...: rows {0-3, 12-15} of k-chunk fk meet rows {4-11} of fk ^ 1) — measured
    // 34 % of the LDS cycles of cfg 18 as bank conflicts (profiles/81_*) against 2-5 % for PIPE 2.  Exhaustive
    // search over the GF(2)-linear maps of the row bits (bench/lds_bank_sim.py --shifted): 192 of 32,768 are
    // conflict-free for shifts {0, 1, 2}; this one also for {0, 2, 4} (POOL).  B keeps PIPE 2's swizzle.
    auto swz_img = [](int row) -> int { return (((row >> 2) & 1) << 1) | (((row >> 1) & 1) << 2); };
    const int lch_img = pch ^ swz_img(srow);
    const int hw3 = p.H * p.W;
    const int mp0 = m0 - DXR + srow;          // output row whose dx = 0 tap reads image row srow
    unsigned am = 0;                          // 3 bits per chunk: row valid for dy = -1, 0, +1
    unsigned kpack = 0;                       // POOL: 4 bits per chunk: image-row wraps since chunk 0
    long a_first;                             // descriptor base, bytes from p.A (may lie before the tensor)
    unsigned a_voff;
    if constexpr (!POOL) {
      const int step_h = 64 / p.W, step_w = 64 - step_h * p.W;
      const int rem = (mp0 + hw3) % hw3;       // mp0 >= -1: pixel -1 walks on like the last pixel of an image
      int h = rem / p.W;
      int w = rem - h * p.W;
#pragma unroll
      for (int c = 0; c <= NCH; ++c) {
        const int m = mp0 + c * 64;
        const bool ok = m >= 0 && m < p.M && (c < NCH || tid < 16);
        const unsigned bits = (h > 0 ? 1u : 0u) | 2u | (h < p.H - 1 ? 4u : 0u);
        am |= (ok ? bits : 0u) << (3 * c);
        w += step_w;
        h += step_h;
        if (w >= p.W) { w -= p.W; ++h; }
        while (h >= p.H) h -= p.H;
      }
      a_first = (long)(m0 - 1 - p.W) * p.C * 2;
      a_voff = (unsigned)((srow * p.C + lch_img * 8) * 2);
    } else {
      const int H2 = p.H >> 1;
      const int r = mp0 & 1;                           // row of the pair (two's complement: -1 -> 1, -2 -> 0)
      const int qs = (mp0 >> 1) + H2 * p.W;            // (w, pair) index, shifted by one image: >= 0
      const int gs = qs / p.W;
      int w = qs - gs * p.W;
      int hp = gs % H2;
      const int pix0 = (2 * (gs - H2) + r) * p.W + w;  // this lane's chunk-0 pixel in the tensor's own (n, h, w) order
      // wave-uniform: first pixel the tile's dy = 0 taps can touch
      const int qf = ((m0 - DXR) >> 1) + H2 * p.W;
      const int gf = qf / p.W;
      const int base_pix = 2 * (gf - H2) * p.W + (qf - gf * p.W);
      const int step_g = 32 / p.W, step_w = 32 - step_g * p.W;
      int kacc = 0;
#pragma unroll
      for (int c = 0; c <= NCH; ++c) {
        const int m = mp0 + c * 64;
        const int h = 2 * hp + r;
        const bool ok = m >= 0 && m < p.M && (c < NCH || tid < 32);
        const unsigned bits = (h > 0 ? 1u : 0u) | 2u | (h < p.H - 1 ? 4u : 0u);
        am |= (ok ? bits : 0u) << (3 * c);
        kpack |= (unsigned)(kacc & 15) << (4 * c);
        int inc = step_g;
        w += step_w;
        if (w >= p.W) { w -= p.W; ++inc; }
        hp += inc;
        while (hp >= H2) hp -= H2;
        kacc += inc;
      }
      a_first = (long)(base_pix - p.W) * p.C * 2;             // the dy = -1 taps read one image row earlier
      a_voff = (unsigned)(((pix0 - base_pix) * p.C + lch_img * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(p.A + a_first, (long)p.M * p.C * 2 - a_first);
    // bytes between two chunks' rows: 64 pixels, or (POOL) 32 pixel columns + the image rows wrapped (per lane)
    const unsigned cstride = (unsigned)((POOL ? 32 : 64) * p.C * 2);
    const unsigned wrap_bytes = (unsigned)(p.W * p.C * 2);
    auto issue_img = [&](auto c_, auto ib_, int dyi, unsigned soff) {          // chunk c of the super-tile at (dyi, soff)
      constexpr int c = decltype(c_)::value, ib = decltype(ib_)::value;
      unsigned vo = a_voff;
      if constexpr (POOL) vo += ((kpack >> (4 * c)) & 15u) * wrap_bytes;
      vo = ((am >> (3 * c + dyi)) & 1u) ? vo : DTP_OOB_OFF;
      const int dst = c < NCH ? ib * IMG + c * 8192 + wave * 1024
                              : (wave == 0 ? ib * IMG + NCH * 8192 : ZOFF + (wave - 1) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (__attribute__((address_space(3))) void*)(smem + dst), 16, vo,
                                               soff + c * cstride, 0, 0);
    };
    auto issue_slot = [&](auto s_, auto ib_, int dyi, unsigned soff) {         // even-phase slot s of a super-tile
      constexpr int sl = decltype(s_)::value;
      using IC = std::integral_constant<int, 0>;
      (void)sizeof(IC);
      if constexpr (NCH == 4) {
        if constexpr (sl == 0) { issue_img(std::integral_constant<int, 0>{}, ib_, dyi, soff); issue_img(std::integral_constant<int, 4>{}, ib_, dyi, soff); }
        if constexpr (sl == 1) issue_img(std::integral_constant<int, 1>{}, ib_, dyi, soff);
        if constexpr (sl == 2) issue_img(std::integral_constant<int, 2>{}, ib_, dyi, soff);
        if constexpr (sl == 3) issue_img(std::integral_constant<int, 3>{}, ib_, dyi, soff);
      } else {
        if constexpr (sl == 0) {
          issue_img(std::integral_constant<int, 0>{}, ib_, dyi, soff);
          issue_img(std::integral_constant<int, 6>{}, ib_, dyi, soff);
          issue_img(std::integral_constant<int, 1>{}, ib_, dyi, soff);
        }
        if constexpr (sl == 1) { issue_img(std::integral_constant<int, 2>{}, ib_, dyi, soff); issue_img(std::integral_constant<int, 3>{}, ib_, dyi, soff); }
        if constexpr (sl == 2) issue_img(std::integral_constant<int, 4>{}, ib_, dyi, soff);
        if constexpr (sl == 3) issue_img(std::integral_constant<int, 5>{}, ib_, dyi, soff);
      }
    };
    // B half-tiles: LDS row rho = 64*j + srow <-> tile row ((rho>>5)<<6) + h*32 + (rho&31), as PIPE 2
    const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(p.B + (long)n0 * p.ldb * 2, (long)(p.Ncols - n0) * p.ldb * 2);
    const unsigned b_voff = (unsigned)(((((srow >> 5) << 6) + (srow & 31)) * p.ldb + lch * 8) * 2);
    const unsigned bstride_h = (unsigned)(32 * p.ldb * 2), bstride_j = (unsigned)(128 * p.ldb * 2);
    auto issue_B3 = [&](auto h_, auto d_, unsigned kb) {                       // half-tile B_h of the K-tile at column kb
      constexpr int h = decltype(h_)::value, d = decltype(d_)::value;
      const int dst = BOFF + d * (2 * SB) + h * SB + wave * 1024;
#pragma unroll
      for (int j = 0; j < NB; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (__attribute__((address_space(3))) void*)(smem + dst + j * 8192), 16,
                                                 b_voff, kb + h * bstride_h + j * bstride_j, 0, 0);
    };
    // wave-uniform K bookkeeping: super-tile S = 3*cc + dyi; its K-tiles' weight columns start at
    // kb(S) + j*2C; its image at soffset si(S) (chunk c: + c*cstride)
    const unsigned c2 = (unsigned)(p.C * 2);
    auto kb_of = [&](int S) -> unsigned { const int cc = S / 3, dyi = S - cc * 3; return (unsigned)((3 * dyi * p.C + cc * 64) * 2); };
    auto si_of = [&](int S) -> unsigned { const int cc = S / 3, dyi = S - cc * 3; return (unsigned)((dyi * p.W * p.C + cc * 64) * 2); };
    auto dy_of = [&](int S) -> int { return S - (S / 3) * 3; };
    // ---- fragment read addresses: image row of tile row R for tap dx is R + 1 + dx
    const unsigned lds0 = lds_addr(smem);
    // (computed by late_setup(), behind the prologue's DMA issue: this arithmetic then runs under the loads'
    // latency instead of in front of it — at one workgroup per CU nothing else covers either)
    unsigned fb[3][2];
    const unsigned zb[2] = {lds0 + ZOFF, lds0 + ZOFF - IMG};       // + ib*IMG + immediate lands in the zero region
    // ---- dx border masks: fragment f = FH*h + i of this wave row = tile rows wm*WR + 16*f + frow.  A
    // fragment's 16 pixels are lanes 0..15 (lanes l, l+16, l+32, l+48 hold the same pixel), so a mask is a
    // 16-bit pattern: 16 patterns packed in 8 SGPRs, widened to a wave mask where they are used (SALU)
    unsigned pm_m1[FH], pm_p1[FH];
    auto late_setup = [&]() {
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) {
        const int rr = frow + DXR * dxi;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          fb[dxi][ks] = lds0 + (wm * WR + rr) * 128 + (((ks * 4 + fk) ^ swz_img(rr)) << 4);
      }
      // (POOL: a tile row is half a pixel column step: 16 rows = 8 columns)
      const int step16 = (POOL ? 8 : 16) % p.W;
      int w = (POOL ? (m0 + wm * WR + frow) >> 1 : m0 + wm * WR + frow) % p.W;
#pragma unroll
      for (int f = 0; f < 2 * FH; ++f) {
        const unsigned b0 = (unsigned)__builtin_amdgcn_ballot_w64(w == 0) & 0xFFFFu;
        const unsigned b1 = (unsigned)__builtin_amdgcn_ballot_w64(w == p.W - 1) & 0xFFFFu;
        if (f & 1) { pm_m1[f >> 1] |= b0 << 16; pm_p1[f >> 1] |= b1 << 16; }
        else { pm_m1[f >> 1] = b0; pm_p1[f >> 1] = b1; }
        w += step16;
        if (w >= p.W) w -= p.W;
      }
    };
    auto wave_mask = [](unsigned packed, int f) -> unsigned long long {
      const unsigned p16 = (f & 1) ? packed >> 16 : packed & 0xFFFFu;
      const unsigned m32 = p16 | (p16 << 16);
      return ((unsigned long long)m32 << 32) | m32;
    };
    bf16x8_t xa[2][FH], b0f[2][2], b1f[2][2];
    typedef const __attribute__((address_space(3))) bf16x8_t* lds_frag_t;
#if defined(DTP_DIAG_NOSEL)   // diagnostic build (wrong math at the w borders): no zero-region select
#define DTP3_SEL(dst, normal, zero, mask) (void)0
#elif defined(DTP_DIAG_SELSAME)   // diagnostic build (wrong math): the select runs but picks the lane's own address
#define DTP3_SEL(dst, normal, zero, mask) asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(dst) : "v"(normal), "v"(normal), "s"(mask))
#else
#define DTP3_SEL(dst, normal, zero, mask) asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(dst) : "v"(normal), "v"(zero), "s"(mask))
#endif
#define DTP3_READ_A(h, ib, dxi)                                                                       \
  {                                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < FH; ++i) { \
      unsigned a_ = fb[dxi][ks];                                                                      \
      if ((dxi) == 0) DTP3_SEL(a_, fb[dxi][ks], zb[ib], wave_mask(pm_m1[(FH * (h) + i) >> 1], FH * (h) + i)); \
      if ((dxi) == 2) DTP3_SEL(a_, fb[dxi][ks], zb[ib], wave_mask(pm_p1[(FH * (h) + i) >> 1], FH * (h) + i)); \
      xa[ks][i] = *(lds_frag_t)(size_t)(a_ + ((ib) * IMG + (h) * HS + i * 2048));                     \
    }                                                                                                 \
  }
#define DTP3_READ_B(h, d, dst)                                                                        \
  {                                                                                                   \
    const char* s_ = smem + BOFF + (d) * (2 * SB) + (h) * SB + (wn * 32) * 128;                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j)    \
        dst[ks][j] = *(const bf16x8_t*)(s_ + j * 2048 + (ks ? foff1 : foff0));                        \
  }
#define DTP3_MMA(mh, nh, bf)                                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < FH; ++i)     \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[(mh) * FH + i][(nh) * 2 + j] =                \
          __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], xa[ks][i], acc[(mh) * FH + i][(nh) * 2 + j], 0, 0, 0);
#define DTP3_SYNC_IN(VM)                                      \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");   \
  __builtin_amdgcn_s_barrier();                               \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          \
  __builtin_amdgcn_sched_barrier(0);                          \
  __builtin_amdgcn_s_setprio(1);
#define DTP3_SYNC_OUT()                                       \
  __builtin_amdgcn_s_setprio(0);                              \
  __builtin_amdgcn_sched_barrier(0);                          \
  __builtin_amdgcn_s_barrier();                               \
  __builtin_amdgcn_sched_barrier(0);
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>;
    using J3 = std::integral_constant<int, 3>;
    using J4 = std::integral_constant<int, 4>;
    // One K-tile = four quadrant phases (A0B0, A0B1, A1B1, A1B0).  j = dx + 1, ib = image buffer,
    // d = B double buffer of this K-tile (T & 1), S / T = its super-tile / K-tile index.
    // DMA issue points:  q1: B1(T+1)   q2: image slot   q3: B0(T+2)   q4: image slot
    //   image slots of a super-tile (next image, buffer 1-ib): (j0,q2) chunk 0 + halo, (j0,q4) chunk 1,
    //   (j1,q2) chunk 2, (j1,q4) chunk 3, (j2,*) nothing.
    // Counted waits (loads that may stay in flight behind the one the NEXT phase reads), with L_s = image
    // loads of slot s and NB = loads of a B half-tile (256x256: L = 2 1 1 1 0 0, NB = 2 -> 4 7 6 / 7 7 4):
    //   after q1: B1(T), issued at q1 of T-1:   the two slots of T-1 + 2 NB
    //   after q4: B0(T+1), issued at q3 of T-1: slot (T-1,q4) + the two slots of T + 2 NB; after (j2,q4) the
    //             whole next image as well (its last chunk went out at (j1,q4)): L4 + L5 + 2 NB
    //   q2, q3: the next phase reads nothing new (image rows are static within a super-tile)
    // q3 and q4 are ONE phase (A1B1 + A1B0 behind the same barrier: A1B0 reads nothing new); both issues of the
    // pair go out in front of it IN THE SAME ORDER, so the counted waits above are unchanged: 6 barrier pairs per
    // two K-tiles instead of 8 (+0.3-0.8 % isolated; the same merge is worth +2-4 % in wgrad_tt8_kernel).
#define DTP3_TILE(j, ib, d, KB1, KB2, VQ1, VQ4, SLOT2, SLOT4)                                         \
      DTP3_READ_A(0, ib, j) DTP3_READ_B(0, d, b0f)                                     /* q1 */       \
      issue_B3(J1{}, std::integral_constant<int, 1 - (d)>{}, KB1);                                    \
      DTP3_SYNC_IN(VQ1); DTP3_MMA(0, 0, b0f) DTP3_SYNC_OUT();                                         \
      DTP3_READ_B(1, d, b1f)                                                           /* q2 */       \
      SLOT2                                                                                           \
      DTP3_SYNC_IN(63); DTP3_MMA(0, 1, b1f) DTP3_SYNC_OUT();                                          \
      DTP3_READ_A(1, ib, j)                                                            /* q3 + q4 */  \
      issue_B3(J0{}, std::integral_constant<int, (d)>{}, KB2);                                        \
      SLOT4                                                                                           \
      DTP3_SYNC_IN(VQ4); DTP3_MMA(1, 1, b1f) DTP3_MMA(1, 0, b0f) DTP3_SYNC_OUT();
    // one super-tile: kb = its weight column, kbn / sin / dyn = weight column, image offset, dy index of the NEXT one
#define DTP3_SLOT(sl, ib, dyn, sin) issue_slot(std::integral_constant<int, sl>{}, std::integral_constant<int, 1 - (ib)>{}, dyn, sin);
#define DTP3_SUPER(ib, d0, kb, kbn, sin, dyn)                                                         \
      DTP3_TILE(0, ib, d0, (kb) + c2, (kb) + 2 * c2, VQ1_0, VQ4_0, DTP3_SLOT(0, ib, dyn, sin), DTP3_SLOT(1, ib, dyn, sin)) \
      DTP3_TILE(1, ib, 1 - (d0), (kb) + 2 * c2, (kbn), VQ1_1, VQ4_1, DTP3_SLOT(2, ib, dyn, sin), DTP3_SLOT(3, ib, dyn, sin)) \
      DTP3_TILE(2, ib, d0, (kbn), (kbn) + c2, VQ1_2, VQ4_2, , )

    // prologue (issue order = the steady state's: B0(0); image 0; B1(0); B0(1))
    issue_B3(J0{}, J0{}, 0u);
    issue_slot(J0{}, J0{}, 0, 0u);
    issue_slot(J1{}, J0{}, 0, 0u);
    issue_slot(J2{}, J0{}, 0, 0u);
    issue_slot(J3{}, J0{}, 0, 0u);
    issue_B3(J1{}, J0{}, 0u);
    issue_B3(J0{}, J1{}, c2);
    late_setup();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the zero-region stores
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();    // stagger the second wave row by one barrier
    __builtin_amdgcn_sched_barrier(0);

    DTP_STAMP(2)
    for (int S = 0; S < NS; S += 2) {
      // (loads issued for super-tiles >= NS at the tail read in-range garbage or hardware zeros into
      // slots nobody reads any more; they only keep the counted waits uniform)
      const unsigned kb0 = kb_of(S), kb1 = kb_of(S + 1), kb2 = kb_of(S + 2);
      const bool pf = !POOL && p.pf_rows > 0 && S + 2 >= NS;      // last iteration: prefetch for the XCD's next tile
      const unsigned si1 = si_of(S + 1), si2 = pf ? (unsigned)p.pf_rows * c2 : si_of(S + 2);
      const int dy1 = dy_of(S + 1), dy2 = pf ? 0 : dy_of(S + 2);
      DTP3_SUPER(0, 0, kb0, kb1, si1, dy1)         // K-tiles 3S .. 3S+2   (B dbuf 0, 1, 0)
      DTP3_SUPER(1, 1, kb1, kb2, si2, dy2)         // K-tiles 3S+3 .. 3S+5 (B dbuf 1, 0, 1)
    }
    DTP_STAMP(3)
    if (grp == 0) __builtin_amdgcn_s_barrier();    // pair the stagger barrier of the other group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain (dummy) DMA before the C image reuses LDS
#undef DTP3_SUPER
#undef DTP3_SLOT
#undef DTP3_TILE
#undef DTP3_SYNC_OUT
#undef DTP3_SYNC_IN
#undef DTP3_MMA
#undef DTP3_READ_B
#undef DTP3_READ_A
#undef DTP3_SEL
  }

  // ---------------- main loop: 2 LDS stages, one barrier per K-step
  if (PIPE < 2 && kbeg < kend) stage(kbeg, 0);
  int buf = 0;
  if (PIPE >= 2) {
    // handled above
  } else if (PIPE == 0) {
    for (int kt = kbeg; kt < kend; ++kt) {
      sync_lds_dma();    // stage kt landed everywhere; everyone is done reading the other buffer
      if (kt + 1 < kend) stage(kt + 1, buf ^ 1);
      compute(buf);
      buf ^= 1;
    }
  } else {
    for (int kt = kbeg; kt < kend - 1; ++kt) {
      sync_lds_dma();
      step(kt, buf, std::true_type{});
      buf ^= 1;
    }
    if (kbeg < kend) {
      sync_lds_dma();
      step(kend - 1, buf, std::false_type{});
    }
  }

  // ---------------- epilogue
  // the un-pooling store loop exists in the 3x3 bf16 kernels only, and not in the two-stage / pipelined 256x256 ones
  // (256 VGPRs and ~100 SGPRs already: its few extra kernel-uniform values made hipcc spill prologue values there)
  constexpr bool UNPOOL = TAPS9 && !FP8 && !POOL && !(BM == 256 && BN == 256 && PIPE < 2);
  if (p.dq_a) {                        // kernel-uniform: fp8 dequantisation of the accumulators
    const float dq = p.dq_a[0] * p.dq_b[0];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] *= dq;
  }
  const int prow = lane & 15;          // pixel within a 16-block
  const int cq = (lane >> 4) * 4;      // first of 4 consecutive channels within a 16-block
  if (gridDim.y > 1 || p.out_fp32) {
    // direct register -> global f32 (split-K slabs, or small f32 outputs such as the logits)
    float* dst = gridDim.y > 1 ? p.partial + (size_t)blockIdx.y * p.M * p.Ncols : (float*)p.out;
    const int ld = gridDim.y > 1 ? p.Ncols : p.ldo;
    const bool fuse = gridDim.y == 1;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WM + i * 16 + prow;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WN + j * 16 + cq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e < p.Ncols) {
            float v = acc[i][j][e];
            if (fuse) {
              if (p.bias) v += p.bias[n + e];
              if (p.relu) v = fmaxf(v, 0.f);
            }
            dst[(size_t)m * ld + n + e] = v;
          }
        }
      }
    }
    return;
  }

  // ---- store-loop geometry and the 1-bit ReLU masks of this lane's 16-byte chunks: the byte loads
  // are issued here, before the C image is written, so their latency hides under that phase
  constexpr int CPR = BN / 8;                 // 16-B chunks per C row
  constexpr int TOTAL = BM * CPR;
  // T % CPR == 0: a lane keeps its 8-column chunk q and walks rows pl0, pl0 + G, ...
  constexpr int G = T / CPR;
  constexpr int ITER = TOTAL / T;
  static_assert(T % CPR == 0 && TOTAL % T == 0, "store loop geometry");
  const int q = tid % CPR, pl0 = tid / CPR;
  const int n = n0 + q * 8;
  const bool n_ok = n < p.Ncols;
  // The store loop used to be one pass per row with the kernel-uniform feature tests (mask kind,
  // 1-bit masks out, column sums) and the row / column tail tests inside it: ~270 instructions and
  // one exposed LDS read latency per 16-byte store, 6,500 cycles per 256x256 tile whether 49 or
  // 256 CUs were storing (in-kernel stamps: issue-bound, not a write burst).  Now: four LDS reads at
  // a time, the optional masking as separate unrolled passes, then four stores back to back through
  // a buffer descriptor over this tile's valid rows (row tail = out of range = dropped by the
  // hardware, column tail = out-of-range offset), row stride in the scalar offset: 3,000 cycles.
  const int rows_valid = (p.M - m0) < BM ? (p.M - m0) : BM;
  const long tile_bytes = (long)rows_valid * p.ldo * 2;
  const unsigned voff = n_ok ? (unsigned)((pl0 * p.ldo + n) * 2) : DTP_OOB_OFF;
  const int row_step = G * p.ldo * 2;         // bytes between two rows of this lane (kernel-uniform)
  uint32_t mb[ITER];
  if (p.mask_bits) {
    const __amdgpu_buffer_rsrc_t mb_rs = make_rsrc(p.mask_bits + (((size_t)m0 * p.ldo) >> 3), tile_bytes >> 4);
#pragma unroll
    for (int it = 0; it < ITER; ++it)
      mb[it] = __builtin_amdgcn_raw_buffer_load_b8(mb_rs, (int)(n_ok ? voff >> 4 : DTP_OOB_OFF), it * (row_step >> 4), 0);
  } else if (UNPOOL && p.unpool_codes) {
    // un-pooling: the 4-bit arg-max codes of this lane's chunks, 4 bytes per 16-byte chunk, in the same registers
    const __amdgpu_buffer_rsrc_t cd_rs = make_rsrc((const char*)p.unpool_codes + (((size_t)m0 * p.ldo) >> 1), tile_bytes >> 2);
#pragma unroll
    for (int it = 0; it < ITER; ++it)
      mb[it] = __builtin_amdgcn_raw_buffer_load_b32(cd_rs, (int)(n_ok ? voff >> 2 : DTP_OOB_OFF), it * (row_step >> 2), 0);
  }
  // raw barriers: hipcc's __syncthreads() waits vmcnt(0) first, which would expose the mask loads above
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // all waves finished reading the staging buffers
  __builtin_amdgcn_sched_barrier(0);
  DTP_STAMP(5)
  {
    const float keep_scale = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    const unsigned long long seed = (p.drop_p > 0.f && p.seed_ptr) ? p.seed + *p.seed_ptr : p.seed;
    // all bias values of this lane in ONE batch of 16-byte loads (Ncols % 8 == 0 on this path, so a
    // 4-channel group is entirely inside or outside): loading them per j, right before their use,
    // cost one exposed global-load latency per j — 5 us of an 8-phase tile's 7.5 us epilogue
    f32x4_t bvv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * WN + j * 16 + cq;
      bvv[j] = (p.bias && n < p.Ncols) ? *(const f32x4_t*)(p.bias + n) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    if (p.drop_p > 0.f) {          // kernel-uniform: classifier layers with dropout, generic path
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int nl = wn * WN + j * 16 + cq;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int pl = wm * WM + i * 16 + prow;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[i][j][e] + bvv[j][e];
            if (p.relu) x = fmaxf(x, 0.f);
            const uint64_t idx = (uint64_t)(m0 + pl) * (uint64_t)p.Ncols + (uint64_t)(n0 + nl + e);
            v[e] = rng_uniform(seed, idx) >= p.drop_p ? x * keep_scale : 0.f;
          }
          u32x2_t pk;
          pk[0] = pack_bf16x2(v[0], v[1]);
          pk[1] = pack_bf16x2(v[2], v[3]);
          const int q = nl >> 3;
          const int addr = pl * (BN * 2) + ((q ^ (pl & 7)) << 4) + ((nl >> 2) & 1) * 8;
          *(u32x2_t*)(smem + addr) = pk;
        }
      }
    } else {
      // Fast path (every convolution): 128 accumulator values per lane at one wave-instruction per
      // 4 cycles made this phase VALU-bound (4.5 us of a 256x256 tile).  Vector adds (v_pk_add_f32),
      // one v_cvt_pk_bf16_f32 per pair, and ReLU AFTER the rounding as a packed signed-16-bit max
      // on the bf16 bit patterns (negative floats are negative int16s; round(relu(x)) ==
      // relu(round(x))), with the ReLU switch hoisted out of the loops.
      auto body = [&](auto relu_tag) {
        constexpr bool RELU = decltype(relu_tag)::value;
        typedef short s16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int nl = wn * WN + j * 16 + cq;
          const int q = nl >> 3, sub = ((nl >> 2) & 1) * 8;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int pl = wm * WM + i * 16 + prow;
            const f32x4_t a = acc[i][j] + bvv[j];
            // (scalars, not pk[h] in a loop: hipcc folded the subscripted form into pk[1] = pk[0])
            uint32_t lo = pack_bf16x2(a[0], a[1]), hi = pack_bf16x2(a[2], a[3]);
            if (RELU) {
              lo = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, lo), (s16x2_t){0, 0}));
              hi = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, hi), (s16x2_t){0, 0}));
            }
            const u32x2_t pk = {lo, hi};
            *(u32x2_t*)(smem + pl * (BN * 2) + ((q ^ (pl & 7)) << 4) + sub) = pk;
          }
        }
      };
      if (p.relu) body(std::true_type{});
      else body(std::false_type{});
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // C image complete
  __builtin_amdgcn_sched_barrier(0);
  DTP_STAMP(6)
  if constexpr (POOL) {
    // 2x2 max-pool of the (bias + ReLU, bf16) C image: window k = tile rows 4k (h, w), 4k+2 (h, w+1),
    // 4k+1 (h+1, w), 4k+3 (h+1, w+1); same 4-bit codes {bit 2: max > 0, bits 1:0: FIRST maximum in that
    // order} as maxpool2x2_fwd_kernel (pool.hip), so the pool backward is the existing one.  A thread
    // owns one 16-byte channel chunk and walks the windows: pooled rows go out as coalesced 16-byte
    // stores, the full-resolution activation never leaves the CU.
    constexpr int CPRp = BN / 8, NW = BM / 4;
    static_assert(T % CPRp == 0 && (NW * CPRp) % T == 0, "pooled store geometry");
    const int qq = tid % CPRp;
    const int nq = n0 + qq * 8;
    const int wins_valid = ((p.M - m0) < BM ? (p.M - m0) : BM) >> 2;
    const long win0 = (long)m0 >> 2;
    const int cvn = p.Ncols >> 3;
#pragma unroll
    for (int k = tid / CPRp; k < NW; k += T / CPRp) {
      u32x4_t v[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int pl = 4 * k + r4;
        v[r4] = *(const u32x4_t*)(smem + pl * (BN * 2) + ((qq ^ (pl & 7)) << 4));
      }
      u32x4_t o;
      uint32_t code = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const float fa = hf ? __uint_as_float(v[0][e] & 0xFFFF0000u) : __uint_as_float(v[0][e] << 16);
          const float fb = hf ? __uint_as_float(v[2][e] & 0xFFFF0000u) : __uint_as_float(v[2][e] << 16);
          const float fc = hf ? __uint_as_float(v[1][e] & 0xFFFF0000u) : __uint_as_float(v[1][e] << 16);
          const float fd = hf ? __uint_as_float(v[3][e] & 0xFFFF0000u) : __uint_as_float(v[3][e] << 16);
          uint32_t arg = 0; float mx = fa;
          if (fb > mx) { mx = fb; arg = 1; }
          if (fc > mx) { mx = fc; arg = 2; }
          if (fd > mx) { mx = fd; arg = 3; }
          code |= (arg | (mx > 0.f ? 4u : 0u)) << (4 * (2 * e + hf));
          const uint32_t bits = __float_as_uint(mx);          // exact: the inputs are bf16
          if (hf) o[e] |= bits & 0xFFFF0000u; else o[e] = bits >> 16;
        }
      }
      if (k < wins_valid && nq < p.Ncols) {
        *(u32x4_t*)(p.out + ((win0 + k) * p.ldo + nq) * 2) = o;
        p.pool_codes[(win0 + k) * cvn + (nq >> 3)] = code;
      }
    }
    return;
  }
  {
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float cq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bsc[8], bsh[8];                     // bn_y mode: scale / shift of this lane's 8 channels
    if (p.bn_y && p.bn_scale) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bsc[e] = n_ok ? p.bn_scale[n + e] : 0.f;
        bsh[e] = n_ok ? p.bn_shift[n + e] : 0.f;
      }
    }
    const __amdgpu_buffer_rsrc_t o_rs = make_rsrc(p.out + (size_t)m0 * p.ldo * 2, tile_bytes);
    const bool unit_scale = p.mask_scale == 1.0f;      // kernel-uniform (every convolution)
    // Un-pooling epilogue (p.unpool_codes, kernel-uniform): tile row m is the pooled pixel (img*H + oh)*W + ow whose 2x2
    // window starts at pixel 4m - 2ow of the full-resolution gradient [N][2H][2W][ldo].  A lane walks rows G apart, so it
    // carries (ow, byte offset of its window) from row to row instead of dividing per row; offsets are relative to the
    // window of tile row 0 (the map m -> 4m - 2ow is monotonic, so they are >= 0 and < (4 BM + 2W) pixels).
    __amdgpu_buffer_rsrc_t up_rs = o_rs;
    unsigned up_off = 0;
    int up_ow = 0, up_gm = 0;
    uint32_t up_keep = 0, up_flip = 0;
    if (UNPOOL && p.unpool_codes) {
      const int ow_m0 = m0 % p.W;
      up_ow = (m0 + pl0) % p.W;
      up_gm = G % p.W;
      const long base_px = 4l * m0 - 2l * ow_m0;
      up_rs = make_rsrc(p.out + (size_t)base_px * p.ldo * 2, (4l * p.M - base_px) * p.ldo * 2);
      up_off = (unsigned)((4 * pl0 - 2 * (up_ow - ow_m0)) * p.ldo + n) * 2u;
      up_keep = p.unpool_relu ? 0x00070007u : 0x00030003u;     // code nibble: bits 1:0 arg-max, bit 2 "max > 0"
      up_flip = p.unpool_relu ? 0x00040004u : 0u;              // bit 2 clear -> the value matches no window position
    }
    constexpr int CH = ITER < 4 ? ITER : 4;     // rows per lane and pass: 4 LDS reads in flight, 4 stores back to back
    static_assert(ITER % CH == 0, "store loop chunking");
#pragma unroll
    for (int c0 = 0; c0 < ITER; c0 += CH) {
    u32x4_t v[CH];
#pragma unroll
    for (int it = 0; it < CH; ++it) {
      const int pl = pl0 + (c0 + it) * G;
      v[it] = *(const u32x4_t*)(smem + pl * (BN * 2) + ((q ^ (pl & 7)) << 4));
    }
    if (UNPOOL && p.unpool_codes) {
      typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
      typedef short s16x2_t __attribute__((ext_vector_type(2)));
      const int row_b = p.ldo * 2, row_c = 2 * p.W * p.ldo * 2;      // (h, w+1) and (h+1, w) of the window, bytes
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const int pl = pl0 + (c0 + it) * G;
        const unsigned off = (n_ok && pl < rows_valid) ? up_off : DTP_OOB_OFF;
        const uint32_t code = mb[c0 + it];
        u32x4_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // both bf16 halves of a word at once: A = (arg-max | positive << 2) of channel 2e in the low half, of
          // channel 2e+1 in the high half; half == k  <=>  (half ^ k) - 1 underflows  -> arithmetic >> 15 gives 0xFFFF
          const uint32_t c8 = (code >> (8 * e)) & 0xFFu;
          const uint32_t A = (((c8 << 12) | c8) & up_keep) ^ up_flip;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const u16x2_t t = __builtin_bit_cast(u16x2_t, A ^ (uint32_t)(k * 0x00010001u)) - (u16x2_t){1, 1};
            const uint32_t mk = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2_t, t) >> (s16x2_t){15, 15});
            o[k][e] = v[it][e] & mk;
          }
          v[it][e] = o[0][e] | o[1][e] | o[2][e] | o[3][e];       // what the column sums below see: the routed values
        }
        __builtin_amdgcn_raw_buffer_store_b128(o[0], up_rs, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o[1], up_rs, (int)off, row_b, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o[2], up_rs, (int)off, row_c, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o[3], up_rs, (int)off, row_c + row_b, 0);
        int ow2 = up_ow + up_gm;
        ow2 -= ow2 >= p.W ? p.W : 0;
        up_off += (unsigned)((4 * G - 2 * (ow2 - up_ow)) * p.ldo * 2);
        up_ow = ow2;
      }
    } else {
    if (p.mask_bits) {
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        if (unit_scale) {
          // keep / clear each bf16 by AND with a mask built from two sign-extended bit extracts
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int s0 = (int)(mb[c0 + it] << (31 - 2 * e)) >> 31;        // bit 2e   -> 0 / ~0
            const int s1 = (int)(mb[c0 + it] << (30 - 2 * e)) >> 31;        // bit 2e+1 -> 0 / ~0
            v[it][e] &= ((uint32_t)s0 & 0xFFFFu) | ((uint32_t)s1 & 0xFFFF0000u);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t vw = v[it][e];
            float lo = ((mb[c0 + it] >> (2 * e)) & 1u) ? bf16_to_f32((bf16_t)(vw & 0xFFFFu)) * p.mask_scale : 0.f;
            float hi = ((mb[c0 + it] >> (2 * e + 1)) & 1u) ? bf16_to_f32((bf16_t)(vw >> 16)) * p.mask_scale : 0.f;
            v[it][e] = pack_bf16x2(lo, hi);
          }
        }
      }
    } else if (p.mask_src) {
      const __amdgpu_buffer_rsrc_t ms_rs = make_rsrc(p.mask_src + (size_t)m0 * p.ldo * 2, tile_bytes);
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const u32x4_t mk = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ms_rs, (int)voff, (c0 + it) * row_step, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t mw = mk[e], vw = v[it][e];
          // bf16 > 0  <=>  sign clear and magnitude non-zero
          const bool lo_pos = ((mw & 0x8000u) == 0) && ((mw & 0x7FFFu) != 0);
          const bool hi_pos = ((mw & 0x80000000u) == 0) && ((mw & 0x7FFF0000u) != 0);
          float lo = lo_pos ? bf16_to_f32((bf16_t)(vw & 0xFFFFu)) * p.mask_scale : 0.f;
          float hi = hi_pos ? bf16_to_f32((bf16_t)(vw >> 16)) * p.mask_scale : 0.f;
          v[it][e] = pack_bf16x2(lo, hi);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < CH; ++it) __builtin_amdgcn_raw_buffer_store_b128(v[it], o_rs, (int)voff, (c0 + it) * row_step, 0);
    if (p.bits_out) {
      const __amdgpu_buffer_rsrc_t bo_rs = make_rsrc(p.bits_out + (((size_t)m0 * p.ldo) >> 3), tile_bytes >> 4);
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const u32x4_t x = v[it];
        const uint32_t b = bf16x2_pos_bits(x[0]) | (bf16x2_pos_bits(x[1]) << 2) | (bf16x2_pos_bits(x[2]) << 4) | (bf16x2_pos_bits(x[3]) << 6);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)b, bo_rs, (int)(n_ok ? voff >> 4 : DTP_OOB_OFF), (c0 + it) * (row_step >> 4), 0);
      }
    }
    }
    if (p.colsum_part && p.bn_y) {       // kernel-uniform: BatchNorm-backward reductions of this dz tile
      const __amdgpu_buffer_rsrc_t by_rs = make_rsrc(p.bn_y + (size_t)m0 * p.ldo * 2, tile_bytes);
      u32x4_t yv[CH];
#pragma unroll
      for (int it = 0; it < CH; ++it)
        yv[it] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(by_rs, (int)voff, (c0 + it) * row_step, 0));
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        if (n_ok && pl0 + (c0 + it) * G < rows_valid) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ylo = __uint_as_float(yv[it][e] << 16), yhi = __uint_as_float(yv[it][e] & 0xFFFF0000u);
            float dlo = __uint_as_float(v[it][e] << 16), dhi = __uint_as_float(v[it][e] & 0xFFFF0000u);
            if (p.bn_scale) {
              dlo = fmaf(ylo, bsc[2 * e], bsh[2 * e]) > 0.f ? dlo : 0.f;
              dhi = fmaf(yhi, bsc[2 * e + 1], bsh[2 * e + 1]) > 0.f ? dhi : 0.f;
            }
            cs[2 * e] += dlo;
            cs[2 * e + 1] += dhi;
            cq[2 * e] = fmaf(dlo, ylo, cq[2 * e]);
            cq[2 * e + 1] = fmaf(dhi, yhi, cq[2 * e + 1]);
          }
        }
      }
    } else if (p.colsum_part) {
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        if (n_ok && pl0 + (c0 + it) * G < rows_valid) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(v[it][e] << 16), hi = __uint_as_float(v[it][e] & 0xFFFF0000u);
            cs[2 * e] += lo;
            cs[2 * e + 1] += hi;
            if (p.colsum_sq) {       // kernel-uniform
              cq[2 * e] = fmaf(lo, lo, cq[2 * e]);
              cq[2 * e + 1] = fmaf(hi, hi, cq[2 * e + 1]);
            }
          }
        }
      }
    }
    }
    DTP_STAMP(4)
    if (p.colsum_part) {   // kernel-uniform
      // T % CPR == 0: a thread keeps the same 8-column chunk q for all its rows; fold the T/CPR
      // row-groups through LDS (the C image is dead after the barrier), one partial row per tile
      static_assert(BN <= T, "colsum epilogue geometry");
      __syncthreads();
      float* red = (float*)smem;
      const int g = pl0;
#pragma unroll
      for (int e = 0; e < 8; ++e) red[g * BN + q * 8 + e] = cs[e];
      __syncthreads();
      if (tid < BN && n0 + tid < p.Ncols) {
        float s = 0.f;
#pragma unroll 4
        for (int k = 0; k < G; ++k) s += red[k * BN + tid];
        p.colsum_part[(size_t)tm * (p.colsum_sq ? 2 : 1) * p.Ncols + n0 + tid] = s;
      }
      if (p.colsum_sq) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[g * BN + q * 8 + e] = cq[e];
        __syncthreads();
        if (tid < BN && n0 + tid < p.Ncols) {
          float s = 0.f;
#pragma unroll 4
          for (int k = 0; k < G; ++k) s += red[k * BN + tid];
          p.colsum_part[((size_t)tm * 2 + 1) * p.Ncols + n0 + tid] = s;
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// split-K reduce + fused epilogue: out = epi(sum_z partial[z])
__global__ void igemm_splitk_reduce_kernel(const float* __restrict__ partial, int nsplit, int M, int Ncols,
                                           const float* __restrict__ bias, const bf16_t* __restrict__ mask_src,
                                           void* __restrict__ out, int ldo, int out_fp32, int relu,
                                           float mask_scale, float drop_p, unsigned long long seed,
                                           const unsigned long long* seed_ptr) {
  if (drop_p > 0.f && seed_ptr) seed += *seed_ptr;
  const size_t total = (size_t)M * Ncols;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / Ncols), n = (int)(i - (size_t)m * Ncols);
    float v = 0.f;
    for (int z = 0; z < nsplit; ++z) v += partial[(size_t)z * total + i];
    if (bias) v += bias[n];
    if (relu) v = fmaxf(v, 0.f);
    if (drop_p > 0.f) v = rng_uniform(seed, i) >= drop_p ? v * keep_scale : 0.f;
    const size_t o = (size_t)m * ldo + n;
    if (mask_src) v = bf16_to_f32(mask_src[o]) > 0.f ? v * mask_scale : 0.f;
    if (out_fp32) ((float*)out)[o] = v;
    else ((bf16_t*)out)[o] = f32_to_bf16(v);
  }
}

// -------------------------------------------------------------------------------------------
template <int BM, int BN, int WGM, int WGN, bool TAPS9, int PIPE, bool FP8 = false, bool POOL = false>
static int launch_cfg(const IgemmParams& p0, int nsplit, hipStream_t stream) {
  IgemmParams p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Ncols + BN - 1) / BN;
  constexpr int lds = PIPE == 3 ? 2 * (BM + 8) * 128 + 4 * (WGN * 4096) + ((2 * (BM / WGM / 32) - 1) * 2048 + 512)
                                : 2 * (BM + BN) * 128;
  static bool attr_set = false;
  auto kern = igemm_nt_kernel<BM, BN, WGM, WGN, TAPS9, PIPE, FP8, POOL>;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return -2;
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n, nsplit, 1);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, stream, p);
  return DTP_CHECK_LAUNCH();
}

extern "C" int dtp_igemm_nt(const void* A, const void* B, const float* bias, const void* mask_src,
                            const void* mask_bits, void* bits_out, void* out,
                            float* partial, float* colsum_part, const void* zero_page, int M, int Ncols, int C,
                            int taps, int H,
                            int W, int lda, int ldb, int ldo, int out_fp32, int relu, float mask_scale,
                            float drop_p, unsigned long long seed, const void* seed_ptr, int fp8,
                            const float* dq_a, const float* dq_b, int colsum_sq, const void* bn_y,
                            const float* bn_scale, const float* bn_shift, void* pool_codes,
                            const void* unpool_codes, int unpool_relu, int nsplit, int cfg,
                            hipStream_t stream) {
  // fp8 != 0: 8-bit operands, A [M][lda] (fp8 == 1: OCP e4m3, fp8 == 2: e5m2), B [Ncols][ldb] e4m3, bytes; K = C a multiple of 128
  if (M <= 0 || Ncols <= 0 || C <= 0 || (C & (fp8 ? 127 : 63))) return -10;   // K chunks are 128 bytes wide
  if (fp8 && (taps != 1 || (lda & 15) || (ldb & 15))) return -18;
  if (taps != 1 && taps != 9) return -11;
  if (taps == 9 && (H <= 0 || W <= 0 || M % (H * W) != 0)) return -12;
  if (!out_fp32 && ((Ncols & 7) || (ldo & 7))) return -13;               // 16-byte bf16 row chunks
  if ((lda & 7) || (ldb & 7)) return -14;
  IgemmParams p;
  p.A = (const char*)A; p.B = (const char*)B; p.bias = bias; p.mask_src = (const char*)mask_src;
  p.out = (char*)out; p.partial = partial; p.zero_page = (const char*)zero_page;
  p.colsum_part = colsum_part;
  p.colsum_sq = colsum_part ? colsum_sq : 0;
  p.bn_y = (const char*)bn_y; p.bn_scale = bn_scale; p.bn_shift = bn_shift;
  p.pool_codes = (uint32_t*)pool_codes;
  p.unpool_codes = (const uint32_t*)unpool_codes; p.unpool_relu = unpool_relu;
  // un-pooling epilogue: 3x3 rows = pooled pixels, plain bf16 C-image store loop only (its offsets stay far below 2^31:
  // a tile spans < 4*BM + 2W full-resolution pixels)
  if (unpool_codes && (taps != 9 || out_fp32 || ldo != Ncols || mask_src || mask_bits || bits_out || bn_y || pool_codes ||
                       fp8 || drop_p > 0.f || relu || bias || (long)(4 * 512 + 2 * (long)W + 8) * ldo * 2 > 0x3FFFFFFFl))
    return -25;
  if ((cfg == 20 || cfg == 21) != (pool_codes != nullptr)) return -23;      // pooled output <-> pooling kernel
  if (bn_y && (!colsum_part || !colsum_sq || out_fp32 || (bn_scale && !bn_shift))) return -21;
  p.mask_bits = (const unsigned char*)mask_bits; p.bits_out = (unsigned char*)bits_out;
  p.M = M; p.Ncols = Ncols; p.C = C; p.H = H; p.W = W; p.lda = lda; p.ldb = ldb; p.ldo = ldo;
  p.out_fp32 = out_fp32; p.relu = relu; p.mask_scale = mask_scale; p.drop_p = drop_p; p.seed = seed;
  p.seed_ptr = (const unsigned long long*)seed_ptr;
  p.dq_a = dq_a; p.dq_b = dq_b;
  p.a_bf8 = fp8 == 2;
  static const int k_order_env = std::getenv("DTP_IGEMM_KORDER") ? std::atoi(std::getenv("DTP_IGEMM_KORDER")) : 1;
  p.k_order = k_order_env;
  static const int pf_env = std::getenv("DTP_IGEMM_PREFETCH_NEXT") ? std::atoi(std::getenv("DTP_IGEMM_PREFETCH_NEXT")) : 1;
  p.pf_rows = 0;      // set per configuration below
  p.stamps = nullptr;
  if (const char* e = std::getenv("DTP_IGEMM_STAMPS")) p.stamps = (unsigned long long*)std::strtoull(e, nullptr, 0);
  if (fp8 && (!dq_a || !dq_b)) return -19;
  p.ksteps_total = taps * (C >> (fp8 ? 7 : 6));
  if (nsplit < 1) nsplit = 1;
  if (nsplit > p.ksteps_total) nsplit = p.ksteps_total;
  p.ksteps_per_split = (p.ksteps_total + nsplit - 1) / nsplit;
  nsplit = (p.ksteps_total + p.ksteps_per_split - 1) / p.ksteps_per_split;
  if (nsplit > 1 && partial == nullptr) return -15;
  if (colsum_part && (nsplit > 1 || out_fp32)) return -17;   // only the bf16 C-image epilogue emits it
  if (unpool_codes && (nsplit > 1 || cfg == 0 || cfg == 8)) return -25;
  if ((mask_bits || bits_out) && (nsplit > 1 || out_fp32)) return -20;   // 1-bit masks: same epilogue only
  int rc;
  const bool t9 = taps == 9;
#define DTP_IG(BM_, BN_, WGM_, WGN_, PIPE_)                                     \
  (t9 ? launch_cfg<BM_, BN_, WGM_, WGN_, true, PIPE_>(p, nsplit, stream)        \
      : launch_cfg<BM_, BN_, WGM_, WGN_, false, PIPE_>(p, nsplit, stream))
  if (fp8) {
    switch (cfg) {
      case 1: rc = launch_cfg<256, 128, 4, 2, false, 0, true>(p, nsplit, stream); break;
      case 3: rc = launch_cfg<128, 128, 2, 2, false, 0, true>(p, nsplit, stream); break;
      case 4: rc = launch_cfg<128, 64, 2, 1, false, 0, true>(p, nsplit, stream); break;
      default: return -16;
    }
  } else
  switch (cfg) {   // cfg 0-4: plain 2-stage loop; 8-12: software-pipelined K-step
    case 0: rc = DTP_IG(256, 256, 2, 4, 0); break;
    case 1: rc = DTP_IG(256, 128, 4, 2, 0); break;
    case 2: rc = DTP_IG(256, 64, 4, 1, 0); break;
    case 3: rc = DTP_IG(128, 128, 2, 2, 0); break;
    case 4: rc = DTP_IG(128, 64, 2, 1, 0); break;
    case 8: rc = DTP_IG(256, 256, 2, 4, 1); break;
    case 9: rc = DTP_IG(256, 128, 4, 2, 1); break;
    case 10: rc = DTP_IG(256, 64, 4, 1, 1); break;
    case 11: rc = DTP_IG(128, 128, 2, 2, 1); break;
    case 12: rc = DTP_IG(128, 64, 2, 1, 1); break;
    case 16: rc = DTP_IG(256, 256, 2, 4, 2); break;   // 8-phase, counted vmcnt
    case 17: rc = DTP_IG(512, 128, 4, 2, 2); break;   // 8-phase for outputs of <= 128 channels
    case 18:        // 8-phase over one staged activation image per tap row (3x3 only, C % 128 == 0, no split-K)
      if (!t9 || nsplit != 1 || (C & 127)) return -22;
      if (pf_env) p.pf_rows = (32 / ((Ncols + 255) / 256 > 32 ? 32 : (Ncols + 255) / 256)) * 256;
      rc = launch_cfg<256, 256, 2, 4, true, 3>(p, nsplit, stream);
      break;
    case 19:        // the same schedule as a 384x128 tile (4x2 waves of 96x64) for outputs of <= 128 channels
      if (!t9 || nsplit != 1 || (C & 127)) return -22;
      if (pf_env) p.pf_rows = 32 * 384;
      rc = launch_cfg<384, 128, 4, 2, true, 3>(p, nsplit, stream);
      break;
    case 20: case 21:   // cfg 18 / 19 + ReLU + 2x2 max-pool in the epilogue: `out` is the POOLED tensor [M/4][ldo]
      if (!t9 || nsplit != 1 || (C & 127) || !relu || out_fp32 || (H & 1) || (W & 1) || W < 14 || mask_src || mask_bits ||
          bits_out || colsum_part || bn_y)
        return -24;
      rc = cfg == 20 ? launch_cfg<256, 256, 2, 4, true, 3, false, true>(p, nsplit, stream)
                     : launch_cfg<384, 128, 4, 2, true, 3, false, true>(p, nsplit, stream);
      break;
    default: return -16;
  }
#undef DTP_IG
  if (rc != 0) return rc;
  if (nsplit > 1) {
    const size_t total = (size_t)M * Ncols;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(igemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, partial, nsplit, M, Ncols,
                       bias, (const bf16_t*)mask_src, out, ldo, out_fp32, relu, mask_scale, drop_p, seed,
                       (const unsigned long long*)seed_ptr);
    return DTP_CHECK_LAUNCH();
  }
  return 0;
}
