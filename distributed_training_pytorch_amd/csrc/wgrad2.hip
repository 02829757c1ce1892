// Weight-gradient GEMM, second generation (gfx950): same math and LDS images as wgrad.hip
//
//   dW[m][tap*C + c] = sum_k  dY[k][m] * X[pix(k) + shift(tap)][c]          (3x3 conv, TAPS9)
//   dW[m][n]         = sum_k  dY[k][m] * X[k][n]                            (linear)
//
// but with the two things fixed that held wgrad.hip at 0.73-0.96 PF/s (measured causes, see
// docs/KERNELS.md "wgrad2"):
//
//  1. hipcc puts `s_waitcnt vmcnt(0)` in front of every `__builtin_amdgcn_ds_read_tr16_b64` that
//     follows an LDS-DMA (the builtin carries no memory operand, so the waitcnt pass assumes it
//     may read what the DMA is still writing): the prefetch of K-tile t+1 was drained before the
//     first fragment read of K-tile t, i.e. global->LDS loads and MFMAs never overlapped.  Here the
//     transposed reads are inline-asm `ds_read_b64_tr_b16` with hand-counted `s_waitcnt lgkmcnt(N)`
//     (form (ii) of cdna_hip_programming.md 5.7: the wait statement names every destination "+v"),
//     and the DMA of tile t+1 stays in flight under all 64 MFMAs of tile t.
//     (The transposing read itself is NOT slow: bench/micro/lds_probe.hip measures 270-285 B/clk
//     per CU, the same as ds_read_b64/b128; round 1's 53 B/clk was a VALU-bound micro-benchmark.)
//  2. Staging cost ~230 instructions per K-step with four exec-masked loops (per-row (h, w)
//     tracking, 64-bit pointer selects against a zero page).  Here LDS-DMA goes through a buffer
//     descriptor whose base advances by one K-step of rows in scalar registers; per-lane byte
//     offsets are constants, out-of-range lanes get zeros written by the hardware (K/M/N tails,
//     rows past the tensor), and the 3x3 border mask is ONE ballot per wave and K-step: lane l
//     tracks pixel (h, w) of K-step row l, each staging pass picks its rows' bits with a scalar
//     shift.  ~25 VALU, no branches.
//
// Reference parity: replaces cuDNN bwd-filter and the cuBLAS dW GEMMs
// triggered by loss.backward() at example_trainer.py:86.
#include "dtp_common.h"
#include <cstdlib>
#include <type_traits>

struct Wgrad2Params {
  const char* dY;     // [K][lda] bf16
  const char* X;      // [K][ldb] bf16 (conv: NHWC pixels x C)
  float* partial;     // [nsplit][M][Ntot] f32
  char* out;          // direct output (nsplit == 1)
  int K, M, Ntot;     // Ntot = taps * C (conv) or in_features (linear)
  int C, H, W;
  int lda, ldb;
  int ksteps_total, ksteps_per_split;
  int tiles_m, tiles_n;
  int out_fp32;
  float beta;
  int nwrap;          // how often (h >= H) can hold after advancing a pixel row by one K-step
  unsigned long long* stamps;   // DEBUG (-DDTP_WG2_STAMPS build + DTP_WG2_STAMPS=<address>): per workgroup
                                // {loop ticks, ticks in vmcnt(0), ticks in the barrier, K-steps, t_entry, t_exit}
};

template <int RB>
DTP_DEVINL int s_of2(int k) {   // swizzle (in 16-B chunks) of LDS row k for an image with RB-byte rows (= wgrad.hip s_of)
  if (RB >= 256) return ((k & 3) | (((k >> 3) & 1) << 2)) << 1;
  return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1;
}

template <int BM, int BN, int WGM, int WGN, bool TAPS9>
__global__ __launch_bounds__(WGM * WGN * 64) void wgrad_tt2_kernel(const Wgrad2Params p) {
  constexpr int T = WGM * WGN * 64;
  constexpr int BK = 64;
  constexpr int RBA = BM * 2, RBB = BN * 2;        // LDS row bytes
  constexpr int LA = BM / 8, LB = BN / 8;          // lanes (16-B chunks) per row
  constexpr int RPA = T / LA, RPB = T / LB;        // rows per staging pass
  constexpr int AP = BK / RPA, BP = BK / RPB;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int MIH = MI / 2;                      // dY fragments per (k-substep, m-half) sub-step
  constexpr int STAGE = BK * (RBA + RBB);
  constexpr int RWB = 64 / LB;                     // B rows covered by one wave-instruction
  static_assert(AP >= 1 && BP >= 1 && LA <= 64 && LB <= 64 && MI >= 2 && (MI % 2) == 0, "geometry");
  static_assert(2 * STAGE <= 160 * 1024 && 32 * RBA + 4 * RBA < 65536 && 32 * RBB + 4 * RBB < 65536, "LDS / imm range");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  const int ntiles = p.tiles_m * p.tiles_n;
  // 1-D grid over (k-split z, tile), tile fastest, XCD-remapped: all taps x channel blocks of one
  // k-chunk run on the same XCD at the same time and share its L2 (see wgrad.hip)
  const int nsplit_g = gridDim.x / ntiles;
  const int pair = xcd_remap(blockIdx.x, gridDim.x);
  const int zsplit = pair / ntiles;
  const int tile = pair - zsplit * ntiles;
  const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;
  const int m0 = tm * BM;
  int n0, tap = 0, c0 = 0;
  if (TAPS9) {
    const int cblocks = p.C / BN;
    tap = tn / cblocks;
    c0 = (tn - tap * cblocks) * BN;
    n0 = tap * p.C + c0;
  } else {
    n0 = tn * BN;
    c0 = n0;
  }
  const int dy = TAPS9 ? tap / 3 - 1 : 0, dx = TAPS9 ? tap % 3 - 1 : 0;

  const int kbeg = zsplit * p.ksteps_per_split;
  int kend = kbeg + p.ksteps_per_split;
  if (kend > p.ksteps_total) kend = p.ksteps_total;

  // ---------------- staging: constant per-lane byte offsets, descriptor bases advance per K-step
  const int ra = tid / LA, pca = tid % LA;
  const int lca = pca ^ s_of2<RBA>(ra);
  const bool a_col_ok = (m0 + lca * 8) < p.lda;
  const int rb = tid / LB, pcb = tid % LB;
  const int lcb = pcb ^ s_of2<RBB>(rb);
  const bool b_col_ok = TAPS9 ? true : (n0 + lcb * 8) < p.Ntot;
  const long ldb_e = TAPS9 ? p.C : p.ldb;
  unsigned a_voff[AP], b_voff[BP];
#pragma unroll
  for (int i = 0; i < AP; ++i)
    a_voff[i] = a_col_ok ? (unsigned)(((ra + i * RPA) * p.lda + lca * 8) * 2) : DTP_OOB_OFF;
#pragma unroll
  for (int i = 0; i < BP; ++i)
    b_voff[i] = b_col_ok ? (unsigned)(((long)(rb + i * RPB) * ldb_e + lcb * 8) * 2) : DTP_OOB_OFF;
  const unsigned b_lanebit = 1u << (lane / LB);
  int th = 0, tw = 0;                                 // pixel (h, w) of K-step row `lane` of the NEXT tile to stage
  if (TAPS9) {
    const int rem = (kbeg * BK + lane) % (p.H * p.W);
    th = rem / p.W;
    tw = rem - th * p.W;
  }
  const int step_dh = TAPS9 ? BK / p.W : 0;
  const int step_dw = TAPS9 ? BK - step_dh * p.W : 0;
  // scalar state: descriptor bases and the bytes left from them to the end of each tensor
  const long shift = TAPS9 ? (long)dy * p.W + dx : 0;
  const char* a_base = p.dY + ((long)kbeg * BK * p.lda + m0) * 2;
  long a_left = ((long)p.K - (long)kbeg * BK) * p.lda * 2 - (long)m0 * 2;
  const char* b_base = p.X + (((long)kbeg * BK + shift) * ldb_e + c0) * 2;
  long b_left = ((long)p.K - (long)kbeg * BK - shift) * ldb_e * 2 - (long)c0 * 2;
  const long a_step = (long)BK * p.lda * 2, b_step = (long)BK * ldb_e * 2;

  auto stage = [&](int buf) {      // stages the tile the scalar state points at, then advances it
    char* lA = smem + buf * STAGE + wave * 1024;
    char* lB = smem + buf * STAGE + BK * RBA + wave * 1024;
    const __amdgpu_buffer_rsrc_t ra_ = make_rsrc(a_base, a_left);
    const __amdgpu_buffer_rsrc_t rb_ = make_rsrc(b_base, b_left);
#pragma unroll
    for (int i = 0; i < AP; ++i) glds16_buf(ra_, a_voff[i], lA + i * (T * 16));
    if (TAPS9) {
      const bool valid = (unsigned)(th + dy) < (unsigned)p.H && (unsigned)(tw + dx) < (unsigned)p.W;
      const unsigned long long mask = __ballot(valid);
      tw += step_dw; th += step_dh;                 // row `lane` of the following K-step (branch-free)
      const bool cw = tw >= p.W;
      tw -= cw ? p.W : 0; th += cw ? 1 : 0;
      for (int r = 0; r < p.nwrap; ++r) th -= th >= p.H ? p.H : 0;
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const unsigned bits = (unsigned)(mask >> (wave * RWB + i * RPB));   // scalar shift: wave is uniform
        glds16_buf(rb_, (bits & b_lanebit) ? b_voff[i] : DTP_OOB_OFF, lB + i * (T * 16));
      }
    } else {
#pragma unroll
      for (int i = 0; i < BP; ++i) glds16_buf(rb_, b_voff[i], lB + i * (T * 16));
    }
    a_base += a_step; a_left -= a_step;
    b_base += b_step; b_left -= b_step;
  };

  // ---------------- transposed fragment read addresses (LDS byte addresses, buffer 0)
  const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int krow = 8 * g4 + q;                       // + 32*ks, +4 for the second read of a fragment
  const int sa = s_of2<RBA>(krow), sb = s_of2<RBB>(krow);
  const int a_cw = (wm * WM) >> 3, b_cw = (wn * WN) >> 3;
  const int sub = (pp >> 1), half8 = (pp & 1) * 8;
  const unsigned lds0 = lds_addr(smem);
  unsigned a_ad[MI], b_ad[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) a_ad[i] = lds0 + krow * RBA + ((((a_cw + 2 * i) ^ sa) + sub) << 4) + half8;
#pragma unroll
  for (int j = 0; j < NI; ++j) b_ad[j] = lds0 + BK * RBA + krow * RBB + ((((b_cw + 2 * j) ^ sb) + sub) << 4) + half8;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment registers: halves of a bf16x8 operand as the two 8-byte transposed reads deliver them
  u32x2_t xl[2][MIH], xh[2][MIH], wl[2][NI], wh[2][NI];

#define RD_A(slot, ks, mh)                                                                     \
  _Pragma("unroll") for (int i = 0; i < MIH; ++i)                                              \
      DTP_TR2(xl[slot][i], xh[slot][i], a_ad[(mh) * MIH + i], (ks) * 32 * RBA, RBA);
#define RD_B(slot, ks)                                                                         \
  _Pragma("unroll") for (int j = 0; j < NI; ++j) DTP_TR2(wl[slot][j], wh[slot][j], b_ad[j], (ks) * 32 * RBB, RBB);
// wait until at most N LDS reads of this wave are outstanding; naming the consumed registers "+v"
// keeps every use (and any compiler copy) of them behind the wait
#define WAIT_A(N, slot)                                                                        \
  _Pragma("unroll") for (int i = 0; i < MIH; ++i)                                              \
      DTP_LGKM_WAIT2(N, xl[slot][i], xh[slot][i]);
#define WAIT_B(N, slot)                                                                        \
  _Pragma("unroll") for (int j = 0; j < NI; ++j)                                               \
      DTP_LGKM_WAIT2(N, wl[slot][j], wh[slot][j]);
#define MMA(aslot, bslot, mh)                                                                  \
  _Pragma("unroll") for (int i = 0; i < MIH; ++i) {                                            \
    const bf16x8_t xa_ = frag_of(xl[aslot][i], xh[aslot][i]);                                  \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                           \
      const bf16x8_t wb_ = frag_of(wl[bslot][j], wh[bslot][j]);                                \
      acc[(mh) * MIH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb_, xa_, acc[(mh) * MIH + i][j], 0, 0, 0); \
    }                                                                                          \
  }
  constexpr int NRA = 2 * MIH, NRB = 2 * NI;      // LDS read instructions per RD_A / RD_B

#ifdef DTP_WG2_STAMPS
  unsigned long long st_vm = 0, st_bar = 0;
  const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif
  if (kbeg < kend) stage(0);
  int buf = 0;
#ifdef DTP_WG2_STAMPS
  const unsigned long long st_loop0 = __builtin_amdgcn_s_memtime();
#endif
  for (int kt = kbeg; kt < kend; ++kt) {
    // tile kt landed (own DMA: vmcnt; everybody's: barrier); everyone is done reading the other buffer
#ifdef DTP_WG2_STAMPS
    const unsigned long long st_a = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_b = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    const unsigned long long st_c = __builtin_amdgcn_s_memtime();
    st_vm += st_b - st_a;
    st_bar += st_c - st_b;
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_sched_barrier(0);
    // sub-steps (ks, mh): (0,0) (0,1) (1,0) (1,1); reads of sub-step s+1 are issued before the MFMAs of s
    RD_A(0, 0, 0)
    RD_B(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < kend) stage(buf ^ 1);             // DMA of tile kt+1: in flight under all MFMAs below
    __builtin_amdgcn_sched_barrier(0);
    RD_A(1, 0, 1)
    WAIT_A(NRA, 0) WAIT_B(NRA, 0)
    __builtin_amdgcn_sched_barrier(0);
    MMA(0, 0, 0)
    __builtin_amdgcn_sched_barrier(0);
    RD_A(0, 1, 0)
    RD_B(1, 1)
    WAIT_A((NRA + NRB > 15 ? 15 : NRA + NRB), 1)      // lgkmcnt is a 4-bit counter
    __builtin_amdgcn_sched_barrier(0);
    MMA(1, 0, 1)
    __builtin_amdgcn_sched_barrier(0);
    RD_A(1, 1, 1)
    WAIT_A(NRA, 0) WAIT_B(NRA, 1)
    __builtin_amdgcn_sched_barrier(0);
    MMA(0, 1, 0)
    __builtin_amdgcn_sched_barrier(0);
    WAIT_A(0, 1)
    __builtin_amdgcn_sched_barrier(0);
    MMA(1, 1, 1)
    __builtin_amdgcn_sched_barrier(0);
    buf ^= 1;
    // fragment addresses follow the buffer
    const int delta = buf ? STAGE : -STAGE;
#pragma unroll
    for (int i = 0; i < MI; ++i) a_ad[i] += delta;
#pragma unroll
    for (int j = 0; j < NI; ++j) b_ad[j] += delta;
  }
#undef RD_A
#undef RD_B
#undef WAIT_A
#undef WAIT_B
#undef MMA

#ifdef DTP_WG2_STAMPS
  const unsigned long long st_loop1 = __builtin_amdgcn_s_memtime();
  if (p.stamps && tid == 0) {
    unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
    o[0] = st_loop1 - st_loop0; o[1] = st_vm; o[2] = st_bar; o[3] = (unsigned long long)(kend - kbeg);
    o[4] = st_entry; o[5] = st_loop0;
  }
#endif
  // ---------------- epilogue: lane owns dW[m = .. + (lane&15)][n .. n+3]
  const bool direct = nsplit_g == 1;
  float* slab = p.partial + (size_t)zsplit * p.M * p.Ntot;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WM + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * WN + j * 16 + g4 * 4;
      if (n >= p.Ntot) continue;      // Ntot is a multiple of 8, n of 4: whole quad in or out
      const size_t o = (size_t)m * p.Ntot + n;
      f32x4_t v = acc[i][j];
      if (!direct) {
        *(f32x4_t*)(slab + o) = v;
      } else if (p.out_fp32) {
        float* d = (float*)p.out + o;
        if (p.beta != 0.f) {
          const f32x4_t old = *(const f32x4_t*)d;
          v = v + old * p.beta;
        }
        *(f32x4_t*)d = v;
      } else {
        bf16_t* d = (bf16_t*)p.out + o;
        if (p.beta != 0.f) {
          const u32x2_t old = *(const u32x2_t*)d;
          v[0] += p.beta * bf16_to_f32((bf16_t)(old[0] & 0xFFFFu));
          v[1] += p.beta * bf16_to_f32((bf16_t)(old[0] >> 16));
          v[2] += p.beta * bf16_to_f32((bf16_t)(old[1] & 0xFFFFu));
          v[3] += p.beta * bf16_to_f32((bf16_t)(old[1] >> 16));
        }
        u32x2_t pk;
        pk[0] = pack_bf16x2(v[0], v[1]);
        pk[1] = pack_bf16x2(v[2], v[3]);
        *(u32x2_t*)d = pk;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 8-phase weight gradient (round 3): the 256x256 tile on the schedule of igemm.hip PIPE 2 — 8 LDS slots
// [dbuf][A0, A1, B0, B1] of half-tiles (64 K-rows x 128 channels: A_h = columns wm*128 + h*64 + [0, 64) of
// dY for both wave rows, B_h = columns wn*64 + h*32 + [0, 32) of X for the four wave columns), quadrant order
// A0B0 -> A0B1 -> A1B1 -> A1B0 so one operand half stays in registers, every phase = issue ONE half-tile of
// LDS-DMA, counted `s_waitcnt vmcnt(8)` (four half-tiles stay in flight), raw s_barrier, 16 MFMAs, s_barrier;
// the two wave rows (= the two waves of every SIMD) are staggered by one barrier, so one's transposed reads
// and their LDS latency run under the other's MFMAs.  The last two quadrants (A1B1, A1B0) are ONE phase here:
// A1B0 reads nothing new, so its MFMAs join A1B1's behind the same barrier (both half-tile issues of the pair
// go out in front of it; the counted waits are unchanged because the issue ORDER is): 6 barriers-pairs per
// two K-tiles instead of 8 and a 32-MFMA block that covers the other group's 24-read phase: +2-4 % isolated.  Why now: in-kernel stamps of the 2-stage loop above
// (profiles/86_*) show a wave needing ~2,490 cycles for the 1,024 cycles of MFMAs of its K-step — 48
// transposed 8-byte reads with at most 16 in flight are three LDS round trips nothing of its own covers —
// and waiting another ~1,160 cycles in the K-step's barrier for its SIMD-mate to do the same.  Round 1's
// port of this schedule lost 5-9 % because hipcc drained vmcnt in front of every transposed-read builtin
// (found in round 2); here the reads are inline asm with explicit waits, as in the kernel above.
template <bool TAPS9>
__global__ __launch_bounds__(512) void wgrad_tt8_kernel(const Wgrad2Params p) {
  constexpr int BM = 256, BN = 256, BK = 64;
  constexpr int RB = 256;                  // LDS row bytes of a half-tile image [64 k][128 channels]
  constexpr int SLOT = BK * RB;            // 16 KB
  constexpr int DB = 4 * SLOT;             // one K-tile: [A0, A1, B0, B1]
  constexpr int VMC = 8;                   // 2 loads per lane and half-tile: the 4 youngest half-tiles stay in flight
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int grp = wm;

  const int ntiles = p.tiles_m * p.tiles_n;
  const int nsplit_g = gridDim.x / ntiles;
  const int pair = xcd_remap(blockIdx.x, gridDim.x);
  const int zsplit = pair / ntiles;
  const int tile = pair - zsplit * ntiles;
  const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;
  const int m0 = tm * BM;
  int n0, tap = 0, c0 = 0;
  if (TAPS9) {
    const int cblocks = p.C / BN;
    tap = tn / cblocks;
    c0 = (tn - tap * cblocks) * BN;
    n0 = tap * p.C + c0;
  } else {
    n0 = tn * BN;
    c0 = n0;
  }
  const int dy = TAPS9 ? tap / 3 - 1 : 0, dx = TAPS9 ? tap % 3 - 1 : 0;
  const int kbeg = zsplit * p.ksteps_per_split;
  int kend = kbeg + p.ksteps_per_split;
  if (kend > p.ksteps_total) kend = p.ksteps_total;

  // ---------------- staging: a wave-instruction covers 4 image rows (64 lanes x 16 B = 4 x 256 B), a pass 32 rows
  const int srow = wave * 4 + (lane >> 4);                // + 32 * pass
  const int lc = (lane & 15) ^ s_of2<RB>(srow);           // logical chunk this lane fetches (same in both passes)
  const long ldb_e = TAPS9 ? p.C : p.ldb;
  unsigned a_voff[2][2], b_voff[2][2];                    // [half][pass]
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int mcol = (lc >> 3) * 128 + h * 64 + (lc & 7) * 8;
    const int ncol = (lc >> 2) * 64 + h * 32 + (lc & 3) * 8;
    const bool a_ok = (m0 + mcol) < p.lda;
    const bool b_ok = TAPS9 ? true : (n0 + ncol) < p.Ntot;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      a_voff[h][j] = a_ok ? (unsigned)(((long)(srow + 32 * j) * p.lda + mcol) * 2) : DTP_OOB_OFF;
      b_voff[h][j] = b_ok ? (unsigned)(((long)(srow + 32 * j) * ldb_e + ncol) * 2) : DTP_OOB_OFF;
    }
  }
  const unsigned b_lanebit = 1u << (lane >> 4);
  int th = 0, tw = 0;                                     // pixel (h, w) of K-tile row `lane` of the next B0 to issue
  if (TAPS9) {
    const int rem = (kbeg * BK + lane) % (p.H * p.W);
    th = rem / p.W;
    tw = rem - th * p.W;
  }
  const int step_dh = TAPS9 ? BK / p.W : 0;
  const int step_dw = TAPS9 ? BK - step_dh * p.W : 0;
  const long shift = TAPS9 ? (long)dy * p.W + dx : 0;
  const char* a_base0 = p.dY + ((long)kbeg * BK * p.lda + m0) * 2;
  const long a_left0 = ((long)p.K - (long)kbeg * BK) * p.lda * 2 - (long)m0 * 2;
  const char* b_base0 = p.X + (((long)kbeg * BK + shift) * ldb_e + c0) * 2;
  const long b_left0 = ((long)p.K - (long)kbeg * BK - shift) * ldb_e * 2 - (long)c0 * 2;
  const long a_step = (long)BK * p.lda * 2, b_step = (long)BK * ldb_e * 2;
  unsigned long long bmask = ~0ull;                       // border mask of the tile whose B halves are being issued

  auto issue_A = [&](auto h_, auto d_, int Tt) {          // half-tile A_h of K-tile Tt -> slot [d][A_h]
    constexpr int h = decltype(h_)::value, d = decltype(d_)::value;
    const long adv = (long)(Tt - kbeg) * a_step;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a_base0 + adv, Tt < kend ? a_left0 - adv : 0);
    char* dst = smem + d * DB + h * SLOT + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16_buf(rs, a_voff[h][j], dst + j * 8192);
  };
  auto issue_B = [&](auto h_, auto d_, int Tt) {
    constexpr int h = decltype(h_)::value, d = decltype(d_)::value;
    const long adv = (long)(Tt - kbeg) * b_step;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(b_base0 + adv, Tt < kend ? b_left0 - adv : 0);
    char* dst = smem + d * DB + (2 + h) * SLOT + wave * 1024;
    if (TAPS9) {
      if (h == 0) {                                       // B0 opens a tile: its border mask, then (h, w) moves on
        const bool valid = (unsigned)(th + dy) < (unsigned)p.H && (unsigned)(tw + dx) < (unsigned)p.W;
        bmask = __ballot(valid);
        tw += step_dw; th += step_dh;
        const bool cw = tw >= p.W;
        tw -= cw ? p.W : 0; th += cw ? 1 : 0;
        for (int r = 0; r < p.nwrap; ++r) th -= th >= p.H ? p.H : 0;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned bits = (unsigned)(bmask >> (32 * j + wave * 4));      // scalar shift: wave is uniform
        glds16_buf(rs, (bits & b_lanebit) ? b_voff[h][j] : DTP_OOB_OFF, dst + j * 8192);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) glds16_buf(rs, b_voff[h][j], dst + j * 8192);
    }
  };

  // ---------------- transposed fragment read addresses (slot A0 / B0 of dbuf 0; halves and K-substeps are immediates)
  const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int krow = 8 * g4 + q;
  const int sw = s_of2<RB>(krow);
  const int sub = pp >> 1, half8 = (pp & 1) * 8;
  const unsigned lds0 = lds_addr(smem);
  unsigned a_ad[4], b_ad[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_ad[i] = lds0 + krow * RB + ((((wm * 8 + 2 * i) ^ sw) + sub) << 4) + half8;
#pragma unroll
  for (int j = 0; j < 2; ++j) b_ad[j] = lds0 + 2 * SLOT + krow * RB + ((((wn * 4 + 2 * j) ^ sw) + sub) << 4) + half8;

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  u32x2_t xl[2][4], xh[2][4], b0l[2][2], b0h[2][2], b1l[2][2], b1h[2][2];

#define W8_READ_A(h)                                                                           \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) \
      DTP_TR2(xl[ks][i], xh[ks][i], a_ad[i], (h) * SLOT + ks * 32 * RB, RB);
#define W8_READ_B(h, bl, bh)                                                                   \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      DTP_TR2(bl[ks][j], bh[ks][j], b_ad[j], (h) * SLOT + ks * 32 * RB, RB);
#define W8_WAIT_A()                                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) \
      DTP_LGKM_WAIT2(0, xl[ks][i], xh[ks][i]);
#define W8_WAIT_B(bl, bh)                                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
      DTP_LGKM_WAIT2(0, bl[ks][j], bh[ks][j]);
#define W8_MMA(mh, nh, bl, bh)                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
    const bf16x8_t xa_ = frag_of(xl[ks][i], xh[ks][i]);                                        \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                              \
        acc[(mh) * 4 + i][(nh) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(           \
            frag_of(bl[ks][j], bh[ks][j]), xa_, acc[(mh) * 4 + i][(nh) * 2 + j], 0, 0, 0);     \
  }
#define W8_SYNC_IN()                                          \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");  \
  __builtin_amdgcn_s_barrier();                               \
  __builtin_amdgcn_sched_barrier(0);
#define W8_GO()                                               \
  __builtin_amdgcn_sched_barrier(0);                          \
  __builtin_amdgcn_s_setprio(1);
#define W8_SYNC_OUT()                                         \
  __builtin_amdgcn_s_setprio(0);                              \
  __builtin_amdgcn_sched_barrier(0);                          \
  __builtin_amdgcn_s_barrier();                               \
  __builtin_amdgcn_sched_barrier(0);
#define W8_FLIP(delta)                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) a_ad[i] += (delta); \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) b_ad[j] += (delta);
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // prologue: tile kbeg entirely, tile kbeg+1's B0 / A0 (issue order = consumption order)
  issue_B(I0{}, I0{}, kbeg);
  issue_A(I0{}, I0{}, kbeg);
  issue_B(I1{}, I0{}, kbeg);
  issue_A(I1{}, I0{}, kbeg);
  issue_B(I0{}, I1{}, kbeg + 1);
  issue_A(I0{}, I1{}, kbeg + 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMC) : "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();    // stagger the second wave row (the SIMD-mates) by one barrier
  __builtin_amdgcn_sched_barrier(0);

  for (int t = kbeg; t < kend; t += 2) {
    // ---- K-tile t (dbuf 0)
    W8_READ_A(0) W8_READ_B(0, b0l, b0h)                                        // P1
    issue_B(I1{}, I1{}, t + 1);
    W8_SYNC_IN() W8_WAIT_A() W8_WAIT_B(b0l, b0h) W8_GO() W8_MMA(0, 0, b0l, b0h) W8_SYNC_OUT()
    W8_READ_B(1, b1l, b1h)                                                     // P2
    issue_A(I1{}, I1{}, t + 1);
    W8_SYNC_IN() W8_WAIT_B(b1l, b1h) W8_GO() W8_MMA(0, 1, b1l, b1h) W8_SYNC_OUT()
    W8_READ_A(1)                                                               // P3 + P4: nothing new to read for A1B0
    issue_B(I0{}, I0{}, t + 2);
    issue_A(I0{}, I0{}, t + 2);
    W8_SYNC_IN() W8_WAIT_A() W8_GO() W8_MMA(1, 1, b1l, b1h) W8_MMA(1, 0, b0l, b0h) W8_SYNC_OUT()
    W8_FLIP(DB)
    // ---- K-tile t+1 (dbuf 1); with an odd tile count an all-zero tile (loads past kend return zeros)
    W8_READ_A(0) W8_READ_B(0, b0l, b0h)                                        // P5
    issue_B(I1{}, I0{}, t + 2);
    W8_SYNC_IN() W8_WAIT_A() W8_WAIT_B(b0l, b0h) W8_GO() W8_MMA(0, 0, b0l, b0h) W8_SYNC_OUT()
    W8_READ_B(1, b1l, b1h)                                                     // P6
    issue_A(I1{}, I0{}, t + 2);
    W8_SYNC_IN() W8_WAIT_B(b1l, b1h) W8_GO() W8_MMA(0, 1, b1l, b1h) W8_SYNC_OUT()
    W8_READ_A(1)                                                               // P7 + P8
    issue_B(I0{}, I1{}, t + 3);
    issue_A(I0{}, I1{}, t + 3);
    W8_SYNC_IN() W8_WAIT_A() W8_GO() W8_MMA(1, 1, b1l, b1h) W8_MMA(1, 0, b0l, b0h) W8_SYNC_OUT()
    W8_FLIP(-DB)
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();    // pair the stagger barrier of the other group
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef W8_READ_A
#undef W8_READ_B
#undef W8_WAIT_A
#undef W8_WAIT_B
#undef W8_MMA
#undef W8_SYNC_IN
#undef W8_GO
#undef W8_SYNC_OUT
#undef W8_FLIP

  // ---------------- epilogue (as wgrad_tt2): lane owns dW[m = .. + (lane & 15)][n .. n + 3]
  const bool direct = nsplit_g == 1;
  float* slab = p.partial + (size_t)zsplit * p.M * p.Ntot;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + g4 * 4;
      if (n >= p.Ntot) continue;
      const size_t o = (size_t)m * p.Ntot + n;
      f32x4_t v = acc[i][j];
      if (!direct) {
        *(f32x4_t*)(slab + o) = v;
      } else if (p.out_fp32) {
        float* d = (float*)p.out + o;
        if (p.beta != 0.f) {
          const f32x4_t old = *(const f32x4_t*)d;
          v = v + old * p.beta;
        }
        *(f32x4_t*)d = v;
      } else {
        bf16_t* d = (bf16_t*)p.out + o;
        if (p.beta != 0.f) {
          const u32x2_t old = *(const u32x2_t*)d;
          v[0] += p.beta * bf16_to_f32((bf16_t)(old[0] & 0xFFFFu));
          v[1] += p.beta * bf16_to_f32((bf16_t)(old[0] >> 16));
          v[2] += p.beta * bf16_to_f32((bf16_t)(old[1] & 0xFFFFu));
          v[3] += p.beta * bf16_to_f32((bf16_t)(old[1] >> 16));
        }
        u32x2_t pk;
        pk[0] = pack_bf16x2(v[0], v[1]);
        pk[1] = pack_bf16x2(v[2], v[3]);
        *(u32x2_t*)d = pk;
      }
    }
  }
}

template <bool TAPS9>
static int launch_wg8(const Wgrad2Params& p0, int nsplit, hipStream_t stream) {
  Wgrad2Params p = p0;
  p.tiles_m = (p.M + 255) / 256;
  if (TAPS9) {
    if (p.C % 256 != 0) return -20;
    p.tiles_n = 9 * (p.C / 256);
  } else {
    p.tiles_n = (p.Ntot + 255) / 256;
  }
  constexpr int lds = 2 * 4 * 64 * 256;
  static bool attr_set = false;
  auto kern = wgrad_tt8_kernel<TAPS9>;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return -2;
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n * nsplit, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, p);
  return DTP_CHECK_LAUNCH();
}

template <int BM, int BN, int WGM, int WGN, bool TAPS9>
static int launch_wg2(const Wgrad2Params& p0, int nsplit, hipStream_t stream) {
  Wgrad2Params p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  if (TAPS9) {
    if (p.C % BN != 0) return -20;
    p.tiles_n = 9 * (p.C / BN);
  } else {
    p.tiles_n = (p.Ntot + BN - 1) / BN;
  }
  constexpr int lds = 2 * 64 * (BM + BN) * 2;
  static bool attr_set = false;
  auto kern = wgrad_tt2_kernel<BM, BN, WGM, WGN, TAPS9>;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return -2;
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n * nsplit, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, stream, p);
  return DTP_CHECK_LAUNCH();
}

// Launches the split-K partial pass only (cfg 0: 256x256 tile, 8 waves; cfg 1: 128x128 tile, 4 waves);
// the caller (dtp_wgrad_tt in wgrad.hip) validates shapes, fixes nsplit and runs the slab reduce.
extern "C" int dtp_wgrad_tt2_launch(const void* dY, const void* X, void* out, float* partial, int K, int M, int Ntot,
                                    int C, int taps, int H, int W, int lda, int ldb, int out_fp32, float beta,
                                    int nsplit, int ksteps_total, int ksteps_per_split, int cfg, hipStream_t stream) {
  Wgrad2Params p;
  p.dY = (const char*)dY; p.X = (const char*)X; p.partial = partial; p.out = (char*)out;
  p.K = K; p.M = M; p.Ntot = Ntot; p.C = C; p.H = H; p.W = W; p.lda = lda; p.ldb = ldb;
  p.out_fp32 = out_fp32; p.beta = beta;
  p.ksteps_total = ksteps_total; p.ksteps_per_split = ksteps_per_split;
  p.nwrap = taps == 9 ? (64 / W + 1 + H - 1) / H : 0;
  p.stamps = nullptr;
  if (const char* e = std::getenv("DTP_WG2_STAMPS")) p.stamps = (unsigned long long*)std::strtoull(e, nullptr, 0);
  // constant per-lane offsets are 32-bit: one K-step of rows must stay below the out-of-range marker
  if ((long)64 * lda * 2 >= 0x40000000l || (long)64 * (taps == 9 ? C : ldb) * 2 >= 0x40000000l) return -21;
  const bool t9 = taps == 9;
  switch (cfg) {
    case 0: return t9 ? launch_wg2<256, 256, 2, 4, true>(p, nsplit, stream)
                      : launch_wg2<256, 256, 2, 4, false>(p, nsplit, stream);
    case 1: return t9 ? launch_wg2<128, 128, 2, 2, true>(p, nsplit, stream)
                      : launch_wg2<128, 128, 2, 2, false>(p, nsplit, stream);
    case 2: return t9 ? launch_wg8<true>(p, nsplit, stream) : launch_wg8<false>(p, nsplit, stream);   // 8-phase 256x256
    default: return -16;
  }
}
