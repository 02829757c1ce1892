// Weight-gradient GEMM ("TT": both operands are reduction-major) for gfx950 — the SMALL-TILE
// configurations (cfg 2 / 3 / 4: 128x64, 64x128, 64x64) that serve outputs narrower than the
// 128-/256-wide tiles of wgrad2.hip (linear3's [C x 4096] gradient, 64-channel layers of user models),
// the split-K reduce kernel and the dispatcher of both files.  The 256x256 / 128x128 tiles and the
// software-pipelined variants that lived here in rounds 1-2 were superseded by wgrad2.hip (asm
// transposed reads, descriptor staging) and are gone.
//
//   dW[m][tap*C + c] = sum_k  dY[k][m] * X[pix(k) + shift(tap)][c]          (3x3 conv, TAPS9)
//   dW[m][n]         = sum_k  dY[k][m] * X[k][n]                            (linear)
//
// k runs over N*H*W output pixels (conv) or the batch (linear): it is the *slow* axis of both
// operands in memory (NHWC rows), so MFMA fragments (which want 8 consecutive k per lane) are
// produced with the gfx950 transposing LDS read ds_read_b64_tr_b16: tiles are staged row-major
// [k][channels] with global_load_lds and read back column-major — no transposed copies of the
// activations ever exist in HBM.
// * LDS images are XOR-swizzled per k-row (on the global source address, LDS-DMA being
//   lane-linear) so that the 8 rows x 4 column-groups a half-wave touches per transposed read
//   fall on 32 distinct 8-byte bank slots (conflict-free; see s_of()).
// * The huge reduction axis (up to N*50176) is split across workgroups (1-D grid of (z, tile)); partial f32
//   slabs are summed by wgrad_reduce_kernel, which also applies beta (gradient accumulation) and
//   casts to the gradient arena's dtype — deterministic, no float atomics
//   (cdna_hip_programming.md Guideline 12).
// * Image-border taps / K, M, N tails are redirected per lane to a zeroed 16-byte page.
//
// Reference parity: replaces cuDNN bwd-filter and the cuBLAS dW GEMMs
// triggered by loss.backward() at example_trainer.py:86.
#include "dtp_common.h"

struct WgradParams {
  const char* dY;     // [K][lda] bf16
  const char* X;      // [K][ldb] bf16 (conv: NHWC pixels x C)
  float* partial;     // [nsplit][M][Ntot] f32
  char* out;          // direct output (nsplit == 1)
  const char* zero_page;
  int K, M, Ntot;     // Ntot = taps * C (conv) or in_features (linear)
  int C, H, W;
  int lda, ldb;
  int ksteps_total, ksteps_per_split;
  int tiles_m, tiles_n;
  int out_fp32;
  float beta;
};

template <int RB>
DTP_DEVINL int s_of(int k) {   // swizzle (in 16-B chunks) of LDS row k for an image with RB-byte rows
  if (RB >= 256) return ((k & 3) | (((k >> 3) & 1) << 2)) << 1;
  return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1;
}

template <int BM, int BN, int WGM, int WGN, bool TAPS9>
__global__ __launch_bounds__(WGM * WGN * 64) void wgrad_tt_kernel(const WgradParams p) {
  constexpr int T = WGM * WGN * 64;
  constexpr int BK = 64;
  constexpr int RBA = BM * 2, RBB = BN * 2;        // LDS row bytes
  constexpr int LA = BM / 8, LB = BN / 8;          // lanes (16-B chunks) per row
  constexpr int RPA = T / LA, RPB = T / LB;        // rows per staging pass
  constexpr int AP = BK / RPA, BP = BK / RPB;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int STAGE = BK * (RBA + RBB);
  static_assert(AP >= 1 && BP >= 1 && (RPA % 16 == 0) && (RPB % 16 == 0), "staging geometry");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((ext_vector_type(4))) short s4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  const int ntiles = p.tiles_m * p.tiles_n;
  // 1-D grid over (k-split z, tile) with the tile index fastest, remapped so that each XCD owns a
  // contiguous run: all tiles (taps x channel blocks) of one k-chunk then run on the SAME XCD at
  // the same time and share its L2 — the activations are fetched from HBM once, not once per tap
  // (measured before: TCC hit rate 14 %, FETCH_SIZE 8x the operand bytes).
  const int nsplit_g = gridDim.x / ntiles;
  const int pair = xcd_remap(blockIdx.x, gridDim.x);
  const int zsplit = pair / ntiles;
  const int tile = pair - zsplit * ntiles;
  const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;   // tm fastest: share the X tile
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

  // ---------------- staging set-up
  // A (dY): lane -> (row ra + i*RPA, physical chunk pca)
  const int ra = tid / LA, pca = tid % LA;
  const int lca = pca ^ s_of<RBA>(ra);
  const bool a_col_ok = (m0 + lca * 8) < p.lda;
  const char* a_src = p.dY + ((size_t)(kbeg * BK + ra) * p.lda + m0 + lca * 8) * 2;
  // B (X)
  const int rb = tid / LB, pcb = tid % LB;
  const int lcb = pcb ^ s_of<RBB>(rb);
  const bool b_col_ok = TAPS9 ? true : (n0 + lcb * 8) < p.Ntot;
  const char* b_src;
  int bh[BP], bw[BP];
  if (TAPS9) {
    b_src = p.X + ((size_t)((long)(kbeg * BK + rb) + dy * p.W + dx) * p.C + c0 + lcb * 8) * 2;
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      const int k = kbeg * BK + rb + i * RPB;
      const int rem = k % (p.H * p.W);
      bh[i] = rem / p.W;
      bw[i] = rem - bh[i] * p.W;
    }
  } else {
    b_src = p.X + ((size_t)(kbeg * BK + rb) * p.ldb + n0 + lcb * 8) * 2;
  }
  const int step_dh = TAPS9 ? BK / p.W : 0;
  const int step_dw = TAPS9 ? BK - step_dh * p.W : 0;
  const size_t a_step = (size_t)BK * p.lda * 2;
  const size_t b_step = (size_t)BK * (TAPS9 ? p.C : p.ldb) * 2;
  const size_t a_pass = (size_t)RPA * p.lda * 2;
  const size_t b_pass = (size_t)RPB * (TAPS9 ? p.C : p.ldb) * 2;

  auto stage = [&](int kt, int buf) {
    char* lA = smem + buf * STAGE + wave * 1024;
    char* lB = smem + buf * STAGE + BK * RBA + wave * 1024;
    const int krow0 = kt * BK;
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const bool ok = a_col_ok && (krow0 + ra + i * RPA) < p.K;
      const char* g = ok ? a_src + i * a_pass : p.zero_page;
      glds16(g, lA + i * (T * 16));
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) {
      bool ok = b_col_ok && (krow0 + rb + i * RPB) < p.K;
      if (TAPS9) {
        const int hh = bh[i] + dy, ww = bw[i] + dx;
        ok = ok && hh >= 0 && hh < p.H && ww >= 0 && ww < p.W;
        // advance (h, w) of this row to the next K-step
        int w2 = bw[i] + step_dw, h2 = bh[i] + step_dh;
        if (w2 >= p.W) { w2 -= p.W; h2 += 1; }
        while (h2 >= p.H) h2 -= p.H;   // BK/W can exceed H for tiny images
        bw[i] = w2; bh[i] = h2;
      }
      const char* g = ok ? b_src + i * b_pass : p.zero_page;
      glds16(g, lB + i * (T * 16));
    }
    a_src += a_step;
    b_src += b_step;
  };

  // ---------------- transposed fragment read addresses
  const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int krow = 8 * g4 + q;                       // + 32*ks, +4 for the second read
  const int sa = s_of<RBA>(krow), sb = s_of<RBB>(krow);
  const int a_cw = (wm * WM) >> 3, b_cw = (wn * WN) >> 3;
  const int sub = (pp >> 1), half8 = (pp & 1) * 8;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* bA = smem + buf * STAGE;
    const char* bB = bA + BK * RBA;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ch = ((a_cw + 2 * i) ^ sa) + sub;
        const char* a0 = bA + (ks * 32 + krow) * RBA + (ch << 4) + half8;
        s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0));
        s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(a0 + 4 * RBA));
        af[i] = (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int ch = ((b_cw + 2 * j) ^ sb) + sub;
        const char* b0 = bB + (ks * 32 + krow) * RBB + (ch << 4) + half8;
        s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(b0));
        s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(b0 + 4 * RBB));
        bf[j] = (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      // rows (accumulator registers) = input channels, columns (lanes) = output channels
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  if (kbeg < kend) stage(kbeg, 0);
  int buf = 0;
  for (int kt = kbeg; kt < kend; ++kt) {
    sync_lds_dma();      // stage kt landed everywhere; everyone is done reading the other buffer
    if (kt + 1 < kend) stage(kt + 1, buf ^ 1);
    compute(buf);
    buf ^= 1;
  }

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

// out = beta*out + sum_z partial[z]   (vectorised x4; total is a multiple of 4)
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, size_t total,
                                    void* __restrict__ out, int out_fp32, float beta) {
  const size_t nvec = total >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (size_t)gridDim.x * blockDim.x) {
    f32x4_t v = *(const f32x4_t*)(partial + (i << 2));
    for (int z = 1; z < nsplit; ++z) v += *(const f32x4_t*)(partial + (size_t)z * total + (i << 2));
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

template <int BM, int BN, int WGM, int WGN, bool TAPS9>
static int launch_wg(const WgradParams& p0, int nsplit, hipStream_t stream) {
  WgradParams p = p0;
  p.tiles_m = (p.M + BM - 1) / BM;
  if (TAPS9) {
    if (p.C % BN != 0) return -20;
    p.tiles_n = 9 * (p.C / BN);
  } else {
    p.tiles_n = (p.Ntot + BN - 1) / BN;
  }
  constexpr int lds = 2 * 64 * (BM + BN) * 2;
  static bool attr_set = false;
  auto kern = wgrad_tt_kernel<BM, BN, WGM, WGN, TAPS9>;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return -2;
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_n * nsplit, 1, 1);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), lds, stream, p);
  return DTP_CHECK_LAUNCH();
}

// second-generation kernels (csrc/wgrad2.hip): cfg 8 = 256x256 tile, cfg 9 = 128x128 tile, cfg 10 = 256x256 8-phase
extern "C" int dtp_wgrad_tt2_launch(const void* dY, const void* X, void* out, float* partial, int K, int M, int Ntot,
                                    int C, int taps, int H, int W, int lda, int ldb, int out_fp32, float beta,
                                    int nsplit, int ksteps_total, int ksteps_per_split, int cfg, hipStream_t stream);

// taps == 9: X is NHWC [K = N*H*W][C], dW is [M][9*C].  taps == 1: X is [K][ldb], dW is [M][Ntot].
extern "C" int dtp_wgrad_tt(const void* dY, const void* X, void* out, float* partial, const void* zero_page,
                            int K, int M, int Ntot, int C, int taps, int H, int W, int lda, int ldb,
                            int out_fp32, float beta, int nsplit, int cfg, hipStream_t stream) {
  if (K <= 0 || M <= 0 || Ntot <= 0) return -10;
  if (taps != 1 && taps != 9) return -11;
  if ((lda & 7) || (ldb & 7) || (Ntot & 7)) return -14;
  if (taps == 9) {
    if (H < 1 || W < 1 || K % (H * W) != 0 || (C & 63) || Ntot != 9 * C || ldb != C) return -12;
  }
  WgradParams p;
  p.dY = (const char*)dY; p.X = (const char*)X; p.partial = partial; p.out = (char*)out;
  p.zero_page = (const char*)zero_page;
  p.K = K; p.M = M; p.Ntot = Ntot; p.C = C; p.H = H; p.W = W; p.lda = lda; p.ldb = ldb;
  p.out_fp32 = out_fp32; p.beta = beta;
  p.ksteps_total = (K + 63) / 64;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > p.ksteps_total) nsplit = p.ksteps_total;
  p.ksteps_per_split = (p.ksteps_total + nsplit - 1) / nsplit;
  nsplit = (p.ksteps_total + p.ksteps_per_split - 1) / p.ksteps_per_split;
  if (nsplit > 1 && partial == nullptr) return -15;
  int rc;
  const bool t9 = taps == 9;
#define DTP_WG(BM_, BN_, WGM_, WGN_) \
  (t9 ? launch_wg<BM_, BN_, WGM_, WGN_, true>(p, nsplit, stream) : launch_wg<BM_, BN_, WGM_, WGN_, false>(p, nsplit, stream))
  switch (cfg) {
    case 2: rc = DTP_WG(128, 64, 2, 2); break;
    case 3: rc = DTP_WG(64, 128, 2, 2); break;
    case 4: rc = DTP_WG(64, 64, 2, 2); break;
    case 8: case 9: case 10:      // 10 = the 256x256 tile on the 8-phase schedule (wgrad_tt8_kernel)
      if (t9 && (C % (cfg == 9 ? 128 : 256))) return -20;
      rc = dtp_wgrad_tt2_launch(dY, X, out, partial, K, M, Ntot, C, taps, H, W, lda, ldb, out_fp32, beta, nsplit,
                                p.ksteps_total, p.ksteps_per_split, cfg - 8, stream);
      break;
    default: return -16;
  }
#undef DTP_WG
  if (rc != 0) return rc;
  if (nsplit > 1) {
    const size_t total = (size_t)M * Ntot;
    int blocks = (int)((total / 4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, partial, nsplit, total, out,
                       out_fp32, beta);
    return DTP_CHECK_LAUNCH();
  }
  return 0;
}
