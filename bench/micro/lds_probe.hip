// Two hardware probes for the weight-gradient kernel design (gfx950):
//  (1) LDS read issue rate with NO consumer VALU: ds_read_b128 / ds_read_b64 / ds_read_b64_tr_b16
//      issued back to back from inline asm (16 per loop iteration, one lgkmcnt(0) per iteration),
//      timed with s_memtime inside the kernel, 1 and 2 waves per SIMD.  The round-1 micro-benchmark
//      (lds_read_rate.hip) spent ~13 VALU instructions per read on address arithmetic and on folding
//      the four 16-bit results, so it measured the VALU, not the LDS.
//  (2) buffer_load_dwordx4 ... lds with an out-of-range voffset: does the LDS-DMA write zeros to its
//      LDS slot (what a register buffer load returns) or leave the slot untouched?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define RD16(INSTR, VT)                                                                     \
  {                                                                                         \
    VT r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;                \
    asm volatile(INSTR " %0, %16\n\t" INSTR " %1, %16 offset:1024\n\t"                      \
                 INSTR " %2, %16 offset:2048\n\t" INSTR " %3, %16 offset:3072\n\t"          \
                 INSTR " %4, %17\n\t" INSTR " %5, %17 offset:1024\n\t"                      \
                 INSTR " %6, %17 offset:2048\n\t" INSTR " %7, %17 offset:3072\n\t"          \
                 INSTR " %8, %16 offset:4096\n\t" INSTR " %9, %16 offset:5120\n\t"          \
                 INSTR " %10, %16 offset:6144\n\t" INSTR " %11, %16 offset:7168\n\t"        \
                 INSTR " %12, %17 offset:4096\n\t" INSTR " %13, %17 offset:5120\n\t"        \
                 INSTR " %14, %17 offset:6144\n\t" INSTR " %15, %17 offset:7168\n\t"        \
                 "s_waitcnt lgkmcnt(0)"                                                     \
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), \
                   "=&v"(r8), "=&v"(r9), "=&v"(r10), "=&v"(r11), "=&v"(r12), "=&v"(r13), "=&v"(r14),       \
                   "=&v"(r15)                                                               \
                 : "v"(a0), "v"(a1)                                                         \
                 : "memory");                                                               \
    keep ^= r0[0] ^ r15[0];                                                                 \
  }

// MODE 0: b128 lane-linear; 1: b64 lane-linear; 2: tr_b16 on wgrad.hip's 512-byte-row swizzle;
// 3: tr_b16 on 256-byte rows with the guide's image (b): ch ^ (((row&3)<<2) | ((row>>2)&3))
template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += blockDim.x) ((unsigned*)smem)[i] = i * 2654435761u;
  __syncthreads();
  unsigned a0, a1;
  if (MODE == 0) {
    a0 = lane * 16; a1 = a0 + 8192;
  } else if (MODE == 1) {
    a0 = lane * 8; a1 = a0 + 8192;
  } else if (MODE == 2) {
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int krow = 8 * g4 + q;
    const int sw = ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1;
    const int ch = (0 ^ sw) + (pp >> 1);
    a0 = krow * 512 + (ch << 4) + (pp & 1) * 8;
    a1 = a0 + 4 * 512;
  } else {
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int row = 8 * g4 + q;                        // block rows r0 + q, two blocks per half 8 rows apart
    const int ch = (0 + (pp >> 1)) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    a0 = row * 256 + (ch << 4) + (pp & 1) * 8;
    a1 = a0 + 4 * 256;
  }
  a0 += (tid >> 6 & 3) * 16;     // a different 16-byte column per wave pair (same banks pattern, shifted)
  a1 += (tid >> 6 & 3) * 16;
  unsigned keep = 0;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) RD16("ds_read_b128", u32x4)
    else if (MODE == 1) RD16("ds_read_b64", u32x2)
    else RD16("ds_read_b64_tr_b16", u32x2)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
  if (keep == 0x12345u) out[4096] = keep;
}

template <int MODE>
static void run_rate(const char* name, int threads, int bytes_per_lane) {
  unsigned long long* d;
  hipMalloc(&d, 8 * 8192);
  hipMemset(d, 0, 8 * 8192);
  const int iters = 2000, grid = 256;
  hipFuncSetAttribute((const void*)rate_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(threads), 65536, 0, d, 50);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(rate_kernel<MODE>, dim3(grid), dim3(threads), 65536, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(2048);
  hipMemcpy(h.data(), d, 2048 * 8, hipMemcpyDeviceToHost);
  double sum = 0; int n = 0;
  const int waves = threads / 64;
  for (int b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) { sum += (double)h[b * 8 + w]; ++n; }
  const double ticks = sum / n;                       // s_memtime ticks (100 MHz constant clock on gfx950? printed raw)
  const double reads = (double)iters * 16;
  printf("%-34s waves/CU %d: %9.0f memtime ticks, %6.3f ticks per wave-read, %7.1f B per tick per CU\n", name, waves,
         ticks, ticks / reads, reads * 64 * bytes_per_lane * waves / ticks);
  hipFree(d);
}

__global__ void oob_kernel(const char* src, unsigned bytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // even lanes in range, odd lanes far out of range
  const unsigned voff = (tid & 1) ? 0x7FFFFFF0u : (unsigned)tid * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + (tid >> 6) * 1024), 16,
                                           voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < 1024 / 4; i += blockDim.x) out[i] = ((unsigned*)smem)[i];
}

int main() {
  run_rate<0>("ds_read_b128 lane-linear", 256, 16);
  run_rate<0>("ds_read_b128 lane-linear", 512, 16);
  run_rate<1>("ds_read_b64 lane-linear", 256, 8);
  run_rate<1>("ds_read_b64 lane-linear", 512, 8);
  run_rate<2>("tr_b16, 512-B rows (wgrad.hip)", 256, 8);
  run_rate<2>("tr_b16, 512-B rows (wgrad.hip)", 512, 8);
  run_rate<3>("tr_b16, 256-B rows image (b)", 256, 8);
  run_rate<3>("tr_b16, 256-B rows image (b)", 512, 8);

  // OOB probe
  char* src; unsigned* out;
  hipMalloc(&src, 4096); hipMalloc(&out, 1024);
  std::vector<unsigned> hs(1024);
  for (int i = 0; i < 1024; ++i) hs[i] = 0x11110000u + i;
  hipMemcpy(src, hs.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(oob_kernel, dim3(1), dim3(64), 4096, 0, src, 4096u, out);
  hipDeviceSynchronize();
  std::vector<unsigned> ho(256);
  hipMemcpy(ho.data(), out, 1024, hipMemcpyDeviceToHost);
  int zeros = 0, untouched = 0, good = 0, other = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const unsigned v = ho[l * 4 + j];
      if (l & 1) { if (v == 0) ++zeros; else if (v == 0xDEADBEEFu) ++untouched; else ++other; }
      else { if (v == 0x11110000u + l * 4 + j) ++good; else ++other; }
    }
  printf("OOB LDS-DMA probe: in-range dwords ok %d/128; out-of-range dwords: zero %d, untouched %d, other %d\n", good,
         zeros, untouched, other);
  return 0;
}
