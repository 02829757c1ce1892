// LDS read throughput on gfx950: ds_read_b128 vs ds_read_b64 vs ds_read_b64_tr_b16, 8 waves per CU,
// conflict-free addresses.  Prints bytes per clock per CU (s_memtime runs at a fixed rate, so the
// result is scaled with the measured kernel time instead).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) short s4;

template <int MODE>
__global__ __launch_bounds__(512) void lds_read_kernel(unsigned* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 4; i += 512) ((unsigned*)smem)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  const char* base = smem + wave * 8192;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {          // ds_read_b128: lane-linear 16 B
        u32x4 v = *(const u32x4*)(base + ((lane * 16 + u * 1024 + it * 1024) & 8191));
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
      } else if (MODE == 1) {   // ds_read_b64: lane-linear 8 B
        u32x2 v = *(const u32x2*)(base + ((lane * 8 + u * 512 + it * 512) & 8191));
        acc += v[0] ^ v[1];
      } else {                  // ds_read_b64_tr_b16: 16 rows of 128 B, 4 lanes per row group (as in wgrad.hip)
        // wgrad.hip's conflict-free addressing: k-row = 8*(lane>>4) + ((lane&15)>>2), 512-byte rows,
        // 16-byte chunk XOR-swizzled per row, (lane&3) picks chunk +0/+1 and the 8-byte half
        const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
        const int krow = 8 * g4 + q;
        const int sw = ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1;
        const int ch = (((u * 2 + it * 2) & 31) ^ sw) + (pp >> 1);
        s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(smem + krow * 512 + (ch << 4) + (pp & 1) * 8 + (wave & 3) * 16384));
        acc += (unsigned)v[0] ^ (unsigned)v[1] ^ (unsigned)v[2] ^ (unsigned)v[3];
      }
    }
  }
  if (acc == 0x12345678u) out[tid] = acc;   // keep the loads alive
}

template <int MODE>
static void run(const char* name, int bytes_per_lane) {
  unsigned* d;
  hipMalloc(&d, 4096);
  const int iters = 20000, grid = 256;
  hipFuncSetAttribute((const void*)lds_read_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(lds_read_kernel<MODE>, dim3(grid), dim3(512), 65536, 0, d, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(lds_read_kernel<MODE>, dim3(grid), dim3(512), 65536, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per_cu = (double)iters * 8 * 512 * bytes_per_lane;       // one workgroup per CU
  printf("%-22s %8.3f ms  %7.1f GB/s per CU  (= %5.1f B/clk at 2.4 GHz)\n", name, ms, bytes_per_cu / ms / 1e6,
         bytes_per_cu / (ms * 1e-3) / 2.4e9);
  hipFree(d);
}

int main() {
  run<0>("ds_read_b128", 16);
  run<1>("ds_read_b64", 8);
  run<2>("ds_read_b64_tr_b16", 8);
  return 0;
}
