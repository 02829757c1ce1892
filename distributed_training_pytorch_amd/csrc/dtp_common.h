// Common device helpers for the gfx950 (CDNA4, wave64) kernels of this framework.
// No CUDA-compat layer: this header is written for hipcc --offload-arch=gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                        // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;     // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;     // 8 bytes
typedef __attribute__((ext_vector_type(4))) float f32x4_t;      // 16x16 accumulator tile slice
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define DTP_WAVE 64
#define DTP_DEVINL __device__ __forceinline__

DTP_DEVINL float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// Round-to-nearest-even, NaN-preserving (hipcc lowers the cast to v_cvt_pk_bf16_f32 on gfx950).
DTP_DEVINL bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// Two floats -> packed bf16 pair with ONE v_cvt_pk_bf16_f32 (round-to-nearest-even, same bits as two
// scalar conversions; the scalar form compiled to two converts plus a v_perm per pair).
DTP_DEVINL uint32_t pack_bf16x2(float lo, float hi) {
  typedef float f32v2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16v2_t __attribute__((ext_vector_type(2)));
  const f32v2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16v2_t));
}

// (x > 0) of the two bf16 halves of a word as bits 0 (low half) and 1 (high half).  Plain scalar
// arithmetic on purpose: a packed-int16 clamp over `v[e]` in an unrolled loop was folded by hipcc
// into four copies of word 0's bits (caught by test_relu_bitmask_roundtrip).
DTP_DEVINL uint32_t bf16x2_pos_bits(uint32_t w) {
  const uint32_t lo = ((w & 0xFFFFu) - 1u) < 0x7FFFu ? 1u : 0u;      // 0x0001 .. 0x7FFF
  const uint32_t hi = (int32_t)w >= 0x10000 ? 2u : 0u;               // sign clear, high half non-zero
  return lo | hi;
}

// Async global->LDS copy of 16 bytes per lane. LDS destination is wave-uniform base + lane*16.
DTP_DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Barrier that PUBLISHES LDS-DMA data to the other waves of the workgroup.  A global_load_lds is
// retired only on the issuing wave's vmcnt; hipcc's __syncthreads() emitted the s_waitcnt vmcnt(0)
// in front of its s_barrier in some instantiations and not in others (conv_c64<64, POOL> had none
// inside its row loop: one stale input row per ~10^3 workgroups whenever HBM was busy with another
// stream's kernels).  Every barrier that hands DMA'd LDS to other waves goes through here.
DTP_DEVINL void sync_lds_dma() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Transposing LDS reads from inline asm.  hipcc puts `s_waitcnt vmcnt(0)` in front of every
// __builtin_amdgcn_ds_read_tr16_b64 that follows an LDS-DMA (the builtin has no memory operand, so
// the waitcnt pass assumes it may read what the DMA still writes): a prefetch issued before the
// fragment reads was always drained before the first MFMA.  These forms are invisible to that pass;
// the caller counts lgkmcnt itself and waits with DTP_LGKM_WAIT*, which names the registers it
// releases "+v" so that no use (or compiler copy) of them can move above the wait.
#define DTP_TR1(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr))
// one bf16x8 MFMA fragment = two reads (k rows +0..3 and +4..7 of an 8-row group, RB_ bytes per row)
#define DTP_TR2(lo, hi, addr, IMM, RB_)                                                        \
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"    \
               : "=&v"(lo), "=&v"(hi)                                                          \
               : "v"(addr), "n"(IMM), "n"((IMM) + 4 * (RB_)))
#define DTP_LGKM_WAIT2(N, a, b) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N))
#define DTP_LGKM_WAIT4(N, a, b, c, d) \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N))
// 16-byte LDS read the compiler does not track (same contract as DTP_TR1): hipcc waits lgkmcnt(0)
// where a counted wait would do once reads are software-pipelined past an MFMA group
#define DTP_DSR128(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
DTP_DEVINL bf16x8_t frag_of(u32x2_t lo, u32x2_t hi) {
  return __builtin_bit_cast(bf16x8_t, (u32x4_t){lo[0], lo[1], hi[0], hi[1]});
}
DTP_DEVINL unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// Buffer descriptor over [base, base + bytes): loads / stores / LDS-DMA through it are range-checked
// per lane by the hardware — an out-of-range lane loads zeros (LDS-DMA: writes zeros to its LDS slot,
// measured in bench/micro/lds_probe.hip) and its store is dropped.  Tails need no branches.
#define DTP_OOB_OFF 0x80000000u
DTP_DEVINL __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes) {
  const unsigned n = bytes <= 0 ? 0u : (bytes > 0x7FFFFFFFl ? 0x7FFFFFFFu : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)n, 0x00020000);
}
DTP_DEVINL void glds16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0,
                                           0, 0);
}

// Bijective XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each
// XCD a contiguous run of tile ids for L2 reuse (cdna_hip_programming.md T1). Speed only.
DTP_DEVINL int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

DTP_DEVINL float wave_reduce_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
DTP_DEVINL float wave_reduce_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Counter-based RNG for dropout: 64-bit mix of (seed, element index) -> uniform [0,1).
DTP_DEVINL float rng_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + idx * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
}

#define DTP_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -1)
