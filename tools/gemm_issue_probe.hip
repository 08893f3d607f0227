// Round 4: what does the side work of the bf16 GEMM's K step cost when it sits in the slots behind the MFMAs (one wave per SIMD,
// 64 accumulator tiles in AGPRs, v_mfma_f32_16x16x32_bf16 from inline assembly exactly as csrc/gemm_bf16.hip issues it)?
// One workgroup per CU, 4 waves, a "step" = 128 MFMAs; variants add, per step: 16 x (s_add_i32 m0 + buffer_load ... lds through an
// EMPTY descriptor: zero-fill, no memory traffic), 32 ds_read_b128, the step's s_waitcnt + s_barrier.  Cycles per MFMA from s_memtime.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/gemm_issue_probe tools/gemm_issue_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int STEPS = 300;

// MODE bits: 1 = DMA issue (m0 + load, 2 chunks per row in rows 0..7), 2 = fragment reads (2 per row in rows 0..7, 1 in rows 8..14, 9 in row 15),
// 4 = per-step s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier in front of row 15, 8 = the DMA load WITHOUT the m0 write, 16 = the m0 write without the load
template <int MODE>
__global__ __launch_bounds__(256, 1)
void probe(unsigned long long* cycles, float* sink, u4v desc_in, const unsigned char* slab) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 32768; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3f803f80u;
  __syncthreads();
  bf16x8 bf[2][8], afc, afn;
  for (int i = 0; i < 8; ++i) { afc[i] = (__bf16)(1.0f + 0.01f * ((lane + i) & 7)); }
  afn = afc;
  for (int h = 0; h < 2; ++h) for (int b = 0; b < 8; ++b) bf[h][b] = afc;
  f32x4 acc[8][8];
  for (int a = 0; a < 8; ++a) for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned fa = lds0 + lane * 16, fb = lds0 + 32768 + lane * 16;
  // MODE & 32: a REAL descriptor over this workgroup's own 64 KiB slab (L2 resident after the first step): the DMA writes land in LDS
  u4v desc = desc_in;
  if (MODE & 32) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(slab) + (unsigned long long)blockIdx.x * 65536ull;
    desc = u4v{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu)), 65536u, 0x00020000u};
  }
  const unsigned voff = lane * 16 + wave * 1024;
  unsigned dst = lds0 + 65536 + wave * 1024;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < STEPS; ++it) {
    asm volatile("" : "+v"(fa), "+v"(fb));
#pragma unroll
    for (int R = 0; R < 16; ++R) {
      const int kh = R >> 3, a = R & 7;
      if (R == 15 && (MODE & 4)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[a][b]) : "v"(afc), "v"(bf[kh][b]));
        if (MODE & 2) {
          if (R < 15) {
            if (b == 0) afn = *(const __attribute__((address_space(3))) bf16x8*)(uintptr_t)(fa + ((R + 1) & 7) * 2048);
            if (kh == 0 && b == 1) bf[1][a] = *(const __attribute__((address_space(3))) bf16x8*)(uintptr_t)(fb + a * 2048);
          } else {
            bf[0][b] = *(const __attribute__((address_space(3))) bf16x8*)(uintptr_t)(fb + 16384 + b * 2048);
            if (b == 7) afn = *(const __attribute__((address_space(3))) bf16x8*)(uintptr_t)(fa + 16384);
          }
        }
        if ((MODE & (1 | 8 | 16)) && kh == 0) {
          if (b == 2 && !(MODE & 8)) asm volatile("s_add_i32 m0, %0, %1" :: "s"(dst), "s"(a * 4096) : "scc", "memory");
          if (b == 3 && !(MODE & 16)) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(desc), "s"(a * 4096) : "memory");
          if (b == 4 && !(MODE & 8)) asm volatile("s_add_i32 m0, %0, %1" :: "s"(dst), "s"(32768 + a * 4096) : "scc", "memory");
          if (b == 5 && !(MODE & 16)) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(desc), "s"(32768 + a * 4096) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      afc = afn;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  float r = 0.f;
  for (int a = 0; a < 8; ++a) for (int b = 0; b < 8; ++b) r += acc[a][b][0];
  if (r == 1234.5f) sink[0] = r;
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  unsigned long long* d_cyc;
  float* d_sink;
  CK(hipMalloc(&d_cyc, cus * sizeof(unsigned long long)));
  CK(hipMalloc(&d_sink, 64));
  unsigned char* d_slab;
  CK(hipMalloc(&d_slab, (size_t)cus * 65536));
  CK(hipMemset(d_slab, 0x3f, (size_t)cus * 65536));
  unsigned long long* h = (unsigned long long*)malloc(cus * sizeof(unsigned long long));
  const u4v desc = {0u, 0u, 0u, 0x00020000u};          // num_records 0: every lane out of range, zero-fill
  auto report = [&](const char* name, double ms) {
    CK(hipMemcpy(h, d_cyc, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < cus; ++i) m += (double)h[i];
    printf("%-72s %7.2f memtime ticks per MFMA   %8.3f us per 128-MFMA step\n", name, m / cus / ((double)STEPS * 128), ms * 1e3 / STEPS);
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
#define RUN(MODE, NAME) do { hipLaunchKernelGGL((probe<MODE>), dim3(cus), dim3(256), 0, 0, d_cyc, d_sink, desc, d_slab); CK(hipDeviceSynchronize()); \
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((probe<MODE>), dim3(cus), dim3(256), 0, 0, d_cyc, d_sink, desc, d_slab); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); report(NAME, ms); } while (0)
  RUN(0, "MFMAs only (64 AGPR tiles, inline asm)");
  RUN(16, "+ 16 x s_add_i32 m0 per step");
  RUN(8, "+ 16 x buffer_load ... lds (empty descriptor) per step, no m0 write");
  RUN(1, "+ 16 x (m0 + buffer_load ... lds)");
  RUN(2, "+ 32 ds_read_b128 per step");
  RUN(4, "+ s_waitcnt + s_barrier per step");
  RUN(3, "+ DMA + fragment reads");
  RUN(7, "+ DMA + fragment reads + barrier (the GEMM's K step)");
  RUN(33, "+ 16 x (m0 + buffer_load ... lds) from a REAL 64 KiB slab in the L2: 64 KiB of LDS writes per step");
  RUN(33 | 4, "+ real DMA + barrier");
  RUN(33 | 2, "+ real DMA + fragment reads");
  RUN(33 | 6, "+ real DMA + fragment reads + barrier (the GEMM's K step with its LDS traffic)");
  return 0;
}
