// What does ONE wave per SIMD pay for side instructions behind its MFMAs when they are INDEPENDENT of each other and spread evenly?
// (round 6; follows tools/occupancy_probe.hip, whose side work was one dependent v_fma chain: 1 / 2 / 3 per MFMA = 17.5 / 24.7 / 39.6
// cycles).  The dy epilogue of conv_nhwc.hip was cut into micro-ops of up to four DEPENDENT instructions behind every second MFMA;
// this probe prices the alternatives: K independent VALU instructions per MFMA (four rotating chains), packed-fp32 against scalar,
// a transcendental every MFMA / every second MFMA, an accvgpr read, and the same with an LDS fragment read + s_waitcnt per 4 MFMAs.
// Reported: cycles per 16x16x32 MFMA per SIMD (16 = the pipe rate).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/epilogue_slot_probe tools/epilogue_slot_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 1500, UNROLL = 8;

// MODE 0: K independent v_fma_f32 (chain c = slot % 4: distance 4 between dependent ones)
// MODE 1: K independent v_pk_fma_f32
// MODE 2: one v_exp_f32 per MFMA (K = 1) or per second MFMA (K = 0) + one independent v_fma
// MODE 3: K dependent v_fma (the old probe's chain, no transcendental)
// MODE 4: K independent v_fma + one ds_read_b128 and one s_waitcnt lgkmcnt per 4 MFMAs (the conv's fragment read rate)
// MODE 5: K independent v_fma + one v_accvgpr_read per MFMA
template <int MODE, int K>
__global__ __launch_bounds__(256)
void probe(unsigned long long* cycles, float* sink) {
  __shared__ u32x4 lds[1024];
  const int lane = threadIdx.x & 63;
  lds[threadIdx.x] = u32x4{1u, 2u, 3u, 4u};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + 0.01f * ((lane + i) & 7)); b[i] = (__bf16)(0.5f + 0.01f * ((lane * 3 + i) & 7)); }
  f32x4 acc[UNROLL];
  for (int u = 0; u < UNROLL; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  float c[4] = {0.001f * lane, 0.5f, 1.0f, 0.25f};
  f32x2 p[4] = {{0.1f, 0.2f}, {0.3f, 0.4f}, {0.5f, 0.6f}, {0.7f, 0.8f}};
  const float m = 0.999f;
  const f32x2 m2 = {0.999f, 0.998f};
  float e = 0.01f * lane, ex = 0.f, rd = 0.f;
  u32x4 fr = {0, 0, 0, 0};
  const unsigned laddr = (unsigned)(uintptr_t)(lds) + lane * 16;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
      if (MODE == 0 || MODE == 4 || MODE == 5) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(c[(u * K + k) & 3]) : "v"(m));
      }
      if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[(u * K + k) & 3]) : "v"(m2));
      }
      if (MODE == 2) {
        if (K == 1 || (u & 1) == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(ex) : "v"(e));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(c[u & 3]) : "v"(m));
      }
      if (MODE == 3) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(c[0]) : "v"(m));
      }
      if (MODE == 4 && (u & 3) == 0) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fr) : "v"(laddr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (MODE == 5) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(rd) : "a"(c[3]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) atomicMax(cycles + blockIdx.x, t1 - t0);
  float r = c[0] + c[1] + c[2] + c[3] + ex + rd + p[0].x + p[1].y + p[2].x + p[3].y + (float)fr[0];
  for (int u = 0; u < UNROLL; ++u) r += acc[u][0];
  if (r == 1234.5f) sink[0] = r;
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  unsigned long long* d_cyc;
  float* d_sink;
  CK(hipMalloc(&d_cyc, cus * sizeof(unsigned long long)));
  CK(hipMalloc(&d_sink, 64));
  unsigned long long* h = (unsigned long long*)malloc(cus * sizeof(unsigned long long));
  auto report = [&](const char* name) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d_cyc, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mm = 0;
    for (int i = 0; i < cus; ++i) mm += (double)h[i];
    printf("%-72s %7.2f cycles per MFMA\n", name, mm / cus / ((double)ITERS * UNROLL));
  };
#define RUN(MODE, K, NAME) do { for (int rep = 0; rep < 2; ++rep) { CK(hipMemset(d_cyc, 0, cus * sizeof(unsigned long long))); \
    hipLaunchKernelGGL((probe<MODE, K>), dim3(cus), dim3(256), 0, 0, d_cyc, d_sink); } report(NAME); } while (0)
  RUN(0, 0, "MFMA only");
  RUN(0, 1, "1 independent v_fma per MFMA");
  RUN(0, 2, "2 independent v_fma per MFMA");
  RUN(0, 3, "3 independent v_fma per MFMA");
  RUN(0, 4, "4 independent v_fma per MFMA");
  RUN(3, 1, "1 dependent v_fma per MFMA");
  RUN(3, 2, "2 dependent v_fma per MFMA");
  RUN(3, 3, "3 dependent v_fma per MFMA");
  RUN(1, 1, "1 independent v_pk_fma_f32 per MFMA");
  RUN(1, 2, "2 independent v_pk_fma_f32 per MFMA");
  RUN(2, 0, "1 v_fma per MFMA + v_exp every second MFMA");
  RUN(2, 1, "1 v_fma + 1 v_exp per MFMA");
  RUN(4, 0, "ds_read_b128 + s_waitcnt per 4 MFMAs");
  RUN(4, 1, "ds_read_b128 + s_waitcnt per 4 MFMAs, 1 independent v_fma per MFMA");
  RUN(4, 2, "ds_read_b128 + s_waitcnt per 4 MFMAs, 2 independent v_fma per MFMA");
  RUN(5, 0, "1 v_accvgpr_read per MFMA");
  RUN(5, 1, "1 v_accvgpr_read + 1 independent v_fma per MFMA");
  return 0;
}
