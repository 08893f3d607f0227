// Does a second wave on the SIMD hide the side work of a matrix-pipe kernel?  (the structural choice
// between 32x32x16 MFMAs -- longer gaps for one wave's own side instructions -- and two waves per SIMD.)
// One workgroup per CU of 256 threads (one wave per SIMD) or 512 threads (two per SIMD); every wave runs the same loop:
// independent MFMAs with K VALU instructions (a dependent chain of v_fma_f32, every fourth one a v_exp_f32) after each.
// Reported: cycles per MFMA PER SIMD (elapsed / MFMAs issued on the SIMD by all its waves): 16 (32) = the pipe rate.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/occupancy_probe tools/occupancy_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ITERS = 1500, UNROLL = 8;

template <int K, bool BIG, int WAVES>
__global__ __launch_bounds__(64 * WAVES)
void probe(unsigned long long* cycles, float* sink) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + 0.01f * ((lane + i) & 7)); b[i] = (__bf16)(0.5f + 0.01f * ((lane * 3 + i) & 7)); }
  f32x4 acc[UNROLL];
  f32x16 big[4];
  for (int u = 0; u < UNROLL; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int u = 0; u < 4; ++u) for (int e = 0; e < 16; ++e) big[u][e] = 0.f;
  float v0 = 0.001f * lane, v1 = 0.5f, v2 = 1.0f;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (BIG) big[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[u & 3], 0, 0, 0);
      else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if ((u * K + k) % 4 == 3) asm volatile("v_exp_f32 %0, %1" : "=v"(v2) : "v"(v0));
        else if (k & 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(v1), "v"(v2));
        else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v1) : "v"(v0), "v"(v2));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) atomicMax(cycles + blockIdx.x, t1 - t0);
  float r = v0 + v1 + v2;
  for (int u = 0; u < UNROLL; ++u) r += acc[u][0];
  for (int u = 0; u < 4; ++u) r += big[u][0];
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
  auto report = [&](const char* name, int waves_per_simd) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d_cyc, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < cus; ++i) m += (double)h[i];
    printf("%-56s %7.2f cycles per MFMA per SIMD\n", name, m / cus / ((double)ITERS * UNROLL * waves_per_simd));
  };
#define RUN(K, BIG, W, NAME) do { for (int rep = 0; rep < 2; ++rep) { CK(hipMemset(d_cyc, 0, cus * sizeof(unsigned long long))); \
    hipLaunchKernelGGL((probe<K, BIG, W>), dim3(cus), dim3(64 * W), 0, 0, d_cyc, d_sink); } report(NAME, W / 4); } while (0)
  RUN(0, false, 4, "16x16x32, 1 wave/SIMD, 0 VALU per MFMA");
  RUN(1, false, 4, "16x16x32, 1 wave/SIMD, 1 VALU per MFMA");
  RUN(2, false, 4, "16x16x32, 1 wave/SIMD, 2 VALU per MFMA");
  RUN(3, false, 4, "16x16x32, 1 wave/SIMD, 3 VALU per MFMA");
  RUN(4, false, 4, "16x16x32, 1 wave/SIMD, 4 VALU per MFMA");
  RUN(0, false, 8, "16x16x32, 2 waves/SIMD, 0 VALU per MFMA");
  RUN(1, false, 8, "16x16x32, 2 waves/SIMD, 1 VALU per MFMA");
  RUN(2, false, 8, "16x16x32, 2 waves/SIMD, 2 VALU per MFMA");
  RUN(3, false, 8, "16x16x32, 2 waves/SIMD, 3 VALU per MFMA");
  RUN(4, false, 8, "16x16x32, 2 waves/SIMD, 4 VALU per MFMA");
  RUN(0, true, 4, "32x32x16, 1 wave/SIMD, 0 VALU per MFMA");
  RUN(2, true, 4, "32x32x16, 1 wave/SIMD, 2 VALU per MFMA");
  RUN(4, true, 4, "32x32x16, 1 wave/SIMD, 4 VALU per MFMA");
  RUN(6, true, 4, "32x32x16, 1 wave/SIMD, 6 VALU per MFMA");
  RUN(8, true, 4, "32x32x16, 1 wave/SIMD, 8 VALU per MFMA");
  RUN(4, true, 8, "32x32x16, 2 waves/SIMD, 4 VALU per MFMA");
  RUN(8, true, 8, "32x32x16, 2 waves/SIMD, 8 VALU per MFMA");
  return 0;
}
