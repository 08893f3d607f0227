// What does an instruction placed between two MFMAs cost a wave that has its SIMD to itself (the conv / weight-gradient
// kernels run one wave per SIMD)?  One workgroup per CU, 4 waves; a loop of independent v_mfma_f32_16x16x32_bf16 (4 passes =
// 16 cycles each) or v_mfma_f32_32x32x16_bf16 (8 passes) with K filler instructions of one class after every MFMA;
// cycles per MFMA from s_memtime.  With nothing in between the loop runs at the pipe rate; the K at which the time starts
// to grow is the issue budget of that class.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/issue_probe tools/issue_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int ITERS = 2000, UNROLL = 8;

// CLASS: 0 none, 1 SALU (s_add_i32), 2 VALU (v_add_u32), 3 s_waitcnt lgkmcnt(15), 4 ds_read_b128, 5 s_nop 0, 6 v_add + ds_read_b128 pairs
template <int CLASS, int K, bool BIG>
__global__ __launch_bounds__(256, 1)
void probe(unsigned long long* cycles, float* sink, int zero) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[16 * 1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3f803f80u;
  __syncthreads();
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + 0.01f * ((lane + i) & 7)); b[i] = (__bf16)(0.5f + 0.01f * ((lane * 3 + i) & 7)); }
  f32x4 acc[UNROLL];
  f32x16 big[4];
  for (int u = 0; u < UNROLL; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int u = 0; u < 4; ++u) for (int e = 0; e < 16; ++e) big[u][e] = 0.f;
  unsigned vsum = lane, ldsaddr = (unsigned)(uintptr_t)smem + lane * 16;
  int ssum = zero;
  u4v ld = {0, 0, 0, 0};
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (BIG) big[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, big[u & 3], 0, 0, 0);
      else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (CLASS == 1) asm volatile("s_add_i32 %0, %0, 1" : "+s"(ssum));
        else if (CLASS == 2) asm volatile("v_add_u32 %0, %0, 1" : "+v"(vsum));
        else if (CLASS == 3) asm volatile("s_waitcnt lgkmcnt(15)");
        else if (CLASS == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(ldsaddr));
        else if (CLASS == 5) asm volatile("s_nop 0");
        else if (CLASS == 6) { asm volatile("v_add_u32 %0, %0, 0" : "+v"(ldsaddr)); asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(ldsaddr)); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (CLASS == 4 || CLASS == 6) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  float r = (float)ssum + (float)vsum + (float)ld[0];
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
  auto report = [&](const char* name) {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d_cyc, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < cus; ++i) m += (double)h[i];
    printf("%-44s %7.2f cycles per MFMA\n", name, m / cus / ((double)ITERS * UNROLL));
  };
#define RUN(CLASS, K, BIG, NAME) do { for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<CLASS, K, BIG>), dim3(cus), dim3(256), 0, 0, d_cyc, d_sink, 0); report(NAME); } while (0)
  RUN(0, 0, false, "16x16x32: nothing between");
  RUN(1, 1, false, "16x16x32: 1 SALU"); RUN(1, 2, false, "16x16x32: 2 SALU"); RUN(1, 3, false, "16x16x32: 3 SALU"); RUN(1, 4, false, "16x16x32: 4 SALU");
  RUN(1, 6, false, "16x16x32: 6 SALU"); RUN(1, 8, false, "16x16x32: 8 SALU"); RUN(1, 12, false, "16x16x32: 12 SALU");
  RUN(2, 1, false, "16x16x32: 1 VALU"); RUN(2, 2, false, "16x16x32: 2 VALU"); RUN(2, 3, false, "16x16x32: 3 VALU"); RUN(2, 4, false, "16x16x32: 4 VALU"); RUN(2, 6, false, "16x16x32: 6 VALU");
  RUN(3, 1, false, "16x16x32: 1 s_waitcnt"); RUN(3, 2, false, "16x16x32: 2 s_waitcnt"); RUN(3, 4, false, "16x16x32: 4 s_waitcnt"); RUN(3, 8, false, "16x16x32: 8 s_waitcnt");
  RUN(5, 2, false, "16x16x32: 2 s_nop"); RUN(5, 4, false, "16x16x32: 4 s_nop"); RUN(5, 8, false, "16x16x32: 8 s_nop");
  RUN(4, 1, false, "16x16x32: 1 ds_read_b128"); RUN(4, 2, false, "16x16x32: 2 ds_read_b128");
  RUN(6, 1, false, "16x16x32: 1 (v_add + ds_read_b128)"); RUN(6, 2, false, "16x16x32: 2 (v_add + ds_read_b128)");
  RUN(0, 0, true, "32x32x16: nothing between");
  RUN(1, 4, true, "32x32x16: 4 SALU"); RUN(1, 8, true, "32x32x16: 8 SALU"); RUN(1, 16, true, "32x32x16: 16 SALU"); RUN(1, 24, true, "32x32x16: 24 SALU");
  RUN(2, 2, true, "32x32x16: 2 VALU"); RUN(2, 4, true, "32x32x16: 4 VALU"); RUN(2, 6, true, "32x32x16: 6 VALU"); RUN(2, 8, true, "32x32x16: 8 VALU");
  RUN(6, 1, true, "32x32x16: 1 (v_add + ds_read_b128)"); RUN(6, 2, true, "32x32x16: 2 (v_add + ds_read_b128)"); RUN(6, 3, true, "32x32x16: 3 (v_add + ds_read_b128)");
  return 0;
}
