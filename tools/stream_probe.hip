// Probe (round 6): what limits the BatchNorm apply pass -- 16 bytes per lane in, 16 out, 1.48 GB each way -- at 4.7-5.1 TB/s when a float4
// copy reaches 6.3 (MI355X_MICROARCH.md)?  The pass without its arithmetic, in variants of grid shape, loop shape and cache policy.
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_probe tools/stream_probe.hip && tools/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ u4v work(u4v v) { v[0] ^= 0x00010001u; return v; }      // (keeps the compiler from turning the loop into a memcpy)

// LD / ST: 0 = plain, 1 = nontemporal.  NF: pieces in flight per lane.  grid-stride over all pieces.
template <int LD, int ST, int NF>
__global__ __launch_bounds__(256) void stride_kernel(const u4v* __restrict__ z, u4v* __restrict__ a, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (NF - 1) * stride < n; i += NF * stride) {
    u4v v[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) v[k] = LD ? __builtin_nontemporal_load(z + i + k * stride) : z[i + k * stride];
#pragma unroll
    for (int k = 0; k < NF; ++k) { if (ST) __builtin_nontemporal_store(work(v[k]), a + i + k * stride); else a[i + k * stride] = work(v[k]); }
  }
  for (; i < n; i += stride) { if (ST) __builtin_nontemporal_store(work(z[i]), a + i); else a[i] = work(z[i]); }
}
// every workgroup owns ONE contiguous chunk of the tensor
template <int LD, int ST, int NF>
__global__ __launch_bounds__(256) void chunk_kernel(const u4v* __restrict__ z, u4v* __restrict__ a, long long n) {
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  long long i = lo + threadIdx.x;
  for (; i + (NF - 1) * 256 < hi; i += NF * 256) {
    u4v v[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) v[k] = LD ? __builtin_nontemporal_load(z + i + k * 256) : z[i + k * 256];
#pragma unroll
    for (int k = 0; k < NF; ++k) { if (ST) __builtin_nontemporal_store(work(v[k]), a + i + k * 256); else a[i + k * 256] = work(v[k]); }
  }
  for (; i < hi; i += 256) { if (ST) __builtin_nontemporal_store(work(z[i]), a + i); else a[i] = work(z[i]); }
}

template <typename K>
static float run(K kern, int grid, const u4v* z, u4v* a, long long n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, z, a, n);
  hipEventRecord(e0, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, z, a, n);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}

int main() {
  const long long n = 64LL * 301 * 601 * 8;            // 16-byte pieces of one [64, 301, 601, 64] bf16 tensor
  u4v *z, *a;
  CK(hipMalloc(&z, n * 16)); CK(hipMalloc(&a, n * 16));
  CK(hipMemset(z, 1, n * 16)); CK(hipMemset(a, 0, n * 16));
  const double gb = 2.0 * n * 16 / 1e9;
#define ROW(name, kern, grid) { float ms = run(kern, grid, z, a, n); printf("%-58s grid %7d  %.3f ms  %.2f TB/s\n", name, grid, ms, gb / ms); }
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    ROW("stride, nt load + nt store, 2 in flight (the product)", (stride_kernel<1, 1, 2>), grid);
  }
  ROW("stride, nt + nt, 1 in flight", (stride_kernel<1, 1, 1>), 2048);
  ROW("stride, nt + nt, 4 in flight", (stride_kernel<1, 1, 4>), 2048);
  ROW("stride, nt + nt, 8 in flight", (stride_kernel<1, 1, 8>), 2048);
  ROW("stride, plain load + plain store, 2 in flight", (stride_kernel<0, 0, 2>), 2048);
  ROW("stride, nt load + plain store, 2 in flight", (stride_kernel<1, 0, 2>), 2048);
  ROW("stride, plain load + nt store, 2 in flight", (stride_kernel<0, 1, 2>), 2048);
  ROW("stride, plain + plain, 4 in flight", (stride_kernel<0, 0, 4>), 2048);
  ROW("stride, one sweep (grid = pieces / 512), nt + nt", (stride_kernel<1, 1, 2>), (int)(n / 512));
  ROW("stride, one sweep, plain + plain", (stride_kernel<0, 0, 2>), (int)(n / 512));
  for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
    ROW("chunk per workgroup, nt + nt, 4 in flight", (chunk_kernel<1, 1, 4>), grid);
  }
  ROW("chunk per workgroup, plain + plain, 4 in flight", (chunk_kernel<0, 0, 4>), 2048);
  ROW("chunk per workgroup, nt + nt, 8 in flight", (chunk_kernel<1, 1, 8>), 2048);
  ROW("chunk per workgroup, nt + nt, 2 in flight", (chunk_kernel<1, 1, 2>), 2048);
  {   // the runtime's own copy
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipMemcpyAsync(a, z, n * 16, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipMemcpyAsync(a, z, n * 16, hipMemcpyDeviceToDevice, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-58s               %.3f ms  %.2f TB/s\n", "hipMemcpyAsync device to device", ms, gb / ms);
  }
  return 0;
}
