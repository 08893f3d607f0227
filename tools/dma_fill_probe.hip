// How fast can one MI355X move bytes HBM -> LDS with global_load_lds_dwordx4 (the staging path of conv_nhwc.hip,
// wgrad_nhwc.hip, gemm_bf16.hip), as a function of the bytes a workgroup keeps in flight and of the access pattern?
// One persistent workgroup per CU, 4 waves; every round a wave issues `depth` 1 KiB DMA instructions, then waits for
// them (vmcnt(0)) -- the structure of the kernels' group loop with the compute removed.  Patterns:
//   0  1 KiB contiguous per instruction (a dz / conv window row)
//   1  64-byte runs at a 128-byte stride (the weight gradient's half-pixel a rows)
//   2  like 0, but `ahead` rounds stay in flight (counted vmcnt) instead of draining every round
// Also the same stream through global_load_dwordx4 into registers, for reference.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/dma_fill_probe tools/dma_fill_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// bytes_per_wg: the contiguous slab a workgroup streams; depth: DMA instructions per wave and round
template <int PATTERN>
__global__ __launch_bounds__(256, 1)
void fill_kernel(const unsigned char* __restrict__ src, long long bytes_per_wg, int depth, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned char* base = src + (long long)blockIdx.x * bytes_per_wg;
  const long long round_bytes = (long long)depth * 4 * 1024;                 // all four waves
  const long long rounds = bytes_per_wg / (PATTERN == 1 ? 2 * round_bytes : round_bytes);
  for (long long r = 0; r < rounds; ++r) {
    for (int d = 0; d < depth; ++d) {
      const long long chunk = (r * depth + d) * 4 + wave;                     // 1 KiB units
      const unsigned char* p;
      if (PATTERN == 1) p = base + chunk * 2048 + (lane >> 2) * 128 + (lane & 3) * 16;   // 16 pixels x 64 of their 128 bytes
      else p = base + chunk * 1024 + lane * 16;
      glds16(p, lds0 + (unsigned)(((d * 4 + wave) * 1024) & (128 * 1024 - 1)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(volatile unsigned*)smem;
}

// L2-resident source: the workgroup re-reads a `wrap`-byte slab (its own, or -- shared -- one slab for the whole chip, as
// the workgroups of a GEMM share operand panels), so after the first sweep every line is an L2 hit: the rate of the
// L2 -> LDS path itself, which the HBM-streaming runs above cannot show.
template <bool SHARED>
__global__ __launch_bounds__(256, 1)
void fill_l2_kernel(const unsigned char* __restrict__ src, long long bytes_per_wg, long long wrap, int depth, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned char* base = SHARED ? src : src + (long long)blockIdx.x * wrap;
  const long long round_bytes = (long long)depth * 4 * 1024;
  const long long rounds = bytes_per_wg / round_bytes;
  for (long long r = 0; r < rounds; ++r) {
    for (int d = 0; d < depth; ++d) {
      const long long chunk = (r * depth + d) * 4 + wave;
      const unsigned char* p = base + ((chunk * 1024) & (wrap - 1)) + lane * 16;
      glds16(p, lds0 + (unsigned)(((d * 4 + wave) * 1024) & (128 * 1024 - 1)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(volatile unsigned*)smem;
}

// `ahead` rounds in flight: round r waits only for round r - ahead (vmcnt counts this wave's later instructions)
template <int DEPTH, int AHEAD>
__global__ __launch_bounds__(256, 1)
void fill_ahead_kernel(const unsigned char* __restrict__ src, long long bytes_per_wg, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const unsigned char* base = src + (long long)blockIdx.x * bytes_per_wg;
  const long long rounds = bytes_per_wg / ((long long)DEPTH * 4 * 1024);
  for (long long r = 0; r < rounds; ++r) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const long long chunk = (r * DEPTH + d) * 4 + wave;
      glds16(base + chunk * 1024 + lane * 16, lds0 + (unsigned)((((r % (AHEAD + 1)) * DEPTH + d) * 4 + wave) * 1024));
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * AHEAD) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(volatile unsigned*)smem;
}

__global__ __launch_bounds__(256, 1)
void reg_kernel(const u4v* __restrict__ src, long long pieces_per_wg, int depth, unsigned* sink) {
  const u4v* base = src + (long long)blockIdx.x * pieces_per_wg;
  u4v acc = {0, 0, 0, 0};
  const long long rounds = pieces_per_wg / ((long long)depth * 256);
  for (long long r = 0; r < rounds; ++r) {
    for (int d = 0; d < depth; ++d) {
      const u4v v = __builtin_nontemporal_load(base + (r * depth + d) * 256 + threadIdx.x);
      acc ^= v;
    }
  }
  if (sink && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = 1;
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const long long per_wg = 24LL << 20;                      // 24 MiB per workgroup: 6 GiB in all at 256 CUs
  const long long total = per_wg * cus;
  unsigned char* src;
  unsigned* sink;
  CK(hipMalloc(&src, total));
  CK(hipMalloc(&sink, 4096 * sizeof(unsigned)));
  CK(hipMemset(src, 1, total));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, double bytes) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %8.3f ms  %7.2f TB/s  %6.1f GB/s per CU\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e6 / cus);
  };
  printf("CUs: %d, %.2f GiB streamed per run\n", cus, total / 1073741824.0);
  char name[128];
  for (int depth : {2, 4, 8, 14, 24, 32}) {
    snprintf(name, sizeof name, "LDS-DMA contiguous, %d KiB in flight per workgroup, drain per round", depth * 4);
    run(name, [&] { hipLaunchKernelGGL(fill_kernel<0>, dim3(cus), dim3(256), 0, 0, src, per_wg, depth, sink); }, (double)total);
  }
  for (int depth : {8, 14, 32}) {
    snprintf(name, sizeof name, "LDS-DMA 64 B runs / 128 B stride, %d KiB in flight, drain per round", depth * 4);
    run(name, [&] { hipLaunchKernelGGL(fill_kernel<1>, dim3(cus), dim3(256), 0, 0, src, per_wg, depth, sink); }, (double)total / 2);
  }
  run("LDS-DMA contiguous, 8 per round, 1 round ahead (64 KiB in flight)", [&] { hipLaunchKernelGGL((fill_ahead_kernel<8, 1>), dim3(cus), dim3(256), 0, 0, src, per_wg, sink); }, (double)total);
  run("LDS-DMA contiguous, 8 per round, 2 rounds ahead (96 KiB in flight)", [&] { hipLaunchKernelGGL((fill_ahead_kernel<8, 2>), dim3(cus), dim3(256), 0, 0, src, per_wg, sink); }, (double)total);
  run("LDS-DMA contiguous, 4 per round, 3 rounds ahead (64 KiB in flight)", [&] { hipLaunchKernelGGL((fill_ahead_kernel<4, 3>), dim3(cus), dim3(256), 0, 0, src, per_wg, sink); }, (double)total);
  run("LDS-DMA contiguous, 2 per round, 7 rounds ahead (64 KiB in flight)", [&] { hipLaunchKernelGGL((fill_ahead_kernel<2, 7>), dim3(cus), dim3(256), 0, 0, src, per_wg, sink); }, (double)total);
  for (long long wrap : {64LL << 10, 1LL << 20}) {
    for (int depth : {4, 8, 16, 32}) {
      snprintf(name, sizeof name, "LDS-DMA from L2: own %lld KiB slab re-read, %d KiB in flight, drain per round", wrap >> 10, depth * 4);
      run(name, [&] { hipLaunchKernelGGL(fill_l2_kernel<false>, dim3(cus), dim3(256), 0, 0, src, per_wg, wrap, depth, sink); }, (double)total);
    }
  }
  for (int depth : {8, 16, 32}) {
    snprintf(name, sizeof name, "LDS-DMA from L2: ONE 2 MiB slab shared by all workgroups, %d KiB in flight", depth * 4);
    run(name, [&] { hipLaunchKernelGGL(fill_l2_kernel<true>, dim3(cus), dim3(256), 0, 0, src, per_wg, 2LL << 20, depth, sink); }, (double)total);
  }
  for (int depth : {2, 4, 8}) {
    snprintf(name, sizeof name, "global_load_dwordx4 to registers, %d x 4 KiB per workgroup and round", depth);
    run(name, [&] { hipLaunchKernelGGL(reg_kernel, dim3(cus), dim3(256), 0, 0, (const u4v*)src, per_wg / 16, depth, sink); }, (double)total);
  }
  return 0;
}
