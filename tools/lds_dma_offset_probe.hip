// buffer_load_dwordx4 ... offen offset:IMM lds on gfx950: is the instruction's immediate offset added to the memory address
// only, or to the LDS address (M0 base + 16 * lane) as well?  conv_nhwc.hip / wgrad_nhwc.hip issue the 1 KiB chunks of a row
// with one descriptor and immediates 0 / 1024 / 2048 / 3072.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/lds_dma_offset_probe tools/lds_dma_offset_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned* src /* 4096 words */, unsigned* out /* 4096 words = 16 KiB of LDS */) {
  __shared__ __attribute__((aligned(16))) unsigned smem[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) smem[i] = 0xeeeeeeeeu;
  __syncthreads();
  const unsigned long long p = (unsigned long long)src;
  const u4v d = {(unsigned)p, (unsigned)(p >> 32) & 0xffffu, 4096 * 4, 0x00020000u};
  const unsigned voff = threadIdx.x * 16, lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)smem;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen offset:1024 lds\n\t"
               "buffer_load_dwordx4 %1, %2, 0 offen offset:3072 lds\n\t"
               "s_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(voff), "s"(d), "s"(lds) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = smem[i];
}

int main() {
  unsigned *src, *out, h[4096], o[4096];
  CK(hipMalloc(&src, sizeof h));
  CK(hipMalloc(&out, sizeof o));
  for (int i = 0; i < 4096; ++i) h[i] = i;
  CK(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
  // report, per 1 KiB block of LDS, which source word sits at its start
  for (int blk = 0; blk < 16; ++blk) {
    const unsigned v = o[blk * 256];
    if (v == 0xeeeeeeeeu) printf("LDS block %2d: untouched\n", blk);
    else printf("LDS block %2d: source words %u.. (source block %u)%s\n", blk, v, v / 256, o[blk * 256 + 255] == v + 255 ? "" : "  [not contiguous]");
  }
  return 0;
}
