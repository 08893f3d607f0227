// Where do the workgroups of a launch land?  Prints HW_REG_XCC_ID and HW_REG_HW_ID per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg(6164);                 // hwreg(HW_REG_XCC_ID, 0, 4)
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // hwreg(HW_REG_HW_ID, 0, 32)
  }
  // stay resident for a while so that later blocks cannot reuse the slot
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(64);
}
int main() {
  const int n = 560;
  unsigned* d;
  hipMalloc(&d, n * 8);
  hipMemset(d, 0xff, n * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    int cnt[16] = {0}, mism = 0;
    for (int i = 0; i < n; ++i) { cnt[h[2 * i] & 15]++; if ((h[2 * i] & 15) != (unsigned)(i % 8)) ++mism; }
    printf("rep %d: per-XCC counts:", rep);
    for (int x = 0; x < 16; ++x) printf(" %d", cnt[x]);
    printf("   blocks with xcc != blockIdx %% 8: %d\n", mism);
    printf("  first 24 blocks (xcc, hw_id): ");
    for (int i = 0; i < 24; ++i) printf("(%u,%08x) ", h[2 * i], h[2 * i + 1]);
    printf("\n");
  }
  return 0;
}
