// Issue rate of ds_read_b64_tr_b16 (the transposing LDS read of wgrad_nhwc.hip / gemm_bf16.hip) for the address patterns
// the kernels use, against plain ds_read_b64 / ds_read_b128 on conflict-free addresses: one workgroup per CU, 4 waves,
// every wave issues NREAD reads back to back per iteration (results xor-ed so nothing is dropped), cycles from s_memtime.
//   lin64      ds_read_b64, lane * 8                      (512 B per instruction, conflict-free)
//   lin128     ds_read_b128, lane * 16                    (1 KiB per instruction)
//   tr_lin     ds_read_b64_tr_b16, lane * 8
//   tr_a(df)   wgrad's a-row image: pixel p = df + 8 g + 4 hf + i/4 (64 B per pixel), unit = cbi ^ bit3(p)
//   tr_dz      wgrad's dz-row image: pixel p = 8 g + 4 hf + i/4 (128 B per pixel), unit = cb ^ (bit1(p) | bit3(p) << 1)
//   tr_plain_a / tr_plain_dz   the same without the XOR swizzle
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/lds_tr_probe tools/lds_tr_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int NREAD = 64, ITERS = 200;

template <int KIND>
__global__ __launch_bounds__(256, 1)
void probe(const unsigned* __restrict__ offs /* [256] byte offsets per thread */, unsigned long long* cycles, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)smem + offs[threadIdx.x];
  unsigned acc = 0;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < NREAD; r += 8) {
      u4v w[8];
      u2v v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned a = base + (unsigned)((r + q) & 7) * 4608u;          // eight row images, like the kernels' rings
        if (KIND == 0) asm volatile("ds_read_b64 %0, %1" : "=v"(v[q]) : "v"(a));
        else if (KIND == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(w[q]) : "v"(a));
        else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[q]) : "v"(a));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) acc ^= (KIND == 1) ? (w[q][0] ^ w[q][3]) : (v[q][0] ^ v[q][1]);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == 0x12345u) sink[0] = acc;
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  unsigned* d_off;
  unsigned long long* d_cyc;
  unsigned* d_sink;
  CK(hipMalloc(&d_off, 256 * sizeof(unsigned)));
  CK(hipMalloc(&d_cyc, cus * sizeof(unsigned long long)));
  CK(hipMalloc(&d_sink, 64));
  unsigned h_off[256];
  unsigned long long* h_cyc = (unsigned long long*)malloc(cus * sizeof(unsigned long long));
  auto run = [&](const char* name, int kind, auto&& offset_of /* (wave, lane) -> bytes */, int bytes_per_lane) {
    for (int t = 0; t < 256; ++t) h_off[t] = offset_of(t >> 6, t & 63);
    CK(hipMemcpy(d_off, h_off, sizeof h_off, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(cus), dim3(256), 0, 0, d_off, d_cyc, d_sink);
      else if (kind == 1) hipLaunchKernelGGL(probe<1>, dim3(cus), dim3(256), 0, 0, d_off, d_cyc, d_sink);
      else hipLaunchKernelGGL(probe<2>, dim3(cus), dim3(256), 0, 0, d_off, d_cyc, d_sink);
      CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h_cyc, d_cyc, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < cus; ++i) mean += (double)h_cyc[i];
    mean /= cus;
    // __builtin_readcyclecounter = s_memtime: shader-clock cycles
    const double instr = (double)ITERS * NREAD * 4;          // per CU
    printf("%-28s %8.2f cycles per CU-instruction  %7.1f B/cycle per CU\n", name, mean / instr, instr * 64 * bytes_per_lane / mean);
  };
  run("lin64", 0, [](int w, int l) { return (unsigned)(l * 8); }, 8);
  run("lin128", 1, [](int w, int l) { return (unsigned)(l * 16); }, 16);
  run("tr_lin", 2, [](int w, int l) { return (unsigned)(l * 8); }, 8);
  for (int df = 0; df < 5; ++df) {
    char nm[32];
    snprintf(nm, sizeof nm, "tr_a df=%d", df);
    run(nm, 2, [df](int w, int l) { const int g = l >> 4, i = l & 15, cbi = w & 1; const int p = df + 8 * g + (i >> 2); return (unsigned)(p * 64 + ((cbi ^ ((p >> 3) & 1)) << 5) + (i & 3) * 8); }, 8);
  }
  run("tr_a df=0 hf=1", 2, [](int w, int l) { const int g = l >> 4, i = l & 15, cbi = w & 1; const int p = 8 * g + 4 + (i >> 2); return (unsigned)(p * 64 + ((cbi ^ ((p >> 3) & 1)) << 5) + (i & 3) * 8); }, 8);
  run("tr_plain_a", 2, [](int w, int l) { const int g = l >> 4, i = l & 15, cbi = w & 1; const int p = 8 * g + (i >> 2); return (unsigned)(p * 64 + (cbi << 5) + (i & 3) * 8); }, 8);
  for (int cb = 0; cb < 4; ++cb) {
    char nm[32];
    snprintf(nm, sizeof nm, "tr_dz cb=%d", cb);
    run(nm, 2, [cb](int w, int l) { const int g = l >> 4, i = l & 15; const int p = 8 * g + (i >> 2); const int u = ((p >> 1) & 1) | (((p >> 3) & 1) << 1); return (unsigned)(p * 128 + ((cb ^ u) << 5) + (i & 3) * 8); }, 8);
  }
  run("tr_plain_dz", 2, [](int w, int l) { const int g = l >> 4, i = l & 15; const int p = 8 * g + (i >> 2); return (unsigned)(p * 128 + (i & 3) * 8); }, 8);
  // the conv kernel's B-fragment read: ds_read_b128 of pixel n + df, piece g ^ swz(p)
  for (int df = 0; df < 5; ++df) {
    char nm[32];
    snprintf(nm, sizeof nm, "conv b128 df=%d", df);
    run(nm, 1, [df](int w, int l) { const int n = l & 15, g = l >> 4; const int p = n + df; return (unsigned)(p * 128 + ((g ^ (((p >> 1) & 3) << 1)) << 4)); }, 16);
  }
  run("conv b128 plain", 1, [](int w, int l) { const int n = l & 15, g = l >> 4; return (unsigned)(n * 128 + (g << 4)); }, 16);
  return 0;
}
