// Hardware probes behind the channels-last bf16 conv path (csrc/conv_nhwc.hip): what the guide states about
// three gfx950 instructions, checked on the device before kernels are built on it.
//   A  ds_read_b64_tr_b16: which 4 halves lane l receives for a given per-lane address pattern
//   B  global_load_lds_dwordx4 issued from inline asm with M0 = wave-uniform LDS base: lane i lands at base + 16 i,
//      the per-lane source address is free (swizzled source, redirect to a zero page)
//   C  operand / result layout of v_mfma_f32_16x16x32_bf16
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe_cdna4 tools/probe_cdna4.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// ---- A ---------------------------------------------------------------------------------------
__global__ void probe_tr(const int* __restrict__ addr_halves, unsigned short* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)addr_halves[threadIdx.x] * 2u;      // byte address inside the dynamic LDS block
  u2v r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = (unsigned short)(r[0] & 0xffff);
  out[threadIdx.x * 4 + 1] = (unsigned short)(r[0] >> 16);
  out[threadIdx.x * 4 + 2] = (unsigned short)(r[1] & 0xffff);
  out[threadIdx.x * 4 + 3] = (unsigned short)(r[1] >> 16);
}

// ---- B ---------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ void probe_glds(const u4v* __restrict__ src, const int* __restrict__ pick, u4v* __restrict__ out, int lds_base_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u4v* l4 = reinterpret_cast<u4v*>(smem);
  for (int i = threadIdx.x; i < 1024; i += 256) l4[i] = u4v{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(lds_base_bytes + wave * 1024);
  glds16(src + pick[threadIdx.x], dst);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = l4[i];
}

// ---- C ---------------------------------------------------------------------------------------
__global__ void probe_mfma16(const __bf16* __restrict__ A /*[16][32]*/, const __bf16* __restrict__ Bm /*[32][16]*/, float* __restrict__ C /*[16][16]*/) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = Bm[(8 * (l >> 4) + j) * 16 + (l & 15)];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

int main() {
  // A: three address patterns
  {
    int* d_addr; unsigned short* d_out;
    CK(hipMalloc(&d_addr, 64 * 4)); CK(hipMalloc(&d_out, 256 * 2));
    for (int pat = 0; pat < 3; ++pat) {
      std::vector<int> addr(64);
      for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        if (pat == 0) addr[l] = g * 64 + i * 4;                        // the guide's canonical 4x16 block per 16-lane group
        if (pat == 1) addr[l] = g * 1000 + (i >> 2) * 72 + (i & 3) * 4; // free row stride (72 halves), blocks 1000 halves apart
        if (pat == 2) addr[l] = (i >> 2) * 64 + g * 16 + (i & 3) * 4;   // [4 rows][64 ch] image: group g = channel quarter
      }
      CK(hipMemcpy(d_addr, addr.data(), 256, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 16384, 0, d_addr, d_out);
      CK(hipDeviceSynchronize());
      std::vector<unsigned short> out(256);
      CK(hipMemcpy(out.data(), d_out, 512, hipMemcpyDeviceToHost));
      // hypothesis: out[lane (g,i)][j] = value at the address supplied by lane (g, 4*j + i/4), element i%4
      int bad = 0;
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
          const int g = l >> 4, i = l & 15;
          const int expect = addr[g * 16 + 4 * j + (i >> 2)] + (i & 3);
          if (out[l * 4 + j] != expect) ++bad;
        }
      printf("probe A pattern %d: hypothesis out[g,i][j] = in[lane (g,4j+i/4)][i%%4] -> %s (%d mismatches)\n", pat, bad ? "FAIL" : "PASS", bad);
      if (bad || pat == 0) {
        for (int l = 0; l < 64; l += (bad ? 1 : 17))
          printf("   lane %2d addr %5d -> %5d %5d %5d %5d\n", l, addr[l], out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
      }
    }
  }
  // B
  {
    const int NSRC = 4096;
    std::vector<unsigned> src(NSRC * 4);
    for (int i = 0; i < NSRC; ++i) for (int e = 0; e < 4; ++e) src[i * 4 + e] = (unsigned)(i * 4 + e);
    for (int e = 0; e < 4; ++e) src[7 * 4 + e] = 0;                       // element 7 plays the zero page
    std::vector<int> pick(256);
    for (int t = 0; t < 256; ++t) pick[t] = (t % 5 == 0) ? 7 : ((t * 37 + 11) % NSRC);   // scattered sources, every 5th to the zero page
    u4v *d_src, *d_out; int* d_pick;
    CK(hipMalloc(&d_src, NSRC * 16)); CK(hipMalloc(&d_out, 1024 * 16)); CK(hipMalloc(&d_pick, 1024));
    CK(hipMemcpy(d_src, src.data(), NSRC * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pick, pick.data(), 1024, hipMemcpyHostToDevice));
    for (int base : {0, 4096}) {
      hipLaunchKernelGGL(probe_glds, dim3(1), dim3(256), 16384, 0, d_src, d_pick, d_out, base);
      CK(hipDeviceSynchronize());
      std::vector<unsigned> out(1024 * 4);
      CK(hipMemcpy(out.data(), d_out, 1024 * 16, hipMemcpyDeviceToHost));
      int bad = 0, untouched_bad = 0;
      for (int t = 0; t < 256; ++t) {
        const int slot = base / 16 + t;               // wave w lands at base + w*1024 + lane*16
        for (int e = 0; e < 4; ++e) if (out[slot * 4 + e] != src[pick[t] * 4 + e]) ++bad;
      }
      for (int s = 0; s < 1024; ++s) {
        if (s >= base / 16 && s < base / 16 + 256) continue;
        for (int e = 0; e < 4; ++e) if (out[s * 4 + e] != 0xdeadbeefu) ++untouched_bad;
      }
      printf("probe B lds base %5d: lane i -> base + wave*1024 + 16 i with a free per-lane source: %s (%d wrong words, %d stray writes)\n",
             base, (bad || untouched_bad) ? "FAIL" : "PASS", bad, untouched_bad);
      if (bad) for (int t = 0; t < 8; ++t) printf("   t %d expect %u got %u\n", t, src[pick[t] * 4], out[(base / 16 + t) * 4]);
    }
  }
  // C
  {
    std::vector<unsigned short> A(16 * 32), Bm(32 * 16);
    std::vector<float> Af(16 * 32), Bf(32 * 16);
    srand(3);
    for (int i = 0; i < 16 * 32; ++i) { Af[i] = (float)(rand() % 9 - 4); A[i] = f2bf(Af[i]); }
    for (int i = 0; i < 32 * 16; ++i) { Bf[i] = (float)(rand() % 7 - 3); Bm[i] = f2bf(Bf[i]); }
    __bf16 *dA, *dB; float* dC;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, Bm.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_mfma16, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    CK(hipDeviceSynchronize());
    std::vector<float> C(256);
    CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int m = 0; m < 16; ++m)
      for (int n = 0; n < 16; ++n) {
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s += Af[m * 32 + k] * Bf[k * 16 + n];
        if (C[m * 16 + n] != s) ++bad;
      }
    printf("probe C mfma_f32_16x16x32_bf16: A[m=l&15][k=8(l>>4)+j], B[k=8(l>>4)+j][n=l&15], C[m=4(l>>4)+r][n=l&15]: %s (%d mismatches)\n",
           bad ? "FAIL" : "PASS", bad);
  }
  return 0;
}
