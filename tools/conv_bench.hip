// Standalone A/B harness for the 64->64 split-f16 conv kernels (no Python: a fresh GPU box pays
// ~2 minutes for its first `import torch`).  Links libvoicesplit_hip.so and drives it through the
// C ABI only.  For every case: the one-tile-per-workgroup kernel (vs_set_conv_kernel(1)) and the
// persistent pipelined kernel (2) run on the same operands; outputs must be BIT-identical (same K
// order, fp32 accumulation), repeated launches of the persistent kernel must be bit-stable (race
// screen), and both are timed with HIP events.
//   tools/conv_bench [B=64] [reps=10]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../include/voicesplit_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define VS(x) do { int rc_ = (x); if (rc_) { printf("vs error %d (%s) at %s:%d\n", rc_, vs_last_error(), __FILE__, __LINE__); exit(3); } } while (0)

static unsigned g_seed = 12345u;
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffffff) / 8388608.0f - 1.0f; }

__global__ void diff_kernel(const unsigned* a, const unsigned* b, long long n, unsigned long long* cnt, unsigned* firstbad) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long c = 0;
  for (; i < n; i += (long long)gridDim.x * blockDim.x)
    if (a[i] != b[i]) { ++c; atomicMin(firstbad, (unsigned)(i < 0xffffffffLL ? i : 0xfffffffe)); }
  if (c) atomicAdd(cnt, c);
}

// ---- FETCH_SIZE / WRITE_SIZE calibration (MI355X_MICROARCH.md: "calibrate on a known byte count in your own
// access pattern"): stream n floats once with (a) one dword per lane per instruction through a buffer
// descriptor -- the conv kernels' window loads -- and (b) one dwordx4 per lane; write n floats once with dword
// stores.  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over `conv_bench 0 0 calib` then shows what the counters
// report for exactly 1 GiB read / written.
__global__ void calib_read_dword(const float* x, long long n, float* sink) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0xffffffffu, 0x00020000);
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (unsigned)(i * 4), 0, 0));
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void calib_read_dwordx4(const float4* x, long long n4, float* sink) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = x[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void calib_write_dword(float* y, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = (float)i;
}

struct Case { int B, T, F, dil, act; };

int main(int argc, char** argv) {
  const int Bbig = argc > 1 ? atoi(argv[1]) : 64;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  if (argc > 3 && !strcmp(argv[3], "calib")) {
    const long long n = 1LL << 28;      // 2^28 floats = exactly 1 GiB
    float *x, *y, *sink;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 1, n * 4));
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(calib_read_dword, dim3(4096), dim3(256), 0, nullptr, x, n, sink);
      hipLaunchKernelGGL(calib_read_dwordx4, dim3(4096), dim3(256), 0, nullptr, (const float4*)x, n / 4, sink);
      hipLaunchKernelGGL(calib_write_dword, dim3(4096), dim3(256), 0, nullptr, y, n);
    }
    CK(hipDeviceSynchronize());
    printf("calib: 3 x (1 GiB dword read, 1 GiB dwordx4 read, 1 GiB dword write)\n");
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "mall")) {
    // Does a buffer that was just streamed come back faster the second time (memory-side cache, 256 MB)?
    // For each size: read the SAME region twice in a row vs read two DIFFERENT regions, 16-byte loads.
    const long long total = 3LL << 30;      // 3 GiB arena (bytes)
    float *x, *sink;
    CK(hipMalloc(&x, total)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 1, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (long long mb : {32LL, 64LL, 92LL, 128LL, 184LL, 256LL, 512LL}) {
      const long long n4 = mb * 1024 * 1024 / 16;
      auto run = [&](bool same) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          // flush: stream 1 GiB of something else
          hipLaunchKernelGGL(calib_read_dwordx4, dim3(4096), dim3(256), 0, nullptr, (const float4*)(x + (2LL << 30) / 4), (1LL << 30) / 16, sink);
          hipLaunchKernelGGL(calib_read_dwordx4, dim3(4096), dim3(256), 0, nullptr, (const float4*)x, n4, sink);
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(calib_read_dwordx4, dim3(4096), dim3(256), 0, nullptr, (const float4*)(same ? x : x + (1LL << 30) / 4), n4, sink);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
        }
        return best;
      };
      const float a = run(true), b = run(false);
      printf("  %4lld MB: re-read %.3f ms (%.2f TB/s)   fresh region %.3f ms (%.2f TB/s)\n", mb, a, mb / 1024.0 / 1024.0 / a * 1e3 * 1.048576 * 1.048576 ,
             b, mb / 1024.0 / 1024.0 / b * 1e3 * 1.048576 * 1.048576);
    }
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "wgrad")) {
    // weight-gradient ring kernel (5x5): default vs timing ablations, random and zero operands
    const int B = Bbig, T = 301, F = 601;
    const long long n = (long long)B * 64 * T * F;
    float *dz, *in, *part, *dw, *dw2, *scr;
    CK(hipMalloc(&dz, n * 4)); CK(hipMalloc(&in, n * 4));
    CK(hipMalloc(&part, vs_conv64_wgrad_partial_floats(5, 5) * 4)); CK(hipMalloc(&dw, 64 * 64 * 25 * 4)); CK(hipMalloc(&dw2, 64 * 64 * 25 * 4));
    CK(hipMalloc(&scr, 64));
    {
      std::vector<float> h(n);
      for (long long i = 0; i < n; ++i) h[i] = frand() * 2.0f;
      CK(hipMemcpy(dz, h.data(), n * 4, hipMemcpyHostToDevice));
      for (long long i = 0; i < n; ++i) h[i] = frand() * 3.0f;
      CK(hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gflop = 2.0 * 64 * 64 * 25 * (double)B * T * F / 1e9;
    auto timeit = [&](int mode, int dil, const char* what) {
      VS(vs_set_wgrad_kernel(mode));
      CK(hipMemset(scr, 0, 64));
      VS(vs_conv64_wgrad_f16x3(dz, in, part, dw, scr, B, T, F, 5, 5, dil, nullptr));
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r) VS(vs_conv64_wgrad_f16x3(dz, in, part, dw, scr, B, T, F, 5, 5, dil, nullptr));
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("  dil=%-2d %-52s %.3f ms  (%.0f TF algorithmic; includes 2 operand |max| passes ~0.25 ms)\n", dil, what, ms, gflop / ms);
    };
    if (argc > 4) {          // one variant only (for counter passes): tools/conv_bench 64 3 wgrad <mode>
      timeit(atoi(argv[4]), 1, "selected variant");
      return 0;
    }
    printf("wgrad ring kernel 5x5, B=%d (random operands):\n", B);
    for (int dil : {1, 4, 16}) timeit(1, dil, "eight-wave ring kernel (round 1)");
    for (int dil : {1, 4, 16}) timeit(3, dil, "four-wave ring kernel");
    timeit(164, 1, "  - every load out of range (no memory access)");
    timeit(168, 1, "  - loads only + out of range");
    timeit(228, 1, "  - every load from the slab's first 64 KB (L2-hot)");
    timeit(232, 1, "  - loads only + L2-hot");
    timeit(356, 1, "  - all staging pieces behind K block 3");
    timeit(612, 1, "  - same bytes per step as ONE contiguous 28 KB piece (HBM-cold, one page)");
    timeit(616, 1, "  - loads only + contiguous piece");
    timeit(101, 1, "  - fragments read once per step");
    timeit(102, 1, "  - no staging (loads, conversion, LDS writes)");
    timeit(108, 1, "  - no barriers");
    timeit(104, 1, "  - loads issued + waited for, no conversion / LDS writes");
    timeit(116, 1, "  - conversion + LDS writes, no loads");
    timeit(103, 1, "  - no staging, fragments once");
    timeit(111, 1, "  - MFMA stream only");
    CK(hipMemset(in, 0, n * 4));
    printf("same, all-zero input operand:\n");
    timeit(1, 1, "eight-wave ring kernel, zero input");
    timeit(3, 1, "four-wave ring kernel, zero input");
    timeit(111, 1, "  - MFMA stream only, zero input");
    VS(vs_set_wgrad_kernel(0));
    return 0;
  }
  const bool prof = argc > 3 && !strcmp(argv[3], "prof");
  std::vector<Case> cases = {
      {2, 19, 37, 1, 1}, {1, 23, 70, 2, 1}, {2, 21, 133, 4, 0}, {1, 50, 64, 8, 1}, {2, 40, 31, 16, 1}, {1, 20, 37, 16, 2},
      {1, 1, 5, 1, 1}, {3, 9, 601, 2, 1}, {1, 70, 37, 1, 1}, {5, 301, 601, 1, 1}, {3, 301, 601, 16, 1},
      {Bbig, 301, 601, 1, 1}, {Bbig, 301, 601, 2, 1}, {Bbig, 301, 601, 4, 1}, {Bbig, 301, 601, 8, 1}, {Bbig, 301, 601, 16, 1},
      {Bbig, 301, 601, 4, 2},
  };
  if (prof) cases = {{Bbig, 301, 601, 1, 1}, {Bbig, 301, 601, 4, 1}};
  size_t maxn = 0;
  for (auto& c : cases) { size_t n = (size_t)c.B * 64 * c.T * c.F; if (n > maxn) maxn = n; }
  float *in, *out0, *out1, *out2, *w, *scale, *shift, *sc_in, *sc_w;
  void *packed, *amax;
  unsigned long long* cnt; unsigned* firstbad;
  CK(hipMalloc(&in, maxn * 4)); CK(hipMalloc(&out0, maxn * 4)); CK(hipMalloc(&out1, maxn * 4)); CK(hipMalloc(&out2, maxn * 4));
  CK(hipMalloc(&w, 64 * 64 * 25 * 4)); CK(hipMalloc(&scale, 256)); CK(hipMalloc(&shift, 256));
  CK(hipMalloc(&sc_in, 64)); CK(hipMalloc(&sc_w, 64)); CK(hipMalloc(&amax, 64));
  CK(hipMalloc(&packed, vs_conv64_packed_f16_floats(5, 5) * 4));
  CK(hipMalloc(&cnt, 8)); CK(hipMalloc(&firstbad, 4));
  {
    std::vector<float> h(maxn);
    for (size_t i = 0; i < maxn; ++i) h[i] = frand() * 3.0f;
    CK(hipMemcpy(in, h.data(), maxn * 4, hipMemcpyHostToDevice));
    std::vector<float> hw(64 * 64 * 25), hs(64), hb(64);
    for (auto& v : hw) v = frand() * 0.05f;
    for (auto& v : hs) v = 0.5f + 0.5f * (frand() + 1.0f);
    for (auto& v : hb) v = 0.3f * frand();
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(scale, hs.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(shift, hb.data(), 256, hipMemcpyHostToDevice));
  }
  VS(vs_conv64_pack_f16(w, packed, 5, 5, 0, amax, sc_w, nullptr));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int bad_total = 0;
  for (auto& c : cases) {
    const long long n = (long long)c.B * 64 * c.T * c.F;
    VS(vs_pow2_scale(in, n, amax, sc_in, nullptr));
    auto run = [&](int mode, float* out) {
      VS(vs_set_conv_kernel(mode));
      VS(vs_conv64_f16x3_fwd(in, packed, scale, shift, sc_in, sc_w, out, c.B, c.T, c.F, 5, 5, c.dil, c.act, nullptr));
    };
    auto ndiff = [&](const float* a, const float* b) {
      CK(hipMemset(cnt, 0, 8)); CK(hipMemset(firstbad, 0xff, 4));
      hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, nullptr, (const unsigned*)a, (const unsigned*)b, n, cnt, firstbad);
      unsigned long long h; unsigned fb;
      CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&fb, firstbad, 4, hipMemcpyDeviceToHost));
      if (h) printf("    first mismatch at flat index %u (b %lld co %lld t %lld f %lld)\n", fb, fb / (64LL * c.T * c.F), (fb / ((long long)c.T * c.F)) % 64,
                    (fb / c.F) % c.T, (long long)fb % c.F);
      return h;
    };
    CK(hipMemset(out0, 0xcd, n * 4)); CK(hipMemset(out1, 0xab, n * 4)); CK(hipMemset(out2, 0xef, n * 4));
    run(1, out0);
    run(2, out1);
    run(2, out2);
    CK(hipDeviceSynchronize());
    const unsigned long long d01 = ndiff(out0, out1), d12 = ndiff(out1, out2);
    float ms[3] = {0, 0, 0};
    const bool big = n > 50000000LL;
    if (big) {
      for (int mode = 1; mode <= 2; ++mode) {
        run(mode, out2);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) run(mode, out2);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms[mode], e0, e1));
        ms[mode] /= reps;
      }
      // stability of the persistent kernel under back-to-back launches
      const unsigned long long d_again = ndiff(out1, out2);
      if (d_again) { printf("    UNSTABLE: %llu elements differ between launches\n", d_again); ++bad_total; }
    }
    const double gflop = 2.0 * 64 * 64 * 25 * (double)c.B * c.T * c.F / 1e9;
    printf("B=%d T=%d F=%d dil=%d act=%d: legacy-vs-pk mismatches %llu, pk-vs-pk %llu", c.B, c.T, c.F, c.dil, c.act, d01, d12);
    if (big) printf("  | legacy %.3f ms (%.0f TF)  pk %.3f ms (%.0f TF)  x%.3f", ms[1], gflop / ms[1], ms[2], gflop / ms[2], ms[1] / ms[2]);
    printf("\n");
    if (d01 || d12) ++bad_total;
  }
  // ---- timing ablations of the persistent kernel and the data-dependence of the clock -------------
  if (!prof) {
    Case c{Bbig, 301, 601, 1, 1};
    const long long n = (long long)c.B * 64 * c.T * c.F;
    const double gflop = 2.0 * 64 * 64 * 25 * (double)c.B * c.T * c.F / 1e9;
    VS(vs_pow2_scale(in, n, amax, sc_in, nullptr));
    auto timeit = [&](int mode, const char* what) {
      VS(vs_set_conv_kernel(mode));
      VS(vs_conv64_f16x3_fwd(in, packed, scale, shift, sc_in, sc_w, out2, c.B, c.T, c.F, 5, 5, c.dil, c.act, nullptr));
      CK(hipEventRecord(e0));
      for (int r = 0; r < reps; ++r)
        VS(vs_conv64_f16x3_fwd(in, packed, scale, shift, sc_in, sc_w, out2, c.B, c.T, c.F, 5, 5, c.dil, c.act, nullptr));
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
      printf("  %-58s %.3f ms  (%.0f TF algorithmic, %.0f TF on the f16 pipe)\n", what, ms, gflop / ms, 3 * gflop / ms);
    };
    printf("ablations, B=%d dil=1 (random data):\n", c.B);
    timeit(1, "legacy (one tile per workgroup)");
    timeit(2, "persistent pipeline");
    timeit(123, "  pinned issue pattern, 1 MFMA : 3 others");
    timeit(124, "  pinned issue pattern, 1 MFMA : 4 others");
    timeit(125, "  pinned issue pattern, 1 MFMA : 5 others");
    timeit(126, "  pinned issue pattern, 1 MFMA : 6 others");
    timeit(128, "  pinned issue pattern, 1 MFMA : 8 others");
    timeit(101, "  - no fragment ds_reads");
    timeit(102, "  - no staging (window/weight loads, cvt, ds_write)");
    timeit(104, "  - no in-loop epilogue");
    timeit(108, "  - no barriers");
    timeit(106, "  - no staging, no epilogue");
    timeit(107, "  - no staging, no epilogue, no fragment reads");
    timeit(115, "  - MFMA stream only (all of the above + no barriers)");
    CK(hipMemset(in, 0, n * 4));
    printf("same, all-zero input (the guide: +15..21 %% from DVFS on zero operands):\n");
    timeit(1, "legacy, zero input");
    timeit(2, "persistent, zero input");
    timeit(115, "  - MFMA stream only, zero input");
    VS(vs_set_conv_kernel(0));
  }
  printf(bad_total ? "FAILED: %d case(s) differ\n" : "ALL BITWISE EQUAL\n", bad_total);
  return bad_total ? 1 : 0;
}
