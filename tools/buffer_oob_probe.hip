// Is the SGPR offset of a raw buffer access part of the range check on gfx950?  conv_nhwc.hip's epilogue predicates whole
// rows by handing an out-of-range soffset to buffer_store / buffer_load: if the check ignored soffset those accesses
// would land 2 GiB past the tensor.  One wave: loads and stores with (voffset in range, soffset = 0x7FFFFFF0),
// (voffset = 0x7FFFFFF0, soffset small) and both in range, on a buffer surrounded by canaries.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/buffer_oob_probe tools/buffer_oob_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__global__ void probe(unsigned* buf /* 4096 words: [1024, 3072) is the buffer */, unsigned* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf + 1024, 0, 2048 * 4, 0x00020000);
  const unsigned lane = threadIdx.x, kOob = 0x7FFFFFF0u;
  const u2v a = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, 0, 0));            // in range
  const u2v b = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, kOob, 0));         // soffset out
  const u2v c = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(r, kOob, 64, 0));               // voffset out
  const u2v d = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, 2048 * 4 - 256, 0)); // sum out for lanes >= 32
  out[lane * 8 + 0] = a[0]; out[lane * 8 + 1] = a[1]; out[lane * 8 + 2] = b[0]; out[lane * 8 + 3] = b[1];
  out[lane * 8 + 4] = c[0]; out[lane * 8 + 5] = c[1]; out[lane * 8 + 6] = d[0]; out[lane * 8 + 7] = d[1];
  const u2v s = {0xdead0000u + lane, 0xbeef0000u + lane};
  __builtin_amdgcn_raw_buffer_store_b64(s, r, lane * 8, kOob, 0);               // must vanish
  __builtin_amdgcn_raw_buffer_store_b64(s, r, kOob, 64, 0);                     // must vanish
  __builtin_amdgcn_raw_buffer_store_b64(s, r, lane * 8, 2048 * 4 - 256, 0);     // lanes < 32 land in the last 256 bytes; the rest vanish
}

int main() {
  unsigned *buf, *out, h[4096], o[512];
  CK(hipMalloc(&buf, sizeof h));
  CK(hipMalloc(&out, sizeof o));
  for (int i = 0; i < 4096; ++i) h[i] = 0x11110000u + i;
  CK(hipMemcpy(buf, h, sizeof h, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
  unsigned g[4096];
  CK(hipMemcpy(g, buf, sizeof g, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    if (o[l * 8] != h[1024 + 2 * l]) { ++bad; printf("in-range load lane %d wrong\n", l); }
    if (o[l * 8 + 2] != 0 || o[l * 8 + 3] != 0) { ++bad; if (l < 2) printf("soffset-out load lane %d returned %08x (not 0): soffset is NOT range checked\n", l, o[l * 8 + 2]); }
    if (o[l * 8 + 4] != 0) { ++bad; if (l < 2) printf("voffset-out load lane %d returned %08x\n", l, o[l * 8 + 4]); }
    const unsigned want = l < 32 ? h[1024 + 2048 - 64 + 2 * l] : 0;
    if (o[l * 8 + 6] != want) { ++bad; if (l < 34) printf("sum-out load lane %d: %08x want %08x\n", l, o[l * 8 + 6], want); }
  }
  for (int i = 0; i < 4096; ++i) {
    unsigned want = h[i];
    const int k = i - (1024 + 2048 - 64);
    if (k >= 0 && k < 64) want = (k & 1 ? 0xbeef0000u : 0xdead0000u) + k / 2;
    if (g[i] != want) { ++bad; if (bad < 12) printf("word %d: %08x want %08x\n", i, g[i], want); }
  }
  printf(bad ? "FAIL: %d mismatches\n" : "OK: soffset and voffset are both range checked, loads return 0, stores vanish (%d mismatches)\n", bad);
  return bad != 0;
}
