// Small streaming kernels of the backward pass: the sigmoid gradient of the mask head and the
// column sums that produce bias gradients (models/voicesplit/model.py:83-87 backwards).
#include "vs_common.h"

namespace {

// dlogits = dmask * mask * (1 - mask)            (mask = sigmoid(logits), model.py:87)
__global__ __launch_bounds__(256)
void sigmoid_bwd_kernel(const float* __restrict__ dmask, const float* __restrict__ mask, float* __restrict__ dlogits, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float m = mask[i];
    dlogits[i] = dmask[i] * m * (1.f - m);
  }
}

// the same, and a bf16 row copy of dlogits [rows][Kp] (zero padded) beside it: the A operand of the head's first backward contraction
// on gemm_bf16.hip (vs_backward, VS_MATH_BF16) without a conversion pass; a thread = 2 consecutive columns of a row (lanes side by side)
__global__ __launch_bounds__(256)
void sigmoid_bwd_rows_kernel(const float* __restrict__ dmask, const float* __restrict__ mask, float* __restrict__ dlogits, long long rows, int N,
                             unsigned* __restrict__ rows_bf16, int Kp) {
  const int pairs = Kp >> 1;
  const long long total = rows * pairs;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / pairs;
    const int k = (int)(i - r * pairs) * 2;
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e)
      if (k + e < N) {
        const float m = mask[r * N + k + e];
        v[e] = dmask[r * N + k + e] * m * (1.f - m);
        dlogits[r * N + k + e] = v[e];
      }
    rows_bf16[i] = vs_pack_bf16(v[0], v[1]);
  }
}

// out[g][n] = sum_{r < rows} X[(g*rows + r)*ld + n]: one thread per column, rows walked serially
// (coalesced across the 256 columns of a block).  Used twice for a full column sum: per
// utterance (g = b, rows = T), then over the utterances.
__global__ __launch_bounds__(256)
void colsum_kernel(const float* __restrict__ x, int ld, int rows, int N, float* __restrict__ out, int ldo) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float* p = x + (size_t)blockIdx.y * rows * ld + n;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 3 < rows; r += 4) {
    s0 += p[(size_t)r * ld];
    s1 += p[(size_t)(r + 1) * ld];
    s2 += p[(size_t)(r + 2) * ld];
    s3 += p[(size_t)(r + 3) * ld];
  }
  for (; r < rows; ++r) s0 += p[(size_t)r * ld];
  out[(size_t)blockIdx.y * ldo + n] = (s0 + s1) + (s2 + s3);
}

}  // namespace

int vs_sigmoid_bwd_impl(const float* dmask, const float* mask, float* dlogits, long long n, hipStream_t stream) {
  VS_REQUIRE(n > 0, "sigmoid_bwd: n=%lld", n);
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, stream, dmask, mask, dlogits, n);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_sigmoid_bwd_rows_impl(const float* dmask, const float* mask, float* dlogits, long long rows, int N, void* rows_bf16, int Kp,
                             hipStream_t stream) {
  VS_REQUIRE(rows > 0 && N > 0 && rows_bf16 && Kp >= N && Kp % 8 == 0 && (reinterpret_cast<uintptr_t>(rows_bf16) & 15) == 0,
             "sigmoid_bwd_rows: bad argument");
  const long long nb = (rows * (Kp >> 1) + 255) / 256;
  hipLaunchKernelGGL(sigmoid_bwd_rows_kernel, dim3((unsigned)(nb < 16384 ? nb : 16384)), dim3(256), 0, stream, dmask, mask, dlogits, rows, N,
                     reinterpret_cast<unsigned*>(rows_bf16), Kp);
  VS_LAUNCH_CHECK();
  return 0;
}

// out[g][0..N) = column sums of the g-th block of `rows` rows of X [groups*rows][ld]
int vs_colsum_impl(const float* x, int ld, int groups, int rows, int N, float* out, int ldo, hipStream_t stream) {
  VS_REQUIRE(groups > 0 && rows > 0 && N > 0 && ld >= N && ldo >= N && groups <= 65535,
             "colsum: bad shape groups=%d rows=%d N=%d ld=%d", groups, rows, N, ld);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, groups), dim3(256), 0, stream, x, ld, rows, N, out, ldo);
  VS_LAUNCH_CHECK();
  return 0;
}
