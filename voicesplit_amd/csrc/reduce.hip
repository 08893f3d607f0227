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

// out[g][0..N) = column sums of the g-th block of `rows` rows of X [groups*rows][ld]
int vs_colsum_impl(const float* x, int ld, int groups, int rows, int N, float* out, int ldo, hipStream_t stream) {
  VS_REQUIRE(groups > 0 && rows > 0 && N > 0 && ld >= N && ldo >= N && groups <= 65535,
             "colsum: bad shape groups=%d rows=%d N=%d ld=%d", groups, rows, N, ld);
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, groups), dim3(256), 0, stream, x, ld, rows, N, out, ldo);
  VS_LAUNCH_CHECK();
  return 0;
}
