// fp32 "NT" GEMM on the f32 matrix cores with fused prologue/epilogue:
//
//   C[m][n] = act( sum_k opA(A[m][k]) * W[n][k] + bias1[n] + bias2[n] + rowbias[m / group][n] )
//
// Both operands are K-contiguous (activations [rows][features], nn.Linear / nn.LSTM weights
// [out][in]), which is exactly how the reference stores them, so no transposes are needed.
// Used for
//   * the LSTM input projection x @ W_ih^T for both directions (models/voicesplit/model.py:82),
//     with the d-vector columns of W_ih folded into a per-utterance row bias
//     (models/voicesplit/model.py:77-81: repeat+cat of the speaker embedding);
//   * the d-vector fold itself  dvec @ W_ih[:, 8F:]^T + b_ih + b_hh;
//   * the head: relu -> fc1 -> relu -> fc2 -> sigmoid (models/voicesplit/model.py:83-87).
//
// Tile 128x128x32, 256 threads = 2x2 waves, each wave 64x64 = 2x2 accumulators of
// v_mfma_f32_32x32x2_f32.  Global -> VGPR (float4, next K tile prefetched during the MFMAs)
// -> LDS rows padded to 36 floats so the ds_read_b128 fragment reads (lane = row, 4 consecutive
// k per lane half) are bank-conflict free; one b128 read feeds four K-steps.
#include "vs_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = BK + 4;   // 36 floats = 144 B: 16-B aligned rows, conflict-free b128 reads

struct GemmArgs {
  const float* A; int lda;
  const float* W; int ldw;
  const float* W_hi; int n_split;   // rows n >= n_split of the weight come from W_hi (two stacked matrices)
  float* C; int ldc;
  int M, N, K;
  const float* bias1;     // [N] or null
  const float* bias2;     // [N] or null
  const float* rowbias;   // [ceil(M/group)][ldrb] or null
  int ldrb, group;
};

template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ base, int ld, int row, int nrows, int k, int K,
                                        const float* __restrict__ base_hi = nullptr, int split = 0x7fffffff) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < nrows) {
    const float* p = (row < split ? base + (size_t)row * ld : base_hi + (size_t)(row - split) * ld) + k;
    if (VEC) {
      if (k < K) v = *reinterpret_cast<const float4*>(p);
    } else {
      if (k + 0 < K) v.x = p[0];
      if (k + 1 < K) v.y = p[1];
      if (k + 2 < K) v.z = p[2];
      if (k + 3 < K) v.w = p[3];
    }
  }
  return v;
}

template <bool VEC, bool A_RELU, int ACT>
__global__ __launch_bounds__(256, 2)
void gemm_nt_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float sA[BM * PITCH];
  __shared__ __attribute__((aligned(16))) float sW[BN * PITCH];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[4], rw[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + 256 * i;
      const int row = v >> 3, c4 = (v & 7) * 4;
      ra[i] = load4<VEC>(g.A, g.lda, m0 + row, g.M, k0 + c4, g.K);
      rw[i] = load4<VEC>(g.W, g.ldw, n0 + row, g.N, k0 + c4, g.K, g.W_hi, g.n_split);
      if (A_RELU) {
        ra[i].x = fmaxf(ra[i].x, 0.f); ra[i].y = fmaxf(ra[i].y, 0.f);
        ra[i].z = fmaxf(ra[i].z, 0.f); ra[i].w = fmaxf(ra[i].w, 0.f);
      }
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + 256 * i;
      const int row = v >> 3, c4 = (v & 7) * 4;
      *reinterpret_cast<float4*>(&sA[row * PITCH + c4]) = ra[i];
      *reinterpret_cast<float4*>(&sW[row * PITCH + c4]) = rw[i];
    }
  };

  const float* fa = sA + (wm * 64 + l31) * PITCH + half * 4;
  const float* fw = sW + (wn * 64 + l31) * PITCH + half * 4;

  gload(0);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (k0 + BK < g.K) gload(k0 + BK);
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      float4 a4[2], b4[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        a4[x] = *reinterpret_cast<const float4*>(fa + x * 32 * PITCH + kq * 8);
        b4[x] = *reinterpret_cast<const float4*>(fw + x * 32 * PITCH + kq * 8);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const float av = j == 0 ? a4[mb].x : j == 1 ? a4[mb].y : j == 2 ? a4[mb].z : a4[mb].w;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const float bv = j == 0 ? b4[nb].x : j == 1 ? b4[nb].y : j == 2 ? b4[nb].z : b4[nb].w;
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mb][nb], 0, 0, 0);
          }
        }
      }
    }
  }

  // epilogue. D: col = lane&31 -> n, row = (r&3)+8*(r>>2)+4*(lane>>5) -> m
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = n0 + wn * 64 + nb * 32 + l31;
    if (n >= g.N) continue;
    float bcol = 0.f;
    if (g.bias1) bcol += g.bias1[n];
    if (g.bias2) bcol += g.bias2[n];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < g.M) {
          float v = acc[mb][nb][r] + bcol;
          if (g.rowbias) v += g.rowbias[(size_t)(m / g.group) * g.ldrb + n];
          g.C[(size_t)m * g.ldc + n] = vs_act<ACT>(v);
        }
      }
    }
  }
}

template <bool VEC, bool A_RELU>
int launch_act(const GemmArgs& g, int act, hipStream_t stream) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM), block(256);
  switch (act) {
    case VS_ACT_NONE: hipLaunchKernelGGL((gemm_nt_kernel<VEC, A_RELU, VS_ACT_NONE>), grid, block, 0, stream, g); break;
    case VS_ACT_RELU: hipLaunchKernelGGL((gemm_nt_kernel<VEC, A_RELU, VS_ACT_RELU>), grid, block, 0, stream, g); break;
    case VS_ACT_SIGMOID: hipLaunchKernelGGL((gemm_nt_kernel<VEC, A_RELU, VS_ACT_SIGMOID>), grid, block, 0, stream, g); break;
    default: VS_REQUIRE(false, "gemm: unsupported activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int vs_gemm_nt2_impl(const float* A, int lda, const float* W, const float* W_hi, int n_split, int ldw,
                     float* C, int ldc, int M, int N, int K,
                     const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                     int a_relu, int act, hipStream_t stream);

int vs_gemm_nt_impl(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                    const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                    int a_relu, int act, hipStream_t stream) {
  return vs_gemm_nt2_impl(A, lda, W, nullptr, 0x7fffffff, ldw, C, ldc, M, N, K, bias1, bias2, rowbias, ldrb, group,
                          a_relu, act, stream);
}

// Same GEMM with the weight given as two stacked row blocks (rows [0,n_split) from W, the rest
// from W_hi, same leading dimension): the forward and reverse W_ih of the BiLSTM in one launch.
int vs_gemm_nt2_impl(const float* A, int lda, const float* W, const float* W_hi, int n_split, int ldw,
                     float* C, int ldc, int M, int N, int K,
                     const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                     int a_relu, int act, hipStream_t stream) {
  VS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  VS_REQUIRE(lda >= K && ldw >= K && ldc >= N, "gemm: leading dims lda=%d ldw=%d ldc=%d vs K=%d N=%d", lda, ldw, ldc, K, N);
  VS_REQUIRE(!rowbias || (group > 0 && ldrb >= N), "gemm: rowbias needs group>0 and ldrb>=N");
  VS_REQUIRE((M + BM - 1) / BM <= 65535, "gemm: M=%d too large", M);
  VS_REQUIRE(n_split >= N || W_hi != nullptr, "gemm: W_hi is NULL but n_split=%d < N=%d", n_split, N);
  GemmArgs g{A, lda, W, ldw, W_hi, n_split, C, ldc, M, N, K, bias1, bias2, rowbias, ldrb, group > 0 ? group : 1};
  const bool vec = (K % 4 == 0) && (lda % 4 == 0) && (ldw % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(W_hi) & 15) == 0);
  if (vec) return a_relu ? launch_act<true, true>(g, act, stream) : launch_act<true, false>(g, act, stream);
  return a_relu ? launch_act<false, true>(g, act, stream) : launch_act<false, false>(g, act, stream);
}
