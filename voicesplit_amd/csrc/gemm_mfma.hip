// fp32 GEMM on the f32 matrix cores with fused prologue/epilogue, for every dense contraction of
// the path that is not a convolution:
//
//   C[m][n] (+)= act( sum_k opA(A)(m,k) * opW(W)(n,k) + bias1[n] + bias2[n] + rowbias[m/group][n] ) * (gate[m][n] > 0)
//
// Each operand is either K-contiguous ("row" layout: A[m][k], W[n][k] -- activations
// [rows][features] and nn.Linear / nn.LSTM weights [out][in], exactly how the reference stores
// them) or K-major ("col" layout: A[k][m], W[k][n]), so the forward products (NT), the data
// gradients (NN: dX = dY @ W) and the weight gradients (TN: dW = dY^T @ X) all run without a
// transpose pass:
//   forward   * LSTM input projection x @ W_ih^T, both directions   models/voicesplit/model.py:82
//             * d-vector fold dvec @ W_ih[:, 8F:]^T + b_ih + b_hh   models/voicesplit/model.py:77-81
//             * head relu -> fc1 -> relu -> fc2 -> sigmoid          models/voicesplit/model.py:83-87
//   backward  * dX  = dY @ W          (NN)   gate = relu mask of the layer input
//             * dW  = dY^T @ X        (TN)   split-K for the small fc / W_hh outputs
//             * dW_hh = sum_t dgates_t^T h_{t-1}: TN with the X rows shifted by one frame inside
//               each utterance (w_shift / w_group)
//
// Tile 128x128x32, 256 threads = 2x2 waves, each wave 64x64 = 2x2 accumulators of
// v_mfma_f32_32x32x2_f32.  Global -> VGPR (float4, next K tile prefetched during the MFMAs) ->
// LDS.  Row-layout operands sit in LDS as [row][36]: the fragment read (lane = row, 4 consecutive
// k per lane half) is one conflict-free ds_read_b128 feeding four K-steps.  Col-layout operands
// stay K-major in LDS ([k][132]) and are read with ds_read_b32 (lanes = 32 consecutive rows:
// conflict-free); the k -> (K-step, lane half) mapping is the same for both so they can be mixed.
#include "vs_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = BK + 4;    // row layout: 36 floats = 144 B rows, conflict-free b128 reads
constexpr int PITCH_T = BM + 4;  // col layout: 132 floats = 528 B rows (16-B aligned)
constexpr int LDS_FLOATS = BM * PITCH;   // 4608 >= BK * PITCH_T = 4224

struct GemmArgs {
  const float* A; int lda;
  const float* W; int ldw;
  const float* W_hi; int n_split;   // row layout W only: rows n >= n_split come from W_hi
  float* C; int ldc;
  int M, N, K;
  const float* bias1;     // [N] or null
  const float* bias2;     // [N] or null
  const float* rowbias;   // [ceil(M/group)][ldrb] or null
  int ldrb, group;
  const float* gate; int ldg;   // epilogue mask: C = gate[m][n] > 0 ? C : 0
  int a_relu, w_relu, act, accumulate;
  int w_shift, w_group;   // col layout W only: row k is read from row k + w_shift, zero when
                          // (k % w_group) + w_shift falls outside [0, w_group)
  int k_chunk;            // split-K: blockIdx.z covers k in [z*k_chunk, (z+1)*k_chunk)
  long long c_split_stride;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// 8 fp32 values -> the 8 bf16 of one v_mfma_f32_32x32x16_bf16 operand (RNE, v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16x8 to_bf16x8(const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[j];
  return r;
}

// 4 consecutive elements p[i..i+3] of a row of n valid elements, zero beyond
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int i, int n) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (VEC && i + 3 < n) return *reinterpret_cast<const float4*>(p + i);
  if (i + 0 < n) v.x = p[i + 0];
  if (i + 1 < n) v.y = p[i + 1];
  if (i + 2 < n) v.z = p[i + 2];
  if (i + 3 < n) v.w = p[i + 3];
  return v;
}

__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

// BF: VS_MATH_BF16 instances -- same staging, same epilogue, but the fragments are rounded to bf16 on their way out of
// LDS and multiplied by v_mfma_f32_32x32x16_bf16 (k = 16 ks + 8 half + j): 1/32 of the matrix-pipe time of the fp32
// instruction; the head GEMMs of the bf16 configuration then run at the speed of their staging.
template <int LA, int LB, bool VEC, bool BF = false>
__global__ __launch_bounds__(256, 3)
void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float sA[LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float sW[LDS_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kb = blockIdx.z * g.k_chunk;
  const int ke = kb + g.k_chunk < g.K ? kb + g.k_chunk : g.K;
  float* Cz = g.C + (long long)blockIdx.z * g.c_split_stride;

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
      if (LA == 0) {
        const int row = m0 + (v >> 3), c4 = k0 + (v & 7) * 4;
        ra[i] = row < g.M ? load4<VEC>(g.A + (size_t)row * g.lda, c4, ke) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int k = k0 + (v >> 5), c4 = m0 + (v & 31) * 4;
        ra[i] = k < ke ? load4<VEC>(g.A + (size_t)k * g.lda, c4, g.M) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (g.a_relu) ra[i] = relu4(ra[i]);
      if (LB == 0) {
        const int row = n0 + (v >> 3), c4 = k0 + (v & 7) * 4;
        const float* p = row < g.n_split ? g.W + (size_t)row * g.ldw : g.W_hi + (size_t)(row - g.n_split) * g.ldw;
        rw[i] = row < g.N ? load4<VEC>(p, c4, ke) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int k = k0 + (v >> 5), c4 = n0 + (v & 31) * 4;
        bool ok = k < ke;
        if (g.w_group > 0) {
          const int pos = k % g.w_group + g.w_shift;
          ok = ok && pos >= 0 && pos < g.w_group;
        }
        rw[i] = ok ? load4<VEC>(g.W + ((long long)k + g.w_shift) * g.ldw, c4, g.N) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (g.w_relu) rw[i] = relu4(rw[i]);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + 256 * i;
      if (LA == 0) *reinterpret_cast<float4*>(&sA[(v >> 3) * PITCH + (v & 7) * 4]) = ra[i];
      else *reinterpret_cast<float4*>(&sA[(v >> 5) * PITCH_T + (v & 31) * 4]) = ra[i];
      if (LB == 0) *reinterpret_cast<float4*>(&sW[(v >> 3) * PITCH + (v & 7) * 4]) = rw[i];
      else *reinterpret_cast<float4*>(&sW[(v >> 5) * PITCH_T + (v & 31) * 4]) = rw[i];
    }
  };

  // fragment bases: K-step j of k-quad kq multiplies k = 8*kq + 4*half + j in both layouts
  const float* fa = LA == 0 ? sA + (wm * 64 + l31) * PITCH + half * 4 : sA + (half * 4) * PITCH_T + wm * 64 + l31;
  const float* fw = LB == 0 ? sW + (wn * 64 + l31) * PITCH + half * 4 : sW + (half * 4) * PITCH_T + wn * 64 + l31;

  if (kb < ke) gload(kb);
  for (int k0 = kb; k0 < ke; k0 += BK) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (k0 + BK < ke) gload(k0 + BK);
    if constexpr (BF) {
      const float* fa8 = LA == 0 ? sA + (wm * 64 + l31) * PITCH + half * 8 : sA + (half * 8) * PITCH_T + wm * 64 + l31;
      const float* fw8 = LB == 0 ? sW + (wn * 64 + l31) * PITCH + half * 8 : sW + (half * 8) * PITCH_T + wn * 64 + l31;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 a8[2], b8[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          float t[8];
          if (LA == 0) {
            const float4 t0 = *reinterpret_cast<const float4*>(fa8 + x * 32 * PITCH + ks * 16);
            const float4 t1 = *reinterpret_cast<const float4*>(fa8 + x * 32 * PITCH + ks * 16 + 4);
            t[0] = t0.x; t[1] = t0.y; t[2] = t0.z; t[3] = t0.w; t[4] = t1.x; t[5] = t1.y; t[6] = t1.z; t[7] = t1.w;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = fa8[(ks * 16 + j) * PITCH_T + x * 32];
          }
          a8[x] = to_bf16x8(t);
          if (LB == 0) {
            const float4 t0 = *reinterpret_cast<const float4*>(fw8 + x * 32 * PITCH + ks * 16);
            const float4 t1 = *reinterpret_cast<const float4*>(fw8 + x * 32 * PITCH + ks * 16 + 4);
            t[0] = t0.x; t[1] = t0.y; t[2] = t0.z; t[3] = t0.w; t[4] = t1.x; t[5] = t1.y; t[6] = t1.z; t[7] = t1.w;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = fw8[(ks * 16 + j) * PITCH_T + x * 32];
          }
          b8[x] = to_bf16x8(t);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[mb], b8[nb], acc[mb][nb], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int kq = 0; kq < BK / 8; ++kq) {
      float a4[2][4], b4[2][4];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        if (LA == 0) {
          const float4 t = *reinterpret_cast<const float4*>(fa + x * 32 * PITCH + kq * 8);
          a4[x][0] = t.x; a4[x][1] = t.y; a4[x][2] = t.z; a4[x][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) a4[x][j] = fa[(kq * 8 + j) * PITCH_T + x * 32];
        }
        if (LB == 0) {
          const float4 t = *reinterpret_cast<const float4*>(fw + x * 32 * PITCH + kq * 8);
          b4[x][0] = t.x; b4[x][1] = t.y; b4[x][2] = t.z; b4[x][3] = t.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) b4[x][j] = fw[(kq * 8 + j) * PITCH_T + x * 32];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[mb][j], b4[nb][j], acc[mb][nb], 0, 0, 0);
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
          v = vs_act_rt(v, g.act);
          if (g.gate) v = g.gate[(size_t)m * g.ldg + n] > 0.f ? v : 0.f;
          float* c = Cz + (size_t)m * g.ldc + n;
          if (g.accumulate) v += *c;
          *c = v;
        }
      }
    }
  }
}

template <int LA, int LB>
void launch_layout(const GemmArgs& g, bool vec, bool bf, dim3 grid, hipStream_t stream) {
  if (bf) {
    if (vec) hipLaunchKernelGGL((gemm_kernel<LA, LB, true, true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_kernel<LA, LB, false, true>), grid, dim3(256), 0, stream, g);
  } else if (vec) hipLaunchKernelGGL((gemm_kernel<LA, LB, true>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_kernel<LA, LB, false>), grid, dim3(256), 0, stream, g);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__global__ void splitk_reduce_kernel(const float* __restrict__ part, int S, long long n, float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  float s = 0.f;
  for (int q = 0; q < S; ++q) s += part[(long long)q * n + idx];
  out[idx] = s;
}

}  // namespace

// General entry.  layout_a / layout_w: 0 = K contiguous (A[m][k], W[n][k]); 1 = K-major
// (A[k][m], W[k][n]).  splits > 1: split-K through `partials` ([splits][M][N] floats), summed in
// a fixed order into C (which must then be dense, ldc == N, and take no epilogue terms).
static int gemm_general(bool bf, int layout_a, int layout_w, const float* A, int lda, const float* W, const float* W_hi,
                        int n_split, int ldw, float* C, int ldc, int M, int N, int K,
                        const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                        const float* gate, int ldg, int a_relu, int w_relu, int act, int accumulate,
                        int w_shift, int w_group, int splits, float* partials, hipStream_t stream) {
  VS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  VS_REQUIRE((layout_a == 0 || layout_a == 1) && (layout_w == 0 || layout_w == 1), "gemm: bad layout");
  VS_REQUIRE(lda >= (layout_a ? M : K) && ldw >= (layout_w ? N : K) && ldc >= N,
             "gemm: leading dims lda=%d ldw=%d ldc=%d vs M=%d N=%d K=%d", lda, ldw, ldc, M, N, K);
  VS_REQUIRE(!rowbias || (group > 0 && ldrb >= N), "gemm: rowbias needs group>0 and ldrb>=N");
  VS_REQUIRE(!gate || ldg >= N, "gemm: gate needs ldg>=N");
  VS_REQUIRE((M + BM - 1) / BM <= 65535, "gemm: M=%d too large", M);
  VS_REQUIRE(act == VS_ACT_NONE || act == VS_ACT_RELU || act == VS_ACT_SIGMOID, "gemm: unsupported activation %d", act);
  VS_REQUIRE(layout_w == 0 || (W_hi == nullptr || n_split >= N), "gemm: stacked W needs the K-contiguous layout");
  VS_REQUIRE(n_split >= N || W_hi != nullptr, "gemm: W_hi is NULL but n_split=%d < N=%d", n_split, N);
  VS_REQUIRE(w_group == 0 || layout_w == 1, "gemm: w_shift/w_group need the K-major W layout");
  if (splits < 1) splits = 1;
  int k_chunk = K;
  if (splits > 1) {
    VS_REQUIRE(partials != nullptr && ldc == N && !bias1 && !bias2 && !rowbias && !gate && act == VS_ACT_NONE && !accumulate,
               "gemm: split-K needs a partial buffer, a dense C and no epilogue terms");
    k_chunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (K + k_chunk - 1) / k_chunk;
  }
  GemmArgs g{A, lda, W, ldw, W_hi, n_split, splits > 1 ? partials : C, ldc, M, N, K, bias1, bias2, rowbias, ldrb,
             group > 0 ? group : 1, gate, ldg, a_relu, w_relu, act, accumulate, w_shift, w_group, k_chunk,
             (long long)M * N};
  const bool vec = (lda % 4 == 0) && (ldw % 4 == 0) && aligned16(A) && aligned16(W) && aligned16(W_hi);
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, splits);
  if (layout_a == 0 && layout_w == 0) launch_layout<0, 0>(g, vec, bf, grid, stream);
  else if (layout_a == 0 && layout_w == 1) launch_layout<0, 1>(g, vec, bf, grid, stream);
  else if (layout_a == 1 && layout_w == 0) launch_layout<1, 0>(g, vec, bf, grid, stream);
  else launch_layout<1, 1>(g, vec, bf, grid, stream);
  if (splits > 1) {
    const long long n = (long long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, partials, splits, n, C);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_gemm_general_impl(int layout_a, int layout_w, const float* A, int lda, const float* W, const float* W_hi,
                         int n_split, int ldw, float* C, int ldc, int M, int N, int K,
                         const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                         const float* gate, int ldg, int a_relu, int w_relu, int act, int accumulate,
                         int w_shift, int w_group, int splits, float* partials, hipStream_t stream) {
  return gemm_general(false, layout_a, layout_w, A, lda, W, W_hi, n_split, ldw, C, ldc, M, N, K, bias1, bias2, rowbias, ldrb, group,
                      gate, ldg, a_relu, w_relu, act, accumulate, w_shift, w_group, splits, partials, stream);
}

// The same contraction with both operands rounded to bf16 (VS_MATH_BF16: the head of the bf16 configuration)
int vs_gemm_general_bf16_impl(int layout_a, int layout_w, const float* A, int lda, const float* W, const float* W_hi,
                              int n_split, int ldw, float* C, int ldc, int M, int N, int K,
                              const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                              const float* gate, int ldg, int a_relu, int w_relu, int act, int accumulate,
                              int w_shift, int w_group, int splits, float* partials, hipStream_t stream) {
  return gemm_general(true, layout_a, layout_w, A, lda, W, W_hi, n_split, ldw, C, ldc, M, N, K, bias1, bias2, rowbias, ldrb, group,
                      gate, ldg, a_relu, w_relu, act, accumulate, w_shift, w_group, splits, partials, stream);
}

// C = act(opA(A) @ W^T + biases): both operands K-contiguous (the forward products).
int vs_gemm_nt2_impl(const float* A, int lda, const float* W, const float* W_hi, int n_split, int ldw,
                     float* C, int ldc, int M, int N, int K,
                     const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                     int a_relu, int act, hipStream_t stream) {
  return vs_gemm_general_impl(0, 0, A, lda, W, W_hi, n_split, ldw, C, ldc, M, N, K, bias1, bias2, rowbias, ldrb, group,
                              nullptr, 0, a_relu, 0, act, 0, 0, 0, 1, nullptr, stream);
}

// vs_gemm_nt_impl with bf16-rounded operands (VS_MATH_BF16 head)
int vs_gemm_nt_bf16_impl(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                         const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                         int a_relu, int act, hipStream_t stream) {
  return vs_gemm_general_bf16_impl(0, 0, A, lda, W, nullptr, 0x7fffffff, ldw, C, ldc, M, N, K, bias1, bias2, rowbias, ldrb, group,
                                   nullptr, 0, a_relu, 0, act, 0, 0, 0, 1, nullptr, stream);
}

int vs_gemm_nt_impl(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                    const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                    int a_relu, int act, hipStream_t stream) {
  return vs_gemm_nt2_impl(A, lda, W, nullptr, 0x7fffffff, ldw, C, ldc, M, N, K, bias1, bias2, rowbias, ldrb, group,
                          a_relu, act, stream);
}
