// The same GEMM as gemm_mfma.hip in split-f16 arithmetic (VS_MATH_F16X3; scheme in conv_f16x3.hip):
//   C (+)= act( (A*sA) x (W*sW) / (sA*sW) + bias terms ) * gate,     x = three f16 MFMA products
// for the three large LSTM contractions of the path (models/voicesplit/model.py:82 and its
// backward): x @ W_ih^T (NT), dxg @ W_ih (NN), dxg^T @ feat (TN).
//
// Tile 128x128x32, 4 waves (2x2), each wave 64x64 = 2x2 accumulators of v_mfma_f32_32x32x16_f16,
// three workgroups per CU.  Both operands are converted to f16 hi/lo halves while they are staged
// and sit in LDS as [row][18 dwords] (16 k-pairs + 2 pad): a lane's fragment (8 consecutive k of its
// row) is two ds_read_b64, conflict-free with lane = row (18*row mod 64 hits 32 distinct even banks).
// K-contiguous operands arrive as float4 along k and are written with ds_write_b64; K-major
// operands arrive as dwords with lanes along the row index (coalesced per k row) through a buffer
// descriptor (one 32-bit lane offset, everything else scalar; flat loads with a 64-bit address and
// a bounds branch per dword cost 50-80 more VGPRs and 1.6x the time), each thread holding the two
// floats of a k-pair, and are written with ds_write_b32 (2-way conflict = free): the transpose
// costs no extra pass.  Workgroups are renumbered so that each XCD (workgroup id % 8) owns a
// contiguous range of the (m, n) tile list.
// Measured at B=64 (rocprofv3): 19264x3200x4808 NT 3.4 ms (MFMA pipe 25 % busy, waves stalled on
// memory 59 % of the time), 19264x4808x1600 NN 1.15 ms, 1600x4808x19264 TN 1.19 ms.  A 64-deep K
// block, 2 vs 3 workgroups per CU and the XCD renumbering all leave the NT case within 3 %: it
// moves 18.6 GB of operand tiles in 3.4 ms (5.5 TB/s if few of them hit in L2), which points at
// operand re-reads rather than at the pipeline - a reading of the counters above, not yet
// confirmed by a FETCH_SIZE pass; a larger tile is the next thing to try.
#include "vs_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128;
// per K-block depth BK: LDS row pitch in dwords, floats of one operand tile per thread
constexpr int pitch_of(int BK) { return BK / 2 + 2; }
constexpr int regs_of(int BK) { return BM * BK / 256; }

struct Gemm16Args {
  const float* A; int lda;
  const float* W; int ldw;
  const float* W_hi; int n_split;   // row layout W only
  float* C; int ldc;
  int M, N, K;
  const float* bias1;
  const float* bias2;
  const float* rowbias;
  int ldrb, group;
  const float* gate; int ldg;
  int a_relu, w_relu, act, accumulate;
  const float* a_scale;   // {s, 1/s}
  const float* w_scale;   // {s, 1/s}
  int tiles_m, tiles_n;
};

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  const h2 l = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]));
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

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

// One operand tile [128 rows][BK k] : global -> registers (RT floats per thread)
template <int LAYOUT, bool VEC, int BK>
__device__ __forceinline__ void tile_load(float (&r)[regs_of(BK)], const __amdgpu_buffer_rsrc_t rsrc, const float* __restrict__ base, const float* __restrict__ base_hi,
                                          int split, int ld, int row0, int nrows, int k0, int K, int tid) {
  constexpr int RT = regs_of(BK);
  if (LAYOUT == 0) {
#pragma unroll
    for (int i = 0; i < RT / 4; ++i) {
      const int v = tid + 256 * i;
      const int row = row0 + v / (BK / 4), c4 = k0 + (v % (BK / 4)) * 4;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) {
        const float* p = row < split ? base + (size_t)row * ld : base_hi + (size_t)(row - split) * ld;
        t = load4<VEC>(p, c4, K);
      }
      r[4 * i + 0] = t.x; r[4 * i + 1] = t.y; r[4 * i + 2] = t.z; r[4 * i + 3] = t.w;
    }
  } else {
    // item = (row m = idx & 127, k-pair kp = idx >> 7): lanes along m, two dword loads (k, k+1)
    // through a buffer descriptor over the whole operand: one 32-bit lane offset, the K block and
    // pair index in the scalar offset, rows k >= K out of range (-> 0).  Columns m >= nrows read
    // the neighbouring row: they only reach output rows / columns that are never stored.
    const unsigned voff = ((unsigned)(2 * (tid >> 7)) * (unsigned)ld + (unsigned)(tid & 127)) * 4u;
    const unsigned so0 = ((unsigned)k0 * (unsigned)ld + (unsigned)row0) * 4u;
#pragma unroll
    for (int i = 0; i < RT / 2; ++i) {
      const unsigned so = so0 + (unsigned)(4 * i) * (unsigned)ld * 4u;
      r[2 * i + 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, so, 0));
      r[2 * i + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, so + (unsigned)ld * 4u, 0));
    }
  }
}

// registers -> f16 hi/lo halves in LDS [row][PW dwords]
template <int LAYOUT, int BK>
__device__ __forceinline__ void tile_store(const float (&r)[regs_of(BK)], unsigned* __restrict__ sh, unsigned* __restrict__ sl,
                                           float s, bool relu, int tid) {
  constexpr int RT = regs_of(BK), PW = pitch_of(BK);
  if (LAYOUT == 0) {
#pragma unroll
    for (int i = 0; i < RT / 4; ++i) {
      const int v = tid + 256 * i;
      float x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = (relu ? fmaxf(r[4 * i + j], 0.f) : r[4 * i + j]) * s;
      unsigned h0, l0, h1, l1;
      split_pair(x[0], x[1], h0, l0);
      split_pair(x[2], x[3], h1, l1);
      u2v hi, lo;
      hi[0] = h0; hi[1] = h1;
      lo[0] = l0; lo[1] = l1;
      *reinterpret_cast<u2v*>(&sh[(v / (BK / 4)) * PW + (v % (BK / 4)) * 2]) = hi;
      *reinterpret_cast<u2v*>(&sl[(v / (BK / 4)) * PW + (v % (BK / 4)) * 2]) = lo;
    }
  } else {
#pragma unroll
    for (int i = 0; i < RT / 2; ++i) {
      const int idx = tid + 256 * i;
      const float x0 = (relu ? fmaxf(r[2 * i], 0.f) : r[2 * i]) * s;
      const float x1 = (relu ? fmaxf(r[2 * i + 1], 0.f) : r[2 * i + 1]) * s;
      unsigned hi, lo;
      split_pair(x0, x1, hi, lo);
      sh[(idx & 127) * PW + (idx >> 7)] = hi;
      sl[(idx & 127) * PW + (idx >> 7)] = lo;
    }
  }
}

template <int LA, int LB, bool VEC, int BK, int OCC>
__global__ __launch_bounds__(256, OCC)
void gemm_f16x3_kernel(Gemm16Args g) {
  constexpr int RT = regs_of(BK), PW = pitch_of(BK);
  __shared__ __attribute__((aligned(16))) unsigned sAh[BM * PW], sAl[BM * PW], sWh[BN * PW], sWl[BN * PW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  // XCD x takes tiles [x*per, (x+1)*per) of the row-major (m, n) tile list
  const int per = gridDim.x >> 3;
  const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tile >= g.tiles_m * g.tiles_n) return;
  const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
  const float sa = g.a_scale[0], sw = g.w_scale[0];
  const float inv = g.a_scale[1] * g.w_scale[1];

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float ra[RT], rw[RT];
  // K-major operands are read through buffer descriptors ((K-1)*ld + rows floats)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, LA ? (int)(((size_t)(g.K - 1) * g.lda + g.M) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.W), 0, LB ? (int)(((size_t)(g.K - 1) * g.ldw + g.N) * 4) : 0, 0x00020000);
  // fragment of block x, k-step ks: 8 halves = dwords [row*PW + 8*ks + 4*half .. +3]
  const int fa = (wm * 64 + l31) * PW + 4 * half;
  const int fw = (wn * 64 + l31) * PW + 4 * half;

  tile_load<LA, VEC, BK>(ra, rsA, g.A, nullptr, 0x7fffffff, g.lda, m0, g.M, 0, g.K, tid);
  tile_load<LB, VEC, BK>(rw, rsW, g.W, g.W_hi, g.n_split, g.ldw, n0, g.N, 0, g.K, tid);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    __syncthreads();
    tile_store<LA, BK>(ra, sAh, sAl, sa, g.a_relu != 0, tid);
    tile_store<LB, BK>(rw, sWh, sWl, sw, g.w_relu != 0, tid);
    __syncthreads();
    if (k0 + BK < g.K) {
      tile_load<LA, VEC, BK>(ra, rsA, g.A, nullptr, 0x7fffffff, g.lda, m0, g.M, k0 + BK, g.K, tid);
      tile_load<LB, VEC, BK>(rw, rsW, g.W, g.W_hi, g.n_split, g.ldw, n0, g.N, k0 + BK, g.K, tid);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int oa = fa + x * 32 * PW + 8 * ks, ow = fw + x * 32 * PW + 8 * ks;
        u4v t;
        u2v p0 = *reinterpret_cast<const u2v*>(&sAh[oa]), p1 = *reinterpret_cast<const u2v*>(&sAh[oa + 2]);
        t[0] = p0[0]; t[1] = p0[1]; t[2] = p1[0]; t[3] = p1[1];
        ah[x] = __builtin_bit_cast(h8, t);
        p0 = *reinterpret_cast<const u2v*>(&sAl[oa]); p1 = *reinterpret_cast<const u2v*>(&sAl[oa + 2]);
        t[0] = p0[0]; t[1] = p0[1]; t[2] = p1[0]; t[3] = p1[1];
        al[x] = __builtin_bit_cast(h8, t);
        p0 = *reinterpret_cast<const u2v*>(&sWh[ow]); p1 = *reinterpret_cast<const u2v*>(&sWh[ow + 2]);
        t[0] = p0[0]; t[1] = p0[1]; t[2] = p1[0]; t[3] = p1[1];
        bh[x] = __builtin_bit_cast(h8, t);
        p0 = *reinterpret_cast<const u2v*>(&sWl[ow]); p1 = *reinterpret_cast<const u2v*>(&sWl[ow + 2]);
        t[0] = p0[0]; t[1] = p0[1]; t[2] = p1[0]; t[3] = p1[1];
        bl[x] = __builtin_bit_cast(h8, t);
      }
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al[mb] : ah[mb], term == 1 ? bl[nb] : bh[nb],
                                                                 acc[mb][nb], 0, 0, 0);
    }
  }

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
          float v = fmaf(acc[mb][nb][r], inv, bcol);
          if (g.rowbias) v += g.rowbias[(size_t)(m / g.group) * g.ldrb + n];
          v = vs_act_rt(v, g.act);
          if (g.gate) v = g.gate[(size_t)m * g.ldg + n] > 0.f ? v : 0.f;
          float* c = g.C + (size_t)m * g.ldc + n;
          if (g.accumulate) v += *c;
          *c = v;
        }
      }
    }
  }
}

template <int LA, int LB>
void launch_layout(const Gemm16Args& g, bool vec, dim3 grid, hipStream_t stream) {
  if (vec) hipLaunchKernelGGL((gemm_f16x3_kernel<LA, LB, true, 32, 3>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_f16x3_kernel<LA, LB, false, 32, 3>), grid, dim3(256), 0, stream, g);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int vs_pow2_scale_impl(const float* x, long long n, unsigned* amax_scratch, float* scale2, hipStream_t stream);

// Same contract as vs_gemm_general_impl (no split-K, no w_shift).  a_scale2 / w_scale2: {s, 1/s}
// of the two operands (see vs_pow2_scale_impl); the caller derives them once per tensor.
int vs_gemm_f16x3_impl(int layout_a, int layout_w, const float* A, int lda, const float* W, const float* W_hi,
                       int n_split, int ldw, float* C, int ldc, int M, int N, int K,
                       const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                       const float* gate, int ldg, int a_relu, int w_relu, int act, int accumulate,
                       const float* a_scale2, const float* w_scale2, hipStream_t stream) {
  VS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_f16x3: bad shape M=%d N=%d K=%d", M, N, K);
  VS_REQUIRE((layout_a == 0 || layout_a == 1) && (layout_w == 0 || layout_w == 1), "gemm_f16x3: bad layout");
  VS_REQUIRE(lda >= (layout_a ? M : K) && ldw >= (layout_w ? N : K) && ldc >= N,
             "gemm_f16x3: leading dims lda=%d ldw=%d ldc=%d vs M=%d N=%d K=%d", lda, ldw, ldc, M, N, K);
  VS_REQUIRE(!rowbias || (group > 0 && ldrb >= N), "gemm_f16x3: rowbias needs group>0 and ldrb>=N");
  VS_REQUIRE(!gate || ldg >= N, "gemm_f16x3: gate needs ldg>=N");
  VS_REQUIRE(act == VS_ACT_NONE || act == VS_ACT_RELU || act == VS_ACT_SIGMOID, "gemm_f16x3: unsupported activation %d", act);
  VS_REQUIRE(layout_w == 0 || (W_hi == nullptr || n_split >= N), "gemm_f16x3: stacked W needs the K-contiguous layout");
  VS_REQUIRE(n_split >= N || W_hi != nullptr, "gemm_f16x3: W_hi is NULL but n_split=%d < N=%d", n_split, N);
  VS_REQUIRE(a_scale2 && w_scale2, "gemm_f16x3: NULL scale");
  VS_REQUIRE(!layout_a || ((size_t)(K - 1) * lda + M) * 4 < (1ull << 32), "gemm_f16x3: K-major A above 4 GiB");
  VS_REQUIRE(!layout_w || ((size_t)(K - 1) * ldw + N) * 4 < (1ull << 32), "gemm_f16x3: K-major W above 4 GiB");
  Gemm16Args g{A, lda, W, ldw, W_hi, n_split, C, ldc, M, N, K, bias1, bias2, rowbias, ldrb, group > 0 ? group : 1,
               gate, ldg, a_relu, w_relu, act, accumulate, a_scale2, w_scale2, (M + BM - 1) / BM, (N + BN - 1) / BN};
  const bool vec = (lda % 4 == 0) && (ldw % 4 == 0) && aligned16(A) && aligned16(W) && aligned16(W_hi);
  VS_REQUIRE((long long)g.tiles_m * g.tiles_n < (1LL << 30), "gemm_f16x3: too many tiles");
  dim3 grid((unsigned)((g.tiles_m * g.tiles_n + 7) / 8 * 8));
  if (layout_a == 0 && layout_w == 0) launch_layout<0, 0>(g, vec, grid, stream);
  else if (layout_a == 0 && layout_w == 1) launch_layout<0, 1>(g, vec, grid, stream);
  else if (layout_a == 1 && layout_w == 0) launch_layout<1, 0>(g, vec, grid, stream);
  else launch_layout<1, 1>(g, vec, grid, stream);
  VS_LAUNCH_CHECK();
  return 0;
}
