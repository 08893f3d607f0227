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
// operands arrive as dwords with lanes along the row index (coalesced per k row), each thread
// holding the two floats of a k-pair, and are written with ds_write_b32 (2-way conflict = free):
// the transpose costs no extra pass.  Every global access is a buffer load whose offset is pushed
// out of range where the row / k index leaves the matrix (hardware returns 0): one 32-bit lane
// offset, no bounds branches, all loads of a tile in flight together.  (Flat loads behind
// per-access `if (row < M)` / `if (k < K)` tests compiled to one exec-masked basic block per load,
// 50-80 more VGPRs, and 1.3-1.6x the time.)  Workgroups are renumbered so that each XCD (workgroup
// id % 8) owns a contiguous range of the (m, n) tile list.
// Measured at B=64 (rocprofv3, same box before -> after the branch-free loads): 19264x3200x4808 NT
// 3.48 -> 2.60 ms, 19264x4808x1600 NN 1.43 -> 1.34 ms, 1600x4808x19264 TN 1.54 -> 1.44 ms.
// Counters of the NT case before the change: MFMA pipe 25 % busy, waves waiting 59 % of their
// time, L2 hit rate 83 %, 2.5 GB fetched over the fabric per launch (1.5 TB/s) - neither HBM nor L2
// bound; a 64-deep K block, 2 vs 3 workgroups per CU and the XCD renumbering each moved it < 3 %.
// What was left was the fp32 -> f16 hi/lo conversion in the staging path (~280 VALU per wave and K
// block against 24 MFMAs = 768 pipe cycles).  The forward input GEMM now takes the pre-split form
// at the bottom of this file (split pass + gemm_pre_kernel: 1.95 ms, MFMA pipe 46 % busy, 88M
// instead of 697M VALU instructions per launch); the two backward contractions stay here: they
// would need transposed split copies of dxg and feat (~0.45 ms) to save ~1 ms.
#include <stdlib.h>

#include "vs_internal.h"

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
  int band;               // tile rows per band of the workgroup -> tile map (gemm_tile_of)
};

// BF (VS_MATH_BF16): one bf16 rounding per element in the hi image, nothing in the lo image; the kernels
// then issue only the hi x hi product on v_mfma_f32_32x32x16_bf16
template <bool BF = false>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  if (BF) { hi = vs_pack_bf16(x0, x1); lo = 0u; return; }
  const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  const h2 l = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]));
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

template <bool BF, class HV>
__device__ __forceinline__ f32x16 g_mma(HV a, HV b, f32x16 c, int, int, int) {
  if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vs_bf16x8, a), __builtin_bit_cast(vs_bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// One operand's view for the loads: K-contiguous operands may be two stacked matrices (rows
// < split from `lo`, the rest from `hi`: W_ih of the two LSTM directions); a tile lying on one side
// uses one descriptor, a straddling tile reads both (out-of-range lanes return 0) and adds.
struct OperandView {
  __amdgpu_buffer_rsrc_t lo, hi;
  int row_lo, rows_lo;    // first row of this tile inside `lo`, rows of `lo`
  int row_hi, rows_hi;    // same for `hi` (row_hi may be negative in a straddling tile)
  int use_lo, use_hi;
};

constexpr unsigned kOob = 0xFFFFFFF0u;    // beyond any descriptor (operands are < 4 GiB)

// Workgroup id -> (m, n) tile.  XCD x (= workgroup id % 8, the dispatcher's round robin) owns the contiguous range
// [x*per, (x+1)*per) of a BANDED tile list: bands of `band` tile rows (8; 1 and 1024 measured the same in round 3, 0 = the row-major list), inside a band the m index runs fastest.
// The ~64 workgroups an XCD has resident at a time then cover an 8 x 8 block of tiles: per K step they share 8 + 8
// operand panels in that XCD's L2, where the row-major list (n fastest, 25 tiles per row in the LSTM input GEMM)
// made them span 2.6 tile rows = 3 + 25 panels and re-stream all of W from the Infinity Cache for every row panel
// (r02 counters: 9.87 GB fetched per launch against 0.68 GB of operands).
__device__ __forceinline__ bool gemm_tile_of(int tiles_m, int tiles_n, int band_rows, int& tm, int& tn) {
  const int per = (int)(gridDim.x >> 3);
  const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (t >= tiles_m * tiles_n) return false;
  if (band_rows <= 0) { tm = t / tiles_n; tn = t - tm * tiles_n; return true; }   // row-major list (round 2), kept for A/B timing
  const int band_tiles = band_rows * tiles_n;
  const int band = t / band_tiles;
  const int r = t - band * band_tiles;
  const int rows = min(band_rows, tiles_m - band * band_rows);   // the last band may be shorter
  tn = r / rows;
  tm = band * band_rows + (r - tn * rows);
  return true;
}

// One operand tile [128 rows][BK k] : global -> registers (RT floats per thread).  Every access
// is a buffer load whose offset is pushed out of range when the row or k index is outside the
// matrix (hardware returns 0): no branches, so all loads of a tile are in flight together.
template <int LAYOUT, bool VEC, int BK>
__device__ __forceinline__ void tile_load(float (&r)[regs_of(BK)], const OperandView& o, int ld, int row0_kmajor, int k0, int K, int tid) {
  constexpr int RT = regs_of(BK);
  typedef float f4 __attribute__((ext_vector_type(4)));
  if (LAYOUT == 0) {
#pragma unroll
    for (int i = 0; i < RT / 4; ++i) {
      const int v = tid + 256 * i;
      const int lr = v / (BK / 4), c4 = k0 + (v % (BK / 4)) * 4;
      f4 t = {0.f, 0.f, 0.f, 0.f};
      auto fetch = [&](const __amdgpu_buffer_rsrc_t rs, int row, int nrows) {
        const bool okr = row >= 0 && row < nrows;
        const unsigned base = ((unsigned)row * (unsigned)ld + (unsigned)c4) * 4u;
        if (VEC) {      // K % 4 == 0: a float4 is entirely inside or entirely outside the row
          return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (okr && c4 < K) ? base : kOob, 0, 0));
        } else {
          f4 q;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            q[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (okr && c4 + e < K) ? base + 4u * e : kOob, 0, 0));
          return q;
        }
      };
      if (o.use_lo) t = fetch(o.lo, o.row_lo + lr, o.rows_lo);
      if (o.use_hi) t += fetch(o.hi, o.row_hi + lr, o.rows_hi);
      r[4 * i + 0] = t[0]; r[4 * i + 1] = t[1]; r[4 * i + 2] = t[2]; r[4 * i + 3] = t[3];
    }
  } else {
    // item = (row m = idx & 127, k-pair kp = idx >> 7): lanes along m, two dword loads (k, k+1)
    // through a descriptor over the whole operand: one 32-bit lane offset, the K block and pair
    // index in the scalar offset, rows k >= K out of range (-> 0).  Columns m >= nrows read the
    // neighbouring row: they only reach output rows / columns that are never stored.
    const unsigned voff = ((unsigned)(2 * (tid >> 7)) * (unsigned)ld + (unsigned)(tid & 127)) * 4u;
    const unsigned so0 = ((unsigned)k0 * (unsigned)ld + (unsigned)row0_kmajor) * 4u;
#pragma unroll
    for (int i = 0; i < RT / 2; ++i) {
      const unsigned so = so0 + (unsigned)(4 * i) * (unsigned)ld * 4u;
      r[2 * i + 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(o.lo, voff, so, 0));
      r[2 * i + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(o.lo, voff, so + (unsigned)ld * 4u, 0));
    }
  }
}

// registers -> f16 hi/lo halves in LDS [row][PW dwords]
template <int LAYOUT, int BK, bool BF = false>
__device__ __forceinline__ void tile_store(const float (&r)[regs_of(BK)], unsigned* __restrict__ sh, unsigned* __restrict__ sl,
                                           float s, bool relu, int tid) {
  constexpr int RT = regs_of(BK), PW = pitch_of(BK);
  const float floor_ = relu ? 0.f : -__builtin_inff();      // max(x, -inf) = x: one v_max either way
  if (LAYOUT == 0) {
#pragma unroll
    for (int i = 0; i < RT / 4; ++i) {
      const int v = tid + 256 * i;
      float x[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = fmaxf(r[4 * i + j], floor_) * s;
      unsigned h0, l0, h1, l1;
      split_pair<BF>(x[0], x[1], h0, l0);
      split_pair<BF>(x[2], x[3], h1, l1);
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
      const float x0 = fmaxf(r[2 * i], floor_) * s;
      const float x1 = fmaxf(r[2 * i + 1], floor_) * s;
      unsigned hi, lo;
      split_pair<BF>(x0, x1, hi, lo);
      sh[(idx & 127) * PW + (idx >> 7)] = hi;
      sl[(idx & 127) * PW + (idx >> 7)] = lo;
    }
  }
}

template <int LA, int LB, bool VEC, int BK, int OCC, bool BF = false>
__global__ __launch_bounds__(256, OCC)
void gemm_f16x3_kernel(Gemm16Args g) {
  constexpr int RT = regs_of(BK), PW = pitch_of(BK);
  __shared__ __attribute__((aligned(16))) unsigned sAh[BM * PW], sAl[BM * PW], sWh[BN * PW], sWl[BN * PW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  int tile_m, tile_n;
  if (!gemm_tile_of(g.tiles_m, g.tiles_n, g.band, tile_m, tile_n)) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
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
  // descriptors: K-major operands span (K-1)*ld + rows floats, K-contiguous ones (rows-1)*ld + K
  auto desc = [](const float* p, size_t floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000);
  };
  OperandView va, vw;
  va.lo = va.hi = desc(g.A, LA ? (size_t)(g.K - 1) * g.lda + g.M : (size_t)(g.M - 1) * g.lda + g.K);
  va.row_lo = m0; va.rows_lo = g.M; va.row_hi = 0; va.rows_hi = 0; va.use_lo = 1; va.use_hi = 0;
  const int nlo = g.n_split < g.N ? g.n_split : g.N;          // rows of W held by g.W
  vw.lo = desc(g.W, LB ? (size_t)(g.K - 1) * g.ldw + g.N : (size_t)(nlo - 1) * g.ldw + g.K);
  vw.hi = (!LB && g.N > nlo) ? desc(g.W_hi, (size_t)(g.N - nlo - 1) * g.ldw + g.K) : vw.lo;
  vw.row_lo = n0; vw.rows_lo = nlo; vw.row_hi = n0 - nlo; vw.rows_hi = g.N - nlo;
  vw.use_lo = LB || n0 < nlo; vw.use_hi = !LB && n0 + BN > nlo && g.N > nlo;

  // fragment of block x, k-step ks: 8 halves = dwords [row*PW + 8*ks + 4*half .. +3]
  const int fa = (wm * 64 + l31) * PW + 4 * half;
  const int fw = (wn * 64 + l31) * PW + 4 * half;

  tile_load<LA, VEC, BK>(ra, va, g.lda, m0, 0, g.K, tid);
  tile_load<LB, VEC, BK>(rw, vw, g.ldw, n0, 0, g.K, tid);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    __syncthreads();
    tile_store<LA, BK, BF>(ra, sAh, sAl, sa, g.a_relu != 0, tid);
    tile_store<LB, BK, BF>(rw, sWh, sWl, sw, g.w_relu != 0, tid);
    __syncthreads();
    if (k0 + BK < g.K) {
      tile_load<LA, VEC, BK>(ra, va, g.lda, m0, k0 + BK, g.K, tid);
      tile_load<LB, VEC, BK>(rw, vw, g.ldw, n0, k0 + BK, g.K, tid);
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
      for (int term = BF ? 2 : 0; term < 3; ++term)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = g_mma<BF>(term == 0 ? al[mb] : ah[mb], term == 1 ? bl[nb] : bh[nb],
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
void launch_layout(const Gemm16Args& g, bool vec, dim3 grid, hipStream_t stream, bool bf) {
  if (bf) {
    if (vec) hipLaunchKernelGGL((gemm_f16x3_kernel<LA, LB, true, 32, 3, true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_f16x3_kernel<LA, LB, false, 32, 3, true>), grid, dim3(256), 0, stream, g);
    return;
  }
  if (vec) hipLaunchKernelGGL((gemm_f16x3_kernel<LA, LB, true, 32, 3>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_f16x3_kernel<LA, LB, false, 32, 3>), grid, dim3(256), 0, stream, g);
}

// ---------------------------------------------------------------------------------------------
// Pre-split operands.  In the kernel above every operand tile is converted fp32 -> f16 hi/lo while it
// is staged, i.e. once per tile that uses it: an A row-panel 25 times (N tiles) and the weights 151
// times (M tiles) in the LSTM input GEMM, ~280 VALU instructions per wave and K block against 24
// MFMAs.  Here the split is a pass of its own (split_rows_kernel: x*s -> hi, lo as two f16 arrays
// [rows][Kp], Kp = K rounded up to the K block, zero padded) and the GEMM streams halves:
// global b128 -> register -> ds_write_b128, no conversion, and a lane's fragment (8 consecutive k)
// is ONE ds_read_b128 (row pitch 80 B: 5 x 16 B, conflict-free for b128 with lane = row).
// ---------------------------------------------------------------------------------------------
constexpr int PBK = VS_GEMM_KPAD;       // K block in halves: 64 (one barrier pair per 48 MFMAs of a wave; 32 measured 10 % slower)
constexpr int PPW = PBK / 2 + 4;        // LDS row pitch in dwords (64 halves + 8 pad = 144 B: conflict-free b128 reads, lane = row)
constexpr int PCPR = PBK / 8;           // 16-byte chunks per row of a tile
constexpr int PRPP = 256 / PCPR;        // rows one pass of the workgroup stages
constexpr int PNP = 128 / PRPP;         // passes per 128-row tile

// src [rows][ld] fp32 (K valid columns) -> hi, lo [rows][Kp] f16 of src * scale[0]
__global__ __launch_bounds__(256)
void split_rows_kernel(const float* __restrict__ src, int rows, int K, int ld, const float* __restrict__ scale,
                       _Float16* __restrict__ hi, _Float16* __restrict__ lo, int Kp, int relu, int bf) {
  const float s = scale[0];
  const int groups = Kp >> 3;                                   // 8 halves = 16 B per thread
  const long long total = (long long)rows * groups;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / groups;
    const int k0 = (int)(i - r * groups) * 8;
    const float* p = src + r * ld + k0;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = (k0 + e < K) ? p[e] : 0.f;
      if (relu) v = fmaxf(v, 0.f);
      x[e] = v * s;
    }
    u4v h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned hq, lq;
      if (bf) split_pair<true>(x[2 * q], x[2 * q + 1], hq, lq);
      else split_pair<false>(x[2 * q], x[2 * q + 1], hq, lq);
      h[q] = hq;
      l[q] = lq;
    }
    *reinterpret_cast<u4v*>(hi + r * Kp + k0) = h;
    *reinterpret_cast<u4v*>(lo + r * Kp + k0) = l;
  }
}

struct GemmPreArgs {
  const _Float16* Ah; const _Float16* Al;     // [M][Kp]
  const _Float16* Wh; const _Float16* Wl;     // [N][Kp]
  float* C; int ldc;
  int M, N, Kp;
  const float* bias1; const float* bias2;
  const float* rowbias; int ldrb, group;
  int act, accumulate;
  const float* a_scale; const float* w_scale;
  int tiles_m, tiles_n;
  int band;
};

// C (+)= act((Ah+Al)(Wh+Wl)^T / (sA*sW) + bias terms): 128x128x64 tile, 4 waves (2x2) of 64x64
template <bool BF>
__global__ __launch_bounds__(256, 2)
void gemm_pre_kernel(GemmPreArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned sAh[BM * PPW], sAl[BM * PPW], sWh[BN * PPW], sWl[BN * PPW];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  int tile_m, tile_n;
  if (!gemm_tile_of(g.tiles_m, g.tiles_n, g.band, tile_m, tile_n)) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const float inv = g.a_scale[1] * g.w_scale[1];

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging: thread -> (row = tid>>2 (+64), 16-byte chunk q = tid&3) of each of the four arrays;
  // rows beyond the matrix are pushed out of the descriptor's range (-> 0)
  auto desc = [](const _Float16* p, size_t halves) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p), 0, (int)(halves * 2), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rAh = desc(g.Ah, (size_t)g.M * g.Kp), rAl = desc(g.Al, (size_t)g.M * g.Kp);
  const __amdgpu_buffer_rsrc_t rWh = desc(g.Wh, (size_t)g.N * g.Kp), rWl = desc(g.Wl, (size_t)g.N * g.Kp);
  const int srow = tid / PCPR, sq = tid % PCPR;                  // (row, 16-byte chunk) staged by this thread, per pass
  unsigned offA[PNP], offW[PNP];
#pragma unroll
  for (int i = 0; i < PNP; ++i) {
    const int row = srow + PRPP * i;
    offA[i] = (m0 + row < g.M) ? (unsigned)(((size_t)(m0 + row) * g.Kp + 8 * sq) * 2) : kOob;
    offW[i] = (n0 + row < g.N) ? (unsigned)(((size_t)(n0 + row) * g.Kp + 8 * sq) * 2) : kOob;
  }
  u4v ra[PNP][2], rw[PNP][2];      // [pass][hi, lo]
  auto tile_load = [&](int k0) {
    const unsigned so = (unsigned)k0 * 2u;
#pragma unroll
    for (int i = 0; i < PNP; ++i) {
      ra[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rAh, offA[i], so, 0);
      ra[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rAl, offA[i], so, 0);
      rw[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rWh, offW[i], so, 0);
      rw[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rWl, offW[i], so, 0);
    }
  };
  const int st = srow * PPW + 4 * sq;                           // dword index of this thread's chunk, pass 0
  auto tile_store = [&]() {
#pragma unroll
    for (int i = 0; i < PNP; ++i) {
      *reinterpret_cast<u4v*>(&sAh[st + PRPP * i * PPW]) = ra[i][0];
      *reinterpret_cast<u4v*>(&sAl[st + PRPP * i * PPW]) = ra[i][1];
      *reinterpret_cast<u4v*>(&sWh[st + PRPP * i * PPW]) = rw[i][0];
      *reinterpret_cast<u4v*>(&sWl[st + PRPP * i * PPW]) = rw[i][1];
    }
  };
  // fragment of 32-row block x, k-step ks: dwords [row*PPW + 8*ks + 4*half .. +3]
  const int fa = (wm * 64 + l31) * PPW + 4 * half;
  const int fw = (wn * 64 + l31) * PPW + 4 * half;

  tile_load(0);
  for (int k0 = 0; k0 < g.Kp; k0 += PBK) {
    __syncthreads();
    tile_store();
    __syncthreads();
    if (k0 + PBK < g.Kp) tile_load(k0 + PBK);
#pragma unroll
    for (int ks = 0; ks < PBK / 16; ++ks) {
      h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int oa = fa + x * 32 * PPW + 8 * ks, ow = fw + x * 32 * PPW + 8 * ks;
        ah[x] = __builtin_bit_cast(h8, *reinterpret_cast<const u4v*>(&sAh[oa]));
        al[x] = __builtin_bit_cast(h8, *reinterpret_cast<const u4v*>(&sAl[oa]));
        bh[x] = __builtin_bit_cast(h8, *reinterpret_cast<const u4v*>(&sWh[ow]));
        bl[x] = __builtin_bit_cast(h8, *reinterpret_cast<const u4v*>(&sWl[ow]));
      }
#pragma unroll
      for (int term = BF ? 2 : 0; term < 3; ++term)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mb][nb] = g_mma<BF>(term == 0 ? al[mb] : ah[mb], term == 1 ? bl[nb] : bh[nb],
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
          float* c = g.C + (size_t)m * g.ldc + n;
          if (g.accumulate) v += *c;
          *c = v;
        }
      }
    }
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int gemm_band() { return 8; }      // tile rows per band of the raster (1 / 8 / 1024 measured the same in round 3: profiles/r03_gemm_l2_prefetch.md)

}  // namespace

int vs_pow2_scale_impl(const float* x, long long n, unsigned* amax_scratch, float* scale2, hipStream_t stream);

// Same contract as vs_gemm_general_impl (no split-K, no w_shift).  a_scale2 / w_scale2: {s, 1/s}
// of the two operands (see vs_pow2_scale_impl); the caller derives them once per tensor.
int vs_gemm_f16x3_impl(int layout_a, int layout_w, const float* A, int lda, const float* W, const float* W_hi,
                       int n_split, int ldw, float* C, int ldc, int M, int N, int K,
                       const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                       const float* gate, int ldg, int a_relu, int w_relu, int act, int accumulate,
                       const float* a_scale2, const float* w_scale2, hipStream_t stream, int math) {
  const bool bf = math == VS_MATH_CODE_BF16;
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
               gate, ldg, a_relu, w_relu, act, accumulate, a_scale2, w_scale2, (M + BM - 1) / BM, (N + BN - 1) / BN, gemm_band()};
  const bool vec = (lda % 4 == 0) && (ldw % 4 == 0) && (K % 4 == 0) && aligned16(A) && aligned16(W) && aligned16(W_hi);
  VS_REQUIRE(layout_a || ((size_t)(M - 1) * lda + K) * 4 < (1ull << 32) - 64, "gemm_f16x3: A above 4 GiB");
  VS_REQUIRE(layout_w || ((size_t)(N - 1) * ldw + K) * 4 < (1ull << 32) - 64, "gemm_f16x3: W above 4 GiB");
  VS_REQUIRE((long long)g.tiles_m * g.tiles_n < (1LL << 30), "gemm_f16x3: too many tiles");
  dim3 grid((unsigned)((g.tiles_m * g.tiles_n + 7) / 8 * 8));
  if (layout_a == 0 && layout_w == 0) launch_layout<0, 0>(g, vec, grid, stream, bf);
  else if (layout_a == 0 && layout_w == 1) launch_layout<0, 1>(g, vec, grid, stream, bf);
  else if (layout_a == 1 && layout_w == 0) launch_layout<1, 0>(g, vec, grid, stream, bf);
  else launch_layout<1, 1>(g, vec, grid, stream, bf);
  VS_LAUNCH_CHECK();
  return 0;
}

// bytes of scratch the pre-split form of C[M][N] = A[M][K] W[N][K]^T needs (both operands, hi + lo)
size_t vs_gemm_presplit_bytes(int M, int N, int K) {
  const size_t Kp = (size_t)(K + PBK - 1) / PBK * PBK;
  return ((size_t)M + (size_t)N) * Kp * 2 * sizeof(_Float16) + 1024;
}

// x [rows][ld] (K valid columns) * scale2[0] -> hi, lo [rows][Kp]
int vs_split_rows_impl(const float* x, int rows, int K, int ld, const float* scale2, _Float16* hi, _Float16* lo, int relu,
                       hipStream_t stream, int math) {
  VS_REQUIRE(rows > 0 && K > 0 && ld >= K, "split_rows: bad shape rows=%d K=%d ld=%d", rows, K, ld);
  const int Kp = (K + PBK - 1) / PBK * PBK;
  const long long total = (long long)rows * (Kp >> 3);
  const long long nb = (total + 255) / 256;
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, stream, x, rows, K, ld, scale2, hi, lo, Kp, relu,
                     math == VS_MATH_CODE_BF16 ? 1 : 0);
  VS_LAUNCH_CHECK();
  return 0;
}

// C[M][N] (+)= act(A W^T + bias terms) over operands already split by vs_split_rows_impl (Kp = K
// rounded up to VS_GEMM_KPAD; scales as used for the split)
int vs_gemm_presplit_impl(const _Float16* Ah, const _Float16* Al, const _Float16* Wh, const _Float16* Wl, int Kp,
                          float* C, int ldc, int M, int N, const float* bias1, const float* bias2,
                          const float* rowbias, int ldrb, int group, int act, int accumulate,
                          const float* a_scale2, const float* w_scale2, hipStream_t stream, int math) {
  VS_REQUIRE(M > 0 && N > 0 && Kp > 0 && Kp % PBK == 0 && ldc >= N, "gemm_presplit: bad shape M=%d N=%d Kp=%d ldc=%d", M, N, Kp, ldc);
  VS_REQUIRE(((size_t)M * Kp) * 2 < (1ull << 32) - 64 && ((size_t)N * Kp) * 2 < (1ull << 32) - 64, "gemm_presplit: operand above 4 GiB");
  VS_REQUIRE(!rowbias || (group > 0 && ldrb >= N), "gemm_presplit: rowbias needs group>0 and ldrb>=N");
  VS_REQUIRE(aligned16(Ah) && aligned16(Al) && aligned16(Wh) && aligned16(Wl), "gemm_presplit: operands must be 16-byte aligned");
  GemmPreArgs g{Ah, Al, Wh, Wl, C, ldc, M, N, Kp, bias1, bias2, rowbias, ldrb, group > 0 ? group : 1, act, accumulate,
                a_scale2, w_scale2, (M + BM - 1) / BM, (N + BN - 1) / BN, gemm_band()};
  if (math == VS_MATH_CODE_BF16) hipLaunchKernelGGL(gemm_pre_kernel<true>, dim3((unsigned)((g.tiles_m * g.tiles_n + 7) / 8 * 8)), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL(gemm_pre_kernel<false>, dim3((unsigned)((g.tiles_m * g.tiles_n + 7) / 8 * 8)), dim3(256), 0, stream, g);
  VS_LAUNCH_CHECK();
  return 0;
}
