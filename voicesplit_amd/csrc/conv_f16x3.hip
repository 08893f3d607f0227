// 64->64 channel dilated "same" convolution (forward of cnn2..cnn7 and, with transposed/flipped
// weights, their data gradient) on the f16 matrix cores with fp32-class accuracy.
//
// Same tiling and dilation decomposition as conv_mfma.hip (tile = 64 co x R rows x 32 f, one
// residue class of t at a time, window staged through LDS, weights pre-packed in fragment order);
// what changes is the arithmetic.  gfx950 has no TF32-like path: its f32 MFMA runs at the f32
// vector rate, 1/16 of the f16/bf16 rate.  So every fp32 operand is split into two halves,
//     x * s = hi + lo,   hi = f16(x*s) (round toward zero),   lo = f16(x*s - hi)     (exact remainder)
// with s a per-tensor power of two that puts max|x| at ~2^10 (f16 then holds 22 significant bits
// of every element that matters; anything below 2^-24 of the tensor's maximum is dropped, like an
// fp32 accumulator would drop it), and the product is rebuilt from three f16 MFMAs accumulated in
// fp32:  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi   (the dropped a_lo*b_lo is 2^-22 relative).
// v_mfma_f32_32x32x16_f16 does 16 K per 32 cycles where v_mfma_f32_32x32x2_f32 does 2 per 64, so
// the same contraction costs 3/16 of the matrix-pipe time: the layer's ceiling moves from 157
// to ~780 algorithmic TFLOP/s at unchanged accuracy class (parity tests: same tolerances).
//
// Data path per workgroup (256 threads, 4 waves; wave w owns rows w*P..w*P+P-1, both co blocks):
//  * ci is walked in 4 chunks of 16.  Staging: thread -> pixel of the [R+KT-1][32+KF-1] window, 16
//    raw buffer loads (out-of-range -> 0 = ZeroPad2d), scale, split, four ds_write_b128 into the
//    pixel-major LDS window [pixel][slot 4][8 x f16]: slot = 2*part + (ci>>3), part 0 = hi, 1 = lo.
//    The 16-byte slot index is XOR-swizzled with (pixel>>2)&3 so that the b128 fragment reads of
//    32 consecutive pixels (one slot each) are bank-conflict free.
//  * B fragment (K = 16 ci, N = 32 f): lane (f = lane&31, half = lane>>5) needs ci 8*half..8*half+7
//    of its pixel: ONE ds_read_b128 per part.
//  * A fragment: weights packed as [chunk][tap][co block][part][lane][8 x f16] -- already the LDS
//    image.  One kt row of taps (20 KB; all 7 taps for the 7x1 layer) is copied global -> VGPR ->
//    LDS by the whole workgroup into one of two buffers while the previous row is multiplied,
//    then each wave reads its fragments with ds_read_b128.  (Per-wave fragment loads straight
//    from L2, as in the fp32 kernel, would need 42 B/clk/CU at this MFMA rate: first version,
//    7.0 ms per layer against a 2.9 ms matrix-pipe floor.)
//  * Epilogue: y = act(acc * (1/(s_in*s_w)) * scale[co] + shift[co]).
#include "vs_internal.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCo = 64;
constexpr int kCi = 64;
constexpr int kChunk = 16;              // input channels per LDS stage = K of one MFMA
constexpr int kNChunk = kCi / kChunk;   // 4
constexpr unsigned kOob = 0x7FFFFFF0u;
constexpr int kTileF = 32;
constexpr int kPadTaps = 2;             // dummy (zero) tap blocks behind the packed weights
constexpr int kTapBytes = 2 * 2 * 64 * 16;   // one (chunk, tap): 2 co blocks x 2 parts x 64 lanes x 16 B

// ---- per-tensor power-of-two scale --------------------------------------------------------------
// absmax over a tensor as an atomicMax on the bit pattern (non-negative floats order like uints)
__global__ __launch_bounds__(256)
void absmax_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
  float m = 0.f;
  const long long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = x4[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// scale[0] = s = 2^(10 - e) with max = f * 2^e, f in [0.5,1);  scale[1] = 1/s.  0 / inf / nan -> 1.
// amax: n uints (1 from absmax_kernel, VS_AMAX_SLOTS from vs_absmax_commit producers); one wave.
__global__ void scale_from_absmax_kernel(const unsigned* __restrict__ amax, int n, float* __restrict__ scale) {
  unsigned mb = 0;
  for (int i = threadIdx.x; i < n; i += 64) mb = amax[i] > mb ? amax[i] : mb;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned other = __shfl_down(mb, o, 64); mb = other > mb ? other : mb; }
  if (threadIdx.x != 0) return;
  const float m = __uint_as_float(mb);
  int e = 0;
  float s = 1.f, inv = 1.f;
  if (m > 0.f && m < 3.0e38f) {
    (void)frexpf(m, &e);
    int k = 10 - e;
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    s = ldexpf(1.f, k);
    inv = ldexpf(1.f, -k);
  }
  scale[0] = s;
  scale[1] = inv;
}

__device__ __forceinline__ void split2(float x0, float x1, f16x2& hi, f16x2& lo) {
  hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0 - (float)hi[0], x1 - (float)hi[1]));
}

// packed weight layout: [chunk][tap][cb(2)][part(2)][lane(64)][8 x f16]; element j of the lane vector:
//   part(W[co = cb*32 + (lane&31)][ci = chunk*16 + 8*(lane>>5) + j][kt][kf] * s_w)
// transpose_flip = 1: the data-gradient weights W'[co'][ci'][kt][kf] = W[ci'][co'][KT-1-kt][KF-1-kf]
__global__ void conv_pack_weights_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int KT, int KF,
                                             int transpose_flip, const float* __restrict__ wscale, int bf16) {
  const int NT = KT * KF;
  const int total = kNChunk * NT * 2 * 2 * 64 * 8;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total + kPadTaps * (kTapBytes / 2)) return;
  if (idx >= total) { wp[idx] = (_Float16)0.f; return; }
  const int j = idx & 7;
  const int lane = (idx >> 3) & 63;
  const int part = (idx >> 9) & 1;
  const int cb = (idx >> 10) & 1;
  const int tg = idx >> 11;           // chunk*NT + tap
  const int chunk = tg / NT, tap = tg - chunk * NT;
  const int kt = tap / KF, kf = tap - kt * KF;
  const int co = cb * 32 + (lane & 31);
  const int ci = chunk * kChunk + 8 * (lane >> 5) + j;
  const float v = (transpose_flip ? w[((ci * kCi + co) * KT + (KT - 1 - kt)) * KF + (KF - 1 - kf)]
                                  : w[((co * kCi + ci) * KT + kt) * KF + kf]) * wscale[0];
  if (bf16) {      // VS_MATH_BF16: the hi slot carries the bf16 rounding of the weight, the lo slot is unused
    const unsigned short b = (unsigned short)(vs_pack_bf16(v, 0.f) & 0xffffu);
    wp[idx] = part ? (_Float16)0.f : __builtin_bit_cast(_Float16, b);
    return;
  }
  f16x2 hi, lo;
  split2(v, 0.f, hi, lo);
  wp[idx] = part ? lo[0] : hi[0];
}

template <int KT, int KF, int P, int ACT, int NTERM = 3, bool STATS = false>
__global__ __launch_bounds__(256, 2)
void conv64_f16x3_kernel(const float* __restrict__ in, const _Float16* __restrict__ wp,
                         const float* __restrict__ scale, const float* __restrict__ shift,
                         const float* __restrict__ in_scale, const float* __restrict__ w_scale,
                         float* __restrict__ out, int T, int F, int dil, int n_rt, int n_ft, unsigned* amax_out,
                         int i_base, int i_end,       // this launch covers rows [i_base, i_end) of every residue class
                         double* bn_stats = nullptr) {   // STATS: [VS_BN_STAT_SLOTS][64][2] partial {sum, sum of squares} of the output
  constexpr int R = 4 * P;
  constexpr int ROWS = R + KT - 1;
  constexpr int PX = kTileF + KF - 1 + ((kTileF + KF - 1) % 4 ? 4 - (kTileF + KF - 1) % 4 : 0);   // multiple of 4
  constexpr int NPIX = ROWS * PX;
  constexpr int NPP = (NPIX + 255) / 256;
  constexpr int NT = KT * KF;
  constexpr int GT = KF == 1 ? KT : KF;          // taps per weight group (one kt row; all taps for 7x1)
  constexpr int NGRP = NT / GT;                  // weight groups per ci chunk
  constexpr int kTapVec = kTapBytes / 16;        // 256 x 16 B per tap
  // activations [pixel][4 slots][8 x f16] = 64 B per pixel (+ overrun rows for the one-tap-ahead
  // prefetch); weights: two buffers of GT taps in fragment order
  __shared__ __attribute__((aligned(16))) u32x4 sIn[(NPIX + (P + 1) * PX) * 4];
  constexpr int NWBUF = NGRP > 1 ? 2 : 1;        // one group per chunk: the chunk barrier already separates reuse
  __shared__ __attribute__((aligned(16))) u32x4 sW[NWBUF][GT * kTapVec];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int half = lane >> 5;

  int bid = blockIdx.x;
  const int ft = bid % n_ft; bid /= n_ft;
  const int rt = bid % n_rt; bid /= n_rt;
  const int cls = bid % dil;
  const int b = bid / dil;
  const int n_all = (T - cls + dil - 1) / dil;
  const int n_c = n_all < i_end ? n_all : i_end;          // rows of the class this launch may write
  const int i0 = i_base + rt * R;
  if (i0 >= n_c) return;
  const int f0 = ft * kTileF;
  const size_t plane = (size_t)T * F;
  const float* in_b = in + (size_t)b * kCi * plane;
  const unsigned plane_bytes = (unsigned)(plane * sizeof(float));
  const unsigned slab_bytes = plane_bytes * kChunk;
  const float s_in = in_scale[0];
  const float inv = in_scale[1] * w_scale[1];

  unsigned voff[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int pix = tid + 256 * i;
    const int rr = pix / PX;
    const int x = pix - rr * PX;
    const int iin = i0 - KT / 2 + rr;
    const int f = f0 - KF / 2 + x;
    const bool ok = (pix < NPIX) && (iin >= 0) && (iin < n_all) && (f >= 0) && (f < F);
    voff[i] = ok ? (unsigned)(((cls + dil * iin) * F + f) * 4) : kOob;
  }
  float stage[NPP][kChunk];
  auto load_chunk = [&](int chunk) {
    const float* src = in_b + (size_t)chunk * kChunk * plane;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, slab_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPP; ++i)
#pragma unroll
      for (int c = 0; c < kChunk; ++c)
        stage[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[i], c * plane_bytes, 0));
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int pix = tid + 256 * i;
      if (pix < NPIX) {
        const int sw = ((pix % PX) >> 2) & 3;     // swizzle by window column (a row is 9*64 banks: no effect)
#pragma unroll
        for (int h = 0; h < 2; ++h) {        // ci 8h..8h+7
          if (NTERM == 1) {                  // VS_MATH_BF16: one bf16 rounding per element, no lo half
            u32x4 vb;
#pragma unroll
            for (int q = 0; q < 4; ++q) vb[q] = vs_pack_bf16(stage[i][8 * h + 2 * q] * s_in, stage[i][8 * h + 2 * q + 1] * s_in);
            sIn[pix * 4 + ((0 + h) ^ sw)] = vb;
            continue;
          }
          f16x2 hi[4], lo[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) split2(stage[i][8 * h + 2 * q] * s_in, stage[i][8 * h + 2 * q + 1] * s_in, hi[q], lo[q]);
          u32x4 vh, vl;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            vh[q] = __builtin_bit_cast(unsigned, hi[q]);
            vl[q] = __builtin_bit_cast(unsigned, lo[q]);
          }
          sIn[pix * 4 + ((0 + h) ^ sw)] = vh;
          sIn[pix * 4 + ((2 + h) ^ sw)] = vl;
        }
      }
    }
  };

  // weights: the packed array is already the LDS image (fragment order), group gg = GT consecutive
  // taps = GT*256 16-byte vectors, copied global -> VGPR -> LDS by all 256 threads
  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(wp), 0, (unsigned)((kNChunk * NT + kPadTaps) * kTapBytes), 0x00020000);
  u32x4 wreg[GT];
  auto load_w = [&](int gg) {
#pragma unroll
    for (int i = 0; i < GT; ++i)
      wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)((tid + 256 * i) * 16), gg * (GT * kTapBytes), 0);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < GT; ++i) sW[buf][tid + 256 * i] = wreg[i];
  };

  f32x16 acc[2][P];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][p][r] = 0.f;

  const bool wave_active = (i0 + wave * P) < n_c;
  // A fragments of tap g of the group in sW[buf]: [cb][part]
  auto load_a = [&](int buf, int g, u32x4 (&a)[2][2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int part = 0; part < (NTERM == 1 ? 1 : 2); ++part) a[cb][part] = sW[buf][g * kTapVec + (cb * 2 + part) * 64 + lane];
  };
  // B fragments of one tap: [row p][part]; pixel = (wave*P + p + kt)*PX + l31 + kf.  The lane-
  // dependent part of the (swizzled) address is precomputed per (kf, part); the row part is a
  // compile-time immediate, so a fragment read costs no VALU.
  const u32x4* bbase[KF][2];
#pragma unroll
  for (int kf = 0; kf < KF; ++kf) {
    const int x = l31 + kf;
    const int sw = (x >> 2) & 3;
    bbase[kf][0] = sIn + ((wave * P) * PX + x) * 4 + ((0 + half) ^ sw);
    bbase[kf][1] = sIn + ((wave * P) * PX + x) * 4 + ((2 + half) ^ sw);
  }
  auto load_b = [&](int kt, int kf, u32x4 (&bf)[P][2]) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bf[p][0] = bbase[kf][0][(p + kt) * PX * 4];
      if (NTERM != 1) bf[p][1] = bbase[kf][1][(p + kt) * PX * 4];
    }
  };

  load_w(0);
  load_chunk(0);
  int gg = 0;
#pragma unroll 1
  for (int chunk = 0; chunk < kNChunk; ++chunk) {
    __syncthreads();          // every wave is done with the previous chunk's window
    store_chunk();
#pragma unroll
    for (int grp = 0; grp < NGRP; ++grp, ++gg) {
      const int buf = NWBUF > 1 ? (gg & 1) : 0;
      store_w(buf);           // buffer last read two groups ago: everyone has passed the barrier since
      __syncthreads();
      if (gg + 1 < kNChunk * NGRP) load_w(gg + 1);
      if (grp == 0 && chunk + 1 < kNChunk) load_chunk(chunk + 1);
      if (wave_active) {
        u32x4 a_cur[2][2], a_nxt[2][2], b_cur[P][2], b_nxt[P][2];
        const int kt0 = KF == 1 ? 0 : grp;     // 5x5: the group is kt row `grp`; 7x1: all kt, kf = 0
        load_a(buf, 0, a_cur);
        load_b(kt0, 0, b_cur);
#pragma unroll
        for (int g = 0; g < GT; ++g) {
          const int gn = g + 1 < GT ? g + 1 : g;                 // next tap of the group (clamped)
          load_a(buf, gn, a_nxt);
          load_b(KF == 1 ? gn : kt0, KF == 1 ? 0 : gn, b_nxt);
          __builtin_amdgcn_sched_barrier(0);
          // product term outermost: consecutive MFMAs go to 2*P different accumulators, so no
          // MFMA waits on the result of the one issued just before it
          if (NTERM == 1) {
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb)
                acc[cb][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vs_bf16x8, a_cur[cb][0]),
                                                                    __builtin_bit_cast(vs_bf16x8, b_cur[p][0]), acc[cb][p], 0, 0, 0);
          } else {
#pragma unroll
          for (int term = 0; term < 3; ++term) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const f16x8 bq = __builtin_bit_cast(f16x8, b_cur[p][term == 1 ? 1 : 0]);      // lo only for ah*bl
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) {
                const f16x8 aq = __builtin_bit_cast(f16x8, a_cur[cb][term == 0 ? 1 : 0]);  // lo only for al*bh
                acc[cb][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq, bq, acc[cb][p], 0, 0, 0);
              }
            }
          }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int part = 0; part < 2; ++part) a_cur[cb][part] = a_nxt[cb][part];
#pragma unroll
          for (int p = 0; p < P; ++p) { b_cur[p][0] = b_nxt[p][0]; b_cur[p][1] = b_nxt[p][1]; }
        }
      }
    }
  }

  if (!STATS && !wave_active) return;
  const int f = f0 + l31;
  float* out_b = out + (size_t)b * kCo * plane;
  float m = 0.f;
  float stS[2][16], stQ[2][16];      // STATS: this lane's sums over its rows, per (co block, register)
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float sc = scale[co] * inv, sh = shift[co];
      stS[cb][r] = 0.f; stQ[cb][r] = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = i0 + wave * P + p;
        if (wave_active && i < n_c && f < F) {
          const int t = cls + dil * i;
          const float y = vs_act_fast<ACT>(fmaf(acc[cb][p][r], sc, sh));
          out_b[(size_t)co * plane + (size_t)t * F + f] = y;
          m = fmaxf(m, fabsf(y));
          if (STATS) { stS[cb][r] += y; stQ[cb][r] = fmaf(y, y, stQ[cb][r]); }
        }
      }
    }
  }
  vs_absmax_commit(m, amax_out);     // the output is the next layer's operand
  if (STATS) {
    // train-mode BatchNorm statistics of this layer (same scheme as conv_f16x3_pk.hip): half-wave
    // reduction (32 lanes share their channels), the four waves through LDS, one fp64 atomic per channel
    // and statistic into one of VS_BN_STAT_SLOTS partial slots (spreads ~10^4 workgroups over 64 addresses each)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = stS[cb][r], qv = stQ[cb][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); }
        stS[cb][r] = sv; stQ[cb][r] = qv;
      }
    __syncthreads();                              // everyone is done with the LDS window
    float* red = reinterpret_cast<float*>(sIn);    // [4 waves][64 co][2]
    if (l31 == 0) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          red[(wave * 64 + co) * 2 + 0] = stS[cb][r];
          red[(wave * 64 + co) * 2 + 1] = stQ[cb][r];
        }
    }
    __syncthreads();
    if (tid < 128) {
      const double v = (double)red[tid] + (double)red[128 + tid] + (double)red[256 + tid] + (double)red[384 + tid];
      atomicAdd(bn_stats + (size_t)(blockIdx.x % VS_BN_STAT_SLOTS) * 128 + tid, v);
    }
  }
}

template <int KT, int KF, int P>
int launch_conv(const float* in, const _Float16* wp, const float* scale, const float* shift, const float* in_scale,
                const float* w_scale, float* out, int B, int T, int F, int dil, int act, unsigned* amax_out, hipStream_t stream,
                int i_base = 0, int i_end = 0x7fffffff, int math = VS_MATH_CODE_F16X3, double* bn_stats = nullptr) {
  constexpr int R = 4 * P;
  const int rows_all = (T + dil - 1) / dil;
  const int rows_max = (rows_all < i_end ? rows_all : i_end) - i_base;
  const int n_rt = (rows_max + R - 1) / R;
  const int n_ft = (F + kTileF - 1) / kTileF;
  const long long nblk = (long long)B * dil * n_rt * n_ft;
  VS_REQUIRE(nblk > 0 && nblk < 2147483647LL, "conv64_f16x3: grid of %lld blocks out of range", nblk);
  dim3 grid((unsigned)nblk), block(256);
  if (bn_stats) {        // train-mode forward: conv + bias, statistics fused into the epilogue
    VS_REQUIRE(act == VS_ACT_NONE, "conv64_f16x3: fused BatchNorm statistics need act = NONE");
    if (math == VS_MATH_CODE_BF16) hipLaunchKernelGGL((conv64_f16x3_kernel<KT, KF, P, VS_ACT_NONE, 1, true>), grid, block, 0, stream, in, wp, scale, shift, in_scale, w_scale, out, T, F, dil, n_rt, n_ft, amax_out, i_base, i_end, bn_stats);
    else hipLaunchKernelGGL((conv64_f16x3_kernel<KT, KF, P, VS_ACT_NONE, 3, true>), grid, block, 0, stream, in, wp, scale, shift, in_scale, w_scale, out, T, F, dil, n_rt, n_ft, amax_out, i_base, i_end, bn_stats);
    VS_LAUNCH_CHECK();
    return 0;
  }
  if (math == VS_MATH_CODE_BF16) {
    switch (act) {
      case VS_ACT_RELU: hipLaunchKernelGGL((conv64_f16x3_kernel<KT, KF, P, VS_ACT_RELU, 1>), grid, block, 0, stream, in, wp, scale, shift, in_scale, w_scale, out, T, F, dil, n_rt, n_ft, amax_out, i_base, i_end, (double*)nullptr); break;
      case VS_ACT_MISH: hipLaunchKernelGGL((conv64_f16x3_kernel<KT, KF, P, VS_ACT_MISH, 1>), grid, block, 0, stream, in, wp, scale, shift, in_scale, w_scale, out, T, F, dil, n_rt, n_ft, amax_out, i_base, i_end, (double*)nullptr); break;
      case VS_ACT_NONE: hipLaunchKernelGGL((conv64_f16x3_kernel<KT, KF, P, VS_ACT_NONE, 1>), grid, block, 0, stream, in, wp, scale, shift, in_scale, w_scale, out, T, F, dil, n_rt, n_ft, amax_out, i_base, i_end, (double*)nullptr); break;
      default: VS_REQUIRE(false, "conv64_f16x3: unknown activation %d", act);
    }
    VS_LAUNCH_CHECK();
    return 0;
  }
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL((conv64_f16x3_kernel<KT, KF, P, VS_ACT_RELU>), grid, block, 0, stream, in, wp, scale, shift, in_scale, w_scale, out, T, F, dil, n_rt, n_ft, amax_out, i_base, i_end, (double*)nullptr); break;
    case VS_ACT_MISH: hipLaunchKernelGGL((conv64_f16x3_kernel<KT, KF, P, VS_ACT_MISH>), grid, block, 0, stream, in, wp, scale, shift, in_scale, w_scale, out, T, F, dil, n_rt, n_ft, amax_out, i_base, i_end, (double*)nullptr); break;
    case VS_ACT_NONE: hipLaunchKernelGGL((conv64_f16x3_kernel<KT, KF, P, VS_ACT_NONE>), grid, block, 0, stream, in, wp, scale, shift, in_scale, w_scale, out, T, F, dil, n_rt, n_ft, amax_out, i_base, i_end, (double*)nullptr); break;
    default: VS_REQUIRE(false, "conv64_f16x3: unknown activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

long long tile_rows(int T, int dil, int R) {
  long long tot = 0;
  for (int c = 0; c < dil && c < T; ++c) {
    int n = (T - c + dil - 1) / dil;
    tot += (long long)((n + R - 1) / R) * R;
  }
  return tot;
}

}  // namespace

// packed f16 hi/lo weights, in floats (4-byte units) so that callers size one buffer for either path
extern "C" size_t vs_conv64_packed_f16_floats(int KT, int KF) { return (size_t)(kNChunk * KT * KF + kPadTaps) * (kTapBytes / 4); }

// scale2 <- {s, 1/s} from a running |max| that the producer of the tensor already folded into *amax
int vs_scale_from_absmax_impl(const unsigned* amax, int n, float* scale2, hipStream_t stream) {
  hipLaunchKernelGGL(scale_from_absmax_kernel, dim3(1), dim3(64), 0, stream, amax, n, scale2);
  VS_LAUNCH_CHECK();
  return 0;
}

// folds max|x| of x[0..n) into *amax (not reset: several tensors can share one scale)
int vs_absmax_accum_impl(const float* x, long long n, unsigned* amax, hipStream_t stream) {
  VS_REQUIRE(n > 0, "absmax: n=%lld", n);
  VS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "absmax: tensor must be 16-byte aligned");
  const long long nb = ((n >> 2) + 255) / 256;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(nb < 2048 ? (nb > 0 ? nb : 1) : 2048)), dim3(256), 0, stream, x, n, amax);
  VS_LAUNCH_CHECK();
  return 0;
}

// scale2 <- {s, 1/s} for the tensor x[0..n): s = the power of two that maps max|x| into [2^9, 2^10)
// (one extra pass over x; the orchestration avoids it by having the producer of x track the max)
int vs_pow2_scale_impl(const float* x, long long n, unsigned* amax_scratch, float* scale2, hipStream_t stream) {
  VS_REQUIRE(n > 0, "pow2_scale: n=%lld", n);
  VS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "pow2_scale: tensor must be 16-byte aligned");
  VS_CHECK_HIP(hipMemsetAsync(amax_scratch, 0, sizeof(unsigned), stream));
  const long long nb = ((n >> 2) + 255) / 256;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(nb < 2048 ? (nb > 0 ? nb : 1) : 2048)), dim3(256), 0, stream, x, n, amax_scratch);
  return vs_scale_from_absmax_impl(amax_scratch, 1, scale2, stream);
}

// w [64][64][KT][KF] fp32 -> fragment-ordered hi/lo f16 (scaled by w_scale2[0], which this call computes)
int vs_conv64_pack_f16_impl(const float* w, _Float16* wp, int KT, int KF, int transpose_flip, unsigned* amax_scratch,
                            float* w_scale2, hipStream_t stream, int math) {
  VS_REQUIRE((KT == 7 && KF == 1) || (KT == 5 && KF == 5), "conv64_f16x3: unsupported kernel %dx%d", KT, KF);
  if (int rc = vs_pow2_scale_impl(w, (long long)kCo * kCi * KT * KF, amax_scratch, w_scale2, stream)) return rc;
  const int total = (int)(vs_conv64_packed_f16_floats(KT, KF) * 2);
  hipLaunchKernelGGL(conv_pack_weights_f16_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, wp, KT, KF, transpose_flip, w_scale2,
                     math == VS_MATH_CODE_BF16 ? 1 : 0);
  VS_LAUNCH_CHECK();
  return 0;
}

// which 5x5 kernel vs_conv64_f16x3_fwd_impl launches: 0 = default (persistent pipeline), 1 = one tile per
// workgroup (this file), 2 = persistent pipeline (conv_f16x3_pk.hip).  Test / A-B switch, process-global.
static int g_conv_kernel = 0;
extern "C" int vs_set_conv_kernel(int mode) {
  VS_REQUIRE((mode >= 0 && mode <= 2) || (mode >= 100 && mode <= 130), "vs_set_conv_kernel: mode %d", mode);   // 100+abl: timing ablations
  g_conv_kernel = mode;
  return 0;
}

int vs_conv64_f16x3_fwd_impl(const float* in, const _Float16* wp, const float* scale, const float* shift,
                             const float* in_scale2, const float* w_scale2, float* out,
                             int B, int T, int F, int KT, int KF, int dil, int act, unsigned* amax_out, hipStream_t stream, int math,
                             double* bn_stats) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "conv64_f16x3: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((long long)kChunk * T * F * 4 < (long long)kOob, "conv64_f16x3: T*F=%lld too large for 32-bit slab offsets", (long long)T * F);
  if (KT == 5 && KF == 5 && g_conv_kernel != 1 && (long long)kCo * T * F * 4 < (long long)kOob) {
    // persistent pipelined kernel on 8-row tiles; when the residue classes end 1..4 rows past a multiple
    // of 8 (dil = 16 at T = 301: 19 rows) the 4-row one-tile kernel takes that tail, as in the legacy path
    const int rows_all = (T + dil - 1) / dil, full8 = rows_all / 8 * 8, rem = rows_all - full8;
    const bool tail = full8 > 0 && rem > 0 && rem <= 4;
    if (int rc = vs_conv64_f16x3_pk_impl(in, wp, scale, shift, in_scale2, w_scale2, out, B, T, F, dil, act, amax_out, stream,
                                         g_conv_kernel >= 100 ? g_conv_kernel - 100 : 0, tail ? full8 : 0x7fffffff, math, bn_stats)) return rc;
    return tail ? launch_conv<5, 5, 1>(in, wp, scale, shift, in_scale2, w_scale2, out, B, T, F, dil, act, amax_out, stream, full8, 0x7fffffff, math, bn_stats) : 0;
  }
  // 8-row tiles (P = 2) amortise the KT-1 halo rows and the weight staging over twice the MFMAs
  // (measured 6.6 vs 7.45 ms per layer at equal padding); 4-row tiles only win when they avoid
  // more than ~12 % of padded rows (short residue classes: dil = 16 at T = 301).
  const bool p2 = tile_rows(T, dil, 8) * 100 <= tile_rows(T, dil, 4) * 112;
  // Mixed tiling: when the residue classes end 1..4 rows past a multiple of 8 (dil = 16 at T = 301:
  // 19 rows), the 8-row kernel takes the full tiles and the 4-row kernel the tail, instead of
  // running everything on the less efficient 4-row tiles or padding a third 8-row tile.
  const int rows_all = (T + dil - 1) / dil, full8 = rows_all / 8 * 8, rem = rows_all - full8;
  if (KT == 5 && KF == 5 && full8 > 0 && rem > 0 && rem <= 4) {
    if (int rc = launch_conv<5, 5, 2>(in, wp, scale, shift, in_scale2, w_scale2, out, B, T, F, dil, act, amax_out, stream, 0, full8, math, bn_stats)) return rc;
    return launch_conv<5, 5, 1>(in, wp, scale, shift, in_scale2, w_scale2, out, B, T, F, dil, act, amax_out, stream, full8, 0x7fffffff, math, bn_stats);
  }
  if (KT == 7 && KF == 1) {
    return p2 ? launch_conv<7, 1, 2>(in, wp, scale, shift, in_scale2, w_scale2, out, B, T, F, dil, act, amax_out, stream, 0, 0x7fffffff, math, bn_stats)
              : launch_conv<7, 1, 1>(in, wp, scale, shift, in_scale2, w_scale2, out, B, T, F, dil, act, amax_out, stream, 0, 0x7fffffff, math, bn_stats);
  }
  if (KT == 5 && KF == 5) {
    return p2 ? launch_conv<5, 5, 2>(in, wp, scale, shift, in_scale2, w_scale2, out, B, T, F, dil, act, amax_out, stream, 0, 0x7fffffff, math, bn_stats)
              : launch_conv<5, 5, 1>(in, wp, scale, shift, in_scale2, w_scale2, out, B, T, F, dil, act, amax_out, stream, 0, 0x7fffffff, math, bn_stats);
  }
  VS_REQUIRE(false, "conv64_f16x3: unsupported kernel %dx%d", KT, KF);
  return -1;
}
