// The HBM-bound kernels around the channels-last bf16 convs of the VS_MATH_BF16 path (conv_nhwc.hip):
//   cnn1  (models/voicesplit/model.py:17-19)  x [B][T][F] fp32 -> [B][T][F][64] bf16, 1x7 conv (+ statistics)
//   BatchNorm + activation apply              z -> a, both [B][T][F][64] bf16
//   cnn8  (:51-52) + transpose/view (:72-74)  [B][T][F][64] bf16 -> [B][T][8][F] fp32 (the LSTM feature layout)
// and their backward counterparts.  A pixel is 128 contiguous bytes = eight 16-byte pieces of 8 channels; every
// streaming kernel gives a lane one piece, so a wave moves 1 KiB per instruction and a lane's channels are fixed
// for the whole launch (grid strides are multiples of 8 pieces): per-channel constants live in registers and
// per-channel sums are per-lane partials, folded once at the end.
#include "vs_internal.h"

namespace {

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// fold per-lane partial sums of a lane's 8 channels (channel piece = lane & 7) over the workgroup and add them to
// stats[slot][channel][which] (doubles): v[j] for channel 8*(lane&7)+j
template <int NV>
__device__ __forceinline__ void fold_channel_sums(float (&v)[NV][8], double* __restrict__ stats, float* lds /* [4 waves][NV][64] */,
                                                  unsigned* turn = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NV; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = v[q][j];
      s += __shfl_xor(s, 8, 64);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < 8) lds[(wave * NV + q) * 64 + lane * 8 + j] = s;
    }
  __syncthreads();
  // deterministic mode: the workgroups that share a slot add in index order (vs_common.h)
  const unsigned slot = blockIdx.x % VS_BN_STAT_SLOTS, rank_in_slot = blockIdx.x / VS_BN_STAT_SLOTS;
  unsigned* my_turn = turn ? turn + VS_TURN_SLOT + slot : nullptr;
  vs_turn_begin(my_turn, rank_in_slot);
  for (int i = threadIdx.x; i < NV * 64; i += blockDim.x) {
    const int q = i / 64, c = i - q * 64;
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += lds[(w * NV + q) * 64 + c];
    atomicAdd(stats + (size_t)(blockIdx.x % VS_BN_STAT_SLOTS) * 128 + c * 2 + q, (double)s);
  }
  vs_turn_end(my_turn, rank_in_slot, (gridDim.x - slot + VS_BN_STAT_SLOTS - 1) / VS_BN_STAT_SLOTS);
}

// The 7 input samples x[f-3 .. f+3] of a pixel for all 8 lanes that share it (one lane per 8 channels): lane `piece`
// loads sample `piece` (one load instruction per wave instead of seven mostly redundant ones) and the group exchanges
// them by lane permutes.
__device__ __forceinline__ float load_x7(const float* xr, int f, int F, int piece) {
  const int ff = f + piece - 3;
  return (piece < 7 && ff >= 0 && ff < F) ? xr[ff] : 0.f;
}
__device__ __forceinline__ void exchange_x7(float mine, float (&xv)[7]) {
  const int base = (int)(threadIdx.x & 56u);               // first lane of the pixel's group inside the wave
#pragma unroll
  for (int k = 0; k < 7; ++k) xv[k] = __shfl(mine, base + k, 64);
}
__device__ __forceinline__ void gather_x7(const float* xr, int f, int F, int piece, float (&xv)[7]) {
  exchange_x7(load_x7(xr, f, F, piece), xv);
}

// ---- cnn1 ---------------------------------------------------------------------------------------------------
// out[b][t][f][co] = act(scale[co] * sum_j w[co][j] x[b][t][f+j-3] + shift[co]); STATS: sum / sum of squares of out
template <int ACT, bool STATS>
__global__ __launch_bounds__(256)
void nhwc_conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                            const float* __restrict__ shift, unsigned short* __restrict__ out, long long npix, int F,
                            double* __restrict__ stats, const float* __restrict__ bias) {
  __shared__ float red[4 * 2 * 64];
  const int piece = threadIdx.x & 7;                      // channels 8*piece .. 8*piece+7
  // the lane's 8 channels as 4 pairs: v_pk_fma_f32 does two channels per issue slot (with 64 channels x 7 taps + Mish per
  // pixel this kernel is VALU-, not HBM-bound once the BatchNorm + activation ride in it)
  vs_f32x2 wr[4][7], sc[4], sh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c0 = piece * 8 + 2 * q;
    sc[q] = vs_f32x2{scale[c0], scale[c0 + 1]};
    // bias != NULL: scale / shift are a BatchNorm's constants for z = conv + bias: act(scale (conv + bias) + shift)
    sh[q] = bias ? vs_f32x2{fmaf(bias[c0], sc[q].x, shift[c0]), fmaf(bias[c0 + 1], sc[q].y, shift[c0 + 1])} : vs_f32x2{shift[c0], shift[c0 + 1]};
#pragma unroll
    for (int k = 0; k < 7; ++k) wr[q][k] = vs_f32x2{w[c0 * 7 + k], w[(c0 + 1) * 7 + k]};
  }
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  const long long stride = (long long)gridDim.x * 32;     // pixels per sweep of the grid
  const int sr = (int)(stride % F);                       // the column is carried along: one 64-bit division per launch, not per pixel
  long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  int f = (int)(p % F);
  // the next pixel's input sample is loaded before this one is computed: the loop is otherwise load -> ~150 VALU -> store with
  // nothing in flight (4 waves per SIMD do not cover a 2 us miss)
  float mine = p < npix ? load_x7(x + (p - f), f, F, piece) : 0.f;
  for (; p < npix; p += stride) {
    int fn = f + sr;
    if (fn >= F) fn -= F;
    const long long pn = p + stride;
    const float mine_next = pn < npix ? load_x7(x + (pn - fn), fn, F, piece) : 0.f;
    float xv[7];
    exchange_x7(mine, xv);
    mine = mine_next;
    f = fn;
    u4v pk;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vs_f32x2 s2 = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 7; ++k) s2 = __builtin_elementwise_fma(wr[q][k], vs_f32x2{xv[k], xv[k]}, s2);
      const vs_f32x2 y = vs_act_fast2<ACT>(__builtin_elementwise_fma(s2, sc[q], sh[q]));
      if (STATS) {
        acc[0][2 * q] += y.x; acc[0][2 * q + 1] += y.y;
        acc[1][2 * q] = fmaf(y.x, y.x, acc[1][2 * q]); acc[1][2 * q + 1] = fmaf(y.y, y.y, acc[1][2 * q + 1]);
      }
      pk[q] = vs_pack_bf16(y.x, y.y);
    }
    __builtin_nontemporal_store(pk, reinterpret_cast<u4v*>(out + p * 64 + piece * 8));
  }
  if (STATS) fold_channel_sums<2>(acc, stats, red);
}

// ---- BatchNorm + activation apply -------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(256)
void nhwc_bn_apply_kernel(const u4v* __restrict__ z, u4v* __restrict__ a, const float* __restrict__ scale,
                          const float* __restrict__ shift, long long npieces) {
  const int piece = threadIdx.x & 7;
  vs_f32x2 sc[4], sh[4];                                   // the lane's 8 channels as the 4 pairs its 32-bit words hold
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sc[q] = vs_f32x2{scale[piece * 8 + 2 * q], scale[piece * 8 + 2 * q + 1]};
    sh[q] = vs_f32x2{shift[piece * 8 + 2 * q], shift[piece * 8 + 2 * q + 1]};
  }
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  auto apply = [&](const u4v v) {
    u4v o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const vs_f32x2 y = vs_act_fast2<ACT>(__builtin_elementwise_fma(vs_f32x2{bf_lo(v[q]), bf_hi(v[q])}, sc[q], sh[q]));
      o[q] = vs_pack_bf16(y.x, y.y);
    }
    return o;
  };
  for (; i + stride < npieces; i += 2 * stride) {         // two pieces in flight per lane (four: 0.628 vs 0.625 ms at B = 64, round 4)
    const u4v v0 = __builtin_nontemporal_load(z + i), v1 = __builtin_nontemporal_load(z + i + stride);
    __builtin_nontemporal_store(apply(v0), a + i);
    __builtin_nontemporal_store(apply(v1), a + i + stride);
  }
  if (i < npieces) __builtin_nontemporal_store(apply(__builtin_nontemporal_load(z + i)), a + i);
}

// ---- cnn8 + transpose/view ---------------------------------------------------------------------------------------
// out[b][t][co][f] = act(scale[co] * sum_ci w[co][ci] in[b][t][f][ci] + shift[co]), co < 8.  One wave = 16 pixels of a
// row per step: the B operand of v_mfma_f32_16x16x32_bf16 is the pixels as they lie in memory (lane = (pixel, 8
// channels) = one 16-byte load), A = the 8 x 64 weights zero-padded to 16 rows; C rows 0..7 -> 8 feature rows.
// STATS: the per-channel sum and sum of squares of what is stored (train-mode BatchNorm of cnn8: models/voicesplit/model.py:52)
// accumulate per lane over the launch and flush once: stats[slot = block % VS_BN_STAT_SLOTS][8 channels][2] doubles
// (vs_bn_finalize_impl folds the slots).  Replaces a pass over the 370 MB feature tensor.
// PRE >= 0: `in` is the UN-normalised output z7 of the layer below and the kernel applies that layer's BatchNorm +
// activation PRE on its way into the matrix pipe (a7 = act(z7 * pre_scale + pre_shift), rounded to bf16 exactly as
// nhwc_bn_apply_kernel would have stored it): cnn8 is an HBM-bound consumer with idle VALU, so the training step needs no
// BatchNorm-apply pass for cnn7 and no a7 tensor at all (the backward recomputes it the same way).
template <int ACT, bool STATS = false, int PRE = -1>
__global__ __launch_bounds__(256)
void nhwc_conv_last_kernel(const unsigned short* __restrict__ in, const float* __restrict__ w, const float* __restrict__ scale,
                           const float* __restrict__ shift, float* __restrict__ out, long long nrows /* B*T */, int F,
                           double* __restrict__ stats, const float* __restrict__ pre_scale = nullptr, const float* __restrict__ pre_shift = nullptr,
                           unsigned short* __restrict__ rows_bf16 = nullptr, int Kp = 0, unsigned* turn = nullptr) {
  // rows_bf16 != NULL (eval forward, whole path): the output goes out as the bf16 A operand of the LSTM input GEMM instead -- rows
  // [nrows][Kp] (element co * F + f of row (b, t), the K padding zeroed by the wave that owns a row's first block), `out` unused
  __shared__ float red[4 * 16];
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  vs_f32x2 psc[2][4], psh[2][4];      // PRE: constants of this lane's 16 channels (k-chunk kc: 32 kc + 8 g + 2 q, + 1)
  if (PRE >= 0) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = kc * 32 + g * 8 + 2 * q;
        psc[kc][q] = vs_f32x2{pre_scale[ch], pre_scale[ch + 1]};
        psh[kc][q] = vs_f32x2{pre_shift[ch], pre_shift[ch + 1]};
      }
  }
  vs_bf16x8 wa[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int j = 0; j < 8; ++j) wa[kc][j] = (__bf16)(n < 8 ? w[n * 64 + kc * 32 + g * 8 + j] : 0.f);
  float sc[4], sh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = (g * 4 + r) & 7;
    sc[r] = scale[co];
    sh[r] = shift[co];
  }
  const int blocks_per_row = (F + 15) >> 4;
  const long long nblk = nrows * blocks_per_row;
  const long long wstride = (long long)gridDim.x * 4;
  const long long sq = wstride / blocks_per_row;          // (row, block column) are carried along: one division per launch
  const int sr = (int)(wstride - sq * blocks_per_row);
  // workgroups go round the 8 XCDs: the ones of an XCD take neighbouring blocks, so that the 64-byte pieces of an output line meet in one L2
  const int per_xcd = gridDim.x >> 3;
  const int vblock = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  long long blk = (long long)vblock * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  long long row = blk / blocks_per_row;
  int bc = (int)(blk - row * blocks_per_row);
  for (; blk < nblk; blk += wstride, row += sq, bc += sr) {
    if (bc >= blocks_per_row) { bc -= blocks_per_row; ++row; }
    const int f = bc * 16 + n;
    const bool ok = f < F;
    const u4v* src = reinterpret_cast<const u4v*>(in + ((row * F + (ok ? f : 0)) << 6)) + g;
    u4v b0 = ok ? src[0] : u4v{0u, 0u, 0u, 0u}, b1 = ok ? src[4] : u4v{0u, 0u, 0u, 0u};
    if (PRE >= 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const vs_f32x2 y0 = vs_act_fast2<(PRE >= 0 ? PRE : VS_ACT_NONE)>(__builtin_elementwise_fma(vs_f32x2{bf_lo(b0[q]), bf_hi(b0[q])}, psc[0][q], psh[0][q]));
        const vs_f32x2 y1 = vs_act_fast2<(PRE >= 0 ? PRE : VS_ACT_NONE)>(__builtin_elementwise_fma(vs_f32x2{bf_lo(b1[q]), bf_hi(b1[q])}, psc[1][q], psh[1][q]));
        b0[q] = vs_pack_bf16(y0.x, y0.y);
        b1[q] = vs_pack_bf16(y1.x, y1.y);
      }
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], __builtin_bit_cast(vs_bf16x8, b0), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], __builtin_bit_cast(vs_bf16x8, b1), c, 0, 0, 0);
    if (!STATS && PRE < 0 && rows_bf16) {
      if (ok && g < 2) {
        unsigned short* o = rows_bf16 + (size_t)row * Kp + (size_t)(g * 4) * F + f;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(size_t)r * F] = (unsigned short)(vs_pack_bf16(vs_act_fast<ACT>(fmaf(c[r], sc[r], sh[r])), 0.f) & 0xffffu);
      }
      if (bc == 0)
        for (int k = 8 * F + lane; k < Kp; k += 64) rows_bf16[(size_t)row * Kp + k] = 0;
    } else if (ok && g < 2) {
      float* o = out + (row * 8 + g * 4) * F + f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = vs_act_fast<ACT>(fmaf(c[r], sc[r], sh[r]));
        o[(size_t)r * F] = v;
        if (STATS) { ssum[r] += v; ssq[r] = fmaf(v, v, ssq[r]); }
      }
    }
  }
  if (STATS) {
    // lanes (g, n): channel 4 g + r for g < 2; fold the 16 columns of a group, then the four waves
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = ssum[r], b = ssq[r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (n == 0 && g < 2) { red[wave * 16 + (g * 4 + r) * 2] = a; red[wave * 16 + (g * 4 + r) * 2 + 1] = b; }
    }
    __syncthreads();
    const unsigned slot = blockIdx.x % VS_BN_STAT_SLOTS, rank_in_slot = blockIdx.x / VS_BN_STAT_SLOTS;
    unsigned* my_turn = turn ? turn + VS_TURN_SLOT + slot : nullptr;
    vs_turn_begin(my_turn, rank_in_slot);
    if (threadIdx.x < 16)
      atomicAdd(stats + (size_t)slot * 16 + threadIdx.x,
                (double)red[threadIdx.x] + (double)red[16 + threadIdx.x] + (double)red[32 + threadIdx.x] + (double)red[48 + threadIdx.x]);
    vs_turn_end(my_turn, rank_in_slot, (gridDim.x - slot + VS_BN_STAT_SLOTS - 1) / VS_BN_STAT_SLOTS);
  }
}

// ---- BatchNorm + activation backward -----------------------------------------------------------------------------
// dy = da * act'(z * scale + shift);  pass 1: per-channel sum dy, sum dy * xhat (xhat = (z - mean) * invstd);
// pass 2: dz = cA * dy + cB * z + cC (coefficients from bn_bwd_finalize, conv_bwd.hip), bf16, may overwrite da.
template <int ACT>
__device__ __forceinline__ float nhwc_act_grad(float y) {
  if (ACT == VS_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (ACT == VS_ACT_MISH) return vs_mish_grad_fast(y);
  return 1.f;
}

template <int ACT>
__global__ __launch_bounds__(256)
void nhwc_bn_bwd_stats_kernel(const u4v* __restrict__ da, const u4v* __restrict__ z, long long npieces,
                              const float* __restrict__ scale, const float* __restrict__ shift,
                              const float* __restrict__ mean, const float* __restrict__ invstd, double* __restrict__ stats) {
  __shared__ float red[4 * 2 * 64];
  const int piece = threadIdx.x & 7;
  float sc[8], sh[8], mu[8], is[8], acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = piece * 8 + j;
    sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; is[j] = invstd[c];
    acc[0][j] = 0.f; acc[1][j] = 0.f;
  }
  const long long stride = (long long)gridDim.x * 256;
  auto eat = [&](const u4v g, const u4v v) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float z0 = bf_lo(v[q]), z1 = bf_hi(v[q]);
      const float d0 = bf_lo(g[q]) * nhwc_act_grad<ACT>(fmaf(z0, sc[2 * q], sh[2 * q]));
      const float d1 = bf_hi(g[q]) * nhwc_act_grad<ACT>(fmaf(z1, sc[2 * q + 1], sh[2 * q + 1]));
      acc[0][2 * q] += d0;
      acc[0][2 * q + 1] += d1;
      acc[1][2 * q] = fmaf(d0, (z0 - mu[2 * q]) * is[2 * q], acc[1][2 * q]);
      acc[1][2 * q + 1] = fmaf(d1, (z1 - mu[2 * q + 1]) * is[2 * q + 1], acc[1][2 * q + 1]);
    }
  };
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < npieces; i += 2 * stride) {
    const u4v g0 = __builtin_nontemporal_load(da + i), v0 = __builtin_nontemporal_load(z + i);
    const u4v g1 = __builtin_nontemporal_load(da + i + stride), v1 = __builtin_nontemporal_load(z + i + stride);
    eat(g0, v0);
    eat(g1, v1);
  }
  if (i < npieces) eat(__builtin_nontemporal_load(da + i), __builtin_nontemporal_load(z + i));
  fold_channel_sums<2>(acc, stats, red);
}

template <int ACT>
__global__ __launch_bounds__(256)
void nhwc_bn_bwd_apply_kernel(const u4v* da, const u4v* __restrict__ z, u4v* dz, long long npieces,
                              const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ coef) {
  const int piece = threadIdx.x & 7;
  float sc[8], sh[8], cA[8], cB[8], cC[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = piece * 8 + j;
    sc[j] = scale[c]; sh[j] = shift[c]; cA[j] = coef[c]; cB[j] = coef[64 + c]; cC[j] = coef[128 + c];
  }
  const long long stride = (long long)gridDim.x * 256;
  auto apply = [&](const u4v g, const u4v v) {
    u4v o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float z0 = bf_lo(v[q]), z1 = bf_hi(v[q]);
      const float d0 = bf_lo(g[q]) * nhwc_act_grad<ACT>(fmaf(z0, sc[2 * q], sh[2 * q]));
      const float d1 = bf_hi(g[q]) * nhwc_act_grad<ACT>(fmaf(z1, sc[2 * q + 1], sh[2 * q + 1]));
      o[q] = vs_pack_bf16(fmaf(cA[2 * q], d0, fmaf(cB[2 * q], z0, cC[2 * q])),
                          fmaf(cA[2 * q + 1], d1, fmaf(cB[2 * q + 1], z1, cC[2 * q + 1])));
    }
    return o;
  };
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < npieces; i += 2 * stride) {
    const u4v g0 = __builtin_nontemporal_load(da + i), v0 = __builtin_nontemporal_load(z + i);
    const u4v g1 = __builtin_nontemporal_load(da + i + stride), v1 = __builtin_nontemporal_load(z + i + stride);
    __builtin_nontemporal_store(apply(g0, v0), dz + i);
    __builtin_nontemporal_store(apply(g1, v1), dz + i + stride);
  }
  if (i < npieces) __builtin_nontemporal_store(apply(__builtin_nontemporal_load(da + i), __builtin_nontemporal_load(z + i)), dz + i);
}

// cnn1: pass 2 fused with the 1x7 weight gradient: dz1 = cA dy + cB z + cC is contracted with the 7 shifted inputs on
// the spot (dz1 is never written): acc[c][k] += dz1[b][t][f][c] * x[b][t][f + k - 3]
template <int ACT>
__global__ __launch_bounds__(256)
void nhwc_bn_bwd_first_kernel(const u4v* __restrict__ da, const u4v* __restrict__ z, const float* __restrict__ x,
                              long long npix, int F, const float* __restrict__ scale, const float* __restrict__ shift,
                              const float* __restrict__ coef, double* __restrict__ acc_out /* [64][7] */) {
  __shared__ float red[4 * 7 * 64];
  const int piece = threadIdx.x & 7;
  float sc[8], sh[8], cA[8], cB[8], cC[8], acc[7][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = piece * 8 + j;
    sc[j] = scale[c]; sh[j] = shift[c]; cA[j] = coef[c]; cB[j] = coef[64 + c]; cC[j] = coef[128 + c];
#pragma unroll
    for (int k = 0; k < 7; ++k) acc[k][j] = 0.f;
  }
  const long long stride = (long long)gridDim.x * 32;
  const int sr = (int)(stride % F);
  long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  int f = (int)(p % F);
  for (; p < npix; p += stride, f += sr) {
    if (f >= F) f -= F;
    float xv[7];
    gather_x7(x + (p - f), f, F, piece, xv);
    const u4v g = __builtin_nontemporal_load(da + p * 8 + piece), v = __builtin_nontemporal_load(z + p * 8 + piece);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = 2 * q + e;
        const float zv = e ? bf_hi(v[q]) : bf_lo(v[q]);
        const float dy = (e ? bf_hi(g[q]) : bf_lo(g[q])) * nhwc_act_grad<ACT>(fmaf(zv, sc[j], sh[j]));
        const float dzv = fmaf(cA[j], dy, fmaf(cB[j], zv, cC[j]));
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[k][j] = fmaf(dzv, xv[k], acc[k][j]);
      }
    }
  }
  // fold: lanes sharing a channel piece, then the waves; one fp64 atomic per (channel, tap) and workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = acc[k][j];
      s += __shfl_xor(s, 8, 64);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < 8) red[(wave * 7 + k) * 64 + lane * 8 + j] = s;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 7 * 64; i += 256) {
    const int k = i / 64, c = i - k * 64;
    atomicAdd(acc_out + c * 7 + k, (double)(red[(0 * 7 + k) * 64 + c] + red[(1 * 7 + k) * 64 + c] + red[(2 * 7 + k) * 64 + c] + red[(3 * 7 + k) * 64 + c]));
  }
}

__global__ void nhwc_cvt_f64_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

// cnn8 backward in one pass over a7: din[b][t][f][ci] = sum_co w[co][ci] dz8[b][t][co][f]  (data gradient, bf16) and
// dw[co][ci] = sum_pixels dz8[..][co][..] a7[..][ci]  (weight gradient: per-workgroup partial sums part[block][co][ci])
// DYACT >= 0: the data gradient is turned into dy = din * act'(z7 * scale + shift) before it is stored, and the
// per-channel sums of dy and dy * xhat go to bn_stats (the first pass of cnn7's BatchNorm backward, fused)
struct LastBwdBn {
  const u4v* z;
  const float *scale, *shift, *mean, *invstd;
  double* stats;
  unsigned* turn;          // deterministic mode: the statistics are flushed in workgroup order; else NULL
};

// activation and its derivative of one value; Mish: both from ONE exp2 and ONE rcp -- u, n, r as in vs_mish_fast2, so the
// activation is bitwise what nhwc_bn_apply_kernel / the PRE form of nhwc_conv_last_kernel produce, and
// Mish' = r (n + 4 y u (u + 1) r) (the identity derived at conv_nhwc.hip's dy epilogue)
template <int ACT>
__device__ __forceinline__ void nhwc_act_both(float y, float& a, float& d) {
  if (ACT == VS_ACT_MISH) {
    const float u = __builtin_amdgcn_exp2f(fminf(y, 20.0f) * 1.44269504088896340736f);
    const float n = u * (u + 2.0f);
    const float r = __builtin_amdgcn_rcpf(n + 2.0f);
    a = y * (n * r);
    const float dd = r * fmaf(4.0f * y * r, u * (u + 1.0f), n);
    d = y > 20.0f ? 1.0f : dd;
  } else if (ACT == VS_ACT_RELU) {
    a = fmaxf(y, 0.0f);
    d = y > 0.0f ? 1.0f : 0.0f;
  } else {
    a = y;
    d = 1.0f;
  }
}

// RECOMP: a7 is not read but recomputed from z7 (a7 = bf16(act(z7 * scale + shift)), the forward's own values: see
// nhwc_conv_last_kernel's PRE form) -- one 1.48 GB operand stream less
// activation and derivative for a pair of channels (the packed form of nhwc_act_both: the activation is bitwise vs_mish_fast2's,
// i.e. what nhwc_bn_apply_kernel and the PRE form of nhwc_conv_last_kernel store; no selects: at the y = 20 clamp n r = 1 and the
// derivative's second term vanishes in fp32)
template <int ACT>
__device__ __forceinline__ void nhwc_act_both2(vs_f32x2 y, vs_f32x2& a, vs_f32x2& d) {
  if (ACT == VS_ACT_MISH) {
    const vs_f32x2 yc = {fminf(y.x, 20.0f), fminf(y.y, 20.0f)};
    const vs_f32x2 e = yc * 1.44269504088896340736f;
    const vs_f32x2 u = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    const vs_f32x2 n = u * (u + 2.0f);
    const vs_f32x2 dn = n + 2.0f;
    const vs_f32x2 r = {__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y)};
    a = y * (n * r);
    d = r * __builtin_elementwise_fma((yc * 4.0f) * r, __builtin_elementwise_fma(u, u, u), n);
  } else if (ACT == VS_ACT_RELU) {
    a = vs_f32x2{fmaxf(y.x, 0.0f), fmaxf(y.y, 0.0f)};
    d = vs_f32x2{y.x > 0.0f ? 1.0f : 0.0f, y.y > 0.0f ? 1.0f : 0.0f};
  } else {
    a = y;
    d = vs_f32x2{1.0f, 1.0f};
  }
}

// Round 4: channel PAIRS throughout (v_pk_fma_f32: the 64 + 64 FMAs of the data and weight gradient and the activation /
// derivative were ~310 VALU instructions per pixel piece -- the kernel ran at 3 TB/s, VALU-bound), and the next pixel's
// operands are in flight while this one is computed (2 waves per SIMD do not hide a load issued at the top of its own iteration).
template <int DYACT, bool RECOMP = false>
__global__ __launch_bounds__(256)
void nhwc_conv_last_bwd_kernel(const float* __restrict__ dz8, const float* __restrict__ w, const u4v* __restrict__ a7,
                               u4v* __restrict__ din, float* __restrict__ part, long long npix, int F, LastBwdBn bn) {
  __shared__ float red[4 * 8 * 64];
  const int piece = threadIdx.x & 7;
  vs_f32x2 wr[8][4], acc[8][4];            // [co][pair], ci = 8 piece + 2 pair + {0, 1}
  vs_f32x2 sc[4], sh[4], mu[4], is[4], bacc[2][4];
#pragma unroll
  for (int co = 0; co < 8; ++co)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wr[co][q] = vs_f32x2{w[co * 64 + piece * 8 + 2 * q], w[co * 64 + piece * 8 + 2 * q + 1]};
      acc[co][q] = vs_f32x2{0.f, 0.f};
    }
  if (DYACT >= 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = piece * 8 + 2 * q;
      sc[q] = vs_f32x2{bn.scale[c], bn.scale[c + 1]}; sh[q] = vs_f32x2{bn.shift[c], bn.shift[c + 1]};
      mu[q] = vs_f32x2{bn.mean[c], bn.mean[c + 1]}; is[q] = vs_f32x2{bn.invstd[c], bn.invstd[c + 1]};
      bacc[0][q] = vs_f32x2{0.f, 0.f}; bacc[1][q] = vs_f32x2{0.f, 0.f};
    }
  }
  const long long stride = (long long)gridDim.x * 32;
  const int sr = (int)(stride % F);
  const long long sq = stride / F;
  long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  long long row = p / F;
  int f = (int)(p - row * F);
  // operands of a pixel piece, fetched one iteration ahead
  struct In { float d[8]; u4v zv, av; };
  auto fetch = [&](long long pq, long long rq, int fq, In& in) {
    const float* dzr = dz8 + rq * 8 * F + fq;
#pragma unroll
    for (int co = 0; co < 8; ++co) in.d[co] = dzr[(size_t)co * F];
    if (DYACT >= 0) in.zv = __builtin_nontemporal_load(bn.z + pq * 8 + piece);
    if (!RECOMP) in.av = __builtin_nontemporal_load(a7 + pq * 8 + piece);
  };
  In cur, nxt;
  cur.zv = cur.av = nxt.zv = nxt.av = u4v{0u, 0u, 0u, 0u};
  if (p < npix) fetch(p, row, f, cur);
  for (; p < npix; p += stride) {
    int fn = f + sr;
    long long rown = row + sq;
    if (fn >= F) { fn -= F; ++rown; }
    const long long pn = p + stride;
    if (pn < npix) fetch(pn, rown, fn, nxt);
    vs_f32x2 zf[4], dact[4];
    u4v av = cur.av;
    if (DYACT >= 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        zf[q] = vs_f32x2{bf_lo(cur.zv[q]), bf_hi(cur.zv[q])};
        vs_f32x2 aq;
        nhwc_act_both2<(DYACT >= 0 ? DYACT : VS_ACT_NONE)>(__builtin_elementwise_fma(zf[q], sc[q], sh[q]), aq, dact[q]);
        if (RECOMP) av[q] = vs_pack_bf16(aq.x, aq.y);
      }
    }
    vs_f32x2 g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = vs_f32x2{0.f, 0.f};
#pragma unroll
    for (int co = 0; co < 8; ++co) {
      const vs_f32x2 dd = {cur.d[co], cur.d[co]};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        g[q] = __builtin_elementwise_fma(wr[co][q], dd, g[q]);
        acc[co][q] = __builtin_elementwise_fma(dd, vs_f32x2{bf_lo(av[q]), bf_hi(av[q])}, acc[co][q]);
      }
    }
    if (DYACT >= 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        g[q] *= dact[q];
        bacc[0][q] += g[q];
        bacc[1][q] = __builtin_elementwise_fma(g[q], (zf[q] - mu[q]) * is[q], bacc[1][q]);
      }
    }
    __builtin_nontemporal_store(u4v{vs_pack_bf16(g[0].x, g[0].y), vs_pack_bf16(g[1].x, g[1].y), vs_pack_bf16(g[2].x, g[2].y), vs_pack_bf16(g[3].x, g[3].y)},
                                din + p * 8 + piece);
    cur = nxt;
    f = fn;
    row = rown;
  }
  if (DYACT >= 0) {
    float b8[2][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      b8[0][2 * q] = bacc[0][q].x; b8[0][2 * q + 1] = bacc[0][q].y;
      b8[1][2 * q] = bacc[1][q].x; b8[1][2 * q + 1] = bacc[1][q].y;
    }
    fold_channel_sums<2>(b8, bn.stats, red, bn.turn);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int co = 0; co < 8; ++co)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = (j & 1) ? acc[co][j >> 1].y : acc[co][j >> 1].x;
      s += __shfl_xor(s, 8, 64);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < 8) red[(wave * 8 + co) * 64 + lane * 8 + j] = s;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256)
    part[(size_t)blockIdx.x * 512 + i] = red[i] + red[512 + i] + red[1024 + i] + red[1536 + i];
}

// ---- cnn1 by recomputation (round 4) ---------------------------------------------------------------------------------
// cnn1 is 7 MACs per output and its input is 1/64 of its output: nothing of it needs to be WRITTEN to be available again.
// Train-mode BatchNorm needs the batch statistics of z1 = conv(x) + bias before the activation can be applied -- but z1 is
// linear in seven shifted copies of x, so its per-channel sum and sum of squares follow from the 7 sums S[k] and the 28
// products R[k][k'] of the shifted inputs (ONE pass over the 46 MB input, independent of the 64 channels):
//   sum z_c   = sum_k w[c][k] S[k] + N b_c          sum z_c^2 = sum_kk' w[c][k] w[c][k'] R[k][k'] + 2 b_c sum_k w[c][k] S[k] + N b_c^2
// The forward then writes a1 = act(BN(z1)) in one pass (no z1 tensor, no apply pass); the backward recomputes z1 from x
// beside the derivative and needs ONE pass over da1: the BatchNorm backward dz = cA dy + cB z + cC is linear in dy and z,
// so dw[c][k] = sum dz x_k = cA sum(dy x_k) + cB sum(z x_k) + cC S[k], with sum(z x_k) = sum_k' w[c][k'] R[k'][k] + b_c S[k].
// x_k = x[b][t][f + k - 3], zero outside the row (ZeroPad2d((3, 3, 0, 0)), models/voicesplit/model.py:17).
constexpr int kMomN = 7 + 28;                            // S[0..6], then R[k][k'] for k <= k' row by row

__global__ __launch_bounds__(256)
void nhwc_first_moments_kernel(const float* __restrict__ x, long long npix, int F, double* __restrict__ mom, unsigned* turn) {
  __shared__ float red[4][kMomN];
  float acc[kMomN];
#pragma unroll
  for (int i = 0; i < kMomN; ++i) acc[i] = 0.f;
  const long long stride = (long long)gridDim.x * 256;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += stride) {
    const int f = (int)(p % F);
    const float* xr = x + (p - f);
    float xv[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int ff = f + k - 3;
      xv[k] = (ff >= 0 && ff < F) ? xr[ff] : 0.f;
    }
    int i = 7;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      acc[k] += xv[k];
#pragma unroll
      for (int k2 = k; k2 < 7; ++k2) { acc[i] = fmaf(xv[k], xv[k2], acc[i]); ++i; }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < kMomN; ++i) {
    float v = acc[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  // deterministic mode: mom is [VS_BN_STAT_SLOTS][64] then (the workgroups of a slot add in index order, a fold kernel sums the slots)
  const unsigned slot = blockIdx.x % VS_BN_STAT_SLOTS, rank_in_slot = blockIdx.x / VS_BN_STAT_SLOTS;
  unsigned* my_turn = turn ? turn + VS_TURN_SLOT + slot : nullptr;
  if (turn) mom += (size_t)slot * 64;
  vs_turn_begin(my_turn, rank_in_slot);
  if (threadIdx.x < kMomN) atomicAdd(mom + threadIdx.x, (double)red[0][threadIdx.x] + (double)red[1][threadIdx.x] + (double)red[2][threadIdx.x] + (double)red[3][threadIdx.x]);
  vs_turn_end(my_turn, rank_in_slot, (gridDim.x - slot + VS_BN_STAT_SLOTS - 1) / VS_BN_STAT_SLOTS);
}

__global__ void fold_moment_slots_kernel(const double* __restrict__ slots, double* __restrict__ mom) {      // 64 threads
  if (threadIdx.x >= kMomN) return;
  double v = 0.0;
  for (int k = 0; k < VS_BN_STAT_SLOTS; ++k) v += slots[(size_t)k * 64 + threadIdx.x];
  mom[threadIdx.x] = v;
}

__device__ __forceinline__ double mom_R(const double* mom, int k, int k2) {          // R[k][k2], symmetric
  if (k > k2) { const int t = k; k = k2; k2 = t; }
  return mom[7 + k * 7 - k * (k - 1) / 2 + (k2 - k)];
}

// stats[c] = {sum z_c, sum z_c^2} over the `count` pixels, from the moments (one thread per channel, fp64)
__global__ void nhwc_first_stats_kernel(const double* __restrict__ mom, const float* __restrict__ w, const float* __restrict__ bias,
                                        double count, double* __restrict__ stats) {
  const int c = threadIdx.x;
  if (c >= 64) return;
  const double b = bias[c];
  double ws = 0.0, q = 0.0;
  for (int k = 0; k < 7; ++k) {
    ws += (double)w[c * 7 + k] * mom[k];
    for (int k2 = 0; k2 < 7; ++k2) q += (double)w[c * 7 + k] * (double)w[c * 7 + k2] * mom_R(mom, k, k2);
  }
  stats[2 * c] = ws + count * b;
  stats[2 * c + 1] = q + 2.0 * b * ws + count * b * b;
}

// one pass over da1: per channel sum dy, sum dy xhat, sum dy x_k (k = 0..6) with z1 recomputed from x
// act'(y) for a pair of channels; Mish' = r (n + 4 y u (u + 1) r), u = e^y, n = u (u + 2), r = 1 / (n + 2) (the identity derived at
// conv_nhwc.hip's dy epilogue): one exp2 and one rcp per channel, the rest packed; y clamped at 20, where the expression is 1 to fp32
template <int ACT>
__device__ __forceinline__ vs_f32x2 nhwc_act_grad2(vs_f32x2 y) {
  if (ACT == VS_ACT_RELU) return vs_f32x2{y.x > 0.f ? 1.f : 0.f, y.y > 0.f ? 1.f : 0.f};
  if (ACT == VS_ACT_MISH) {
    const vs_f32x2 yc = {fminf(y.x, 20.0f), fminf(y.y, 20.0f)};
    const vs_f32x2 e = yc * 1.44269504088896340736f;
    const vs_f32x2 u = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    const vs_f32x2 n = u * (u + 2.0f);
    const vs_f32x2 d = n + 2.0f;
    const vs_f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const vs_f32x2 t = __builtin_elementwise_fma(u, u, u);
    return r * __builtin_elementwise_fma((yc * 4.0f) * r, t, n);
  }
  return vs_f32x2{1.f, 1.f};
}

template <int ACT>
__global__ __launch_bounds__(256)
void nhwc_first_bwd_kernel(const u4v* __restrict__ da, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                           long long npix, int F, const float* __restrict__ scale, const float* __restrict__ shift,
                           const float* __restrict__ mean, const float* __restrict__ invstd, double* __restrict__ acc_out /* [64][9] */,
                           unsigned* turn) {
  __shared__ float red[4 * 9 * 64];
  const int piece = threadIdx.x & 7;
  // channel pairs throughout (v_pk_*_f32): per pixel and pair 7 FMAs for z, the derivative, 9 FMAs into the sums
  vs_f32x2 wr[4][7], bs[4], sc[4], sh[4], mu[4], is[4], acc[9][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c0 = piece * 8 + 2 * q;
    bs[q] = vs_f32x2{bias[c0], bias[c0 + 1]};
    sc[q] = vs_f32x2{scale[c0], scale[c0 + 1]};
    sh[q] = vs_f32x2{shift[c0], shift[c0 + 1]};
    mu[q] = vs_f32x2{mean[c0], mean[c0 + 1]};
    is[q] = vs_f32x2{invstd[c0], invstd[c0 + 1]};
#pragma unroll
    for (int k = 0; k < 7; ++k) wr[q][k] = vs_f32x2{w[c0 * 7 + k], w[(c0 + 1) * 7 + k]};
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k][q] = vs_f32x2{0.f, 0.f};
  }
  const long long stride = (long long)gridDim.x * 32;
  const int sr = (int)(stride % F);
  long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  int f = (int)(p % F);
  // The next pixel's gradient piece and input sample are loaded before this one is computed.  (Measured, tools/edge_micro.py at
  // B = 64: 1.07-1.10 ms = 1.4 TB/s with or without this, with a ring three deep, with branch-free loads, and with a two-deep ring
  // of untracked inline-assembly loads + explicit vmcnt at one wave per SIMD -- and the scalar form of the arithmetic ran the
  // same 1.1-1.2 ms as this packed one.  The loop is bound by its VALU work: ~160 instructions + 16 transcendentals per 8-channel
  // piece, and v_pk_fma_f32 buys nothing over two v_fma_f32 here.  Halving it means the matrix pipe: z as a K = 7 contraction and
  // the sums of dy x_k as a contraction over pixels -- not built.)
  auto process = [&](float mine, const u4v& g) {
    float xv[7];
    exchange_x7(mine, xv);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vs_f32x2 zv = bs[q];
#pragma unroll
      for (int k = 0; k < 7; ++k) zv = __builtin_elementwise_fma(wr[q][k], vs_f32x2{xv[k], xv[k]}, zv);
      const vs_f32x2 dy = vs_f32x2{bf_lo(g[q]), bf_hi(g[q])} * nhwc_act_grad2<ACT>(__builtin_elementwise_fma(zv, sc[q], sh[q]));
      acc[0][q] += dy;
      acc[1][q] = __builtin_elementwise_fma(dy, (zv - mu[q]) * is[q], acc[1][q]);
#pragma unroll
      for (int k = 0; k < 7; ++k) acc[2 + k][q] = __builtin_elementwise_fma(dy, vs_f32x2{xv[k], xv[k]}, acc[2 + k][q]);
    }
  };
  float mine = 0.f;
  u4v g = {0u, 0u, 0u, 0u};
  if (p < npix) {
    mine = load_x7(x + (p - f), f, F, piece);
    g = __builtin_nontemporal_load(da + p * 8 + piece);
  }
  for (; p < npix; p += stride) {
    int fn = f + sr;
    if (fn >= F) fn -= F;
    const long long pn = p + stride;
    float mine_next = 0.f;
    u4v g_next = {0u, 0u, 0u, 0u};
    if (pn < npix) {
      mine_next = load_x7(x + (pn - fn), fn, F, piece);
      g_next = __builtin_nontemporal_load(da + pn * 8 + piece);
    }
    process(mine, g);
    mine = mine_next;
    g = g_next;
    f = fn;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = (j & 1) ? acc[k][j >> 1].y : acc[k][j >> 1].x;
      s += __shfl_xor(s, 8, 64);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane < 8) red[(wave * 9 + k) * 64 + lane * 8 + j] = s;
    }
  __syncthreads();
  // deterministic mode: acc_out is [VS_BN_STAT_SLOTS][64][9] then, the workgroups of a slot add in index order, a fold kernel sums the
  // slots in order (vs_common.h)
  const unsigned slot = blockIdx.x % VS_BN_STAT_SLOTS, rank_in_slot = blockIdx.x / VS_BN_STAT_SLOTS;
  unsigned* my_turn = turn ? turn + VS_TURN_SLOT + slot : nullptr;
  double* dst = turn ? acc_out + (size_t)slot * 576 : acc_out;
  vs_turn_begin(my_turn, rank_in_slot);
  for (int i = threadIdx.x; i < 9 * 64; i += 256) {
    const int k = i / 64, c = i - k * 64;
    atomicAdd(dst + c * 9 + k, (double)(red[(0 * 9 + k) * 64 + c] + red[(1 * 9 + k) * 64 + c] + red[(2 * 9 + k) * 64 + c] + red[(3 * 9 + k) * 64 + c]));
  }
  vs_turn_end(my_turn, rank_in_slot, (gridDim.x - slot + VS_BN_STAT_SLOTS - 1) / VS_BN_STAT_SLOTS);
}

__global__ void fold_slots_f64_kernel(const double* __restrict__ slots, int nslots, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (int k = 0; k < nslots; ++k) v += slots[(size_t)k * n + i];
  out[i] = v;
}

// parameter gradients of cnn1 + its BatchNorm from the sums above and the input moments (one thread per channel, fp64)
__global__ void nhwc_first_bwd_finalize_kernel(const double* __restrict__ acc /* [64][9] */, const double* __restrict__ mom,
                                               const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ scale,
                                               const float* __restrict__ mean, const float* __restrict__ invstd, double count, int train,
                                               float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias, float* __restrict__ dw) {
  const int c = threadIdx.x;
  if (c >= 64) return;
  const double sdy = acc[c * 9], sdyx = acc[c * 9 + 1];
  dgamma[c] = (float)sdyx;
  dbeta[c] = (float)sdy;
  const double cA = scale[c];                              // gamma * invstd
  double cB = 0.0, cC = 0.0;
  if (train) {
    cB = -cA * (double)invstd[c] * sdyx / count;
    cC = -cA * sdy / count - cB * (double)mean[c];
  }
  dbias[c] = train ? 0.f : (float)(cA * sdy);              // batch statistics: the gradient of a bias in front of them is exactly zero
  const double b = bias[c];
  for (int k = 0; k < 7; ++k) {
    double zx = b * mom[k];                                // sum z x_k
    for (int k2 = 0; k2 < 7; ++k2) zx += (double)w[c * 7 + k2] * mom_R(mom, k2, k);
    dw[c * 7 + k] = (float)(cA * acc[c * 9 + 2 + k] + cB * zx + cC * mom[k]);
  }
}

int stream_blocks(long long items_per_block_sweep, long long total, long long cap = 2048) {
  long long nb = (total + items_per_block_sweep - 1) / items_per_block_sweep;
  if (nb > cap) nb = cap;
  return nb < 1 ? 1 : (int)nb;
}
// The pure elementwise passes (no sums, no per-block flush): ONE sweep -- a workgroup per `items_per_block_sweep`, no grid-stride loop.
// [r6, calls 28-29] A 2048-block grid-stride loop over a 1.48 GB tensor (stride 8 MB) reads and writes at 4.7-5.3 TB/s; the same kernel
// launched with a workgroup per 512 pieces at 6.35 (BatchNorm apply 0.627 -> 0.467 ms, the backward pass from dy 0.918 -> 0.758), and the
// rate rises monotonically with the grid in between (tools/stream_probe.hip, profiles/r06_experiments.md section 15): workgroups dispatched in
// index order walk the tensor as one front, a strided loop keeps 2048 fronts 4 KB wide in flight.
int stream_blocks_wide(long long items_per_block_sweep, long long total) {
  long long nb = (total + items_per_block_sweep - 1) / items_per_block_sweep;
  if (nb > (1LL << 24)) nb = 1LL << 24;
  return nb < 1 ? 1 : (int)nb;
}

}  // namespace

int vs_nhwc_conv_first_impl(const float* x, const float* w, const float* scale, const float* shift, void* out,
                            int B, int T, int F, int act, double* bn_stats, hipStream_t stream, const float* bias) {
  VS_REQUIRE(x && w && scale && shift && out, "nhwc conv_first: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "nhwc conv_first: bad shape");
  const long long npix = (long long)B * T * F;
  // [r6, call 31] 2048 workgroups: 437 us, 4096: 412, 16384: 408, 65536: 508 (per-lane weights), one sweep: 1885
  const dim3 grid(stream_blocks(32, npix, (bn_stats && g_vs_turn) ? 2048 : 8192)), block(256);
  unsigned short* o = reinterpret_cast<unsigned short*>(out);
  if (bn_stats) {
    VS_REQUIRE(act == VS_ACT_NONE, "nhwc conv_first: fused statistics go with no activation");
    hipLaunchKernelGGL((nhwc_conv_first_kernel<VS_ACT_NONE, true>), grid, block, 0, stream, x, w, scale, shift, o, npix, F, bn_stats, bias);
  } else if (act == VS_ACT_NONE) hipLaunchKernelGGL((nhwc_conv_first_kernel<VS_ACT_NONE, false>), grid, block, 0, stream, x, w, scale, shift, o, npix, F, nullptr, bias);
  else if (act == VS_ACT_MISH) hipLaunchKernelGGL((nhwc_conv_first_kernel<VS_ACT_MISH, false>), grid, block, 0, stream, x, w, scale, shift, o, npix, F, nullptr, bias);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL((nhwc_conv_first_kernel<VS_ACT_RELU, false>), grid, block, 0, stream, x, w, scale, shift, o, npix, F, nullptr, bias);
  else VS_REQUIRE(false, "nhwc conv_first: unsupported activation %d", act);
  VS_LAUNCH_CHECK();
  return 0;
}

// cnn1 by recomputation: the input moments (35 doubles: S[7], R[k <= k'] row by row), the batch statistics of z1 they imply
int vs_nhwc_first_moments_impl(const float* x, int B, int T, int F, double* mom, hipStream_t stream, double* det_slots, int mom_is_zero) {
  VS_REQUIRE(x && mom && B > 0 && T > 0 && F > 0, "nhwc first_moments: bad argument");
  const long long npix = (long long)B * T * F;
  if (!mom_is_zero) VS_CHECK_HIP(hipMemsetAsync(mom, 0, sizeof(double) * kMomN, stream));
  // at most two workgroups per CU: every workgroup ends in 35 fp64 atomics on the SAME 35 addresses, which the L2 serialises (2048
  // workgroups: 90 us for a pass over 46 MB)
  long long nb = (npix + 256 * 16 - 1) / (256 * 16);
  if (nb > 512) nb = 512;
  unsigned* turn = det_slots ? g_vs_turn : nullptr;           // deterministic mode needs the caller's slot scratch ([VS_BN_STAT_SLOTS][64] doubles)
  if (turn) {      // (the full grid: a workgroup waits for the one 64 below it, which was dispatched before it -- see vs_turn_begin)
    VS_CHECK_HIP(hipMemsetAsync(det_slots, 0, sizeof(double) * VS_BN_STAT_SLOTS * 64, stream));
  }
  hipLaunchKernelGGL(nhwc_first_moments_kernel, dim3((unsigned)(nb < 1 ? 1 : nb)), dim3(256), 0, stream, x, npix, F, turn ? det_slots : mom, turn);
  if (turn) hipLaunchKernelGGL(fold_moment_slots_kernel, dim3(1), dim3(64), 0, stream, det_slots, mom);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_nhwc_first_stats_impl(const double* mom, const float* w, const float* bias, double count, double* stats, hipStream_t stream) {
  VS_REQUIRE(mom && w && bias && stats && count > 0, "nhwc first_stats: bad argument");
  hipLaunchKernelGGL(nhwc_first_stats_kernel, dim3(1), dim3(64), 0, stream, mom, w, bias, count, stats);
  VS_LAUNCH_CHECK();
  return 0;
}

// cnn1 + its BatchNorm + activation backward in ONE pass over da1 (z1 recomputed from x): dgamma, dbeta, dbias, dw [64][7].
// scratch: 64 * 9 + 35 doubles
int vs_nhwc_first_bwd_impl(const void* da, const float* x, const float* w, const float* bias, int B, int T, int F, int act, int train,
                           const float* scale, const float* shift, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, float* dbias, float* dw, double* scratch, hipStream_t stream, const double* moments, double* det_slots) {
  VS_REQUIRE(da && x && w && bias && scale && shift && mean && invstd && dgamma && dbeta && dbias && dw && scratch, "nhwc first_bwd: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "nhwc first_bwd: bad shape");
  const long long npix = (long long)B * T * F;
  double* acc = scratch;
  VS_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 64 * 9, stream));
  const double* mom = moments;                             // the forward's (vs_forward_train keeps them in the tape), or recomputed here
  if (!mom) {
    if (int rc = vs_nhwc_first_moments_impl(x, B, T, F, scratch + 64 * 9, stream)) return rc;
    mom = scratch + 64 * 9;
  }
  const u4v* g = reinterpret_cast<const u4v*>(da);
  unsigned* turn = det_slots ? g_vs_turn : nullptr;          // deterministic mode needs the caller's slot scratch ([VS_BN_STAT_SLOTS][576] doubles)
  const dim3 grid(stream_blocks(32, npix)), block(256);
  double* sums = turn ? det_slots : acc;
  if (turn) VS_CHECK_HIP(hipMemsetAsync(det_slots, 0, sizeof(double) * VS_BN_STAT_SLOTS * 576, stream));
  if (act == VS_ACT_MISH) hipLaunchKernelGGL(nhwc_first_bwd_kernel<VS_ACT_MISH>, grid, block, 0, stream, g, x, w, bias, npix, F, scale, shift, mean, invstd, sums, turn);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL(nhwc_first_bwd_kernel<VS_ACT_RELU>, grid, block, 0, stream, g, x, w, bias, npix, F, scale, shift, mean, invstd, sums, turn);
  else if (act == VS_ACT_NONE) hipLaunchKernelGGL(nhwc_first_bwd_kernel<VS_ACT_NONE>, grid, block, 0, stream, g, x, w, bias, npix, F, scale, shift, mean, invstd, sums, turn);
  else VS_REQUIRE(false, "nhwc first_bwd: unsupported activation %d", act);
  if (turn) hipLaunchKernelGGL(fold_slots_f64_kernel, dim3(3), dim3(192), 0, stream, det_slots, VS_BN_STAT_SLOTS, 576, acc);
  hipLaunchKernelGGL(nhwc_first_bwd_finalize_kernel, dim3(1), dim3(64), 0, stream, acc, mom, w, bias, scale, mean, invstd, (double)npix, train,
                     dgamma, dbeta, dbias, dw);
  VS_LAUNCH_CHECK();
  return 0;
}

// a = act(z * scale[c] + shift[c]) over [npix][64] bf16; a may alias z
int vs_nhwc_bn_apply_impl(const void* z, void* a, long long npix, int act, const float* scale, const float* shift, hipStream_t stream) {
  VS_REQUIRE(z && a && scale && shift && npix > 0, "nhwc bn_apply: bad argument");
  const long long npieces = npix * 8;
  const dim3 grid(stream_blocks_wide(512, npieces)), block(256);
  const u4v* zi = reinterpret_cast<const u4v*>(z);
  u4v* ao = reinterpret_cast<u4v*>(a);
  if (act == VS_ACT_MISH) hipLaunchKernelGGL(nhwc_bn_apply_kernel<VS_ACT_MISH>, grid, block, 0, stream, zi, ao, scale, shift, npieces);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL(nhwc_bn_apply_kernel<VS_ACT_RELU>, grid, block, 0, stream, zi, ao, scale, shift, npieces);
  else if (act == VS_ACT_NONE) hipLaunchKernelGGL(nhwc_bn_apply_kernel<VS_ACT_NONE>, grid, block, 0, stream, zi, ao, scale, shift, npieces);
  else VS_REQUIRE(false, "nhwc bn_apply: unsupported activation %d", act);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_nhwc_conv_last_impl(const void* in, const float* w, const float* scale, const float* shift, float* out,
                           int B, int T, int F, int act, hipStream_t stream, double* bn_stats,
                           const float* pre_scale, const float* pre_shift, int pre_act, void* rows_bf16, int Kp) {
  VS_REQUIRE(in && w && scale && shift && (out || rows_bf16), "nhwc conv_last: NULL argument");
  VS_REQUIRE(!rows_bf16 || (!bn_stats && !pre_scale && Kp >= 8 * F), "nhwc conv_last: the row form is the plain eval layer's (Kp = %d)", Kp);
  unsigned short* rb = reinterpret_cast<unsigned short*>(rows_bf16);
  VS_REQUIRE(!pre_scale == !pre_shift, "nhwc conv_last: pre_scale and pre_shift come together");
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "nhwc conv_last: bad shape");
  const long long nrows = (long long)B * T;
  const long long nblk = nrows * ((F + 15) / 16);
  // [r6, call 31] 2048 workgroups: 427 us, 4096: 411, 8192: 403, 16384: 444 (the statistics flush), 32768: 548; deterministic mode keeps 2048 (its
  // workgroups take turns per statistics slot: 128 turns instead of 32 cost the step 1.4 ms)
  const dim3 grid(stream_blocks(4, nblk, (bn_stats && g_vs_turn) ? 2048 : 8192)), block(256);
  const unsigned short* i = reinterpret_cast<const unsigned short*>(in);
  if (pre_scale) {     // `in` is z7: the layer below's BatchNorm + activation applied on the way in; output unactivated (+ statistics)
    VS_REQUIRE(act == VS_ACT_NONE, "nhwc conv_last: the pre-activation form writes the unactivated output");
    VS_REQUIRE(pre_act == VS_ACT_MISH || pre_act == VS_ACT_RELU, "nhwc conv_last: pre-activation %d", pre_act);
#define VS_LAST_PRE(ST_, PRE_) hipLaunchKernelGGL((nhwc_conv_last_kernel<VS_ACT_NONE, ST_, PRE_>), grid, block, 0, stream, i, w, scale, shift, out, nrows, F, bn_stats, pre_scale, pre_shift, \
                                                  (unsigned short*)nullptr, 0, ST_ ? g_vs_turn : (unsigned*)nullptr)
    if (bn_stats) { if (pre_act == VS_ACT_MISH) VS_LAST_PRE(true, VS_ACT_MISH); else VS_LAST_PRE(true, VS_ACT_RELU); }
    else { if (pre_act == VS_ACT_MISH) VS_LAST_PRE(false, VS_ACT_MISH); else VS_LAST_PRE(false, VS_ACT_RELU); }
#undef VS_LAST_PRE
    VS_LAUNCH_CHECK();
    return 0;
  }
  if (bn_stats) {      // train mode: z8 = conv + bias unactivated, statistics of it ([VS_BN_STAT_SLOTS][8][2] doubles, zeroed by the caller)
    VS_REQUIRE(act == VS_ACT_NONE, "nhwc conv_last: statistics are those of the unactivated output");
    hipLaunchKernelGGL((nhwc_conv_last_kernel<VS_ACT_NONE, true>), grid, block, 0, stream, i, w, scale, shift, out, nrows, F, bn_stats,
                       (const float*)nullptr, (const float*)nullptr, (unsigned short*)nullptr, 0, g_vs_turn);
    VS_LAUNCH_CHECK();
    return 0;
  }
  const float* nof = nullptr;
  if (act == VS_ACT_MISH) hipLaunchKernelGGL(nhwc_conv_last_kernel<VS_ACT_MISH>, grid, block, 0, stream, i, w, scale, shift, out, nrows, F, (double*)nullptr, nof, nof, rb, Kp);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL(nhwc_conv_last_kernel<VS_ACT_RELU>, grid, block, 0, stream, i, w, scale, shift, out, nrows, F, (double*)nullptr, nof, nof, rb, Kp);
  else if (act == VS_ACT_NONE) hipLaunchKernelGGL(nhwc_conv_last_kernel<VS_ACT_NONE>, grid, block, 0, stream, i, w, scale, shift, out, nrows, F, (double*)nullptr, nof, nof, rb, Kp);
  else VS_REQUIRE(false, "nhwc conv_last: unsupported activation %d", act);
  VS_LAUNCH_CHECK();
  return 0;
}

// BatchNorm + activation backward over [npix][64] bf16 (dz may alias da): parameter gradients + dz
int vs_nhwc_bn_act_bwd_impl(const void* da, const void* z, void* dz, long long npix, int act, int train,
                            const float* scale, const float* shift, const float* mean, const float* invstd,
                            float* dgamma, float* dbeta, float* dbias, double* stats /* [VS_BN_STAT_SLOTS][64][2] */, float* coef,
                            hipStream_t stream, int beside_wgrad) {
  VS_REQUIRE(da && z && dz && scale && shift && mean && invstd && stats && coef && npix > 0, "nhwc bn_act_bwd: bad argument");
  const long long npieces = npix * 8;
  int nb = stream_blocks(512, npieces);
  if (beside_wgrad) {      // both passes throttled like vs_nhwc_bn_bwd_from_dy_impl's (the round-6 re-run of the two-pass A/B: +4 ms per step, profiles/r06_experiments.md)
    const int want = 256;          // (128: +2.2 ms per step, 512: +0.2 -- round 5, call 10)
    if (want < nb) nb = want;
  }
  const dim3 grid(nb), block(256);
  const u4v* g = reinterpret_cast<const u4v*>(da);
  const u4v* zz = reinterpret_cast<const u4v*>(z);
  VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * VS_BN_STAT_SLOTS * 128, stream));
  if (act == VS_ACT_MISH) hipLaunchKernelGGL(nhwc_bn_bwd_stats_kernel<VS_ACT_MISH>, grid, block, 0, stream, g, zz, npieces, scale, shift, mean, invstd, stats);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL(nhwc_bn_bwd_stats_kernel<VS_ACT_RELU>, grid, block, 0, stream, g, zz, npieces, scale, shift, mean, invstd, stats);
  else if (act == VS_ACT_NONE) hipLaunchKernelGGL(nhwc_bn_bwd_stats_kernel<VS_ACT_NONE>, grid, block, 0, stream, g, zz, npieces, scale, shift, mean, invstd, stats);
  else VS_REQUIRE(false, "nhwc bn_act_bwd: unsupported activation %d", act);
  if (int rc = vs_bn_bwd_finalize_impl(stats, VS_BN_STAT_SLOTS, (double)npix, train, 64, scale, mean, invstd, dgamma, dbeta, dbias, coef, stream)) return rc;
  u4v* o = reinterpret_cast<u4v*>(dz);
  if (act == VS_ACT_MISH) hipLaunchKernelGGL(nhwc_bn_bwd_apply_kernel<VS_ACT_MISH>, grid, block, 0, stream, g, zz, o, npieces, scale, shift, coef);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL(nhwc_bn_bwd_apply_kernel<VS_ACT_RELU>, grid, block, 0, stream, g, zz, o, npieces, scale, shift, coef);
  else hipLaunchKernelGGL(nhwc_bn_bwd_apply_kernel<VS_ACT_NONE>, grid, block, 0, stream, g, zz, o, npieces, scale, shift, coef);
  VS_LAUNCH_CHECK();
  return 0;
}

// The same with the first pass already done by the kernel that produced dy = da * act'(.) (conv_nhwc.hip's dy
// epilogue, cnn8's backward below): stats holds the per-channel sums; parameter gradients + dz = cA dy + cB z + cC
int vs_nhwc_bn_bwd_from_dy_impl(const void* dy, const void* z, void* dz, long long npix, int train,
                                const float* scale, const float* mean, const float* invstd,
                                float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, hipStream_t stream, int rezero_doubles,
                                int beside_wgrad) {
  VS_REQUIRE(dy && z && dz && scale && mean && invstd && stats && coef && npix > 0, "nhwc bn_bwd_from_dy: bad argument");
  const long long npieces = npix * 8;
  int nb = beside_wgrad ? stream_blocks(512, npieces) : stream_blocks_wide(512, npieces);
  // beside the weight gradient on the side stream (vs_backward): ONE block per CU.  The pass then takes 1.8 ms instead of 1.2 -- still
  // inside the weight gradient's 1.95 -- and takes less from it (2.06 -> 1.94 ms per layer, -0.4 ms per step: profiles/r05_bn_finalize_ab.md)
  if (beside_wgrad) {
    const int want = 256;          // (128: +2.2 ms per step, 512: +0.2 -- round 5, call 10; [r6, call 32] a contiguous part per workgroup instead of
    if (want < nb) nb = want;      // the strided loop: the pass 1.9 -> 1.7 ms, the weight gradient beside it and the step unchanged -- not kept)
  }
  const dim3 grid(nb), block(256);
  if (int rc = vs_bn_bwd_finalize_impl(stats, VS_BN_STAT_SLOTS, (double)npix, train, 64, scale, mean, invstd, dgamma, dbeta, dbias, coef, stream,
                                       rezero_doubles)) return rc;
  hipLaunchKernelGGL(nhwc_bn_bwd_apply_kernel<VS_ACT_NONE>, grid, block, 0, stream, reinterpret_cast<const u4v*>(dy), reinterpret_cast<const u4v*>(z),
                     reinterpret_cast<u4v*>(dz), npieces, scale, scale, coef);
  VS_LAUNCH_CHECK();
  return 0;
}

// cnn1, from dy: finalize + the pass that contracts dz1 with the input (dw [64][7])
int vs_nhwc_bn_bwd_first_from_dy_impl(const void* dy, const void* z, const float* x, int B, int T, int F, int train,
                                      const float* scale, const float* mean, const float* invstd,
                                      float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc /* 448 */,
                                      hipStream_t stream) {
  VS_REQUIRE(dy && z && x && dw && stats && coef && acc, "nhwc bn_bwd_first_from_dy: NULL argument");
  const long long npix = (long long)B * T * F;
  if (int rc = vs_bn_bwd_finalize_impl(stats, VS_BN_STAT_SLOTS, (double)npix, train, 64, scale, mean, invstd, dgamma, dbeta, dbias, coef, stream)) return rc;
  VS_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 448, stream));
  const dim3 grid(stream_blocks(32, npix)), block(256);
  hipLaunchKernelGGL(nhwc_bn_bwd_first_kernel<VS_ACT_NONE>, grid, block, 0, stream, reinterpret_cast<const u4v*>(dy), reinterpret_cast<const u4v*>(z),
                     x, npix, F, scale, scale, coef, acc);
  hipLaunchKernelGGL(nhwc_cvt_f64_f32_kernel, dim3(2), dim3(256), 0, stream, acc, dw, 448);
  VS_LAUNCH_CHECK();
  return 0;
}

// cnn1: the same backward with pass 2 contracted against the input on the spot: dw [64][7] (dz1 is not written)
int vs_nhwc_bn_act_bwd_first_impl(const void* da, const void* z, const float* x, int B, int T, int F, int act, int train,
                                  const float* scale, const float* shift, const float* mean, const float* invstd,
                                  float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc /* 448 */,
                                  hipStream_t stream) {
  VS_REQUIRE(da && z && x && dw && stats && coef && acc, "nhwc bn_act_bwd_first: NULL argument");
  const long long npix = (long long)B * T * F, npieces = npix * 8;
  const u4v* g = reinterpret_cast<const u4v*>(da);
  const u4v* zz = reinterpret_cast<const u4v*>(z);
  {
    const dim3 grid(stream_blocks(512, npieces)), block(256);
    VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * VS_BN_STAT_SLOTS * 128, stream));
    if (act == VS_ACT_MISH) hipLaunchKernelGGL(nhwc_bn_bwd_stats_kernel<VS_ACT_MISH>, grid, block, 0, stream, g, zz, npieces, scale, shift, mean, invstd, stats);
    else if (act == VS_ACT_RELU) hipLaunchKernelGGL(nhwc_bn_bwd_stats_kernel<VS_ACT_RELU>, grid, block, 0, stream, g, zz, npieces, scale, shift, mean, invstd, stats);
    else if (act == VS_ACT_NONE) hipLaunchKernelGGL(nhwc_bn_bwd_stats_kernel<VS_ACT_NONE>, grid, block, 0, stream, g, zz, npieces, scale, shift, mean, invstd, stats);
    else VS_REQUIRE(false, "nhwc bn_act_bwd_first: unsupported activation %d", act);
  }
  if (int rc = vs_bn_bwd_finalize_impl(stats, VS_BN_STAT_SLOTS, (double)npix, train, 64, scale, mean, invstd, dgamma, dbeta, dbias, coef, stream)) return rc;
  VS_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 448, stream));
  const dim3 grid(stream_blocks(32, npix)), block(256);
  if (act == VS_ACT_MISH) hipLaunchKernelGGL(nhwc_bn_bwd_first_kernel<VS_ACT_MISH>, grid, block, 0, stream, g, zz, x, npix, F, scale, shift, coef, acc);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL(nhwc_bn_bwd_first_kernel<VS_ACT_RELU>, grid, block, 0, stream, g, zz, x, npix, F, scale, shift, coef, acc);
  else hipLaunchKernelGGL(nhwc_bn_bwd_first_kernel<VS_ACT_NONE>, grid, block, 0, stream, g, zz, x, npix, F, scale, shift, coef, acc);
  hipLaunchKernelGGL(nhwc_cvt_f64_f32_kernel, dim3(2), dim3(256), 0, stream, acc, dw, 448);
  VS_LAUNCH_CHECK();
  return 0;
}

// cnn8 backward: dz8 [B][T][8][F] fp32, a7 [B][T][F][64] bf16 -> din (bf16, same layout as a7) and dw [8][64];
// part: VS_NHWC_LAST_BWD_BLOCKS x 512 floats of scratch
// z7 != NULL: din is dy7 = da7 * act'(z7 * bn_scale + bn_shift) and bn_stats ([VS_BN_STAT_SLOTS][64][2] doubles, zeroed
// by the caller) receives the sums the BatchNorm backward of cnn7 starts from (vs_nhwc_bn_bwd_from_dy_impl)
int vs_nhwc_conv_last_bwd_impl(const float* dz8, const float* w, const void* a7, void* din, float* part, float* dw,
                               int B, int T, int F, const void* z7, int act, const float* bn_scale, const float* bn_shift,
                               const float* bn_mean, const float* bn_invstd, double* bn_stats, hipStream_t stream) {
  VS_REQUIRE(dz8 && w && din && part && dw, "nhwc conv_last_bwd: NULL argument");
  VS_REQUIRE(a7 || z7, "nhwc conv_last_bwd: a7 may only be NULL in the dy form (it is then recomputed from z7)");
  VS_REQUIRE(!z7 || (bn_scale && bn_shift && bn_mean && bn_invstd && bn_stats), "nhwc conv_last_bwd: the dy form needs the BatchNorm constants and statistics slots");
  const long long npix = (long long)B * T * F;
  long long nb = (npix + 31) / 32;
  if (nb > VS_NHWC_LAST_BWD_BLOCKS) nb = VS_NHWC_LAST_BWD_BLOCKS;
  const LastBwdBn bn{reinterpret_cast<const u4v*>(z7), bn_scale, bn_shift, bn_mean, bn_invstd, bn_stats, z7 ? g_vs_turn : nullptr};
  const dim3 grid((unsigned)nb), block(256);
  const u4v* a = reinterpret_cast<const u4v*>(a7);
  u4v* o = reinterpret_cast<u4v*>(din);
  if (!z7) hipLaunchKernelGGL(nhwc_conv_last_bwd_kernel<-1>, grid, block, 0, stream, dz8, w, a, o, part, npix, F, bn);
  else if (act == VS_ACT_MISH && !a7) hipLaunchKernelGGL((nhwc_conv_last_bwd_kernel<VS_ACT_MISH, true>), grid, block, 0, stream, dz8, w, a, o, part, npix, F, bn);
  else if (act == VS_ACT_RELU && !a7) hipLaunchKernelGGL((nhwc_conv_last_bwd_kernel<VS_ACT_RELU, true>), grid, block, 0, stream, dz8, w, a, o, part, npix, F, bn);
  else if (act == VS_ACT_MISH) hipLaunchKernelGGL(nhwc_conv_last_bwd_kernel<VS_ACT_MISH>, grid, block, 0, stream, dz8, w, a, o, part, npix, F, bn);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL(nhwc_conv_last_bwd_kernel<VS_ACT_RELU>, grid, block, 0, stream, dz8, w, a, o, part, npix, F, bn);
  else VS_REQUIRE(false, "nhwc conv_last_bwd: dy form for activation %d", act);
  VS_LAUNCH_CHECK();
  return vs_reduce_partials_impl(part, (int)nb, 512, dw, stream);
}

// ---- the same two edges for the channels-last split-f16 forward (conv_nhwc_f16x3.hip: VS_MATH_F16X3, eval BatchNorm) ---------------
// cnn1 computes in fp32 (7 FMAs per output) and writes y * s_y as hi / lo f16 planes; s_y comes from vs_nhwc_first_plan_impl's bound.
// cnn8 reads the planes of cnn7's output and contracts them on the f16 matrix pipe with its 8 x 64 weights split the same way.
namespace {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

// out_scale2 <- {s, 1 / s}, s = the power of two that maps max_c (|scale_c| sum_j |w_cj| max|x| + |shift_c|) (>= max |y|) into [2^14, 2^15);
// w = [nco][taps] (cnn1: 64 x 7, cnn8: 8 x 64)
__global__ void nhwc_first_plan_kernel(const unsigned* __restrict__ amax_in, int n_amax, const float* __restrict__ w, const float* __restrict__ scale,
                                       const float* __restrict__ shift, float* __restrict__ out_scale2, int nco, int taps) {
  const int c = threadIdx.x;                                  // 64 threads
  unsigned mb = 0;
  for (int i = c; i < n_amax; i += 64) mb = amax_in[i] > mb ? amax_in[i] : mb;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned other = __shfl_xor(mb, o, 64); mb = other > mb ? other : mb; }
  float bound = 0.3125f;
  if (c < nco) {
    float l1 = 0.f;
    for (int k = 0; k < taps; ++k) l1 += fabsf(w[c * taps + k]);
    bound = fmaxf(fabsf(scale[c]) * l1 * __uint_as_float(mb) + fabsf(shift[c]), 0.3125f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o, 64));
  if (c == 0) {
    float s = 1.f, si = 1.f;
    if (bound > 0.f && bound < 3.0e38f) {
      int e = 0;
      (void)frexpf(bound, &e);
      int k = 15 - e;
      k = k > 100 ? 100 : (k < -100 ? -100 : k);
      s = ldexpf(1.f, k);
      si = ldexpf(1.f, -k);
    }
    out_scale2[0] = s;
    out_scale2[1] = si;
  }
}

// max |x| of the path's input (any 4-byte alignment: the mixture may be a slice of a batch), folded into *amax (bits of a float >= 0);
// one atomic per workgroup (one per wave made 8192 of them queue up on the one address: 0.1 ms for 46 MB)
__global__ __launch_bounds__(256)
void absmax_any_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ amax) {
  __shared__ float red[4];
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax(amax, __float_as_uint(m));
  }
}

template <int ACT>
__global__ __launch_bounds__(256)
void nhwc_conv_first_split_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                  const float* __restrict__ shift, const float* __restrict__ out_scale2, unsigned short* __restrict__ out_hi,
                                  unsigned short* __restrict__ out_lo, unsigned* __restrict__ amax_out, long long npix, int F) {
  const int piece = threadIdx.x & 7;
  vs_f32x2 wr[4][7], sc[4], sh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c0 = piece * 8 + 2 * q;
    sc[q] = vs_f32x2{scale[c0], scale[c0 + 1]};
    sh[q] = vs_f32x2{shift[c0], shift[c0 + 1]};
#pragma unroll
    for (int k = 0; k < 7; ++k) wr[q][k] = vs_f32x2{w[c0 * 7 + k], w[(c0 + 1) * 7 + k]};
  }
  const float sy = out_scale2[0];
  float am = 0.f;
  const long long stride = (long long)gridDim.x * 32;
  const int sr = (int)(stride % F);
  long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  int f = (int)(p % F);
  float mine = p < npix ? load_x7(x + (p - f), f, F, piece) : 0.f;
  for (; p < npix; p += stride) {
    int fn = f + sr;
    if (fn >= F) fn -= F;
    const long long pn = p + stride;
    const float mine_next = pn < npix ? load_x7(x + (pn - fn), fn, F, piece) : 0.f;
    float xv[7];
    exchange_x7(mine, xv);
    mine = mine_next;
    f = fn;
    u4v ph, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vs_f32x2 s2 = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 7; ++k) s2 = __builtin_elementwise_fma(wr[q][k], vs_f32x2{xv[k], xv[k]}, s2);
      vs_f32x2 y = vs_act_fast2<ACT>(__builtin_elementwise_fma(s2, sc[q], sh[q]));
      am = fmaxf(am, fmaxf(fabsf(y.x), fabsf(y.y)));
      y = y * sy;
      const unsigned hb = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(y.x, y.y));
      const h2v h = __builtin_bit_cast(h2v, hb);
      ph[q] = hb;
      pl[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(y.x - (float)h[0], y.y - (float)h[1]));
    }
    __builtin_nontemporal_store(ph, reinterpret_cast<u4v*>(out_hi + p * 64 + piece * 8));
    __builtin_nontemporal_store(pl, reinterpret_cast<u4v*>(out_lo + p * 64 + piece * 8));
  }
  vs_absmax_commit(am, amax_out);
}

// out[b][t][co][f] = act(scale[co] * sum_ci w[co][ci] x[b][t][f][ci] + shift[co]), co < 8, x = (in_hi + in_lo) / s_x.
// One wave = 16 pixels per step: B operands = the two planes' pixels as they lie in memory, A = the weights (rows 8..15 repeat rows 0..7)
// split into hi / lo after a power-of-two scale found here (512 values: one wave reduction); three products, fp32 accumulate.
// ROWS: the output goes out as the LSTM input GEMM's A operand instead -- hi / lo f16 rows [nrows][Kp] of out * out_scale2[0] (element
// co * F + f of row (b, t): the feature order of the fp32 form), columns 8 F .. Kp zeroed by the workgroups that own a row's first block.
template <int ACT, bool ROWS>
__global__ __launch_bounds__(256)
void nhwc_conv_last_split_kernel(const unsigned short* __restrict__ in_hi, const unsigned short* __restrict__ in_lo, const float* __restrict__ in_scale2,
                                 const float* __restrict__ w, const float* __restrict__ scale, const float* __restrict__ shift,
                                 float* __restrict__ out, long long nrows, int F,
                                 unsigned short* __restrict__ row_hi, unsigned short* __restrict__ row_lo, int Kp, const float* __restrict__ out_scale2) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
  float wv[2][8], wmax = 0.f;
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wv[kc][j] = w[(n & 7) * 64 + kc * 32 + g * 8 + j];      // A rows 8..15 repeat 0..7: accumulator rows g >= 2 repeat g - 2
      wmax = fmaxf(wmax, fabsf(wv[kc][j]));
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
  float sw = 1.f, swi = 1.f;
  if (wmax > 0.f && wmax < 3.0e38f) {
    int e = 0;
    (void)frexpf(wmax, &e);
    sw = ldexpf(1.f, 10 - e);
    swi = ldexpf(1.f, e - 10);
  }
  h8v wh[2], wl[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = wv[kc][j] * sw;
      wh[kc][j] = (_Float16)v;
      wl[kc][j] = (_Float16)(v - (float)wh[kc][j]);
    }
  const float inv = in_scale2[1] * swi;
  float sc[4], sh[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = (g * 4 + r) & 7;
    sc[r] = scale[co] * inv;
    sh[r] = shift[co];
  }
  const float so = ROWS ? out_scale2[0] : 1.f;
  const int blocks_per_row = (F + 15) >> 4;
  const long long nblk = nrows * blocks_per_row;
  const long long wstride = (long long)gridDim.x * 4;
  const long long sq = wstride / blocks_per_row;
  const int sr = (int)(wstride - sq * blocks_per_row);
  // workgroups go round the 8 XCDs: the ones of an XCD take neighbouring blocks, so that the 32- and 64-byte pieces of an output
  // line meet in ONE L2 instead of leaving eight of them as partial writes
  const int per_xcd = gridDim.x >> 3;
  const int vblock = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  long long blk = (long long)vblock * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  long long row = blk / blocks_per_row;
  int bc = (int)(blk - row * blocks_per_row);
  const u4v z4 = {0u, 0u, 0u, 0u};
  // the next block's pixels are loaded before this one is contracted (the loop is otherwise load -> wait -> 6 MFMAs -> store)
  u4v h0 = z4, h1 = z4, l0 = z4, l1 = z4;
  auto fetch = [&](long long r, int c, u4v& a0, u4v& a1, u4v& b0, u4v& b1) {
    const int f = c * 16 + n;
    if (f < F) {
      const size_t e0 = (size_t)((r * F + f) << 6);
      const u4v* sh_ = reinterpret_cast<const u4v*>(in_hi + e0) + g;
      const u4v* sl_ = reinterpret_cast<const u4v*>(in_lo + e0) + g;
      a0 = __builtin_nontemporal_load(sh_); a1 = __builtin_nontemporal_load(sh_ + 4);
      b0 = __builtin_nontemporal_load(sl_); b1 = __builtin_nontemporal_load(sl_ + 4);
    } else {
      a0 = a1 = b0 = b1 = z4;
    }
  };
  if (blk < nblk) fetch(row, bc, h0, h1, l0, l1);
  for (; blk < nblk; blk += wstride) {
    long long rown = row + sq;
    int bcn = bc + sr;
    if (bcn >= blocks_per_row) { bcn -= blocks_per_row; ++rown; }
    u4v nh0 = z4, nh1 = z4, nl0 = z4, nl1 = z4;
    if (blk + wstride < nblk) fetch(rown, bcn, nh0, nh1, nl0, nl1);
    const int f = bc * 16 + n;
    const bool ok = f < F;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[0], __builtin_bit_cast(h8v, h0), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[1], __builtin_bit_cast(h8v, h1), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[0], __builtin_bit_cast(h8v, l0), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[1], __builtin_bit_cast(h8v, l1), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[0], __builtin_bit_cast(h8v, h0), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[1], __builtin_bit_cast(h8v, h1), c, 0, 0, 0);
    if (ROWS) {
      if (ok) {                                                 // lanes g < 2 write the hi halves, their repeats (g >= 2) the lo halves
        unsigned short* dst = (g < 2 ? row_hi : row_lo) + (size_t)row * Kp + (size_t)((g & 1) * 4) * F + f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = vs_act_fast<ACT>(fmaf(c[r], sc[r], sh[r])) * so;
          const _Float16 vh = __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(v, 0.f))[0];
          const _Float16 vl = __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(v - (float)vh, 0.f))[0];
          dst[(size_t)r * F] = __builtin_bit_cast(unsigned short, g < 2 ? vh : vl);
        }
      }
      if (bc == 0) {                                            // this wave owns the row's padding
        for (int k = 8 * F + lane; k < Kp; k += 64) {
          row_hi[(size_t)row * Kp + k] = 0;
          row_lo[(size_t)row * Kp + k] = 0;
        }
      }
    } else if (ok && g < 2) {
      float* o = out + (row * 8 + g * 4) * F + f;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(size_t)r * F] = vs_act_fast<ACT>(fmaf(c[r], sc[r], sh[r]));
    }
    h0 = nh0; h1 = nh1; l0 = nl0; l1 = nl1;
    row = rown; bc = bcn;
  }
}

}  // namespace

int vs_absmax_any_impl(const float* x, long long n, unsigned* amax, hipStream_t stream) {
  VS_REQUIRE(x && amax && n > 0, "absmax: bad argument");
  hipLaunchKernelGGL(absmax_any_kernel, dim3(stream_blocks(256 * 16, n) > 1024 ? 1024 : stream_blocks(256 * 16, n)), dim3(256), 0, stream, x, n, amax);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_nhwc_first_plan_impl(const unsigned* amax_in, int n_amax, const float* w, const float* scale, const float* shift, float* out_scale2,
                            hipStream_t stream) {
  VS_REQUIRE(amax_in && n_amax > 0 && w && scale && shift && out_scale2, "nhwc first_plan: bad argument");
  hipLaunchKernelGGL(nhwc_first_plan_kernel, dim3(1), dim3(64), 0, stream, amax_in, n_amax, w, scale, shift, out_scale2, 64, 7);
  VS_LAUNCH_CHECK();
  return 0;
}

// the same for cnn8 (8 output channels x 64 taps) when its output is written as the LSTM input GEMM's split operand
int vs_nhwc_last_plan_impl(const unsigned* amax_in, int n_amax, const float* w, const float* scale, const float* shift, float* out_scale2,
                           hipStream_t stream) {
  VS_REQUIRE(amax_in && n_amax > 0 && w && scale && shift && out_scale2, "nhwc last_plan: bad argument");
  hipLaunchKernelGGL(nhwc_first_plan_kernel, dim3(1), dim3(64), 0, stream, amax_in, n_amax, w, scale, shift, out_scale2, 8, 64);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_nhwc_conv_first_split_impl(const float* x, const float* w, const float* scale, const float* shift, const float* out_scale2,
                                  void* out_hi, void* out_lo, unsigned* amax_out, int B, int T, int F, int act, hipStream_t stream) {
  VS_REQUIRE(x && w && scale && shift && out_scale2 && out_hi && out_lo, "nhwc conv_first_split: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "nhwc conv_first_split: bad shape");
  const long long npix = (long long)B * T * F;
  const dim3 grid(stream_blocks(32, npix, 8192)), block(256);
  unsigned short* oh = reinterpret_cast<unsigned short*>(out_hi);
  unsigned short* ol = reinterpret_cast<unsigned short*>(out_lo);
  if (act == VS_ACT_NONE) hipLaunchKernelGGL((nhwc_conv_first_split_kernel<VS_ACT_NONE>), grid, block, 0, stream, x, w, scale, shift, out_scale2, oh, ol, amax_out, npix, F);
  else if (act == VS_ACT_MISH) hipLaunchKernelGGL((nhwc_conv_first_split_kernel<VS_ACT_MISH>), grid, block, 0, stream, x, w, scale, shift, out_scale2, oh, ol, amax_out, npix, F);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL((nhwc_conv_first_split_kernel<VS_ACT_RELU>), grid, block, 0, stream, x, w, scale, shift, out_scale2, oh, ol, amax_out, npix, F);
  else VS_REQUIRE(false, "nhwc conv_first_split: unsupported activation %d", act);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_nhwc_conv_last_split_impl(const void* in_hi, const void* in_lo, const float* in_scale2, const float* w, const float* scale,
                                 const float* shift, float* out, int B, int T, int F, int act, hipStream_t stream,
                                 void* row_hi, void* row_lo, int Kp, const float* out_scale2) {
  VS_REQUIRE(in_hi && in_lo && in_scale2 && w && scale && shift && (out || row_hi), "nhwc conv_last_split: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "nhwc conv_last_split: bad shape");
  VS_REQUIRE(!row_hi || (row_lo && out_scale2 && Kp >= 8 * F), "nhwc conv_last_split: bad row-form arguments (Kp = %d)", Kp);
  const long long nrows = (long long)B * T;
  const long long nblk = nrows * ((F + 15) / 16);
  const dim3 grid(stream_blocks(4, nblk, 8192)), block(256);
  const unsigned short* ih = reinterpret_cast<const unsigned short*>(in_hi);
  const unsigned short* il = reinterpret_cast<const unsigned short*>(in_lo);
  unsigned short* rh = reinterpret_cast<unsigned short*>(row_hi);
  unsigned short* rl = reinterpret_cast<unsigned short*>(row_lo);
#define VS_LAST_SPLIT(A)                                                                                                                   \
  do {                                                                                                                                     \
    if (rh) hipLaunchKernelGGL((nhwc_conv_last_split_kernel<A, true>), grid, block, 0, stream, ih, il, in_scale2, w, scale, shift, out, nrows, F, rh, rl, Kp, out_scale2); \
    else hipLaunchKernelGGL((nhwc_conv_last_split_kernel<A, false>), grid, block, 0, stream, ih, il, in_scale2, w, scale, shift, out, nrows, F, rh, rl, Kp, out_scale2);  \
  } while (0)
  if (act == VS_ACT_NONE) VS_LAST_SPLIT(VS_ACT_NONE);
  else if (act == VS_ACT_MISH) VS_LAST_SPLIT(VS_ACT_MISH);
  else if (act == VS_ACT_RELU) VS_LAST_SPLIT(VS_ACT_RELU);
  else VS_REQUIRE(false, "nhwc conv_last_split: unsupported activation %d", act);
#undef VS_LAST_SPLIT
  VS_LAUNCH_CHECK();
  return 0;
}
