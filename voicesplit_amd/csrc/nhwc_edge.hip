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
__device__ __forceinline__ void fold_channel_sums(float (&v)[NV][8], double* __restrict__ stats, float* lds /* [4 waves][NV][64] */) {
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
  for (int i = threadIdx.x; i < NV * 64; i += blockDim.x) {
    const int q = i / 64, c = i - q * 64;
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += lds[(w * NV + q) * 64 + c];
    atomicAdd(stats + (size_t)(blockIdx.x % VS_BN_STAT_SLOTS) * 128 + c * 2 + q, (double)s);
  }
}

// ---- cnn1 ---------------------------------------------------------------------------------------------------
// out[b][t][f][co] = act(scale[co] * sum_j w[co][j] x[b][t][f+j-3] + shift[co]); STATS: sum / sum of squares of out
template <int ACT, bool STATS>
__global__ __launch_bounds__(256)
void nhwc_conv_first_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                            const float* __restrict__ shift, unsigned short* __restrict__ out, long long npix, int F,
                            double* __restrict__ stats) {
  __shared__ float red[4 * 2 * 64];
  const int piece = threadIdx.x & 7;                      // channels 8*piece .. 8*piece+7
  float wr[8][7], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[piece * 8 + j];
    sh[j] = shift[piece * 8 + j];
#pragma unroll
    for (int k = 0; k < 7; ++k) wr[j][k] = w[(piece * 8 + j) * 7 + k];
  }
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  const long long stride = (long long)gridDim.x * 32;     // pixels per sweep of the grid
  for (long long p = (long long)blockIdx.x * 32 + (threadIdx.x >> 3); p < npix; p += stride) {
    const int f = (int)(p % F);
    const float* xr = x + (p - f);
    float xv[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int ff = f + k - 3;
      xv[k] = (ff >= 0 && ff < F) ? xr[ff] : 0.f;
    }
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 7; ++k) s = fmaf(wr[j][k], xv[k], s);
      y[j] = vs_act_fast<ACT>(fmaf(s, sc[j], sh[j]));
      if (STATS) { acc[0][j] += y[j]; acc[1][j] = fmaf(y[j], y[j], acc[1][j]); }
    }
    const u4v pk = {vs_pack_bf16(y[0], y[1]), vs_pack_bf16(y[2], y[3]), vs_pack_bf16(y[4], y[5]), vs_pack_bf16(y[6], y[7])};
    *reinterpret_cast<u4v*>(out + p * 64 + piece * 8) = pk;
  }
  if (STATS) fold_channel_sums<2>(acc, stats, red);
}

// ---- BatchNorm + activation apply -------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(256)
void nhwc_bn_apply_kernel(const u4v* __restrict__ z, u4v* __restrict__ a, const float* __restrict__ scale,
                          const float* __restrict__ shift, long long npieces) {
  const int piece = threadIdx.x & 7;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = scale[piece * 8 + j]; sh[j] = shift[piece * 8 + j]; }
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  auto apply = [&](const u4v v) {
    u4v o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float lo = vs_act_fast<ACT>(fmaf(bf_lo(v[q]), sc[2 * q], sh[2 * q]));
      const float hi = vs_act_fast<ACT>(fmaf(bf_hi(v[q]), sc[2 * q + 1], sh[2 * q + 1]));
      o[q] = vs_pack_bf16(lo, hi);
    }
    return o;
  };
  for (; i + stride < npieces; i += 2 * stride) {         // two pieces in flight per lane
    const u4v v0 = __builtin_nontemporal_load(z + i), v1 = __builtin_nontemporal_load(z + i + stride);
    a[i] = apply(v0);
    a[i + stride] = apply(v1);
  }
  if (i < npieces) a[i] = apply(__builtin_nontemporal_load(z + i));
}

// ---- cnn8 + transpose/view ---------------------------------------------------------------------------------------
// out[b][t][co][f] = act(scale[co] * sum_ci w[co][ci] in[b][t][f][ci] + shift[co]), co < 8.  One wave = 16 pixels of a
// row per step: the B operand of v_mfma_f32_16x16x32_bf16 is the pixels as they lie in memory (lane = (pixel, 8
// channels) = one 16-byte load), A = the 8 x 64 weights zero-padded to 16 rows; C rows 0..7 -> 8 feature rows.
template <int ACT>
__global__ __launch_bounds__(256)
void nhwc_conv_last_kernel(const unsigned short* __restrict__ in, const float* __restrict__ w, const float* __restrict__ scale,
                           const float* __restrict__ shift, float* __restrict__ out, long long nrows /* B*T */, int F) {
  const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
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
  for (long long blk = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); blk < nblk; blk += wstride) {
    const long long row = blk / blocks_per_row;
    const int f = (int)(blk - row * blocks_per_row) * 16 + n;
    const bool ok = f < F;
    const u4v* src = reinterpret_cast<const u4v*>(in + ((row * F + (ok ? f : 0)) << 6)) + g;
    const u4v b0 = ok ? src[0] : u4v{0u, 0u, 0u, 0u}, b1 = ok ? src[4] : u4v{0u, 0u, 0u, 0u};
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0], __builtin_bit_cast(vs_bf16x8, b0), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1], __builtin_bit_cast(vs_bf16x8, b1), c, 0, 0, 0);
    if (ok && g < 2) {
      float* o = out + (row * 8 + g * 4) * F + f;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(size_t)r * F] = vs_act_fast<ACT>(fmaf(c[r], sc[r], sh[r]));
    }
  }
}

int stream_blocks(long long items_per_block_sweep, long long total) {
  long long nb = (total + items_per_block_sweep - 1) / items_per_block_sweep;
  if (nb > 2048) nb = 2048;
  return nb < 1 ? 1 : (int)nb;
}

}  // namespace

int vs_nhwc_conv_first_impl(const float* x, const float* w, const float* scale, const float* shift, void* out,
                            int B, int T, int F, int act, double* bn_stats, hipStream_t stream) {
  VS_REQUIRE(x && w && scale && shift && out, "nhwc conv_first: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "nhwc conv_first: bad shape");
  const long long npix = (long long)B * T * F;
  const dim3 grid(stream_blocks(32, npix)), block(256);
  unsigned short* o = reinterpret_cast<unsigned short*>(out);
  if (bn_stats) {
    VS_REQUIRE(act == VS_ACT_NONE, "nhwc conv_first: fused statistics go with no activation");
    hipLaunchKernelGGL((nhwc_conv_first_kernel<VS_ACT_NONE, true>), grid, block, 0, stream, x, w, scale, shift, o, npix, F, bn_stats);
  } else if (act == VS_ACT_NONE) hipLaunchKernelGGL((nhwc_conv_first_kernel<VS_ACT_NONE, false>), grid, block, 0, stream, x, w, scale, shift, o, npix, F, nullptr);
  else if (act == VS_ACT_MISH) hipLaunchKernelGGL((nhwc_conv_first_kernel<VS_ACT_MISH, false>), grid, block, 0, stream, x, w, scale, shift, o, npix, F, nullptr);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL((nhwc_conv_first_kernel<VS_ACT_RELU, false>), grid, block, 0, stream, x, w, scale, shift, o, npix, F, nullptr);
  else VS_REQUIRE(false, "nhwc conv_first: unsupported activation %d", act);
  VS_LAUNCH_CHECK();
  return 0;
}

// a = act(z * scale[c] + shift[c]) over [npix][64] bf16; a may alias z
int vs_nhwc_bn_apply_impl(const void* z, void* a, long long npix, int act, const float* scale, const float* shift, hipStream_t stream) {
  VS_REQUIRE(z && a && scale && shift && npix > 0, "nhwc bn_apply: bad argument");
  const long long npieces = npix * 8;
  const dim3 grid(stream_blocks(512, npieces)), block(256);
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
                           int B, int T, int F, int act, hipStream_t stream) {
  VS_REQUIRE(in && w && scale && shift && out, "nhwc conv_last: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "nhwc conv_last: bad shape");
  const long long nrows = (long long)B * T;
  const long long nblk = nrows * ((F + 15) / 16);
  const dim3 grid(stream_blocks(4, nblk)), block(256);
  const unsigned short* i = reinterpret_cast<const unsigned short*>(in);
  if (act == VS_ACT_MISH) hipLaunchKernelGGL(nhwc_conv_last_kernel<VS_ACT_MISH>, grid, block, 0, stream, i, w, scale, shift, out, nrows, F);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL(nhwc_conv_last_kernel<VS_ACT_RELU>, grid, block, 0, stream, i, w, scale, shift, out, nrows, F);
  else if (act == VS_ACT_NONE) hipLaunchKernelGGL(nhwc_conv_last_kernel<VS_ACT_NONE>, grid, block, 0, stream, i, w, scale, shift, out, nrows, F);
  else VS_REQUIRE(false, "nhwc conv_last: unsupported activation %d", act);
  VS_LAUNCH_CHECK();
  return 0;
}
