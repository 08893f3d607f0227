// The two bandwidth-bound ends of the conv stack and the BatchNorm fold.
//
//   cnn1: ZeroPad2d((3,3,0,0)) + Conv2d(1->64,(1,7)) + BN + act   models/voicesplit/model.py:17-19
//   cnn8: Conv2d(64->8,(1,1)) + BN + act, written directly in the [B,T,8*F] layout that
//         x.transpose(1,2).contiguous().view(B,T,-1) produces          models/voicesplit/model.py:51-52,72-74
//
// Both are ~3.5 FLOP/B: one thread per (t,f) pixel, lanes along f so every
// channel plane is read/written in coalesced 256-byte wave rows; the per-channel weights are
// wave-uniform and live in SGPRs.
#include "vs_common.h"

namespace {

// scale = gamma / sqrt(var + eps);  shift = beta + (conv_bias - mean) * scale
// so that  BN(conv + bias) = conv * scale + shift      (nn.BatchNorm2d eval, eps 1e-5)
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var,
                               const float* __restrict__ conv_bias, float eps, int C,
                               float* __restrict__ scale, float* __restrict__ shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = gamma[c] / sqrtf(var[c] + eps);
  float b = conv_bias ? conv_bias[c] : 0.f;
  scale[c] = s;
  shift[c] = beta[c] + (b - mean[c]) * s;
}

// cnn1 is HBM-bound (write 2.96 GB).  A thread owns 4 consecutive pixels of the plane and writes them
// as one 16-byte store per channel (1 KB per wave instruction instead of 256 B; planes are only
// 4-byte aligned -- T*F is odd -- and the stores are issued unaligned): 0.98 -> 0.80 ms.
typedef float edge_f4 __attribute__((ext_vector_type(4)));

template <int ACT>
__global__ __launch_bounds__(256)
void conv_first_kernel(const float* __restrict__ x,      // [B][T][F]
                       const float* __restrict__ w,      // [64][7]
                       const float* __restrict__ scale, const float* __restrict__ shift,
                       float* __restrict__ out,          // [B][64][T][F]
                       int T, int F, unsigned* amax_out) {
  const int plane = T * F;
  const int pix0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int b = blockIdx.y;
  float m = 0.f;
  if (pix0 < plane) {
    const int np = plane - pix0 < 4 ? plane - pix0 : 4;
    float v[4][7];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pix = pix0 + (e < np ? e : 0);
      const int t = pix / F;
      const int f = pix - t * F;
      const float* row = x + (size_t)b * plane + (size_t)t * F;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const int ff = f + k - 3;
        v[e][k] = (ff >= 0 && ff < F) ? row[ff] : 0.f;
      }
    }
    float* o = out + (size_t)b * 64 * plane + pix0;
#pragma unroll 2
    for (int c = 0; c < 64; ++c) {
      edge_f4 y;
      const float sc = scale[c], sh = shift[c];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 7; ++k) a = fmaf(w[c * 7 + k], v[e][k], a);
        y[e] = vs_act_fast<ACT>(fmaf(a, sc, sh));     // Mish on v_exp/v_rcp (3e-7 relative), as in the conv epilogues
        if (e < np) m = fmaxf(m, fabsf(y[e]));
      }
      float* q = o + (size_t)c * plane;
      if (np == 4) {
        __builtin_memcpy(q, &y, 16);
      } else {
        for (int e = 0; e < np; ++e) q[e] = y[e];
      }
    }
  }
  vs_absmax_commit(m, amax_out);
}

// cnn8 reads 64 planes per pixel and writes 8 short rows: one pixel per thread (the 4-pixel form of
// the kernels around it measured the same here: 0.68 vs 0.70 ms)
template <int ACT>
__global__ __launch_bounds__(256)
void conv_last_kernel(const float* __restrict__ in,      // [B][64][T][F]
                      const float* __restrict__ w,       // [8][64]
                      const float* __restrict__ scale, const float* __restrict__ shift,
                      float* __restrict__ out,           // [B][T][8][F]
                      int T, int F) {
  const int plane = T * F;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= plane) return;
  const int t = pix / F;
  const int f = pix - t * F;
  const float* src = in + (size_t)b * 64 * plane + pix;
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = 0.f;
#pragma unroll 8
  for (int c = 0; c < 64; ++c) {
    const float v = src[(size_t)c * plane];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = fmaf(w[o * 64 + c], v, acc[o]);
  }
  float* dst = out + ((size_t)b * T + t) * 8 * F + f;
#pragma unroll
  for (int o = 0; o < 8; ++o) dst[(size_t)o * F] = vs_act<ACT>(fmaf(acc[o], scale[o], shift[o]));
}


// ---- train-mode BatchNorm (model.train(), train.py:84): batch statistics over (B,T,F) --------
// stats[c] = {sum, sum of squares} in double.  Rows [R][L], channel = r % C (NCHW: R = B*C,
// L = T*F); grid (blocks per channel, C): a block walks every gridDim.x-th (row, chunk) item of its
// channel (vs_walk_chunk), fp32 partials per item, double across items, two atomics per block.
__global__ __launch_bounds__(256)
void bn_stats_kernel(const float* __restrict__ x, int C, long long rows_c, int L, double* __restrict__ stats) {
  const int c = blockIdx.y;
  const int gx = vs_row_chunks(L);
  double ds = 0.0, dq = 0.0;
  for (long long it = blockIdx.x; it < rows_c * gx; it += gridDim.x) {
    const float* src = x + (c + (long long)C * (it / gx)) * L;
    float s = 0.f, q = 0.f;
    vs_walk_chunk(L, vs_row_phase(src), (int)(it % gx),
                  [&](int i, auto w) { return vs_ldv<decltype(w)::value>(src + i); },
                  [&](int, auto v) {
#pragma unroll
                    for (int e = 0; e < decltype(v)::N; ++e) { s += v.v[e]; q = fmaf(v.v[e], v.v[e], q); }
                  });
    ds += s;
    dq += q;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ds += __shfl_down(ds, o, 64);
    dq += __shfl_down(dq, o, 64);
  }
  __shared__ double sh[8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[2 * w] = ds; sh[2 * w + 1] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ds = sh[0] + sh[2] + sh[4] + sh[6];
    dq = sh[1] + sh[3] + sh[5] + sh[7];
    atomicAdd(&stats[2 * c], ds);
    atomicAdd(&stats[2 * c + 1], dq);
  }
}

// mean/var -> scale/shift for the apply pass, running buffers updated like nn.BatchNorm2d
// (momentum 0.1, unbiased variance into running_var).
__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, int C,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = stats[2 * c] / count;
  double var = stats[2 * c + 1] / count - mean * mean;
  if (var < 0) var = 0;
  const float is = 1.0f / sqrtf((float)var + eps);
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (mean_out) mean_out[c] = (float)mean;          // saved for the backward pass
  if (invstd_out) invstd_out[c] = is;
  const double unb = count > 1 ? var * (count / (count - 1)) : var;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
}

// y = act(x*scale[c] + shift[c]) over rows [R][L], channel = r % C; y may alias x (in place).
// Same item walk as bn_stats_kernel.  Mish through v_exp/v_rcp (vs_mish_fast): with the accurate
// expf/divide sequence the pass was VALU-bound, not HBM-bound.
template <int ACT>
__global__ __launch_bounds__(256)
void bn_apply_kernel(const float* x, float* y, const float* __restrict__ scale, const float* __restrict__ shift,
                     int C, long long rows_c, int L, unsigned* amax_out) {
  const int c = blockIdx.y;
  const int gx = vs_row_chunks(L);
  const float sc = scale[c], sh = shift[c];
  float m = 0.f;
  for (long long it = blockIdx.x; it < rows_c * gx; it += gridDim.x) {
    const long long off = (c + (long long)C * (it / gx)) * L;
    const float* p = x + off;
    float* q = y + off;
    vs_walk_chunk(L, vs_row_phase(p, q), (int)(it % gx),
                  [&](int i, auto w) { return vs_ldv<decltype(w)::value>(p + i); },
                  [&](int i, auto v) {
#pragma unroll
                    for (int e = 0; e < decltype(v)::N; ++e) {
                      v.v[e] = vs_act_fast<ACT>(fmaf(v.v[e], sc, sh));
                      m = fmaxf(m, fabsf(v.v[e]));
                    }
                    vs_stv(q + i, v);
                  });
  }
  vs_absmax_commit(m, amax_out);
}

// cnn8 keeps the [B][T][8][F] layout: channel stride F inside a frame, frame stride 8F
template <int ACT>
__global__ __launch_bounds__(256)
void bn_apply_feat_kernel(const float* x, float* y, const float* __restrict__ scale, const float* __restrict__ shift,
                          int F, long long total) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)((i / F) & 7);
    y[i] = vs_act<ACT>(fmaf(x[i], scale[c], shift[c]));
  }
}

// The same pass for the bf16 configuration's training forward: besides y it writes the bf16 row-form copy of the features
// the LSTM input GEMM (and later dW_ih) reads -- [B*T][Kp], Kp >= 8F zero padded -- so no conversion pass re-reads the
// 370 MB tensor.  One thread per 8 consecutive elements of a row (rows are 32 F bytes: 16-byte aligned): two float4 in,
// two float4 + one 16-byte bf16 vector out; a group may straddle a channel boundary (F is odd).
typedef unsigned edge_u4 __attribute__((ext_vector_type(4)));
template <int ACT>
__global__ __launch_bounds__(256)
void bn_apply_feat_bf16_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned short* __restrict__ yb,
                               const float* __restrict__ scale, const float* __restrict__ shift, int F, int Kp, long long rows) {
  const int K = 8 * F, groups = Kp >> 3;
  const long long total = rows * groups;
  float sc[8], sh[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { sc[c] = scale[c]; sh[c] = shift[c]; }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / groups;
    const int k0 = (int)(i - r * groups) * 8;
    float v[8];
    if (k0 < K) {                       // K is a multiple of 8: a group is entirely inside or entirely in the padding
      const float4 a = *reinterpret_cast<const float4*>(x + r * K + k0), b = *reinterpret_cast<const float4*>(x + r * K + k0 + 4);
      const float in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      const int c0 = k0 / F, edge = (c0 + 1) * F;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + (k0 + e >= edge ? 1 : 0);
        float s = sc[0], h = sh[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) { s = c == q ? sc[q] : s; h = c == q ? sh[q] : h; }
        v[e] = vs_act<ACT>(fmaf(in[e], s, h));
      }
      *reinterpret_cast<float4*>(y + r * K + k0) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(y + r * K + k0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    const edge_u4 o = {vs_pack_bf16(v[0], v[1]), vs_pack_bf16(v[2], v[3]), vs_pack_bf16(v[4], v[5]), vs_pack_bf16(v[6], v[7])};
    *reinterpret_cast<edge_u4*>(yb + r * Kp + k0) = o;
  }
}

// eval-mode BatchNorm applied to a raw conv output (bias already inside z): constants from the
// running statistics, also kept as mean / invstd for the backward pass.
__global__ void bn_eval_consts_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int C,
                                      float* __restrict__ scale, float* __restrict__ shift,
                                      float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(rvar[c] + eps);
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - rmean[c] * sc;
  mean_out[c] = rmean[c];
  invstd_out[c] = is;
}

__global__ __launch_bounds__(256)
void bn_stats_feat_kernel(const float* __restrict__ x, int F, long long rows /* B*T*8 */, double* __restrict__ stats) {
  // one block per group of rows; row r belongs to channel r & 7
  __shared__ double sh[8][2];
  if (threadIdx.x < 16) sh[threadIdx.x >> 1][threadIdx.x & 1] = 0.0;
  __syncthreads();
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* p = x + r * F;
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < F; i += 256) { const float v = p[i]; s += v; q = fmaf(v, v, q); }
    double ds = s, dq = q;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ds += __shfl_down(ds, o, 64); dq += __shfl_down(dq, o, 64); }
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&sh[r & 7][0], ds);
      atomicAdd(&sh[r & 7][1], dq);
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) atomicAdd(&stats[threadIdx.x], sh[threadIdx.x >> 1][threadIdx.x & 1]);
}

}  // namespace

int vs_bn_apply_impl(const float*, float*, int, int, int, int, const float*, const float*, unsigned*, hipStream_t);
int vs_bn_apply_feat_impl(const float*, float*, int, int, int, int, const float*, const float*, hipStream_t);

int vs_bn_fold_impl(const float* gamma, const float* beta, const float* mean, const float* var,
                    const float* conv_bias, float eps, int C, float* scale, float* shift, hipStream_t stream) {
  VS_REQUIRE(C > 0, "bn_fold: C=%d", C);
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, gamma, beta, mean, var, conv_bias, eps, C, scale, shift);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_conv_first_fwd_impl(const float* x, const float* w, const float* scale, const float* shift, float* out,
                           int B, int T, int F, int act, unsigned* amax_out, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "conv_first: bad shape B=%d T=%d F=%d", B, T, F);
  VS_REQUIRE((long long)T * F < 2147483647LL / 64 && B <= 65535, "conv_first: shape too large");
  dim3 grid((T * F + 1023) / 1024, B), block(256);
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL(conv_first_kernel<VS_ACT_RELU>, grid, block, 0, stream, x, w, scale, shift, out, T, F, amax_out); break;
    case VS_ACT_MISH: hipLaunchKernelGGL(conv_first_kernel<VS_ACT_MISH>, grid, block, 0, stream, x, w, scale, shift, out, T, F, amax_out); break;
    case VS_ACT_NONE: hipLaunchKernelGGL(conv_first_kernel<VS_ACT_NONE>, grid, block, 0, stream, x, w, scale, shift, out, T, F, amax_out); break;
    default: VS_REQUIRE(false, "conv_first: unknown activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_conv_last_fwd_impl(const float* in, const float* w, const float* scale, const float* shift, float* out,
                          int B, int T, int F, int act, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "conv_last: bad shape B=%d T=%d F=%d", B, T, F);
  VS_REQUIRE((long long)T * F < 2147483647LL / 64 && B <= 65535, "conv_last: shape too large");
  dim3 grid((T * F + 255) / 256, B), block(256);
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL(conv_last_kernel<VS_ACT_RELU>, grid, block, 0, stream, in, w, scale, shift, out, T, F); break;
    case VS_ACT_MISH: hipLaunchKernelGGL(conv_last_kernel<VS_ACT_MISH>, grid, block, 0, stream, in, w, scale, shift, out, T, F); break;
    case VS_ACT_NONE: hipLaunchKernelGGL(conv_last_kernel<VS_ACT_NONE>, grid, block, 0, stream, in, w, scale, shift, out, T, F); break;
    default: VS_REQUIRE(false, "conv_last: unknown activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

// Train-mode BatchNorm over a raw conv output x held as [B][C][T*F]:
// statistics -> scale/shift (+ running buffers, + mean/invstd when requested) -> y = act(BN(x)).
// y may alias x (in place, inference-only callers) or be a separate buffer (training keeps x).
// stats_slots > 0: the conv that produced x already accumulated {sum, sum of squares} of every channel into
// `stats_slots` partial slots of [C][2] doubles (its epilogue: conv_f16x3*.hip STATS); they are folded into slot 0
// here and the statistics pass over x is skipped.
static __global__ void bn_stats_fold_kernel(double* __restrict__ stats, int n, int slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (int k = 0; k < slots; ++k) v += stats[(size_t)k * n + i];
  stats[i] = v;
}

int vs_bn_train_impl(const float* x, float* y, int B, int C, int plane, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float eps, float momentum, int act,
                     double* stats /* [C][2] */, float* scale, float* shift, float* mean_out, float* invstd_out,
                     unsigned* amax_out, hipStream_t stream, int stats_slots) {
  VS_REQUIRE(B > 0 && C > 0 && plane > 0 && B <= 65535 && C <= 65535, "bn_train: bad shape B=%d C=%d plane=%d", B, C, plane);
  if (stats_slots > 0) {
    hipLaunchKernelGGL(bn_stats_fold_kernel, dim3((2 * C + 127) / 128), dim3(128), 0, stream, stats, 2 * C, stats_slots);
  } else {
    VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * C, stream));
    hipLaunchKernelGGL(bn_stats_kernel, dim3(vs_bn_blocks_per_channel(C, B, plane), C), dim3(256), 0, stream, x, C, (long long)B, plane, stats);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, stats, (double)B * plane, gamma, beta,
                     eps, momentum, C, running_mean, running_var, scale, shift, mean_out, invstd_out);
  return vs_bn_apply_impl(x, y, B, C, plane, act, scale, shift, amax_out, stream);
}

// Train-mode BatchNorm constants from accumulated statistics: stats = [slots][C][2] doubles {sum, sum of squares}
// (slots > 1: partial sums of the conv epilogues, folded here) over `count` values per channel -> scale / shift of the
// apply pass, mean / invstd for the backward pass, running buffers updated like nn.BatchNorm2d.
// rezero_doubles > 0 (the training orchestration, C <= 64): ONE launch folds the slots, finalizes and clears that many doubles of the
// scratch behind itself (vs_fold_slots); slot 0 does not receive the totals then.
static __global__ __launch_bounds__(VS_FOLD_THREADS)
void bn_fold_finalize_kernel(double* __restrict__ stats, int slots, int rezero, double count,
                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, int C,
                             float* __restrict__ running_mean, float* __restrict__ running_var,
                             float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  __shared__ double part[8 * 128], tot[128];
  vs_fold_slots(stats, 2 * C, slots, rezero, part, tot);
  const int c = threadIdx.x;
  if (c >= C) return;
  const double mean = tot[2 * c] / count;               // (the arithmetic of bn_finalize_kernel, to the letter)
  double var = tot[2 * c + 1] / count - mean * mean;
  if (var < 0) var = 0;
  const float is = 1.0f / sqrtf((float)var + eps);
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (mean_out) mean_out[c] = (float)mean;
  if (invstd_out) invstd_out[c] = is;
  const double unb = count > 1 ? var * (count / (count - 1)) : var;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
}

int vs_bn_finalize_impl(double* stats, int slots, double count, int C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float eps, float momentum,
                        float* scale, float* shift, float* mean_out, float* invstd_out, hipStream_t stream, int rezero_doubles) {
  VS_REQUIRE(stats && C > 0 && count > 0 && slots >= 1, "bn_finalize: bad argument (stats %p, slots %d, C %d, count %g)", (void*)stats, slots, C, count);
  VS_REQUIRE(gamma && beta && scale && shift, "bn_finalize: NULL gamma / beta / scale / shift");
  if (rezero_doubles > 0 && C <= 64) {
    hipLaunchKernelGGL(bn_fold_finalize_kernel, dim3(1), dim3(VS_FOLD_THREADS), 0, stream, stats, slots, rezero_doubles, count, gamma, beta,
                       eps, momentum, C, running_mean, running_var, scale, shift, mean_out, invstd_out);
    VS_LAUNCH_CHECK();
    return 0;
  }
  if (slots > 1) hipLaunchKernelGGL(bn_stats_fold_kernel, dim3((2 * C + 127) / 128), dim3(128), 0, stream, stats, 2 * C, slots);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, stats, count, gamma, beta,
                     eps, momentum, C, running_mean, running_var, scale, shift, mean_out, invstd_out);
  VS_LAUNCH_CHECK();
  return 0;
}

// y = act(x*scale[c] + shift[c]) over [B][C][plane]
int vs_bn_apply_impl(const float* x, float* y, int B, int C, int plane, int act, const float* scale, const float* shift,
                     unsigned* amax_out, hipStream_t stream) {
  VS_REQUIRE(B > 0 && C > 0 && plane > 0 && B <= 65535 && C <= 65535, "bn_apply: bad shape B=%d C=%d plane=%d", B, C, plane);
  dim3 grid(vs_bn_blocks_per_channel(C, B, plane), C), block(256);
  {      // [r6, call 33] 512 workgroups per channel instead of 2048 / C (fp32-class step, B = 64: 1.326 -> 1.227 ms per pass; 128: 1.270, all items: 1.239)
    const long long items = (long long)B * vs_row_chunks(plane);
    if (grid.x < 512) grid.x = (unsigned)(items < 512 ? items : 512);
  }
  const long long rc = B;
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL(bn_apply_kernel<VS_ACT_RELU>, grid, block, 0, stream, x, y, scale, shift, C, rc, plane, amax_out); break;
    case VS_ACT_MISH: hipLaunchKernelGGL(bn_apply_kernel<VS_ACT_MISH>, grid, block, 0, stream, x, y, scale, shift, C, rc, plane, amax_out); break;
    case VS_ACT_NONE: hipLaunchKernelGGL(bn_apply_kernel<VS_ACT_NONE>, grid, block, 0, stream, x, y, scale, shift, C, rc, plane, amax_out); break;
    default: VS_REQUIRE(false, "bn_apply: unknown activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_bn_apply_feat_impl(const float* x, float* y, int B, int T, int F, int act, const float* scale, const float* shift,
                          hipStream_t stream) {
  const long long total = (long long)B * T * 8 * F;
  const int ga = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL(bn_apply_feat_kernel<VS_ACT_RELU>, dim3(ga), dim3(256), 0, stream, x, y, scale, shift, F, total); break;
    case VS_ACT_MISH: hipLaunchKernelGGL(bn_apply_feat_kernel<VS_ACT_MISH>, dim3(ga), dim3(256), 0, stream, x, y, scale, shift, F, total); break;
    case VS_ACT_NONE: hipLaunchKernelGGL(bn_apply_feat_kernel<VS_ACT_NONE>, dim3(ga), dim3(256), 0, stream, x, y, scale, shift, F, total); break;
    default: VS_REQUIRE(false, "bn_apply_feat: unknown activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

// y = act(x*scale[c] + shift[c]) over the cnn8 feature layout, plus its bf16 row-form copy yb [B*T][Kp] (zero padded)
int vs_bn_apply_feat_bf16_impl(const float* x, float* y, void* yb, int Kp, int B, int T, int F, int act, const float* scale, const float* shift,
                               hipStream_t stream) {
  VS_REQUIRE(x && y && yb && B > 0 && T > 0 && F > 0 && Kp >= 8 * F && Kp % 8 == 0, "bn_apply_feat_bf16: bad argument");
  VS_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(yb) & 15) == 0,
             "bn_apply_feat_bf16: buffers must be 16-byte aligned");
  const long long rows = (long long)B * T, total = rows * (Kp >> 3);
  // [r6, call 31] one sweep (a workgroup per 256 pieces, no grid-stride loop): 212 -> 177 us at B = 64 (nhwc_edge.hip: stream_blocks_wide)
  const int ga = (int)((total + 255) / 256 < (1LL << 24) ? (total + 255) / 256 : (1LL << 24));
  unsigned short* o = reinterpret_cast<unsigned short*>(yb);
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL(bn_apply_feat_bf16_kernel<VS_ACT_RELU>, dim3(ga), dim3(256), 0, stream, x, y, o, scale, shift, F, Kp, rows); break;
    case VS_ACT_MISH: hipLaunchKernelGGL(bn_apply_feat_bf16_kernel<VS_ACT_MISH>, dim3(ga), dim3(256), 0, stream, x, y, o, scale, shift, F, Kp, rows); break;
    case VS_ACT_NONE: hipLaunchKernelGGL(bn_apply_feat_bf16_kernel<VS_ACT_NONE>, dim3(ga), dim3(256), 0, stream, x, y, o, scale, shift, F, Kp, rows); break;
    default: VS_REQUIRE(false, "bn_apply_feat_bf16: unknown activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_bn_eval_consts_impl(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, int C,
                           float* scale, float* shift, float* mean_out, float* invstd_out, hipStream_t stream) {
  VS_REQUIRE(C > 0, "bn_eval_consts: C=%d", C);
  hipLaunchKernelGGL(bn_eval_consts_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, gamma, beta, rmean, rvar, eps, C,
                     scale, shift, mean_out, invstd_out);
  VS_LAUNCH_CHECK();
  return 0;
}

// Same for cnn8's output, which already sits in the LSTM feature layout [B][T][8][F].
int vs_bn_train_feat_impl(const float* x, float* y, int B, int T, int F, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float eps, float momentum, int act,
                          double* stats, float* scale, float* shift, float* mean_out, float* invstd_out,
                          hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "bn_train_feat: bad shape");
  VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 16, stream));
  const long long rows = (long long)B * T * 8;
  const int gb = (int)(rows < 4096 ? rows : 4096);
  hipLaunchKernelGGL(bn_stats_feat_kernel, dim3(gb), dim3(256), 0, stream, x, F, rows, stats);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, stream, stats, (double)B * T * F, gamma, beta,
                     eps, momentum, 8, running_mean, running_var, scale, shift, mean_out, invstd_out);
  return vs_bn_apply_feat_impl(x, y, B, T, F, act, scale, shift, stream);
}
