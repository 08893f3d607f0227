// Backward of the conv stack (the `.backward()` half of train.py:94-110 through
// models/voicesplit/model.py:15-52): weight gradients of the 64->64 layers on the fp32 matrix
// cores, BatchNorm+activation backward, and the two bandwidth-bound edge layers.
//
//   data gradient   dIn = conv(dZ, flip/transpose(W))  -> the FORWARD kernel (conv_mfma.hip) fed
//                   with weights packed by conv_pack_weights_kernel(..., transpose_flip = 1)
//   weight gradient dW[co][ci][kt][kf] = sum_{b,t,f} dZ[b][co][t][f] * In[b][ci][t+(kt-KT/2)*dil][f+kf-KF/2]
//                   -> conv64_wgrad_kernel below: an implicit GEMM with M = co, N = ci, K = pixels
//   BatchNorm+act   dZ = scale*(dY - mean(dY) - xhat*mean(dY*xhat)),  dY = dA * act'(Y)
//                   (train: batch statistics; eval: dZ = scale*dY) -> two streaming passes
//
// conv64_wgrad_kernel.  Workgroup = 4 waves, fixed time tap kt, all KF frequency taps; it walks a
// contiguous range of (utterance, frame, 64-bin segment) tiles and keeps its 64x64xKF partial
// sums in accumulator registers the whole time (wave w owns the 32x32 block (co block w>>1,
// ci block w&1) for every kf: 16*KF registers).  Per tile the dZ row segment [64 co][64 f] and
// the input row segment [64 ci][64+KF-1 f] of frame t+(kt-KT/2)*dil are staged global -> VGPR ->
// LDS (raw buffer loads, out-of-range offsets return 0 = ZeroPad2d; the loads of the next tile
// are in flight during the MFMAs).  K runs over pixels: lane (row = lane&31, half = lane>>5)
// reads the four pixels 8*kg+4*half+0..3 of its channel row with ONE ds_read_b128 (row pitch 68
// floats: conflict-free for the b128 lane groups), and the KF shifted input windows are just
// different registers of two such reads -- no data movement per tap.  Frames whose shifted input
// row lies outside the image contribute nothing and are skipped (10 % of the rows at dil = 16).
// The partial sums of the G workgroups of each kt go to a workspace and are summed in a fixed
// order by conv64_wgrad_reduce_kernel (deterministic, no atomics).
#include <type_traits>

#include "vs_internal.h"

namespace {

constexpr unsigned kOob = 0x7FFFFFF0u;
constexpr int kNF = 64;    // output pixels (frequency bins) per tile
constexpr int kPD = 68;    // LDS row pitch in floats: >= 64+4, == 4 (mod 64)

struct WgradArgs {
  const float* dz;   // [B][64][T][F]
  const float* in;   // [B][64][T][F]
  float* part;       // [G][KT][KF][64 co][64 ci]
  int B, T, F, dil, KT, nseg, G;
};

template <int KF>
__global__ __launch_bounds__(256, KF == 1 ? 4 : 3)
void conv64_wgrad_kernel(WgradArgs g) {
  constexpr int PADF = KF / 2;
  __shared__ __attribute__((aligned(16))) float sD[64 * kPD];
  __shared__ __attribute__((aligned(16))) float sA[64 * kPD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // block -> (kt, group): the KT workgroups of one group walk the same tiles at about the same
  // time; bid % 8 is the XCD (observed placement, speed only), so keep them on one L2.
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int kt = (bid >> 3) % g.KT;
  const int grp = ((bid >> 3) / g.KT) * 8 + xcd;

  const int off_t = (kt - g.KT / 2) * g.dil;
  const int t_lo = off_t < 0 ? -off_t : 0;
  const int t_hi = off_t > 0 ? g.T - off_t : g.T;
  const int nvt = t_hi > t_lo ? t_hi - t_lo : 0;
  const int ntiles = g.B * nvt * g.nseg;          // < 2^31, checked by the launcher
  const int per = (ntiles + g.G - 1) / g.G;
  int tile = grp * per;
  const int tile_end = tile + per < ntiles ? tile + per : ntiles;

  const size_t plane = (size_t)g.T * g.F;
  const unsigned plane_bytes = (unsigned)(plane * sizeof(float));

  f32x16 acc[KF];
#pragma unroll
  for (int k = 0; k < KF; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  float sd[16], sa[16], sx = 0.f;
  int nv_next = 0;
  auto issue = [&](int tl) {
    const int bt = tl / g.nseg;
    const int seg = tl - bt * g.nseg;
    const int b = bt / nvt;
    const int t = t_lo + (bt - b * nvt);
    const int f0 = seg * kNF;
    nv_next = g.F - f0 < kNF ? g.F - f0 : kNF;
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.dz + (size_t)b * 64 * plane), 0, 64u * plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.in + (size_t)b * 64 * plane), 0, 64u * plane_bytes, 0x00020000);
    const int fd = f0 + lane;
    const unsigned vd = fd < g.F ? (unsigned)((t * g.F + fd) * 4) : kOob;
    const int fa = f0 - PADF + lane;
    const unsigned va = (fa >= 0 && fa < g.F) ? (unsigned)(((t + off_t) * g.F + fa) * 4) : kOob;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const unsigned so = (unsigned)(wave + 4 * i) * plane_bytes;     // channel plane, wave-uniform
      sd[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, vd, so, 0));
      sa[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, va, so, 0));
    }
    if (KF > 1) {   // the KF-1 halo columns 64..64+KF-2: one (channel, column) per thread
      const int ch = tid >> 2, col = kNF + (tid & 3);
      const int fx = f0 - PADF + col;
      const unsigned vx = (fx >= 0 && fx < g.F) ? (unsigned)(ch * plane_bytes + ((t + off_t) * g.F + fx) * 4) : kOob;
      sx = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, vx, 0, 0));
    }
  };

  const int cb = wave >> 1, nb = wave & 1;
  const float* pd = sD + (cb * 32 + l31) * kPD + 4 * half;
  const float* pa = sA + (nb * 32 + l31) * kPD + 4 * half;

  if (tile < tile_end) issue(tile);
  while (tile < tile_end) {
    __syncthreads();          // every wave is done reading the previous tile
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sD[(wave + 4 * i) * kPD + lane] = sd[i];
      sA[(wave + 4 * i) * kPD + lane] = sa[i];
    }
    if (KF > 1) sA[(tid >> 2) * kPD + kNF + (tid & 3)] = sx;
    const int nkg = (nv_next + 7) >> 3;
    __syncthreads();
    ++tile;
    if (tile < tile_end) issue(tile);
#pragma unroll
    for (int kg = 0; kg < kNF / 8; ++kg) {
      if (kg < nkg) {         // block-uniform
        const float4 d4 = *reinterpret_cast<const float4*>(pd + 8 * kg);
        float w[8];
        {
          const float4 w0 = *reinterpret_cast<const float4*>(pa + 8 * kg);
          w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w;
        }
        if (KF > 1) {
          const float4 w1 = *reinterpret_cast<const float4*>(pa + 8 * kg + 4);
          w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dv = j == 0 ? d4.x : j == 1 ? d4.y : j == 2 ? d4.z : d4.w;
#pragma unroll
          for (int kf = 0; kf < KF; ++kf)
            acc[kf] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv, w[j + kf], acc[kf], 0, 0, 0);
        }
      }
    }
  }

  // partial sums: D row = co (r&3)+8*(r>>2)+4*half, D col = ci = lane&31
  float* out = g.part + ((size_t)grp * g.KT + kt) * KF * 4096;
#pragma unroll
  for (int kf = 0; kf < KF; ++kf)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[(size_t)kf * 4096 + co * 64 + nb * 32 + l31] = acc[kf][r];
    }
}

// dW[co][ci][kt][kf] = sum_g part[g][kt][kf][co][ci]
__global__ void conv64_wgrad_reduce_kernel(const float* __restrict__ part, int G, int NT, float* __restrict__ dw) {
  const int total = NT * 4096;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  float s = 0.f;
  for (int gq = 0; gq < G; ++gq) s += part[(size_t)gq * total + idx];
  const int tap = idx >> 12, co = (idx >> 6) & 63, ci = idx & 63;
  dw[((size_t)co * 64 + ci) * NT + tap] = s;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm + activation backward.  The tensor is seen as rows [R][L] with channel = r % C:
// NCHW activations R = B*C, L = T*F; the LSTM-feature layout of cnn8 R = B*T*8, L = F, C = 8.
// ---------------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float act_grad(float y) {
  if (ACT == VS_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (ACT == VS_ACT_MISH) return vs_mish_grad_fast(y);
  return 1.f;
}

// Both passes walk (row, chunk) items of one channel per block like the forward BatchNorm kernels
// (vs_walk_chunk, conv_edge.hip): grid (blocks per channel, C), float4 middles, two packs in flight.
template <int W> struct BnBwdPack { static constexpr int N = W; VsPack<W> g, z; };

// pass 1: stats[c] += { sum dY, sum dY*xhat },  dY = dA * act'(z*scale+shift), xhat = (z-mean)*invstd
template <int ACT>
__global__ __launch_bounds__(256)
void bn_act_bwd_stats_kernel(const float* __restrict__ da, const float* __restrict__ z, int C, long long rows_c, int L,
                             const float* __restrict__ scale, const float* __restrict__ shift,
                             const float* __restrict__ mean, const float* __restrict__ invstd,
                             double* __restrict__ stats, unsigned* turn = nullptr) {
  const int c = blockIdx.y;
  const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
  const int gx = vs_row_chunks(L);
  double d1 = 0.0, d2 = 0.0;
  for (long long it = blockIdx.x; it < rows_c * gx; it += gridDim.x) {
    const long long off = (c + (long long)C * (it / gx)) * L;
    const float* pz = z + off;
    const float* pg = da + off;
    float s1 = 0.f, s2 = 0.f;
    vs_walk_chunk(L, vs_row_phase(pz, pg), (int)(it % gx),
                  [&](int i, auto w) {
                    BnBwdPack<decltype(w)::value> r;
                    r.g = vs_ldv<decltype(w)::value>(pg + i);
                    r.z = vs_ldv<decltype(w)::value>(pz + i);
                    return r;
                  },
                  [&](int, auto v) {
#pragma unroll
                    for (int e = 0; e < decltype(v)::N; ++e) {
                      const float zv = v.z.v[e];
                      const float dy = v.g.v[e] * act_grad<ACT>(fmaf(zv, sc, sh));
                      s1 += dy;
                      s2 = fmaf(dy, (zv - mu) * is, s2);
                    }
                  });
    d1 += s1;
    d2 += s2;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d1 += __shfl_down(d1, o, 64);
    d2 += __shfl_down(d2, o, 64);
  }
  __shared__ double sh2[8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh2[2 * w] = d1; sh2[2 * w + 1] = d2; }
  __syncthreads();
  unsigned* my_turn = turn ? turn + VS_TURN_CHANNEL + c : nullptr;      // deterministic mode: the workgroups of a channel add in index order (a word per channel)
  vs_turn_begin(my_turn, blockIdx.x);
  if (threadIdx.x == 0) {
    atomicAdd(&stats[2 * c], sh2[0] + sh2[2] + sh2[4] + sh2[6]);
    atomicAdd(&stats[2 * c + 1], sh2[1] + sh2[3] + sh2[5] + sh2[7]);
  }
  vs_turn_end(my_turn, blockIdx.x, gridDim.x);
}

// per channel: parameter gradients and the coefficients of pass 2,  dZ = cA*dY + cB*z + cC
//   train: dZ = scale*(dY - s1/N - xhat*s2/N);   eval: dZ = scale*dY
//   dgamma = s2, dbeta = s1, dbias(conv) = sum dZ = 0 (train) / scale*s1 (eval)
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ stats, double count, int train, int C,
                                       const float* __restrict__ scale, const float* __restrict__ mean,
                                       const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias,
                                       float* __restrict__ coef /* [3][C] */) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double s1 = stats[2 * c], s2 = stats[2 * c + 1];
  if (dgamma) dgamma[c] = (float)s2;
  if (dbeta) dbeta[c] = (float)s1;
  const double sc = scale[c];
  if (train) {
    const double k1 = s1 / count, k2 = s2 / count, is = invstd[c], mu = mean[c];
    coef[c] = (float)sc;
    coef[C + c] = (float)(-sc * k2 * is);
    coef[2 * C + c] = (float)(-sc * k1 + sc * k2 * is * mu);
    if (dbias) dbias[c] = 0.f;
  } else {
    coef[c] = (float)sc;
    coef[C + c] = 0.f;
    coef[2 * C + c] = 0.f;
    if (dbias) dbias[c] = (float)(sc * s1);
  }
}

// pass 2: dz = cA*(dA*act'(z*scale+shift)) + cB*z + cC     (dz may alias da)
template <int ACT>
__global__ __launch_bounds__(256)
void bn_act_bwd_apply_kernel(const float* da, const float* __restrict__ z, int C, long long rows_c, int L,
                             const float* __restrict__ scale, const float* __restrict__ shift,
                             const float* __restrict__ coef, float* dz, unsigned* amax_out) {
  const int c = blockIdx.y;
  const float sc = scale[c], sh = shift[c], cA = coef[c], cB = coef[C + c], cC = coef[2 * C + c];
  const int gx = vs_row_chunks(L);
  float m = 0.f;
  for (long long it = blockIdx.x; it < rows_c * gx; it += gridDim.x) {
    const long long off = (c + (long long)C * (it / gx)) * L;
    const float* pz = z + off;
    const float* pg = da + off;
    float* po = dz + off;
    vs_walk_chunk(L, vs_row_phase(pz, pg, po), (int)(it % gx),
                  [&](int i, auto w) {
                    BnBwdPack<decltype(w)::value> r;
                    r.g = vs_ldv<decltype(w)::value>(pg + i);
                    r.z = vs_ldv<decltype(w)::value>(pz + i);
                    return r;
                  },
                  [&](int i, auto v) {
#pragma unroll
                    for (int e = 0; e < decltype(v)::N; ++e) {
                      const float zv = v.z.v[e];
                      const float dy = v.g.v[e] * act_grad<ACT>(fmaf(zv, sc, sh));
                      const float o = fmaf(cA, dy, fmaf(cB, zv, cC));
                      v.g.v[e] = o;
                      m = fmaxf(m, fabsf(o));
                    }
                    vs_stv(po + i, v.g);
                  });
  }
  vs_absmax_commit(m, amax_out);
}

// ---------------------------------------------------------------------------------------------
// cnn8 (1x1, 64->8, output in the LSTM feature layout [B][T][8][F])
// ---------------------------------------------------------------------------------------------
// dIn[b][ci][t][f] = sum_o W[o][ci] * dZ[b][t][o][f].  A thread owns 4 consecutive pixels of the
// plane and writes them as one 16-byte store per channel (1 KB per wave instruction instead of
// 256 B; planes are only 4-byte aligned -- T*F is odd -- and the stores are issued unaligned).
__global__ __launch_bounds__(256)
void conv_last_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ din, int T, int F) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int plane = T * F;
  const int pix0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int b = blockIdx.y;
  if (pix0 >= plane) return;
  const int np = plane - pix0 < 4 ? plane - pix0 : 4;
  float v[8][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int pix = pix0 + (e < np ? e : 0);
    const int t = pix / F, f = pix - t * F;
    const float* src = dz + ((size_t)b * T + t) * 8 * F + f;
#pragma unroll
    for (int o = 0; o < 8; ++o) v[o][e] = src[(size_t)o * F];
  }
  float* dst = din + (size_t)b * 64 * plane + pix0;
#pragma unroll 2
  for (int c = 0; c < 64; ++c) {
    f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const float wv = w[o * 64 + c];
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = fmaf(wv, v[o][e], a[e]);
    }
    float* q = dst + (size_t)c * plane;
    if (np == 4) {
      __builtin_memcpy(q, &a, 16);          // one global_store_dwordx4, any 4-byte alignment
    } else {
      for (int e = 0; e < np; ++e) q[e] = a[e];
    }
  }
}

// part[blk][o][ci] = sum over this block's (b, t, 256-bin segment) tiles of dZ[o][pix]*In[ci][pix].
// HBM-bound (2.96 GB of input per launch): a tile is 64 channel rows x 256 bins, each row segment one
// 16-byte-per-lane load (1 KB per wave instruction; rows are only 4-byte aligned: issued unaligned),
// the next tile travels to registers while this one is multiplied, and the multiply reads LDS as
// b128 (4 bins per read, row pitch 260 floats: lane = channel hits 16 different bank quads) --
// 3 LDS instructions per 8 FMAs.  (First version: 64-bin tiles, dword loads, 3 scalar LDS reads per
// 2 FMAs: 1.47 ms.)
constexpr int kLwSeg = 256;           // bins per tile
constexpr int kLwPitch = kLwSeg + 4;  // floats
__global__ __launch_bounds__(256)
void conv_last_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ in, float* __restrict__ part,
                            int B, int T, int F, int nseg) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float lw_smem[];
  float* const sA = lw_smem;                       // [64][kLwPitch]
  float* const sD = lw_smem + 64 * kLwPitch;       // [8][kLwPitch]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ci = lane, og = w;
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const long long ntiles = (long long)B * T * nseg;
  const size_t plane = (size_t)T * F;

  f4 ra[16], rd[2];
  // 4 bins of one row starting at p (row has `left` bins from p on): full groups as one 16-byte load
  auto load4 = [&](const float* p, int left) -> f4 {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (left >= 4) {
      __builtin_memcpy(&v, p, 16);
    } else {
      for (int e = 0; e < left; ++e) v[e] = p[e];
    }
    return v;
  };
  auto issue = [&](long long tile) {
    const int seg = (int)(tile % nseg);
    const long long bt = tile / nseg;
    const long long b = bt / T, t = bt % T;
    const int f = seg * kLwSeg + 4 * lane;
    const int left = F - f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ch = w + 4 * i;
      ra[i] = load4(in + ((size_t)(b * 64 + ch)) * plane + (size_t)t * F + f, left);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = w + 4 * i;
      rd[i] = load4(dz + ((size_t)bt * 8 + o) * F + f, left);
    }
  };

  long long tile = blockIdx.x;
  if (tile < ntiles) issue(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const int seg = (int)(tile % nseg);
    const int nv = F - seg * kLwSeg < kLwSeg ? F - seg * kLwSeg : kLwSeg;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<f4*>(&sA[(w + 4 * i) * kLwPitch + 4 * lane]) = ra[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f4*>(&sD[(w + 4 * i) * kLwPitch + 4 * lane]) = rd[i];
    __syncthreads();
    if (tile + gridDim.x < ntiles) issue(tile + gridDim.x);
    const int n4 = (nv + 3) >> 2;                  // bins beyond nv were staged as zeros
    const float* pa = sA + ci * kLwPitch;
    const float* p0 = sD + (2 * og) * kLwPitch;
    const float* p1 = p0 + kLwPitch;
#pragma unroll 4
    for (int q = 0; q < n4; ++q) {
      const f4 a = *reinterpret_cast<const f4*>(pa + 4 * q);
      const f4 d0 = *reinterpret_cast<const f4*>(p0 + 4 * q);
      const f4 d1 = *reinterpret_cast<const f4*>(p1 + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0[e] = fmaf(a[e], d0[e], acc0[e]);
        acc1[e] = fmaf(a[e], d1[e], acc1[e]);
      }
    }
  }
  part[(size_t)blockIdx.x * 512 + (2 * og) * 64 + ci] = (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]);
  part[(size_t)blockIdx.x * 512 + (2 * og + 1) * 64 + ci] = (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
}

// out[i] = sum_g part[g][i].  A thread owns four consecutive outputs (16-byte loads) and keeps eight slabs in flight:
// the 128 slabs x 1.6 MB of a 5x5 weight gradient are a 210 MB stream, not a latency chain.
__global__ __launch_bounds__(256)
void reduce_partials_kernel(const float* __restrict__ part, int G, int n, float* __restrict__ out) {
  const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (idx >= n) return;
  if (idx + 4 <= n && (n & 3) == 0) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int gq = 0;
    for (; gq + 8 <= G; gq += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + (size_t)(gq + u) * n + idx);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; gq < G; ++gq) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)gq * n + idx);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + idx) = s;
    return;
  }
  for (int e = idx; e < n && e < idx + 4; ++e) {
    float s = 0.f;
    for (int gq = 0; gq < G; ++gq) s += part[(size_t)gq * n + e];
    out[e] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// cnn1 (1x7, 1->64): dW[c][k] = sum_{b,t,f} dZ[b][c][t][f] * x[b][t][f+k-3]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void conv_first_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ x, double* __restrict__ acc,
                             int T, int F) {
  const int plane = T * F;
  const int c = blockIdx.y, b = blockIdx.z;
  const float* pz = dz + ((size_t)b * 64 + c) * plane;
  const float* px = x + (size_t)b * plane;
  float s[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) s[k] = 0.f;
  const int e0 = blockIdx.x * 4096;
  const int e1 = e0 + 4096 < plane ? e0 + 4096 : plane;
  for (int e = e0 + threadIdx.x; e < e1; e += 256) {
    const int f = e % F;
    const float v = pz[e];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int ff = f + k - 3;
      const float xv = (ff >= 0 && ff < F) ? px[e + k - 3] : 0.f;
      s[k] = fmaf(v, xv, s[k]);
    }
  }
  __shared__ double red[4][7];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double d = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
    if (lane == 0) red[w][k] = d;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int k = threadIdx.x;
    atomicAdd(&acc[c * 7 + k], red[0][k] + red[1][k] + red[2][k] + red[3][k]);
  }
}

// xp[r][F+6] = {0,0,0, x[r][0..F-1], 0,0,0}: the 1x7 taps of cnn1 read it without edge tests
__global__ void pad_rows3_kernel(const float* __restrict__ x, float* __restrict__ xp, long long rows, int F) {
  const int P = F + 6;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * P; i += (long long)gridDim.x * 256) {
    const long long r = i / P;
    const int j = (int)(i - r * P);
    xp[i] = (j >= 3 && j < F + 3) ? x[r * F + (j - 3)] : 0.f;
  }
}

// cnn1: pass 2 of the BatchNorm backward with the 1x7 weight gradient folded in.  dZ1 has no other
// consumer (the input needs no gradient), so it is formed in registers and contracted with the
// seven shifted inputs on the spot, dW[c][k] += dZ1[b][c][t][f] * x[b][t][f+k-3], instead of being
// written (2.96 GB at B=64) and read back by conv_first_wgrad_kernel.  Same (row, chunk) walk as
// bn_act_bwd_apply_kernel; xp = zero-padded input rows (pad_rows3_kernel).
template <int ACT>
__global__ __launch_bounds__(256)
void bn_act_bwd_first_kernel(const float* __restrict__ da, const float* __restrict__ z, const float* __restrict__ xp,
                             long long B, int T, int F, const float* __restrict__ scale, const float* __restrict__ shift,
                             const float* __restrict__ coef, double* __restrict__ acc /* [64][7] */) {
  constexpr int C = 64;
  const int c = blockIdx.y;
  const int L = T * F, P = F + 6;
  const float sc = scale[c], sh = shift[c], cA = coef[c], cB = coef[C + c], cC = coef[2 * C + c];
  const float invF = 1.0f / (float)F;
  const int gx = vs_row_chunks(L);
  double d[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) d[k] = 0.0;
  for (long long it = blockIdx.x; it < B * gx; it += gridDim.x) {
    const long long b = it / gx;
    const long long off = (c + (long long)C * b) * L;
    const float* pz = z + off;
    const float* pg = da + off;
    const float* xb = xp + b * T * P;
    float s[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) s[k] = 0.f;
    vs_walk_chunk(L, vs_row_phase(pz, pg), (int)(it % gx),
                  [&](int i, auto w) {
                    BnBwdPack<decltype(w)::value> r;
                    r.g = vs_ldv<decltype(w)::value>(pg + i);
                    r.z = vs_ldv<decltype(w)::value>(pz + i);
                    return r;
                  },
                  [&](int i, auto v) {
                    constexpr int W = decltype(v)::N;
                    // (t, f) of element i: float quotient (i < 2^24) with a one-step correction
                    int t = (int)((float)i * invF);
                    int f = i - t * F;
                    if (f < 0) { f += F; --t; } else if (f >= F) { f -= F; ++t; }
                    float dzv[W];
#pragma unroll
                    for (int e = 0; e < W; ++e) {
                      const float zv = v.z.v[e];
                      const float dy = v.g.v[e] * act_grad<ACT>(fmaf(zv, sc, sh));
                      dzv[e] = fmaf(cA, dy, fmaf(cB, zv, cC));
                    }
                    const float* xr = xb + (size_t)t * P + f;      // tap k of element e of this row: xr[e + k]
                    if (f + W <= F) {
                      float xs[W + 6];
#pragma unroll
                      for (int j = 0; j < W + 6; ++j) xs[j] = xr[j];
#pragma unroll
                      for (int e = 0; e < W; ++e)
#pragma unroll
                        for (int k = 0; k < 7; ++k) s[k] = fmaf(dzv[e], xs[e + k], s[k]);
                    } else {                                        // the pack runs into the next frame
#pragma unroll
                      for (int e = 0; e < W; ++e) {
                        const float* q = (f + e < F) ? xr + e : xr + e + 6;
#pragma unroll
                        for (int k = 0; k < 7; ++k) s[k] = fmaf(dzv[e], q[k], s[k]);
                      }
                    }
                  });
#pragma unroll
    for (int k = 0; k < 7; ++k) d[k] += s[k];
  }
  __shared__ double red[4][7];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double x = d[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    if (lane == 0) red[w][k] = x;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int k = threadIdx.x;
    atomicAdd(&acc[c * 7 + k], red[0][k] + red[1][k] + red[2][k] + red[3][k]);
  }
}

__global__ void cvt_f64_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------

// Persistent grid = one resident round: KT * groups <= 256 CUs x workgroups per CU (3 for the
// 5x5 kernel at <= 168 VGPRs, 4 for the 7x1 one), groups a multiple of 8 (XCD mapping).  A grid
// even slightly above the resident capacity would run a second, nearly empty round.
extern "C" int vs_conv64_wgrad_groups(int KT) { return KT == 7 ? 144 : 152; }

extern "C" size_t vs_conv64_wgrad_partial_floats(int KT, int KF) {
  // fp32 kernel: vs_conv64_wgrad_groups slabs; split-f16 ring kernel: 128 (5x5) / 256 (7x1) slabs
  const size_t g32 = (size_t)vs_conv64_wgrad_groups(KT), g16 = KF == 5 ? 128 : 256;
  return (g32 > g16 ? g32 : g16) * KT * KF * 4096;
}

int vs_conv64_wgrad_impl(const float* dz, const float* in, float* part, float* dw,
                         int B, int T, int F, int KT, int KF, int dil, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "conv64_wgrad: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((KT == 7 && KF == 1) || (KT == 5 && KF == 5), "conv64_wgrad: unsupported kernel %dx%d", KT, KF);
  VS_REQUIRE((long long)64 * T * F * 4 < (long long)kOob, "conv64_wgrad: T*F=%lld too large for 32-bit offsets", (long long)T * F);
  VS_REQUIRE((long long)B * T * ((F + kNF - 1) / kNF) < 2147483647LL, "conv64_wgrad: too many tiles");
  const int G = vs_conv64_wgrad_groups(KT);
  WgradArgs a{dz, in, part, B, T, F, dil, KT, (F + kNF - 1) / kNF, G};
  dim3 grid(G * KT), block(256);
  if (KF == 5) hipLaunchKernelGGL(conv64_wgrad_kernel<5>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(conv64_wgrad_kernel<1>, grid, block, 0, stream, a);
  const int total = KT * KF * 4096;
  hipLaunchKernelGGL(conv64_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, part, G, KT * KF, dw);
  VS_LAUNCH_CHECK();
  return 0;
}

// BatchNorm+activation backward over rows [R][L], channel = r % C (see above).
//   stats [C][2] double scratch, coef [3][C] float scratch; dz may alias da.
int vs_bn_act_bwd_impl(const float* da, const float* z, float* dz, int C, long long R, int L, int act, int train,
                       const float* scale, const float* shift, const float* mean, const float* invstd,
                       float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, unsigned* amax_out,
                       hipStream_t stream) {
  VS_REQUIRE(C > 0 && R > 0 && L > 0 && R % C == 0, "bn_act_bwd: bad shape C=%d R=%lld L=%d", C, R, L);
  VS_REQUIRE(R <= 2147483647LL && C <= 65535, "bn_act_bwd: too many rows");
  VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * C, stream));
  const long long rows_per_c = R / C;
  unsigned* turn = C <= 64 ? g_vs_turn : nullptr;          // (the turn region holds a word for each of 64 channels: the features' BatchNorm, C = 8, is the user)
  const int bpc_full = vs_bn_blocks_per_channel(C, rows_per_c, L);
  // deterministic mode: chains of at most 64 turns per channel behind the pass (~0.2 ms), 64 * C workgroups all resident at once; the
  // apply pass below keeps the full grid (round 6: sixteen per channel on BOTH passes cost 4.2 ms at B = 64 -- 128 workgroups cannot
  // stream 1.9 GB)
  int bpc = (turn && bpc_full > 64) ? 64 : bpc_full;
  int bpc_apply = bpc_full;
  if (C <= 8) {      // [r6, calls 34-35] the features' BatchNorm (8 channels of 601-element rows): more, shorter-lived workgroups -- statistics pass
    // 256 -> 512 per channel (373 -> 330 us; 4096: 736, its flush), apply pass 256 -> 4096 (304 -> 216 us; all rows: 250); step -0.09 ms
    const long long items = rows_per_c * vs_row_chunks(L);
    if (!turn && bpc < 512) bpc = (int)(items < 512 ? items : 512);
    if (bpc_apply < 4096) bpc_apply = (int)(items < 4096 ? items : 4096);
  }
  dim3 grid(bpc, C), block(256);
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL(bn_act_bwd_stats_kernel<VS_ACT_RELU>, grid, block, 0, stream, da, z, C, rows_per_c, L, scale, shift, mean, invstd, stats, turn); break;
    case VS_ACT_MISH: hipLaunchKernelGGL(bn_act_bwd_stats_kernel<VS_ACT_MISH>, grid, block, 0, stream, da, z, C, rows_per_c, L, scale, shift, mean, invstd, stats, turn); break;
    case VS_ACT_NONE: hipLaunchKernelGGL(bn_act_bwd_stats_kernel<VS_ACT_NONE>, grid, block, 0, stream, da, z, C, rows_per_c, L, scale, shift, mean, invstd, stats, turn); break;
    default: VS_REQUIRE(false, "bn_act_bwd: unknown activation %d", act);
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, stats, (double)rows_per_c * L, train, C,
                     scale, mean, invstd, dgamma, dbeta, dbias, coef);
  const dim3 grid_apply(bpc_apply, C);
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL(bn_act_bwd_apply_kernel<VS_ACT_RELU>, grid_apply, block, 0, stream, da, z, C, rows_per_c, L, scale, shift, coef, dz, amax_out); break;
    case VS_ACT_MISH: hipLaunchKernelGGL(bn_act_bwd_apply_kernel<VS_ACT_MISH>, grid_apply, block, 0, stream, da, z, C, rows_per_c, L, scale, shift, coef, dz, amax_out); break;
    default: hipLaunchKernelGGL(bn_act_bwd_apply_kernel<VS_ACT_NONE>, grid_apply, block, 0, stream, da, z, C, rows_per_c, L, scale, shift, coef, dz, amax_out); break;
  }
  VS_LAUNCH_CHECK();
  return 0;
}

namespace {
__global__ void bn_bwd_fold_slots_kernel(double* __restrict__ stats, int n, int slots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (int k = 0; k < slots; ++k) v += stats[(size_t)k * n + i];
  stats[i] = v;
}
}  // namespace

// stats: [slots][C][2] doubles {sum dy, sum dy*xhat} (slots > 1: folded into slot 0 first) -> parameter gradients and
// the three per-channel coefficients of the apply pass
namespace {
// one launch: fold the slots, the coefficients, clear the scratch (vs_fold_slots; the arithmetic of bn_bwd_finalize_kernel, to the letter)
__global__ __launch_bounds__(VS_FOLD_THREADS)
void bn_bwd_fold_finalize_kernel(double* __restrict__ stats, int slots, int rezero, double count, int train, int C,
                                 const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ invstd,
                                 float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias, float* __restrict__ coef) {
  __shared__ double part[8 * 128], tot[128];
  vs_fold_slots(stats, 2 * C, slots, rezero, part, tot);
  const int c = threadIdx.x;
  if (c >= C) return;
  const double s1 = tot[2 * c], s2 = tot[2 * c + 1];
  if (dgamma) dgamma[c] = (float)s2;
  if (dbeta) dbeta[c] = (float)s1;
  const double sc = scale[c];
  if (train) {
    const double k1 = s1 / count, k2 = s2 / count, is = invstd[c], mu = mean[c];
    coef[c] = (float)sc;
    coef[C + c] = (float)(-sc * k2 * is);
    coef[2 * C + c] = (float)(-sc * k1 + sc * k2 * is * mu);
    if (dbias) dbias[c] = 0.f;
  } else {
    coef[c] = (float)sc;
    coef[C + c] = 0.f;
    coef[2 * C + c] = 0.f;
    if (dbias) dbias[c] = (float)(sc * s1);
  }
}
}  // namespace

int vs_bn_bwd_finalize_impl(double* stats, int slots, double count, int train, int C, const float* scale, const float* mean,
                            const float* invstd, float* dgamma, float* dbeta, float* dbias, float* coef, hipStream_t stream, int rezero_doubles) {
  VS_REQUIRE(stats && coef && C > 0 && count > 0, "bn_bwd_finalize: bad argument");
  if (rezero_doubles > 0 && C <= 64) {
    hipLaunchKernelGGL(bn_bwd_fold_finalize_kernel, dim3(1), dim3(VS_FOLD_THREADS), 0, stream, stats, slots, rezero_doubles, count, train, C,
                       scale, mean, invstd, dgamma, dbeta, dbias, coef);
    VS_LAUNCH_CHECK();
    return 0;
  }
  if (slots > 1) hipLaunchKernelGGL(bn_bwd_fold_slots_kernel, dim3((2 * C + 127) / 128), dim3(128), 0, stream, stats, 2 * C, slots);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, stats, count, train, C,
                     scale, mean, invstd, dgamma, dbeta, dbias, coef);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_conv_last_dgrad_impl(const float* dz, const float* w, float* din, int B, int T, int F, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && B <= 65535, "conv_last_dgrad: bad shape B=%d T=%d F=%d", B, T, F);
  hipLaunchKernelGGL(conv_last_dgrad_kernel, dim3((T * F + 1023) / 1024, B), dim3(256), 0, stream, dz, w, din, T, F);
  VS_LAUNCH_CHECK();
  return 0;
}

extern "C" int vs_conv_last_wgrad_blocks(void) { return 512; }   // two resident workgroups per CU: one round

int vs_conv_last_wgrad_impl(const float* dz, const float* in, float* part /* [blocks][512] */, float* dw,
                            int B, int T, int F, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0, "conv_last_wgrad: bad shape B=%d T=%d F=%d", B, T, F);
  const int nblk = vs_conv_last_wgrad_blocks();
  const size_t lds = (size_t)72 * kLwPitch * sizeof(float);        // 74.9 KB: two workgroups per CU
  VS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_last_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(conv_last_wgrad_kernel, dim3(nblk), dim3(256), lds, stream, dz, in, part, B, T, F, (F + kLwSeg - 1) / kLwSeg);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, stream, part, nblk, 512, dw);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_reduce_partials_impl(const float* part, int G, int n, float* out, hipStream_t stream) {
  VS_REQUIRE(G > 0 && n > 0, "reduce_partials: bad shape G=%d n=%d", G, n);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((n + 1023) / 1024), dim3(256), 0, stream, part, G, n, out);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_conv_first_wgrad_impl(const float* dz, const float* x, double* acc /* [64][7] */, float* dw,
                             int B, int T, int F, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && B <= 65535, "conv_first_wgrad: bad shape B=%d T=%d F=%d", B, T, F);
  VS_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 448, stream));
  hipLaunchKernelGGL(conv_first_wgrad_kernel, dim3((T * F + 4095) / 4096, 64, B), dim3(256), 0, stream, dz, x, acc, T, F);
  hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3(2), dim3(256), 0, stream, acc, dw, 448);
  VS_LAUNCH_CHECK();
  return 0;
}

// cnn1: BatchNorm+activation backward and the 1x7 weight gradient in one go (dZ1 is never stored).
//   da, z: [B][64][T][F]; x: [B][T][F]; xpad: B*T*(F+6) floats of scratch; acc: 448 doubles.
int vs_bn_act_bwd_first_impl(const float* da, const float* z, const float* x, float* xpad, int B, int T, int F, int act, int train,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc,
                             hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F >= 4 && (long long)T * F < (1 << 24), "bn_act_bwd_first: bad shape B=%d T=%d F=%d (needs F >= 4, T*F < 2^24)", B, T, F);
  constexpr int C = 64;
  const int L = T * F;
  VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * C, stream));
  VS_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 448, stream));
  dim3 grid(vs_bn_blocks_per_channel(C, B, L), C), block(256);
  const long long rows = (long long)B * T;
  hipLaunchKernelGGL(pad_rows3_kernel, dim3((unsigned)((rows * (F + 6) + 255) / 256 < 4096 ? (rows * (F + 6) + 255) / 256 : 4096)), block, 0, stream,
                     x, xpad, rows, F);
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL(bn_act_bwd_stats_kernel<VS_ACT_RELU>, grid, block, 0, stream, da, z, C, (long long)B, L, scale, shift, mean, invstd, stats); break;
    case VS_ACT_MISH: hipLaunchKernelGGL(bn_act_bwd_stats_kernel<VS_ACT_MISH>, grid, block, 0, stream, da, z, C, (long long)B, L, scale, shift, mean, invstd, stats); break;
    case VS_ACT_NONE: hipLaunchKernelGGL(bn_act_bwd_stats_kernel<VS_ACT_NONE>, grid, block, 0, stream, da, z, C, (long long)B, L, scale, shift, mean, invstd, stats); break;
    default: VS_REQUIRE(false, "bn_act_bwd_first: unknown activation %d", act);
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(64), 0, stream, stats, (double)B * L, train, C,
                     scale, mean, invstd, dgamma, dbeta, dbias, coef);
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL(bn_act_bwd_first_kernel<VS_ACT_RELU>, grid, block, 0, stream, da, z, xpad, (long long)B, T, F, scale, shift, coef, acc); break;
    case VS_ACT_MISH: hipLaunchKernelGGL(bn_act_bwd_first_kernel<VS_ACT_MISH>, grid, block, 0, stream, da, z, xpad, (long long)B, T, F, scale, shift, coef, acc); break;
    default: hipLaunchKernelGGL(bn_act_bwd_first_kernel<VS_ACT_NONE>, grid, block, 0, stream, da, z, xpad, (long long)B, T, F, scale, shift, coef, acc); break;
  }
  hipLaunchKernelGGL(cvt_f64_f32_kernel, dim3(2), dim3(256), 0, stream, acc, dw, 448);
  VS_LAUNCH_CHECK();
  return 0;
}

// =================================================================================================
// Weight gradient in split-f16 arithmetic (VS_MATH_F16X3; see conv_f16x3.hip for the scheme):
// same decomposition as conv64_wgrad_kernel (workgroup = one time tap kt, all KF frequency taps,
// a contiguous range of (utterance, frame, 64-bin segment) tiles, partial sums kept in registers),
// with K = 16 pixels per v_mfma_f32_32x32x16_f16.  Both operands sit in LDS as f16 hi / lo rows
// [channel][pixel] (pitch 72 halves = 9 x 16 B: conflict-free b128 fragment reads with lane = row);
// a lane's A fragment is the 8 pixels 16*kb + 8*half .. +7 of its dz row, and the KF shifted input
// windows are cut out of ONE aligned 12-pixel read (b128 + b64) per part with v_alignbit for the
// odd shifts -- no data movement in LDS per tap.  All kt workgroups of a group walk the SAME tile
// sequence (tiles whose shifted input row is outside the image are skipped), so the dz segment is
// fetched from HBM once and served to the other four time taps by the XCD's L2.
// =================================================================================================
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

constexpr int kPH = 72;          // LDS row pitch in halves (144 B)
constexpr int kPW = kPH / 2;     // ... in dwords (pixel pairs)

// BF (VS_MATH_BF16): one bf16 rounding per element in the hi image, nothing in the lo image; the
// kernels then issue only the hi x hi product on v_mfma_f32_32x32x16_bf16
template <bool BF>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  if (BF) { hi = vs_pack_bf16(x0, x1); lo = 0u; return; }
  const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  const h2 l = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x0 - (float)h[0], x1 - (float)h[1]));
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

template <bool BF>
__device__ __forceinline__ f32x16 wg_mma(h8 a, h8 b, f32x16 c, int, int, int) {
  if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vs_bf16x8, a), __builtin_bit_cast(vs_bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

struct Wgrad16Args {
  const float* dz;   // [B][64][T][F]
  const float* in;   // [B][64][T][F]
  const float* dz_scale;   // {s, 1/s}
  const float* in_scale;   // {s, 1/s}
  float* part;       // [G][KT][KF][64 co][64 ci]   (scaled by s_dz*s_in)
  int B, T, F, dil, KT, nseg, G;
};

template <int KF, bool BF = false>
__global__ __launch_bounds__(256, 2)
void conv64_wgrad_f16x3_kernel(Wgrad16Args g) {
  constexpr int PADF = KF / 2;
  __shared__ __attribute__((aligned(16))) unsigned sDh[64 * kPW], sDl[64 * kPW], sAh[64 * kPW], sAl[64 * kPW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int kt = (bid >> 3) % g.KT;
  const int grp = ((bid >> 3) / g.KT) * 8 + xcd;
  const int off_t = (kt - g.KT / 2) * g.dil;

  const int ntiles = g.B * g.T * g.nseg;          // same sequence for every kt
  const int per = (ntiles + g.G - 1) / g.G;
  int tile = grp * per;
  const int tile_end = tile + per < ntiles ? tile + per : ntiles;

  const size_t plane = (size_t)g.T * g.F;
  const unsigned plane_bytes = (unsigned)(plane * sizeof(float));
  const float s_dz = g.dz_scale[0], s_in = g.in_scale[0];

  f32x16 acc[KF];
#pragma unroll
  for (int k = 0; k < KF; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

  // skip tiles whose input row t+off_t is outside the image
  auto advance = [&](int tl) {
    while (tl < tile_end) {
      const int t = (tl / g.nseg) % g.T;
      if (t + off_t >= 0 && t + off_t < g.T) break;
      ++tl;
    }
    return tl;
  };

  // staging: one 16-byte buffer load per 4 consecutive pixels of a channel row (rows are only
  // 4-byte aligned, which raw buffer loads accept); pixels outside the row are zeroed afterwards
  // (a neighbouring row's data, not an out-of-range address), which only edge tiles need
  constexpr int NQA = (kNF + KF - 1 + 3) / 4;      // float4 groups per input row: 17 (KF=5) / 16 (KF=1)
  constexpr int NIA4 = (64 * NQA + 255) / 256;     // staging iterations for the input tile
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 sd[4], sa[NIA4];
  int nv_next = 0, f0_next = 0;
  auto issue = [&](int tl) {
    const int bt = tl / g.nseg;
    const int seg = tl - bt * g.nseg;
    const int b = bt / g.T;
    const int t = bt - b * g.T;
    const int f0 = seg * kNF;
    f0_next = f0;
    nv_next = g.F - f0 < kNF ? g.F - f0 : kNF;
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.dz + (size_t)b * 64 * plane), 0, 64u * plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.in + (size_t)b * 64 * plane), 0, 64u * plane_bytes, 0x00020000);
    const unsigned base_d = (unsigned)((t * g.F + f0) * 4);
    const int base_a = ((t + off_t) * g.F + f0 - PADF) * 4;           // may be negative at the very first pixels
    // a 16-byte group that straddles the start or the end of the utterance's slab is split into
    // dwords (the hardware zeroes the whole out-of-range access, valid pixels included); only the
    // first row of channel 0 and the last row of channel 63 can get there
    const long long slab = 64ll * plane_bytes;
    auto load16 = [&](__amdgpu_buffer_rsrc_t r, long long off, bool live) -> f4 {
      if (!live || (off >= 0 && off + 16 <= slab))
        return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, live ? (unsigned)off : kOob, 0, 0));
      f4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long long o = off + 4 * e;
        x[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (o >= 0 && o < slab) ? (unsigned)o : kOob, 0, 0));
      }
      return x;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      sd[i] = load16(rd, (long long)(idx >> 4) * plane_bytes + base_d + (idx & 15) * 16, true);
    }
#pragma unroll
    for (int i = 0; i < NIA4; ++i) {
      const int idx = tid + 256 * i;
      const int ch = idx / NQA, q = idx - ch * NQA;
      sa[i] = load16(ra, (long long)ch * plane_bytes + base_a + q * 16, idx < 64 * NQA);
    }
  };

  const int cb = wave >> 1, nb = wave & 1;
  const int rowd = (cb * 32 + l31) * kPW + 4 * half;     // dword index of this lane's dz fragment, K-block 0
  const int rowa = (nb * 32 + l31) * kPW + 4 * half;

  tile = advance(tile);
  if (tile < tile_end) issue(tile);
  while (tile < tile_end) {
    __syncthreads();          // every wave is done reading the previous tile
    {
      const int f0 = f0_next;
      const bool edge = f0 < PADF || f0 + kNF + KF - 1 - PADF > g.F;     // block-uniform: some pixel of the window is off the row
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx & 15;
        f4 x = sd[i];
        if (edge) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = (f0 + 4 * q + e < g.F) ? x[e] : 0.f;
        }
        unsigned h0, l0, h1, l1;
        split_pair<BF>(x[0] * s_dz, x[1] * s_dz, h0, l0);
        split_pair<BF>(x[2] * s_dz, x[3] * s_dz, h1, l1);
        u2v hv, lv;
        hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
        *reinterpret_cast<u2v*>(&sDh[(idx >> 4) * kPW + 2 * q]) = hv;
        *reinterpret_cast<u2v*>(&sDl[(idx >> 4) * kPW + 2 * q]) = lv;
      }
#pragma unroll
      for (int i = 0; i < NIA4; ++i) {
        const int idx = tid + 256 * i;
        if (idx < 64 * NQA) {
          const int ch = idx / NQA, q = idx - ch * NQA;
          f4 x = sa[i];
          if (edge) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int f = f0 - PADF + 4 * q + e;
              x[e] = (f >= 0 && f < g.F) ? x[e] : 0.f;
            }
          }
          unsigned h0, l0, h1, l1;
          split_pair<BF>(x[0] * s_in, x[1] * s_in, h0, l0);
          split_pair<BF>(x[2] * s_in, x[3] * s_in, h1, l1);
          u2v hv, lv;
          hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
          *reinterpret_cast<u2v*>(&sAh[ch * kPW + 2 * q]) = hv;
          *reinterpret_cast<u2v*>(&sAl[ch * kPW + 2 * q]) = lv;
        }
      }
    }
    const int nkb = (nv_next + 15) >> 4;
    __syncthreads();
    tile = advance(tile + 1);
    if (tile < tile_end) issue(tile);
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
      {
        const u4 dh = *reinterpret_cast<const u4*>(&sDh[rowd + 8 * kb]);
        const u4 dl = *reinterpret_cast<const u4*>(&sDl[rowd + 8 * kb]);
        unsigned wh[6], wl[6];
        {
          const u4 a = *reinterpret_cast<const u4*>(&sAh[rowa + 8 * kb]);
          const u4 c = *reinterpret_cast<const u4*>(&sAl[rowa + 8 * kb]);
          wh[0] = a[0]; wh[1] = a[1]; wh[2] = a[2]; wh[3] = a[3];
          wl[0] = c[0]; wl[1] = c[1]; wl[2] = c[2]; wl[3] = c[3];
          if (KF > 1) {
            const u2v a2 = *reinterpret_cast<const u2v*>(&sAh[rowa + 8 * kb + 4]);
            const u2v c2 = *reinterpret_cast<const u2v*>(&sAl[rowa + 8 * kb + 4]);
            wh[4] = a2[0]; wh[5] = a2[1];
            wl[4] = c2[0]; wl[5] = c2[1];
          }
        }
        // window of tap kf = halves kf .. kf+7 of the 12 read (cut out right before use: few live registers)
        auto tap = [&](const unsigned (&w)[6], int kf) {
          const int m = kf >> 1;
          u4 r;
#pragma unroll
          for (int q = 0; q < 4; ++q) r[q] = (kf & 1) ? __builtin_amdgcn_alignbit(w[m + q + 1], w[m + q], 16) : w[m + q];
          return __builtin_bit_cast(h8, r);
        };
        const h8 dhv = __builtin_bit_cast(h8, dh), dlv = __builtin_bit_cast(h8, dl);
#pragma unroll
        for (int term = BF ? 2 : 0; term < 3; ++term) {
#pragma unroll
          for (int kf = 0; kf < KF; ++kf)
            acc[kf] = wg_mma<BF>(term == 0 ? dlv : dhv, term == 1 ? tap(wl, kf) : tap(wh, kf),
                                                             acc[kf], 0, 0, 0);
        }
      }
    }
  }

  float* out = g.part + ((size_t)grp * g.KT + kt) * KF * 4096;
#pragma unroll
  for (int kf = 0; kf < KF; ++kf)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[(size_t)kf * 4096 + co * 64 + nb * 32 + l31] = acc[kf][r];
    }
}

// dW[co][ci][kt][kf] = (1/(s_dz*s_in)) * sum_g part[g][kt][kf][co][ci]
__global__ void conv64_wgrad_reduce_scaled_kernel(const float* __restrict__ part, int G, int NT, float* __restrict__ dw,
                                                  const float* __restrict__ dz_scale, const float* __restrict__ in_scale) {
  const int total = NT * 4096;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  float s = 0.f;
  for (int gq = 0; gq < G; ++gq) s += part[(size_t)gq * total + idx];
  const int tap = idx >> 12, co = (idx >> 6) & 63, ci = idx & 63;
  dw[((size_t)co * 64 + ci) * NT + tap] = s * (dz_scale[1] * in_scale[1]);
}

}  // namespace

// =================================================================================================
// Weight gradient, ring form (split-f16 arithmetic, same fragments as conv64_wgrad_f16x3_kernel).
//
// The kt-split kernel above stages one dz row segment and ONE shifted input row segment per tile
// and multiplies them for the KF taps of its kt: every input row is staged KT times (rocprofv3:
// 20 GB fetched per launch against 5.9 GB of operands), and 2 staged rows feed only KF taps.
// Here a workgroup of 8 waves owns ALL KT*KF taps and walks a column = (utterance, 64-bin segment,
// residue class r of the frame index modulo the dilation): frames t = r, r+d, r+2d, ...  Consecutive
// steps of a column share KT-1 of their KT input rows, which stay in an LDS ring (row m of the
// column lives in slot m % KT): one step stages ONE new input row and one dz row and issues all
// KT*KF taps from LDS -- every operand element is read from HBM once and converted once.
//   5x5: the 64x64x25 accumulators (400 KB) exceed what one workgroup can hold beside its
//        fragments, so a group = two workgroups with 32 input channels each (same XCD: the second
//        read of the dz row is an L2 hit).  Wave = (co block, tap group): group g owns time tap
//        kt = g (5 taps) and a share of kt = 4 ({0,1},{2},{3},{4}) -> 7/6/6/6 accumulators.
//   7x1: one workgroup holds all 7 taps x 64x64.  Wave = (co block, ci block, tap half {0..3},{4..6}).
// Rows outside the image are never staged; their taps are skipped (wave-uniform test).  The next
// step's rows travel global -> registers while the current step is multiplied; a column's first
// step brings KT/2+1 input rows at once, issued under the previous column's last step.
// Partial sums stay in registers for the whole launch and are reduced in a fixed order afterwards.
// =================================================================================================
namespace {

struct WgradRingArgs {
  const float* dz;   // [B][64][T][F]
  const float* in;   // [B][64][T][F]
  const float* dz_scale;   // {s, 1/s}
  const float* in_scale;   // {s, 1/s}
  float* part;       // [G][KT*KF][64 co][64 ci]   (scaled by s_dz*s_in)
  int B, T, F, dil, nseg, G, nchunk;
};

// CK: columns are cut into chunks of steps (g.nchunk > 1); without it the chunk bookkeeping and the
// row pre-load events compile away (they cost the 5x5 instance, which sits at 256 VGPRs, ~10 %).
template <int KT, int KF, bool CK, bool BF = false>
__global__ __launch_bounds__(512)
void conv64_wgrad_ring_kernel(WgradRingArgs g) {
  constexpr int P = KT / 2, PADF = KF / 2;
  constexpr int CH = KF == 5 ? 32 : 64;            // input channels per workgroup
  constexpr int NH = 64 / CH;                      // workgroups per group
  constexpr int NACC = KF == 5 ? 7 : 4;
  constexpr int NQA = (kNF + KF - 1 + 3) / 4;      // 16-byte groups per input row segment: 17 / 16
  constexpr int NIA = (CH * NQA + 511) / 512;
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  constexpr bool DB = KF == 5;                     // double-buffered LDS: ring of KT+1 slots, two dz buffers
  constexpr int NS = DB ? KT + 1 : KT;             // ring slots
  constexpr int NDZ = DB ? 2 : 1;
  constexpr int D = DB ? 1 : 2;                    // steps in flight in registers
  unsigned* const sD = smem;                       // [NDZ][hi, lo][64 rows][kPW]
  unsigned* const sA = smem + NDZ * 2 * 64 * kPW;  // [NS slots][hi, lo][CH rows][kPW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int hh = NH == 2 ? (jj & 1) : 0;                       // which 32 input channels
  const int grp = (NH == 2 ? (jj >> 1) : jj) * 8 + xcd;
  int cb, nb, tg;
  if (KF == 5) { cb = wave & 1; nb = 0; tg = wave >> 1; }
  else { cb = (wave >> 1) & 1; nb = wave & 1; tg = wave >> 2; }
  const int rowd = (cb * 32 + l31) * kPW + 4 * half;            // dword index of this lane's dz fragment, K-block 0
  const int rowa = (nb * 32 + l31) * kPW + 4 * half;

  const size_t plane = (size_t)g.T * g.F;
  const unsigned plane_bytes = (unsigned)(plane * sizeof(float));
  const long long slab = 64ll * plane_bytes;
  const float s_dz = g.dz_scale[0], s_in = g.in_scale[0];
  const int NC = g.B * g.nseg * g.dil * (CK ? g.nchunk : 1);    // columns (x chunks of their steps)

  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // column c -> (utterance b, residue r, segment, chunk); rows m = 0..klast of the residue class
  // (frame t = r + m*dil), this chunk's steps k = k0..k1
  struct Col { int b, r, f0, nv, klast, k0, k1; };
  auto decode = [&](int c) {
    Col o;
    const int nchunk = CK ? g.nchunk : 1;
    const int chunk = CK ? c % nchunk : 0;
    if (CK) c /= nchunk;
    const int seg = c % g.nseg;
    const int rest = c / g.nseg;
    o.r = rest % g.dil;
    o.b = rest / g.dil;
    o.f0 = seg * kNF;
    o.nv = g.F - o.f0 < kNF ? g.F - o.f0 : kNF;
    o.klast = o.r < g.T ? (g.T - 1 - o.r) / g.dil : -1;
    const int len = (o.klast + nchunk) / nchunk;               // ceil((klast+1)/nchunk)
    o.k0 = chunk * len;
    o.k1 = o.k0 + len - 1 < o.klast ? o.k0 + len - 1 : o.klast;
    return o;
  };
  auto first_col = [&](int c) {                                 // skip empty chunks / residues beyond the last frame
    while (c < NC) {
      const Col o = decode(c);
      if (o.k0 <= o.k1) break;
      c += g.G;
    }
    return c;
  };

  struct Regs { f4 sd[2]; f4 sa[P + 1][NIA]; };
  // kind 0: step k (brings input row k+P and dz row k); 1: first step of a chunk (input rows
  // k..k+P); 2: a chunk that starts inside its column first brings rows k0-P..k0-1, one per event (k = row)
  struct Ev { Col c; int col, k, kind; bool valid; };
  // a 16-byte group that straddles the start or the end of the utterance's slab is split into
  // dwords (the hardware zeroes the whole out-of-range access, valid pixels included)
  auto load16 = [&](__amdgpu_buffer_rsrc_t r, long long off, bool live) -> f4 {
    if (!live || (off >= 0 && off + 16 <= slab))
      return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, live ? (unsigned)off : kOob, 0, 0));
    f4 x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long o = off + 4 * e;
      x[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (o >= 0 && o < slab) ? (unsigned)o : kOob, 0, 0));
    }
    return x;
  };
  // rows of one step: dz row k, and input rows 0..P (k == 0) or k+P
  auto issue = [&](const Ev& ev, Regs& R) {
    const Col& o = ev.c;
    const int k = ev.k;
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.dz + (size_t)o.b * 64 * plane), 0, 64u * plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.in + (size_t)o.b * 64 * plane), 0, 64u * plane_bytes, 0x00020000);
    const int t = o.r + k * g.dil;
    const unsigned base_d = (unsigned)((t * g.F + o.f0) * 4);
    if (!CK || ev.kind != 2) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 512 * i;
        R.sd[i] = load16(rd, (long long)(idx >> 4) * plane_bytes + base_d + (idx & 15) * 16, true);
      }
    }
#pragma unroll
    for (int j = 0; j <= P; ++j) {
      if (j > 0 && ev.kind != 1) break;                         // only a chunk's first step brings P+1 rows
      const int m = ev.kind == 0 ? k + P : k + j;
      const bool row_ok = m <= o.klast;
      const int base_a = ((o.r + m * g.dil) * g.F + o.f0 - PADF) * 4;      // may be negative at the very first pixels
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        const int idx = tid + 512 * i;
        const int ch = idx / NQA, q = idx - ch * NQA;
        R.sa[j][i] = load16(ra, (long long)(hh * 32 + ch) * plane_bytes + base_a + q * 16, row_ok && idx < CH * NQA);
      }
    }
  };
  // registers -> f16 hi/lo rows in LDS (dz buffer `par`, input row m -> slot m % NS)
  auto stash = [&](const Ev& ev, const Regs& R, int par) {
    const Col& o = ev.c;
    const int k = ev.k;
    const bool edge = o.f0 < PADF || o.f0 + kNF + KF - 1 - PADF > g.F;      // block-uniform: some pixel of the window is off the row
    unsigned* const zh = sD + (size_t)(par * 2) * 64 * kPW;
    unsigned* const zl = zh + 64 * kPW;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (CK && ev.kind == 2) break;
      const int idx = tid + 512 * i;
      const int q = idx & 15;
      f4 x = R.sd[i];
      if (edge) {
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = (o.f0 + 4 * q + e < g.F) ? x[e] : 0.f;
      }
      unsigned h0, l0, h1, l1;
      split_pair<BF>(x[0] * s_dz, x[1] * s_dz, h0, l0);
      split_pair<BF>(x[2] * s_dz, x[3] * s_dz, h1, l1);
      u2v hv, lv;
      hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
      *reinterpret_cast<u2v*>(&zh[(idx >> 4) * kPW + 2 * q]) = hv;
      *reinterpret_cast<u2v*>(&zl[(idx >> 4) * kPW + 2 * q]) = lv;
    }
#pragma unroll
    for (int j = 0; j <= P; ++j) {
      if (j > 0 && ev.kind != 1) break;
      const int m = ev.kind == 0 ? k + P : k + j;
      if (m > o.klast) continue;                                // never staged, never multiplied
      unsigned* const dh = sA + (size_t)((m % NS) * 2) * CH * kPW;
      unsigned* const dl = dh + CH * kPW;
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        const int idx = tid + 512 * i;
        if (idx < CH * NQA) {
          const int ch = idx / NQA, q = idx - ch * NQA;
          f4 x = R.sa[j][i];
          if (edge) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int f = o.f0 - PADF + 4 * q + e;
              x[e] = (f >= 0 && f < g.F) ? x[e] : 0.f;
            }
          }
          unsigned h0, l0, h1, l1;
          split_pair<BF>(x[0] * s_in, x[1] * s_in, h0, l0);
          split_pair<BF>(x[2] * s_in, x[3] * s_in, h1, l1);
          u2v hv, lv;
          hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
          *reinterpret_cast<u2v*>(&dh[ch * kPW + 2 * q]) = hv;
          *reinterpret_cast<u2v*>(&dl[ch * kPW + 2 * q]) = lv;
        }
      }
    }
  };
  // window of tap kf = halves kf .. kf+7 of the 12 read
  auto tap = [&](const unsigned (&w)[6], int kf) {
    const int m = kf >> 1;
    u4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = (kf & 1) ? __builtin_amdgcn_alignbit(w[m + q + 1], w[m + q], 16) : w[m + q];
    return __builtin_bit_cast(h8, r);
  };
  auto window = [&](int slot, int kb, unsigned (&wh)[6], unsigned (&wl)[6]) {
    const unsigned* ph = sA + (size_t)(slot * 2) * CH * kPW + rowa + 8 * kb;
    const unsigned* pl = ph + CH * kPW;
    const u4 a = *reinterpret_cast<const u4*>(ph);
    const u4 c = *reinterpret_cast<const u4*>(pl);
    wh[0] = a[0]; wh[1] = a[1]; wh[2] = a[2]; wh[3] = a[3];
    wl[0] = c[0]; wl[1] = c[1]; wl[2] = c[2]; wl[3] = c[3];
    const u2v a2 = *reinterpret_cast<const u2v*>(ph + 4);
    const u2v c2 = *reinterpret_cast<const u2v*>(pl + 4);
    wh[4] = a2[0]; wh[5] = a2[1];
    wl[4] = c2[0]; wl[5] = c2[1];
  };
  auto slot_of = [&](int m) { return ((m % NS) + NS) % NS; };
  // all taps of this wave for one step; its rows are in LDS
  auto compute = [&](const Ev& ev, int par) {
    if (CK && ev.kind == 2) return;
    const Col& o = ev.c;
    const int k = ev.k;
    const int nkb = (o.nv + 15) >> 4;
    const unsigned* const zh = sD + (size_t)(par * 2) * 64 * kPW + rowd;
    const unsigned* const zl = zh + 64 * kPW;
    if (KF == 5) {
      const int m0 = k + tg - P, m4 = k + P;                     // input rows of time taps kt = tg and kt = 4
      const bool ok0 = m0 >= 0 && m0 <= o.klast, ok4 = m4 <= o.klast;
      const int s0 = slot_of(m0), s4 = slot_of(m4);
#pragma unroll 1
      for (int kb = 0; kb < nkb; ++kb) {
        const h8 dhv = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(zh + 8 * kb));
        const h8 dlv = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(zl + 8 * kb));
        unsigned wh[6], wl[6], xh[6], xl[6];
        if (ok0) window(s0, kb, wh, wl);
        if (ok4) window(s4, kb, xh, xl);
#pragma unroll
        for (int term = BF ? 2 : 0; term < 3; ++term) {
          const h8 av = term == 0 ? dlv : dhv;
          if (ok0) {
#pragma unroll
            for (int kf = 0; kf < 5; ++kf)
              acc[kf] = wg_mma<BF>(av, term == 1 ? tap(wl, kf) : tap(wh, kf), acc[kf], 0, 0, 0);
          }
          if (ok4) {
            // kt = 4 is shared out: tap group 0 takes kf 0,1; groups 1..3 take kf 2,3,4
            switch (tg) {
              case 0:
                acc[5] = wg_mma<BF>(av, term == 1 ? tap(xl, 0) : tap(xh, 0), acc[5], 0, 0, 0);
                acc[6] = wg_mma<BF>(av, term == 1 ? tap(xl, 1) : tap(xh, 1), acc[6], 0, 0, 0);
                break;
              case 1: acc[5] = wg_mma<BF>(av, term == 1 ? tap(xl, 2) : tap(xh, 2), acc[5], 0, 0, 0); break;
              case 2: acc[5] = wg_mma<BF>(av, term == 1 ? tap(xl, 3) : tap(xh, 3), acc[5], 0, 0, 0); break;
              default: acc[5] = wg_mma<BF>(av, term == 1 ? tap(xl, 4) : tap(xh, 4), acc[5], 0, 0, 0); break;
            }
          }
        }
      }
    } else {
      bool ok[4];
      int sl[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int kt = tg * 4 + a;
        const int m = k + kt - P;
        ok[a] = kt < KT && m >= 0 && m <= o.klast;
        sl[a] = slot_of(m);
      }
#pragma unroll 1
      for (int kb = 0; kb < nkb; ++kb) {
        const h8 dhv = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(zh + 8 * kb));
        const h8 dlv = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(zl + 8 * kb));
        h8 bh[4], bl[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (ok[a]) {
            const unsigned* ph = sA + (size_t)(sl[a] * 2) * CH * kPW + rowa + 8 * kb;
            bh[a] = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(ph));
            bl[a] = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(ph + CH * kPW));
          }
        }
#pragma unroll
        for (int term = BF ? 2 : 0; term < 3; ++term)
#pragma unroll
          for (int a = 0; a < 4; ++a)
            if (ok[a]) acc[a] = wg_mma<BF>(term == 0 ? dlv : dhv, term == 1 ? bl[a] : bh[a], acc[a], 0, 0, 0);
      }
    }
  };
  auto first_ev = [&](int c) {
    Ev e;
    e.col = first_col(c);
    e.valid = e.col < NC;
    e.c = decode(e.valid ? e.col : 0);
    const int pre = e.c.k0 - P > 0 ? e.c.k0 - P : 0;             // first row this chunk has to bring before k0
    e.kind = (CK && pre < e.c.k0) ? 2 : 1;
    e.k = (CK && pre < e.c.k0) ? pre : e.c.k0;
    return e;
  };
  auto next_ev = [&](const Ev& e) {
    Ev n = e;
    if (CK && e.kind == 2) {
      if (e.k + 1 < e.c.k0) { n.k = e.k + 1; return n; }
      n.kind = 1; n.k = e.c.k0;
      return n;
    }
    if (e.k < e.c.k1) { n.kind = 0; n.k = e.k + 1; return n; }
    return first_ev(e.col + g.G);
  };

  // Software pipeline.  The rows of step e+1 are in registers R0 while step e is multiplied (7x1:
  // step e+2 is on its way in R1 as well: a step is shorter than a trip to HBM there).  With the
  // double-buffered LDS of the 5x5 case a wave writes them right after its own share of step e and
  // one barrier per step publishes them; a column's first step overwrites live slots and waits
  // for everyone, as every step of the single-buffered 7x1 ring does.
  Regs R0, R1;
  Ev cur = first_ev(grp);
  if (cur.valid) {
    issue(cur, R0);
    stash(cur, R0, 0);
    Ev nxt = next_ev(cur);
    if (D == 2 && nxt.valid) issue(nxt, R0);
    __syncthreads();
    int par = 0;
    while (true) {
      Ev nn = nxt;
      if (D == 2) {
        if (nxt.valid) {
          nn = next_ev(nxt);
          if (nn.valid) issue(nn, R1);
        }
      } else if (nxt.valid) {
        issue(nxt, R0);
      }
      compute(cur, par);
      const int npar = DB ? par ^ 1 : 0;
      if (nxt.valid) {
        if (!DB || nxt.kind != 0) __syncthreads();
        stash(nxt, R0, npar);
      }
      __syncthreads();
      if (!nxt.valid) break;
      if (D == 2) { R0 = R1; cur = nxt; nxt = nn; }
      else { cur = nxt; nxt = next_ev(cur); }
      par = npar;
    }
  }

  // acc[a] of this wave -> tap index, block (cb, ci block) of the group's slab
  float* out = g.part + (size_t)grp * (KT * KF) * 4096;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    int tp;
    if (KF == 5) {
      if (a < 5) tp = tg * 5 + a;
      else {
        const int kf = tg == 0 ? a - 5 : tg + 1;
        if (tg != 0 && a == 6) continue;
        tp = 4 * 5 + kf;
      }
    } else {
      tp = tg * 4 + a;
      if (tp >= KT) continue;
    }
    const int ci = (KF == 5 ? hh * 32 : nb * 32) + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[(size_t)tp * 4096 + co * 64 + ci] = acc[a][r];
    }
  }
}

// =================================================================================================
// 5x5 ring kernel, four waves per workgroup (one per SIMD, up to 512 registers each).
//
// Why a second form (round 2, profiles/r02_conv_ablation.md): in the eight-wave kernel above the K-block
// loop tests `ok0` / `ok4` and switches on the tap group INSIDE every product term, which cuts a term into
// 2-3 basic blocks of 1-5 MFMAs; with staging, fragment reads and barriers ablated away that stream still
// left the matrix pipe idle ~40 % of the cycles.  Making the loop straight-line needs registers the
// eight-wave form does not have (256 per wave, all used: the in-flight rows of the next step were spilled
// and every step waited for HBM).  Here a wave is (co block, tap half): half 0 owns time taps kt = 0,1 and
// kf = 0,1,2 of kt = 4, half 1 owns kt = 2,3 and kf = 3,4 of kt = 4 -- 13 / 12 accumulators (a 4 % imbalance
// instead of 7 / 6), 208 accumulator registers, no spill.  Interior steps (all three input rows of the wave
// inside the image) run a straight-line K-block loop specialised on the tap half; the first / last two
// steps of a column keep a generic loop.  Products are issued hi*hi, lo*hi, hi*lo so that the two that share
// the hi windows come first and the lo windows reuse their registers.  Same staging, ring, barriers,
// chunking and partial-sum layout as the eight-wave kernel; results agree to summation order.
// ABL: timing ablations (results are garbage unless 0): 1 = fragments read once per step, 2 = no staging
// (global loads, conversion, LDS writes), 8 = no barriers, 4 = loads only, 16 = conversion only, 64 = every load out of
// range, 128 = every load from the slab's first 64 KB, 256 = all staging behind K block 3, 512 = the step's bytes
// as one contiguous piece (profiles/r02_wgrad_ablation.md).
// =================================================================================================
template <bool CK, bool BF = false, int ABL = 0>
__global__ __launch_bounds__(256, 1)
void conv64_wgrad_ring4_kernel(WgradRingArgs g) {
  constexpr int KT = 5, KF = 5;
  constexpr int NTHR = 256;
  constexpr int P = KT / 2, PADF = KF / 2;
  constexpr int CH = KF == 5 ? 32 : 64;            // input channels per workgroup
  constexpr int NH = 64 / CH;                      // workgroups per group
  constexpr int NACC = 13;
  constexpr int NQA = (kNF + KF - 1 + 3) / 4;      // 16-byte groups per input row segment: 17 / 16
  constexpr int RPP = NTHR / NQA;                  // input channel rows one pass of the workgroup covers: 15
  constexpr int NIA = (CH + RPP - 1) / RPP;        // passes: thread -> (channel tid / NQA + RPP * pass, group tid % NQA),
                                                   // the SAME group (= pixel mask) in every pass
  constexpr int NID = 64 * 16 / NTHR;              // 16-byte groups of the dz row per thread
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  constexpr bool DB = KF == 5;                     // double-buffered LDS: ring of KT+1 slots, two dz buffers
  constexpr int NS = DB ? KT + 1 : KT;             // ring slots
  constexpr int NDZ = DB ? 2 : 1;
  unsigned* const sD = smem;                       // [NDZ][hi, lo][64 rows][kPW]
  unsigned* const sA = smem + NDZ * 2 * 64 * kPW;  // [NS slots][hi, lo][CH rows][kPW]

  const int tid = threadIdx.x;
  // every ring slot holds finite values from the start (border steps multiply stale slots by zero, see compute)
  for (int i = tid; i < (NDZ * 2 * 64 + NS * 2 * CH) * kPW; i += NTHR) smem[i] = 0u;
  __syncthreads();
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int hh = NH == 2 ? (jj & 1) : 0;                       // which 32 input channels
  const int grp = (NH == 2 ? (jj >> 1) : jj) * 8 + xcd;
  const int cb = wave & 1, nb = 0, th = wave >> 1;             // co block, (ci block), tap half
  const int rowd = (cb * 32 + l31) * kPW + 4 * half;            // dword index of this lane's dz fragment, K-block 0
  const int rowa = (nb * 32 + l31) * kPW + 4 * half;
  const int chl = tid / NQA, qa = tid - chl * NQA, qd = tid & 15;   // staging role: input (channel row, group), dz group
  unsigned* const dump = smem + (NDZ * 2 * 64 + NS * 2 * CH) * kPW;  // where idle staging lanes write (16 dwords)

  const size_t plane = (size_t)g.T * g.F;
  const unsigned plane_bytes = (unsigned)(plane * sizeof(float));
  const long long slab = 64ll * plane_bytes;
  const float s_dz = g.dz_scale[0], s_in = g.in_scale[0];
  const int NC = g.B * g.nseg * g.dil * (CK ? g.nchunk : 1);    // columns (x chunks of their steps)

  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // column c -> (utterance b, residue r, segment, chunk); rows m = 0..klast of the residue class
  // (frame t = r + m*dil), this chunk's steps k = k0..k1
  struct Col { int b, r, f0, nv, klast, k0, k1; };
  auto decode = [&](int c) {
    Col o;
    const int nchunk = CK ? g.nchunk : 1;
    const int chunk = CK ? c % nchunk : 0;
    if (CK) c /= nchunk;
    const int seg = c % g.nseg;
    const int rest = c / g.nseg;
    o.r = rest % g.dil;
    o.b = rest / g.dil;
    o.f0 = seg * kNF;
    o.nv = g.F - o.f0 < kNF ? g.F - o.f0 : kNF;
    o.klast = o.r < g.T ? (g.T - 1 - o.r) / g.dil : -1;
    const int len = (o.klast + nchunk) / nchunk;               // ceil((klast+1)/nchunk)
    o.k0 = chunk * len;
    o.k1 = o.k0 + len - 1 < o.klast ? o.k0 + len - 1 : o.klast;
    return o;
  };
  auto first_col = [&](int c) {                                 // skip empty chunks / residues beyond the last frame
    while (c < NC) {
      const Col o = decode(c);
      if (o.k0 <= o.k1) break;
      c += g.G;
    }
    return c;
  };

  struct Regs { f4 sd[NID]; f4 sa[P + 1][NIA]; };
  // kind 0: step k (brings input row k+P and dz row k); 1: first step of a chunk (input rows
  // k..k+P); 2: a chunk that starts inside its column first brings rows k0-P..k0-1, one per event (k = row)
  struct Ev { Col c; int col, k, kind; bool valid; };
  // a 16-byte group that straddles the start or the end of the utterance's slab is split into
  // dwords (the hardware zeroes the whole out-of-range access, valid pixels included)
  auto load16 = [&](__amdgpu_buffer_rsrc_t r, long long off, bool live) -> f4 {
    if (!live || (off >= 0 && off + 16 <= slab))
      return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, live ? (unsigned)off : kOob, 0, 0));
    f4 x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long o = off + 4 * e;
      x[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (o >= 0 && o < slab) ? (unsigned)o : kOob, 0, 0));
    }
    return x;
  };
  // rows of one step: dz row k, and input rows 0..P (k == 0) or k+P.
  // A 16-byte group can only straddle the utterance's slab in the first row (first segment) or the last row (last
  // segment) of the image -- a block-uniform test; everywhere else the loads are plain 32-bit-offset b128 loads
  // with no per-lane bounds logic.  (load16's per-lane test expanded to 40-90 instructions around EVERY load:
  // ~800 instructions per step and thread, as much issue time as the step's 156 MFMAs.)
  auto fast16 = [&](__amdgpu_buffer_rsrc_t r, int off, bool live) -> f4 {
    if (ABL & 64) live = false;                                  // ablation: every load out of range (no memory access)
    if (ABL & 128) off = (unsigned)off % (64u * 1024u) & ~15u;   // ablation: every load from the slab's first 64 KB
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, live ? (unsigned)off : kOob, 0, 0));
  };
  auto issue = [&](const Ev& ev, Regs& R) {
    if ((ABL & 2) || (ABL & 16)) return;
    const Col& o = ev.c;
    const int k = ev.k;
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.dz + (size_t)o.b * 64 * plane), 0, 64u * plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(g.in + (size_t)o.b * 64 * plane), 0, 64u * plane_bytes, 0x00020000);
    const int t = o.r + k * g.dil;
    const bool last_seg = o.f0 + kNF + KF - 1 - PADF > g.F;
    const unsigned base_d = (unsigned)((t * g.F + o.f0) * 4);
    if (!CK || ev.kind != 2) {
      const bool risky = !(ABL & (192 | 512)) && t == g.T - 1 && last_seg;
      // ablation 512: the same number of bytes per step, but as ONE contiguous 28 KB piece that walks through the
      // slab (cold in L2, a single page) instead of 96 row pieces on 96 channel planes
      const unsigned walk = (unsigned)(((unsigned)ev.col * 331u + (unsigned)k) % ((unsigned)(slab >> 15) - 2u)) << 15;
#pragma unroll
      for (int i = 0; i < NID; ++i) {
        const int idx = tid + NTHR * i;
        if (ABL & 512) R.sd[i] = fast16(rd, (int)(walk + idx * 16), true);
        else
        if (!risky) R.sd[i] = fast16(rd, (int)((idx >> 4) * plane_bytes + base_d + (idx & 15) * 16), true);
        else R.sd[i] = load16(rd, (long long)(idx >> 4) * plane_bytes + base_d + (idx & 15) * 16, true);
      }
    }
#pragma unroll
    for (int j = 0; j <= P; ++j) {
      if (j > 0 && ev.kind != 1) break;                         // only a chunk's first step brings P+1 rows
      const int m = ev.kind == 0 ? k + P : k + j;
      const bool row_ok = m <= o.klast;
      const int tm = o.r + m * g.dil;
      const int base_a = (tm * g.F + o.f0 - PADF) * 4;          // may be negative at the very first pixels
      const bool risky = !(ABL & (192 | 512)) && ((tm == 0 && o.f0 < PADF) || (tm >= g.T - 1 && last_seg));
      const unsigned walk_a = (unsigned)(((unsigned)ev.col * 331u + (unsigned)m) % ((unsigned)(slab >> 15) - 2u)) << 15;
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        const int ch = chl + RPP * i;
        const bool live = row_ok && chl < RPP && ch < CH;
        if (ABL & 512) R.sa[j][i] = fast16(ra, (int)(walk_a + 16384u + (unsigned)(ch * NQA + qa) * 16u), live);
        else
        if (!risky) R.sa[j][i] = fast16(ra, (int)((hh * 32 + ch) * plane_bytes) + base_a + qa * 16, live);
        else R.sa[j][i] = load16(ra, (long long)(hh * 32 + ch) * plane_bytes + base_a + qa * 16, live);
      }
    }
  };
  // Pixel masks of a column's segment as scale factors: the operand scale where the pixel is on the row, 0 where it
  // is not (what was loaded there is a finite value of the neighbouring row).  The staging multiplies by the scale
  // anyway, so masking costs nothing and needs no branch.  One group per thread and operand: 4 + 4 factors.
  struct Mask { float dz[4], in[4]; };
  auto masks_of = [&](const Col& o) {
    Mask mk;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mk.dz[e] = (o.f0 + 4 * qd + e < g.F) ? s_dz : 0.f;
      mk.in[e] = ((unsigned)(o.f0 - PADF + 4 * qa + e) < (unsigned)g.F) ? s_in : 0.f;
    }
    return mk;
  };
  // registers -> f16 hi/lo rows in LDS (dz buffer `par`, input row m -> slot m % NS)
  auto stash = [&](const Ev& ev, const Regs& R, int par) {
    if (ABL & 2) return;
    const Col& o = ev.c;
    const int k = ev.k;
    const Mask mk = masks_of(o);
    unsigned* const zh = sD + (size_t)(par * 2) * 64 * kPW;
    unsigned* const zl = zh + 64 * kPW;
#pragma unroll
    for (int i = 0; i < NID; ++i) {
      if (CK && ev.kind == 2) break;
      const int idx = tid + NTHR * i;
      const f4 x = R.sd[i];
      unsigned h0, l0, h1, l1;
      split_pair<BF>(x[0] * mk.dz[0], x[1] * mk.dz[1], h0, l0);
      split_pair<BF>(x[2] * mk.dz[2], x[3] * mk.dz[3], h1, l1);
      u2v hv, lv;
      hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
      *reinterpret_cast<u2v*>(&zh[(idx >> 4) * kPW + 2 * qd]) = hv;
      *reinterpret_cast<u2v*>(&zl[(idx >> 4) * kPW + 2 * qd]) = lv;
    }
#pragma unroll
    for (int j = 0; j <= P; ++j) {
      if (j > 0 && ev.kind != 1) break;
      const int m = ev.kind == 0 ? k + P : k + j;
      if (m > o.klast) continue;                                // never staged, never multiplied
      unsigned* const dh = sA + (size_t)((m % NS) * 2) * CH * kPW;
      unsigned* const dl = dh + CH * kPW;
#pragma unroll
      for (int i = 0; i < NIA; ++i) {
        const int ch = chl + RPP * i;
        if (chl < RPP && ch < CH) {
          const f4 x = R.sa[j][i];
          unsigned h0, l0, h1, l1;
          split_pair<BF>(x[0] * mk.in[0], x[1] * mk.in[1], h0, l0);
          split_pair<BF>(x[2] * mk.in[2], x[3] * mk.in[3], h1, l1);
          u2v hv, lv;
          hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
          *reinterpret_cast<u2v*>(&dh[ch * kPW + 2 * qa]) = hv;
          *reinterpret_cast<u2v*>(&dl[ch * kPW + 2 * qa]) = lv;
        }
      }
    }
  };
  // One piece of a regular step's (kind 0) staging: u < NID = a 16-byte group of its dz row, else a group of its
  // new input row.  compute() places these BETWEEN the MFMAs of the previous step's last two K blocks: with one wave
  // per SIMD nothing else covers them (the eight-wave kernel has the SIMD's second wave for that).  Straight-line on
  // purpose -- a piece with a branch in it ends the basic block, and the compiler then runs the whole staging AFTER
  // the K block's MFMAs instead of under them (that was 2.3 ms of an 8.2 ms launch).  So: pixels off the row are
  // multiplied by a zero scale (Mask), lanes without a group write to a dump word, and a row beyond the image is
  // staged as the zeros its out-of-range loads returned.  Both targets are free by construction: the other dz
  // buffer and the one ring slot the current step does not read.
  auto stash_unit = [&](const Ev& ev, const Regs& R, int par, int u, const Mask& mk) __attribute__((always_inline)) {
    if (ABL & 2) return;
    if (ABL & 4) {      // keep the loads alive (and waited for) without converting them
      if (u < NID) asm volatile("" :: "v"(R.sd[u])); else asm volatile("" :: "v"(R.sa[0][u - NID]));
      return;
    }
    if (u < NID) {
      unsigned* const zh = sD + (size_t)(par * 2) * 64 * kPW;
      unsigned* const zl = zh + 64 * kPW;
      const int idx = tid + NTHR * u;
      const f4 x = R.sd[u];
      unsigned h0, l0, h1, l1;
      split_pair<BF>(x[0] * mk.dz[0], x[1] * mk.dz[1], h0, l0);
      split_pair<BF>(x[2] * mk.dz[2], x[3] * mk.dz[3], h1, l1);
      u2v hv, lv;
      hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
      *reinterpret_cast<u2v*>(&zh[(idx >> 4) * kPW + 2 * qd]) = hv;
      *reinterpret_cast<u2v*>(&zl[(idx >> 4) * kPW + 2 * qd]) = lv;
    } else {
      const int i2 = u - NID;
      const int m = ev.k + P;
      unsigned* const dh = sA + (size_t)((m % NS) * 2) * CH * kPW;
      const int ch = chl + RPP * i2;
      const bool ok = chl < RPP && ch < CH;
      unsigned* const ph = ok ? dh + ch * kPW + 2 * qa : dump;
      unsigned* const pl = ok ? dh + CH * kPW + ch * kPW + 2 * qa : dump + 2;
      const f4 x = R.sa[0][i2];
      unsigned h0, l0, h1, l1;
      split_pair<BF>(x[0] * mk.in[0], x[1] * mk.in[1], h0, l0);
      split_pair<BF>(x[2] * mk.in[2], x[3] * mk.in[3], h1, l1);
      u2v hv, lv;
      hv[0] = h0; hv[1] = h1; lv[0] = l0; lv[1] = l1;
      *reinterpret_cast<u2v*>(ph) = hv;
      *reinterpret_cast<u2v*>(pl) = lv;
    }
  };
  // window of tap kf = halves kf .. kf+7 of the 12 read
  auto tap = [&](const unsigned (&w)[6], int kf) {
    const int m = kf >> 1;
    u4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = (kf & 1) ? __builtin_amdgcn_alignbit(w[m + q + 1], w[m + q], 16) : w[m + q];
    return __builtin_bit_cast(h8, r);
  };
  auto slot_of = [&](int m) { return ((m % NS) + NS) % NS; };
  // half-window reader: the hi (part 0) or lo (part 1) image of a ring slot, 12 halves of the lane's row
  auto window1 = [&](int slot, int kb, int part, unsigned (&w)[6]) {
    const unsigned* ph = sA + (size_t)(slot * 2 + part) * CH * kPW + rowa + 8 * kb;
    const u4 a = *reinterpret_cast<const u4*>(ph);
    const u2v a2 = *reinterpret_cast<const u2v*>(ph + 4);
    w[0] = a[0]; w[1] = a[1]; w[2] = a[2]; w[3] = a[3]; w[4] = a2[0]; w[5] = a2[1];
  };
  // all taps of this wave for one step; its rows are in LDS.  TH = the wave's tap half, compile-time.
  // nxe / Rn / npar: the rows of a regular next step, in registers; they are converted and written to LDS (dz buffer
  // npar, the free ring slot) one piece every third MFMA of K blocks 2 and 3.  When the next step is not a regular
  // one the caller passes a stand-in with the same free targets and the pieces write harmless values there.
  auto compute = [&](const Ev& ev, int par, auto THc, const Ev& nxe, const Regs& Rn, int npar) __attribute__((always_inline)) {
    constexpr int TH = decltype(THc)::value;
    constexpr int KFX0 = TH == 0 ? 0 : 3, NX = TH == 0 ? 3 : 2;      // this half's share of time tap kt = 4: kf KFX0 .. KFX0+NX-1
    if (CK && ev.kind == 2) return;
    const Col& o = ev.c;
    const int k = ev.k;
    const int nkb = (o.nv + 15) >> 4;
    const unsigned* const zh = sD + (size_t)(par * 2) * 64 * kPW + rowd;
    const unsigned* const zl = zh + 64 * kPW;
    // input rows of time taps kt = 2*TH, 2*TH + 1 and 4
    const int ma = k + 2 * TH - P, mb = ma + 1, m4 = k + P;
    const bool oka = ma >= 0 && ma <= o.klast, okb = mb >= 0 && mb <= o.klast, ok4 = m4 <= o.klast;
    const int sa_ = slot_of(ma), sb_ = slot_of(mb), s4 = slot_of(m4);
    // A row outside the image (the first / last two steps of a column) was never staged: its ring slot holds
    // some other row of the column -- finite f16 values, the whole ring is zeroed when the kernel starts -- and
    // its taps are multiplied by a ZERO dz fragment instead of being branched around: ONE straight-line K-block
    // loop for every step.  (Two loops -- a fast one and a branchy one for the borders -- made the compiler
    // keep a second copy of the 208 accumulator registers: 512 registers and spills.)
    const unsigned ka = oka ? 0xffffffffu : 0u, kb_ = okb ? 0xffffffffu : 0u, k4 = ok4 ? 0xffffffffu : 0u;
    const Mask mk = masks_of(nxe.c);
    constexpr int NU = NID + NIA;                 // staging pieces of a regular step
    constexpr int U2 = (ABL & 256) ? 0 : (NU + 1) / 2;   // pieces 0..U2-1 behind K block 2, the rest behind K block 3
    constexpr int PER = BF ? 1 : 3;               // one piece every PER MFMAs
    constexpr int NMF = 10 + NX;                  // MFMAs of a product term
    // the piece that goes behind MFMA number i of K block kb (if any)
    auto hook = [&](int kb, int i) __attribute__((always_inline)) {
      if (kb < 2 || i % PER != 0) return;
      const int u = (kb == 2 ? 0 : U2) + i / PER;
      if (u < (kb == 2 ? U2 : NU)) stash_unit(nxe, Rn, npar, u, mk);
    };
    // one product term: dz fragment a4 against the windows wa, wb (5 taps each) and x (this half's share of kt = 4)
    auto term = [&](u4 a4, const unsigned (&wa)[6], const unsigned (&wb)[6], const unsigned (&x)[6], int kb, int i0) __attribute__((always_inline)) {
      u4 ma_, mb_, mx_;
#pragma unroll
      for (int q = 0; q < 4; ++q) { ma_[q] = a4[q] & ka; mb_[q] = a4[q] & kb_; mx_[q] = a4[q] & k4; }
      const h8 ava = __builtin_bit_cast(h8, ma_), avb = __builtin_bit_cast(h8, mb_), avx = __builtin_bit_cast(h8, mx_);
#pragma unroll
      for (int kf = 0; kf < 5; ++kf) { acc[kf] = wg_mma<BF>(ava, tap(wa, kf), acc[kf], 0, 0, 0); hook(kb, i0 + kf); }
#pragma unroll
      for (int kf = 0; kf < 5; ++kf) { acc[5 + kf] = wg_mma<BF>(avb, tap(wb, kf), acc[5 + kf], 0, 0, 0); hook(kb, i0 + 5 + kf); }
#pragma unroll
      for (int j = 0; j < NX; ++j) { acc[10 + j] = wg_mma<BF>(avx, tap(x, KFX0 + j), acc[10 + j], 0, 0, 0); hook(kb, i0 + 10 + j); }
    };
    // pieces of K block kb that found no MFMA to hide behind (the tap half with 12 MFMAs a term), or all of them
    auto rest = [&](int kb, int i_from) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < PER * NU; ++i)
        if (i >= i_from) hook(kb, i);
    };
    // Fragments are read one product term ahead: the lo windows of a K block while its two hi-window terms run, the
    // hi windows and dz fragments of the NEXT K block while the lo-window term runs (the hi windows' registers are
    // free by then).  Reading them where they are used left the matrix pipe idle for a trip to LDS twice per K block:
    // 8 x ~450 of a step's ~6500 cycles.
    unsigned ha[6], hb[6], hx[6];
    window1(sa_, 0, 0, ha); window1(sb_, 0, 0, hb); window1(s4, 0, 0, hx);
    u4 dh4 = *reinterpret_cast<const u4*>(zh), dl4 = dh4;
    if (!BF) dl4 = *reinterpret_cast<const u4*>(zl);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {                // a segment is at most 64 pixels = 4 K blocks
      if (kb < nkb) {
        const int kbr = (ABL & 1) ? 0 : kb, kbn = (ABL & 1) ? 0 : kb + 1;
        if (!BF) {
          unsigned la[6], lb[6], lx[6];
          window1(sa_, kbr, 1, la); window1(sb_, kbr, 1, lb); window1(s4, kbr, 1, lx);
          __builtin_amdgcn_sched_barrier(0);        // reads first: left alone, the scheduler sinks them to their first use
          term(dh4, ha, hb, hx, kb, 0);
          term(dl4, ha, hb, hx, kb, NMF);
          __builtin_amdgcn_sched_barrier(0);        // the next K block's hi windows take over the registers of this one's
          u4 dhn = dh4, dln = dl4;
          if (kb < 3) {
            window1(sa_, kbn, 0, ha); window1(sb_, kbn, 0, hb); window1(s4, kbn, 0, hx);
            dhn = *reinterpret_cast<const u4*>(zh + 8 * kbn);
            dln = *reinterpret_cast<const u4*>(zl + 8 * kbn);
          }
          __builtin_amdgcn_sched_barrier(0);
          term(dh4, la, lb, lx, kb, 2 * NMF);
          rest(kb, 3 * NMF);
          dh4 = dhn; dl4 = dln;
        } else {
          unsigned na[6], nb_[6], nx[6];
          u4 dhn = dh4;
          if (kb < 3) {
            window1(sa_, kbn, 0, na); window1(sb_, kbn, 0, nb_); window1(s4, kbn, 0, nx);
            dhn = *reinterpret_cast<const u4*>(zh + 8 * kbn);
          }
          __builtin_amdgcn_sched_barrier(0);
          term(dh4, ha, hb, hx, kb, 0);
          rest(kb, NMF);
          if (kb < 3) {
#pragma unroll
            for (int q = 0; q < 6; ++q) { ha[q] = na[q]; hb[q] = nb_[q]; hx[q] = nx[q]; }
          }
          dh4 = dhn;
        }
      } else {
        rest(kb, 0);                                // a short segment: nothing to hide behind
      }
    }
  };
  auto first_ev = [&](int c) {
    Ev e;
    e.col = first_col(c);
    e.valid = e.col < NC;
    e.c = decode(e.valid ? e.col : 0);
    const int pre = e.c.k0 - P > 0 ? e.c.k0 - P : 0;             // first row this chunk has to bring before k0
    e.kind = (CK && pre < e.c.k0) ? 2 : 1;
    e.k = (CK && pre < e.c.k0) ? pre : e.c.k0;
    return e;
  };
  auto next_ev = [&](const Ev& e) {
    Ev n = e;
    if (CK && e.kind == 2) {
      if (e.k + 1 < e.c.k0) { n.k = e.k + 1; return n; }
      n.kind = 1; n.k = e.c.k0;
      return n;
    }
    if (e.k < e.c.k1) { n.kind = 0; n.k = e.k + 1; return n; }
    return first_ev(e.col + g.G);
  };

  // Software pipeline.  The rows of step e+1 are in registers R0 while step e is multiplied (7x1:
  // step e+2 is on its way in R1 as well: a step is shorter than a trip to HBM there).  With the
  // double-buffered LDS of the 5x5 case a wave writes them right after its own share of step e and
  // one barrier per step publishes them; a column's first step overwrites live slots and waits
  // for everyone, as every step of the single-buffered 7x1 ring does.
  // Software pipeline: the rows of step e+1 are loaded into registers when step e starts and are converted and
  // written to LDS inside compute(e) (behind its last two K blocks); one barrier per step publishes them.  A
  // column's first step (and a chunk's pre-load events) overwrite live ring slots: they wait for everyone and are
  // staged in one piece, as in the eight-wave kernel.  Instantiated per tap half; a wave picks its copy once, so
  // TH is a compile-time constant inside.
  auto pipeline = [&](auto THw) __attribute__((always_inline)) {
    Regs R0;
    Ev cur = first_ev(grp);
    if (!cur.valid) return;
    issue(cur, R0);
    stash(cur, R0, 0);
    Ev nxt = next_ev(cur);
    __syncthreads();
    int par = 0;
    while (true) {
      Ev nn = nxt;
      if (nxt.valid) {
        issue(nxt, R0);
        nn = next_ev(nxt);
      }
      const int npar = par ^ 1;
      // a regular next step is staged inside compute(); anything else (a column's first step, a chunk's pre-load
      // event) after it, in one piece.  compute()'s staging pieces then get a stand-in with the same free targets
      // -- the other dz buffer and the ring slot of row cur.k + 1 + P -- and write values nobody reads there.
      const bool inl = nxt.valid && nxt.kind == 0;
      Ev standin = cur;
      standin.k = cur.k + 1;
      compute(cur, par, THw, inl ? nxt : standin, R0, npar);
      if (nxt.valid && !inl) {
        if (!(ABL & 8)) __syncthreads();
        stash(nxt, R0, npar);
      }
      if (!(ABL & 8)) __syncthreads();
      if (!nxt.valid) break;
      cur = nxt;
      nxt = nn;
      par = npar;
    }
  };
  if (th == 0) pipeline(std::integral_constant<int, 0>());
  else pipeline(std::integral_constant<int, 1>());

  // acc[a] of this wave -> tap index, block (cb, ci block) of the group's slab
  float* out = g.part + (size_t)grp * (KT * KF) * 4096;
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
    int tp;
    if (a < 10) tp = (2 * th + a / 5) * 5 + a % 5;              // time taps 2*th and 2*th + 1
    else {
      const int j = a - 10;
      if (th == 1 && j >= 2) continue;                           // half 1 owns two taps of kt = 4
      tp = 4 * 5 + (th == 0 ? j : 3 + j);
    }
    const int ci = hh * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[(size_t)tp * 4096 + co * 64 + ci] = acc[a][r];
    }
  }
}

}  // namespace

static int g_wgrad_kernel = 0;
// 0: choose by problem size (default), 1: ring kernel, 2: kt-split kernel.  Both compute the same
// sums in a fixed order each; the unit tests run every shape through both.
extern "C" int vs_set_wgrad_kernel(int mode) {
  // 3 = the four-wave 5x5 ring kernel (default where it applies), 1 = the eight-wave ring kernel, 2 = kt-split;
  // 100 + ABL: timing ablations of the four-wave kernel
  if ((mode < 0 || mode > 3) && (mode < 100 || mode > 100 + 1023)) return -1;
  g_wgrad_kernel = mode;
  return 0;
}

// groups for the f16 kernel: 2 workgroups per CU resident (LDS 36.9 KB, <= 256 VGPRs)
extern "C" int vs_conv64_wgrad_f16_groups(int KT) { return KT == 7 ? 72 : 96; }

int vs_conv64_wgrad_f16x3_impl(const float* dz, const float* in, const float* dz_scale2, const float* in_scale2,
                               float* part, float* dw, int B, int T, int F, int KT, int KF, int dil, hipStream_t stream, int math) {
  const bool bf = math == VS_MATH_CODE_BF16;
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "conv64_wgrad_f16x3: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((KT == 7 && KF == 1) || (KT == 5 && KF == 5), "conv64_wgrad_f16x3: unsupported kernel %dx%d", KT, KF);
  VS_REQUIRE((long long)64 * T * F * 4 < (long long)kOob, "conv64_wgrad_f16x3: T*F=%lld too large for 32-bit offsets", (long long)T * F);
  VS_REQUIRE((long long)B * T * ((F + kNF - 1) / kNF) < 2147483647LL, "conv64_wgrad_f16x3: too many tiles");
  // ring kernel: 256 workgroups (one per CU); 5x5: 128 groups of two 32-channel workgroups.
  // Columns are cut along their steps until every group has ~4 of them (chunks of >= 32 steps);
  // a problem too small to give every group two (a few utterances) goes to the kt-split kernel,
  // whose (utterance, frame, segment) tiles spread over the chip at any batch size.
  const int nseg = (F + kNF - 1) / kNF;
  const int Gr = KF == 5 ? 128 : 256;
  const long long ncol = (long long)B * nseg * dil;
  const int steps = (T + dil - 1) / dil;
  int nchunk = (int)((4LL * Gr + ncol - 1) / ncol);
  if (nchunk > steps / 32) nchunk = steps / 32;
  if (nchunk < 1) nchunk = 1;
  const bool ring = g_wgrad_kernel == 1 || g_wgrad_kernel == 3 || g_wgrad_kernel >= 100 ||
                    (g_wgrad_kernel == 0 && ncol * nchunk >= 2LL * Gr);
  if (ring) {
    const int G = Gr;
    VS_REQUIRE(ncol * nchunk < 2147483647LL, "conv64_wgrad_f16x3: too many columns");
    WgradRingArgs a{dz, in, dz_scale2, in_scale2, part, B, T, F, dil, nseg, G, nchunk};
    // LDS: dz rows (two buffers for 5x5) + the input-row ring (KT+1 slots of 32 channels / KT of 64)
    const size_t lds = KF == 5 ? (size_t)(2 * 2 * 64 + (KT + 1) * 2 * 32) * kPW * 4 + 256 : (size_t)(2 * 64 + KT * 2 * 64) * kPW * 4;
    auto launch = [&](auto kernel, int threads = 512) -> int {
      VS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), lds, stream, a);
      return 0;
    };
    int rc;
    if (KF == 5 && g_wgrad_kernel != 1) {         // four waves, straight-line K-block loop
      if (g_wgrad_kernel >= 100 && nchunk == 1 && !bf) {
        switch (g_wgrad_kernel - 100) {
          case 1: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 1>, 256); break;
          case 2: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 2>, 256); break;
          case 3: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 3>, 256); break;
          case 8: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 8>, 256); break;
          case 11: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 11>, 256); break;
          case 4: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 4>, 256); break;
          case 16: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 16>, 256); break;
          case 64: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 64>, 256); break;
          case 128: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 128>, 256); break;
          case 256: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 256>, 256); break;
          case 512: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 512>, 256); break;
          case 516: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 516>, 256); break;
          case 68: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 68>, 256); break;
          case 132: rc = launch(&conv64_wgrad_ring4_kernel<false, false, 132>, 256); break;
          default: rc = launch(&conv64_wgrad_ring4_kernel<false>, 256); break;
        }
      } else if (bf) rc = nchunk > 1 ? launch(&conv64_wgrad_ring4_kernel<true, true>, 256) : launch(&conv64_wgrad_ring4_kernel<false, true>, 256);
      else rc = nchunk > 1 ? launch(&conv64_wgrad_ring4_kernel<true>, 256) : launch(&conv64_wgrad_ring4_kernel<false>, 256);
    } else
    if (bf) {
      if (KF == 5) rc = nchunk > 1 ? launch(&conv64_wgrad_ring_kernel<5, 5, true, true>) : launch(&conv64_wgrad_ring_kernel<5, 5, false, true>);
      else rc = nchunk > 1 ? launch(&conv64_wgrad_ring_kernel<7, 1, true, true>) : launch(&conv64_wgrad_ring_kernel<7, 1, false, true>);
    } else {
      if (KF == 5) rc = nchunk > 1 ? launch(&conv64_wgrad_ring_kernel<5, 5, true>) : launch(&conv64_wgrad_ring_kernel<5, 5, false>);
      else rc = nchunk > 1 ? launch(&conv64_wgrad_ring_kernel<7, 1, true>) : launch(&conv64_wgrad_ring_kernel<7, 1, false>);
    }
    if (rc) return rc;
    const int total = KT * KF * 4096;
    hipLaunchKernelGGL(conv64_wgrad_reduce_scaled_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, part, G, KT * KF, dw,
                       dz_scale2, in_scale2);
    VS_LAUNCH_CHECK();
    return 0;
  }
  const int G = vs_conv64_wgrad_f16_groups(KT);
  Wgrad16Args a{dz, in, dz_scale2, in_scale2, part, B, T, F, dil, KT, (F + kNF - 1) / kNF, G};
  dim3 grid(G * KT), block(256);
  if (bf) {
    if (KF == 5) hipLaunchKernelGGL((conv64_wgrad_f16x3_kernel<5, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv64_wgrad_f16x3_kernel<1, true>), grid, block, 0, stream, a);
  } else if (KF == 5) hipLaunchKernelGGL(conv64_wgrad_f16x3_kernel<5>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(conv64_wgrad_f16x3_kernel<1>, grid, block, 0, stream, a);
  const int total = KT * KF * 4096;
  hipLaunchKernelGGL(conv64_wgrad_reduce_scaled_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, part, G, KT * KF, dw,
                     dz_scale2, in_scale2);
  VS_LAUNCH_CHECK();
  return 0;
}
