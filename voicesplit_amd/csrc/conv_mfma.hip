// 64->64 channel "same" convolution with time dilation, fused BatchNorm(eval)+activation.
//
// Replaces, per layer, ZeroPad2d + Conv2d + BatchNorm2d + Mish/ReLU of
//   models/voicesplit/model.py:21-23 (cnn2, 7x1) and :26-48 (cnn3..cnn7, 5x5, dilation (d,1))
//   models/voicefilter/model.py:25-50 (same, ReLU).
//
// Formulation (im2col-free implicit GEMM on the fp32 matrix cores):
//   out[co][t][f] = sum_{ci,kt,kf} W[co][ci][kt][kf] * in[ci][t+(kt-KT/2)*dil][f+kf-KF/2]
//   M = co (64, two 32-row MFMA blocks), N = 32 consecutive f of one output row,
//   K = ci pair per v_mfma_f32_32x32x2_f32 step, looped over taps and ci chunks.
// A dilated layer is decomposed into `dil` residue classes of t (t = cls + dil*i): inside a
// class it is an ordinary dense conv over rows i, so an LDS tile of R output rows needs only
// R+KT-1 input rows whatever the dilation.
//
// Workgroup = 256 threads (4 waves, one per SIMD; two workgroups per CU so one computes while
// the other stages/stores).  Tile = 64 co x (R = 4*P rows) x 32 f.  Wave w owns rows
// w*P..w*P+P-1 for both co blocks: 2*P accumulators of 32x32 (16 VGPR each).
// The ci dimension is walked in 8 chunks of 8 channels: the chunk's input window
// [8][R+KT-1][32+KF-1] is staged global->VGPR->LDS with raw buffer loads whose out-of-range
// offsets return 0 -- that zero fill outside the image IS the reference's ZeroPad2d; loads for
// chunk c+1 are in flight while chunk c is multiplied.
// B fragments (activations) are ds_read_b32 with compile-time offsets: lanes 0-31 read 32
// consecutive f (conflict free), lanes 32-63 the next input channel.  A fragments (weights)
// come pre-packed in fragment order (conv_pack_weights_kernel) as one coalesced dwordx4 per
// lane per 4 K-steps (= one tap of one chunk) straight from L2/L1, prefetched one tap ahead.
#include <stdlib.h>

#include "vs_common.h"

namespace {

constexpr int kCo = 64;
constexpr int kCi = 64;
constexpr int kChunk = 8;               // input channels per LDS stage
constexpr int kNChunk = kCi / kChunk;   // 8
constexpr int kPairs = kChunk / 2;      // 4 MFMA K-steps per (chunk, tap) = one float4 of A
constexpr unsigned kOob = 0x7FFFFFF0u;  // buffer offset guaranteed >= num_records -> load returns 0
constexpr int kTileF = 32;
constexpr int kPadTaps = 4;              // dummy tap blocks behind the packed weights (prefetch overrun)

// packed weight layout: [chunk][tap][cb(2)][lane(64)][4]; element j of the float4 is
//   W[co = cb*32 + (lane&31)][ci = chunk*8 + 2*j + (lane>>5)][kt][kf]
// plus kPadTaps dummy tap blocks at the end so the weight prefetch never reads out of bounds.
__global__ void conv_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int KT, int KF) {
  const int NT = KT * KF;
  const int total = kNChunk * NT * 2 * 64 * 4;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total + kPadTaps * 512) return;
  if (idx >= total) { wp[idx] = 0.f; return; }
  int j = idx & 3;
  int lane = (idx >> 2) & 63;
  int cb = (idx >> 8) & 1;
  int tg = idx >> 9;           // chunk*NT + tap
  int chunk = tg / NT, tap = tg - chunk * NT;
  int kt = tap / KF, kf = tap - kt * KF;
  int co = cb * 32 + (lane & 31);
  int ci = chunk * kChunk + 2 * j + (lane >> 5);
  wp[idx] = w[((co * kCi + ci) * KT + kt) * KF + kf];
}

template <int KT, int KF, int P, int ACT, int VAR, int NW>
__global__ __launch_bounds__(NW * 64, NW == 1 ? 3 : 2)
void conv64_mfma_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                        const float* __restrict__ scale, const float* __restrict__ shift,
                        float* __restrict__ out, int T, int F, int dil, int n_rt, int n_ft) {
  constexpr int NTHR = NW * 64;
  constexpr int R = NW * P;
  constexpr int ROWS = R + KT - 1;
  constexpr int PITCH = kTileF + KF - 1;
  constexpr int NELEM = kChunk * ROWS * PITCH;
  constexpr int NPT = (NELEM + NTHR - 1) / NTHR;
  constexpr int NT = KT * KF;
  __shared__ float sIn[NELEM + P * PITCH];     // + P rows: the one-tap-ahead B prefetch may overrun

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
  const int n_c = (T - cls + dil - 1) / dil;   // rows of this residue class
  const int i0 = rt * R;
  if (i0 >= n_c) return;                       // uniform: nothing to do for this tile
  const int f0 = ft * kTileF;
  const size_t plane = (size_t)T * F;
  const float* in_b = in + (size_t)b * kCi * plane;

  // ---- staging: element e of the LDS window <-> (ci_l, rr, x); byte offset inside the
  //      chunk's [8][T][F] slab, or kOob where the window leaves the image --------------------
  unsigned voff[NPT];
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int e = tid + NTHR * i;
    const int ci_l = e / (ROWS * PITCH);
    const int rem = e - ci_l * (ROWS * PITCH);
    const int rr = rem / PITCH;
    const int x = rem - rr * PITCH;
    const int iin = i0 - KT / 2 + rr;
    const int f = f0 - KF / 2 + x;
    const bool ok = (e < NELEM) && (iin >= 0) && (iin < n_c) && (f >= 0) && (f < F);
    const int t = cls + dil * iin;
    voff[i] = ok ? (unsigned)(((ci_l * T + t) * F + f) * 4) : kOob;
  }
  const unsigned slab_bytes = (unsigned)(kChunk * plane * sizeof(float));
  float stage[NPT];
  auto load_chunk = [&](int chunk) {
    const float* src = in_b + (size_t)chunk * kChunk * plane;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, slab_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPT; ++i)
      stage[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[i], 0, 0));
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int e = tid + NTHR * i;
      if (e < NELEM) sIn[e] = stage[i];
    }
  };

  f32x16 acc[2][P];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][p][r] = 0.f;

  const bool wave_active = (i0 + wave * P) < n_c;   // rows ascend: first row decides
  const float4* wp4 = reinterpret_cast<const float4*>(wp) + lane;
  // this lane's B-fragment base inside the window
  const float* sB = sIn + (half * ROWS + wave * P) * PITCH + l31;

  float4 a_cur[2], a_nxt[2];
  a_cur[0] = wp4[0];
  a_cur[1] = wp4[64];
  // VAR 3: weight fragments DEPTH taps ahead in a register ring (L2 latency under load exceeds one tap)
  constexpr int RING = (NT % 5 == 0) ? 5 : 7;
  constexpr int DEPTH = 3;
  static_assert(NT % RING == 0 && DEPTH < RING && DEPTH <= kPadTaps, "ring must tile the taps");
  float4 a_ring[RING][2];
  if (VAR == 3) {
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) {
      a_ring[t][0] = wp4[(size_t)t * 128];
      a_ring[t][1] = wp4[(size_t)t * 128 + 64];
    }
  }

  if (VAR < 10 || ((VAR - 10) & 4)) load_chunk(0);
#pragma unroll 1
  for (int chunk = 0; chunk < kNChunk; ++chunk) {
    constexpr bool kAblate = VAR >= 10;                 // timing experiments only (wrong results)
    constexpr bool kDoB = !kAblate || ((VAR - 10) & 1);
    constexpr bool kDoA = !kAblate || ((VAR - 10) & 2);
    constexpr bool kDoS = !kAblate || ((VAR - 10) & 4);
    if (kDoS) {
      __syncthreads();            // previous chunk's window fully consumed
      store_chunk();
      __syncthreads();
      if (VAR != 3 && chunk + 1 < kNChunk) load_chunk(chunk + 1);   // in flight during the MFMA block below
    }
    if (VAR == 3 && !wave_active) {
      if (chunk + 1 < kNChunk) load_chunk(chunk + 1);      // rows past the class: staging duty only
    } else if (VAR == 3) {
      // Fully unrolled tap loop, software pipelined by hand:
      //  * vmcnt retires in order, so a wait for a weight fragment also waits for every older
      //    staging load of the next chunk.  Instead of one burst of NPT staging loads before
      //    the first tap they are spread SPT per tap behind that tap's weight prefetch, each
      //    with a full tap of MFMAs (>= 1024 cycles) to land before the next wait;
      //  * the B fragments of K-step s+1 are read from LDS before the MFMAs of K-step s issue.
      constexpr int SPT = (NPT + NT - 1) / NT < 2 ? 2 : (NPT + NT - 1) / NT;
      const bool more = chunk + 1 < kNChunk;
      const float* src = in_b + (size_t)(chunk + 1) * kChunk * plane;
      __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, more ? slab_bytes : 0u, 0x00020000);
      float b_cur[P], b_nxt[P];
#pragma unroll
      for (int p = 0; p < P; ++p) b_cur[p] = sB[p * PITCH];
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        const int tg = chunk * NT + tap;
        // weight fragment of tap tg+DEPTH -> ring slot (tap+DEPTH) % RING (NT % RING == 0, so the
        // slot numbering is the same in every chunk and stays a compile-time register index)
        {
          const float4* nxt = wp4 + (size_t)(tg + DEPTH) * 128;
          a_ring[(tap + DEPTH) % RING][0] = nxt[0];
          a_ring[(tap + DEPTH) % RING][1] = nxt[64];
        }
#pragma unroll
        for (int i = tap * SPT; i < (tap + 1) * SPT && i < NPT; ++i)
          stage[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[i], 0, 0));
        const float4 a0v = a_ring[tap % RING][0], a1v = a_ring[tap % RING][1];
#pragma unroll
        for (int pr = 0; pr < kPairs; ++pr) {
          const int ntap = (pr + 1 < kPairs) ? tap : tap + 1;       // K-step after this one
          const int npr = (pr + 1 < kPairs) ? pr + 1 : 0;
          const int nkt = ntap / KF, nkf = ntap % KF;               // ntap == NT reads the pad rows
#pragma unroll
          for (int p = 0; p < P; ++p)
            b_nxt[p] = sB[((2 * npr) * ROWS + p + nkt) * PITCH + nkf];
          __builtin_amdgcn_sched_barrier(0);
          const float a0 = (pr == 0) ? a0v.x : (pr == 1) ? a0v.y : (pr == 2) ? a0v.z : a0v.w;
          const float a1 = (pr == 0) ? a1v.x : (pr == 1) ? a1v.y : (pr == 2) ? a1v.z : a1v.w;
#pragma unroll
          for (int p = 0; p < P; ++p) {
            acc[0][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b_cur[p], acc[0][p], 0, 0, 0);
            acc[1][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b_cur[p], acc[1][p], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < P; ++p) b_cur[p] = b_nxt[p];
        }
      }
    } else if (wave_active && VAR == 2) {
      // software pipeline, one tap deep: while the 4*2P MFMAs of tap t run, the A fragment
      // (global, L1/L2) and the 4*P B fragments (LDS) of tap t+1 are already in flight.
      float b_cur[kPairs][P], b_nxt[kPairs][P];
#pragma unroll
      for (int pr = 0; pr < kPairs; ++pr)
#pragma unroll
        for (int p = 0; p < P; ++p) b_cur[pr][p] = sB[((2 * pr) * ROWS + p) * PITCH];
#pragma unroll 1
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) {
          const int tg = chunk * NT + kt * KF + kf;
          // first K-step of this tap
#pragma unroll
          for (int p = 0; p < P; ++p) {
            acc[0][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0].x, b_cur[0][p], acc[0][p], 0, 0, 0);
            acc[1][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1].x, b_cur[0][p], acc[1][p], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          // prefetch tap t+1 (the tap after the chunk's last reads P rows past the window: the
          // LDS array is padded for it and the values are discarded)
          const float4* nxt = wp4 + (size_t)(tg + 1) * 128;
          a_nxt[0] = nxt[0];
          a_nxt[1] = nxt[64];
          const int nkt = (kf + 1 < KF) ? 0 : 1;          // next tap: same row block or the next
          const int nkf = (kf + 1 < KF) ? kf + 1 : 0;
#pragma unroll
          for (int pr = 0; pr < kPairs; ++pr)
#pragma unroll
            for (int p = 0; p < P; ++p)
              b_nxt[pr][p] = sB[((2 * pr) * ROWS + p + kt + nkt) * PITCH + nkf];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int pr = 1; pr < kPairs; ++pr) {
            const float a0 = (pr == 1) ? a_cur[0].y : (pr == 2) ? a_cur[0].z : a_cur[0].w;
            const float a1 = (pr == 1) ? a_cur[1].y : (pr == 2) ? a_cur[1].z : a_cur[1].w;
#pragma unroll
            for (int p = 0; p < P; ++p) {
              acc[0][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b_cur[pr][p], acc[0][p], 0, 0, 0);
              acc[1][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b_cur[pr][p], acc[1][p], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          a_cur[0] = a_nxt[0];
          a_cur[1] = a_nxt[1];
#pragma unroll
          for (int pr = 0; pr < kPairs; ++pr)
#pragma unroll
            for (int p = 0; p < P; ++p) b_cur[pr][p] = b_nxt[pr][p];
        }
      }
    } else if (wave_active) {
#pragma unroll 1
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) {
          const int tg = chunk * NT + kt * KF + kf;
          const float4* nxt = wp4 + (VAR == 5 ? (size_t)0 : (size_t)(tg + 1) * 128);   // VAR 5: perf experiment only (L1-resident weights, wrong results)
          if (kDoA) {
            a_nxt[0] = nxt[0];
            a_nxt[1] = nxt[64];
          } else {
            a_nxt[0] = a_cur[1];
            a_nxt[1] = a_cur[0];
          }
          if (VAR == 1 || VAR == 5 || kAblate) __builtin_amdgcn_sched_barrier(0);   // keep the prefetch issued ahead of this tap's MFMAs
#pragma unroll
          for (int pr = 0; pr < kPairs; ++pr) {
            float bfrag[P];
#pragma unroll
            for (int p = 0; p < P; ++p)
              bfrag[p] = kDoB ? sB[((2 * pr) * ROWS + p + kt) * PITCH + kf] : (pr == 0 ? a_cur[0].w : a_cur[1].w) + (float)p;
            const float a0 = (pr == 0) ? a_cur[0].x : (pr == 1) ? a_cur[0].y : (pr == 2) ? a_cur[0].z : a_cur[0].w;
            const float a1 = (pr == 0) ? a_cur[1].x : (pr == 1) ? a_cur[1].y : (pr == 2) ? a_cur[1].z : a_cur[1].w;
#pragma unroll
            for (int p = 0; p < P; ++p) {
              acc[0][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bfrag[p], acc[0][p], 0, 0, 0);
              acc[1][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bfrag[p], acc[1][p], 0, 0, 0);
            }
          }
          a_cur[0] = a_nxt[0];
          a_cur[1] = a_nxt[1];
        }
      }
    }
  }

  // ---- epilogue: y = act(acc*scale[co] + shift[co]) ; D layout: col = lane&31 (f),
  //      row = (r&3) + 8*(r>>2) + 4*(lane>>5) (co within the 32-block) -------------------
  if (!wave_active) return;
  const int f = f0 + l31;
  float* out_b = out + (size_t)b * kCo * plane;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float sc = scale[co], sh = shift[co];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int i = i0 + wave * P + p;
        if (i < n_c && f < F) {
          const int t = cls + dil * i;
          out_b[(size_t)co * plane + (size_t)t * F + f] = vs_act<ACT>(fmaf(acc[cb][p][r], sc, sh));
        }
      }
    }
  }
}

int conv_lds_pad() {   // experiment: extra dynamic LDS per workgroup to lower occupancy
  static int v = -1;
  if (v < 0) { const char* e = getenv("VS_CONV_LDS_PAD"); v = e ? atoi(e) : 0; }
  return v;
}

int conv_variant() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("VS_CONV_VARIANT"); v = e ? atoi(e) : -1; if (v < 0) v = 100; }   // 100 = per-shape default (see launch_conv)
  return v;
}

template <int KT, int KF, int P, int VAR, int NW>
int launch_var(const float* in, const float* wp, const float* scale, const float* shift, float* out,
               int B, int T, int F, int dil, int act, hipStream_t stream) {
  constexpr int R = NW * P;
  const int rows_max = (T + dil - 1) / dil;
  const int n_rt = (rows_max + R - 1) / R;
  const int n_ft = (F + kTileF - 1) / kTileF;
  const long long nblk = (long long)B * dil * n_rt * n_ft;
  VS_REQUIRE(nblk > 0 && nblk < 2147483647LL, "conv64: grid of %lld blocks out of range", nblk);
  dim3 grid((unsigned)nblk), block(NW * 64);
  switch (act) {
    case VS_ACT_RELU:
      hipLaunchKernelGGL((conv64_mfma_kernel<KT, KF, P, VS_ACT_RELU, VAR, NW>), grid, block, conv_lds_pad(), stream, in, wp, scale, shift, out, T, F, dil, n_rt, n_ft);
      break;
    case VS_ACT_MISH:
      hipLaunchKernelGGL((conv64_mfma_kernel<KT, KF, P, VS_ACT_MISH, VAR, NW>), grid, block, conv_lds_pad(), stream, in, wp, scale, shift, out, T, F, dil, n_rt, n_ft);
      break;
    case VS_ACT_NONE:
      hipLaunchKernelGGL((conv64_mfma_kernel<KT, KF, P, VS_ACT_NONE, VAR, NW>), grid, block, conv_lds_pad(), stream, in, wp, scale, shift, out, T, F, dil, n_rt, n_ft);
      break;
    default:
      VS_REQUIRE(false, "conv64: unknown activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

// VS_CONV_VARIANT (experiments): 0 plain, 1 pinned weight prefetch, 2/3 hand-pipelined 4-wave
// tiles, 4 = wave-private tiles (one wave per workgroup, no barriers).
template <int KT, int KF>
int launch_conv(const float* in, const float* wp, const float* scale, const float* shift, float* out,
                int B, int T, int F, int dil, int act, bool p2, hipStream_t stream) {
#define VS_GO(P_, V_, NW_) return launch_var<KT, KF, P_, V_, NW_>(in, wp, scale, shift, out, B, T, F, dil, act, stream)
  switch (conv_variant()) {
    case 10: VS_GO(2, 10, 4);
    case 11: VS_GO(2, 11, 4);
    case 12: VS_GO(2, 12, 4);
    case 13: VS_GO(2, 13, 4);
    case 14: VS_GO(2, 14, 4);
    case 17: VS_GO(2, 17, 4);
    case 5: if (p2) VS_GO(2, 5, 4); else VS_GO(1, 5, 4);
    case 4: VS_GO(2, 3, 1);
    case 3: if (p2) VS_GO(2, 3, 4); else VS_GO(1, 3, 4);
    case 2: if (p2) VS_GO(2, 2, 4); else VS_GO(1, 2, 4);
    case 0: if (p2) VS_GO(2, 0, 4); else VS_GO(1, 0, 4);
    case 1: if (p2) VS_GO(2, 1, 4); else VS_GO(1, 1, 4);
    // default (measured, profiles/r01_conv_micro_variants.jsonl): 8-row tiles run fastest with the
    // pinned one-tap weight prefetch, 4-row tiles (twice the weight traffic per MFMA) with the
    // hand-pipelined loop and its 3-tap-deep weight ring.
    default: if (p2) VS_GO(2, 1, 4); else VS_GO(1, 3, 4);
  }
#undef VS_GO
}

// rows wasted by tiling each residue class with R-row tiles
long long tile_rows(int T, int dil, int R) {
  long long tot = 0;
  for (int c = 0; c < dil && c < T; ++c) {
    int n = (T - c + dil - 1) / dil;
    tot += (long long)((n + R - 1) / R) * R;
  }
  return tot;
}

}  // namespace

extern "C" size_t vs_conv64_packed_floats(int KT, int KF) { return (size_t)(kNChunk * KT * KF + kPadTaps) * 512; }

int vs_conv64_pack_impl(const float* w, float* wp, int KT, int KF, hipStream_t stream) {
  VS_REQUIRE((KT == 7 && KF == 1) || (KT == 5 && KF == 5), "conv64: unsupported kernel %dx%d", KT, KF);
  const int total = (int)vs_conv64_packed_floats(KT, KF);
  hipLaunchKernelGGL(conv_pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, wp, KT, KF);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_conv64_fwd_impl(const float* in, const float* wp, const float* scale, const float* shift, float* out,
                       int B, int T, int F, int KT, int KF, int dil, int act, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "conv64: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((long long)kChunk * T * F * 4 < (long long)kOob, "conv64: T*F=%lld too large for 32-bit slab offsets", (long long)T * F);
  // pick the row-tile height (8 or 4 rows per residue class) that wastes fewer rows
  const bool p2 = tile_rows(T, dil, 8) <= tile_rows(T, dil, 4);
  if (KT == 7 && KF == 1) return launch_conv<7, 1>(in, wp, scale, shift, out, B, T, F, dil, act, p2, stream);
  if (KT == 5 && KF == 5) return launch_conv<5, 5>(in, wp, scale, shift, out, B, T, F, dil, act, p2, stream);
  VS_REQUIRE(false, "conv64: unsupported kernel %dx%d", KT, KF);
  return -1;
}
