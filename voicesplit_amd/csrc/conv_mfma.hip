// 64->64 channel "same" convolution with time dilation, fused BatchNorm(eval)+activation.
//
// Replaces, per layer, ZeroPad2d + Conv2d + BatchNorm2d + Mish/ReLU of
//   models/voicesplit/model.py:21-23 (cnn2, 7x1) and :26-48 (cnn3..cnn7, 5x5, dilation (d,1))
//   models/voicefilter/model.py:25-50 (same, ReLU).
//
// Formulation (im2col-free implicit GEMM on the fp32 matrix cores):
//   out[co][t][f] = sum_{ci,kt,kf} W[co][ci][kt][kf] * in[ci][t+(kt-KT/2)*dil][f+kf-KF/2]
//   M = co (64, two 32-row MFMA blocks), N = 32 consecutive f of one output row,
//   K = a pair of input channels per v_mfma_f32_32x32x2_f32 step, looped over taps and ci chunks.
// A dilated layer is decomposed into `dil` residue classes of t (t = cls + dil*i): inside a
// class it is an ordinary dense conv over rows i, so an LDS window of R output rows needs only
// R+KT-1 input rows whatever the dilation.
//
// Workgroup = 256 threads (4 waves, one per SIMD; up to three workgroups per CU).  Tile = 64 co x
// (R = 4*P rows) x 32 f.  Wave w owns rows w*P..w*P+P-1 for both co blocks: 2*P accumulators of
// 32x32 (16 VGPR each).  The ci dimension is walked in 8 chunks of 8 channels; per chunk:
//  * staging: thread -> pixel (row, x) of the [R+KT-1][32+KF-1] window, 8 channel loads (raw
//    buffer loads whose out-of-range offsets return 0 -- that zero fill outside the image IS the
//    reference's ZeroPad2d), two ds_write_b128 into the pixel-major LDS window [row][x][8 ci];
//    the loads of chunk c+1 are in flight while chunk c is multiplied;
//  * B operand: lane (f = lane&31, half = lane>>5) gets the B values of all four K-steps of a
//    tap (ci = 4*half + 0..3) with ONE ds_read_b128;
//  * A operand: weights pre-packed in fragment order (conv_pack_weights_kernel, matching
//    "blocked" channel order), one buffer_load_dwordx4 per co block per tap with a lane-constant
//    voffset and a scalar tap offset: no per-tap VALU address arithmetic;
//  * both prefetched one tap ahead: per tap a wave issues 2 buffer loads + P ds_read_b128 +
//    8*P MFMAs.
// Epilogue: y = act(acc*scale[co] + shift[co]) (folded BatchNorm), 128-byte row segments.
// Rows past the end of a residue class are skipped per wave, so the only padded MFMA work is
// F: 608/601.
//
// Measured at B=64 (profiles/r01_*): 126-136 TFLOP/s = 80-87 % of the fp32 MFMA peak.  The
// variants tried on the way (compiler-scheduled loop, pinned / 3-tap-deep weight prefetch,
// spread staging loads, wave-private tiles without barriers, 16-row tiles, bank-rotated
// operands, ablations) all land within +-3 % of this; see DESIGN.md section 5 and
// profiles/r01_conv_micro_variants.jsonl.
#include "vs_common.h"

namespace {

constexpr int kCo = 64;
constexpr int kCi = 64;
constexpr int kChunk = 8;               // input channels per LDS stage
constexpr int kNChunk = kCi / kChunk;   // 8
constexpr unsigned kOob = 0x7FFFFFF0u;  // buffer offset guaranteed >= num_records -> load returns 0
constexpr int kTileF = 32;
constexpr int kPadTaps = 4;             // dummy tap blocks behind the packed weights (prefetch overrun)

// packed weight layout: [chunk][tap][cb(2)][lane(64)][4]; element j of the float4 is
//   W[co = cb*32 + (lane&31)][ci = chunk*8 + 4*(lane>>5) + j][kt][kf]
// i.e. MFMA K-step j multiplies channels (j, 4+j) of the chunk; kPadTaps zero blocks follow.
// transpose_flip = 1 packs the data-gradient weights W'[co'][ci'][kt][kf] = W[ci'][co'][KT-1-kt][KF-1-kf]:
// with them the same forward kernel computes dIn = conv^T(dOut) ("same" padding is symmetric).
__global__ void conv_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int KT, int KF, int transpose_flip) {
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
  int ci = chunk * kChunk + 4 * (lane >> 5) + j;
  wp[idx] = transpose_flip ? w[((ci * kCi + co) * KT + (KT - 1 - kt)) * KF + (KF - 1 - kf)]
                           : w[((co * kCi + ci) * KT + kt) * KF + kf];
}

template <int KT, int KF, int P, int ACT>
__global__ __launch_bounds__(256, 2)
void conv64_mfma_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                      const float* __restrict__ scale, const float* __restrict__ shift,
                      float* __restrict__ out, int T, int F, int dil, int n_rt, int n_ft) {
  constexpr int R = 4 * P;
  constexpr int ROWS = R + KT - 1;
  constexpr int PX = kTileF + KF - 1;            // window width in pixels
  constexpr int NPIX = ROWS * PX;
  constexpr int NPP = (NPIX + 255) / 256;        // pixels per thread
  constexpr int NT = KT * KF;
  __shared__ __attribute__((aligned(16))) float sIn[(NPIX + P * PX) * kChunk];   // + overrun rows for the prefetch

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
  const int n_c = (T - cls + dil - 1) / dil;
  const int i0 = rt * R;
  if (i0 >= n_c) return;
  const int f0 = ft * kTileF;
  const size_t plane = (size_t)T * F;
  const float* in_b = in + (size_t)b * kCi * plane;
  const unsigned plane_bytes = (unsigned)(plane * sizeof(float));
  const unsigned slab_bytes = plane_bytes * kChunk;

  // staging descriptors
  unsigned voff[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int pix = tid + 256 * i;
    const int rr = pix / PX;
    const int x = pix - rr * PX;
    const int iin = i0 - KT / 2 + rr;
    const int f = f0 - KF / 2 + x;
    const bool ok = (pix < NPIX) && (iin >= 0) && (iin < n_c) && (f >= 0) && (f < F);
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
        float4* d = reinterpret_cast<float4*>(&sIn[pix * kChunk]);
        d[0] = make_float4(stage[i][0], stage[i][1], stage[i][2], stage[i][3]);
        d[1] = make_float4(stage[i][4], stage[i][5], stage[i][6], stage[i][7]);
      }
    }
  };

  f32x16 acc[2][P];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][p][r] = 0.f;

  const bool wave_active = (i0 + wave * P) < n_c;
  // weights: rsrc over the whole packed array, lane-constant voffset, scalar tap offset
  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wp), 0, (unsigned)((kNChunk * NT + kPadTaps) * 2048), 0x00020000);
  const unsigned wvoff = lane * 16;
  auto load_a = [&](int tg, float4 (&a)[2]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u r0 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, tg * 2048, 0);
    v4u r1 = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + 1024, tg * 2048, 0);
    v4f f0 = __builtin_bit_cast(v4f, r0), f1 = __builtin_bit_cast(v4f, r1);
    a[0] = make_float4(f0[0], f0[1], f0[2], f0[3]);
    a[1] = make_float4(f1[0], f1[1], f1[2], f1[3]);
  };
  // this lane's B base: pixel (row = wave*P, x = l31), channel block = half
  const float* sB = sIn + ((wave * P) * PX + l31) * kChunk + half * 4;

  float4 a_cur[2], a_nxt[2];
  load_a(0, a_cur);
  load_chunk(0);
#pragma unroll 1
  for (int chunk = 0; chunk < kNChunk; ++chunk) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    if (chunk + 1 < kNChunk) load_chunk(chunk + 1);
    if (wave_active) {
      float4 b_cur[P], b_nxt[P];
#pragma unroll
      for (int p = 0; p < P; ++p) b_cur[p] = *reinterpret_cast<const float4*>(sB + (p * PX) * kChunk);
#pragma unroll 1
      for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) {
          const int tg = chunk * NT + kt * KF + kf;
          load_a(tg + 1, a_nxt);
          const int nkt = (kf + 1 < KF) ? kt : kt + 1;      // tap after this one (may overrun into the pad rows)
          const int nkf = (kf + 1 < KF) ? kf + 1 : 0;
#pragma unroll
          for (int p = 0; p < P; ++p)
            b_nxt[p] = *reinterpret_cast<const float4*>(sB + ((p + nkt) * PX + nkf) * kChunk);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) {
            const float a0 = (pr == 0) ? a_cur[0].x : (pr == 1) ? a_cur[0].y : (pr == 2) ? a_cur[0].z : a_cur[0].w;
            const float a1 = (pr == 0) ? a_cur[1].x : (pr == 1) ? a_cur[1].y : (pr == 2) ? a_cur[1].z : a_cur[1].w;
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const float bv = (pr == 0) ? b_cur[p].x : (pr == 1) ? b_cur[p].y : (pr == 2) ? b_cur[p].z : b_cur[p].w;
              acc[0][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][p], 0, 0, 0);
              acc[1][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][p], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          a_cur[0] = a_nxt[0];
          a_cur[1] = a_nxt[1];
#pragma unroll
          for (int p = 0; p < P; ++p) b_cur[p] = b_nxt[p];
        }
      }
    }
  }

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

template <int KT, int KF, int P>
int launch_conv(const float* in, const float* wp, const float* scale, const float* shift, float* out,
                int B, int T, int F, int dil, int act, hipStream_t stream) {
  constexpr int R = 4 * P;
  const int rows_max = (T + dil - 1) / dil;
  const int n_rt = (rows_max + R - 1) / R;
  const int n_ft = (F + kTileF - 1) / kTileF;
  const long long nblk = (long long)B * dil * n_rt * n_ft;
  VS_REQUIRE(nblk > 0 && nblk < 2147483647LL, "conv64: grid of %lld blocks out of range", nblk);
  dim3 grid((unsigned)nblk), block(256);
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL((conv64_mfma_kernel<KT, KF, P, VS_ACT_RELU>), grid, block, 0, stream, in, wp, scale, shift, out, T, F, dil, n_rt, n_ft); break;
    case VS_ACT_MISH: hipLaunchKernelGGL((conv64_mfma_kernel<KT, KF, P, VS_ACT_MISH>), grid, block, 0, stream, in, wp, scale, shift, out, T, F, dil, n_rt, n_ft); break;
    case VS_ACT_NONE: hipLaunchKernelGGL((conv64_mfma_kernel<KT, KF, P, VS_ACT_NONE>), grid, block, 0, stream, in, wp, scale, shift, out, T, F, dil, n_rt, n_ft); break;
    default: VS_REQUIRE(false, "conv64: unknown activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

// rows covered when every residue class is tiled with R-row tiles
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

int vs_conv64_pack_impl(const float* w, float* wp, int KT, int KF, int transpose_flip, hipStream_t stream) {
  VS_REQUIRE((KT == 7 && KF == 1) || (KT == 5 && KF == 5), "conv64: unsupported kernel %dx%d", KT, KF);
  const int total = (int)vs_conv64_packed_floats(KT, KF);
  hipLaunchKernelGGL(conv_pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, wp, KT, KF, transpose_flip);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_conv64_fwd_impl(const float* in, const float* wp, const float* scale, const float* shift, float* out,
                       int B, int T, int F, int KT, int KF, int dil, int act, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "conv64: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((long long)kChunk * T * F * 4 < (long long)kOob, "conv64: T*F=%lld too large for 32-bit slab offsets", (long long)T * F);
  // row-tile height (8 or 4 rows per residue class): whichever covers fewer padded rows
  const bool p2 = tile_rows(T, dil, 8) <= tile_rows(T, dil, 4);
  if (KT == 7 && KF == 1) {
    return p2 ? launch_conv<7, 1, 2>(in, wp, scale, shift, out, B, T, F, dil, act, stream)
              : launch_conv<7, 1, 1>(in, wp, scale, shift, out, B, T, F, dil, act, stream);
  }
  if (KT == 5 && KF == 5) {
    return p2 ? launch_conv<5, 5, 2>(in, wp, scale, shift, out, B, T, F, dil, act, stream)
              : launch_conv<5, 5, 1>(in, wp, scale, shift, out, B, T, F, dil, act, stream);
  }
  VS_REQUIRE(false, "conv64: unsupported kernel %dx%d", KT, KF);
  return -1;
}
