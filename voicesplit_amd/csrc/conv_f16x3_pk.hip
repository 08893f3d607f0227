// Persistent, software-pipelined form of the 5x5 64->64 split-f16 convolution (conv_f16x3.hip has
// the arithmetic, the packed-weight layout and the one-tile-per-workgroup kernel this one replaces
// for cnn3..cnn7 forward + data gradient: models/voicesplit/model.py:26-48).
//
// What rocprofv3 showed for the one-tile kernel (profiles/r01_train_rocprof): matrix pipe 60 % busy,
// waves parked 25 % of their cycles -- not on memory (every s_waitcnt in front of its barriers is
// already satisfied) but on (a) the fill/drain of a workgroup that lives for ONE tile: address
// set-up + first HBM round trip at the head, 64 scale/Mish/store sequences per lane at the tail,
// with nothing but the SIMD's second wave to cover them, (b) two rendezvous per weight group where
// either of the two waves of a SIMD can strand the pipe, (c) the chunk boundary, where all four
// waves convert and write the next window while no MFMA of the workgroup is in flight.
//
// This kernel removes all three by construction:
//  * ONE workgroup per CU (4 waves, one per SIMD, 110 KB of LDS), persistent: it walks
//    tiles  k*gridDim + (xcd*gridDim/8 + j)  so the 32 workgroups of an XCD sweep a contiguous band
//    of tiles (shared halo rows/columns are L2 hits) and never pays a prologue again;
//  * the (tile, ci chunk) steps form one flat pipeline: the window of step s+1 is loaded at the
//    start of step s, converted and written into the OTHER LDS window buffer in the shadow of step
//    s's MFMAs (a few VALU per tap), the weights of group g+1 are fetched during group g, and the
//    epilogue of tile k (scale, shift, activation, stores, |max|) is spread one accumulator element
//    per tap over the first 16 taps of each chunk of tile k+1;
//  * therefore ONE barrier per weight group (60 MFMAs per wave) is the only synchronisation, the
//    group body is branch-free straight-line code (non-existent next steps / out-of-tile stores are
//    predicated through the buffer descriptor's range check), and a wave's instruction stream is
//    MFMA-dense: 12 MFMAs + 8 ds_read_b128 + <= ~35 staging/epilogue instructions per tap.
// Results are bit-identical to conv64_f16x3_kernel (same K order, same fp32 accumulation).
#include "vs_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCo = 64;
constexpr int kCi = 64;
constexpr int kChunk = 16;
constexpr int kNChunk = 4;
constexpr unsigned kOob = 0x7FFFFFF0u;
constexpr int kTileF = 32;
constexpr int KT = 5, KF = 5;
constexpr int kTapBytes = 2 * 2 * 64 * 16;     // one (chunk, tap): 2 co blocks x 2 parts x 64 lanes x 16 B
constexpr int kTapVec = kTapBytes / 16;        // 256
constexpr int kGroupVec = KF * kTapVec;        // one kt row of taps: 1280 x 16 B = 20 KB
constexpr int kNGroups = kNChunk * KT;         // 20 weight groups per tile

struct PkArgs {
  const float* in;
  const _Float16* wp;
  const float* scale;
  const float* shift;
  const float* in_scale;
  const float* w_scale;
  float* out;
  unsigned* amax_out;
  double* bn_stats;       // [VS_BN_STAT_SLOTS][64][2] partial {sum z, sum z^2} accumulated by the STATS instances (train-mode BatchNorm), else NULL
  int B, T, F, dil, n_rt, n_ft, i_base, i_end;
  int rows_q, rows_r;     // T / dil, T % dil: residue class c has rows_q + (c < rows_r) rows
  long long n_tiles;
};

struct Tile {
  int b, cls, i0, f0, n_all, n_c;
  bool valid;
};

__device__ __forceinline__ void split2(float x0, float x1, f16x2& hi, f16x2& lo) {
  hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0 - (float)hi[0], x1 - (float)hi[1]));
}

// Workgroup rendezvous that orders LDS traffic only: the pending ds_writes of this wave are retired
// (lgkmcnt) and nothing else is waited for -- global loads issued for later steps and the epilogue's
// stores stay in flight across the barrier (a __syncthreads() would be free to drain them).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Position of a workgroup in the tile space as mixed-radix digits (ft, rt, cls, b) and the digits of
// its stride: stepping to the next tile is four adds with carry instead of three integer divisions.
struct TileWalk {
  int ft, rt, cls, b;
  int d_ft, d_rt, d_cls, d_b;
};

__device__ __forceinline__ void walk_init(TileWalk& w, const PkArgs& a, unsigned first, unsigned stride) {
  unsigned id = first;
  w.ft = id % a.n_ft; id /= a.n_ft;
  w.rt = id % a.n_rt; id /= a.n_rt;
  w.cls = id % a.dil;
  w.b = id / a.dil;
  id = stride;
  w.d_ft = id % a.n_ft; id /= a.n_ft;
  w.d_rt = id % a.n_rt; id /= a.n_rt;
  w.d_cls = id % a.dil;
  w.d_b = id / a.dil;
}

__device__ __forceinline__ void walk_next(TileWalk& w, const PkArgs& a) {
  w.ft += w.d_ft;
  int c = w.ft >= a.n_ft;
  w.ft -= c ? a.n_ft : 0;
  w.rt += w.d_rt + c;
  c = w.rt >= a.n_rt;
  w.rt -= c ? a.n_rt : 0;
  w.cls += w.d_cls + c;
  c = w.cls >= a.dil;
  w.cls -= c ? a.dil : 0;
  w.b += w.d_b + c;
}

template <int P>
__device__ __forceinline__ Tile tile_at(const TileWalk& w, const PkArgs& a) {
  constexpr int R = 4 * P;
  Tile t;
  t.b = w.b;
  t.cls = w.cls;
  t.n_all = a.rows_q + (w.cls < a.rows_r ? 1 : 0);
  t.n_c = t.n_all < a.i_end ? t.n_all : a.i_end;
  t.i0 = a.i_base + w.rt * R;
  t.f0 = w.ft * kTileF;
  t.valid = w.b < a.B && t.i0 < t.n_c;
  return t;
}

// first valid tile at or after the walk's position (b >= B: past the end)
template <int P>
__device__ __forceinline__ Tile seek_tile(TileWalk& w, const PkArgs& a) {
  Tile t = tile_at<P>(w, a);
  while (!t.valid && w.b < a.B) { walk_next(w, a); t = tile_at<P>(w, a); }
  return t;
}

// ABL: timing ablations (results are garbage unless 0): 1 = no fragment reads, 2 = no staging (window and
// weight loads / conversions / LDS writes), 4 = no in-loop epilogue, 8 = no barriers.
// FILL > 0: pin the issue pattern of a tap to (1 MFMA, up to FILL other instructions) x 12.
// STATS: the epilogue also accumulates the per-channel sum and sum of squares of what it stores (train-mode
// BatchNorm statistics of the layer, SURVEY.md 7 hard-part 3): per-lane fp32 partial sums over all tiles of the
// workgroup, combined across lanes / waves once at the end, one fp64 atomicAdd per channel and workgroup.
// NT: product terms per fp32 product: 3 = split f16 (hi*hi + hi*lo + lo*hi), 1 = single-pass bf16 (VS_MATH_BF16:
// operands rounded to bf16, the lo slots of the LDS images stay unused).
template <int P, int ACT, int ABL = 0, int FILL = 0, int NT = 3, bool STATS = false>
__global__ __launch_bounds__(256, 1)
void conv64_f16x3_pk_kernel(PkArgs a) {
  constexpr int R = 4 * P;
  constexpr int ROWS = R + KT - 1;
  constexpr int PX = 36;                               // 32 + 4 halo columns (already a multiple of 4)
  constexpr int NPIX = ROWS * PX;
  constexpr int NPP = (NPIX + 255) / 256;
  constexpr int WINVEC = (NPIX + (P + 1) * PX) * 4;    // u32x4 per window buffer (with the overrun rows)
  constexpr int DUMMY = NPIX;                          // first overrun pixel: sink for threads without a pixel
  constexpr int NSL = 2 * P;                           // accumulator slices (co block, row) per wave
  constexpr int SPC = (NSL + kNChunk - 1) / kNChunk;   // slices retired per chunk of the next tile
  // ONE LDS object: two window buffers, a ring of three weight groups, the epilogue constants
  __shared__ __attribute__((aligned(16))) u32x4 smem[2 * WINVEC + 3 * kGroupVec + 32];
  u32x4* const sIn0 = smem;
  u32x4* const sW0 = smem + 2 * WINVEC;
  float* const sSc = reinterpret_cast<float*>(smem + 2 * WINVEC + 3 * kGroupVec);   // [64] scale*inv, [64] shift

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  const int half = lane >> 5;
  const int T = a.T, F = a.F, dil = a.dil;
  const size_t plane = (size_t)T * F;
  const unsigned plane_bytes = (unsigned)(plane * sizeof(float));
  const unsigned slab_bytes = plane_bytes * kChunk;
  const unsigned batch_bytes_lo = plane_bytes * kCo;   // fits: checked by the launcher
  const float s_in = a.in_scale[0];
  const float inv = a.in_scale[1] * a.w_scale[1];

  // XCD-contiguous tile walk (block b runs on XCD b % 8: a speed assumption only)
  const unsigned nwg = gridDim.x;
  const unsigned per_xcd = nwg / 8;
  const unsigned first = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;

  // per-thread window pixels (compile-time geometry): row / column inside the window
  int prr[NPP], pxx[NPP], pdst[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int pix = tid + 256 * i;
    prr[i] = pix / PX;
    pxx[i] = pix - prr[i] * PX;
    const bool has = pix < NPIX;
    const int sw = (pxx[i] >> 2) & 3;
    pdst[i] = has ? pix * 4 : DUMMY * 4;               // u32x4 index of the pixel's 4 slots
    pdst[i] |= has ? (sw << 28) : 0;                   // the swizzle rides in the top bits
  }
  unsigned voff[NPP];
  auto set_voff = [&](const Tile& t) {
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int iin = t.i0 - KT / 2 + prr[i];
      const int f = t.f0 - KF / 2 + pxx[i];
      const bool ok = (tid + 256 * i < NPIX) && (iin >= 0) && (iin < t.n_all) && (f >= 0) && (f < F);
      voff[i] = ok ? (unsigned)(((t.cls + dil * iin) * F + f) * 4) : kOob;
    }
  };
  float stage[NPP][kChunk];
  // exists == false: a zero-sized descriptor, every load returns 0 without touching memory
  auto load_chunk = [&](const Tile& t, int chunk, bool exists) {
    const float* src = a.in + ((size_t)t.b * kCi + (size_t)chunk * kChunk) * plane;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, exists ? slab_bytes : 0u, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPP; ++i)
#pragma unroll
      for (int c = 0; c < kChunk; ++c)
        stage[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[i], c * plane_bytes, 0));
  };
  // one unit = 8 channels (one 16-byte hi slot + one lo slot) of staged pixel i
  auto store_unit = [&](u32x4* win, int i, int h) {
    const int sw = (unsigned)pdst[i] >> 28;
    u32x4* dst = win + (pdst[i] & 0x0fffffff);
    asm volatile("" : "+v"(stage[i][8 * h]));     // pins the unit's conversion to the tap it was placed in
    u32x4 vh, vl;
    if (NT == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) vh[q] = vs_pack_bf16(stage[i][8 * h + 2 * q] * s_in, stage[i][8 * h + 2 * q + 1] * s_in);
      dst[(0 + h) ^ sw] = vh;
      return;
    }
    f16x2 hi[4], lo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split2(stage[i][8 * h + 2 * q] * s_in, stage[i][8 * h + 2 * q + 1] * s_in, hi[q], lo[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vh[q] = __builtin_bit_cast(unsigned, hi[q]);
      vl[q] = __builtin_bit_cast(unsigned, lo[q]);
    }
    dst[(0 + h) ^ sw] = vh;
    dst[(2 + h) ^ sw] = vl;
  };

  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(a.wp), 0, (unsigned)(kNGroups * KF * kTapBytes), 0x00020000);
  u32x4 wreg[KF];
  auto load_w = [&](int gg) {
#pragma unroll
    for (int i = 0; i < KF; ++i)
      wreg[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)((tid + 256 * i) * 16), gg * (KF * kTapBytes), 0);
  };
  auto store_w = [&](u32x4* buf) {
#pragma unroll
    for (int i = 0; i < KF; ++i) buf[tid + 256 * i] = wreg[i];
  };

  f32x16 acc[NSL], accE[NSL];          // index = cb * P + p
#pragma unroll
  for (int q = 0; q < NSL; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[q][r] = 0.f; accE[q][r] = 0.f; }

  // lane-dependent part of the (swizzled) B-fragment address per (kf, part), relative to a window buffer
  int boff[KF][2];
#pragma unroll
  for (int kf = 0; kf < KF; ++kf) {
    const int x = l31 + kf;
    const int sw = (x >> 2) & 3;
    boff[kf][0] = ((wave * P) * PX + x) * 4 + ((0 + half) ^ sw);
    boff[kf][1] = ((wave * P) * PX + x) * 4 + ((2 + half) ^ sw);
  }

  // ---- epilogue state of the tile whose accumulators sit in accE -----------------------------------
  // (no global load may be consumed inside the pipeline: loads return in order, so a scale[co] fetch
  // used at once would wait for every window load issued before it -- the constants live in LDS)
  if (tid < 64) { sSc[tid] = a.scale[tid] * inv; sSc[64 + tid] = a.shift[tid]; }
  __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0u, 0x00020000);   // empty: stores dropped
  int e_i0 = 0, e_f0 = 0, e_cls = 0, e_nc = 0;
  float e_max = 0.f;
  unsigned e_voff[SPC];            // byte offset of (co = cb*32 + 4*half, t, f) of the slice being retired, or kOob
  const float* e_tab[SPC];         // sSc + cb*32 + 4*half
  bool e_ok[SPC];
  auto epilogue_slice = [&](int s, int q_rt /* runtime slice id = cb*P + p */) {
    const int cb = q_rt / P, p = q_rt - cb * P;
    const int i = e_i0 + wave * P + p;
    const int f = e_f0 + l31;
    e_ok[s] = (i < e_nc) && (f < F);
    e_voff[s] = e_ok[s] ? (unsigned)((((cb * 32 + 4 * half) * T + (e_cls + dil * i)) * F + f) * 4) : kOob;
    e_tab[s] = sSc + cb * 32 + 4 * half;
  };
  // element r of the slice: out[b][co][t][f] = act(accE * scale[co]*inv + shift[co]), co = cb*32 + cr + 4*half
  // STATS: stS/stQ[0] belong to the co block whose slices are being retired, [1] to the other one; the two
  // swap when the retiring slices move to the other co block (static register indices throughout)
  float stS[2][16], stQ[2][16];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) { stS[k][r] = 0.f; stQ[k][r] = 0.f; }
  auto epilogue_elem = [&](int s, const f32x16& v, int r, float sc, float sh, int slot = 0) {
    const int cr = (r & 3) + 8 * (r >> 2);
    float vr = v[r];
    asm volatile("" : "+v"(vr));        // opaque here: the element's arithmetic cannot be hoisted out of its tap
    const float y = vs_act_fast<ACT>(fmaf(vr, sc, sh));
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), orsrc, e_voff[s], cr * plane_bytes, 0);
    e_max = e_ok[s] ? fmaxf(e_max, fabsf(y)) : e_max;
    if (STATS) {
      const float yz = e_ok[s] ? y : 0.f;
      stS[slot][r] += yz;
      stQ[slot][r] = fmaf(yz, yz, stQ[slot][r]);
    }
  };

  TileWalk walk;
  walk_init(walk, a, first, nwg);
  Tile cur = seek_tile<P>(walk, a);
  if (!cur.valid) return;

  // fragments of the tap being multiplied / of the tap after it (carried across groups, chunks and tiles)
  u32x4 a_cur[2][2], a_nxt[2][2], b_cur[P][2], b_nxt[P][2];
  auto load_a = [&](const u32x4* wb, int g, u32x4 (&af)[2][2]) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int part = 0; part < (NT == 1 ? 1 : 2); ++part) af[cb][part] = wb[g * kTapVec + (cb * 2 + part) * 64 + lane];
  };
  auto load_b = [&](const u32x4* wn, int kt, int kf, u32x4 (&bf)[P][2]) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      bf[p][0] = wn[boff[kf][0] + (p + kt) * PX * 4];
      if (NT != 1) bf[p][1] = wn[boff[kf][1] + (p + kt) * PX * 4];
    }
  };

  // ---- pipeline head: window (cur, chunk 0), weight groups 0 and 1 ---------------------------------
  set_voff(cur);
  load_w(0);
  load_chunk(cur, 0, true);
  store_w(sW0);
  load_w(1);
#pragma unroll
  for (int i = 0; i < NPP; ++i) { store_unit(sIn0, i, 0); store_unit(sIn0, i, 1); }
  store_w(sW0 + kGroupVec);
  lds_barrier();
  load_a(sW0, 0, a_cur);
  load_b(sIn0, 0, 0, b_cur);
  int step = 0;        // parity of the window buffer holding the current step
  int wr = 0;          // ring slot (0..2) of the weight group being multiplied

  while (true) {
    walk_next(walk, a);
    Tile nxt = seek_tile<P>(walk, a);

#pragma unroll 1
    for (int chunk = 0; chunk < kNChunk; ++chunk) {
      const bool last_chunk = chunk == kNChunk - 1;
      const bool has_next = !last_chunk || nxt.valid;
      const u32x4* const win = sIn0 + (step & 1) * WINVEC;
      u32x4* const win_n = sIn0 + ((step + 1) & 1) * WINVEC;
      // the slices of the previous tile retired during this chunk, their constants in registers
      float e_sc[SPC][16], e_sh[SPC][16];
#pragma unroll
      for (int s = 0; s < SPC; ++s) {
        epilogue_slice(s, chunk * SPC + s);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cr = (r & 3) + 8 * (r >> 2);
          e_sc[s][r] = e_tab[s][cr];
          e_sh[s][r] = e_tab[s][64 + cr];
        }
      }
#pragma unroll
      for (int grp = 0; grp < KT; ++grp) {
        const int wr1 = wr == 2 ? 0 : wr + 1;
        const int wr2 = wr1 == 2 ? 0 : wr1 + 1;
        const u32x4* const wbuf = sW0 + wr * kGroupVec;         // this group (visible since the previous barrier)
        const u32x4* const wbuf1 = sW0 + wr1 * kGroupVec;       // next group (written last group, visible after this barrier)
        u32x4* const wbuf2 = sW0 + wr2 * kGroupVec;             // group after next: filled during this group
        // One rendezvous per group.  It publishes what was written during the previous group (weights
        // of group+1; the next window once grp == 4) and frees what the previous group read (ring slot
        // wr2, and at grp == 0 the other window buffer).  This group's own operands were published one
        // barrier earlier, so its first fragments were fetched before the barrier.
        if (!(ABL & 8)) lds_barrier();
        if (!(ABL & 2)) {
          int ng = chunk * KT + grp + 2;
          ng = ng >= kNGroups ? ng - kNGroups : ng;
          load_w(ng);
        }
        if (grp == 0 && !(ABL & 2)) {
          if (last_chunk) { if (nxt.valid) set_voff(nxt); }
          load_chunk(last_chunk ? nxt : cur, last_chunk ? 0 : chunk + 1, has_next);
        }
#pragma unroll
        for (int g = 0; g < KF; ++g) {
          const int ti = grp * KF + g;                           // tap index inside the chunk, compile-time
          // fragments of the next tap: same group / next group / next step
          if (ABL & 1) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) { a_nxt[cb][0] = a_cur[cb][1]; a_nxt[cb][1] = a_cur[cb][0]; }
#pragma unroll
            for (int p = 0; p < P; ++p) { b_nxt[p][0] = b_cur[p][1]; b_nxt[p][1] = b_cur[p][0]; }
          } else if (g + 1 < KF) {
            load_a(wbuf, g + 1, a_nxt);
            load_b(win, grp, g + 1, b_nxt);
          } else {
            load_a(wbuf1, 0, a_nxt);
            if (grp + 1 < KT) load_b(win, grp + 1, 0, b_nxt);
            else load_b(win_n, 0, 0, b_nxt);
          }
          // --- shadow work of this tap ---------------------------------------------------------------
          // (1) one accumulator element of the previous tile per retiring slice
          if (ti < 16 && !(ABL & 4)) {
#pragma unroll
            for (int s = 0; s < SPC; ++s) epilogue_elem(s, accE[s], ti, e_sc[s][ti], e_sh[s][ti]);
          }
          // (2) the next step's window: 2*NPP units spread over the taps of groups 2 and 3
          if ((grp == 2 || grp == 3) && !(ABL & 2)) {
            const int slot = (grp - 2) * KF + g;                 // 0..9
            constexpr int NU = 2 * NPP;
#pragma unroll
            for (int u = 0; u < NU; ++u)
              if (slot == (u * 10) / NU) store_unit(win_n, u >> 1, u & 1);
          }
          // (3) the weights fetched at the head of this group go to the ring
          if (g == 3 && !(ABL & 2)) store_w(wbuf2);
          // --- the tap: term outermost so consecutive MFMAs hit different accumulators ---------------
          if (NT == 1) {
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb)
                acc[cb * P + p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vs_bf16x8, a_cur[cb][0]),
                                                                         __builtin_bit_cast(vs_bf16x8, b_cur[p][0]), acc[cb * P + p], 0, 0, 0);
          } else {
#pragma unroll
          for (int term = 0; term < 3; ++term) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const f16x8 bq = __builtin_bit_cast(f16x8, b_cur[p][term == 1 ? 1 : 0]);
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) {
                const f16x8 aq = __builtin_bit_cast(f16x8, a_cur[cb][term == 0 ? 1 : 0]);
                acc[cb * P + p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq, bq, acc[cb * P + p], 0, 0, 0);
              }
            }
          }
          }
          // issue pattern of the tap: one MFMA, then up to FILL instructions of any other kind -- the
          // matrix pipe is busy 32 cycles per MFMA, which covers ~7 issue slots of this (only) wave
          if (FILL > 0) {
#pragma unroll
            for (int m = 0; m < NT * 2 * P; ++m) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
              __builtin_amdgcn_sched_group_barrier(0x6f6, FILL, 0);   // VALU | SALU | VMEM | DS | TRANS
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
        wr = wr1;
      }
      ++step;
      // retire the slices handled in this chunk: accE[s] <- accE[s + SPC]
#pragma unroll
      for (int s = 0; s + SPC < NSL; ++s) accE[s] = accE[s + SPC];
      if (STATS) {
        // slices are retired in the order q = cb*P + p: the co block changes after chunk (P/SPC - 1) and after the last
        constexpr int CPB = P / SPC;                 // chunks per co block (2 for P = 2)
        if (chunk % CPB == CPB - 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float a0 = stS[0][r], q0 = stQ[0][r];
            stS[0][r] = stS[1][r]; stQ[0][r] = stQ[1][r];
            stS[1][r] = a0; stQ[1][r] = q0;
          }
        }
      }
    }

    // ---- tile done: its accumulators become the pending epilogue --------------------------------------
    vs_absmax_commit(e_max, a.amax_out);            // |max| of the tile retired during this one (0 the first time: no-op)
    e_max = 0.f;
#pragma unroll
    for (int q = 0; q < NSL; ++q) {
      accE[q] = acc[q];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    }
    e_i0 = cur.i0; e_f0 = cur.f0; e_cls = cur.cls; e_nc = cur.n_c;
    orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)cur.b * kCo * plane, 0, batch_bytes_lo, 0x00020000);
    if (!nxt.valid) break;
    cur = nxt;
  }

  // ---- drain: the last tile's epilogue has no next tile to hide behind ---------------------------------
#pragma unroll
  for (int q = 0; q < NSL; ++q) {
    epilogue_slice(0, q);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = (r & 3) + 8 * (r >> 2);
      epilogue_elem(0, accE[q], r, e_tab[0][cr], e_tab[0][64 + cr], q / P);     // after whole tiles slot k = co block k
    }
  }
  vs_absmax_commit(e_max, a.amax_out);
  if (STATS) {
    // lanes of one half-wave hold the same 32 channels: co = cb*32 + cr(r) + 4*half
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = stS[k][r], qv = stQ[k][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); }
        stS[k][r] = sv; stQ[k][r] = qv;
      }
    __syncthreads();                             // every wave is out of the pipeline: the LDS is free
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][64 co][2]
    if (l31 == 0) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = k * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          red[(wave * 64 + co) * 2 + 0] = stS[k][r];
          red[(wave * 64 + co) * 2 + 1] = stQ[k][r];
        }
    }
    __syncthreads();
    if (tid < 128) {
      const double v = (double)red[tid] + (double)red[128 + tid] + (double)red[256 + tid] + (double)red[384 + tid];
      atomicAdd(a.bn_stats + (size_t)(blockIdx.x % VS_BN_STAT_SLOTS) * 128 + tid, v);     // tid = co*2 + {0: sum, 1: sum of squares}
    }
  }
}

template <int P>
int launch_pk(const PkArgs& a0, int act, int i_base, int i_end, hipStream_t stream, int abl = 0, int math = VS_MATH_CODE_F16X3) {
  constexpr int R = 4 * P;
  PkArgs a = a0;
  a.i_base = i_base;
  a.i_end = i_end;
  const int rows_all = (a.T + a.dil - 1) / a.dil;
  const int rows_max = (rows_all < i_end ? rows_all : i_end) - i_base;
  if (rows_max <= 0) return 0;
  a.n_rt = (rows_max + R - 1) / R;
  a.n_ft = (a.F + kTileF - 1) / kTileF;
  a.n_tiles = (long long)a.B * a.dil * a.n_rt * a.n_ft;
  a.rows_q = a.T / a.dil;
  a.rows_r = a.T % a.dil;
  VS_REQUIRE(a.n_tiles < 2147483647LL, "conv64_f16x3_pk: %lld tiles", a.n_tiles);
  int dev = 0, cus = 256;
  VS_CHECK_HIP(hipGetDevice(&dev));
  VS_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  long long nwg = cus / 8 * 8;
  if (nwg < 8) nwg = 8;
  if (nwg > (a.n_tiles + 7) / 8 * 8) nwg = (a.n_tiles + 7) / 8 * 8;
  dim3 grid((unsigned)nwg), block(256);
#ifdef VS_ABLATION        // timing ablations of the kernel (tools/conv_bench): make -C voicesplit_amd/csrc ABLATION=1
  if (abl) {      // timing ablations (Mish epilogue), see the kernel's ABL parameter
    switch (abl) {
      case 1: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 1>), grid, block, 0, stream, a); break;
      case 2: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 2>), grid, block, 0, stream, a); break;
      case 4: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 4>), grid, block, 0, stream, a); break;
      case 8: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 8>), grid, block, 0, stream, a); break;
      case 6: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 6>), grid, block, 0, stream, a); break;
      case 7: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 7>), grid, block, 0, stream, a); break;
      case 15: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 15>), grid, block, 0, stream, a); break;
      case 23: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 0, 3>), grid, block, 0, stream, a); break;
      case 24: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 0, 4>), grid, block, 0, stream, a); break;
      case 25: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 0, 5>), grid, block, 0, stream, a); break;
      case 26: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 0, 6>), grid, block, 0, stream, a); break;
      case 28: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 0, 8>), grid, block, 0, stream, a); break;
      default: VS_REQUIRE(false, "conv64_f16x3_pk: unknown ablation %d", abl);
    }
    VS_LAUNCH_CHECK();
    return 0;
  }
#else
  VS_REQUIRE(abl == 0, "conv64_f16x3_pk: timing ablations are not compiled into this library (build with ABLATION=1)");
#endif
  if (a.bn_stats) {      // train-mode forward: conv + bias, no activation, statistics fused
    VS_REQUIRE(act == VS_ACT_NONE, "conv64_f16x3_pk: fused BatchNorm statistics need act = NONE");
    if (math == VS_MATH_CODE_BF16) hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_NONE, 0, 0, 1, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_NONE, 0, 0, 3, true>), grid, block, 0, stream, a);
    VS_LAUNCH_CHECK();
    return 0;
  }
  if (math == VS_MATH_CODE_BF16) {
    switch (act) {
      case VS_ACT_RELU: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_RELU, 0, 0, 1>), grid, block, 0, stream, a); break;
      case VS_ACT_MISH: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH, 0, 0, 1>), grid, block, 0, stream, a); break;
      case VS_ACT_NONE: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_NONE, 0, 0, 1>), grid, block, 0, stream, a); break;
      default: VS_REQUIRE(false, "conv64_f16x3_pk: unknown activation %d", act);
    }
    VS_LAUNCH_CHECK();
    return 0;
  }
  switch (act) {
    case VS_ACT_RELU: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_RELU>), grid, block, 0, stream, a); break;
    case VS_ACT_MISH: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_MISH>), grid, block, 0, stream, a); break;
    case VS_ACT_NONE: hipLaunchKernelGGL((conv64_f16x3_pk_kernel<P, VS_ACT_NONE>), grid, block, 0, stream, a); break;
    default: VS_REQUIRE(false, "conv64_f16x3_pk: unknown activation %d", act);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// 5x5 layers only.  Same contract as vs_conv64_f16x3_fwd_impl; covers rows [0, i_end) of every residue class.
int vs_conv64_f16x3_pk_impl(const float* in, const _Float16* wp, const float* scale, const float* shift,
                            const float* in_scale2, const float* w_scale2, float* out,
                            int B, int T, int F, int dil, int act, unsigned* amax_out, hipStream_t stream, int abl,
                            int i_end, int math, double* bn_stats) {
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "conv64_f16x3_pk: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((long long)kCo * T * F * 4 < (long long)kOob, "conv64_f16x3_pk: T*F=%lld too large for 32-bit offsets", (long long)T * F);
  PkArgs a{in, wp, scale, shift, in_scale2, w_scale2, out, amax_out, bn_stats, B, T, F, dil, 0, 0, 0, 0, 0, 0, 0};
  return launch_pk<2>(a, act, 0, i_end, stream, abl, math);
}
