// Weight gradient of the 64 -> 64 convs in the bf16 configuration (VS_MATH_BF16): cnn2 (7x1) and cnn3..cnn7 (5x5,
// time dilation 1..16), what autograd computes for models/voicesplit/model.py:21-48 under train.py:110:
//   dw[co][ci][dt][df] = sum_{b,t,f} dz[b][t][f][co] * a[b][t + (dt-KT/2) dil][f + df - KF/2][ci]
// dz, a: channels-last bf16 [B][T][F][64].  As a GEMM: M = co, N = ci (per tap), K = pixels -- K runs ALONG the
// pixel index while both operands are stored channel-contiguous, so their MFMA fragments (8 consecutive k for a
// fixed m / n) are transposes of what lies in memory.  gfx950's ds_read_b64_tr_b16 does that transpose on the way
// out of LDS: the rows go HBM -> LDS by LDS-DMA exactly as they are ([pixel][channel]) and a lane's fragment is two
// transposing 8-byte reads -- no VALU, no second copy of either tensor.  (tools/probe_cdna4.hip pins the lane map of
// the instruction; the fp32-layout kernel conv64_wgrad_ring4 gets K-contiguous fragments from its [channel][pixel]
// layout instead and pays the layout on the forward side.)
//
// Decomposition: the same walk as the forward kernel (conv_nhwc.hip) -- item = (utterance, class, 32-column strip,
// row segment), groups of R = 8 rows, LDS rings for the a rows (R + KT - 1 per group, R new per group) and the dz rows,
// DMA one group ahead, zero page for everything outside the image / the item.  The 64 x 64 x taps accumulators do
// not fit one workgroup's registers, so a PAIR of workgroups shares every item: workgroup h = (blockIdx >> 3) & 1 owns input
// channels [32h, 32h+32) and stages only those 64 bytes of each a pixel.  Inside it wave (cbi, th) owns input-channel
// block cbi (16) and one half of the taps: 4 (co blocks) x 13 (taps) = 52 accumulator tiles = 208 registers that live
// for the whole launch.  Per output row: 4 dz fragments (shared by all its taps) + one a fragment per tap -> 4 MFMAs
// per a fragment (0.65 8-byte LDS reads per MFMA).  Partial sums go to part[pair][co][ci][tap]; vs_reduce_partials
// adds the pairs.
#include <cstdlib>
#include <utility>

#include "vs_internal.h"

namespace {

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ u4v g_wg_zero_page[4];

constexpr int STRIP = 32;
constexpr int R = 8;

struct WgradArgs {
  const unsigned short* dz;         // [B][T][F][64] bf16, channel = co
  const unsigned short* a;          // [B][T][F][64] bf16, channel = ci
  float* part;                      // [pairs][64 co][64 ci][taps]
  int B, T, F, dil;
  int nstrip, nseg, seg_rows, n_items;
  int abl;                          // VS_ABLATION builds only (tools/wgrad_ablation.py): selects a timing-ablation instance
                                    // (1 = no DMA after a workgroup's first groups, 2 = no LDS fragment reads, 4 = no MFMAs; results are wrong)
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// transposing LDS read: the 16 lanes of a group supply the addresses of a [4 rows][16 columns] block of halves
// (lane i: row i/4, columns 4(i%4)..+3) and lane i receives column i, rows 0..3 (tools/probe_cdna4.hip, probe A).
// The builtin (not inline asm): the compiler then counts the read on lgkmcnt itself.
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ s4v ds_read_tr16(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)addr);
}
__device__ __forceinline__ vs_bf16x8 frag_of(s4v lo, s4v hi) {
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(vs_bf16x8, v);
}

template <int KT, int KF>
struct WGeo {
  static constexpr int P = KT / 2, PF = KF / 2, H = KT - 1, NTAP = KT * KF;
  static constexpr int NT0 = (NTAP + 1) / 2;             // taps of tap half 0 (half 1: the rest)
  static constexpr int APX = STRIP + 2 * PF;             // staged a pixels per row, 64 bytes each (this workgroup's 32 channels)
  static constexpr int ACPR = (APX * 4 + 63) / 64;       // 1 KiB DMA chunks per a row
  static constexpr int AROWB = ACPR * 1024;
  static constexpr int ZROWB = STRIP * 128;              // a dz row: 32 pixels x 64 channels = 4 chunks
  static constexpr int ZCPR = 4;
  static constexpr int WIN = R + H;
  static constexpr int NRA = 2 * WIN;                    // a ring rows
  static constexpr int NRZ = 2 * R;                      // dz ring rows
  static constexpr int A_BYTES = NRA * AROWB, LDS_BYTES = A_BYTES + NRZ * ZROWB;
};

struct WItem { int b, cls, strip, o0, o1, nk, ngroups; };

template <int KT, int KF>
struct WgradWalk {
  using G = WGeo<KT, KF>;
  static constexpr int P = G::P, PF = G::PF, H = G::H, NTAP = G::NTAP, NT0 = G::NT0;

  const WgradArgs& a;
  int lane, wave, h;                 // h: input-channel half of this workgroup
  unsigned lds0;                     // LDS byte address of the a ring; the dz ring follows at + A_BYTES

  __device__ __forceinline__ WgradWalk(const WgradArgs& a_, const lds_byte* smem) : a(a_) {
    lane = threadIdx.x & 63;
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    h = (int)((blockIdx.x >> 3) & 1);
    lds0 = (unsigned)(uintptr_t)smem;
  }

  __device__ __forceinline__ bool decode(int it, WItem& r) const {
    r.strip = it % a.nstrip;
    const int t1 = it / a.nstrip;
    const int seg = t1 % a.nseg;
    const int t2 = t1 / a.nseg;
    r.cls = t2 % a.dil;
    r.b = t2 / a.dil;
    r.nk = r.cls < a.T ? (a.T - r.cls + a.dil - 1) / a.dil : 0;
    r.o0 = seg * a.seg_rows;
    r.o1 = min(r.nk, r.o0 + a.seg_rows);
    if (r.o0 >= r.o1) return false;
    r.ngroups = (r.o1 - r.o0 + R - 1) / R;
    return true;
  }

  // a rows [w_first, w_first + nrows) of the item's class -> ring positions pos_first.. (mod NRA).  LDS piece q (16 bytes,
  // 4 per pixel) of pixel px holds channel piece q ^ (2 * bit3(px)) of this workgroup's half: rows px and px + 8 of a
  // transposing read land on different bank halves.
  __device__ __forceinline__ void issue_a(const WItem& x, int w_first, int nrows, int pos_first) const {
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_wg_zero_page);
    const long long rel0 = reinterpret_cast<const unsigned char*>(a.a) - zp;
    const long long row0 = (long long)x.b * a.T + x.cls;
    const int nchunks = nrows * G::ACPR;
    for (int c = wave; c < nchunks; c += 4) {
      const int row = c / G::ACPR, part = c - row * G::ACPR;
      const int w = w_first + row;
      int pos = pos_first + row;
      if (pos >= G::NRA) pos -= G::NRA;
      const int e = part * 64 + lane;
      const int px = e >> 2, q = e & 3;
      const int col = x.strip * STRIP - PF + px;
      const bool ok = (w >= 0) & (w < x.nk) & (px < G::APX) & (col >= 0) & (col < a.F);
      const int piece = 4 * h + (q ^ (((px >> 3) & 1) << 1));
      const long long off = rel0 + ((((row0 + (long long)w * a.dil) * a.F + col) << 7) + (piece << 4));
      const unsigned char* src = zp + (off & -(long long)ok);
      glds16(src, (unsigned)__builtin_amdgcn_readfirstlane(lds0 + (unsigned)(pos * G::AROWB + part * 1024)));
    }
  }

  // dz rows [k_first, k_first + R) -> dz ring rows zpos_first.. ; rows outside [o0, o1) of the item are zero.
  // LDS piece q (8 per pixel) of pixel px holds channel piece q ^ (2 u(px)), u = bit1(px) | bit3(px) << 1.
  __device__ __forceinline__ void issue_z(const WItem& x, int k_first, int zpos_first) const {
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_wg_zero_page);
    const long long rel0 = reinterpret_cast<const unsigned char*>(a.dz) - zp;
    const long long row0 = (long long)x.b * a.T + x.cls;
    for (int c = wave; c < R * G::ZCPR; c += 4) {
      const int row = c / G::ZCPR, part = c - row * G::ZCPR;
      const int k = k_first + row;
      int pos = zpos_first + row;
      if (pos >= G::NRZ) pos -= G::NRZ;
      const int px = part * 8 + (lane >> 3), q = lane & 7;
      const int col = x.strip * STRIP + px;
      const bool ok = (k >= x.o0) & (k < x.o1) & (col < a.F);
      const int u = ((px >> 1) & 1) | (((px >> 3) & 1) << 1);
      const long long off = rel0 + ((((row0 + (long long)k * a.dil) * a.F + col) << 7) + ((q ^ (u << 1)) << 4));
      const unsigned char* src = zp + (off & -(long long)ok);
      glds16(src, (unsigned)__builtin_amdgcn_readfirstlane(lds0 + (unsigned)(G::A_BYTES + pos * G::ZROWB + part * 1024)));
    }
  }
};

// Which (window row, column shift) fragments a wave reads and what it does with each.  A wave owns taps [T0, T0 + NTW) of
// the row-major tap order, i.e. the time offsets dt in [DTLO, DTHI] (the first / last of them possibly for a subset of
// the column shifts).  The a fragment of window row w, shift df is the operand of EVERY owned tap (dt, df) -- for output
// row w - dt -- so the group is walked by window rows: one fragment read feeds up to DTN x 4 MFMAs (the walk by output
// rows read it once per tap: 4 MFMAs, and the kernel sat on the LDS read port).  The dz fragments of the DTN output rows in
// play stay in registers (ring of DTN + 1 rows x 4 co blocks).
template <int KT, int KF, int TH>
struct WSched {
  using G = WGeo<KT, KF>;
  static constexpr int NTAP = G::NTAP, H = G::H;
  static constexpr int T0 = TH ? G::NT0 : 0, NTW = TH ? NTAP - G::NT0 : G::NT0;
  static constexpr int DTLO = T0 / KF, DTHI = (T0 + NTW - 1) / KF, DTN = DTHI - DTLO + 1, ZR = DTN + 1;
  static constexpr bool owned(int dt, int df) { const int t = dt * KF + df; return dt >= DTLO && dt <= DTHI && t >= T0 && t < T0 + NTW; }
  // does fragment (w, df) feed any MFMA of a group of RV output rows?
  static constexpr bool valid(int RV, int w, int df) {
    for (int dt = DTLO; dt <= DTHI; ++dt)
      if (owned(dt, df) && w - dt >= 0 && w - dt < RV) return true;
    return false;
  }
  static constexpr int nsteps(int RV) {
    int n = 0;
    for (int w = 0; w < RV + H; ++w)
      for (int df = 0; df < KF; ++df) n += valid(RV, w, df) ? 1 : 0;
    return n;
  }
  static constexpr int step_at(int RV, int s) {                   // w * KF + df of step s
    int n = 0;
    for (int w = 0; w < RV + H; ++w)
      for (int df = 0; df < KF; ++df)
        if (valid(RV, w, df)) { if (n == s) return w * KF + df; ++n; }
    return -1;
  }
  static constexpr int steps_of_row(int RV, int w) { int n = 0; for (int df = 0; df < KF; ++df) n += valid(RV, w, df) ? 1 : 0; return n; }
  static constexpr int rank_in_row(int RV, int w, int df) { int n = 0; for (int d = 0; d < df; ++d) n += valid(RV, w, d) ? 1 : 0; return n; }
};

template <int KT, int KF, int TH, int ABL>
struct WgradCore {
  using G = WGeo<KT, KF>;
  using S = WSched<KT, KF, TH>;
  static constexpr int NTW = S::NTW, T0 = S::T0, ZR = S::ZR;

  f32x4 acc[4][NTW];
  unsigned aoff[KF][2], zoff[4][2];
  unsigned lds0;
  static constexpr int abl = ABL;      // VS_ABLATION instances: 1 = no DMA after the first groups, 2 = no fragment reads, 4 = no MFMAs

  struct GroupRegs {
    s4v bq[3][2];
    s4v zf[ZR][4][2];
    int cq, zq, rv;
  };

  __device__ __forceinline__ unsigned a_addr(const GroupRegs& gr, int w, int df, int hf) const {
    int pos = gr.cq + w;
    if (pos >= G::NRA) pos -= G::NRA;
    return lds0 + (unsigned)(pos * G::AROWB) + aoff[df][hf];
  }
  __device__ __forceinline__ unsigned z_addr(const GroupRegs& gr, int r, int cb, int hf) const {
    int pos = gr.zq + r;
    if (pos >= G::NRZ) pos -= G::NRZ;
    return lds0 + (unsigned)(G::A_BYTES + pos * G::ZROWB) + zoff[cb][hf];
  }
  template <int RV, int SI>
  __device__ __forceinline__ void read_b(GroupRegs& gr) const {
    constexpr int code = S::step_at(RV, SI), w = code / KF, df = code % KF;
    if constexpr (abl & 2) return;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) gr.bq[SI % 3][hf] = ds_read_tr16(a_addr(gr, w, df, hf));
  }
  template <int r, int cb>
  __device__ __forceinline__ void read_z(GroupRegs& gr) const {
    if constexpr (abl & 2) return;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) gr.zf[r % ZR][cb][hf] = ds_read_tr16(z_addr(gr, r, cb, hf));
  }

  // step SI of a group of RV rows: the read two steps ahead, this window row's share of the next output row's dz
  // fragments, then the MFMAs of fragment SI
  template <int RV, int SI>
  __device__ __forceinline__ void step(GroupRegs& gr) {
    constexpr int NS = S::nsteps(RV);
    constexpr int code = S::step_at(RV, SI), w = code / KF, df = code % KF;
    if constexpr (SI + 2 < NS) read_b<RV, SI + 2>(gr);
    // output row rn = w + 1 - DTLO is first used at window row w + 1: its 4 co blocks are spread over this row's steps
    constexpr int rn = w + 1 - S::DTLO;
    if constexpr (rn >= 1 && rn < RV) {
      constexpr int nw = S::steps_of_row(RV, w), k = S::rank_in_row(RV, w, df);
      if constexpr (0 % nw == k) read_z<rn, 0>(gr);
      if constexpr (1 % nw == k) read_z<rn, 1>(gr);
      if constexpr (2 % nw == k) read_z<rn, 2>(gr);
      if constexpr (3 % nw == k) read_z<rn, 3>(gr);
    }
    __builtin_amdgcn_sched_barrier(0);
    const vs_bf16x8 bfrag = frag_of(gr.bq[SI % 3][0], gr.bq[SI % 3][1]);
    mfmas<RV, w, df>(gr, bfrag, std::make_integer_sequence<int, S::DTN>());
    __builtin_amdgcn_sched_barrier(0);
  }
  template <int RV, int w, int df, int... Ds>
  __device__ __forceinline__ void mfmas(GroupRegs& gr, const vs_bf16x8& bfrag, std::integer_sequence<int, Ds...>) {
    (mfma4<RV, w, df, S::DTLO + Ds>(gr, bfrag), ...);
  }
  template <int RV, int w, int df, int dt>
  __device__ __forceinline__ void mfma4(GroupRegs& gr, const vs_bf16x8& bfrag) {
    constexpr int r = w - dt;
    if constexpr (S::owned(dt, df) && r >= 0 && r < RV) {
      constexpr int j = dt * KF + df - T0;
      if (r < gr.rv && !(abl & 4)) {               // wave-uniform: the group's last rows may not exist (their dz rows are zeros either way)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_of(gr.zf[r % ZR][cb][0], gr.zf[r % ZR][cb][1]), bfrag, acc[cb][j], 0, 0, 0);
      }
    }
  }
  template <int RV, int... SIs>
  __device__ __forceinline__ void steps(GroupRegs& gr, std::integer_sequence<int, SIs...>) {
    (step<RV, SIs>(gr), ...);
  }

  // a group of rv <= R output rows (R, or the even tail of an item; a row past the item's end is a zero dz row).  ONE
  // body with a wave-uniform guard per tap instead of one instance per row count: with several instances the
  // accumulators, which live across groups, get a copy per instance (512 registers + spills).
  __device__ __forceinline__ void group(int cq, int zq, int rv) {
    constexpr int RV = R;
    GroupRegs gr;
    gr.cq = cq;
    gr.zq = zq;
    gr.rv = rv;
    if constexpr (abl & 2) {          // no fragment reads: defined (not foldable) register contents instead
      const short v = (short)(0x3c00 + (threadIdx.x & 7));
#pragma unroll
      for (int q = 0; q < 3; ++q) { gr.bq[q][0] = s4v{v, v, v, v}; gr.bq[q][1] = s4v{v, v, v, v}; }
#pragma unroll
      for (int q = 0; q < ZR; ++q)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) { gr.zf[q][cb][0] = s4v{v, v, v, v}; gr.zf[q][cb][1] = s4v{v, v, v, v}; }
    }
    read_z<0, 0>(gr);
    read_z<0, 1>(gr);
    read_z<0, 2>(gr);
    read_z<0, 3>(gr);
    read_b<RV, 0>(gr);
    if constexpr (S::nsteps(RV) > 1) read_b<RV, 1>(gr);
    __builtin_amdgcn_sched_barrier(0);
    steps<RV>(gr, std::make_integer_sequence<int, S::nsteps(RV)>());
  }
};

template <int KT, int KF, int TH, int ABL>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const lds_byte* smem) {
  using G = WGeo<KT, KF>;
  constexpr int P = G::P, H = G::H;
  WgradWalk<KT, KF> wk(a, smem);
  WgradCore<KT, KF, TH, ABL> core;
  constexpr int NTW = WgradCore<KT, KF, TH, ABL>::NTW, T0 = WgradCore<KT, KF, TH, ABL>::T0;
  const int lane = wk.lane, g = lane >> 4, i = lane & 15, cbi = wk.wave & 1;
  core.lds0 = wk.lds0;
  int abl_groups = 0;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int t = 0; t < NTW; ++t) core.acc[cb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane byte offsets of the transposing reads inside a row image, for both 4-pixel halves of a fragment.
  //   a image (64 bytes / pixel): pixel p = df + 8g + 4 half + i/4, unit (32 bytes) = cbi ^ bit3(p), + 8 (i%4)
  //   dz image (128 bytes / pixel): pixel p = 8g + 4 half + i/4, unit = cb ^ (bit1(p) | bit3(p) << 1), + 8 (i%4)
#pragma unroll
  for (int df = 0; df < KF; ++df)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int p = df + 8 * g + 4 * hf + (i >> 2);
      core.aoff[df][hf] = (unsigned)(p * 64 + ((cbi ^ ((p >> 3) & 1)) << 5) + (i & 3) * 8);
    }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int p = 8 * g + 4 * hf + (i >> 2);
      const int u = ((p >> 1) & 1) | (((p >> 3) & 1) << 1);
      core.zoff[cb][hf] = (unsigned)(p * 128 + ((cb ^ u) << 5) + (i & 3) * 8);
    }

  // prefetch cursor (the group after the one being computed)
  WItem pf;
  // the two workgroups of a pair read the same dz rows and the two halves of the same a lines: they are blocks b and
  // b + 8, which the dispatcher places on the same XCD (block % 8), so the second read of a line is an L2 hit
  // (pairs (2p, 2p+1) sat on different XCDs: 6.3 GB fetched per launch against 2.96 GB of operands; now 3.2 GB)
  const int pair = (int)((blockIdx.x & 7) + 8 * (blockIdx.x >> 4)), npairs = (int)(gridDim.x >> 1);
  int pf_it = pair, pf_g = 0;
  bool pf_live = false;
  auto pf_seek = [&]() {
    pf_live = false;
    while (pf_it < a.n_items) {
      if (wk.decode(pf_it, pf)) { pf_g = 0; pf_live = true; return; }
      pf_it += npairs;
    }
  };
  int wp = 0, zwp = 0;               // ring positions the next DMA'd a row / dz row go to
  auto pf_issue = [&]() {
    const int ro = pf.o0 + pf_g * R;
    const int first = pf_g == 0 ? 0 : H;
    const int nrows = G::WIN - first;
    const bool dma = !(ABL & 1) || abl_groups++ < 2;
    if (dma) wk.issue_a(pf, ro - P + first, nrows, wp);
    wp += nrows;
    if (wp >= G::NRA) wp -= G::NRA;
    if (dma) wk.issue_z(pf, ro, zwp);
    zwp += R;
    if (zwp >= G::NRZ) zwp -= G::NRZ;
    if (++pf_g >= pf.ngroups) { pf_it += npairs; pf_seek(); }
  };
  pf_seek();
  if (pf_live) pf_issue();

  WItem cur;
  int cq = 0, zq = 0;
  for (int it = pair; it < a.n_items; it += npairs) {
    if (!wk.decode(it, cur)) continue;
    for (int gidx = 0; gidx < cur.ngroups; ++gidx) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int wp_before = wp;
      const bool pf_new_item = pf_live && pf_g == 0;
      if (pf_live) pf_issue();
      const int left = cur.o1 - (cur.o0 + gidx * R);
      core.group(cq, zq, left > 6 ? 8 : left > 4 ? 6 : left > 2 ? 4 : 2);
      if (gidx + 1 < cur.ngroups) {
        cq += R; if (cq >= G::NRA) cq -= G::NRA;
      } else if (pf_new_item) {
        cq = wp_before;
      }
      zq += R; if (zq >= G::NRZ) zq -= G::NRZ;
    }
  }

  // partial sums of this pair: part[pair][co][ci][tap]
  float* dst = a.part + (size_t)pair * 64 * 64 * G::NTAP;
  const int ci = 32 * wk.h + 16 * cbi + i;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = 16 * cb + 4 * g + q;
        dst[((size_t)co * 64 + ci) * G::NTAP + T0 + j] = core.acc[cb][j][q];
      }
}

template <int KT, int KF, int ABL = 0>
__global__ __launch_bounds__(256, 1)
void nhwc_wgrad_kernel(WgradArgs a) {
  using G = WGeo<KT, KF>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
  const int th = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7));
  if (th == 0) wgrad_body<KT, KF, 0, ABL>(a, (const lds_byte*)smem);
  else wgrad_body<KT, KF, 1, ABL>(a, (const lds_byte*)smem);
}

int wg_num_cus() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  return v;
}

template <int KT, int KF>
int launch_wgrad(WgradArgs a, float* dw, hipStream_t stream) {
  using G = WGeo<KT, KF>;
  const int nk_max = (a.T + a.dil - 1) / a.dil;
  const long long base_items = (long long)a.B * a.dil * a.nstrip;
  int nseg = (int)((2048 + base_items - 1) / base_items);
  if (nseg > nk_max / 16) nseg = nk_max / 16;
  if (nseg < 1) nseg = 1;
  int seg_rows = (nk_max + nseg - 1) / nseg;
  seg_rows = (seg_rows + R - 1) / R * R;
  nseg = (nk_max + seg_rows - 1) / seg_rows;
  a.nseg = nseg;
  a.seg_rows = seg_rows;
  const long long n_items = base_items * nseg;
  VS_REQUIRE(n_items < (1LL << 30), "nhwc wgrad: too many work items");
  a.n_items = (int)n_items;
  static int cus = wg_num_cus();
  long long pairs = cus / 16 * 8;                       // blocks come in groups of 16: 8 pairs (b, b + 8)
  if (pairs > (n_items + 7) / 8 * 8) pairs = (n_items + 7) / 8 * 8;
  if (pairs > VS_NHWC_WGRAD_MAX_PAIRS) pairs = VS_NHWC_WGRAD_MAX_PAIRS;
  if (pairs < 8) pairs = 8;
#ifdef VS_ABLATION
  switch (a.abl) {
#define VS_WG_CASE(N) case N: hipLaunchKernelGGL((nhwc_wgrad_kernel<KT, KF, N>), dim3((unsigned)(2 * pairs)), dim3(256), 0, stream, a); break;
    VS_WG_CASE(1) VS_WG_CASE(2) VS_WG_CASE(3) VS_WG_CASE(4) VS_WG_CASE(5) VS_WG_CASE(6)
#undef VS_WG_CASE
    default: hipLaunchKernelGGL((nhwc_wgrad_kernel<KT, KF>), dim3((unsigned)(2 * pairs)), dim3(256), 0, stream, a);
  }
#else
  hipLaunchKernelGGL((nhwc_wgrad_kernel<KT, KF>), dim3((unsigned)(2 * pairs)), dim3(256), 0, stream, a);
#endif
  VS_LAUNCH_CHECK();
  return vs_reduce_partials_impl(a.part, (int)pairs, 64 * 64 * G::NTAP, dw, stream);
}

}  // namespace

size_t vs_nhwc_wgrad_partial_floats(int KT, int KF) { return (size_t)VS_NHWC_WGRAD_MAX_PAIRS * 64 * 64 * KT * KF; }

// dw [64][64][KT][KF] fp32 <- dz, a channels-last bf16; part: vs_nhwc_wgrad_partial_floats(KT, KF) floats of scratch
int vs_nhwc_wgrad_impl(const void* dz, const void* a_in, float* part, float* dw, int B, int T, int F, int KT, int KF, int dil,
                       hipStream_t stream) {
  VS_REQUIRE(dz && a_in && part && dw, "nhwc wgrad: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "nhwc wgrad: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((reinterpret_cast<uintptr_t>(dz) & 15) == 0 && (reinterpret_cast<uintptr_t>(a_in) & 15) == 0, "nhwc wgrad: operands must be 16-byte aligned");
  WgradArgs a{reinterpret_cast<const unsigned short*>(dz), reinterpret_cast<const unsigned short*>(a_in), part, B, T, F, dil,
              (F + STRIP - 1) / STRIP, 1, 0, 0, 0};
#ifdef VS_ABLATION
  if (const char* e = getenv("VOICESPLIT_WGRAD_ABL")) a.abl = atoi(e);
#endif
  if (KT == 5 && KF == 5) return launch_wgrad<5, 5>(a, dw, stream);
  if (KT == 7 && KF == 1) return launch_wgrad<7, 1>(a, dw, stream);
  VS_REQUIRE(false, "nhwc wgrad: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  return -1;
}
