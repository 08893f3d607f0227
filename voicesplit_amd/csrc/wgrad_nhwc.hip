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

// LDS-DMA through a buffer descriptor: lane i of the wave moves 16 bytes from desc.base + voff (+ soff) to LDS
// m0 + 16 i, and a lane whose offset is outside [0, desc.num_records) moves zeros -- the image border and the rows an
// item does not own cost no address arithmetic: a row is one descriptor (base = the row, num_records = its bytes, or 0
// for a row that does not exist) and the per-lane offset is a launch constant plus a wave-uniform column term.
__device__ __forceinline__ void blds16(u4v desc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(desc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ u4v row_desc(const void* base, long long byte_off, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base) + (unsigned long long)byte_off;
  return u4v{(unsigned)p, (unsigned)(p >> 32) & 0xffffu, bytes, 0x00020000u};
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
  // LDS: two window buffers of WIN a rows and two of R dz rows.  A group's rows sit at FIXED offsets of its buffer (the
  // H rows it shares with the previous group are fetched again -- L2 hits -- instead of being kept in a ring), so every
  // fragment read is base register + immediate: one wave per SIMD issues one instruction per 4 cycles, an address add
  // in front of each read (+ the dependency on it) cost more issue time than the read (tools/issue_probe.hip).
  static constexpr int ABUF = WIN * AROWB, ZBUF = R * ZROWB;
  static constexpr int A_BYTES = 2 * ABUF, LDS_BYTES = A_BYTES + 2 * ZBUF;
  static constexpr int NROWS = WIN + R;                  // DMA rows of a group: a rows 0..WIN-1, then dz rows
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
    {
      const int px = lane >> 2, q = lane & 3;                       // a chunk: 16 pixels x 4 pieces of this workgroup's half
      va_lane = (px - PF) * 128 + ((4 * h + (q ^ (((px >> 3) & 1) << 1))) << 4);
    }
    {
      const int px = lane >> 3, q = lane & 7;                       // dz chunk: 8 pixels x 8 pieces (bit3(px) is added per chunk)
      vz_lane = px * 128 + ((q ^ (((px >> 1) & 1) << 1)) << 4);
    }
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

  // ---- DMA of one group's rows, one ROW per unit ------------------------------------------------------------------------
  // a rows w_first .. w_first + WIN - 1 of the item's class -> a buffer `buf`, dz rows k_first .. k_first + R - 1 -> dz
  // buffer `buf`; rows outside the image / outside [o0, o1) are zeros (descriptor with no records), columns outside the
  // image likewise (range check).  Row rho of the NROWS = WIN + R rows belongs to wave rho % 4; a row is ONE descriptor,
  // ONE M0 and its 1 KiB chunks by immediate offset (which moves the memory and the LDS address alike,
  // tools/lds_dma_offset_probe.hip).  LDS images: a piece q (16 bytes, 4 per pixel) of pixel px holds channel piece
  // q ^ (2 * bit3(px)) of this workgroup's half (rows px and px + 8 of a transposing read land on different bank halves);
  // dz piece q (8 per pixel) of pixel px holds channel piece q ^ (2 u(px)), u = bit1(px) | bit3(px) << 1.
  struct Batch {
    long long row0;                // b * T + cls
    int nk, o0, o1;
    int w_first, k_first, buf;
    int live;                      // 0: nothing to fetch
    unsigned va, vz0, vz1;         // per-lane source offsets inside a row: a; dz chunks 0 / 2 and 1 / 3 (bit3(px) flips piece bit 2)
  };
  int va_lane, vz_lane;            // launch constants of the per-lane source offsets (set by the constructor)

  __device__ __forceinline__ void begin(Batch& bt, const WItem& x, int w_first, int k_first, int buf) const {
    bt.row0 = (long long)x.b * a.T + x.cls;
    bt.nk = x.nk; bt.o0 = x.o0; bt.o1 = x.o1;
    bt.w_first = w_first; bt.k_first = k_first; bt.buf = buf;
    bt.live = 1;
    const int colb = (x.strip * STRIP) << 7;
    bt.va = (unsigned)(va_lane + colb);
    bt.vz0 = (unsigned)(vz_lane + colb);
    bt.vz1 = (unsigned)((vz_lane ^ 64) + colb);
  }
  // unit j of this wave: row rho = wave + 4 j (j < UNITS)
  static constexpr int UNITS = (G::NROWS + 3) / 4;
  template <int J>
  __device__ __forceinline__ void row_unit(const Batch& bt) const {
    if (!bt.live) return;
    const int rho = wave + 4 * J;
    const unsigned rowbytes = (unsigned)a.F * 128u;
    if (4 * J + 3 < G::WIN || rho < G::WIN) {              // an a row (compile-time for all J but the one straddling WIN)
      if (4 * J >= G::WIN) return;
      const int w = bt.w_first + rho;
      const bool ok = (w >= 0) & (w < bt.nk);
      const u4v d = row_desc(a.a, ((bt.row0 + (long long)w * a.dil) * a.F) << 7, ok ? rowbytes : 0u);
      const unsigned dst = lds0 + (unsigned)(bt.buf * G::ABUF + rho * G::AROWB);
      unsigned keep;
      // chunk p: 16 pixels = 2 KiB of the tensor row, 1 KiB of the image (this half's 64 bytes per pixel): the immediate
      // moves both addresses by 1 KiB, the other KiB of the memory step is in the offset register
      static_assert(G::ACPR >= 1 && G::ACPR <= 3, "a row: 1..3 chunks");
      if (G::ACPR == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                     "buffer_load_dwordx4 %4, %2, 0 offen offset:1024 lds\n\tbuffer_load_dwordx4 %5, %2, 0 offen offset:2048 lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(bt.va), "s"(d), "s"(dst), "v"(bt.va + 1024u), "v"(bt.va + 2048u) : "memory");
      else if (G::ACPR == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                     "buffer_load_dwordx4 %4, %2, 0 offen offset:1024 lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(bt.va), "s"(d), "s"(dst), "v"(bt.va + 1024u) : "memory");
      else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(bt.va), "s"(d), "s"(dst) : "memory");
    } else {
      const int r = rho - G::WIN;
      if (r >= R) return;
      const int k2 = bt.k_first + r;
      const bool ok = (k2 >= bt.o0) & (k2 < bt.o1);
      const u4v d = row_desc(a.dz, ((bt.row0 + (long long)k2 * a.dil) * a.F) << 7, ok ? rowbytes : 0u);
      const unsigned dst = lds0 + (unsigned)(G::A_BYTES + bt.buf * G::ZBUF + r * G::ZROWB);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                   "buffer_load_dwordx4 %4, %2, 0 offen offset:1024 lds\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:2048 lds\n\t"
                   "buffer_load_dwordx4 %4, %2, 0 offen offset:3072 lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(bt.vz0), "s"(d), "s"(dst), "v"(bt.vz1) : "memory");
    }
  }
  template <int... Js>
  __device__ __forceinline__ void all_units(const Batch& bt, std::integer_sequence<int, Js...>) const { (row_unit<Js>(bt), ...); }
  __device__ __forceinline__ void fetch_all(const Batch& bt) const { all_units(bt, std::make_integer_sequence<int, UNITS>()); }
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
  // a step at a group's edge has as few as 4 MFMAs (64 cycles); two steps ahead did not cover the LDS latency there
  static constexpr int PD = 4;

  f32x4 acc[4][NTW];
  unsigned aoff[KF][2], zoff[4][2];    // per-lane byte offsets of the transposing reads inside a row image (launch constants)
  unsigned lds0;
  static constexpr int abl = ABL;      // VS_ABLATION instances: 1 = no DMA after the first groups, 2 = no fragment reads, 4 = no MFMAs

  struct GroupRegs {
    s4v bq[PD + 1][2];                 // a fragments in flight: read PD steps ahead of their MFMAs
    s4v zf[ZR][4][2];
    unsigned ab[KF][2], zb[4][2];      // the read bases of this group's buffers; window row w / output row r are immediates
    int rv;
  };

  __device__ __forceinline__ unsigned a_addr(const GroupRegs& gr, int w, int df, int hf) const { return gr.ab[df][hf] + (unsigned)(w * G::AROWB); }
  __device__ __forceinline__ unsigned z_addr(const GroupRegs& gr, int r, int cb, int hf) const { return gr.zb[cb][hf] + (unsigned)(r * G::ZROWB); }
  template <int RV, int SI, int hf>
  __device__ __forceinline__ void read_b(GroupRegs& gr) const {
    constexpr int code = S::step_at(RV, SI), w = code / KF, df = code % KF;
    if constexpr (abl & 2) return;
    gr.bq[SI % (PD + 1)][hf] = ds_read_tr16(a_addr(gr, w, df, hf));
  }
  template <int r, int cb, int hf>
  __device__ __forceinline__ void read_z(GroupRegs& gr) const {
    if constexpr (abl & 2) return;
    gr.zf[r % ZR][cb][hf] = ds_read_tr16(z_addr(gr, r, cb, hf));
  }

  using Walk = WgradWalk<KT, KF>;
  using Batch = typename Walk::Batch;

  // Step SI of a group = the MFMAs of a fragment (w, df): every owned tap (dt, df) whose output row w - dt lies in the
  // group, 4 co blocks each.  A wave issues in order, so whatever else the step has to do -- the DMA chunk, the two reads
  // of the fragment PD steps ahead, this window row's share of the next output row's dz fragments -- only hides
  // under the matrix pipe if it sits BETWEEN the MFMAs, one unit per MFMA (batched in front of them it cost 20-45 idle
  // cycles per step: the weight gradient ran at MFMA-only time + read time + DMA time).
  static constexpr int nvalid_dt(int RV, int w, int df) {
    int n = 0;
    for (int dt = S::DTLO; dt <= S::DTHI; ++dt) n += (S::owned(dt, df) && w - dt >= 0 && w - dt < RV) ? 1 : 0;
    return n;
  }
  static constexpr int dt_at(int RV, int w, int df, int idx) {        // idx-th valid dt
    int n = 0;
    for (int dt = S::DTLO; dt <= S::DTHI; ++dt)
      if (S::owned(dt, df) && w - dt >= 0 && w - dt < RV) { if (n == idx) return dt; ++n; }
    return -1;
  }
  static constexpr int rmin(int RV, int w, int df) {                  // smallest output row the step touches
    int m = 1 << 20;
    for (int dt = S::DTLO; dt <= S::DTHI; ++dt)
      if (S::owned(dt, df) && w - dt >= 0 && w - dt < RV && w - dt < m) m = w - dt;
    return m;
  }
  // dz fragments read in step (w, df): co blocks cb with cb % (steps of window row w) == rank of df, of output row w + 1 - DTLO
  static constexpr int nz_cb(int RV, int w, int df) {
    const int rn = w + 1 - S::DTLO;
    if (rn < 1 || rn >= RV) return 0;
    const int nw = S::steps_of_row(RV, w), k = S::rank_in_row(RV, w, df);
    int n = 0;
    for (int cb = 0; cb < 4; ++cb) n += (cb % nw == k) ? 1 : 0;
    return n;
  }
  static constexpr int z_cb_at(int RV, int w, int df, int idx) {
    const int nw = S::steps_of_row(RV, w), k = S::rank_in_row(RV, w, df);
    int n = 0;
    for (int cb = 0; cb < 4; ++cb)
      if (cb % nw == k) { if (n == idx) return cb; ++n; }
    return -1;
  }

  // side unit U of step SI: 0 = DMA chunk, 1 / 2 = the halves of fragment SI + PD, 3.. = dz fragment halves
  template <int RV, int SI, int U>
  __device__ __forceinline__ void unit(GroupRegs& gr, const Walk& wk, const Batch& bt) {
    constexpr int NS = S::nsteps(RV);
    constexpr int code = S::step_at(RV, SI), w = code / KF, df = code % KF;
    if constexpr (U == 0) {
      if constexpr (SI < Walk::UNITS) wk.template row_unit<SI>(bt);      // the next group's rows: one row per step
    } else if constexpr (U <= 2) {
      if constexpr (SI + PD < NS) read_b<RV, SI + PD, U - 1>(gr);
    } else if constexpr (U < 3 + 2 * nz_cb(RV, w, df)) {
      constexpr int cb = z_cb_at(RV, w, df, (U - 3) / 2);
      read_z<w + 1 - S::DTLO, cb, (U - 3) % 2>(gr);
    }
  }
  template <int RV, int SI, int M>
  __device__ __forceinline__ void mfma_unit(GroupRegs& gr, const Walk& wk, const Batch& bt, const vs_bf16x8& bfrag) {
    constexpr int code = S::step_at(RV, SI), w = code / KF, df = code % KF;
    constexpr int dt = dt_at(RV, w, df, M / 4), cb = M % 4, r = w - dt, j = dt * KF + df - T0;
    if constexpr (!(abl & 4))
      acc[cb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_of(gr.zf[r % ZR][cb][0], gr.zf[r % ZR][cb][1]), bfrag, acc[cb][j], 0, 0, 0);
    unit<RV, SI, M>(gr, wk, bt);
    __builtin_amdgcn_sched_barrier(0);
  }
  template <int RV, int SI, int... Ms>
  __device__ __forceinline__ void mfma_units(GroupRegs& gr, const Walk& wk, const Batch& bt, const vs_bf16x8& bfrag, std::integer_sequence<int, Ms...>) {
    (mfma_unit<RV, SI, Ms>(gr, wk, bt, bfrag), ...);
  }
  template <int RV, int SI, int U0, int... Us>
  __device__ __forceinline__ void tail_units(GroupRegs& gr, const Walk& wk, const Batch& bt, std::integer_sequence<int, Us...>) {
    (unit<RV, SI, U0 + Us>(gr, wk, bt), ...);
  }

  template <int RV, int SI>
  __device__ __forceinline__ void step(GroupRegs& gr, const Walk& wk, const Batch& bt) {
    constexpr int code = S::step_at(RV, SI), w = code / KF, df = code % KF;
    constexpr int NM = 4 * nvalid_dt(RV, w, df), NU = 3 + 2 * nz_cb(RV, w, df);
    // wave-uniform: a step none of whose output rows exists in this (tail) group is skipped -- those steps form a suffix;
    // inside an executed step rows past the group's end are zero dz rows
    if (rmin(RV, w, df) < gr.rv) {
      const vs_bf16x8 bfrag = frag_of(gr.bq[SI % (PD + 1)][0], gr.bq[SI % (PD + 1)][1]);
      mfma_units<RV, SI>(gr, wk, bt, bfrag, std::make_integer_sequence<int, NM>());
      if constexpr (NU > NM) tail_units<RV, SI, NM>(gr, wk, bt, std::make_integer_sequence<int, NU - NM>());
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  template <int RV, int... SIs>
  __device__ __forceinline__ void steps(GroupRegs& gr, const Walk& wk, const Batch& bt, std::integer_sequence<int, SIs...>) {
    (step<RV, SIs>(gr, wk, bt), ...);
  }

  // a group of rv <= R output rows (R, or the even tail of an item; a row past the item's end is a zero dz row).  ONE
  // body with a wave-uniform guard per tap instead of one instance per row count: with several instances the
  // accumulators, which live across groups, get a copy per instance (512 registers + spills).
  __device__ __forceinline__ void group(const Walk& wk, const Batch& bt, int buf, int rv) {
    constexpr int RV = R;
    static_assert(S::nsteps(R) >= Walk::UNITS, "every DMA row unit needs a step");
    GroupRegs gr;
    gr.rv = rv;
#pragma unroll
    for (int df = 0; df < KF; ++df)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) gr.ab[df][hf] = lds0 + (unsigned)(buf * G::ABUF) + aoff[df][hf];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) gr.zb[cb][hf] = lds0 + (unsigned)(G::A_BYTES + buf * G::ZBUF) + zoff[cb][hf];
    if constexpr (abl & 2) {          // no fragment reads: defined (not foldable) register contents instead
      const short v = (short)(0x3c00 + (threadIdx.x & 7));
#pragma unroll
      for (int q = 0; q < PD + 1; ++q) { gr.bq[q][0] = s4v{v, v, v, v}; gr.bq[q][1] = s4v{v, v, v, v}; }
#pragma unroll
      for (int q = 0; q < ZR; ++q)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) { gr.zf[q][cb][0] = s4v{v, v, v, v}; gr.zf[q][cb][1] = s4v{v, v, v, v}; }
    }
    read_b<RV, 0, 0>(gr); read_b<RV, 0, 1>(gr);
    read_z<0, 0, 0>(gr); read_z<0, 0, 1>(gr);
    read_z<0, 1, 0>(gr); read_z<0, 1, 1>(gr);
    read_z<0, 2, 0>(gr); read_z<0, 2, 1>(gr);
    read_z<0, 3, 0>(gr); read_z<0, 3, 1>(gr);
    if constexpr (PD > 1 && S::nsteps(RV) > 1) { read_b<RV, 1, 0>(gr); read_b<RV, 1, 1>(gr); }
    if constexpr (PD > 2 && S::nsteps(RV) > 2) { read_b<RV, 2, 0>(gr); read_b<RV, 2, 1>(gr); }
    if constexpr (PD > 3 && S::nsteps(RV) > 3) { read_b<RV, 3, 0>(gr); read_b<RV, 3, 1>(gr); }
    __builtin_amdgcn_sched_barrier(0);
    steps<RV>(gr, wk, bt, std::make_integer_sequence<int, S::nsteps(RV)>());
  }
};

template <int KT, int KF, int TH, int ABL>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const lds_byte* smem) {
  using G = WGeo<KT, KF>;
  constexpr int P = G::P;
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
  typename WgradWalk<KT, KF>::Batch bt;
  bt.live = 0;
  int pbuf = 0;                      // buffer the next fetched group goes to
  auto pf_begin = [&]() {            // describe the fetch of the group at the prefetch cursor (issued row by row), advance the cursor
    const int ro = pf.o0 + pf_g * R;
    wk.begin(bt, pf, ro - P, ro, pbuf);
    if ((ABL & 1) && abl_groups++ >= 2) bt.live = 0;
    pbuf ^= 1;
    if (++pf_g >= pf.ngroups) { pf_it += npairs; pf_seek(); }
  };
  pf_seek();
  if (pf_live) { pf_begin(); wk.fetch_all(bt); }

  WItem cur;
  int cbuf = 0;
  for (int it = pair; it < a.n_items; it += npairs) {
    if (!wk.decode(it, cur)) continue;
    for (int gidx = 0; gidx < cur.ngroups; ++gidx) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      bt.live = 0;
      if (pf_live) pf_begin();
      const int left = cur.o1 - (cur.o0 + gidx * R);
      core.group(wk, bt, cbuf, left > 6 ? 8 : left > 4 ? 6 : left > 2 ? 4 : 2);
      cbuf ^= 1;
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
  a.abl = vs_opt(VS_OPT_ABLATION);
#endif
  if (KT == 5 && KF == 5) return launch_wgrad<5, 5>(a, dw, stream);
  if (KT == 7 && KF == 1) return launch_wgrad<7, 1>(a, dw, stream);
  VS_REQUIRE(false, "nhwc wgrad: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  return -1;
}
