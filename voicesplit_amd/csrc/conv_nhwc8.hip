// 64 -> 64 5x5 convolution of the conv stack in the bf16 configuration (cnn3..cnn7 of models/voicesplit/model.py:26-48, forward and
// data gradient), the EIGHT-WAVE form: two waves per SIMD.
//
// Why (round 5, VERDICT next #1; tools/occupancy_probe.hip, profiles/r05_probes.txt).  conv_nhwc.hip runs one wave per SIMD with
// the weights of 16 output channels x 64 input channels x 25 taps in 200 registers.  A lone wave issues in order: whatever sits
// between two of its 16-cycle MFMAs beyond ~1 instruction stretches the stream (probe: 16.5 / 17.5 / 24.7 / 39.6 cycles per MFMA
// at 0 / 1 / 2 / 3 VALU instructions per MFMA), and the data gradient's activation-derivative epilogue (the dy form) is ~1.3
// instructions per MFMA: 2.05 ms against the plain conv's 1.65.  With a second wave on the SIMD the same instruction mix costs
// 16.2 / 16.6 / 19.7 / 25.2: one wave's side work issues under the other's MFMAs.  Two waves per SIMD means 256 registers per
// wave, so the weights are split once more, as conv_nhwc_f16x3.hip does for its two planes:
//   * wave = (block of 16 output channels mblk = wave & 3, K half kh = wave >> 2: input channels [32 kh, 32 kh + 32)):
//     25 A fragments = 100 registers, kept in AGPRs with the accumulators (asm MFMA blocks with `a` operands: mfma_blocks.inc);
//   * strips are 16 columns (one MFMA column block), groups R = 8 output rows: 8 accumulators per wave;
//   * the two K halves of a channel block are summed across its two waves once per group: each hands the other half of its
//     accumulators through LDS (4 rows x 16 bytes per lane) and finishes the other half of the rows;
//   * that epilogue (scale / shift + activation, or the dy form: activation derivative from the output pixel's z + the two
//     BatchNorm-backward sums; bf16 rounding; one 8-byte store per pixel) is DEFERRED into the next group, cut into micro-ops
//     between its MFMA blocks; the z values of the dy form are loaded at the START of the group that computes their rows, a whole
//     group (~6 us) before they are used;
//   * LDS: two window buffers of 12 rows x 20 pixels (+ one scratch row that absorbs the DMA unit of a wave with no row left) and
//     two exchange areas: 142 KiB.  Same swizzle, same DMA units (a descriptor per row, range checks as predication), same
//     item walk as conv_nhwc.hip; the packed weights are conv_nhwc.hip's (vs_nhwc_pack_impl): wave (mblk, kh) takes fragment
//     (q = mblk, k-chunk = kh) of every tap.
// Same arithmetic as conv_nhwc.hip up to the order of the K = 1600 sum (two halves of 800 added at the end): bf16 operands, fp32
// accumulation; the parity tests of that kernel run on this one with the same bounds (tests/test_gpu_nhwc.py).
#include <utility>

#include "vs_internal.h"

namespace {

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
constexpr float kLog2e = 1.44269504088896340736f;
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) const u4v lds_u4v;
typedef __attribute__((address_space(3))) u4v lds_u4v_rw;

#include "mfma_blocks.inc"

constexpr int STRIP = 16;           // output columns per strip = one MFMA column block
constexpr int NW = 8;               // waves per workgroup

struct Conv8Args {
  const unsigned short* in;         // [B][T][F][64] bf16
  const unsigned short* wpk;        // conv_nhwc.hip's packed weights: [4 co blocks][taps][2 k-chunks][64 lanes][8] bf16
  const float* scale;               // [64]  out = act(acc * scale + shift)
  const float* shift;               // [64]
  unsigned short* out;              // [B][T][F][64] bf16
  double* bn_stats;                 // [VS_BN_STAT_SLOTS][64][2] or NULL
  const unsigned short* z2;         // dy form: the output pixel's z, [B][T][F][64] bf16
  const float* bn2_scale; const float* bn2_shift; const float* bn2_mean; const float* bn2_invstd;   // [64] each
  int B, T, F, dil;
  int nstrip, nseg, seg_rows, n_items;
};

__device__ __forceinline__ int swz(int p) { return ((p >> 1) & 3) << 1; }      // conv_nhwc.hip: conflict-free for every tap column

template <int KT, int KF>
struct Geo8 {
  static constexpr int R = 6, HR = R / 2;      // 25 weight fragments + R accumulators = 124 of the 128 AGPRs a wave gets at two waves per SIMD
  static constexpr int P = KT / 2, PF = KF / 2, H = KT - 1;
  static constexpr int NTAP = KT * KF;
  static constexpr int RWPX = STRIP + 2 * PF;            // staged pixels per row (20)
  static constexpr int CPR = (RWPX * 8 + 63) / 64;       // 1 KiB DMA chunks per row (3)
  static constexpr int ROWB = CPR * 1024;
  static constexpr int WIN = R + H;                      // input rows of a group (12)
  static constexpr int WBUF = (WIN + 1) * ROWB;          // + the scratch row of the idle DMA units
  static constexpr int XCH = HR * 1024;                  // what one wave receives per group: HR rows x 64 lanes x 16 bytes
  static constexpr int LDS_BYTES = 2 * WBUF + 2 * NW * XCH;
  static constexpr int UNITS = (WIN + NW - 1) / NW;      // DMA row units of a wave per group (row rho belongs to wave rho % 8)
};

struct Item { int b, cls, strip, o0, o1, in_end, ngroups; };

template <int KT, int KF, int ACT, bool STATS, bool DY>
struct Walk8 {
  using G = Geo8<KT, KF>;
  static constexpr int R = G::R, HR = G::HR, P = G::P, PF = G::PF, H = G::H, NTAP = G::NTAP;
  static constexpr unsigned kOob = 0x7FFFFFF0u;

  const Conv8Args& a;
  int lane, wave, n, g, mblk, kh;
  u4v wf[NTAP];
  f2v csc[2], csh[2], a1[2], a2[2];     // epilogue constants (DY: BatchNorm scale / shift of the layer below, times log2 e); the two sums
  int boff[KF];
  int vdma;
  unsigned lds0, xch0;
  unsigned long long pbase, rstep;      // the input tensor and the byte step of NW class rows: SGPRs
  unsigned vcol;
  float colm;                           // 1 for a column inside the image
  __amdgpu_buffer_rsrc_t rout, rz;
  // the finished group whose epilogue rides on the current one
  f32x4 fin[HR], got[HR];               // this wave's rows: its own K half, the partner's
  u2v zq[HR];                           // DY: their z (4 channels of one pixel each)
  unsigned pro[HR];                     // their row offsets (kOob: no such row)
  unsigned pvcol;
  float pcolm;
  __amdgpu_buffer_rsrc_t prout;
  // micro-op temporaries
  f2v ty, tu, tn, tr, tw, tz, yv[2];

  __device__ __forceinline__ Walk8(const Conv8Args& a_, const lds_byte* smem_) : a(a_) {
    const int tid = threadIdx.x;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    n = lane & 15;
    g = lane >> 4;
    mblk = wave & 3;
    kh = wave >> 2;
    lds0 = (unsigned)(uintptr_t)smem_;
    xch0 = lds0 + 2u * G::WBUF;
    pbase = reinterpret_cast<unsigned long long>(a.in);
    rstep = (unsigned long long)(((long long)NW * a.dil * a.F) << 7);
    asm volatile("" : "+s"(pbase), "+s"(rstep));
    const u4v* wp = reinterpret_cast<const u4v*>(a.wpk) + ((size_t)mblk * NTAP * 2 + kh) * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) wf[tap] = wp[(size_t)tap * 2 * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = mblk * 16 + g * 4 + r;
      csc[r >> 1][r & 1] = DY ? a.bn2_scale[ch] * kLog2e : a.scale[ch];
      csh[r >> 1][r & 1] = DY ? a.bn2_shift[ch] * kLog2e : a.shift[ch];
      a1[r >> 1][r & 1] = 0.f;
      a2[r >> 1][r & 1] = 0.f;
    }
#pragma unroll
    for (int df = 0; df < KF; ++df) {
      const int p = n + df;
      boff[df] = p * 128 + (((kh * 4 + g) ^ swz(p)) << 4);
    }
    {
      const int px = lane >> 3;
      vdma = (px - PF) * 128 + (((lane & 7) ^ swz(px)) << 4);
    }
#pragma unroll
    for (int j = 0; j < HR; ++j) { fin[j] = got[j] = f32x4{0.f, 0.f, 0.f, 0.f}; pro[j] = kOob; zq[j] = u2v{0u, 0u}; }
    pvcol = kOob;
    pcolm = 0.f;
    prout = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out), 0, 0, 0x00020000);
    rout = prout;
    rz = prout;
    vcol = kOob;
    colm = 0.f;
  }

  __device__ __forceinline__ bool decode(int it, Item& r) const {
    r.strip = it % a.nstrip;
    const int t1 = it / a.nstrip;
    const int seg = t1 % a.nseg;
    const int t2 = t1 / a.nseg;
    r.cls = t2 % a.dil;
    r.b = t2 / a.dil;
    const int nk = r.cls < a.T ? (a.T - r.cls + a.dil - 1) / a.dil : 0;
    r.o0 = seg * a.seg_rows;
    r.o1 = min(nk, r.o0 + a.seg_rows);
    if (r.o0 >= r.o1) return false;
    r.in_end = nk;
    r.ngroups = (r.o1 - r.o0 + R - 1) / R;
    return true;
  }

  // ---- LDS-DMA of a group's window: WIN rows, one row per unit; row rho belongs to wave rho % 8.  A unit is branch-free (a branch
  // splits the group into basic blocks): a wave whose unit has no row left (rho >= WIN) moves zeros into the buffer's scratch row.
  struct Batch {
    unsigned long long pcur;       // the next unit's tensor row
    int in_end, wcur;              // rows of the class; the next unit's class row (any sign)
    bool live;
    unsigned dst0;                 // LDS address of this wave's unit 0 (row `wave` of the buffer); the scratch row: dstx
    unsigned dstx;
    unsigned v0, v1, v2;           // per-lane source offsets: chunk 0; chunk 1; chunk 2 (out of range for the lanes whose pixels no tap reads)
  };
  Batch bt;
  __device__ __forceinline__ void begin(Batch& b, const Item& x, int w_first, int buf) const {
    b.wcur = w_first + wave;
    b.pcur = pbase + (unsigned long long)((((long long)x.b * a.T + x.cls + (long long)b.wcur * a.dil) * a.F) << 7);
    b.in_end = x.in_end; b.live = true;
    b.dst0 = lds0 + (unsigned)(buf * G::WBUF + wave * G::ROWB);
    b.dstx = lds0 + (unsigned)(buf * G::WBUF + G::WIN * G::ROWB);
    b.v0 = (unsigned)(vdma + ((x.strip * STRIP) << 7));
    b.v1 = b.v0 + 1024u;
    b.v2 = (lane >> 3) < G::RWPX - 16 ? b.v1 : kOob;
  }
  template <int J>
  __device__ __forceinline__ void row_unit(Batch& b) const {
    static_assert(G::CPR == 3, "a window row is 3 chunks");
    const bool inwin = (NW * J + NW - 1 < G::WIN) || (wave + NW * J < G::WIN);
    const bool ok = b.live & inwin & ((unsigned)b.wcur < (unsigned)b.in_end);
    const u4v d = {(unsigned)b.pcur, (unsigned)(b.pcur >> 32) & 0xffffu, ok ? (unsigned)a.F * 128u : 0u, 0x00020000u};
    const unsigned dst = inwin ? b.dst0 + (unsigned)(J * NW * G::ROWB) : b.dstx;
    // M0 is not live across this block: nothing else in the kernel uses it
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\t"
                 "s_add_u32 m0, %2, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %1, 0 offen lds\n\t"
                 "buffer_load_dwordx4 %4, %1, 0 offen offset:1024 lds"
                 :: "v"(b.v0), "s"(d), "s"(dst), "v"(b.v1), "v"(b.v2) : "memory", "scc");
    b.pcur += rstep;
    b.wcur += NW;
  }
  template <int... Js>
  __device__ __forceinline__ void all_units(Batch& b, std::integer_sequence<int, Js...>) const { (row_unit<Js>(b), ...); }
  __device__ __forceinline__ void fetch_all(Batch& b) const { all_units(b, std::make_integer_sequence<int, G::UNITS>()); }

  __device__ __forceinline__ void begin_item(const Item& x) {
    const size_t ub = (size_t)x.b * a.T * a.F * 128;
    const unsigned bytes = (unsigned)a.T * a.F * 128;
    rout = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out) + ub, 0, bytes, 0x00020000);
    if (DY) rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.z2)) + ub, 0, bytes, 0x00020000);
    const int col = x.strip * STRIP + n;
    vcol = col < a.F ? (unsigned)(col * 128 + mblk * 32 + g * 8) : kOob;
    colm = col < a.F ? 1.f : 0.f;
  }
  __device__ __forceinline__ unsigned row_offset(const Item& x, int k) const {
    return k < x.o1 ? (unsigned)((x.cls + k * a.dil) * a.F) * 128u : kOob;
  }

  // ---- the deferred epilogue of the previous group: HR rows x (2 channel pairs x NSTAGE stages + 1 store) ----------------------
  static constexpr int NSTAGE = DY ? 9 : ACT == VS_ACT_MISH ? 7 : 1;
  static constexpr int NROW = 2 * NSTAGE + 1;
  static constexpr int NMT = HR * NROW;
  template <int Q>
  __device__ __forceinline__ void micro() {
    constexpr int j = Q / NROW, q = Q % NROW;
    if constexpr (q < 2 * NSTAGE) {
      constexpr int pr = q / NSTAGE, sg = q % NSTAGE;
      if constexpr (!DY) {
        // out = act(acc * scale + shift).  Mish(y) = y n / (n + 2), n = u (u + 2), u = e^y (y clamped at 20): conv_nhwc.hip's stages
        if constexpr (sg == 0) {
          const f2v acc2 = f2v{fin[j][2 * pr], fin[j][2 * pr + 1]} + f2v{got[j][2 * pr], got[j][2 * pr + 1]};
          f2v y = __builtin_elementwise_fma(acc2, csc[pr], csh[pr]);
          if constexpr (ACT == VS_ACT_RELU) y = f2v{fmaxf(y.x, 0.f), fmaxf(y.y, 0.f)};
          yv[pr] = y;
          if constexpr (ACT == VS_ACT_MISH) ty = f2v{fminf(y.x, 20.0f), fminf(y.y, 20.0f)} * kLog2e;
        } else if constexpr (sg == 1) {
          tu.x = __builtin_amdgcn_exp2f(ty.x);
        } else if constexpr (sg == 2) {
          tu.y = __builtin_amdgcn_exp2f(ty.y);
        } else if constexpr (sg == 3) {
          tn = tu * (tu + 2.0f);
          tw = tn + 2.0f;
        } else if constexpr (sg == 4) {
          tr.x = __builtin_amdgcn_rcpf(tw.x);
        } else if constexpr (sg == 5) {
          tr.y = __builtin_amdgcn_rcpf(tw.y);
        } else {
          yv[pr] = yv[pr] * (tn * tr);
        }
      } else {
        // dy = da * act'(y), y = z * scale + shift (times log2 e); Mish'(y) = r (n + 4 y u (u + 1) r), u = e^y, n = u (u + 2), r = 1 / (n + 2)
        constexpr bool mish = ACT == VS_ACT_MISH;
        if constexpr (sg == 0) {
          const unsigned u = zq[j][pr];
          tz = f2v{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
          ty = __builtin_elementwise_fma(tz, csc[pr], csh[pr]);
        } else if constexpr (sg == 1) {
          if constexpr (mish) {
            ty = f2v{fminf(ty.x, 20.0f * kLog2e), fminf(ty.y, 20.0f * kLog2e)};
            tu.x = __builtin_amdgcn_exp2f(ty.x);
          }
        } else if constexpr (sg == 2) {
          if constexpr (mish) tu.y = __builtin_amdgcn_exp2f(ty.y);
        } else if constexpr (sg == 3) {
          if constexpr (mish) {
            tn = tu * (tu + 2.0f);
            tw = tn + 2.0f;
          }
        } else if constexpr (sg == 4) {
          if constexpr (mish) tr.x = __builtin_amdgcn_rcpf(tw.x);
        } else if constexpr (sg == 5) {
          if constexpr (mish) tr.y = __builtin_amdgcn_rcpf(tw.y);
        } else if constexpr (sg == 6) {
          if constexpr (mish) {
            tw = __builtin_elementwise_fma(tu, tu, tu);           // u (u + 1)
            tw = tw * ty;
            tu = tr * (4.0f * 0.69314718055994530942f);           // 4 ln 2: y was scaled by log2(e)
          }
        } else if constexpr (sg == 7) {
          const f2v da = f2v{fin[j][2 * pr], fin[j][2 * pr + 1]} + f2v{got[j][2 * pr], got[j][2 * pr + 1]};
          if constexpr (mish) {
            tn = __builtin_elementwise_fma(tw, tu, tn);
            ty = da * (tr * tn);
          } else if constexpr (ACT == VS_ACT_RELU) {
            ty = f2v{ty.x > 0.f ? da.x : 0.f, ty.y > 0.f ? da.y : 0.f};
          } else {
            ty = da;
          }
        } else {
          const float m = pro[j] != kOob ? pcolm : 0.f;
          const f2v dm = ty * m;
          a1[pr] += dm;
          a2[pr] = __builtin_elementwise_fma(dm, tz, a2[pr]);
          yv[pr] = ty;
        }
      }
    } else {
      if constexpr (STATS && !DY) {
        const float m = pro[j] != kOob ? pcolm : 0.f;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const f2v ym = yv[pr] * m;
          a1[pr] += ym;
          a2[pr] = __builtin_elementwise_fma(ym, yv[pr], a2[pr]);
        }
      }
      const u2v pk = {vs_pack_bf16(yv[0].x, yv[0].y), vs_pack_bf16(yv[1].x, yv[1].y)};
      __builtin_amdgcn_raw_buffer_store_b64(pk, prout, pvcol, pro[j], 0);                               // out of range: dropped
    }
  }
  template <int Q0, int... Ds>
  __device__ __forceinline__ void micros(std::integer_sequence<int, Ds...>) { (micro<Q0 + Ds>(), ...); }

  // ---- one group --------------------------------------------------------------------------------------------------------
  // fragment reads run PD steps ahead of their MFMAs (VS_CONV8_PD: the A/B of profiles/r05_conv8.md)
#ifndef VS_CONV8_PD
#define VS_CONV8_PD 2
#endif
  static constexpr int PD = DY ? 2 : VS_CONV8_PD;
  template <int RV>
  struct GroupState {
    f32x4 acc[RV];
    u4v bq[PD + 1];                    // fragments in flight: this step's and the next PD
    u2v zn[HR];                        // DY: z of the rows this wave will finish (used in the NEXT group)
    unsigned vb[KF];
  };
  template <int RV> static constexpr int r_lo(int i) { return i - (KT - 1) > 0 ? i - (KT - 1) : 0; }
  template <int RV> static constexpr int r_hi(int i) { return i < RV - 1 ? i : RV - 1; }
  template <int RV>
  __device__ __forceinline__ u4v frag(const GroupState<RV>& st, int ps) const {
    const int df = ps % KF, i = ps / KF;
    return *(lds_u4v*)(uintptr_t)(st.vb[df] + (unsigned)(i * G::ROWB));
  }

  // Step PS = (window row i, tap column df): the fragment of step PS + 2 is read, then ALL MFMAs of (i, df) -- rows lo .. lo + nm - 1
  // -- issue as one asm statement (the first MFMA of an output row, at (i = r, df = 0), starts from the literal 0).  Behind the
  // block: the next group's window (one row unit per step at the group's start) and the previous group's epilogue micro-ops.
  template <int RV>
  static constexpr int NPS = (RV + H) * KF;
  static constexpr int MS0 = 1;        // first step that carries epilogue micro-ops (the partner's part needs its LDS latency)
  template <int RV, int PS, int... Rs>
  __device__ __forceinline__ void pblock(GroupState<RV>& st, std::integer_sequence<int, Rs...>) {
    constexpr int df = PS % KF, i = PS / KF;
    constexpr int lo = r_lo<RV>(i), nm = sizeof...(Rs);
    const u4v* w[nm] = {&wf[(i - (lo + Rs)) * KF + df]...};
    mfma_block<nm, (df == 0 && i < RV)>(&st.acc[lo], w, st.bq[PS % (PD + 1)]);
  }
  template <int RV, int PS>
  __device__ __forceinline__ void pstep(GroupState<RV>& st) {
    constexpr int i = PS / KF;
    if constexpr (PS + PD < NPS<RV>) st.bq[(PS + PD) % (PD + 1)] = frag<RV>(st, PS + PD);
    __builtin_amdgcn_sched_barrier(0);
    pblock<RV, PS>(st, std::make_integer_sequence<int, r_hi<RV>(i) - r_lo<RV>(i) + 1>());
    constexpr int SPD = NPS<RV> >= 2 * G::UNITS ? 2 : 1;
    static_assert(NPS<RV> >= SPD * G::UNITS, "every DMA unit needs a step");
    if constexpr (PS % SPD == 0 && PS / SPD < G::UNITS) row_unit<PS / SPD>(bt);      // in order: the cursor advances
    constexpr int per = (NMT + NPS<RV> - MS0 - 1) / (NPS<RV> - MS0);
    if constexpr (PS >= MS0) {
      constexpr int m0 = (PS - MS0) * per;
      micros<m0>(std::make_integer_sequence<int, (m0 < NMT ? (NMT - m0 < per ? NMT - m0 : per) : 0)>());
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  template <int RV, int... PSs>
  __device__ __forceinline__ void psteps(GroupState<RV>& st, std::integer_sequence<int, PSs...>) { (pstep<RV, PSs>(st), ...); }

  // Output rows ro .. ro + RV - 1 of item x from window buffer `buf`; `par`: the exchange area of this group.  On return the other K
  // half's share of the accumulators is on its way through LDS, this wave's share sits in fin[] (still without the partner's part:
  // take_partner() after the next barrier), zq[] holds its rows' z (DY) and pro / pvcol / prout describe where its rows go.
  template <int RV>
  __device__ __forceinline__ void group(const Item& x, int ro, int buf, int par) {
    GroupState<RV> st;
    constexpr int HV = RV / 2;
#pragma unroll
    for (int df = 0; df < KF; ++df) st.vb[df] = lds0 + (unsigned)(buf * G::WBUF + boff[df]);
#pragma unroll
    for (int q = 0; q < PD; ++q)
      if (q < NPS<RV>) st.bq[q] = frag<RV>(st, q);
    if constexpr (DY) {
#pragma unroll
      for (int j = 0; j < HR; ++j)
        st.zn[j] = j < HV ? __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(rz, vcol, row_offset(x, ro + kh * HV + j), 0)) : u2v{0u, 0u};
    }
    psteps<RV>(st, std::make_integer_sequence<int, NPS<RV>>());
    // hand-over: K half 0 finishes rows [0, RV / 2), K half 1 rows [RV / 2, RV)
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // the last MFMAs' results (the compiler does not see MFMAs in the asm statements)
    const unsigned xw = xch0 + (unsigned)(par * NW * G::XCH + (wave ^ 4) * G::XCH + lane * 16);      // the partner's area
#pragma unroll
    for (int j = 0; j < HV; ++j) {
      const f32x4 give = kh ? st.acc[j] : st.acc[HV + j];
      const f32x4 keep = kh ? st.acc[HV + j] : st.acc[j];
      *(lds_u4v_rw*)(uintptr_t)(xw + (unsigned)(j * 1024)) = __builtin_bit_cast(u4v, give);
      fin[j] = keep;
    }
#pragma unroll
    for (int j = 0; j < HR; ++j) {
      if (j >= HV) fin[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      pro[j] = j < HV ? row_offset(x, ro + kh * HV + j) : kOob;
      if constexpr (DY) zq[j] = st.zn[j];
    }
    pvcol = vcol;
    pcolm = colm;
    prout = rout;
  }

  // after the barrier that follows a group: fetch the partner's part of this wave's rows (rows the group did not have: stale LDS
  // that ends in a dropped store and a zero statistics mask)
  __device__ __forceinline__ void take_partner(int par) {
    const unsigned xr = xch0 + (unsigned)(par * NW * G::XCH + wave * G::XCH + lane * 16);
#pragma unroll
    for (int j = 0; j < HR; ++j) got[j] = __builtin_bit_cast(f32x4, *(lds_u4v*)(uintptr_t)(xr + (unsigned)(j * 1024)));
    if (pro[HR - 1] == kOob) {       // a short group (wave-uniform, rare): rows it did not have hold stale bits in the partner's area -- any
#pragma unroll                       // bit pattern, NaN included, which a zero mask would not keep out of the sums
      for (int j = 0; j < HR; ++j)
        if (pro[j] == kOob) fin[j] = got[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __device__ __forceinline__ void flush_epilogue() { micros<0>(std::make_integer_sequence<int, NMT>()); }

  __device__ __forceinline__ void flush_stats() {
    if (!(STATS || DY) || a.bn_stats == nullptr) return;
    float s1[4], s2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[r] = a1[r >> 1][r & 1]; s2[r] = a2[r >> 1][r & 1]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        s1[r] += __shfl_xor(s1[r], o, 64);
        s2[r] += __shfl_xor(s2[r], o, 64);
      }
    }
    if (n == 0) {
      double* dst = a.bn_stats + (size_t)(blockIdx.x % VS_BN_STAT_SLOTS) * 128 + (mblk * 16 + g * 4) * 2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double t1 = (double)s1[r], t2 = (double)s2[r];
        if (DY) {                      // sum dy * xhat = invstd * (sum dy * z - mean * sum dy)
          const int ch = mblk * 16 + g * 4 + r;
          t2 = (double)a.bn2_invstd[ch] * (t2 - (double)a.bn2_mean[ch] * t1);
        }
        atomicAdd(dst + 2 * r, t1);
        atomicAdd(dst + 2 * r + 1, t2);
      }
    }
  }
};

// PROBE (VS_ABLATION builds, VS_OPT_SPLITCONV_ABL = 32, the fused-statistics instance): s_memtime around the group boundary; the sums
// over all waves of {total cycles, cycles in s_waitcnt vmcnt(0), cycles in lgkmcnt(0) + s_barrier, groups} are ADDED to bn_stats[0..3]
// (the statistics of such a launch are garbage): tools/nhwc_micro.py prints them.
template <int KT, int KF, int ACT, bool STATS, bool DY, bool PROBE = false>
__global__ __launch_bounds__(64 * NW)
void nhwc_conv8_kernel(Conv8Args a) {
  using G = Geo8<KT, KF>;
  constexpr int R = G::R;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS_BYTES];
  Walk8<KT, KF, ACT, STATS, DY> wk(a, (const lds_byte*)smem);

  Item pf;
  int pf_it = (int)blockIdx.x, pf_g = 0;
  bool pf_live = false;
  auto pf_seek = [&]() {
    pf_live = false;
    while (pf_it < a.n_items) {
      if (wk.decode(pf_it, pf)) { pf_g = 0; pf_live = true; return; }
      pf_it += (int)gridDim.x;
    }
  };
  int pbuf = 0;
  auto pf_begin = [&]() {
    wk.begin(wk.bt, pf, pf.o0 + pf_g * R - G::P, pbuf);
    pbuf ^= 1;
    if (++pf_g >= pf.ngroups) { pf_it += (int)gridDim.x; pf_seek(); }
  };
  wk.bt.live = false;
  wk.bt.pcur = 0; wk.bt.wcur = 0; wk.bt.in_end = 0; wk.bt.dst0 = wk.lds0; wk.bt.dstx = wk.lds0 + (unsigned)(G::WIN * G::ROWB); wk.bt.v0 = wk.bt.v1 = wk.bt.v2 = 0;
  pf_seek();
  if (pf_live) { pf_begin(); wk.fetch_all(wk.bt); }

  unsigned long long t_vm = 0, t_bar = 0, t_grp = 0;
  const unsigned long long t_start = PROBE ? __builtin_amdgcn_s_memtime() : 0ull;
  Item cur;
  int cbuf = 0, par = 0;
  bool pending = false;            // a finished group waits for its partner's part and its epilogue
  for (int it = (int)blockIdx.x; it < a.n_items; it += (int)gridDim.x) {
    if (!wk.decode(it, cur)) continue;
    wk.begin_item(cur);
    for (int gidx = 0; gidx < cur.ngroups; ++gidx) {
      if constexpr (PROBE) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_vm += t1 - t0; t_bar += t2 - t1; t_grp += 1;
      } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's window rows have landed, its hand-over is written
      __builtin_amdgcn_s_barrier();
      }
      if (pending) wk.take_partner(par ^ 1);
      wk.bt.live = false;
      if (pf_live) pf_begin();
      else {                                                            // nothing left to fetch: the units move zeros into the idle buffer
        wk.bt.dst0 = wk.lds0 + (unsigned)((cbuf ^ 1) * G::WBUF + wk.wave * G::ROWB);
        wk.bt.dstx = wk.lds0 + (unsigned)((cbuf ^ 1) * G::WBUF + G::WIN * G::ROWB);
      }
      const int ro = cur.o0 + gidx * R;
      const int left = cur.o1 - ro;                           // > 0
      if (left > 4) wk.template group<6>(cur, ro, cbuf, par);
      else if (left > 2) wk.template group<4>(cur, ro, cbuf, par);
      else wk.template group<2>(cur, ro, cbuf, par);
      cbuf ^= 1;
      par ^= 1;
      pending = true;
    }
  }
  if (pending) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    wk.take_partner(par ^ 1);
    wk.flush_epilogue();
  }
  wk.flush_stats();
  if constexpr (PROBE) {
    if ((threadIdx.x & 63) == 0 && a.bn_stats) {
      atomicAdd(a.bn_stats + 0, (double)(__builtin_amdgcn_s_memtime() - t_start));
      atomicAdd(a.bn_stats + 1, (double)t_vm);
      atomicAdd(a.bn_stats + 2, (double)t_bar);
      atomicAdd(a.bn_stats + 3, (double)t_grp);
    }
  }
}

int conv8_num_cus() {
  static int cus[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!cus[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev] = v;
  }
  return cus[dev];
}

template <int KT, int KF>
int launch_conv8(Conv8Args a, int act, hipStream_t stream) {
  constexpr int R = Geo8<KT, KF>::R;
  const int nk_max = (a.T + a.dil - 1) / a.dil;
  const long long base_items = (long long)a.B * a.dil * a.nstrip;
  // enough items for the persistent grid to balance (>= ~32 per workgroup), segments no shorter than 16 rows
  int nseg = (int)((8192 + base_items - 1) / base_items);
  if (nseg > nk_max / 16) nseg = nk_max / 16;
  if (nseg < 1) nseg = 1;
  int seg_rows = (nk_max + nseg - 1) / nseg;
  seg_rows = (seg_rows + R - 1) / R * R;
  nseg = (nk_max + seg_rows - 1) / seg_rows;
  a.nseg = nseg;
  a.seg_rows = seg_rows;
  const long long n_items = base_items * nseg;
  VS_REQUIRE(n_items < (1LL << 30), "nhwc conv8: too many work items");
  a.n_items = (int)n_items;
  const int cus = conv8_num_cus();
  const dim3 grid((unsigned)(n_items < cus ? n_items : cus)), block(64 * NW);
  const bool stats = a.bn_stats != nullptr;
#define VS_C8_LAUNCH(A, S, D) hipLaunchKernelGGL((nhwc_conv8_kernel<KT, KF, A, S, D>), grid, block, 0, stream, a)
  if (a.z2) {
    VS_REQUIRE(stats && a.bn2_scale && a.bn2_shift && a.bn2_mean && a.bn2_invstd, "nhwc conv8: the dy epilogue needs statistics slots and BatchNorm constants");
    if (act == VS_ACT_MISH) VS_C8_LAUNCH(VS_ACT_MISH, false, true);
    else if (act == VS_ACT_RELU) VS_C8_LAUNCH(VS_ACT_RELU, false, true);
    else VS_REQUIRE(false, "nhwc conv8: dy epilogue for activation %d", act);
  } else if (stats) {
    VS_REQUIRE(act == VS_ACT_NONE, "nhwc conv8: fused statistics go with no activation");
#ifdef VS_ABLATION
    if (vs_opt(VS_OPT_SPLITCONV_ABL) == 32) hipLaunchKernelGGL((nhwc_conv8_kernel<KT, KF, VS_ACT_NONE, true, false, true>), grid, block, 0, stream, a);
    else
#endif
    VS_C8_LAUNCH(VS_ACT_NONE, true, false);
  } else if (act == VS_ACT_NONE) VS_C8_LAUNCH(VS_ACT_NONE, false, false);
  else if (act == VS_ACT_MISH) VS_C8_LAUNCH(VS_ACT_MISH, false, false);
  else if (act == VS_ACT_RELU) VS_C8_LAUNCH(VS_ACT_RELU, false, false);
  else VS_REQUIRE(false, "nhwc conv8: unsupported activation %d", act);
#undef VS_C8_LAUNCH
  VS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// The 5x5 layers of vs_nhwc_conv_impl / vs_nhwc_conv_dy_impl (conv_nhwc.hip) on the eight-wave kernel: same arguments, same packed
// weights; z == NULL: forward / plain data gradient, else the dy form.
int vs_nhwc_conv8_impl(const void* in, const void* packed, const float* scale, const float* shift, void* out, double* bn_stats,
                       const void* z, const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd,
                       int B, int T, int F, int dil, int act, hipStream_t stream) {
  Conv8Args a{reinterpret_cast<const unsigned short*>(in), reinterpret_cast<const unsigned short*>(packed), scale, shift,
              reinterpret_cast<unsigned short*>(out), bn_stats, reinterpret_cast<const unsigned short*>(z), bn_scale, bn_shift, bn_mean, bn_invstd,
              B, T, F, dil, (F + STRIP - 1) / STRIP, 1, 0, 0};
  return launch_conv8<5, 5>(a, act, stream);
}
