// 64 -> 64 convolution of the conv stack, FORWARD in the fp32-class arithmetic (VS_MATH_F16X3, BASELINE configs[1]):
// cnn2 (7x1) and cnn3..cnn7 (5x5, time dilation 1..16) of models/voicesplit/model.py:21-48 with eval-mode BatchNorm and the
// activation in the epilogue.  Round 4: conv_nhwc.hip's design (channels-last operands by LDS-DMA, weights in registers, one
// straight-line block per group of output rows, v_mfma_f32_16x16x32) carried over to the split-f16 arithmetic of
// conv_f16x3.hip:  x = x_hi + x_lo, w = w_hi + w_lo (f16 each, after a power-of-two scale),
//     x w  ~=  x_hi w_hi + x_lo w_hi + x_hi w_lo            (fp32 accumulate; the dropped x_lo w_lo is 2^-22 of the product).
//
// Layout.  An activation tensor is TWO channels-last f16 planes [B][T][F][64] (hi and lo: the fp32 tensor's footprint), the
// values scaled by a power of two s:  x = (hi + lo) / s.  The producer's epilogue writes both planes, so it must know s before
// the tensor exists: s comes from an upper BOUND of max |y| that the layer's plan kernel (vs_nhwc_f16x3_plan_impl) derives from
// the tracked max |x| of the layer's input and the weights' row L1 norms -- loose by the usual sqrt(K), which costs nothing:
// with the bound in [2^14, 2^15) every value down to 2^-12 of it keeps 22 bits and smaller ones an absolute error of 2^-25 s.
//
// What differs from conv_nhwc.hip, and why:
//  * weights in registers are 2 planes x 25 taps: 64 co would need 400 VGPRs per wave.  A WORKGROUP therefore owns 32 output
//    channels (a launch constant: workgroups w and w + 8 -- same XCD, same L2 -- walk the same tiles for the two halves) and its
//    4 waves are 2 blocks of 16 channels x 2 HALVES OF K (input channels 0-31 / 32-63): 200 registers of weights per wave, kept
//    in AGPRs with the accumulators (the MFMAs are asm statements with `a` operands, one statement per window row and tap column:
//    split_mfma_blocks.inc);
//  * the two K halves of a block are summed across waves once per group of R output rows: each wave of a pair hands the other
//    half of its accumulators through LDS (16 bytes per lane and row) and finishes the other half of the rows;
//  * that epilogue (BatchNorm scale / shift, activation, |max|, hi / lo split, two 8-byte stores per pixel) is DEFERRED: cut into
//    micro-ops placed between the MFMA blocks of the next group, spread over its first part (measured: a wave alone on its SIMD
//    does not overlap its VALU work with its own MFMAs -- profiles/r04_split_conv.md -- so the point of the spreading is that the
//    stores have retired and the partner's LDS data has arrived, not that the arithmetic is free);
//  * two planes double the LDS per window row: strips are 16 columns and groups 6 rows (5x5) / 8 rows (7x1); two window
//    buffers + two exchange areas = 144 KiB.
// A fragment read (hi or lo plane, this wave's K half) feeds <= KT taps x {w_hi, w_lo} (hi plane) or {w_hi} (lo plane):
// 2 reads per <= 15 MFMAs.
#include <stdlib.h>

#include <utility>

#include "vs_internal.h"

namespace {

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
constexpr float kLog2e = 1.44269504088896340736f;
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) const u4v lds_u4v;
typedef __attribute__((address_space(3))) u4v lds_u4v_rw;

#include "split_mfma_blocks.inc"

constexpr int STRIP = 16;           // output columns per strip = one MFMA column block

struct SplitConvArgs {
  const unsigned short* in_hi;      // [B][T][F][64] f16, values x * s_x
  const unsigned short* in_lo;      // [B][T][F][64] f16, the remainder
  const unsigned short* wpk;        // [2 co halves][4 waves][taps][2 planes][64 lanes][8] f16 (vs_nhwc_f16x3_pack_impl)
  const float* scale;               // [64]  y = act(acc * scale + shift); scale carries 1 / (s_x s_w)
  const float* shift;               // [64]
  const float* out_scale2;          // device {s_y, 1 / s_y}: the planes written hold y * s_y
  unsigned short* out_hi;
  unsigned short* out_lo;
  unsigned* amax_out;               // [VS_AMAX_SLOTS] running max |y| (bits), or NULL
  int B, T, F, dil;
  int nstrip, nseg, seg_rows, n_tiles;
};

__device__ __forceinline__ int swz(int p) { return ((p >> 1) & 3) << 1; }      // conv_nhwc.hip: conflict-free for every tap column

template <int KT, int KF>
struct SGeo {
  static constexpr int R = KT == 5 ? 6 : 8;              // output rows per group
  static constexpr int P = KT / 2, PF = KF / 2, H = KT - 1;
  static constexpr int NTAP = KT * KF;
  static constexpr int RWPX = STRIP + 2 * PF;            // staged pixels per row (20 / 16)
  static constexpr int CPR = (RWPX * 8 + 63) / 64;       // 1 KiB DMA chunks per row and plane (3 / 2)
  static constexpr int ROWB = CPR * 1024;
  static constexpr int WIN = R + H;                      // input rows of a group (10 / 14)
  static constexpr int WBUF = WIN * 2 * ROWB;            // [row][plane]
  static constexpr int XCH = (R / 2) * 1024;             // what one wave receives per group: R / 2 rows x 64 lanes x 16 bytes
  static constexpr int LDS_BYTES = 2 * WBUF + 2 * 4 * XCH;
  static constexpr int UNITS = (2 * WIN + 3) / 4;        // DMA (row, plane) units of a wave per group
};

struct Item { int b, cls, strip, o0, o1, in_end, ngroups; };

// ABL (instances built with -DVS_ABLATION only, selected by VS_OPT_ABLATION): timing ablations (results invalid), tools/split_conv_micro.py: 1 = no window DMA, 2 = no epilogue micro-ops, 4 = no hand-over,
// 8 = epilogue without its stores, 16 = the stores without the arithmetic, 64 = the H rows two groups share are not fetched again (what a ring buffer would save), 32 = valid results + s_memtime probes of the group boundary
// (written over amax_out: tools/split_conv_micro.py prints them)
template <int KT, int KF, int ACT, int ABL = 0>
struct SplitWalk {
  using G = SGeo<KT, KF>;
  static constexpr int R = G::R, P = G::P, PF = G::PF, H = G::H, NTAP = G::NTAP, HR = G::R / 2;
  static constexpr unsigned kOob = 0x7FFFFFF0u;

  const SplitConvArgs& a;
  int lane, wave, n, g, mblk, kh, cohalf;
  h8v wfh[NTAP], wfl[NTAP];
  f2v csc[2], csh[2], csl[2], chl[2];   // epilogue constants times s_y (the planes' scale rides in them); times log2(e) (Mish's exponent)
  float am;                             // running max |y * s_y|
  int boff[KF];
  int vdma;
  unsigned lds0, xch0;
  unsigned long long pbase, rstep;             // this wave's input plane and the byte step of two class rows: kept in SGPRs (a kernarg
                                               // load per DMA unit stalled the MFMA stream on lgkmcnt(0))
  unsigned vcol;
  __amdgpu_buffer_rsrc_t rout_hi, rout_lo;
  // the finished group whose epilogue rides on the current one
  f32x4 fin[HR], got[HR];           // this wave's R / 2 rows: its own K half, the partner's (added in the first micro-op: LDS latency hidden)
  unsigned pro[HR];                 // their row offsets (kOob: no such row)
  unsigned pvcol;
  __amdgpu_buffer_rsrc_t prout_hi, prout_lo;
  // micro-op temporaries
  f2v ty, tu, tn, tr, tw, yv;
  unsigned hp[2], lp[2];

  __device__ __forceinline__ SplitWalk(const SplitConvArgs& a_, const lds_byte* smem_) : a(a_) {
    const int tid = threadIdx.x;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    n = lane & 15;
    g = lane >> 4;
    mblk = wave & 1;
    kh = wave >> 1;
    cohalf = ((int)blockIdx.x >> 3) & 1;
    lds0 = (unsigned)(uintptr_t)smem_;
    xch0 = lds0 + 2u * G::WBUF;
    pbase = reinterpret_cast<unsigned long long>((wave & 1) ? a.in_lo : a.in_hi);
    rstep = (unsigned long long)(((long long)2 * a.dil * a.F) << 7);
    asm volatile("" : "+s"(pbase), "+s"(rstep));
    const u4v* wp = reinterpret_cast<const u4v*>(a.wpk) + ((size_t)(cohalf * 4 + wave) * NTAP * 2) * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      wfh[tap] = __builtin_bit_cast(h8v, wp[(tap * 2 + 0) * 64]);
      wfl[tap] = __builtin_bit_cast(h8v, wp[(tap * 2 + 1) * 64]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = cohalf * 32 + mblk * 16 + g * 4 + r;
      const float sy = a.out_scale2[0];
      csc[r >> 1][r & 1] = a.scale[ch] * sy;
      csh[r >> 1][r & 1] = a.shift[ch] * sy;
      csl[r >> 1][r & 1] = a.scale[ch] * kLog2e;
      chl[r >> 1][r & 1] = a.shift[ch] * kLog2e;
    }
    am = 0.f;
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
    for (int j = 0; j < HR; ++j) { fin[j] = got[j] = f32x4{0.f, 0.f, 0.f, 0.f}; pro[j] = kOob; }
    pvcol = kOob;
    prout_hi = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out_hi), 0, 0, 0x00020000);
    prout_lo = prout_hi;
    rout_hi = prout_hi;
    rout_lo = prout_hi;
    vcol = kOob;
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

  // ---- LDS-DMA of a group's window: WIN rows x 2 planes, one (row, plane) per unit; unit u = row * 2 + plane belongs to wave u % 4,
  // i.e. wave w moves plane w & 1 of rows (w >> 1) + 2 J.  Everything a unit needs beyond two scalar adds, a compare and a select
  // is prepared per launch (pbase, rstep) or per group (begin): a unit is ~11 instructions in the MFMA stream.
  struct Batch {
    unsigned long long pcur;       // the next unit's tensor row
    int in_end, wcur;              // rows of the class; the next unit's class row (any sign)
    bool live;
    unsigned dst0;                 // LDS address of unit 0; unit J: + J * 4 * ROWB
    unsigned v0, v1, v2;           // per-lane source offsets: chunk 0; chunk 1; chunk 2 by immediate from v2 = v1 or, for the lanes whose
  };                               // pixels (20..23 of the row image) no tap reads, out of range: zeros, no memory traffic
  Batch bt;
  __device__ __forceinline__ void begin(Batch& b, const Item& x, int w_first, int buf) const {
    b.wcur = w_first + (wave >> 1);
    b.pcur = pbase + (unsigned long long)((((long long)x.b * a.T + x.cls + (long long)b.wcur * a.dil) * a.F) << 7);
    b.in_end = x.in_end; b.live = true;
    b.dst0 = lds0 + (unsigned)(buf * G::WBUF + wave * G::ROWB);
    b.v0 = (unsigned)(vdma + ((x.strip * STRIP) << 7));
    b.v1 = b.v0 + 1024u;
    b.v2 = (lane >> 3) < G::RWPX - 16 ? b.v1 : kOob;
  }
  template <int J>
  __device__ __forceinline__ void row_unit(Batch& b) const {
    // branch-free (a branch here splits the group into basic blocks, and the compiler sinks the epilogue's arithmetic across them
    // into clusters): with nothing left to fetch the unit moves zeros into the idle window buffer
    static_assert(4 * G::UNITS == 2 * G::WIN, "every wave has UNITS units");
    if constexpr (ABL & 1) return;
    if constexpr ((ABL & 64) && 2 * J < H) { b.pcur += rstep; b.wcur += 2; return; }      // what a ring of rows would save: the H shared rows are not fetched again
    const bool ok = b.live & ((unsigned)b.wcur < (unsigned)b.in_end);
    const u4v d = {(unsigned)b.pcur, (unsigned)(b.pcur >> 32), ok ? (unsigned)a.F * 128u : 0u, 0x00020000u};
    const unsigned dst = b.dst0 + (unsigned)(J * 4 * G::ROWB);
    static_assert(G::CPR == 2 || G::CPR == 3, "a window row is 2 or 3 chunks");
    // M0 is not live across this block: nothing else in the kernel uses it (LDS instructions of gfx9 take no M0)
    if (G::CPR == 3)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\t"
                   "s_add_u32 m0, %2, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %1, 0 offen lds\n\t"
                   "buffer_load_dwordx4 %4, %1, 0 offen offset:1024 lds"
                   :: "v"(b.v0), "s"(d), "s"(dst), "v"(b.v1), "v"(b.v2) : "memory", "scc");
    else
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\t"
                   "s_add_u32 m0, %2, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %1, 0 offen lds"
                   :: "v"(b.v0), "s"(d), "s"(dst), "v"(b.v1) : "memory", "scc");
    b.pcur += rstep;
    b.wcur += 2;
  }
  template <int... Js>
  __device__ __forceinline__ void all_units(Batch& b, std::integer_sequence<int, Js...>) const { (row_unit<Js>(b), ...); }
  __device__ __forceinline__ void fetch_all(Batch& b) const { all_units(b, std::make_integer_sequence<int, G::UNITS>()); }

  __device__ __forceinline__ void begin_item(const Item& x) {
    const size_t ub = (size_t)x.b * a.T * a.F * 128;
    const unsigned bytes = (unsigned)a.T * a.F * 128;
    rout_hi = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out_hi) + ub, 0, bytes, 0x00020000);
    rout_lo = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out_lo) + ub, 0, bytes, 0x00020000);
    const int col = x.strip * STRIP + n;
    vcol = col < a.F ? (unsigned)(col * 128 + cohalf * 64 + mblk * 32 + g * 8) : kOob;
  }
  __device__ __forceinline__ unsigned row_offset(const Item& x, int k) const {
    return k < x.o1 ? (unsigned)((x.cls + k * a.dil) * a.F) * 128u : kOob;
  }

  // ---- the deferred epilogue of the previous group: R / 2 rows x (2 channel pairs x NSTAGE stages + 2 stores) --------------
  static constexpr int NSTAGE = (ACT == VS_ACT_MISH ? 8 : 2) + 3;
  static constexpr int NROW = 2 * NSTAGE + 2;
  static constexpr int NMT = HR * NROW;
  // a micro-op's results pass through an opaque volatile statement: volatile statements keep their order, and the MFMAs are
  // volatile asm -- the arithmetic stays between the two MFMAs it was written between instead of sinking to its first use
  template <typename V> static __device__ __forceinline__ void pin(V& v) {
#ifdef VS_SPLITCONV_PIN
    asm volatile("" : "+v"(v));
#else
    (void)v;      // the group is one basic block since the DMA units lost their branches: the sched_barriers hold the micro-ops in place, and an
#endif            // opaque statement costs a hazard s_nop each (45 per group)
  }
  template <int Q>
  __device__ __forceinline__ void micro() {
    if constexpr (ABL & 2) return;
    constexpr int j = Q / NROW, q = Q % NROW;
    if constexpr ((ABL & 8) && q >= 2 * NSTAGE) return;        // no stores
    if constexpr ((ABL & 16) && q < 2 * NSTAGE) return;        // stores only
    if constexpr (q < 2 * NSTAGE) {
      constexpr int pr = q / NSTAGE, sg = q % NSTAGE;
      constexpr int base = ACT == VS_ACT_MISH ? 8 : 2;
      if constexpr (sg == 0) {
        yv = f2v{fin[j][2 * pr], fin[j][2 * pr + 1]} + f2v{got[j][2 * pr], got[j][2 * pr + 1]};
        pin(yv);
      } else if constexpr (sg == 1) {
        // y * s_y = act(z) * s_y with the scale folded into the constants (act is positively homogeneous only for ReLU: Mish takes
        // its exponent from the unscaled y, a second fma)
        if constexpr (ACT == VS_ACT_MISH) {
          ty = __builtin_elementwise_fma(yv, csl[pr], chl[pr]);
          ty = f2v{fminf(ty.x, 20.0f * kLog2e), fminf(ty.y, 20.0f * kLog2e)};
          pin(ty);
        }
        f2v y = __builtin_elementwise_fma(yv, csc[pr], csh[pr]);
        if constexpr (ACT == VS_ACT_RELU) y = f2v{fmaxf(y.x, 0.f), fmaxf(y.y, 0.f)};
        yv = y;
        pin(yv);
      } else if constexpr (ACT == VS_ACT_MISH && sg == 2) {
        tu.x = __builtin_amdgcn_exp2f(ty.x);
        pin(tu);
      } else if constexpr (ACT == VS_ACT_MISH && sg == 3) {
        tu.y = __builtin_amdgcn_exp2f(ty.y);
        pin(tu);
      } else if constexpr (ACT == VS_ACT_MISH && sg == 4) {
        tn = tu * (tu + 2.0f);
        tw = tn + 2.0f;
        pin(tn); pin(tw);
      } else if constexpr (ACT == VS_ACT_MISH && sg == 5) {
        tr.x = __builtin_amdgcn_rcpf(tw.x);
        pin(tr);
      } else if constexpr (ACT == VS_ACT_MISH && sg == 6) {
        tr.y = __builtin_amdgcn_rcpf(tw.y);
        pin(tr);
      } else if constexpr (ACT == VS_ACT_MISH && sg == 7) {
        yv = yv * (tn * tr);
        pin(yv);
      } else if constexpr (sg == base) {
        am = fmaxf(am, fmaxf(fabsf(yv.x), fabsf(yv.y)));
        hp[pr] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(yv.x, yv.y));
        pin(hp[pr]); pin(am);
      } else if constexpr (sg == base + 1) {
        const h2v h = __builtin_bit_cast(h2v, hp[pr]);
        yv = yv - f2v{(float)h[0], (float)h[1]};
        pin(yv);
      } else {
        lp[pr] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(yv.x, yv.y));
        pin(lp[pr]);
      }
    } else if constexpr (q == 2 * NSTAGE) {
      __builtin_amdgcn_raw_buffer_store_b64(u2v{hp[0], hp[1]}, prout_hi, pvcol, pro[j], 0);       // out of range: dropped
    } else {
      __builtin_amdgcn_raw_buffer_store_b64(u2v{lp[0], lp[1]}, prout_lo, pvcol, pro[j], 0);
    }
  }
  template <int Q0, int... Ds>
  __device__ __forceinline__ void micros(std::integer_sequence<int, Ds...>) { (micro<Q0 + Ds>(), ...); }

  // ---- one group --------------------------------------------------------------------------------------------------------
  template <int RV>
  struct GroupState {
    f32x4 acc[RV];
    h8v bq[6];                         // fragments in flight: the two planes of this pair-step and of the next two
    unsigned vb[KF];
  };
  // window row i feeds output rows r_lo(i) .. r_hi(i) of the group (tap row i - r)
  template <int RV> static constexpr int r_lo(int i) { return i - (KT - 1) > 0 ? i - (KT - 1) : 0; }
  template <int RV> static constexpr int r_hi(int i) { return i < RV - 1 ? i : RV - 1; }
  template <int RV>
  __device__ __forceinline__ h8v frag(const GroupState<RV>& st, int si) const {
    const int pl = si % 2, df = (si / 2) % KF, i = si / (2 * KF);
    return __builtin_bit_cast(h8v, *(lds_u4v*)(uintptr_t)(st.vb[df] + (unsigned)((i * 2 + pl) * G::ROWB)));
  }

  // Pair-step PS = (window row i, tap column df): the two planes' fragments of pair-step PS + 2 are read, then ALL MFMAs of
  // (i, df) -- rows lo .. lo + nm - 1, three products each -- issue as one asm statement (split_mfma_blocks.inc: weights and
  // accumulators in AGPRs; the first MFMA of an output row, at (i = r, df = 0), starts from the literal 0).  As a builtin the
  // compiler kept the weights in VGPRs and shuttled the epilogue's state through v_accvgpr_read / write in the MFMA stream; one
  // asm statement per MFMA got an s_nop between most of them and a s_waitcnt per fragment.  Behind the block: the next
  // group's window, one (row, plane) DMA unit per pair-step, and the previous group's epilogue micro-ops -- from pair-step 1 on
  // (the partner's part needs its LDS latency) and done early (its stores must have retired by the next group's vmcnt(0)).
  template <int RV>
  static constexpr int NPS = (RV + H) * KF;
  template <int RV, int PS, int... Rs>
  __device__ __forceinline__ void pblock(GroupState<RV>& st, std::integer_sequence<int, Rs...>) {
    constexpr int df = PS % KF, i = PS / KF;
    constexpr int lo = r_lo<RV>(i), nm = sizeof...(Rs);
    const h8v* wh[nm] = {&wfh[(i - (lo + Rs)) * KF + df]...};
    const h8v* wl[nm] = {&wfl[(i - (lo + Rs)) * KF + df]...};
    split_mfma_block<nm, (df == 0 && i < RV)>(&st.acc[lo], wh, wl, st.bq[(2 * PS) % 6], st.bq[(2 * PS + 1) % 6]);
  }
  template <int RV, int PS>
  __device__ __forceinline__ void pstep(GroupState<RV>& st) {
    constexpr int i = PS / KF;
    if constexpr (PS + 2 < NPS<RV>) {
      st.bq[(2 * PS + 4) % 6] = frag<RV>(st, 2 * PS + 4);
      st.bq[(2 * PS + 5) % 6] = frag<RV>(st, 2 * PS + 5);
    }
    __builtin_amdgcn_sched_barrier(0);
    pblock<RV, PS>(st, std::make_integer_sequence<int, r_hi<RV>(i) - r_lo<RV>(i) + 1>());
    // 7x1: a group is 14 pair-steps = 1.5 us, shorter than a DMA round trip under load: its units go out back to back at the group's start
    // (spread over the group, the last one cost 7 % of the kernel in s_waitcnt vmcnt(0): s_memtime probes, profiles/r04_split_conv.md)
    constexpr int SPD = (KF > 1 && NPS<RV> >= 2 * G::UNITS) ? 2 : 1;
    static_assert(NPS<RV> >= SPD * G::UNITS, "every DMA unit needs a pair-step");
    if constexpr (PS % SPD == 0 && PS / SPD < G::UNITS) row_unit<PS / SPD>(bt);      // in order: the cursor advances
    constexpr int per = (NMT + NPS<RV> - 2) / (NPS<RV> - 1);
    if constexpr (PS >= 1) {
      constexpr int m0 = (PS - 1) * per;
      micros<m0>(std::make_integer_sequence<int, (m0 < NMT ? (NMT - m0 < per ? NMT - m0 : per) : 0)>());
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  template <int RV, int... PSs>
  __device__ __forceinline__ void psteps(GroupState<RV>& st, std::integer_sequence<int, PSs...>) { (pstep<RV, PSs>(st), ...); }

  // Output rows ro .. ro + RV - 1 of item x from window buffer `buf`; `par`: the exchange area of this group.  On return the
  // other K half's share of the accumulators is on its way through LDS, this wave's share sits in fin[] (still without the
  // partner's part: take_partner() after the next barrier) and pro / pvcol / prout describe where its rows go.
  template <int RV>
  __device__ __forceinline__ void group(const Item& x, int ro, int buf, int par) {
    GroupState<RV> st;
#pragma unroll
    for (int df = 0; df < KF; ++df) st.vb[df] = lds0 + (unsigned)(buf * G::WBUF + boff[df]);
    st.bq[0] = frag<RV>(st, 0);
    st.bq[1] = frag<RV>(st, 1);
    st.bq[2] = frag<RV>(st, 2);
    st.bq[3] = frag<RV>(st, 3);
    psteps<RV>(st, std::make_integer_sequence<int, NPS<RV>>());
    // hand-over: K half 0 finishes rows [0, RV / 2), K half 1 rows [RV / 2, RV)
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // the last MFMAs' results (the compiler does not see MFMAs in the asm statements)
    constexpr int HV = RV / 2;
    const unsigned xw = xch0 + (unsigned)(par * 4 * G::XCH + (wave ^ 2) * G::XCH + lane * 16);      // the partner's area
#pragma unroll
    for (int j = 0; j < HV; ++j) {
      const f32x4 give = kh ? st.acc[j] : st.acc[HV + j];
      const f32x4 keep = kh ? st.acc[HV + j] : st.acc[j];
      if constexpr (!(ABL & 4)) *(lds_u4v_rw*)(uintptr_t)(xw + (unsigned)(j * 1024)) = __builtin_bit_cast(u4v, give);
      fin[j] = keep;
    }
#pragma unroll
    for (int j = 0; j < HR; ++j) {
      if (j >= HV) fin[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      pro[j] = j < HV ? row_offset(x, ro + kh * HV + j) : kOob;
    }
    pvcol = vcol;
    prout_hi = rout_hi;
    prout_lo = rout_lo;
  }

  // after the barrier that follows a group: fetch the partner's part of this wave's rows (rows the group did not have: stale
  // LDS that ends in a dropped store)
  __device__ __forceinline__ void take_partner(int par) {
    if constexpr (ABL & 4) return;
    const unsigned xr = xch0 + (unsigned)(par * 4 * G::XCH + wave * G::XCH + lane * 16);
#pragma unroll
    for (int j = 0; j < HR; ++j) got[j] = __builtin_bit_cast(f32x4, *(lds_u4v*)(uintptr_t)(xr + (unsigned)(j * 1024)));
    if (pro[HR - 1] == kOob) {       // a short group (wave-uniform, rare): rows it did not have hold stale bits in the partner's area --
#pragma unroll                       // zero them, so that |max| stays under the plan's bound (their stores are dropped anyway)
      for (int j = 0; j < HR; ++j)
        if (pro[j] == kOob) fin[j] = got[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __device__ __forceinline__ void flush_epilogue() { micros<0>(std::make_integer_sequence<int, NMT>()); }
};

template <int KT, int KF, int ACT, int ABL>
__device__ __forceinline__ void nhwc_conv_f16x3_body(const SplitConvArgs& a, const unsigned char* smem) {
  using G = SGeo<KT, KF>;
  constexpr int R = G::R;
  SplitWalk<KT, KF, ACT, ABL> wk(a, (const lds_byte*)smem);
  // workgroups w and w + 8 (one XCD) are a pair: the same tiles, the two halves of the output channels
  const int slot = (((int)blockIdx.x >> 4) << 3) | ((int)blockIdx.x & 7);
  const int nslot = (int)gridDim.x >> 1;

  Item pf;
  int pf_it = slot, pf_g = 0;
  bool pf_live = false;
  auto pf_seek = [&]() {
    pf_live = false;
    while (pf_it < a.n_tiles) {
      if (wk.decode(pf_it, pf)) { pf_g = 0; pf_live = true; return; }
      pf_it += nslot;
    }
  };
  int pbuf = 0;
  auto pf_begin = [&]() {
    wk.begin(wk.bt, pf, pf.o0 + pf_g * R - G::P, pbuf);
    pbuf ^= 1;
    if (++pf_g >= pf.ngroups) { pf_it += nslot; pf_seek(); }
  };
  wk.bt.live = false;
  wk.bt.pcur = 0; wk.bt.wcur = 0; wk.bt.in_end = 0; wk.bt.dst0 = wk.lds0; wk.bt.v0 = wk.bt.v1 = wk.bt.v2 = 0;
  pf_seek();
  if (pf_live) { pf_begin(); wk.fetch_all(wk.bt); }

  unsigned long long t_vm = 0, t_bar = 0, t_grp = 0;
  const unsigned long long t_start = (ABL & 32) ? __builtin_amdgcn_s_memtime() : 0ull;
  Item cur;
  int cbuf = 0, par = 0;
  bool pending = false;            // a finished group waits for its partner's part and its epilogue
  for (int it = slot; it < a.n_tiles; it += nslot) {
    if (!wk.decode(it, cur)) continue;
    wk.begin_item(cur);
    for (int gidx = 0; gidx < cur.ngroups; ++gidx) {
      if constexpr (ABL & 32) {                                // where a group boundary spends its time (cycles of s_memtime per wave)
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
      else wk.bt.dst0 = wk.lds0 + (unsigned)((cbuf ^ 1) * G::WBUF + wk.wave * G::ROWB);      // nothing left to fetch: the units move zeros into the idle buffer
      const int ro = cur.o0 + gidx * R;
      const int left = cur.o1 - ro;                           // > 0
      if (R == 8 && left > 6) wk.template group<R>(cur, ro, cbuf, par);
      else if (left > 4) wk.template group<6>(cur, ro, cbuf, par);
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
  if constexpr (ABL & 32) {          // amax_out doubles as the probe's output: [0..1] total cycles, [2..3] vmcnt waits, [4..5] barrier waits, [6..7] groups
    if ((threadIdx.x & 63) == 0 && a.amax_out) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.amax_out);
      atomicAdd(o + 0, __builtin_amdgcn_s_memtime() - t_start);
      atomicAdd(o + 1, t_vm);
      atomicAdd(o + 2, t_bar);
      atomicAdd(o + 3, t_grp);
    }
  } else
  vs_absmax_commit(wk.am * a.out_scale2[1], a.amax_out);
}

// (the packed-fp32 build: VS_ABLATION instances only; the product launches the _scalar build below -- conv_nhwc.hip has the reason)
template <int KT, int KF, int ACT, int ABL = 0>
__global__ __launch_bounds__(256, 1)
void nhwc_conv_f16x3_kernel(SplitConvArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SGeo<KT, KF>::LDS_BYTES];
  nhwc_conv_f16x3_body<KT, KF, ACT, ABL>(a, smem);
}
template <int KT, int KF, int ACT>
__global__ __launch_bounds__(256, 1) VS_NO_PACKED_FP32
void nhwc_conv_f16x3_scalar_kernel(SplitConvArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SGeo<KT, KF>::LDS_BYTES];
  nhwc_conv_f16x3_body<KT, KF, ACT, 0>(a, smem);
}

// w [co][ci][KT][KF] fp32 -> [co half][wave = (block m, K half kh)][tap][plane][lane][j] f16:
//   hi / lo split of w * s_w at co = 32 half + 16 m + (lane & 15), ci = 32 kh + 8 (lane >> 4) + j
__global__ void nhwc_f16x3_pack_kernel(const float* __restrict__ w, const float* __restrict__ w_scale2, unsigned short* __restrict__ out,
                                       float* __restrict__ l1, int KT, int KF) {
  const int ntap = KT * KF;
  const int total = 2 * 4 * ntap * 64 * 8;                   // one plane's elements
  const float s = w_scale2[0];
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int j = e & 7, lane = (e >> 3) & 63;
    const int tap = (e >> 9) % ntap, wv = ((e >> 9) / ntap) & 3, half = (e >> 9) / ntap / 4;
    const int co = 32 * half + 16 * (wv & 1) + (lane & 15), ci = 32 * (wv >> 1) + 8 * (lane >> 4) + j;
    const int dt = tap / KF, df = tap - dt * KF;
    const float v = w[((size_t)(co * 64 + ci) * KT + dt) * KF + df] * s;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const size_t o = ((((size_t)(half * 4 + wv) * ntap + tap) * 2) * 64 + lane) * 8 + j;
    out[o] = __builtin_bit_cast(unsigned short, hi);
    out[o + 512] = __builtin_bit_cast(unsigned short, lo);
  }
  // L1 norm of every output channel's weights (the plan kernel's bound): block 0, four threads per channel
  if (l1 != nullptr && blockIdx.x == 0) {
    const int co = threadIdx.x >> 2, part = threadIdx.x & 3, n = 64 * ntap;
    float t = 0.f;
    for (int k = part; k < n; k += 4) t += fabsf(w[(size_t)co * n + k]);
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    if (part == 0) l1[co] = t;
  }
}

// One layer's epilogue constants and output scale, on the device (no host round trip):
//   in_scale2 = {s_x, 1 / s_x} of the layer's input planes, w_scale2 = {s_w, 1 / s_w}, amax_in = tracked max |x| (bits, n_amax slots),
//   bn_scale / bn_shift [64]: y = act(z * bn_scale + bn_shift), l1 [64] = sum |w[co]|
//   -> eff_scale [64] = bn_scale / (s_x s_w), eff_shift = bn_shift,
//      out_scale2 = {s_y, 1 / s_y}: the power of two that maps BOUND = max_c (|bn_scale_c| l1_c max|x| + |bn_shift_c|) (>= max |y| for
//      ReLU and Mish: |act(v)| <= max(|v|, 0.31)) into [2^14, 2^15)
__global__ void nhwc_f16x3_plan_kernel(const float* __restrict__ in_scale2, const float* __restrict__ w_scale2, const unsigned* __restrict__ amax_in,
                                       int n_amax, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, const float* __restrict__ l1,
                                       float* __restrict__ eff_scale, float* __restrict__ eff_shift, float* __restrict__ out_scale2) {
  const int c = threadIdx.x;                                  // 64 threads
  unsigned mb = 0;
  for (int i = c; i < n_amax; i += 64) mb = amax_in[i] > mb ? amax_in[i] : mb;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned other = __shfl_xor(mb, o, 64); mb = other > mb ? other : mb; }
  const float xmax = __uint_as_float(mb);
  const float inv = in_scale2[1] * w_scale2[1];
  eff_scale[c] = bn_scale[c] * inv;
  eff_shift[c] = bn_shift[c];
  float bound = fmaxf(fabsf(bn_scale[c]) * l1[c] * xmax + fabsf(bn_shift[c]), 0.3125f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o, 64));
  if (c == 0) {
    float s = 1.f, si = 1.f;
    if (bound > 0.f && bound < 3.0e38f) {
      int e = 0;
      (void)frexpf(bound, &e);                                // bound = f 2^e, f in [0.5, 1)
      int k = 15 - e;
      k = k > 100 ? 100 : (k < -100 ? -100 : k);
      s = ldexpf(1.f, k);
      si = ldexpf(1.f, -k);
    }
    out_scale2[0] = s;
    out_scale2[1] = si;
  }
}

int split_num_cus() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  return v;
}

template <int KT, int KF>
int launch_split(SplitConvArgs a, int act, hipStream_t stream) {
  constexpr int R = SGeo<KT, KF>::R;
  const int nk_max = (a.T + a.dil - 1) / a.dil;
  const long long base_items = (long long)a.B * a.dil * a.nstrip;
  int nseg = (int)((2048 + base_items - 1) / base_items);     // >= ~16 tiles per workgroup pair
  if (nseg > nk_max / 18) nseg = nk_max / 18;
  if (nseg < 1) nseg = 1;
  int seg_rows = (nk_max + nseg - 1) / nseg;
  seg_rows = (seg_rows + R - 1) / R * R;
  nseg = (nk_max + seg_rows - 1) / seg_rows;
  a.nseg = nseg;
  a.seg_rows = seg_rows;
  const long long n_tiles = base_items * nseg;
  VS_REQUIRE(n_tiles < (1LL << 30), "nhwc f16x3 conv: too many tiles");
  a.n_tiles = (int)n_tiles;
  static int cus = 0;
  if (!cus) cus = split_num_cus();
  long long want = (2 * n_tiles + 15) / 16 * 16;              // pairs sit 8 apart: the grid is a multiple of 16
  int grid = cus / 16 * 16;
  if (grid < 16) grid = 16;
  if (want < grid) grid = (int)want;
  const dim3 g((unsigned)grid), block(256);
#ifdef VS_ABLATION        // timing ablations / in-kernel probes (tools/split_conv_micro.py): make -C voicesplit_amd/csrc ABLATION=1
  const int abl = vs_opt(VS_OPT_ABLATION);          // timing ablations, Mish only
  if (abl && act == VS_ACT_MISH) {
    if (abl == 1) hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 1>), g, block, 0, stream, a);
    else if (abl == 2) hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 2>), g, block, 0, stream, a);
    else if (abl == 4) hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 4>), g, block, 0, stream, a);
    else if (abl == 8) hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 8>), g, block, 0, stream, a);
    else if (abl == 32) hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 32>), g, block, 0, stream, a);
    else if (abl == 96) hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 96>), g, block, 0, stream, a);
    else if (abl == 33) hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 33>), g, block, 0, stream, a);
    else if (abl == 34) hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 34>), g, block, 0, stream, a);
    else if (abl == 35) hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 35>), g, block, 0, stream, a);
    else hipLaunchKernelGGL((nhwc_conv_f16x3_kernel<KT, KF, VS_ACT_MISH, 16>), g, block, 0, stream, a);
  } else
#endif
  // the build without packed-fp32 VALU instructions (1-1.5 % faster in round 5, eight of eight pairs; tools/epilogue_slot_probe.hip in
  // round 6 says why: one v_pk_fma_f32 behind an MFMA costs 17 cycles, two scalar v_fma_f32 two).  The packed build is gone.
  if (act == VS_ACT_MISH) hipLaunchKernelGGL((nhwc_conv_f16x3_scalar_kernel<KT, KF, VS_ACT_MISH>), g, block, 0, stream, a);
  else if (act == VS_ACT_RELU) hipLaunchKernelGGL((nhwc_conv_f16x3_scalar_kernel<KT, KF, VS_ACT_RELU>), g, block, 0, stream, a);
  else if (act == VS_ACT_NONE) hipLaunchKernelGGL((nhwc_conv_f16x3_scalar_kernel<KT, KF, VS_ACT_NONE>), g, block, 0, stream, a);
  else VS_REQUIRE(false, "nhwc f16x3 conv: unsupported activation %d", act);
  VS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

size_t vs_nhwc_f16x3_packed_bytes(int KT, int KF) { return (size_t)2 * 64 * 64 * KT * KF * 2; }

// packed <- the split, fragment-ordered weights; w_scale2 (device {s, 1 / s}, e.g. from vs_pow2_scale_impl) is read on the device;
// l1 (64 floats, may be NULL) <- the L1 norm of every output channel's weights
int vs_nhwc_f16x3_pack_impl(const float* w, const float* w_scale2, void* packed, float* l1, int KT, int KF, hipStream_t stream) {
  VS_REQUIRE(w && w_scale2 && packed, "nhwc f16x3 pack: NULL argument");
  VS_REQUIRE((KT == 5 && KF == 5) || (KT == 7 && KF == 1), "nhwc f16x3 pack: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  hipLaunchKernelGGL(nhwc_f16x3_pack_kernel, dim3(64), dim3(256), 0, stream, w, w_scale2, reinterpret_cast<unsigned short*>(packed), l1, KT, KF);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_nhwc_f16x3_plan_impl(const float* in_scale2, const float* w_scale2, const unsigned* amax_in, int n_amax, const float* bn_scale,
                            const float* bn_shift, const float* l1, float* eff_scale, float* eff_shift, float* out_scale2, hipStream_t stream) {
  VS_REQUIRE(in_scale2 && w_scale2 && amax_in && bn_scale && bn_shift && l1 && eff_scale && eff_shift && out_scale2 && n_amax > 0,
             "nhwc f16x3 plan: bad argument");
  hipLaunchKernelGGL(nhwc_f16x3_plan_kernel, dim3(1), dim3(64), 0, stream, in_scale2, w_scale2, amax_in, n_amax, bn_scale, bn_shift, l1,
                     eff_scale, eff_shift, out_scale2);
  VS_LAUNCH_CHECK();
  return 0;
}

// out = act(scale[co] * sum_{ci,dt,df} (w_hi + w_lo)[co][ci][dt][df] (in_hi + in_lo)[b][t + (dt - KT/2) dil][f + df - KF/2][ci] + shift[co]) * s_y,
// written as hi / lo f16 planes; zero padding; amax_out (VS_AMAX_SLOTS uints or NULL) tracks max |out / s_y|.
int vs_nhwc_conv_f16x3_impl(const void* in_hi, const void* in_lo, const void* packed, const float* scale, const float* shift,
                            const float* out_scale2, void* out_hi, void* out_lo, unsigned* amax_out,
                            int B, int T, int F, int KT, int KF, int dil, int act, hipStream_t stream) {
  VS_REQUIRE(in_hi && in_lo && packed && scale && shift && out_scale2 && out_hi && out_lo, "nhwc f16x3 conv: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "nhwc f16x3 conv: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((long long)T * F * 128 < 0x7F000000LL, "nhwc f16x3 conv: an utterance of %d x %d pixels does not fit the 31-bit offsets of the epilogue", T, F);
  VS_REQUIRE(((reinterpret_cast<uintptr_t>(in_hi) | reinterpret_cast<uintptr_t>(in_lo) | reinterpret_cast<uintptr_t>(out_hi) |
               reinterpret_cast<uintptr_t>(out_lo) | reinterpret_cast<uintptr_t>(packed)) & 15) == 0, "nhwc f16x3 conv: buffers must be 16-byte aligned");
  SplitConvArgs a{reinterpret_cast<const unsigned short*>(in_hi), reinterpret_cast<const unsigned short*>(in_lo),
                  reinterpret_cast<const unsigned short*>(packed), scale, shift, out_scale2,
                  reinterpret_cast<unsigned short*>(out_hi), reinterpret_cast<unsigned short*>(out_lo), amax_out,
                  B, T, F, dil, (F + STRIP - 1) / STRIP, 1, 0, 0};
  if (KT == 5 && KF == 5) return launch_split<5, 5>(a, act, stream);
  if (KT == 7 && KF == 1) return launch_split<7, 1>(a, act, stream);
  VS_REQUIRE(false, "nhwc f16x3 conv: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  return -1;
}

namespace {
// fp32 [n] -> hi / lo f16 planes of x * s (s = scale2[0]); and back: x = (hi + lo) * scale2[1]
__global__ void f16x3_split_kernel(const float* __restrict__ x, const float* __restrict__ scale2, unsigned short* __restrict__ hi,
                                   unsigned short* __restrict__ lo, long long n) {
  const float s = scale2[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i] * s;
    const _Float16 h = __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(v, 0.f))[0];
    const _Float16 l = __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(v - (float)h, 0.f))[0];
    hi[i] = __builtin_bit_cast(unsigned short, h);
    lo[i] = __builtin_bit_cast(unsigned short, l);
  }
}
__global__ void f16x3_merge_kernel(const unsigned short* __restrict__ hi, const unsigned short* __restrict__ lo, const float* __restrict__ scale2,
                                   float* __restrict__ x, long long n) {
  const float si = scale2[1];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    x[i] = ((float)__builtin_bit_cast(_Float16, hi[i]) + (float)__builtin_bit_cast(_Float16, lo[i])) * si;
}
}  // namespace

int vs_f16x3_split_impl(const float* x, const float* scale2, void* hi, void* lo, long long n, hipStream_t stream) {
  VS_REQUIRE(x && scale2 && hi && lo && n > 0, "f16x3 split: bad argument");
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(f16x3_split_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, stream, x, scale2,
                     reinterpret_cast<unsigned short*>(hi), reinterpret_cast<unsigned short*>(lo), n);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_f16x3_merge_impl(const void* hi, const void* lo, const float* scale2, float* x, long long n, hipStream_t stream) {
  VS_REQUIRE(x && scale2 && hi && lo && n > 0, "f16x3 merge: bad argument");
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(f16x3_merge_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, stream,
                     reinterpret_cast<const unsigned short*>(hi), reinterpret_cast<const unsigned short*>(lo), scale2, x, n);
  VS_LAUNCH_CHECK();
  return 0;
}

// One layer from fp32 weights and folded BatchNorm constants.
//   wpart (vs_nhwc_f16x3_wpart_bytes, 256-byte aligned): [packed weights][l1: 64 f][w_scale2: 2 f][|max| scratch: 1 u] -- a function of the
//     weights alone; packed_ready != 0: already there (vs_prepare_weights keeps it in the prepared blob, which this call then only reads)
//   plan (128 floats): [eff_scale: 64][eff_shift: 64] -- per call
// vs_nhwc_f16x3_layer_scratch_bytes = the two back to back (the C-ABI's single scratch buffer).
size_t vs_nhwc_f16x3_wpart_bytes(int KT, int KF) { return vs_nhwc_f16x3_packed_bytes(KT, KF) + 512; }
size_t vs_nhwc_f16x3_layer_scratch_bytes(int KT, int KF) { return vs_nhwc_f16x3_wpart_bytes(KT, KF) + 512; }

// the weight part of a layer's scratch: power-of-two scale, split + packed weights, row L1 norms
int vs_nhwc_f16x3_prepare_wpart_impl(const float* w, void* wpart, int KT, int KF, hipStream_t stream) {
  VS_REQUIRE(w && wpart && (reinterpret_cast<uintptr_t>(wpart) & 255) == 0, "nhwc f16x3 weights: NULL or misaligned argument");
  VS_REQUIRE((KT == 5 && KF == 5) || (KT == 7 && KF == 1), "nhwc f16x3 weights: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  char* base = reinterpret_cast<char*>(wpart);
  float* tail = reinterpret_cast<float*>(base + vs_nhwc_f16x3_packed_bytes(KT, KF));
  if (int rc = vs_pow2_scale_impl(w, (long long)64 * 64 * KT * KF, reinterpret_cast<unsigned*>(tail + 66), tail + 64, stream)) return rc;
  return vs_nhwc_f16x3_pack_impl(w, tail + 64, base, tail, KT, KF, stream);
}

int vs_nhwc_f16x3_layer_impl(const void* in_hi, const void* in_lo, const float* in_scale2, const unsigned* amax_in, int n_amax,
                             const float* w, const float* bn_scale, const float* bn_shift, void* wpart, int packed_ready, float* plan,
                             void* out_hi, void* out_lo, float* out_scale2, unsigned* amax_out,
                             int B, int T, int F, int KT, int KF, int dil, int act, hipStream_t stream) {
  VS_REQUIRE(wpart && plan && (reinterpret_cast<uintptr_t>(wpart) & 255) == 0, "nhwc f16x3 layer: the weight scratch must be 256-byte aligned");
  VS_REQUIRE((KT == 5 && KF == 5) || (KT == 7 && KF == 1), "nhwc f16x3 layer: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  char* base = reinterpret_cast<char*>(wpart);
  float* tail = reinterpret_cast<float*>(base + vs_nhwc_f16x3_packed_bytes(KT, KF));
  float* l1 = tail;
  float* w_scale2 = tail + 64;
  unsigned* amax_w = reinterpret_cast<unsigned*>(tail + 66);
  float* eff_scale = plan;
  float* eff_shift = plan + 64;
  (void)amax_w;
  if (!packed_ready) { if (int rc = vs_nhwc_f16x3_prepare_wpart_impl(w, wpart, KT, KF, stream)) return rc; }
  if (int rc = vs_nhwc_f16x3_plan_impl(in_scale2, w_scale2, amax_in, n_amax, bn_scale, bn_shift, l1, eff_scale, eff_shift, out_scale2, stream)) return rc;
  return vs_nhwc_conv_f16x3_impl(in_hi, in_lo, base, eff_scale, eff_shift, out_scale2, out_hi, out_lo, amax_out, B, T, F, KT, KF, dil, act, stream);
}
