// 64 -> 64 convolution of the conv stack in the bf16 configuration (BASELINE configs[2], VS_MATH_BF16):
// cnn2 (7x1) and cnn3..cnn7 (5x5, time dilation 1..16) of models/voicesplit/model.py:21-48, forward and data
// gradient (the data gradient is the same kernel over weights packed transposed + tap-flipped).
//
// Layout: activations are CHANNELS-LAST bf16, [B][T][F][64] -- a pixel's 64 channels are 128 contiguous bytes.
// That is the B operand of v_mfma_f32_16x16x32_bf16 as it lies in memory (a lane's 8 consecutive k = 8
// consecutive channels of one pixel = one 16-byte piece), so the operand goes HBM -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds) and LDS -> register by one ds_read_b128 with no conversion, no transpose and no
// VALU in between.  (The fp32 / split-f16 path keeps [B][64][T][F] fp32 and converts while staging:
// conv_f16x3_pk.hip.)
//
// Decomposition.  A dilated layer is `dil` residue classes of t, each a dense conv along the class rows
// k (t = cls + k*dil).  Work item = (utterance, class, 32-column strip, segment of class rows); a persistent
// workgroup (one per CU, 4 waves, one per SIMD) walks its items in GROUPS of R = 8 output rows:
//   * weights stay in REGISTERS for the whole launch: wave q owns output channels [16q, 16q+16) and holds the
//     A fragments of all KT*KF taps x 2 k-chunks (5x5: 50 fragments = 200 VGPRs).  No weight traffic in the loop,
//     no cross-wave reduction, every wave reads the same pixels from LDS;
//   * the R + KT - 1 input rows of a group (its window) sit at FIXED offsets of one of two LDS window buffers (16-byte
//     pieces XOR-swizzled on the SOURCE address: the fragment reads are conflict-free).  The window of the NEXT group is
//     fetched while this one is computed, a whole group (5x5: 800 MFMAs per wave, ~13k cycles) ahead: each wave owns
//     every fourth row and issues it as ONE unit -- a buffer descriptor for the tensor row, M0, and the row's 1 KiB
//     chunks by immediate offset (buffer_load_dwordx4 ... lds) -- placed between two MFMAs.  Columns and rows outside
//     the image are zeros by the descriptor's range check: no address arithmetic, no zero page.  The KT - 1 rows two
//     consecutive groups share are fetched twice (the second time from L2) -- in exchange every fragment read is
//     base register + immediate.  That matters because a wave alone on its SIMD issues one instruction per 4 cycles
//     (tools/issue_probe.hip: between two 16-cycle MFMAs there is room for ~2 independent instructions; a dependent
//     address add + read pair costs ~15 cycles): the ring version spent more issue slots on addresses than on reads.
//     One s_waitcnt vmcnt(0) + one s_barrier per group;
//   * a group is one straight-line block: for window row i, tap column df, k-chunk, column block: ONE fragment read,
//     multiplied with the <= KT taps dt that send it to output row i - dt of the group (0.3 LDS reads per MFMA;
//     no MFMA is issued for a (row, tap) pair outside the group, so nothing is wasted at group edges).  The MFMAs are volatile
//     inline assembly -- weights in AGPRs, accumulators in VGPRs, the first product of an accumulator with the constant 0 as its
//     addend -- so that the compiler can move nothing across them: the instruction stream is the one written here;
//   * the epilogue (scale / shift, activation or its derivative, BatchNorm sums, bf16 rounding, 8-byte stores of 4 channels of
//     one pixel; lanes / rows outside the image dropped by the stores' range check) is cut into MICRO-OPS of one scalar VALU
//     instruction per channel of a channel pair (or one transcendental, or one store), dealt out ONE PER MFMA [round 6].
//     tools/epilogue_slot_probe.hip: behind an MFMA one or two independent scalar VALU instructions are nearly free (17.4 / 18.5
//     cycles per MFMA against 16.5), a third costs 7 cycles, a dependent chain of three 16 -- and ONE v_pk_fma_f32 17, so the
//     kernels are built without packed-fp32 instructions.  (Rounds 3-5 put up to four dependent, partly packed instructions
//     behind every second MFMA: the dy form ran 2.10-2.27 ms against the plain conv's 1.66; now 1.82-1.93.)
//     Output row r is complete after window row r + KT - 1 and its micro-ops ride on the MFMAs behind that; the last three rows
//     of a group, which complete when its MFMAs are nearly over, are CARRIED: their accumulators, row offsets and descriptors
//     outlive the group and their micro-ops ride on the first MFMAs of the NEXT group (whichever item that is; three dummy rows at
//     the launch's start, a flush behind its end).  Plan<RV> deals the micro-op list out over the group's MFMA slots at compile time;
//   * train mode: per-channel sum / sum of squares of the outputs are accumulated per lane over the whole launch
//     and flushed once (shuffle over the 16 pixels of a fragment, one fp64 atomic per channel and wave).
// The compiler cannot see an MFMA inside an assembly statement and so inserts none of the wait states the hardware wants
// between a matrix-pipe write of a VGPR and a VALU read of it: the structure provides them (>= 4 MFMAs between a row's last
// product and the first micro-op that reads it, an explicit s_nop in front of the hand-over of a group's last row), and
// tools/mfma_hazard_scan.py verifies it on the emitted assembly.
// Every input element is read from HBM once per strip (+12.5 % halo columns for 5x5, + KT-1 halo rows per segment;
// the re-fetched window rows are L2 hits),
// every output written once.
#include <utility>

#include "vs_internal.h"

namespace {

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
constexpr float kLog2e = 1.44269504088896340736f;
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) const u4v lds_u4v;


constexpr int STRIP = 32;           // output columns per strip = 2 MFMA column blocks of 16
constexpr int NB = STRIP / 16;
constexpr int R = 8;                // output rows per group
#ifndef VS_NHWC_SGB
#define VS_NHWC_SGB 0
#endif

struct NhwcConvArgs {
  const unsigned short* in;         // [B][T][F][64] bf16
  const unsigned short* wpk;        // [4 waves][taps][2 k-chunks][64 lanes][8] bf16 (vs_nhwc_pack_impl)
  const float* scale;               // [64]  out = act(acc * scale + shift)
  const float* shift;               // [64]
  unsigned short* out;              // [B][T][F][64] bf16
  double* bn_stats;                 // [VS_BN_STAT_SLOTS][64][2] or NULL
  // data-gradient epilogue of the backward pass (DY instances): the output pixel's z and the BatchNorm constants of the
  // layer whose activation gradient this launch produces
  const unsigned short* z2;         // [B][T][F][64] bf16
  const float* bn2_scale; const float* bn2_shift; const float* bn2_mean; const float* bn2_invstd;   // [64] each
  int B, T, F, dil;
  int nstrip, nseg, seg_rows, n_items;
  unsigned* turn;                   // deterministic mode: the workgroups flush their statistics in workgroup order (vs_common.h); else NULL
};

// XOR swizzle of the 16-byte pieces of a staged pixel (128 bytes = half a bank row; pixel parity picks the half): a
// ds_read_b128 is served in lane groups {0-3, 12-15, 20-27} ... that mix two k-groups, and the 16 pixels of a group
// start at tap column df = 0..4.  ((p >> 1) & 3) << 1 puts every such group on 16 distinct 16-byte slots for every df
// (exhaustive search over the GF(2)-linear maps of the pixel index; the first choice, (p >> 1) & 7, was conflict-free for
// even df only: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.375 in profiles/r03_train_bf16_rocprof).
__device__ __forceinline__ int swz(int p) { return ((p >> 1) & 3) << 1; }

template <int KT, int KF>
struct Geo {
  static constexpr int P = KT / 2, PF = KF / 2, H = KT - 1;
  static constexpr int NTAP = KT * KF;
  static constexpr int RWPX = STRIP + 2 * PF;            // staged pixels per row (36 / 32)
  static constexpr int CPR = (RWPX * 8 + 63) / 64;       // 1 KiB DMA chunks per row (5 / 4); the last may be part padding
  static constexpr int ROWB = CPR * 1024;                // bytes per ring row
  static constexpr int WIN = R + H;                      // input rows of a group
  // LDS: two window buffers of WIN rows.  A group's rows sit at FIXED offsets of its buffer (the H rows it shares with the
  // previous group are fetched again -- L2 hits -- instead of being kept in a ring), so a fragment read is a per-group base
  // register + an immediate: with one wave per SIMD every instruction is a 4-cycle issue slot of the MFMA stream, and an
  // address add in front of a read (plus the dependency on it) cost more than the read (tools/issue_probe.hip).
  static constexpr int WBUF = WIN * ROWB;
  static constexpr int LDS_BYTES = 2 * WBUF;
  static constexpr int UNITS = (WIN + 3) / 4;            // DMA row units of a wave per group (row rho belongs to wave rho % 4)
};

struct Item { int b, cls, strip, o0, o1, in_end, ngroups; };

// DY: the launch is a data gradient whose result da feeds a BatchNorm + activation backward: the epilogue turns it into
// dy = da * act'(z * scale + shift) on the spot (z = the output pixel's pre-BatchNorm value, loaded two epilogue rows ahead),
// accumulates the two sums of the BatchNorm backward (sum dy, sum dy * xhat) like the STATS epilogue accumulates the
// forward's, and stores dy: the separate statistics pass over (da, z) disappears and the apply pass needs no activation
// derivative (dz = cA dy + cB z + cC).  ACT is then the activation whose derivative is taken.
template <int KT, int KF, int ACT, bool STATS, bool DY = false>
struct ConvWalk {
  using G = Geo<KT, KF>;
  static constexpr int P = G::P, PF = G::PF, H = G::H, NTAP = G::NTAP;
  static constexpr unsigned kOob = 0x7FFFFFF0u;

  const NhwcConvArgs& a;
  int lane, wave, n, g;
  vs_bf16x8 wf[NTAP][2];                     // this wave's A fragments: AGPRs for the whole launch (every use is an "a" operand)
  // This lane's 4 output channels as two pairs.  csc / csh: the epilogue's scale and
  // shift; DY: the BatchNorm scale / shift of the layer below, times log2(e).  a1 / a2: the two per-channel sums.
  f2v csc[2], csh[2], a1[2], a2[2];
  int boff[KF][2];                           // per-lane byte offset of the B fragment (column shift df, k-chunk kc) inside a row image
  unsigned vcol[NB];
  int vdma;                                  // per-lane source offset of a DMA chunk inside a tensor row (launch constant)
  __amdgpu_buffer_rsrc_t rout, rz;
  // The epilogues of a group's last three rows ride on the first MFMAs of the NEXT group (whichever item that belongs to), so
  // what they need outlives the group: the accumulators, the rows' byte offsets (kOob: no such row -- the three dummies the launch
  // starts with, the third row of a two-row tail group), the item's column offsets / masks and descriptors, and the ring of z values
  // of the dy form (filled two epilogue rows ahead, across the group boundary).
  f32x4 cacc[3][NB];
  unsigned crow[3];
  unsigned cvcol[NB];
  float ccok[NB];
  __amdgpu_buffer_rsrc_t crout, crz;
  u2v zq3[3][NB];
  unsigned lds0;
  const lds_byte* smem;

  __device__ __forceinline__ ConvWalk(const NhwcConvArgs& a_, const lds_byte* smem_) : a(a_), smem(smem_) {
    const int tid = threadIdx.x;
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    n = lane & 15;
    g = lane >> 4;
    lds0 = (unsigned)(uintptr_t)smem_;
    const u4v* wp = reinterpret_cast<const u4v*>(a.wpk) + (size_t)wave * NTAP * 2 * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) wf[tap][kc] = __builtin_bit_cast(vs_bf16x8, wp[(tap * 2 + kc) * 64]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = wave * 16 + g * 4 + r;
      csc[r >> 1][r & 1] = DY ? a.bn2_scale[ch] * kLog2e : a.scale[ch];
      csh[r >> 1][r & 1] = DY ? a.bn2_shift[ch] * kLog2e : a.shift[ch];
      a1[r >> 1][r & 1] = 0.f;
      a2[r >> 1][r & 1] = 0.f;
    }
#pragma unroll
    for (int df = 0; df < KF; ++df) {
      const int p = n + df;
      boff[df][0] = p * 128 + ((g ^ swz(p)) << 4);         // column block nb: + 2048 (the swizzle is 8-periodic in p)
      boff[df][1] = boff[df][0] ^ 64;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      crow[j] = kOob;
#pragma unroll
      for (int nb2 = 0; nb2 < NB; ++nb2) { cacc[j][nb2] = f32x4{0.f, 0.f, 0.f, 0.f}; zq3[j][nb2] = u2v{0u, 0u}; }
    }
#pragma unroll
    for (int nb2 = 0; nb2 < NB; ++nb2) { cvcol[nb2] = kOob; ccok[nb2] = 0.f; }
    crout = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out), 0, 0, 0x00020000);      // no records: every access out of range
    crz = crout;
    {
      const int px = lane >> 3;                            // chunk c of a row: pixel 8 c + lane / 8, LDS piece lane % 8 holds channel
      vdma = (px - PF) * 128 + (((lane & 7) ^ swz(px)) << 4);     // piece (lane % 8) ^ swz(px) -- swz is 8-periodic: the same for every chunk
    }
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

  // ---- LDS-DMA of a group's WIN window rows, one ROW per unit --------------------------------------------------------
  // Class rows w_first .. w_first + WIN - 1 of item x -> window buffer `buf`.  Row rho belongs to wave rho % 4.  A row is
  // one buffer descriptor (base = the tensor row, records = its bytes, or none for a row outside the image: the DMA then
  // writes zeros, as it does for columns past the right edge), and its CPR chunks of 1 KiB (64 lanes x 16 bytes = 8
  // pixels) by immediate offset, which moves the memory and the LDS address alike (tools/lds_dma_offset_probe.hip).
  // Chunk 0 has its own offset register: in the first strip its first PF pixels lie left of the image -- a negative
  // offset, out of range as it stands -- and an immediate must not be added to that.
  struct Batch {
    long long row0;                // b * T + cls
    int in_end, w_first, buf, live;
    unsigned v0, v1;               // per-lane source offsets: chunk 0; chunks 1.. (= chunk 0's + 1 KiB, by immediate from there)
  };
  Batch bt;                        // the fetch in progress: the window of the group after the one being computed
  __device__ __forceinline__ void begin(Batch& bt, const Item& x, int w_first, int buf) const {
    bt.row0 = (long long)x.b * a.T + x.cls;
    bt.in_end = x.in_end; bt.w_first = w_first; bt.buf = buf; bt.live = 1;
    bt.v0 = (unsigned)(vdma + ((x.strip * STRIP) << 7));
    bt.v1 = bt.v0 + 1024u;
  }
  template <int J>
  __device__ __forceinline__ void row_unit(const Batch& bt) const {
    const int rho = wave + 4 * J;
    if (!bt.live || rho >= G::WIN) return;
    const int w = bt.w_first + rho;
    const bool ok = (w >= 0) & (w < bt.in_end);
    const unsigned long long p = reinterpret_cast<unsigned long long>(a.in) + (unsigned long long)(((bt.row0 + (long long)w * a.dil) * a.F) << 7);
    const u4v d = {(unsigned)p, (unsigned)(p >> 32) & 0xffffu, ok ? (unsigned)a.F * 128u : 0u, 0x00020000u};
    const unsigned dst = lds0 + (unsigned)(bt.buf * G::WBUF + rho * G::ROWB);
    unsigned keep;
    static_assert(G::CPR == 4 || G::CPR == 5, "a window row is 4 or 5 chunks");
    if (G::CPR == 5)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                   "s_add_u32 m0, %3, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %2, 0 offen lds\n\t"
                   "buffer_load_dwordx4 %4, %2, 0 offen offset:1024 lds\n\tbuffer_load_dwordx4 %4, %2, 0 offen offset:2048 lds\n\t"
                   "buffer_load_dwordx4 %4, %2, 0 offen offset:3072 lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(bt.v0), "s"(d), "s"(dst), "v"(bt.v1) : "memory", "scc");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
                   "s_add_u32 m0, %3, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %2, 0 offen lds\n\t"
                   "buffer_load_dwordx4 %4, %2, 0 offen offset:1024 lds\n\tbuffer_load_dwordx4 %4, %2, 0 offen offset:2048 lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(bt.v0), "s"(d), "s"(dst), "v"(bt.v1) : "memory", "scc");
  }
  template <int... Js>
  __device__ __forceinline__ void all_units(const Batch& bt, std::integer_sequence<int, Js...>) const { (row_unit<Js>(bt), ...); }
  __device__ __forceinline__ void fetch_all(const Batch& bt) const { all_units(bt, std::make_integer_sequence<int, G::UNITS>()); }

  // Output-side addressing (the epilogue's stores, the dy form's z loads): a buffer descriptor over the item's utterance,
  // a per-lane byte offset of its pixel (column block nb) and channel quad inside a row -- or an offset no row offset brings
  // back into range when the column is outside the image -- and a wave-uniform row offset, likewise out of range for rows
  // the item does not own.  The range check does the predication: no address arithmetic or branches in the epilogue.
  __device__ __forceinline__ void begin_item(const Item& x) {
    const size_t ub = (size_t)x.b * a.T * a.F * 128;
    const unsigned bytes = (unsigned)a.T * a.F * 128;
    rout = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(a.out) + ub, 0, bytes, 0x00020000);
    if (DY) rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.z2)) + ub, 0, bytes, 0x00020000);
#pragma unroll
    for (int nb2 = 0; nb2 < NB; ++nb2) {
      const int col = x.strip * STRIP + nb2 * 16 + n;
      vcol[nb2] = col < a.F ? (unsigned)(col * 128 + wave * 32 + g * 8) : kOob;
    }
  }
  __device__ __forceinline__ unsigned row_offset(const Item& x, int k) const {
    // (the product through an opaque scalar multiply: written in C++ the compiler branched around it -- a jump per row and store
    // inside the group's straight-line block)
    unsigned off;
    asm("s_mul_i32 %0, %1, %2" : "=s"(off) : "s"(x.cls + k * a.dil), "s"(a.F * 128));
    return k < x.o1 ? off : kOob;
  }

  // One group: output rows ro .. ro+RV-1 of item x (RV = R, or the even tail of the item; rows >= x.o1 are computed and
  // dropped), window rows 0 .. RV+H-1 of window buffer `buf`.
  template <int RV>
  struct GroupState {
    f32x4 acc[RV][NB];
    vs_bf16x8 bq[3];
    f2v y[NB][2];                      // the finished row's values on their way to the store (channel pairs)
    float cok[NB], mk[NB];             // DY: 1 for a column inside the image, else 0; the same for the row being finished
    f2v tz, ty, tu, tn, tr, tw, tp, tq;      // one channel pair in flight through the stages of the epilogue
    u2v pk[NB];                        // the packed pixel on its way to the store
    unsigned vb[KF][2];                // B-fragment read bases of this group's window buffer (column shift, k-chunk); row, column block: immediates
    int ro;
  };
  // micro-ops of one output row: the dy form's z prefetch, NB * 2 channel pairs x NSTAGE stages, NB x (pack, store)
  static constexpr int NSTAGE = DY ? (ACT == VS_ACT_MISH ? 19 : 7) : ACT == VS_ACT_MISH ? 12 : ACT == VS_ACT_RELU ? 2 : STATS ? 4 : 1;
  static constexpr int NMICRO = (DY ? 1 : 0) + NB * 2 * NSTAGE + 2 * NB;

  template <int RV>
  __device__ __forceinline__ vs_bf16x8 frag(const GroupState<RV>& st, int gi) const {
    const int nb = gi % NB, kc = (gi / NB) % 2, df = (gi / (2 * NB)) % KF, i = gi / (2 * NB * KF);
    return __builtin_bit_cast(vs_bf16x8, *(lds_u4v*)(uintptr_t)(st.vb[df][kc] + (unsigned)(i * G::ROWB + nb * 2048)));
  }

  // ---- the epilogue's stages: ONE scalar instruction per channel of the pair (x, y: independent of each other), or one
  // transcendental.  (As volatile assembly the compiler's hazard recognizer, which cannot look inside, put an s_nop in front of every
  // statement that reads a register the statement before wrote -- 349 per group; as C++ expressions they stay where they are
  // written because the MFMAs around them are volatile and a sched_barrier follows each.)
  static __device__ __forceinline__ float i_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
  static __device__ __forceinline__ float i_mul(float a, float b) { return a * b; }
  static __device__ __forceinline__ float i_add(float a, float b) { return a + b; }
  static __device__ __forceinline__ float i_add2(float a) { return a + 2.0f; }
  static __device__ __forceinline__ float i_exp2(float a) { return __builtin_amdgcn_exp2f(a); }
  static __device__ __forceinline__ float i_rcp(float a) { return __builtin_amdgcn_rcpf(a); }
  static __device__ __forceinline__ float i_max0(float a) { return fmaxf(a, 0.f); }
  static __device__ __forceinline__ float i_min_20log2e(float a) { return fminf(a, 20.0f * kLog2e); }
  static __device__ __forceinline__ float i_min_20(float a) { return fminf(a, 20.0f); }
  static __device__ __forceinline__ float i_mul_log2e(float a) { return a * kLog2e; }
  static __device__ __forceinline__ float i_mul_4ln2(float a) { return a * (4.0f * 0.69314718055994530942f); }
  static __device__ __forceinline__ float i_bf_lo(unsigned u) { return __uint_as_float(u << 16); }
  static __device__ __forceinline__ float i_bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

  // The row an epilogue works on: position IDX of the group's epilogue sequence -- 0..2 the rows carried over from the previous group,
  // 3 + r row r of this one.  FLUSH: the carried rows once more, behind the launch's last group (nothing to prefetch).
  static constexpr int epi_rows(int RV) { return RV >= 4 ? RV : 3; }          // epilogue rows a group works off (three carried in + its own but the last three)
  template <int RV, int IDX, int nb2, int e>
  __device__ __forceinline__ float acc_of(GroupState<RV>& st) const {
    if constexpr (IDX < 3) return cacc[IDX][nb2][e];
    else return st.acc[IDX - 3][nb2][e];
  }
  template <int RV, int IDX>
  __device__ __forceinline__ unsigned rowoff_of(const Item& x, const GroupState<RV>& st) const {
    if constexpr (IDX < 3) return crow[IDX];
    else return row_offset(x, st.ro + IDX - 3);
  }
  template <int IDX, int nb2>
  __device__ __forceinline__ unsigned vcol_of() const {
    if constexpr (IDX < 3) return cvcol[nb2];
    else return vcol[nb2];
  }
  // dy form: position IDX's first micro-op fetches z for the position two further on -- a row of this group, one of the rows it
  // will carry out (the next group's positions 0 and 1), or (IDX = 0) the third carried row.  Two epilogue rows = ~160 MFMAs ahead
  // of its first use: one row ahead the epilogue waited on HBM latency (round 3).
  template <int RV, int IDX, bool FLUSH>
  __device__ __forceinline__ void prefetch_z(const Item& x, GroupState<RV>& st) {
    constexpr int t = IDX + 2, P_ = epi_rows(RV);
    if constexpr (t == 2) {
#pragma unroll
      for (int nb2 = 0; nb2 < NB; ++nb2) zq3[2][nb2] = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(crz, cvcol[nb2], crow[2], 0));
    } else if constexpr (!FLUSH) {
      constexpr int row = t < P_ ? t - 3 : (RV >= 4 ? RV - 3 : 0) + (t - P_);      // in-group row, or carry-out row t - P_ (0 or 1)
      const unsigned so = row_offset(x, st.ro + row);
#pragma unroll
      for (int nb2 = 0; nb2 < NB; ++nb2) zq3[t % 3][nb2] = __builtin_bit_cast(u2v, __builtin_amdgcn_raw_buffer_load_b64(rz, vcol[nb2], so, 0));      // out of range: zeros
    }
  }

  // micro-op q of the epilogue of position IDX
  template <int RV, int IDX, int q, bool FLUSH>
  __device__ __forceinline__ void epi(const Item& x, GroupState<RV>& st) {
    constexpr bool mish = ACT == VS_ACT_MISH;
    constexpr int Q0 = DY ? 1 : 0;
    constexpr bool carried = IDX < 3;
    constexpr int r = IDX - 3;                                 // the row inside the group (not for carried rows)
    if constexpr (DY && q == 0) {
      prefetch_z<RV, IDX, FLUSH>(x, st);
#pragma unroll
      for (int nb2 = 0; nb2 < NB; ++nb2) {
        if constexpr (carried) st.mk[nb2] = crow[IDX] != kOob ? ccok[nb2] : 0.f;
        else st.mk[nb2] = (st.ro + r < x.o1) ? st.cok[nb2] : 0.f;
      }
    } else if constexpr (q < Q0 + NB * 2 * NSTAGE) {
      constexpr int v = (q - Q0) / NSTAGE, sg = (q - Q0) % NSTAGE, nb2 = v / 2, pr = v % 2;
      const float acc0 = acc_of<RV, IDX, nb2, 2 * pr>(st), acc1 = acc_of<RV, IDX, nb2, 2 * pr + 1>(st);       // (names: no instruction)
      if constexpr (DY) {
        // dy = da * act'(y), y = z * scale + shift.  Mish'(y) with u = e^y, n = u (u + 2), r = 1 / (n + 2):
        //   tanh(softplus y) = n r,  1 - tanh^2 = 4 (n + 1) r^2 = 4 (u + 1)^2 r^2,  sigmoid = u / (u + 1)
        //   =>  Mish' = r (n + 4 y u (u + 1) r)       (one exp2, one rcp; y clamped at 20, where the expression is 1 to fp32)
        // computed on y log2(e) (folded into the constants).  The sums are of dy and dy * z; flush_stats turns the second
        // into the sum of dy * xhat.
        if constexpr (sg == 0) {
          const unsigned u = zq3[IDX % 3][nb2][pr];
          st.tz.x = i_bf_lo(u); st.tz.y = i_bf_hi(u);
        } else if constexpr (sg == 1) {
          st.ty.x = i_fma(st.tz.x, csc[pr].x, csh[pr].x); st.ty.y = i_fma(st.tz.y, csc[pr].y, csh[pr].y);
        } else if constexpr (mish) {
          if constexpr (sg == 2) { st.ty.x = i_min_20log2e(st.ty.x); st.ty.y = i_min_20log2e(st.ty.y); }
          else if constexpr (sg == 3) st.tu.x = i_exp2(st.ty.x);
          else if constexpr (sg == 4) st.tu.y = i_exp2(st.ty.y);
          else if constexpr (sg == 5) { st.tw.x = i_add2(st.tu.x); st.tw.y = i_add2(st.tu.y); }
          else if constexpr (sg == 6) { st.tn.x = i_mul(st.tu.x, st.tw.x); st.tn.y = i_mul(st.tu.y, st.tw.y); }
          else if constexpr (sg == 7) { st.tw.x = i_add2(st.tn.x); st.tw.y = i_add2(st.tn.y); }
          else if constexpr (sg == 8) st.tr.x = i_rcp(st.tw.x);
          else if constexpr (sg == 9) st.tr.y = i_rcp(st.tw.y);
          else if constexpr (sg == 10) { st.tp.x = i_fma(st.tu.x, st.tu.x, st.tu.x); st.tp.y = i_fma(st.tu.y, st.tu.y, st.tu.y); }          // u (u + 1)
          else if constexpr (sg == 11) { st.tp.x = i_mul(st.tp.x, st.ty.x); st.tp.y = i_mul(st.tp.y, st.ty.y); }
          else if constexpr (sg == 12) { st.tq.x = i_mul_4ln2(st.tr.x); st.tq.y = i_mul_4ln2(st.tr.y); }                                  // y was scaled by log2(e)
          else if constexpr (sg == 13) { st.tn.x = i_fma(st.tp.x, st.tq.x, st.tn.x); st.tn.y = i_fma(st.tp.y, st.tq.y, st.tn.y); }
          else if constexpr (sg == 14) { st.tw.x = i_mul(st.tr.x, st.tn.x); st.tw.y = i_mul(st.tr.y, st.tn.y); }                          // Mish'
          else if constexpr (sg == 15) { st.ty.x = i_mul(acc0, st.tw.x); st.ty.y = i_mul(acc1, st.tw.y); }
          else if constexpr (sg == 16) { st.tp.x = i_mul(st.ty.x, st.mk[nb2]); st.tp.y = i_mul(st.ty.y, st.mk[nb2]); }
          else if constexpr (sg == 17) { a1[pr].x = i_add(a1[pr].x, st.tp.x); a1[pr].y = i_add(a1[pr].y, st.tp.y); }
          else {
            a2[pr].x = i_fma(st.tp.x, st.tz.x, a2[pr].x); a2[pr].y = i_fma(st.tp.y, st.tz.y, a2[pr].y);
            st.y[nb2][pr] = st.ty;
          }
        } else {                       // ReLU (ACT_NONE is not a dy instance)
          if constexpr (sg == 2) st.ty.x = st.ty.x > 0.f ? acc0 : 0.f;
          else if constexpr (sg == 3) st.ty.y = st.ty.y > 0.f ? acc1 : 0.f;
          else if constexpr (sg == 4) { st.tp.x = i_mul(st.ty.x, st.mk[nb2]); st.tp.y = i_mul(st.ty.y, st.mk[nb2]); }
          else if constexpr (sg == 5) { a1[pr].x = i_add(a1[pr].x, st.tp.x); a1[pr].y = i_add(a1[pr].y, st.tp.y); }
          else {
            a2[pr].x = i_fma(st.tp.x, st.tz.x, a2[pr].x); a2[pr].y = i_fma(st.tp.y, st.tz.y, a2[pr].y);
            st.y[nb2][pr] = st.ty;
          }
        }
      } else {
        // out = act(acc * scale + shift).  Mish(y) = y n / (n + 2), n = u (u + 2), u = e^y (y clamped at 20, where n / (n + 2)
        // is 1 to fp32): one exp2 and one rcp per channel
        if constexpr (sg == 0) {
          st.y[nb2][pr].x = i_fma(acc0, csc[pr].x, csh[pr].x); st.y[nb2][pr].y = i_fma(acc1, csc[pr].y, csh[pr].y);
        } else if constexpr (mish) {
          if constexpr (sg == 1) { st.ty.x = i_min_20(st.y[nb2][pr].x); st.ty.y = i_min_20(st.y[nb2][pr].y); }
          else if constexpr (sg == 2) { st.ty.x = i_mul_log2e(st.ty.x); st.ty.y = i_mul_log2e(st.ty.y); }
          else if constexpr (sg == 3) st.tu.x = i_exp2(st.ty.x);
          else if constexpr (sg == 4) st.tu.y = i_exp2(st.ty.y);
          else if constexpr (sg == 5) { st.tw.x = i_add2(st.tu.x); st.tw.y = i_add2(st.tu.y); }
          else if constexpr (sg == 6) { st.tn.x = i_mul(st.tu.x, st.tw.x); st.tn.y = i_mul(st.tu.y, st.tw.y); }
          else if constexpr (sg == 7) { st.tw.x = i_add2(st.tn.x); st.tw.y = i_add2(st.tn.y); }
          else if constexpr (sg == 8) st.tr.x = i_rcp(st.tw.x);
          else if constexpr (sg == 9) st.tr.y = i_rcp(st.tw.y);
          else if constexpr (sg == 10) { st.tn.x = i_mul(st.tn.x, st.tr.x); st.tn.y = i_mul(st.tn.y, st.tr.y); }
          else { st.y[nb2][pr].x = i_mul(st.y[nb2][pr].x, st.tn.x); st.y[nb2][pr].y = i_mul(st.y[nb2][pr].y, st.tn.y); }
        } else if constexpr (ACT == VS_ACT_RELU) {
          st.y[nb2][pr].x = i_max0(st.y[nb2][pr].x); st.y[nb2][pr].y = i_max0(st.y[nb2][pr].y);
        } else if constexpr (STATS) {
          if constexpr (sg == 1) {
            const float m = (int(rowoff_of<RV, IDX>(x, st) != kOob) & int(vcol_of<IDX, nb2>() != kOob)) ? 1.f : 0.f;
            st.tp.x = i_mul(st.y[nb2][pr].x, m); st.tp.y = i_mul(st.y[nb2][pr].y, m);
          } else if constexpr (sg == 2) { a1[pr].x = i_add(a1[pr].x, st.tp.x); a1[pr].y = i_add(a1[pr].y, st.tp.y); }
          else { a2[pr].x = i_fma(st.tp.x, st.y[nb2][pr].x, a2[pr].x); a2[pr].y = i_fma(st.tp.y, st.y[nb2][pr].y, a2[pr].y); }
        }
      }
    } else {
      constexpr int sq = q - Q0 - NB * 2 * NSTAGE, nb2 = sq / 2;
      if constexpr (sq % 2 == 0) {
        st.pk[nb2] = u2v{vs_pack_bf16(st.y[nb2][0].x, st.y[nb2][0].y), vs_pack_bf16(st.y[nb2][1].x, st.y[nb2][1].y)};
      } else {
        if constexpr (carried) __builtin_amdgcn_raw_buffer_store_b64(st.pk[nb2], crout, cvcol[nb2], crow[IDX], 0);
        else __builtin_amdgcn_raw_buffer_store_b64(st.pk[nb2], rout, vcol[nb2], row_offset(x, st.ro + r), 0);              // out of range: dropped
      }
    }
  }

  // ---- which micro-ops ride behind which MFMA -----------------------------------------------------------------------------------------
  // A group's MFMAs are numbered in issue order (slots); its epilogue work is ONE list: the micro-ops of the three carried rows, then
  // of its own rows 0 .. RV - 4.  Carried rows can start at slot 0, row r behind the first MFMA of window row r + KT (its last
  // product was four or more MFMAs earlier).  The list is dealt out at an even pace over the slots that are left -- one micro-op per
  // slot for 5x5 groups of eight rows: 648 micro-ops of the dy form on 800 MFMAs -- limited by what is available; what does not fit
  // (two-row tail groups) runs behind the group's last MFMA.
  static constexpr int nm_of(int RV, int i) { return (i < RV - 1 ? i : RV - 1) - (i - (KT - 1) > 0 ? i - (KT - 1) : 0) + 1; }
  static constexpr int slot_base(int RV, int i) { int s = 0; for (int k = 0; k < i; ++k) s += 2 * NB * KF * nm_of(RV, k); return s; }
  template <int RV>
  struct Plan {
    static constexpr int NSLOT = slot_base(RV, RV + H);
    static constexpr int P_ = epi_rows(RV);
    static constexpr int L = P_ * NMICRO;
    struct Tab { short lo[NSLOT + 1]; };
    static constexpr Tab make() {
      Tab t{};
      int c = 0, s = 0;
      for (int i = 0; i < RV + H; ++i) {
        int rows = i - H;                                  // own rows whose epilogue may have started: r + H + 1 <= i
        if (rows < 0) rows = 0;
        if (rows > P_ - 3) rows = P_ - 3;
        const int avail = (3 + rows) * NMICRO;
        for (int k = 0; k < 2 * NB * KF * nm_of(RV, i); ++k) {
          t.lo[s] = (short)c;
          const int left = NSLOT - s;
          int want = (L - c + left - 1) / left;
          if (want < 1) want = 1;
          int n = avail - c;
          if (n > want) n = want;
          if (n < 0) n = 0;
          c += n;
          ++s;
        }
      }
      t.lo[NSLOT] = (short)c;
      return t;
    }
    static constexpr Tab tab = make();
  };
  template <int RV, bool FLUSH, int M0, int... Ds>
  __device__ __forceinline__ void epis(const Item& x, GroupState<RV>& st, std::integer_sequence<int, Ds...>) {
    (epi<RV, (M0 + Ds) / NMICRO, (M0 + Ds) % NMICRO, FLUSH>(x, st), ...);
  }

  // gstep<RV, GI> is fragment GI of the block: its read two fragments ahead and its MFMAs.  The instruction order is pinned
  // (volatile MFMAs, sched_barrier after every one): hipcc's own order reads a fragment, waits for it and issues two MFMAs, and
  // its sched_group_barrier solver needs 17 minutes for this block.
  template <int RV, int GI, int MM>
  __device__ __forceinline__ void gmfma(const Item& x, GroupState<RV>& st) {
    constexpr int nb = GI % NB, kc = (GI / NB) % 2, df = (GI / (2 * NB)) % KF, i = GI / (2 * NB * KF), ls = GI % (2 * NB * KF);
    constexpr int r_lo = i - (KT - 1) > 0 ? i - (KT - 1) : 0, r_hi = i < RV - 1 ? i : RV - 1;
    constexpr int nm = r_hi - r_lo + 1;                     // MFMAs this fragment feeds
    constexpr int r = r_lo + MM;                            // tap dt = i - r
    // Volatile assembly: the builtin is a pure value to the compiler, which placed it on either side of the stage instructions --
    // the micro-ops then clustered again.  Weights in AGPRs (all 50 fragments: read-only, every use is here), accumulators in
    // VGPRs: the epilogue reads them without v_accvgpr_read (128 per group), and the first product of an accumulator takes the
    // constant 0 as its addend instead of a cleared register (64 v_mov per group).
    constexpr bool first = (i == r) && df == 0 && kc == 0;
    if constexpr (first)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(st.acc[r][nb]) : "a"(wf[(i - r) * KF + df][kc]), "v"(st.bq[GI % 3]));
    else
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(st.acc[r][nb]) : "a"(wf[(i - r) * KF + df][kc]), "v"(st.bq[GI % 3]));
    if constexpr (MM == 0 && GI % 4 == 0 && GI / 4 < G::UNITS) row_unit<GI / 4>(bt);      // the next group's window: one row per unit
    constexpr int sl = slot_base(RV, i) + ls * nm + MM;
    constexpr int m0 = Plan<RV>::tab.lo[sl], m1 = Plan<RV>::tab.lo[sl + 1];
    epis<RV, false, m0>(x, st, std::make_integer_sequence<int, m1 - m0>());
    __builtin_amdgcn_sched_barrier(0);
  }

  template <int RV, int GI, int... MMs>
  __device__ __forceinline__ void gmfmas(const Item& x, GroupState<RV>& st, std::integer_sequence<int, MMs...>) {
    (gmfma<RV, GI, MMs>(x, st), ...);
  }

  template <int RV, int GI>
  __device__ __forceinline__ void gstep(const Item& x, GroupState<RV>& st) {
    constexpr int NG = (RV + H) * KF * 2 * NB;              // fragment reads of the group
    constexpr int i = GI / (2 * NB * KF);
    constexpr int r_lo = i - (KT - 1) > 0 ? i - (KT - 1) : 0, r_hi = i < RV - 1 ? i : RV - 1;
    if constexpr (GI + 2 < NG) st.bq[(GI + 2) % 3] = frag<RV>(st, GI + 2);
    __builtin_amdgcn_sched_barrier(0);
    gmfmas<RV, GI>(x, st, std::make_integer_sequence<int, r_hi - r_lo + 1>());
  }

  template <int RV, int... GIs>
  __device__ __forceinline__ void gsteps(const Item& x, GroupState<RV>& st, std::integer_sequence<int, GIs...>) {
    (gstep<RV, GIs>(x, st), ...);
  }

  template <int RV>
  __device__ __forceinline__ void group(const Item& x, int ro, int buf) {
    static_assert((RV + H) * KF * 2 * NB >= 4 * G::UNITS, "every DMA row unit needs a step");
    using PL = Plan<RV>;
    GroupState<RV> st;
    st.ro = ro;
#pragma unroll
    for (int df = 0; df < KF; ++df)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) st.vb[df][kc] = lds0 + (unsigned)(buf * G::WBUF + boff[df][kc]);
    if constexpr (DY) {
#pragma unroll
      for (int nb2 = 0; nb2 < NB; ++nb2) st.cok[nb2] = (x.strip * STRIP + nb2 * 16 + n < a.F) ? 1.f : 0.f;
    }
    st.bq[0] = frag<RV>(st, 0);
    st.bq[1] = frag<RV>(st, 1);
    gsteps<RV>(x, st, std::make_integer_sequence<int, (RV + H) * KF * 2 * NB>());
    epis<RV, false, PL::tab.lo[PL::NSLOT]>(x, st, std::make_integer_sequence<int, PL::L - PL::tab.lo[PL::NSLOT]>());       // what did not fit (two-row tail groups)
    // Hand the last three rows over to the next group.  The last row's accumulators come from the group's final MFMAs, which the
    // compiler cannot see inside their assembly statements: the wait states between a matrix-pipe write and a VALU read of the same
    // register (11 for an 8-pass instruction) are ours to provide.  The statement takes the accumulators as in-out operands: a bare
    // s_nop does not keep the compiler from moving a read above it (it did, in round 6's first version: the first channel pair of
    // every group's last row came out wrong now and then).
    asm volatile("s_nop 15" : "+v"(st.acc[RV - 1][0]), "+v"(st.acc[RV - 1][NB - 1]));
    constexpr int first = RV >= 4 ? RV - 3 : 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (first + j < RV) {
        crow[j] = row_offset(x, st.ro + first + j);
#pragma unroll
        for (int nb2 = 0; nb2 < NB; ++nb2) cacc[j][nb2] = st.acc[first + j < RV ? first + j : 0][nb2];
      } else {
        crow[j] = kOob;
      }
    }
#pragma unroll
    for (int nb2 = 0; nb2 < NB; ++nb2) {
      cvcol[nb2] = vcol[nb2];
      if constexpr (DY) ccok[nb2] = st.cok[nb2];
    }
    crout = rout;
    if constexpr (DY) {
      crz = rz;
      // the z ring was filled at this group's phase: position PL::P_ of it is the next group's position 0
      constexpr int sh = PL::P_ % 3;
      if constexpr (sh != 0) {
        u2v tmp[3][NB];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int nb2 = 0; nb2 < NB; ++nb2) tmp[k][nb2] = zq3[(k + sh) % 3][nb2];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int nb2 = 0; nb2 < NB; ++nb2) zq3[k][nb2] = tmp[k][nb2];
      }
    }
  }

  // behind the launch's last group the three carried rows still owe their epilogue
  __device__ __forceinline__ void flush_carry() {
    Item x{};
    GroupState<2> st;
    st.ro = 0;
    epis<2, true, 0>(x, st, std::make_integer_sequence<int, 3 * NMICRO>());
  }

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
    // deterministic mode: the workgroups that share a slot add in index order (vs_common.h)
    const unsigned slot = blockIdx.x % VS_BN_STAT_SLOTS, rank_in_slot = blockIdx.x / VS_BN_STAT_SLOTS;
    unsigned* my_turn = a.turn ? a.turn + VS_TURN_SLOT + slot : nullptr;
    vs_turn_begin(my_turn, rank_in_slot);
    if (n == 0) {
      double* dst = a.bn_stats + (size_t)(blockIdx.x % VS_BN_STAT_SLOTS) * 128 + (wave * 16 + g * 4) * 2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double t1 = (double)s1[r], t2 = (double)s2[r];
        if (DY) {                      // sum dy * xhat = invstd * (sum dy * z - mean * sum dy)
          const int ch = wave * 16 + g * 4 + r;
          t2 = (double)a.bn2_invstd[ch] * (t2 - (double)a.bn2_mean[ch] * t1);
        }
        atomicAdd(dst + 2 * r, t1);
        atomicAdd(dst + 2 * r + 1, t2);
      }
    }
    vs_turn_end(my_turn, rank_in_slot, (gridDim.x - slot + VS_BN_STAT_SLOTS - 1) / VS_BN_STAT_SLOTS);
  }
};

template <int KT, int KF, int ACT, bool STATS, bool DY>
__device__ __forceinline__ void nhwc_conv_body(const NhwcConvArgs& a, const unsigned char* smem) {
  using G = Geo<KT, KF>;
  ConvWalk<KT, KF, ACT, STATS, DY> wk(a, (const lds_byte*)smem);

  // prefetch cursor: the group after the one being computed, in the order the compute cursor reaches them
  Item pf;
  int pf_it = (int)blockIdx.x, pf_g = 0;
  bool pf_live = false;
  auto pf_seek = [&]() {           // first item at or after pf_it that decodes
    pf_live = false;
    while (pf_it < a.n_items) {
      if (wk.decode(pf_it, pf)) { pf_g = 0; pf_live = true; return; }
      pf_it += (int)gridDim.x;
    }
  };
  int pbuf = 0;                    // window buffer the next fetched group goes to
  auto pf_begin = [&]() {          // describe the fetch of group (pf, pf_g) -- its rows are issued one per unit inside the group being computed -- and advance the cursor
    wk.begin(wk.bt, pf, pf.o0 + pf_g * R - G::P, pbuf);
    pbuf ^= 1;
    if (++pf_g >= pf.ngroups) { pf_it += (int)gridDim.x; pf_seek(); }
  };
  wk.bt.live = 0;
  pf_seek();
  if (pf_live) { pf_begin(); wk.fetch_all(wk.bt); }

  Item cur;
  int cbuf = 0;                    // window buffer of the group being computed
  for (int it = (int)blockIdx.x; it < a.n_items; it += (int)gridDim.x) {
    if (!wk.decode(it, cur)) continue;
    wk.begin_item(cur);
    for (int gidx = 0; gidx < cur.ngroups; ++gidx) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's rows of the window have landed (and its stores retired)
      __builtin_amdgcn_s_barrier();                         // ... every wave's have; every wave is done with the other buffer
      wk.bt.live = 0;
      if (pf_live) pf_begin();
      const int ro = cur.o0 + gidx * R;
      const int left = cur.o1 - ro;                           // > 0
      if (left > 6) wk.template group<8>(cur, ro, cbuf);
      else if (left > 4) wk.template group<6>(cur, ro, cbuf);
      else if (left > 2) wk.template group<4>(cur, ro, cbuf);
      else wk.template group<2>(cur, ro, cbuf);
      cbuf ^= 1;
    }
  }
  wk.flush_carry();
  wk.flush_stats();
}

// Built without packed-fp32 VALU instructions (tools/epilogue_slot_probe.hip: one v_pk_fma_f32 behind an MFMA costs 17 cycles, two
// scalar v_fma_f32 two): the compiler would otherwise pair the two channels of a stage up again.
template <int KT, int KF, int ACT, bool STATS, bool DY = false>
__global__ __launch_bounds__(256, 1) VS_NO_PACKED_FP32
void nhwc_conv_kernel(NhwcConvArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[Geo<KT, KF>::LDS_BYTES];      // the only LDS object of the kernel
  nhwc_conv_body<KT, KF, ACT, STATS, DY>(a, smem);
}

// w [co][ci][KT][KF] fp32 -> per-wave A fragments: [q][tap][kc][lane][j] = w'[16q + (lane&15)][32kc + 8(lane>>4) + j][tap]
// (transpose_flip: w'[m][k][dt][df] = w[k][m][KT-1-dt][KF-1-df], the data gradient's weights)
__global__ void nhwc_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int KT, int KF, int transpose_flip) {
  const int ntap = KT * KF;
  const int total = 4 * ntap * 2 * 64 * 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, kc = (i >> 9) & 1;
    const int tap = (i >> 10) % ntap, q = (i >> 10) / ntap;
    const int m = 16 * q + (lane & 15), k = 32 * kc + 8 * (lane >> 4) + j;
    const int dt = tap / KF, df = tap - dt * KF;
    const float v = transpose_flip ? w[((size_t)(k * 64 + m) * KT + (KT - 1 - dt)) * KF + (KF - 1 - df)]
                                   : w[((size_t)(m * 64 + k) * KT + dt) * KF + df];
    out[i] = (unsigned short)(vs_pack_bf16(v, 0.f) & 0xffffu);
  }
}

int num_cus() {
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
int launch_conv(NhwcConvArgs a, int act, hipStream_t stream) {
  const int nk_max = (a.T + a.dil - 1) / a.dil;
  const long long base_items = (long long)a.B * a.dil * a.nstrip;
  // enough items for the persistent grid to balance (>= ~16 per workgroup), segments no shorter than 16 rows
  int nseg = (int)((4096 + base_items - 1) / base_items);
  if (nseg > nk_max / 16) nseg = nk_max / 16;
  if (nseg < 1) nseg = 1;
  int seg_rows = (nk_max + nseg - 1) / nseg;
  seg_rows = (seg_rows + R - 1) / R * R;
  nseg = (nk_max + seg_rows - 1) / seg_rows;
  a.nseg = nseg;
  a.seg_rows = seg_rows;
  const long long n_items = base_items * nseg;
  VS_REQUIRE(n_items < (1LL << 30), "nhwc conv: too many work items");
  a.n_items = (int)n_items;
  const int cus = num_cus();
  const dim3 grid((unsigned)(n_items < cus ? n_items : cus)), block(256);
  const size_t lds = 0;   // static LDS: Geo::LDS_BYTES
  const bool stats = a.bn_stats != nullptr;
#define VS_NHWC_LAUNCH3(A, S, D) hipLaunchKernelGGL((nhwc_conv_kernel<KT, KF, A, S, D>), grid, block, lds, stream, a)
#define VS_NHWC_LAUNCH(A, S) VS_NHWC_LAUNCH3(A, S, false)
  if (a.z2) {            // data gradient with the activation-derivative epilogue: act = the activation whose derivative is taken
    VS_REQUIRE(stats && a.bn2_scale && a.bn2_shift && a.bn2_mean && a.bn2_invstd, "nhwc conv: the dy epilogue needs statistics slots and BatchNorm constants");
    if (act == VS_ACT_MISH) VS_NHWC_LAUNCH3(VS_ACT_MISH, false, true);
    else if (act == VS_ACT_RELU) VS_NHWC_LAUNCH3(VS_ACT_RELU, false, true);
    else VS_REQUIRE(false, "nhwc conv: dy epilogue for activation %d", act);
  } else if (stats) {
    VS_REQUIRE(act == VS_ACT_NONE, "nhwc conv: fused statistics go with no activation");
    VS_NHWC_LAUNCH(VS_ACT_NONE, true);
  } else if (act == VS_ACT_NONE) VS_NHWC_LAUNCH(VS_ACT_NONE, false);
  else if (act == VS_ACT_MISH) VS_NHWC_LAUNCH(VS_ACT_MISH, false);
  else if (act == VS_ACT_RELU) VS_NHWC_LAUNCH(VS_ACT_RELU, false);
  else VS_REQUIRE(false, "nhwc conv: unsupported activation %d", act);
#undef VS_NHWC_LAUNCH
#undef VS_NHWC_LAUNCH3
  VS_LAUNCH_CHECK();
  return 0;
}

}  // namespace

size_t vs_nhwc_packed_bytes(int KT, int KF) { return (size_t)64 * 64 * KT * KF * 2; }

int vs_nhwc_pack_impl(const float* w, void* packed, int KT, int KF, int transpose_flip, hipStream_t stream) {
  VS_REQUIRE(w && packed, "nhwc pack: NULL argument");
  VS_REQUIRE((KT == 5 && KF == 5) || (KT == 7 && KF == 1), "nhwc pack: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  hipLaunchKernelGGL(nhwc_pack_kernel, dim3(64), dim3(256), 0, stream, w, reinterpret_cast<unsigned short*>(packed), KT, KF, transpose_flip);
  VS_LAUNCH_CHECK();
  return 0;
}

// out[b][t][f][co] = act(scale[co] * sum_{ci,dt,df} w[co][ci][dt][df] in[b][t + (dt-KT/2) dil][f + df - KF/2][ci] + shift[co]),
// zero padding; in/out channels-last bf16.  bn_stats: per-channel {sum, sum of squares} of the (fp32) outputs,
// [VS_BN_STAT_SLOTS][64][2] doubles the caller zeroed (act must be VS_ACT_NONE then).
int vs_nhwc_conv_impl(const void* in, const void* packed, const float* scale, const float* shift, void* out,
                      int B, int T, int F, int KT, int KF, int dil, int act, double* bn_stats, hipStream_t stream) {
  VS_REQUIRE(in && packed && scale && shift && out, "nhwc conv: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "nhwc conv: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((long long)T * F * 128 < 0x7F000000LL, "nhwc conv: an utterance of %d x %d pixels does not fit the 31-bit offsets of the epilogue", T, F);
  VS_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(packed) & 15) == 0, "nhwc conv: buffers must be 16-byte aligned");
  NhwcConvArgs a{reinterpret_cast<const unsigned short*>(in), reinterpret_cast<const unsigned short*>(packed), scale, shift,
                 reinterpret_cast<unsigned short*>(out), bn_stats, nullptr, nullptr, nullptr, nullptr, nullptr,
                 B, T, F, dil, (F + STRIP - 1) / STRIP, 1, 0, 0, g_vs_turn};
  if (KT == 5 && KF == 5) return launch_conv<5, 5>(a, act, stream);
  if (KT == 7 && KF == 1) return launch_conv<7, 1>(a, act, stream);
  VS_REQUIRE(false, "nhwc conv: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  return -1;
}

// The data gradient of a layer fused with the first half of the BatchNorm + activation backward of the layer below it:
// da = conv(dz, packed (transposed + flipped weights)) is turned into dy = da * act'(z * bn_scale + bn_shift) in the
// epilogue, dy is stored (bf16) and bn_stats receives per-channel {sum dy, sum dy * xhat} ([VS_BN_STAT_SLOTS][64][2]
// doubles the caller zeroed).  z: the lower layer's conv + bias output, channels-last bf16.
int vs_nhwc_conv_dy_impl(const void* dz, const void* packed, void* dy, const void* z, int act,
                         const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd, double* bn_stats,
                         int B, int T, int F, int KT, int KF, int dil, hipStream_t stream) {
  VS_REQUIRE(dz && packed && dy && z && bn_scale && bn_shift && bn_mean && bn_invstd && bn_stats, "nhwc conv dy: NULL argument");
  VS_REQUIRE(B > 0 && T > 0 && F > 0 && dil > 0, "nhwc conv dy: bad shape B=%d T=%d F=%d dil=%d", B, T, F, dil);
  VS_REQUIRE((long long)T * F * 128 < 0x7F000000LL, "nhwc conv dy: an utterance of %d x %d pixels does not fit the 31-bit offsets of the epilogue", T, F);
  VS_REQUIRE((reinterpret_cast<uintptr_t>(dz) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
             (reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(z) & 7) == 0, "nhwc conv dy: buffer alignment");
  NhwcConvArgs a{reinterpret_cast<const unsigned short*>(dz), reinterpret_cast<const unsigned short*>(packed), bn_scale, bn_shift,
                 reinterpret_cast<unsigned short*>(dy), bn_stats, reinterpret_cast<const unsigned short*>(z), bn_scale, bn_shift, bn_mean, bn_invstd,
                 B, T, F, dil, (F + STRIP - 1) / STRIP, 1, 0, 0, g_vs_turn};
  if (KT == 5 && KF == 5) return launch_conv<5, 5>(a, act, stream);
  if (KT == 7 && KF == 1) return launch_conv<7, 1>(a, act, stream);
  VS_REQUIRE(false, "nhwc conv dy: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  return -1;
}
