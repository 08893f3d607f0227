// bf16 GEMM of the VS_MATH_BF16 configuration: the three large LSTM contractions of the path
//   xg    = feat  @ [W_ih; W_ih_reverse]^T   (models/voicesplit/model.py:82; 19264 x 3200 x 4808 at B=64)
//   dfeat = dxg   @ [W_ih; W_ih_reverse]     (its data gradient;              19264 x 4808 x 3200)
//   dW_ih = dxg^T @ feat                     (its weight gradient;             3200 x 4808 x 19264)
// over THREE bf16 arrays, each used in two roles: feat_bf [M][Kp], dxg_bf [M][8H], wih_bf [8H][Kp].  An operand is
// either K-contiguous ("row form": element (i, k) at i*ld + k -- a fragment of v_mfma_f32_16x16x32_bf16 is 16
// contiguous bytes: ds_read_b128) or K-major ("col form": element (i, k) at k*ld + i -- the fragment is a transpose of
// what lies in memory: two ds_read_b64_tr_b16, as in wgrad_nhwc.hip).  So xg is row x row, dfeat row x col, dW_ih
// col x col, and no transposed copy of anything is ever written.
//
// Structure: persistent workgroups (one per CU, 4 waves = 2 x 2, wave tile 128 x 128 = 64 accumulator tiles = 256
// registers, workgroup tile 256 x 256, K step 64), tiles walked in bands of 8 tile rows per XCD (gemm_f16x3.hip's
// map); both operand tiles of a K step go L2 -> LDS by LDS-DMA into the other of two stages while the current one is
// multiplied (a step is 128 MFMAs per wave = 2048 matrix-pipe cycles: the DMA has that long to land; one
// s_waitcnt vmcnt(0) + one raw s_barrier per step); XOR swizzles on the source address make the fragment reads
// conflict-free; per K step a wave issues 32 fragment reads and 128 MFMAs in a pinned order (reads one k-half
// ahead).  The 256 x 256 tile is what the L2 allows: a 128 x 256 tile (first version) moved 48 KB per 1024 pipe
// cycles and workgroup -- 25 TB/s over the chip at full MFMA rate, beyond the L2s -- and measured 0.93-1.0 ms
// (600-640 TF) on the three contractions; this one moves 64 KB per 2048 cycles.  fp32 accumulate, fp32 output
// (+ optional per-row-group bias, or accumulate into C).
#include <stdlib.h>

#include <type_traits>

#include "vs_internal.h"

namespace {

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) const u4v lds_u4v;

__device__ u4v g_gemm_zero_page[4];

constexpr int TM = 256, TN = 256, BK = 64;      // workgroup tile, K step
constexpr int A_BYTES = TM * BK * 2;            // 32 KiB
constexpr int B_BYTES = TN * BK * 2;            // 32 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // 64 KiB
constexpr int NSTAGE = 2;
constexpr int WA = 8, WB = 8;                   // 16 x 16 accumulator tiles of a wave: 128 x 128

struct GemmBf16Args {
  const unsigned short* A; int lda;   // row form: [M][lda], col form: [K][lda]
  const unsigned short* B; int ldb;   // row form: [N][ldb], col form: [K][ldb]
  float* C; int ldc;
  float* C2; int split_m;             // rows >= split_m go to C2 + (m - split_m) * ldc (the two directions of dW_ih); else NULL
  int M, N, K;                        // K = number of valid k (zero-padded operands may be read beyond it up to the next BK)
  const float* rowbias; int ldrb, group;
  int accumulate;
  int tiles_m, tiles_n, band;
  const float* gate; int ldg;         // C = the product where gate[m][n] > 0, else 0 (the relu mask of a backward contraction); or NULL
  int vec_ok;                         // C (C2, rowbias) 16-byte aligned with leading dimensions that are multiples of 4: 16-byte epilogue accesses
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ vs_bf16x8 frag_of(s4v lo, s4v hi) {
  const s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(vs_bf16x8, v);
}
__device__ __forceinline__ s4v ds_read_tr16(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(uintptr_t)addr);
}

// ---- operand tile images in LDS ---------------------------------------------------------------------------------
// row form, R rows x 64 k: 128 bytes per row; LDS piece q (16 bytes) of row r holds k-piece q ^ ((r >> 1) & 7)
//   (the conv kernel's pixel image: 16 rows r..r+15 at the same k-piece hit 16 different 16-byte bank slots)
// col form, 64 k x R rows-of-the-matrix: R*2 bytes per k line; LDS 32-byte unit u of line k holds source unit
//   u ^ ((k & 3) | ((k >> 3) & 1) << 2): the 8 lines {c..c+3, c+8..c+11} a transposing read touches differ in it
template <bool KMAJOR, int R>
struct Operand {
  // DMA of one K step: R*64 halves = R*8 pieces = R/8 chunks of 1 KiB; chunk ch = 4 c + wave, piece e = 64 ch + lane
  // lands at LDS byte 16 e.  What depends on the lane is computed ONCE (LaneDma): with R = 256 the swizzle term of a
  // piece does not depend on c (row form) or only on its parity (col form), so chunk c is the lane's base offset plus a
  // wave-uniform multiple of the leading dimension -- a handful of registers instead of sixteen hoisted address pairs.
  static_assert(R == 256, "the lane decomposition below is for 256-row operand tiles");
  static constexpr int CHUNKS = R / 8 / 4;                 // per wave
  struct LaneDma { unsigned off[2]; int line; int col[2]; };
  __device__ static __forceinline__ LaneDma lane_dma(int ld, int wave, int lane) {
    LaneDma L;
    if (!KMAJOR) {
      const int r = wave * 8 + (lane >> 3), q = lane & 7;  // + 32 c
      const int kp = q ^ ((r >> 1) & 7);                   // (32 c >> 1) & 7 == 0
      L.line = r;
      L.col[0] = L.col[1] = kp * 8;
      L.off[0] = L.off[1] = (unsigned)(r * ld + kp * 8) * 2u;
    } else {
      const int k = 2 * wave + (lane >> 5), q = lane & 31; // + 8 c: (k & 3) unchanged, bit 3 of k = c & 1
      L.line = k;
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int src = q ^ (((k & 3) | (par << 2)) << 1);
        L.col[par] = src * 8;
        L.off[par] = (unsigned)(k * ld + src * 8) * 2u;
      }
    }
    return L;
  }
  // chunk c (0 .. CHUNKS-1) of this wave for the tile at (r0, k0): one LDS-DMA instruction
  __device__ static __forceinline__ void issue_chunk(const LaneDma& L, const unsigned short* base, int ld, int rows /* extent along R */, int kext,
                                                     int r0, int k0, unsigned lds_dst, int wave, int c) {
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_gemm_zero_page);
    // tile origin (wave-uniform): row form (r0, k0), col form (k0, r0)
    const unsigned char* origin = reinterpret_cast<const unsigned char*>(base) + (KMAJOR ? ((long long)k0 * ld + r0) : ((long long)r0 * ld + k0)) * 2;
    bool ok;
    unsigned off;
    if (!KMAJOR) {
      ok = (r0 + 32 * c + L.line < rows) & (k0 + L.col[0] < kext);
      off = L.off[0] + (unsigned)(32 * c) * (unsigned)ld * 2u;
    } else {
      ok = (k0 + 8 * c + L.line < kext) & (r0 + L.col[c & 1] < rows);
      off = L.off[c & 1] + (unsigned)(8 * c) * (unsigned)ld * 2u;
    }
    const unsigned char* src_p = ok ? origin + off : zp;
    glds16(src_p, (unsigned)__builtin_amdgcn_readfirstlane(lds_dst + (unsigned)((c * 4 + wave) * 1024)));
  }
  // The same chunk through a buffer descriptor (the conv kernels' form, conv_nhwc.hip): the descriptor covers the operand
  // from the tile's origin at this K step to the end of its valid rows (row form) / k lines (col form), so rows / lines
  // beyond the matrix are out of range and arrive as zeros without a compare or a select; what depends on the lane is the
  // ONE offset register computed at kernel start (two in col form, by chunk parity), what depends on the chunk is a
  // scalar offset and the LDS address in M0.  ~6 instructions per chunk where the pointer form above needs ~15 -- with 16
  // chunks per wave and step beside 128 MFMAs that was the kernel's bound (profiles/r03_gemm_l2_prefetch.md).
  __device__ static __forceinline__ u4v tile_desc(const unsigned short* base, int ld, int rows, int kext, int r0, int k0, bool live) {
    unsigned long long p;
    long long rec;
    if (!KMAJOR) {
      p = reinterpret_cast<unsigned long long>(base) + ((unsigned long long)r0 * (unsigned)ld + (unsigned)k0) * 2ull;
      rec = ((long long)(rows - r0) * ld - k0) * 2;
      if (k0 >= kext) rec = 0;
    } else {
      p = reinterpret_cast<unsigned long long>(base) + ((unsigned long long)k0 * (unsigned)ld + (unsigned)r0) * 2ull;
      rec = ((long long)(kext - k0) * ld - r0) * 2;
      if (r0 >= rows) rec = 0;
    }
    if (!live || rec < 0) rec = 0;
    if (rec > 0xFFFFFFF0ll) rec = 0xFFFFFFF0ll;
    return u4v{(unsigned)p, (unsigned)(p >> 32) & 0xffffu, (unsigned)rec, 0x00020000u};
  }
  // byte distance between consecutive chunks of a wave (wave-uniform): 32 rows (row form) / 8 k lines (col form)
  __device__ static __forceinline__ unsigned chunk_pitch(int ld) { return (unsigned)(KMAJOR ? 8 : 32) * (unsigned)ld * 2u; }
  __device__ static __forceinline__ void issue_chunk_desc(const LaneDma& L, u4v desc, unsigned pitch, unsigned lds_dst, int wave, int c) {
    const unsigned voff = KMAJOR ? L.off[c & 1] : L.off[0];
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(pitch * (unsigned)c));
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_dst + (unsigned)((c * 4 + wave) * 1024)));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(desc), "s"(dst), "s"(soff) : "memory");
  }
  // Fragment of the 16-row block starting at row rb (of the tile), k-half kh (32 k), for lane (i = lane & 15, g = lane >> 4).
  // The lane-dependent part of the address is ONE register (frag_base, + the stage offset); rb and kh enter as an XOR
  // with a constant and an immediate offset, so that the 32 fragment addresses of a step are not 32 live registers:
  //   row form: r = rb + i, swizzle (r >> 1) & 7 = (i >> 1) & 7 (rb is a multiple of 16); k-half 1 = address ^ 64
  //   col form: k = 32 kh + 8 g + 4 hf + i/4, u(k) = (i/4 & 3) | (g & 1) << 2 does not depend on kh, hf; unit = (rb/16) ^ u
  __device__ static __forceinline__ unsigned frag_base(int lane) {
    const int i = lane & 15, g = lane >> 4;
    if (!KMAJOR) return (unsigned)(i * 128 + ((g ^ ((i >> 1) & 7)) << 4));
    const int u = ((i >> 2) & 3) | ((g & 1) << 2);
    return (unsigned)((8 * g + (i >> 2)) * (R * 2) + (u << 5) + (i & 3) * 8);
  }
  __device__ static __forceinline__ vs_bf16x8 frag(unsigned base /* LDS address of the image + frag_base */, int rb, int kh) {
    if (!KMAJOR) {
      return __builtin_bit_cast(vs_bf16x8, *(lds_u4v*)(uintptr_t)((base ^ (unsigned)(kh << 6)) + (unsigned)(rb * 128)));
    } else {
      const unsigned a = base ^ (unsigned)((rb >> 4) << 5);
      return frag_of(ds_read_tr16(a + (unsigned)((kh * 32) * (R * 2))), ds_read_tr16(a + (unsigned)((kh * 32 + 4) * (R * 2))));
    }
  }
};

__device__ __forceinline__ bool tile_of(const GemmBf16Args& g, int t, int& tm, int& tn) {
  if (t >= g.tiles_m * g.tiles_n) return false;
  const int band_tiles = g.band * g.tiles_n;
  const int band = t / band_tiles;
  const int r = t - band * band_tiles;
  const int rows = min(g.band, g.tiles_m - band * g.band);
  tn = r / rows;
  tm = band * g.band + (r - tn * rows);
  return true;
}

// DM: where the 16 LDS-DMA instructions of the next stage sit inside a step: 0 = all behind the barrier, 1 = two in front
// of each of the first 8 MFMA rows, 2 = one in front of each of the 16 MFMA rows
template <bool AK, bool BK_, int DM>
__global__ __launch_bounds__(256, 1)
void gemm_bf16_kernel(GemmBf16Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE_BYTES];
  using OA = Operand<AK, TM>;
  using OB = Operand<BK_, TN>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;                 // wave tile: rows [128 wm, +128), columns [128 wn, +128)
  const unsigned lds0 = (unsigned)(uintptr_t)(const lds_byte*)smem;
  const int nk = (g.K + BK - 1) / BK;

  // flat sequence of (tile, k step) of this workgroup; XCD x (= blockIdx % 8) owns a contiguous range of the banded list
  const int per = (int)(gridDim.x >> 3);
  const int ntiles = g.tiles_m * g.tiles_n;
  const int tpx = (ntiles + 7) / 8;                         // tiles per XCD
  const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
  auto tile_id = [&](int j) { return xcd * tpx + slot + j * per; };     // j-th tile of this workgroup
  auto tile_ok = [&](int j) { const int t = tile_id(j); return slot + j * per < tpx && t < ntiles; };

  // prefetch cursor
  int pj = 0, pk = 0, ptm = 0, ptn = 0;
  bool plive = tile_ok(0) && tile_of(g, tile_id(0), ptm, ptn);
  int pstage = 0;
  const unsigned fbA = OA::frag_base(lane), fbB = OB::frag_base(lane);
  const typename OA::LaneDma la = OA::lane_dma(g.lda, wave, lane);
  const typename OB::LaneDma lb = OB::lane_dma(g.ldb, wave, lane);
  // The DMA of a stage is 16 instructions per wave (8 A chunks, 8 B chunks) with ~15 scalar / vector instructions of
  // address arithmetic each: issued in one go behind the barrier they cost ~1000 cycles of a 2048-cycle step with the
  // matrix pipe idle, so they are spread over the FIRST half of the step -- two chunks in front of each of its first 8
  // MFMA rows; the second half of the step is their time to land.  (Spread over all 16 rows, the col x col contraction,
  // whose panels stream from HBM, waited for the late chunks: 0.63 -> 1.63 ms.)
  // descriptors of the two operand tiles the prefetch cursor points at (wave-uniform: scalar registers), rebuilt when it moves
  const unsigned pitchA = OA::chunk_pitch(g.lda), pitchB = OB::chunk_pitch(g.ldb);
  u4v dA = OA::tile_desc(g.A, g.lda, g.M, g.K, ptm * TM, 0, plive), dB = OB::tile_desc(g.B, g.ldb, g.N, g.K, ptn * TN, 0, plive);
  auto pf_chunk = [&](int slot /* 0..15 */, bool live) {     // !live (end of the job): empty descriptors, every piece lands as zeros, no branch
    (void)live;
    const unsigned dst = lds0 + (unsigned)(pstage * STAGE_BYTES);
    if (slot < 8) OA::issue_chunk_desc(la, dA, pitchA, dst, wave, slot);
    else OB::issue_chunk_desc(lb, dB, pitchB, dst + A_BYTES, wave, slot - 8);
  };
  auto pf_done = [&]() {
    pstage ^= 1;
    if (++pk == nk) {
      pk = 0;
      ++pj;
      plive = tile_ok(pj) && tile_of(g, tile_id(pj), ptm, ptn);
    }
    dA = OA::tile_desc(g.A, g.lda, g.M, g.K, ptm * TM, pk * BK, plive);
    dB = OB::tile_desc(g.B, g.ldb, g.N, g.K, ptn * TN, pk * BK, plive);
  };
  if (plive) {
#pragma unroll
    for (int c = 0; c < 16; ++c) pf_chunk(c, true);
    pf_done();
  }

  int cstage = 0;
  for (int j = 0; tile_ok(j); ++j) {
    int tm, tn;
    if (!tile_of(g, tile_id(j), tm, tn)) break;
    f32x4 acc[WA][WB];
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
      for (int b = 0; b < WB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < nk; ++ks) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the stage have landed (issued one step ago)
      __builtin_amdgcn_s_barrier();                         // ... every wave's have; every wave is done reading the other stage
      const bool pf_now = plive;                            // the next step's tiles (possibly of the next output tile) go into it
      if (DM == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) pf_chunk(c, pf_now);
      }
      unsigned fa = lds0 + (unsigned)(cstage * STAGE_BYTES) + fbA, fb = lds0 + (unsigned)(cstage * STAGE_BYTES + A_BYTES) + fbB;
      asm volatile("" : "+v"(fa), "+v"(fb));               // opaque: keeps the compiler from hoisting 32 derived addresses per stage out of the loop
      // B fragments of a k-half stay in registers for its 8 accumulator rows; the A fragment of the next row and the B
      // fragments of the other k-half are read while a row's 8 MFMAs issue
      vs_bf16x8 bf[2][WB], afc, afn;
#pragma unroll
      for (int b = 0; b < WB; ++b) bf[0][b] = OB::frag(fb, wn * 128 + b * 16, 0);
      afc = OA::frag(fa, wm * 128, 0);
      afn = afc;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int a = 0; a < WA; ++a) {
          if (a + 1 < WA) afn = OA::frag(fa, wm * 128 + (a + 1) * 16, kh);
          else if (kh == 0) afn = OA::frag(fa, wm * 128, 1);
          if (kh == 0) bf[1][a] = OB::frag(fb, wn * 128 + a * 16, 1);
          if (DM == 1 && kh == 0) { pf_chunk(a, pf_now); pf_chunk(8 + a, pf_now); }
          if (DM == 2) pf_chunk(kh * 8 + a, pf_now);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < WB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afc, bf[kh][b], acc[a][b], 0, 0, 0);
          afc = afn;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (pf_now) pf_done();
      cstage ^= 1;
    }
    // epilogue: lane holds C[m = 16a + 4 (lane>>4) + r][n = 16b + (lane&15)]
    const int i = lane & 15, gq = lane >> 4;
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = tm * TM + wm * 128 + a * 16 + 4 * gq + r;
        if (m >= g.M) continue;
        float* crow = (g.C2 && m >= g.split_m) ? g.C2 + (size_t)(m - g.split_m) * g.ldc : g.C + (size_t)m * g.ldc;
        const float* rb = g.rowbias ? g.rowbias + (size_t)(m / g.group) * g.ldrb : nullptr;
#pragma unroll
        for (int b = 0; b < WB; ++b) {
          const int n = tn * TN + wn * 128 + b * 16 + i;
          if (n < g.N) {
            float v = acc[a][b][r];
            if (rb) v += rb[n];
            if (g.accumulate) v += crow[n];
            if (g.gate) v = g.gate[(size_t)m * g.ldg + n] > 0.f ? v : 0.f;
            crow[n] = v;
          }
        }
      }
  }
}

// ---- round 4: the same kernel with the side work INSIDE the MFMA rows ---------------------------------------------
// What bounded gemm_bf16_kernel above was found in its ISA (tools/isa_loop_stats.py): a K step is 16 rows of 8 back-to-back
// MFMAs with ALL the side work of a row -- two fragment reads, two DMA chunks of 6 instructions each -- issued between two
// rows.  A wave issues in order, one instruction per ~4 cycles: while those ~16 instructions issue the
// matrix pipe finishes the row's last MFMA after 16 cycles and then idles ~50; 128 of 180 cycles per row = the measured
// 0.42-0.46 busy fraction.  Between two MFMAs of a row, on the other hand, the pipe is busy 16 cycles and the wave needs 4
// to issue the next one: up to two more instructions fit there for free.  So:
//  * every MFMA is followed by ONE slot of side work (sched_barrier after each): fragment read, "M0 <- chunk's LDS address",
//    the chunk's buffer_load ... lds, a scalar offset update -- a DMA chunk is 3 instructions in 3 slots (was 7-8 in one
//    place): M0 is set by s_add_i32 straight from the stage base (never saved / restored: nothing else in the kernel uses
//    it), the scalar offset runs (one s_add per chunk), the two tile descriptors advance by a constant per K step (3 scalar
//    instructions each, in slots of the second k-half) and are rebuilt only when the prefetch cursor moves to a new tile;
//  * the per-step bubble is gone: the step's barrier sits in front of the LAST row -- by then every fragment of the stage
//    has been read into registers and the next stage's DMA was issued more than a thousand cycles ago -- and the first nine
//    fragment reads of the next step (8 B + 1 A, into the registers of the finished k-half) ride on the last row's MFMAs.
// An accumulator tile leaves the AGPRs HERE, as four explicit reads: left to the compiler, the copies of all 64 tiles were hoisted to the
// loop exit and ~90 of the 256 registers they needed went to scratch around the epilogue of every tile.
__device__ __forceinline__ f32x4 acc_read(const f32x4& t) {
  f32x4 v;
  asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
               : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]) : "a"(t[0]), "a"(t[1]), "a"(t[2]), "a"(t[3]));
  return v;
}

// DR: the 16 DMA chunks of a step are issued in the first DR rows (8: chunk a of both operands in row a; 4: chunks 2a, 2a+1 in row a
// -- 4 more rows for the lines to arrive before the step's barrier)
// ABL (timing ablations, built with -DVS_ABLATION only; results are meaningless): 1 = no fragment reads inside the K loop (the MFMAs
// run on whatever the registers hold), 2 = no per-step wait + barrier, 3 = both
template <bool AK, bool BK_, int DR, int ABL = 0>
__global__ __launch_bounds__(256, 1)
void gemm_bf16_il_kernel(GemmBf16Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE_BYTES];
  using OA = Operand<AK, TM>;
  using OB = Operand<BK_, TN>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(const lds_byte*)smem;
  const int nk = (g.K + BK - 1) / BK;

  const int per = (int)(gridDim.x >> 3);
  const int ntiles = g.tiles_m * g.tiles_n;
  const int tpx = (ntiles + 7) / 8;
  const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
  auto tile_id = [&](int j) { return xcd * tpx + slot + j * per; };
  auto tile_ok = [&](int j) { const int t = tile_id(j); return slot + j * per < tpx && t < ntiles; };

  int pj = 0, pk = 0, ptm = 0, ptn = 0;
  bool plive = tile_ok(0) && tile_of(g, tile_id(0), ptm, ptn);
  int pstage = 0;
  // fragment base of this lane INCLUDING the wave's tile origin (rows 128 wm / columns 128 wn of the workgroup tile): what is
  // left per fragment is an immediate offset (row form: + 2048 a) or an XOR with an inline constant (col form: ^ (a << 5))
  const unsigned fbA = AK ? (OA::frag_base(lane) ^ (unsigned)((wm * 8) << 5)) : (OA::frag_base(lane) + (unsigned)(wm * 128 * 128));
  const unsigned fbB = BK_ ? (OB::frag_base(lane) ^ (unsigned)((wn * 8) << 5)) : (OB::frag_base(lane) + (unsigned)(wn * 128 * 128));
  const typename OA::LaneDma la = OA::lane_dma(g.lda, wave, lane);
  const typename OB::LaneDma lb = OB::lane_dma(g.ldb, wave, lane);
  const unsigned pitchA = OA::chunk_pitch(g.lda), pitchB = OB::chunk_pitch(g.ldb);
  // per K step the tile origin moves by BK elements along k: BK * 2 bytes (row form) or BK lines (col form)
  const unsigned kstepA = AK ? (unsigned)BK * (unsigned)g.lda * 2u : (unsigned)BK * 2u;
  const unsigned kstepB = BK_ ? (unsigned)BK * (unsigned)g.ldb * 2u : (unsigned)BK * 2u;
  // g.accumulate < 0 (VS_OPT_ABLATION = 9 in a VS_ABLATION build, timing only): every descriptor is empty -- the DMA instructions issue and
  // zero-fill the stage without touching memory: what is left is the kernel's time without the memory side
  const bool feed = g.accumulate >= 0;
  u4v dA = OA::tile_desc(g.A, g.lda, g.M, g.K, ptm * TM, 0, plive && feed), dB = OB::tile_desc(g.B, g.ldb, g.N, g.K, ptn * TN, 0, plive && feed);
  unsigned soffA = 0, soffB = 0;                                     // running scalar offset of the next chunk (wave-uniform)
  unsigned dstA = lds0 + (unsigned)(wave * 1024);                    // LDS address of this wave's chunk 0 in the prefetch stage
  auto dma_load = [&](unsigned voff, const u4v& desc, unsigned soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff), "s"(desc), "s"(soff) : "memory");
  };
  auto adv = [&](u4v& d, unsigned step) {                            // 48-bit base += step (add + add-with-carry), num_records -= step
    const unsigned long long b = (((unsigned long long)d[1] << 32) | d[0]) + step;
    d[0] = (unsigned)b;
    d[1] = (unsigned)(b >> 32);
    d[2] -= step;
  };
  // the step's last piece of cursor bookkeeping (outside the MFMA rows: a handful of scalar instructions; the descriptor
  // rebuild only when the cursor moves to another tile, once per nk steps)
  auto pf_done = [&]() {
    pstage ^= 1;
    dstA = lds0 + (unsigned)(pstage * STAGE_BYTES) + (unsigned)(wave * 1024);
    if (++pk == nk) {
      pk = 0;
      ++pj;
      plive = tile_ok(pj) && tile_of(g, tile_id(pj), ptm, ptn);
      dA = OA::tile_desc(g.A, g.lda, g.M, g.K, ptm * TM, 0, plive && feed);
      dB = OB::tile_desc(g.B, g.ldb, g.N, g.K, ptn * TN, 0, plive && feed);
    }
  };
  if (plive) {                                                       // cold start: the first stage in one go
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dstA + (unsigned)(c * 4096)) : "memory");
      dma_load(AK ? la.off[c & 1] : la.off[0], dA, soffA);
      soffA += pitchA;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(dstA + (unsigned)(A_BYTES + c * 4096)) : "memory");
      dma_load(BK_ ? lb.off[c & 1] : lb.off[0], dB, soffB);
      soffB += pitchB;
    }
    soffA = soffB = 0;
    adv(dA, kstepA);
    adv(dB, kstepB);
    pf_done();
  }

  int cstage = 0;
  vs_bf16x8 bf[2][WB], afc, afn;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int j = 0; tile_ok(j); ++j) {
    int tm, tn;
    if (!tile_of(g, tile_id(j), tm, tn)) break;
    f32x4 acc[WA][WB];
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
      for (int b = 0; b < WB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      // the tile's first fragments.  (The last K step of the previous tile has read them already -- its row 15 does not know it is
      // the last -- but re-reading them here ends their live range in front of the epilogue, which needs the registers: with the
      // 36 of them live across it the allocator spilled ~90 registers per tile.  The stage is complete: the step barrier.)
      unsigned fa = lds0 + (unsigned)(cstage * STAGE_BYTES) + fbA, fb = lds0 + (unsigned)(cstage * STAGE_BYTES + A_BYTES) + fbB;
      asm volatile("" : "+v"(fa), "+v"(fb));
#pragma unroll
      for (int b = 0; b < WB; ++b) bf[0][b] = OB::frag(fb, b * 16, 0);
      afc = OA::frag(fa, 0, 0);
      afn = afc;
    }
    for (int ks = 0; ks < nk; ++ks) {
      // !plive (end of the job): empty descriptors stay empty.  (plive is wave-uniform but lives in a vector register: made
      // scalar explicitly, or the descriptors it touches would leave the scalar registers the DMA instruction needs them in)
      const unsigned lv = (unsigned)__builtin_amdgcn_readfirstlane(plive ? 1 : 0);
      const unsigned stepA = (feed ? lv : 0u) * kstepA, stepB = (feed ? lv : 0u) * kstepB;
      unsigned fa = lds0 + (unsigned)(cstage * STAGE_BYTES) + fbA, fb = lds0 + (unsigned)(cstage * STAGE_BYTES + A_BYTES) + fbB;
      unsigned fan = lds0 + (unsigned)((cstage ^ 1) * STAGE_BYTES) + fbA, fbn = lds0 + (unsigned)((cstage ^ 1) * STAGE_BYTES + A_BYTES) + fbB;
      asm volatile("" : "+v"(fa), "+v"(fb), "+v"(fan), "+v"(fbn));   // opaque: no hoisting of the derived addresses out of the loop
      const unsigned dstB = dstA + (unsigned)A_BYTES;
      __builtin_amdgcn_sched_barrier(0);
      auto row = [&](auto RC) {
        constexpr int R = decltype(RC)::value;
        constexpr int kh = R >> 3, a = R & 7;
        if (R == 15) {
          // every fragment of this stage is in registers (the A fragment of this row was read one row ago), this wave's pieces
          // of the next stage were issued in rows 0..7: after the barrier the next stage is complete and this one is free
          if (!(ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int b = 0; b < WB; ++b) {
          // (inline assembly with the accumulator tied to an AGPR: given the builtin, the register allocator spread the 64
          // accumulator tiles over both register files and moved 280 registers between them every K step)
          // operand roles swapped (the B fragment goes in as the instruction's A): the lane then holds C[m = 16a + (lane & 15)]
          // [n = 16b + 4 (lane >> 4) + 0..3] -- four CONSECUTIVE columns of one row, one 16-byte store per tile in the epilogue
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, %0" : "+a"(acc[a][b]) : "v"(afc), "v"(bf[kh][b]));
          // ---- the slot behind MFMA b of row R ----
          if (ABL & 1) {
          } else if (R < 15) {
            if (b == 0) afn = (a + 1 < WA) ? OA::frag(fa, (a + 1) * 16, kh) : OA::frag(fa, 0, 1);
            if (kh == 0 && b == 1) bf[1][a] = OB::frag(fb, a * 16, 1);
          } else {
            bf[0][b] = OB::frag(fbn, b * 16, 0);          // the next step's first fragments, from the other stage
            if (b == WB - 1) afn = OA::frag(fan, 0, 0);
          }
          if (kh == 0 && DR == 8) {                                  // DMA chunks a of both operands: 3 + 3 instructions in 5 slots
            if (b == 2) asm volatile("s_add_i32 m0, %0, %1" :: "s"(dstA), "n"(a * 4096) : "scc", "memory");
            if (b == 3) dma_load(AK ? la.off[a & 1] : la.off[0], dA, soffA);
            if (b == 4) { soffA += pitchA; asm volatile("s_add_i32 m0, %0, %1" :: "s"(dstB), "n"(a * 4096) : "scc", "memory"); }
            if (b == 5) dma_load(BK_ ? lb.off[a & 1] : lb.off[0], dB, soffB);
            if (b == 6) soffB += pitchB;
          } else if (kh == 0 && DR == 4) {                           // rows 0..3: chunks 2a and 2a+1 of both operands, one instruction per slot
            if (a < 4) {
              constexpr int c0 = (2 * a) & 7, c1 = (2 * a + 1) & 7;
              if (b == 0) asm volatile("s_add_i32 m0, %0, %1" :: "s"(dstA), "n"(c0 * 4096) : "scc", "memory");
              if (b == 1) { dma_load(AK ? la.off[c0 & 1] : la.off[0], dA, soffA); soffA += pitchA; }
              if (b == 2) asm volatile("s_add_i32 m0, %0, %1" :: "s"(dstB), "n"(c0 * 4096) : "scc", "memory");
              if (b == 3) { dma_load(BK_ ? lb.off[c0 & 1] : lb.off[0], dB, soffB); soffB += pitchB; }
              if (b == 4) asm volatile("s_add_i32 m0, %0, %1" :: "s"(dstA), "n"(c1 * 4096) : "scc", "memory");
              if (b == 5) { dma_load(AK ? la.off[c1 & 1] : la.off[0], dA, soffA); soffA += pitchA; }
              if (b == 6) asm volatile("s_add_i32 m0, %0, %1" :: "s"(dstB), "n"(c1 * 4096) : "scc", "memory");
              if (b == 7) { dma_load(BK_ ? lb.off[c1 & 1] : lb.off[0], dB, soffB); soffB += pitchB; }
            }
          } else {
            if (a == 0 && b == 2) adv(dA, stepA);
            if (a == 1 && b == 2) adv(dB, stepB);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        afc = afn;
      };
      row(std::integral_constant<int, 0>{});  row(std::integral_constant<int, 1>{});  row(std::integral_constant<int, 2>{});
      row(std::integral_constant<int, 3>{});  row(std::integral_constant<int, 4>{});  row(std::integral_constant<int, 5>{});
      row(std::integral_constant<int, 6>{});  row(std::integral_constant<int, 7>{});  row(std::integral_constant<int, 8>{});
      row(std::integral_constant<int, 9>{});  row(std::integral_constant<int, 10>{}); row(std::integral_constant<int, 11>{});
      row(std::integral_constant<int, 12>{}); row(std::integral_constant<int, 13>{}); row(std::integral_constant<int, 14>{});
      row(std::integral_constant<int, 15>{});
      soffA = soffB = 0;
      if (lv) pf_done();
      cstage ^= 1;
    }
    // the MFMAs above are opaque to the compiler's hazard recognizer: let the last one drain before its result is read
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // Epilogue.  Round 3's (one dword store per accumulator register, per-element bound checks, 64-bit address arithmetic per
    // store, the row-bias loads chained load -> add -> store) cost 40 us per tile -- a quarter of the xg contraction, more
    // with the row bias (tools/gemm_epilogue_probe.py: T(2K) - T(K) against 2 T(K) - T(2K)).  Now: one row pointer per
    // accumulator row block, eight 16-byte stores at immediate offsets, the bias as one 16-byte load per tile; tiles that
    // stick out of the matrix (and unaligned outputs) take the element-wise path.
    const int i = lane & 15, gq = lane >> 4;
    const bool whole = g.vec_ok && tm * TM + TM <= g.M && tn * TN + TN <= g.N;
    const int n0 = tn * TN + wn * 128 + 4 * gq;
    auto bias_ptr = [&](int a) -> const float* {
      int m = tm * TM + wm * 128 + a * 16 + i;
      m = m < g.M ? m : g.M - 1;
      return g.rowbias + (size_t)(m / g.group) * g.ldrb + n0;
    };
    // the row bias of block a + 1 is loaded while block a is stored: one exposed load latency per tile instead of eight
    f32x4 bias_nx[WB];
    if (whole && g.rowbias) {
      const float* rb0 = bias_ptr(0);
#pragma unroll
      for (int b = 0; b < WB; ++b) bias_nx[b] = *reinterpret_cast<const f32x4*>(rb0 + 16 * b);
    }
#pragma unroll
    for (int a = 0; a < WA; ++a) {
      const int m = tm * TM + wm * 128 + a * 16 + i;
      const int mc = m < g.M ? m : g.M - 1;                                    // (rows beyond the matrix: pointers stay valid, nothing is stored)
      float* crow = ((g.C2 && mc >= g.split_m) ? g.C2 + (size_t)(mc - g.split_m) * g.ldc : g.C + (size_t)mc * g.ldc) + n0;
      const float* rb = g.rowbias ? g.rowbias + (size_t)(mc / g.group) * g.ldrb + n0 : nullptr;
      if (whole) {
        f32x4 v[WB];
#pragma unroll
        for (int b = 0; b < WB; ++b) v[b] = acc_read(acc[a][b]);
        if (rb) {
#pragma unroll
          for (int b = 0; b < WB; ++b) v[b] += bias_nx[b];
          if (a + 1 < WA) {
            const float* rbn = bias_ptr(a + 1);
#pragma unroll
            for (int b = 0; b < WB; ++b) bias_nx[b] = *reinterpret_cast<const f32x4*>(rbn + 16 * b);
          }
        }
        if (g.accumulate > 0) {
#pragma unroll
          for (int b = 0; b < WB; ++b) v[b] += *reinterpret_cast<const f32x4*>(crow + 16 * b);
        }
        if (g.gate) {
          const float* gr = g.gate + (size_t)mc * g.ldg + n0;
#pragma unroll
          for (int b = 0; b < WB; ++b) {
            const f32x4 gt = *reinterpret_cast<const f32x4*>(gr + 16 * b);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[b][e] = gt[e] > 0.f ? v[b][e] : 0.f;
          }
        }
#pragma unroll
        for (int b = 0; b < WB; ++b) *reinterpret_cast<f32x4*>(crow + 16 * b) = v[b];
      } else if (m < g.M) {
        // a tile that sticks out of the matrix: the same 16-byte accesses under a per-lane predicate (N is a multiple of 4 and the
        // leading dimensions are 16-byte multiples here -- vec_ok; the launcher sends everything else to gemm_bf16_kernel, whose
        // epilogue works element by element: with that path unrolled in THIS kernel the allocator spilled 91 registers around it)
#pragma unroll
        for (int b = 0; b < WB; ++b) {
          if (n0 + 16 * b < g.N) {
            f32x4 v = acc_read(acc[a][b]);
            if (rb) v += *reinterpret_cast<const f32x4*>(rb + 16 * b);
            if (g.accumulate > 0) v += *reinterpret_cast<const f32x4*>(crow + 16 * b);
            if (g.gate) {
              const f32x4 gt = *reinterpret_cast<const f32x4*>(g.gate + (size_t)mc * g.ldg + n0 + 16 * b);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gt[e] > 0.f ? v[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(crow + 16 * b) = v;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // one row block at a time: hoisting the next blocks' loads across this point is what spilled
    }
  }
}

// fp32 [rows][ld] (K valid columns) -> bf16 [rows][Kp], zero padded
// VEC: K and ld multiples of 8 / 4 and a 16-byte aligned source -- two 16-byte loads per piece, two pieces in flight per thread [r6: the
// element-wise form below read a piece as eight 4-byte loads and ran the 370 MB gate-gradient conversion, on the backward's critical path, at
// 2.6 TB/s]
template <bool VEC>
__global__ __launch_bounds__(256)
void cvt_rows_bf16_kernel(const float* __restrict__ src, long long rows, int K, int ld, unsigned short* __restrict__ dst, int Kp) {
  const int groups = Kp >> 3;
  const long long total = rows * groups;
  if constexpr (VEC) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const long long stride = (long long)gridDim.x * 256;
    auto conv = [&](long long i, f4 a, f4 b) {
      const long long r = i / groups;
      const int k0 = (int)(i - r * groups) * 8;
      if (k0 >= K) { a = f4{0.f, 0.f, 0.f, 0.f}; b = a; }
      const u4v o = {vs_pack_bf16(a[0], a[1]), vs_pack_bf16(a[2], a[3]), vs_pack_bf16(b[0], b[1]), vs_pack_bf16(b[2], b[3])};
      __builtin_nontemporal_store(o, reinterpret_cast<u4v*>(dst + r * Kp + k0));
    };
    auto addr = [&](long long i) {
      const long long r = i / groups;
      const int k0 = (int)(i - r * groups) * 8;
      return reinterpret_cast<const f4*>(src + r * ld + (k0 < K ? k0 : 0));
    };
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < total; i += 2 * stride) {
      const f4* p0 = addr(i);
      const f4* p1 = addr(i + stride);
      const f4 a0 = __builtin_nontemporal_load(p0), b0 = __builtin_nontemporal_load(p0 + 1);
      const f4 a1 = __builtin_nontemporal_load(p1), b1 = __builtin_nontemporal_load(p1 + 1);
      conv(i, a0, b0);
      conv(i + stride, a1, b1);
    }
    if (i < total) {
      const f4* p0 = addr(i);
      conv(i, __builtin_nontemporal_load(p0), __builtin_nontemporal_load(p0 + 1));
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / groups;
    const int k0 = (int)(i - r * groups) * 8;
    const float* p = src + r * ld + k0;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (k0 + e < K) ? p[e] : 0.f;
    const u4v o = {vs_pack_bf16(x[0], x[1]), vs_pack_bf16(x[2], x[3]), vs_pack_bf16(x[4], x[5]), vs_pack_bf16(x[6], x[7])};
    *reinterpret_cast<u4v*>(dst + r * Kp + k0) = o;
  }
}

int gemm_cus() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  return v;
}

}  // namespace

int vs_cvt_rows_bf16_impl(const float* src, long long rows, int K, int ld, void* dst, int Kp, hipStream_t stream) {
  VS_REQUIRE(src && dst && rows > 0 && K > 0 && ld >= K && Kp >= K && Kp % 8 == 0, "cvt_rows_bf16: bad argument");
  const long long total = rows * (Kp >> 3);
  const long long nb = (total + 255) / 256;
  const long long ccap = 1LL << 24;      // [r6, call 31] one sweep instead of a 4096-block grid-stride loop: 32.8 -> 21.7 us mean over the step's six conversions
  const bool vec = K % 8 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
  if (vec) hipLaunchKernelGGL(cvt_rows_bf16_kernel<true>, dim3((unsigned)(nb < ccap ? nb : ccap)), dim3(256), 0, stream, src, rows, K, ld,
                              reinterpret_cast<unsigned short*>(dst), Kp);
  else
  hipLaunchKernelGGL(cvt_rows_bf16_kernel<false>, dim3((unsigned)(nb < ccap ? nb : ccap)), dim3(256), 0, stream, src, rows, K, ld,
                     reinterpret_cast<unsigned short*>(dst), Kp);
  VS_LAUNCH_CHECK();
  return 0;
}

// C[M][N] (+)= opA(A) opB(B) + rowbias[m / group][n]; a_kmajor / b_kmajor: 0 = element (i, k) at i*ld + k, 1 = at k*ld + i.
// Operands bf16, 16-byte aligned, ld a multiple of 8; row-form operands must be readable (zero padded) up to the next
// multiple of 64 in k.  C2 / split_m: rows >= split_m are written to C2 (two output matrices stacked along M).
// gate [M][ldg] (or NULL): elements where gate <= 0 are written as 0 (the relu mask of the head's backward contractions).
int vs_gemm_bf16_impl(int a_kmajor, int b_kmajor, const void* A, int lda, const void* B, int ldb, float* C, int ldc, float* C2, int split_m,
                      int M, int N, int K, const float* rowbias, int ldrb, int group, int accumulate, hipStream_t stream,
                      const float* gate, int ldg) {
  VS_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "gemm_bf16: bad argument");
  VS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0,
             "gemm_bf16: operands must be 16-byte aligned with ld a multiple of 8");
  VS_REQUIRE(!rowbias || (group > 0 && ldrb >= N), "gemm_bf16: rowbias needs group > 0 and ldrb >= N");
  VS_REQUIRE(!gate || (ldg >= N && !C2), "gemm_bf16: gate needs ldg >= N and a single output");
  VS_REQUIRE(a_kmajor || lda >= (K + BK - 1) / BK * BK, "gemm_bf16: row-form A must be padded to a multiple of %d in k", BK);
  VS_REQUIRE(b_kmajor || ldb >= (K + BK - 1) / BK * BK, "gemm_bf16: row-form B must be padded to a multiple of %d in k", BK);
  GemmBf16Args g{reinterpret_cast<const unsigned short*>(A), lda, reinterpret_cast<const unsigned short*>(B), ldb, C, ldc, C2, split_m,
                 M, N, K, rowbias, ldrb, group > 0 ? group : 1, accumulate, (M + TM - 1) / TM, (N + TN - 1) / TN, 8, gate, ldg, 0};
  // N % 4: the partial-tile epilogue of the interleaved kernel checks a 4-column group by its first column only (ADVICE round 5:
  // with N % 4 != 0 it wrote up to 3 columns past N and read rowbias / gate there)
  g.vec_ok = N % 4 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && (!C2 || (reinterpret_cast<uintptr_t>(C2) & 15) == 0) &&
             (!rowbias || (ldrb % 4 == 0 && (reinterpret_cast<uintptr_t>(rowbias) & 15) == 0)) &&
             (!gate || (ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(gate) & 15) == 0));
  static int cus = gemm_cus();
  const long long ntiles = (long long)g.tiles_m * g.tiles_n;
  long long nwg = cus / 8 * 8;
  if (nwg > (ntiles + 7) / 8 * 8) nwg = (ntiles + 7) / 8 * 8;
  if (nwg < 8) nwg = 8;
  const dim3 grid((unsigned)nwg), block(256);
  const int form = (!a_kmajor && !b_kmajor) ? 0 : (!a_kmajor && b_kmajor) ? 1 : (a_kmajor && b_kmajor) ? 2 : 3;
  VS_REQUIRE(form != 3, "gemm_bf16: the col x row form is not used by the path");
  // round 4: the interleaved kernel (side work in the slots behind the MFMAs, DMA chunks in 8 rows of the step: 4 rows measured
  // slower, profiles/r04_gemm_interleave.md); shapes its 16-byte epilogue cannot serve (vec_ok) take the round-3 kernel below
  if (g.vec_ok) {
#ifdef VS_ABLATION
    const int abl = vs_opt(VS_OPT_ABLATION);
    if (abl == 9) g.accumulate = -1;                       // no DMA: timing only, results are meaningless
    if (abl >= 1 && abl <= 3 && form == 0) {
      if (abl == 1) hipLaunchKernelGGL((gemm_bf16_il_kernel<false, false, 8, 1>), grid, block, 0, stream, g);
      else if (abl == 2) hipLaunchKernelGGL((gemm_bf16_il_kernel<false, false, 8, 2>), grid, block, 0, stream, g);
      else hipLaunchKernelGGL((gemm_bf16_il_kernel<false, false, 8, 3>), grid, block, 0, stream, g);
      VS_LAUNCH_CHECK();
      return 0;
    }
#endif
    if (form == 0) hipLaunchKernelGGL((gemm_bf16_il_kernel<false, false, 8>), grid, block, 0, stream, g);
    else if (form == 1) hipLaunchKernelGGL((gemm_bf16_il_kernel<false, true, 8>), grid, block, 0, stream, g);
    else hipLaunchKernelGGL((gemm_bf16_il_kernel<true, true, 8>), grid, block, 0, stream, g);
    VS_LAUNCH_CHECK();
    return 0;
  }
  // the round-3 kernel with the DMA placement that was measured fastest per operand-form pair -- row x row: two chunks in front
  // of each of the first 8 MFMA rows, the two col forms: one in front of each of the 16 rows (profiles/r03_gemm_l2_prefetch.md)
  if (form == 0) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, 1>), grid, block, 0, stream, g);
  else if (form == 1) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, 2>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((gemm_bf16_kernel<true, true, 2>), grid, block, 0, stream, g);
  VS_LAUNCH_CHECK();
  return 0;
}
