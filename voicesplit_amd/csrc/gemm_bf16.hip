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
// Structure: persistent workgroups (one per CU, 4 waves = 2 x 2, wave tile 64 x 128, workgroup tile 128 x 256, K step
// 64), tiles walked in bands of 8 tile rows per XCD (gemm_f16x3.hip's map); both operand tiles of a K step go
// HBM/L2 -> LDS by LDS-DMA into a ring of three stages, two steps ahead of the step that reads them (counted vmcnt,
// raw s_barrier: nothing drains early); XOR swizzles on the source address make the fragment reads conflict-free;
// per K step a wave issues 24 fragment reads and 64 MFMAs in a pinned order (reads one k-half ahead).  fp32
// accumulate, fp32 output (+ optional per-row-group bias, or accumulate into C).
#include "vs_internal.h"

namespace {

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) const u4v lds_u4v;

__device__ u4v g_gemm_zero_page[4];

constexpr int TM = 128, TN = 256, BK = 64;      // workgroup tile, K step
constexpr int A_BYTES = TM * BK * 2;            // 16 KiB
constexpr int B_BYTES = TN * BK * 2;            // 32 KiB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // 48 KiB
constexpr int NSTAGE = 3;

struct GemmBf16Args {
  const unsigned short* A; int lda;   // row form: [M][lda], col form: [K][lda]
  const unsigned short* B; int ldb;   // row form: [N][ldb], col form: [K][ldb]
  float* C; int ldc;
  float* C2; int split_m;             // rows >= split_m go to C2 + (m - split_m) * ldc (the two directions of dW_ih); else NULL
  int M, N, K;                        // K = number of valid k (zero-padded operands may be read beyond it up to the next BK)
  const float* rowbias; int ldrb, group;
  int accumulate;
  int tiles_m, tiles_n, band;
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
  // DMA of one K step: R*64 halves = R*8 pieces; piece e of the tile -> LDS byte 16 e (lane-linear), source swizzled
  static constexpr int PIECES = R * 8;
  __device__ static __forceinline__ void issue(const unsigned short* base, int ld, int rows /* matrix extent along R */, int kext,
                                               int r0, int k0, unsigned lds_dst, int wave, int lane) {
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_gemm_zero_page);
    const long long rel0 = reinterpret_cast<const unsigned char*>(base) - zp;
#pragma unroll
    for (int c = 0; c < PIECES / 256; ++c) {
      const int e = (c * 4 + wave) * 64 + lane;            // chunk (c, wave) = 64 consecutive pieces = 1 KiB
      long long off;
      bool ok;
      if (!KMAJOR) {
        const int r = e >> 3, q = e & 7;
        const int kp = q ^ ((r >> 1) & 7);
        ok = (r0 + r < rows) & (k0 + kp * 8 < kext);
        off = ((long long)(r0 + r) * ld + k0 + kp * 8) * 2;
      } else {
        constexpr int PPL = R / 8;                         // pieces per k line
        const int k = e / PPL, q = e - k * PPL;
        const int u = (k & 3) | (((k >> 3) & 1) << 2);
        const int src = q ^ (u << 1);                      // piece index inside the line (pairs of pieces = 32-byte units)
        ok = (k0 + k < kext) & (r0 + src * 8 < rows);
        off = ((long long)(k0 + k) * ld + r0 + src * 8) * 2;
      }
      const unsigned char* src_p = zp + ((rel0 + off) & -(long long)ok);
      glds16(src_p, (unsigned)__builtin_amdgcn_readfirstlane(lds_dst + (unsigned)((c * 4 + wave) * 1024)));
    }
  }
  // fragment of the 16-row block starting at row rb (of the tile), k-half kh (32 k): lane (i = lane & 15, g = lane >> 4)
  __device__ static __forceinline__ vs_bf16x8 frag(unsigned img, int rb, int kh, int lane) {
    const int i = lane & 15, g = lane >> 4;
    if (!KMAJOR) {
      const int r = rb + i;
      const unsigned a = img + (unsigned)(r * 128 + (((kh * 4 + g) ^ ((r >> 1) & 7)) << 4));
      return __builtin_bit_cast(vs_bf16x8, *(lds_u4v*)(uintptr_t)a);
    } else {
      s4v h[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int k = kh * 32 + 8 * g + 4 * hf + (i >> 2);
        const int u = (k & 3) | (((k >> 3) & 1) << 2);
        const int unit = (rb >> 4) ^ u;                     // 32-byte unit = 16 rows of the matrix
        h[hf] = ds_read_tr16(img + (unsigned)(k * (R * 2) + (unit << 5) + (i & 3) * 8));
      }
      return frag_of(h[0], h[1]);
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

template <bool AK, bool BK_>
__global__ __launch_bounds__(256, 1)
void gemm_bf16_kernel(GemmBf16Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE_BYTES];
  using OA = Operand<AK, TM>;
  using OB = Operand<BK_, TN>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;                 // wave tile: rows [64 wm, +64), columns [128 wn, +128)
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
  auto pf_issue = [&]() {
    const unsigned dst = lds0 + (unsigned)(pstage * STAGE_BYTES);
    OA::issue(g.A, g.lda, g.M, g.K, ptm * TM, pk * BK, dst, wave, lane);
    OB::issue(g.B, g.ldb, g.N, g.K, ptn * TN, pk * BK, dst + A_BYTES, wave, lane);
    pstage = pstage + 1 == NSTAGE ? 0 : pstage + 1;
    if (++pk == nk) {
      pk = 0;
      ++pj;
      plive = tile_ok(pj) && tile_of(g, tile_id(pj), ptm, ptn);
    }
  };
  constexpr int DMA_PER_STAGE = (OA::PIECES + OB::PIECES) / 256;      // per wave: 4 + 8 = 12
  if (plive) pf_issue();
  if (plive) pf_issue();

  int cstage = 0;
  for (int j = 0; tile_ok(j); ++j) {
    int tm, tn;
    if (!tile_of(g, tile_id(j), tm, tn)) break;
    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < nk; ++ks) {
      // the stage of this step was issued two issues ago: at most the newest stage's DMAs may still be in flight.  (The
      // epilogue's stores of the previous tile are older than both and complete with the same wait.)
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_STAGE) : "memory");
      __builtin_amdgcn_s_barrier();
      // every wave is past the previous step: its stage can be refilled (it is the stage three issues back)
      if (plive) pf_issue();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing issued now: the NEXT step's stage is then the newest one
      const unsigned imgA = lds0 + (unsigned)(cstage * STAGE_BYTES), imgB = imgA + A_BYTES;
      vs_bf16x8 af[2][4], bf[2][8];
#pragma unroll
      for (int a = 0; a < 4; ++a) af[0][a] = OA::frag(imgA, wm * 64 + a * 16, 0, lane);
#pragma unroll
      for (int b = 0; b < 8; ++b) bf[0][b] = OB::frag(imgB, wn * 128 + b * 16, 0, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        if (kh == 0) {
#pragma unroll
          for (int a = 0; a < 4; ++a) af[1][a] = OA::frag(imgA, wm * 64 + a * 16, 1, lane);
#pragma unroll
          for (int b = 0; b < 8; ++b) bf[1][b] = OB::frag(imgB, wn * 128 + b * 16, 1, lane);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kh][a], bf[kh][b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      cstage = cstage + 1 == NSTAGE ? 0 : cstage + 1;
    }
    // epilogue: lane holds C[m = 16a + 4 (lane>>4) + r][n = 16b + (lane&15)]
    const int i = lane & 15, gq = lane >> 4;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = tm * TM + wm * 64 + a * 16 + 4 * gq + r;
        if (m >= g.M) continue;
        float* crow = (g.C2 && m >= g.split_m) ? g.C2 + (size_t)(m - g.split_m) * g.ldc : g.C + (size_t)m * g.ldc;
        const float* rb = g.rowbias ? g.rowbias + (size_t)(m / g.group) * g.ldrb : nullptr;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int n = tn * TN + wn * 128 + b * 16 + i;
          if (n < g.N) {
            float v = acc[a][b][r];
            if (rb) v += rb[n];
            if (g.accumulate) v += crow[n];
            crow[n] = v;
          }
        }
      }
  }
}

// fp32 [rows][ld] (K valid columns) -> bf16 [rows][Kp], zero padded
__global__ __launch_bounds__(256)
void cvt_rows_bf16_kernel(const float* __restrict__ src, long long rows, int K, int ld, unsigned short* __restrict__ dst, int Kp) {
  const int groups = Kp >> 3;
  const long long total = rows * groups;
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
  hipLaunchKernelGGL(cvt_rows_bf16_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, stream, src, rows, K, ld,
                     reinterpret_cast<unsigned short*>(dst), Kp);
  VS_LAUNCH_CHECK();
  return 0;
}

// C[M][N] (+)= opA(A) opB(B) + rowbias[m / group][n]; a_kmajor / b_kmajor: 0 = element (i, k) at i*ld + k, 1 = at k*ld + i.
// Operands bf16, 16-byte aligned, ld a multiple of 8; row-form operands must be readable (zero padded) up to the next
// multiple of 64 in k.  C2 / split_m: rows >= split_m are written to C2 (two output matrices stacked along M).
int vs_gemm_bf16_impl(int a_kmajor, int b_kmajor, const void* A, int lda, const void* B, int ldb, float* C, int ldc, float* C2, int split_m,
                      int M, int N, int K, const float* rowbias, int ldrb, int group, int accumulate, hipStream_t stream) {
  VS_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "gemm_bf16: bad argument");
  VS_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0,
             "gemm_bf16: operands must be 16-byte aligned with ld a multiple of 8");
  VS_REQUIRE(!rowbias || (group > 0 && ldrb >= N), "gemm_bf16: rowbias needs group > 0 and ldrb >= N");
  VS_REQUIRE(a_kmajor || lda >= (K + BK - 1) / BK * BK, "gemm_bf16: row-form A must be padded to a multiple of %d in k", BK);
  VS_REQUIRE(b_kmajor || ldb >= (K + BK - 1) / BK * BK, "gemm_bf16: row-form B must be padded to a multiple of %d in k", BK);
  GemmBf16Args g{reinterpret_cast<const unsigned short*>(A), lda, reinterpret_cast<const unsigned short*>(B), ldb, C, ldc, C2, split_m,
                 M, N, K, rowbias, ldrb, group > 0 ? group : 1, accumulate, (M + TM - 1) / TM, (N + TN - 1) / TN, 8};
  static int cus = gemm_cus();
  const long long ntiles = (long long)g.tiles_m * g.tiles_n;
  long long nwg = cus / 8 * 8;
  if (nwg > (ntiles + 7) / 8 * 8) nwg = (ntiles + 7) / 8 * 8;
  if (nwg < 8) nwg = 8;
  const dim3 grid((unsigned)nwg), block(256);
  if (!a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, stream, g);
  else if (!a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, stream, g);
  else if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, stream, g);
  else VS_REQUIRE(false, "gemm_bf16: the col x row form is not used by the path");
  VS_LAUNCH_CHECK();
  return 0;
}
