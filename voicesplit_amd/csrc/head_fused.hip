// The mask head of models/voicesplit/model.py:83-87 in ONE kernel (VS_MATH_BF16):
//     mask = sigmoid(fc2(relu(fc1(relu(lstm_out)))))
// Round 4.  The two contractions are chained THROUGH REGISTERS: a wave owns 16 rows of the batch; with the operand roles of
// v_mfma_f32_16x16x32_bf16 swapped (weights in the A slot, activations in the B slot) lane (row i, g) ends GEMM 1 holding
// h1[row][16 t + 4 g + 0..3] for every 16-column tile t -- and the B operand of GEMM 2 wants, per lane (row i, g), eight
// k values of a 32-wide chunk.  Two neighbouring tiles give the lane exactly eight: k in {32 c + 4 g + j} U {32 c + 16 + 4 g + j}.
// That is a permutation of the chunk's k, and a contraction does not care in which order k is walked as long as both operands
// agree -- so fc2's weights are PACKED in that order and bias + relu + bf16 rounding turn GEMM 1's accumulators into GEMM 2's
// operand fragments in place: h1 never touches LDS or memory (train mode stores it once for the backward pass, straight from
// the registers, 16 bytes per lane).  Weights reach the waves of a workgroup through LDS: packed in fragment order
// ([chunk][tile][lane][8 x bf16]: a tile image is 1 KiB = one LDS-DMA instruction of one wave), double-buffered, one barrier per
// 32-wide chunk; a workgroup is 5 waves = 80 rows (241 workgroups at B = 64: one round on 256 CUs).  Tile counts are template
// parameters (three instances); a shape runs on the smallest instance that holds it, zero-padded in the packed images.
// Arithmetic: bf16-rounded operands (lstm_out, W1, h1, W2), fp32 accumulation, fp32 bias / sigmoid -- the same roundings as the
// two-launch form it replaces (gemm_mfma.hip's bf16 instances); summation order differs within fp32 rounding.
#include "vs_internal.h"

namespace {

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(3))) const u4v lds_u4v;

constexpr int kWaves = 5;                 // 80 rows per workgroup
constexpr int kDepth = 6;                 // weight fragments in flight from LDS ahead of the MFMA that consumes them

struct HeadArgs {
  const float* x;                         // lstm_out [M][K1]
  const unsigned short* w1p;              // [nk1][NT1][64][8]
  const unsigned short* w2p;              // [2][NC2][NT2 / 2][64][8], k in the register order of h1 (see above)
  const float *b1p, *b2p;                  // biases padded with zeros to 16 NT1 / 16 NT2 floats (part of the packed image)
  float *h1, *logits, *mask;              // h1 [M][FC1] (train) / logits / mask [M][FC2]; each may be NULL
  int M, K1, FC1, FC2, nk1;
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// NC2 32-wide chunks of h1 (NT1 = 2 NC2 tiles of fc1's output), NT2 tiles of fc2's output: compile-time, so that the tile loops are
// straight-line code the scheduler can pipeline (LDS read of tile t + 1.. under the MFMA of tile t).  A shape runs on the smallest
// instance that holds it; the packed images are zero beyond FC1 / FC2.
template <int NC2, int NT2>
__global__ __launch_bounds__(kWaves * 64, 1)
void head_fused_bf16_kernel(HeadArgs g) {
  constexpr int NT1 = 2 * NC2, NH = NT2 / 2;
  static_assert(NT2 % 2 == 0 && NH <= NT1, "fc2's tiles are walked in two halves, each no larger than a stage");
  constexpr int kStageBytes = NT1 * 1024;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kStageBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, gq = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(const lds_byte*)smem;
  const int row = blockIdx.x * (kWaves * 16) + wave * 16 + i;
  const bool row_ok = row < g.M;
  const float* xr = g.x + (size_t)(row_ok ? row : g.M - 1) * g.K1;

  // weights of step s (GEMM 1 chunks, then GEMM 2's [half][chunk]) into stage s & 1: wave w moves tile images w, w + 5, ...
  auto dma = [&](int s) {
    const unsigned dst = lds0 + (unsigned)((s & 1) * kStageBytes);
    if (s < g.nk1) {
      const unsigned short* src = g.w1p + (size_t)s * (NT1 * 512);
#pragma unroll
      for (int t = 0; t < (NT1 + kWaves - 1) / kWaves; ++t)
        if (t * kWaves + wave < NT1)
          glds16(src + ((size_t)(t * kWaves + wave) * 64 + lane) * 8,
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)((t * kWaves + wave) * 1024))));
    } else {
      const unsigned short* src = g.w2p + (size_t)(s - g.nk1) * (NH * 512);
#pragma unroll
      for (int t = 0; t < (NH + kWaves - 1) / kWaves; ++t)
        if (t * kWaves + wave < NH)
          glds16(src + ((size_t)(t * kWaves + wave) * 64 + lane) * 8,
                 (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)((t * kWaves + wave) * 1024))));
    }
  };
  // activations of GEMM 1 chunk kc: relu + bf16 of x[row][32 kc + 8 gq .. + 7]
  auto load_a = [&](int kc, float4& lo, float4& hi) {
    const int k0 = 32 * kc + 8 * gq;
    if (row_ok && k0 < g.K1) {                                // K1 is a multiple of 8: a group of 8 is inside or outside
      lo = *reinterpret_cast<const float4*>(xr + k0);
      hi = *reinterpret_cast<const float4*>(xr + k0 + 4);
    } else {
      lo = hi = float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto pack_a = [&](const float4& lo, const float4& hi) {
    const u4v v = {vs_pack_bf16(fmaxf(lo.x, 0.f), fmaxf(lo.y, 0.f)), vs_pack_bf16(fmaxf(lo.z, 0.f), fmaxf(lo.w, 0.f)),
                   vs_pack_bf16(fmaxf(hi.x, 0.f), fmaxf(hi.y, 0.f)), vs_pack_bf16(fmaxf(hi.z, 0.f), fmaxf(hi.w, 0.f))};
    return __builtin_bit_cast(vs_bf16x8, v);
  };

  f32x4 acc[NT1];
#pragma unroll
  for (int t = 0; t < NT1; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  dma(0);
  float4 a_lo, a_hi;
  load_a(0, a_lo, a_hi);
  // ---- GEMM 1: h1 = relu(x) @ W1^T ----------------------------------------------------------------------------------
  for (int s = 0; s < g.nk1; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's tile images of step s (and its activations) have landed
    __builtin_amdgcn_s_barrier();                             // ... every wave's; every wave is done with the other stage
    const vs_bf16x8 af = pack_a(a_lo, a_hi);
    dma(s + 1);                                               // s + 1 < steps: GEMM 2 has at least one chunk
    if (s + 1 < g.nk1) load_a(s + 1, a_lo, a_hi);
    const unsigned base = lds0 + (unsigned)((s & 1) * kStageBytes) + (unsigned)(lane * 16);
    u4v w[kDepth + 1];
#pragma unroll
    for (int t = 0; t < kDepth && t < NT1; ++t) w[t] = *(lds_u4v*)(uintptr_t)(base + (unsigned)(t * 1024));
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      if (t + kDepth < NT1) w[(t + kDepth) % (kDepth + 1)] = *(lds_u4v*)(uintptr_t)(base + (unsigned)((t + kDepth) * 1024));
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vs_bf16x8, w[t % (kDepth + 1)]), af, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                       // registers: no hoisting of the tile reads beyond kDepth
    }
  }
  // ---- h1 = relu(acc + b1): stored (train), rounded to bf16, and already GEMM 2's activation fragments ------------------
  vs_bf16x8 hf[NC2];
  const bool h1_vec = (g.FC1 & 3) == 0;
#pragma unroll
  for (int c = 0; c < NC2; ++c) {
    u4v pk;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = 2 * c + h;
      const int n = 16 * t + 4 * gq;                          // this lane's four columns of tile t
      const float4 b = *reinterpret_cast<const float4*>(g.b1p + n);
      // columns >= FC1: zero weights and zero bias give relu(0) = 0, and fc2's packed image is zero there as well
      const float v0 = fmaxf(acc[t][0] + b.x, 0.f), v1 = fmaxf(acc[t][1] + b.y, 0.f), v2 = fmaxf(acc[t][2] + b.z, 0.f), v3 = fmaxf(acc[t][3] + b.w, 0.f);
      if (g.h1 != nullptr && 16 * t < g.FC1) {                // wave-uniform
        float* dst = g.h1 + (size_t)row * g.FC1 + n;
        if (h1_vec) {                                          // FC1 % 4 == 0: a group of four is inside or outside
          if (row_ok && n < g.FC1) *reinterpret_cast<float4*>(dst) = float4{v0, v1, v2, v3};
        } else if (row_ok) {
          if (n < g.FC1) dst[0] = v0;
          if (n + 1 < g.FC1) dst[1] = v1;
          if (n + 2 < g.FC1) dst[2] = v2;
          if (n + 3 < g.FC1) dst[3] = v3;
        }
      }
      pk[2 * h] = vs_pack_bf16(v0, v1);
      pk[2 * h + 1] = vs_pack_bf16(v2, v3);
    }
    hf[c] = __builtin_bit_cast(vs_bf16x8, pk);
    __builtin_amdgcn_sched_barrier(0);                         // registers: one chunk's bias loads at a time
  }
  // ---- GEMM 2: logits = h1 @ W2^T, one half of fc2's columns at a time (registers: hf + half the accumulators) -----------
  for (int half = 0; half < 2; ++half) {                      // a real loop: the body is 19 x 19 MFMAs of straight-line code
    f32x4 acc2[NH];
#pragma unroll
    for (int t = 0; t < NH; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC2; ++c) {
      const int s = g.nk1 + half * NC2 + c;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (half * NC2 + c + 1 < 2 * NC2) dma(s + 1);
      const unsigned base = lds0 + (unsigned)((s & 1) * kStageBytes) + (unsigned)(lane * 16);
      u4v w[kDepth + 1];
#pragma unroll
      for (int t = 0; t < kDepth && t < NH; ++t) w[t] = *(lds_u4v*)(uintptr_t)(base + (unsigned)(t * 1024));
#pragma unroll
      for (int t = 0; t < NH; ++t) {
        if (t + kDepth < NH) w[(t + kDepth) % (kDepth + 1)] = *(lds_u4v*)(uintptr_t)(base + (unsigned)((t + kDepth) * 1024));
        acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vs_bf16x8, w[t % (kDepth + 1)]), hf[c], acc2[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue of this half: + b2, sigmoid (the stores drain under the other half's contraction) --------------------
#pragma unroll
    for (int tl = 0; tl < NH; ++tl) {
      const int t = half * NH + tl;
      if (16 * t < g.FC2) {                                   // wave-uniform
        const int n = 16 * t + 4 * gq;
        const float4 b = *reinterpret_cast<const float4*>(g.b2p + n);
        const float z[4] = {acc2[tl][0] + b.x, acc2[tl][1] + b.y, acc2[tl][2] + b.z, acc2[tl][3] + b.w};
        const size_t o = (size_t)row * g.FC2 + n;
        if (16 * t + 16 <= g.FC2) {                           // a whole tile of columns (wave-uniform); rows are 4-byte aligned only (FC2 = 601)
          if (row_ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (g.logits) g.logits[o + r] = z[r];
              if (g.mask) g.mask[o + r] = vs_act<VS_ACT_SIGMOID>(z[r]);
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (row_ok && n + r < g.FC2) {
              if (g.logits) g.logits[o + r] = z[r];
              if (g.mask) g.mask[o + r] = vs_act<VS_ACT_SIGMOID>(z[r]);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// w1p[((kc * nt1 + t) * 64 + lane) * 8 + j] = bf16(W1[16 t + (lane & 15)][32 kc + 8 (lane >> 4) + j]);
// w2p[(((half * nc2 + c) * nh + tl) * 64 + lane) * 8 + j] = bf16(W2[16 (half nh + tl) + (lane & 15)][32 c + (j < 4 ? 4 gq + j : 16 + 4 gq + j - 4)]),
// gq = lane >> 4, nh = nt2 / 2
__global__ void head_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                 const float* __restrict__ b2, unsigned short* __restrict__ w1p, unsigned short* __restrict__ w2p,
                                 float* __restrict__ bp, int K1, int FC1, int FC2, int nk1, int nt1, int nc2, int nt2) {
  const long long n1 = (long long)nk1 * nt1 * 512, n2 = (long long)nc2 * nt2 * 512;
  const int nb = 16 * (nt1 + nt2);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n1 + n2 + nb; e += (long long)gridDim.x * blockDim.x) {
    if (e >= n1 + n2) {                                       // [b1 | 0 ..][b2 | 0 ..]
      const int q = (int)(e - n1 - n2);
      bp[q] = q < 16 * nt1 ? (q < FC1 ? b1[q] : 0.f) : (q - 16 * nt1 < FC2 ? b2[q - 16 * nt1] : 0.f);
      continue;
    }
    const bool second = e >= n1;
    const long long q = second ? e - n1 : e;
    const int j = (int)(q & 7), lane = (int)((q >> 3) & 63);
    const long long ct = q >> 9;
    int chunk, t;
    if (!second) {
      chunk = (int)(ct / nt1), t = (int)(ct - (long long)chunk * nt1);
    } else {                                                  // [half][chunk][tile of the half]
      const int nh = nt2 / 2, s2 = (int)(ct / nh), half = s2 / nc2;
      chunk = s2 - half * nc2, t = half * nh + (int)(ct - (long long)s2 * nh);
    }
    const int n = 16 * t + (lane & 15), gq = lane >> 4;
    float v = 0.f;
    if (!second) {
      const int k = 32 * chunk + 8 * gq + j;
      if (n < FC1 && k < K1) v = w1[(size_t)n * K1 + k];
    } else {
      const int k = 32 * chunk + (j < 4 ? 4 * gq + j : 16 + 4 * gq + (j - 4));
      if (n < FC2 && k < FC1) v = w2[(size_t)n * FC1 + k];
    }
    (second ? w2p : w1p)[q] = (unsigned short)(vs_pack_bf16(v, 0.f) & 0xffffu);
  }
}

struct HeadShape { int nk1, nt1, nc2, nt2; size_t w1_halves, w2_halves; };
// the compiled instances (NC2, NT2): config.json's fc1_dim 600 / fc2_dim 601 exactly, a small one, and one up to 640 x 640
constexpr int kInst[3][2] = {{4, 8}, {19, 38}, {20, 40}};
inline bool head_shape(int K1, int FC1, int FC2, HeadShape* s) {
  if (K1 <= 0 || FC1 <= 0 || FC2 <= 0 || K1 % 8) return false;
  for (int q = 0; q < 3; ++q) {
    if ((FC1 + 31) / 32 > kInst[q][0] || (FC2 + 15) / 16 > kInst[q][1]) continue;
    s->nk1 = (K1 + 31) / 32;
    s->nc2 = kInst[q][0];
    s->nt1 = 2 * s->nc2;                    // GEMM 1's tiles come in pairs: a pair is one k chunk of GEMM 2
    s->nt2 = kInst[q][1];
    s->w1_halves = (size_t)s->nk1 * s->nt1 * 512;
    s->w2_halves = (size_t)s->nc2 * s->nt2 * 512;
    return true;
  }
  return false;
}

}  // namespace

bool vs_head_fused_supported(int K1, int FC1, int FC2) {
  HeadShape s;
  return head_shape(K1, FC1, FC2, &s);
}

size_t vs_head_fused_packed_bytes(int K1, int FC1, int FC2) {
  HeadShape s;
  if (!head_shape(K1, FC1, FC2, &s)) return 0;
  return ((s.w1_halves + s.w2_halves) * 2 + (size_t)16 * (s.nt1 + s.nt2) * 4 + 255) & ~size_t(255);
}

// fc1.weight [FC1][K1], fc2.weight [FC2][FC1], the two biases (fp32) -> the kernel's image: fragment-ordered bf16 weights, zero-padded biases
int vs_head_fused_pack_impl(const float* w1, const float* b1, const float* w2, const float* b2, int K1, int FC1, int FC2, void* packed,
                            hipStream_t stream) {
  VS_REQUIRE(w1 && b1 && w2 && b2 && packed && vs_head_fused_supported(K1, FC1, FC2), "head_fused_pack: bad argument");
  VS_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 15) == 0, "head_fused_pack: the packed buffer must be 16-byte aligned");
  HeadShape s;
  head_shape(K1, FC1, FC2, &s);
  unsigned short* p = reinterpret_cast<unsigned short*>(packed);
  const long long n = (long long)(s.w1_halves + s.w2_halves) + 16 * (s.nt1 + s.nt2);
  hipLaunchKernelGGL(head_pack_kernel, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, stream, w1, b1, w2, b2, p,
                     p + s.w1_halves, reinterpret_cast<float*>(p + s.w1_halves + s.w2_halves), K1, FC1, FC2, s.nk1, s.nt1, s.nc2, s.nt2);
  VS_LAUNCH_CHECK();
  return 0;
}

// relu(lstm_out) -> fc1 -> relu -> fc2 -> sigmoid in one launch; h1_out [M][FC1] (the backward pass's copy), logits, mask: each may be NULL
int vs_head_fused_impl(const float* lstm_out, const void* packed, float* h1_out, float* logits, float* mask,
                       int M, int K1, int FC1, int FC2, hipStream_t stream) {
  VS_REQUIRE(lstm_out && packed && (h1_out || logits || mask) && M > 0, "head_fused: bad argument");
  VS_REQUIRE(vs_head_fused_supported(K1, FC1, FC2), "head_fused: unsupported shape K1=%d FC1=%d FC2=%d", K1, FC1, FC2);
  VS_REQUIRE((reinterpret_cast<uintptr_t>(lstm_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(packed) & 15) == 0, "head_fused: 16-byte alignment");
  HeadShape s;
  head_shape(K1, FC1, FC2, &s);
  const unsigned short* p = reinterpret_cast<const unsigned short*>(packed);
  const float* bp = reinterpret_cast<const float*>(p + s.w1_halves + s.w2_halves);
  HeadArgs g{lstm_out, p, p + s.w1_halves, bp, bp + 16 * s.nt1, h1_out, logits, mask, M, K1, FC1, FC2, s.nk1};
  const dim3 grid((unsigned)((M + kWaves * 16 - 1) / (kWaves * 16))), block(kWaves * 64);
  if (s.nc2 == 4) hipLaunchKernelGGL((head_fused_bf16_kernel<4, 8>), grid, block, 0, stream, g);
  else if (s.nc2 == 19) hipLaunchKernelGGL((head_fused_bf16_kernel<19, 38>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((head_fused_bf16_kernel<20, 40>), grid, block, 0, stream, g);
  VS_LAUNCH_CHECK();
  return 0;
}
