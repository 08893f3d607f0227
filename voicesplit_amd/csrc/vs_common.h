// Shared device/host helpers for the VoiceSplit MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/voicesplit_hip.h"   // every translation unit sees the exported declarations (default visibility)

#define VS_WAVE 64

// A kernel built without gfx950's packed-fp32 VALU instructions (v_pk_fma_f32 ...: pairs of scalar instructions instead; same values).
// The attribute names a device subtarget feature: the host pass of the same translation unit does not know it.
#if defined(__HIP_DEVICE_COMPILE__)
#define VS_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define VS_NO_PACKED_FP32
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// bf16 operands of the single-pass matrix-core mode (VS_MATH_BF16): round-to-nearest-even pairs
typedef __bf16 vs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 vs_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned vs_pack_bf16(float a, float b) {
  vs_bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);       // one v_cvt_pk_bf16_f32
}
// arithmetic codes of the dense contractions (dims.math in include/voicesplit_hip.h)
#define VS_MATH_CODE_FP32 0
#define VS_MATH_CODE_F16X3 1
#define VS_MATH_CODE_BF16 2

// activation codes shared with include/voicesplit_hip.h
#define VS_ACT_RELU 0
#define VS_ACT_MISH 1
#define VS_ACT_NONE 2
#define VS_ACT_SIGMOID 3

// Mish, utils/generic_utils.py:395-399: x*tanh(softplus(x)), softplus threshold 20.
// tanh(log(1+u)) = n/(n+2) with n = u(u+2), u = e^x: no cancellation for x << 0 and
// exactly the reference's pass-through for x > 20 (softplus(x)=x, tanh(x)=1 in fp32).
__device__ __forceinline__ float vs_mish(float x) {
  float u = expf(fminf(x, 20.0f));
  float n = u * (u + 2.0f);
  float y = x * (n / (n + 2.0f));
  return x > 20.0f ? x : y;
}

// d/dx Mish = tanh(sp) + x*(1 - tanh(sp)^2)*sigmoid(x); what autograd gives for
// utils/generic_utils.py:399 (softplus' = sigmoid below the threshold, 1 above, where tanh' = 0).
// 1 - tanh(sp) = 2/(n+2) exactly, so (1 - tanh^2) = (2/(n+2))*(1+tanh) has no cancellation.
__device__ __forceinline__ float vs_mish_grad(float x) {
  float u = expf(fminf(x, 20.0f));
  float n = u * (u + 2.0f);
  float inv = 1.0f / (n + 2.0f);
  float tsp = n * inv;
  float sig = u / (1.0f + u);
  float g = tsp + x * (2.0f * inv) * (1.0f + tsp) * sig;
  return x > 20.0f ? 1.0f : g;
}

__device__ __forceinline__ float vs_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ocml tanhf: accurate near 0 (a (1-e)/(1+e) form cancels there)
__device__ __forceinline__ float vs_tanh(float x) { return tanhf(x); }

// The recurrence's gate arithmetic sits on the step-to-step critical path of ONE wave (lstm.hip): 12 sigmoids and
// 8 tanh per lane and step.  The ocml forms (expf + IEEE divide, tanhf) are ~20-25 dependent VALU each -- ~1 us per
// step for a lone wave; these are 4 and 12: v_exp_f32 / v_rcp_f32 (1 ulp each).  sigmoid: <= ~3e-7 relative.  tanh: the
// 1 - 2/(1+e^2x) form cancels near 0 (absolute error one ulp of 1), so |x| < 1/8 takes the odd series up to x^7
// (truncation 62/2835 x^8 < 2e-9 relative there): <= ~1e-6 relative everywhere.
__device__ __forceinline__ float vs_sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
__device__ __forceinline__ float vs_tanh_fast(float x) {
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681472f));
  const float x2 = x * x;
  const float small = x * fmaf(x2, fmaf(x2, fmaf(x2, -17.0f / 315.0f, 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
  return fabsf(x) < 0.125f ? small : big;
}

template <int ACT>
__device__ __forceinline__ float vs_act(float v) {
  if (ACT == VS_ACT_RELU) return fmaxf(v, 0.0f);
  if (ACT == VS_ACT_MISH) return vs_mish(v);
  if (ACT == VS_ACT_SIGMOID) return vs_sigmoid(v);
  return v;
}

// Mish with the hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each, result within ~3e-7
// relative of vs_mish) for epilogues where the accurate expf/divide sequence (~25 VALU per element)
// would rival the matrix work itself.
__device__ __forceinline__ float vs_mish_fast(float x) {
  float u = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.44269504088896340736f);
  float n = u * (u + 2.0f);
  float y = x * n * __builtin_amdgcn_rcpf(n + 2.0f);
  return x > 20.0f ? x : y;
}

// Two channels at a time: on gfx950 v_pk_mul/add/fma_f32 do two fp32 lanes' worth per issue slot, so everything but the
// exp2 / rcp of the pair costs half.  No select for x > 20: n / (n + 2) is exactly 1 in fp32 at the clamp.
typedef float vs_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vs_f32x2 vs_mish_fast2(vs_f32x2 x) {
  const vs_f32x2 e = vs_f32x2{fminf(x.x, 20.0f), fminf(x.y, 20.0f)} * 1.44269504088896340736f;
  const vs_f32x2 u = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
  const vs_f32x2 n = u * (u + 2.0f);
  const vs_f32x2 d = n + 2.0f;
  const vs_f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  return x * (n * r);
}

template <int ACT>
__device__ __forceinline__ vs_f32x2 vs_act_fast2(vs_f32x2 v) {
  if (ACT == VS_ACT_MISH) return vs_mish_fast2(v);
  if (ACT == VS_ACT_RELU) return vs_f32x2{fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f)};
  return v;
}

template <int ACT>
__device__ __forceinline__ float vs_act_fast(float v) {
  if (ACT == VS_ACT_MISH) return vs_mish_fast(v);
  return vs_act<ACT>(v);
}

// vs_mish_grad with v_exp_f32 and ONE v_rcp_f32: 1/(n+2) and u/(1+u) share the reciprocal of
// (n+2)(1+u) (<= 1.2e26 at the x = 20 clamp).  Within ~5e-7 relative of vs_mish_grad; the accurate
// form (expf + two divides, ~60 VALU per element) made the BatchNorm backward passes VALU-bound.
__device__ __forceinline__ float vs_mish_grad_fast(float x) {
  float u = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * 1.44269504088896340736f);
  float n = u * (u + 2.0f);
  float r = __builtin_amdgcn_rcpf((n + 2.0f) * (1.0f + u));
  float inv = (1.0f + u) * r;
  float sig = u * (n + 2.0f) * r;
  float tsp = n * inv;
  float g = tsp + x * (2.0f * inv) * (1.0f + tsp) * sig;
  return x > 20.0f ? 1.0f : g;
}

__device__ __forceinline__ float vs_act_rt(float v, int act) {
  switch (act) {
    case VS_ACT_RELU: return fmaxf(v, 0.0f);
    case VS_ACT_MISH: return vs_mish(v);
    case VS_ACT_SIGMOID: return vs_sigmoid(v);
    default: return v;
  }
}

// Running |max| of a tensor for the split-f16 conv path: kernels that PRODUCE a conv operand fold
// their outputs' magnitude into an array of VS_AMAX_SLOTS uints (bit pattern of a non-negative
// float orders like the float), so the consumer needs no extra pass over the tensor.  The array
// spreads the atomics of ~10^5 workgroups over 1024 addresses (one address serialises them in L2:
// +40 ms per training step when tried) and a plain load first skips the atomic when it cannot
// raise the slot (a stale value only costs a redundant atomic).  out == nullptr: no-op.
// partial-sum slots of the BatchNorm statistics the conv epilogues accumulate ([slot][64 channels][2] doubles)
// Deterministic mode (vs_set_option(VS_OPT_DETERMINISTIC, 1); round 6): the workgroups of a launch that end in atomic additions to
// shared partial-sum slots (BatchNorm statistics and their backward sums, cnn1's moments) take TURNS, so every address receives its
// addends in the same order in every run and a rerun is bit-identical.  Only workgroups that add to the SAME addresses need an
// order among themselves: the turn region of the tape (vs_tape_layout.det_turn, zeroed at the start of a step) holds one word per
// partial-sum slot (VS_TURN_SLOT + slot: the workgroups blockIdx % VS_BN_STAT_SLOTS == slot queue up in index order: chains of
// four in the persistent convs, of 32 behind a 2048-workgroup streaming pass), one per channel (VS_TURN_CHANNEL + c) and one global word (VS_TURN_GLOBAL) for the two small
// kernels whose workgroups all add to one array.  A turn costs ~8 us (an fp64 atomic round trip, a fence, the next one's poll): one
// chain through all 256 workgroups of a conv launch, the first version, doubled its time.  NULL = off (the default: arrival
// order, last bits differ between runs).  A workgroup waits only for workgroups dispatched before it, which never wait for it: no
// co-residency needed; the last one of a chain re-arms the word for the next launch.
#define VS_TURN_SLOT 0          /* words [0, 64) */
#define VS_TURN_CHANNEL 64      /* words [64, 128) */
#define VS_TURN_GLOBAL 128
#define VS_TURN_WORDS 256
__device__ __forceinline__ void vs_turn_begin(unsigned* turn, unsigned me) {
  if (turn == nullptr) return;
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != me) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
__device__ __forceinline__ void vs_turn_end(unsigned* turn, unsigned me, unsigned total) {
  if (turn == nullptr) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_store(turn, me + 1 == total ? 0u : me + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
#define VS_GEMM_KPAD 64        /* rows of pre-split GEMM operands are zero-padded to a multiple of the GEMM's K block (gemm_f16x3.hip) */
#define VS_BN_STAT_SLOTS 64
#define VS_AMAX_SLOTS 1024
__device__ __forceinline__ void vs_absmax_commit(float m, unsigned* out) {
  if (out == nullptr) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) {
    unsigned* slot = out + ((blockIdx.x * 4u + blockIdx.y * 61u + blockIdx.z * 127u + (threadIdx.x >> 6)) & (VS_AMAX_SLOTS - 1));
    const unsigned bits = __float_as_uint(m);
    if (bits > __builtin_nontemporal_load(slot)) atomicMax(slot, bits);
  }
}

// ---------------------------------------------------------------------------
// Streaming row walk of the BatchNorm passes.  A tensor is seen as rows [R][L] (channel = r % C);
// rows are cut into chunks of VS_ROW_CHUNK elements counted from the 16-byte frame of the row
// (position a = i + ph, ph = (address of element 0 / 4) & 3), so that every chunk but a row's first
// starts on a 16-byte boundary and its middle moves as float4 (T*F = 180901 is odd: rows have every
// phase).  ld(ptr-index i, width tag) -> value pack, fin(i, pack) consumes it; two packs are in
// flight per thread.  vec == 0 (operands with different phases): all-scalar walk.
// ---------------------------------------------------------------------------
#define VS_ROW_CHUNK 8192
template <int W> struct VsWidth { static constexpr int value = W; };
__host__ __device__ __forceinline__ int vs_row_chunks(int L) { return (L + 3 + VS_ROW_CHUNK - 1) / VS_ROW_CHUNK; }

// grid.x of the row-walk kernels (grid.y = channel): ~2048 workgroups = 8 per CU in all
static inline int vs_bn_blocks_per_channel(int C, long long rows_c, int L) {
  const long long items = rows_c * vs_row_chunks(L);
  long long nb = (2048 + C - 1) / C;
  if (nb > items) nb = items;
  return nb < 1 ? 1 : (int)nb;
}

template <class LD, class FIN>
__device__ __forceinline__ void vs_walk_chunk(int L, int ph, int k, LD&& ld, FIN&& fin) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const bool vec = ph >= 0;
  if (!vec) ph = 0;
  const int a0 = k * VS_ROW_CHUNK > ph ? k * VS_ROW_CHUNK : ph;
  const int a1 = (k + 1) * VS_ROW_CHUNK < L + ph ? (k + 1) * VS_ROW_CHUNK : L + ph;
  if (a1 <= a0) return;
  const int i0 = a0 - ph, i1 = a1 - ph;
  if (!vec) {
    for (int i = i0 + tid; i < i1; i += nt) fin(i, ld(i, VsWidth<1>()));
    return;
  }
  int head = (4 - (a0 & 3)) & 3;
  if (head > i1 - i0) head = i1 - i0;
  if (tid < head) fin(i0 + tid, ld(i0 + tid, VsWidth<1>()));
  const int v0 = i0 + head, nv = (i1 - v0) >> 2;
  for (int j = tid; j < nv; j += 2 * nt) {
    const int ia = v0 + 4 * j, ib = ia + 4 * nt;
    auto pa = ld(ia, VsWidth<4>());
    if (j + nt < nv) {
      auto pb = ld(ib, VsWidth<4>());
      fin(ia, pa);
      fin(ib, pb);
    } else {
      fin(ia, pa);
    }
  }
  const int t0 = v0 + 4 * nv;
  if (tid < i1 - t0) fin(t0 + tid, ld(t0 + tid, VsWidth<1>()));
}

// W consecutive floats (W = 1 or 4; the 4-wide form needs 16-byte alignment)
template <int W> struct VsPack { static constexpr int N = W; float v[W]; };
template <int W> __device__ __forceinline__ VsPack<W> vs_ldv(const float* p) {
  VsPack<W> r;
  if constexpr (W == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int W> __device__ __forceinline__ void vs_stv(float* p, const VsPack<W>& r) {
  if constexpr (W == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  else p[0] = r.v[0];
}
// phase shared by all operands of a row walk, or -1 when they differ (scalar walk)
__device__ __forceinline__ int vs_row_phase(const void* a, const void* b = nullptr, const void* c = nullptr) {
  const unsigned pa = (unsigned)((uintptr_t)a >> 2) & 3u;
  if (b && ((unsigned)((uintptr_t)b >> 2) & 3u) != pa) return -1;
  if (c && ((unsigned)((uintptr_t)c >> 2) & 3u) != pa) return -1;
  return (int)pa;
}

// ---------------------------------------------------------------------------
// host-side error plumbing: no exceptions cross the C ABI
// ---------------------------------------------------------------------------
void vs_set_error(const char* fmt, ...);

#define VS_CHECK_HIP(expr)                                                        \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      vs_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return -2;                                                                  \
    }                                                                             \
  } while (0)

#define VS_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      vs_set_error(__VA_ARGS__);     \
      return -1;                     \
    }                                \
  } while (0)

#define VS_LAUNCH_CHECK() VS_CHECK_HIP(hipGetLastError())

// ---- partial-sum slots of the BatchNorm statistics (forward: {sum, sum of squares}, backward: {sum dy, sum dy xhat}) --------------------
// stats = [slots][n] doubles, n = 2 C <= 128.  ONE block of 1024 threads folds the slots (8 groups of threads, independent loads:
// the one-thread-per-value loop of rounds 1-4 was 64 dependent L2 round trips = 17-24 us per layer on the critical path), leaves the
// totals in tot[] (LDS, 128 doubles) and, after every read, zeroes `rezero` doubles of the scratch for the next producer (the memset
// in front of every conv launch of a training step -- its own ~5 us kernel + a launch gap -- goes away).
#define VS_FOLD_THREADS 1024
__device__ __forceinline__ void vs_fold_slots(double* stats, int n, int slots, int rezero, double* part /* LDS [8][128] */, double* tot /* LDS [128] */) {
  const int tid = threadIdx.x, i = tid & 127, sg = tid >> 7;
  double v0 = 0.0, v1 = 0.0;
  if (i < n) {
    int k = sg;
    for (; k + 8 < slots; k += 16) { v0 += stats[(size_t)k * n + i]; v1 += stats[(size_t)(k + 8) * n + i]; }
    if (k < slots) v0 += stats[(size_t)k * n + i];
  }
  part[sg * 128 + i] = v0 + v1;
  __syncthreads();
  if (tid < 128) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += part[q * 128 + tid];
    tot[tid] = t;
  }
  __syncthreads();                                   // every load of stats has returned: the scratch may be cleared
  for (int j = tid; j < rezero; j += VS_FOLD_THREADS) stats[j] = 0.0;
}
