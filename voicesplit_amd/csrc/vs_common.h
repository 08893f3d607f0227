// Shared device/host helpers for the VoiceSplit MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define VS_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

template <int ACT>
__device__ __forceinline__ float vs_act_fast(float v) {
  if (ACT == VS_ACT_MISH) return vs_mish_fast(v);
  return vs_act<ACT>(v);
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
