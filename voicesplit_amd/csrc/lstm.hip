// BiLSTM recurrence (the 301 sequential steps of nn.LSTM, models/voicesplit/model.py:57-61,82).
//
// The input projection x_t @ W_ih^T + b_ih + b_hh (+ d-vector fold) for every t and both
// directions is one big GEMM done beforehand (gemm_mfma.hip) into
//   xg[b][t][dir*4H + gate*H + j]          gate order i, f, g, o  (PyTorch)
// What is left per time step and direction is gates = xg_t + h_{t-1} @ W_hh^T followed by
//   c = sigmoid(f)*c + sigmoid(i)*tanh(g);   h = sigmoid(o)*tanh(c)
//
// One launch per time step (both directions, grid.y = dir); the kernel boundary is the
// device-wide dependency between steps.  Workgroup = 4 waves owns 8 hidden units x 32 batch
// rows: the 32x32 MFMA tile has rows i = 8*gate + unit and columns = batch, so after the
// K reduction each lane holds i,f,g,o of the same (unit, batch) in its own accumulator
// registers (D row = (r&3) + 8*(r>>2) + 4*(lane>>5): r>>2 = gate, (r&3)+4*(lane>>5) = unit) and
// the gate math needs no cross-lane traffic.  The K = H reduction is split over the 4 waves and
// combined through LDS.  W_hh and h are both kept in MFMA fragment order (one coalesced dwordx4
// per lane per 4 K-steps) and every load of a step is issued before its first MFMA, so a step
// costs one L2 round trip + 13 x 4 MFMAs per wave; c stays [dir][H][Bpad].
#include "vs_internal.h"

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// packed recurrent weights: [dir][jg = H/8][q = H/8][lane 64][4]; element j of the float4:
//   W_hh[dir][(i>>3)*H + jg*8 + (i&7)][8q + 2j + (lane>>5)],  i = lane & 31
__global__ void lstm_pack_whh_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_b,
                                     float* __restrict__ wp, int H) {
  const int HQ = H / 8;
  const long long total = 2LL * HQ * HQ * 256;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 3;
  const int lane = (idx >> 2) & 63;
  long long rest = idx >> 8;
  const int q = rest % HQ; rest /= HQ;
  const int jg = rest % HQ;
  const int dir = rest / HQ;
  const int i = lane & 31;
  const int row = (i >> 3) * H + jg * 8 + (i & 7);
  const int k = 8 * q + 2 * j + (lane >> 5);
  const float* w = dir ? whh_b : whh_f;
  wp[idx] = w[(size_t)row * H + k];
}

struct LstmStepArgs {
  const float* xg;      // [B][T][8H]
  const float* wp;      // packed W_hh
  const float* h_prev;  // fragment order [2 dir][Bpad/32][H/8][64 lane][4]
  float* h_next;        // same layout
  float* c;             // [2][H][Bpad]  (updated in place: each (unit,batch) has one owner)
  float* out;           // [B][T][2H]
  float* gates_save;    // training: [B][T][8H] activated gates i,f,g,o (may alias xg), else null
  float* c_save;        // training: [B][T][2H] cell state c_t, else null
  int B, T, H, Bpad, step;
};

// h is kept in MFMA B-fragment order so a K-quad is ONE coalesced dwordx4 per lane:
//   hfrag[dir][bt][q][lane][j] = h[dir][k = 8q + 2j + (lane>>5)][b = 32*bt + (lane&31)]
__device__ __forceinline__ size_t hfrag_index(int dir, int nbt, int bt, int hq, int k, int b31) {
  const int q = k >> 3, j = (k & 7) >> 1, half = k & 1;
  return ((((size_t)dir * nbt + bt) * hq + q) * 64 + half * 32 + b31) * 4 + j;
}

constexpr int kMaxQ = 13;   // K-quads per wave held in flight at once (H <= 416 in one pass)

__global__ __launch_bounds__(256)
void lstm_step_kernel(LstmStepArgs a) {
  __shared__ float sRed[3 * 16 * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int HQ = a.H / 8;
  const int NBT = a.Bpad / 32;
  const int jg = blockIdx.x % HQ;
  const int bt = blockIdx.x / HQ;
  const int dir = blockIdx.y;
  const int t = dir ? (a.T - 1 - a.step) : a.step;
  const int b = bt * 32 + l31;
  const size_t hb = (size_t)dir * a.H * a.Bpad;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // every load of this step is issued before the first MFMA: one L2 round trip, not 13
  const float4* wq = reinterpret_cast<const float4*>(a.wp) + ((size_t)(dir * HQ + jg) * HQ) * 64 + lane;
  const float4* hq = reinterpret_cast<const float4*>(a.h_prev) + (((size_t)dir * NBT + bt) * HQ) * 64 + lane;
  const bool recur = a.step > 0;      // h_{-1} = 0: nothing to multiply at the first step
  float4 w4[kMaxQ], h4[kMaxQ];
#pragma unroll
  for (int i = 0; i < kMaxQ; ++i) {
    const int q = wave + 4 * i;
    const bool ok = recur && q < HQ;
    w4[i] = ok ? wq[(size_t)q * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
    h4[i] = ok ? hq[(size_t)q * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // wave 0 owns the epilogue: its xg / c reads travel with the operand loads
  float xgv[16], cprev[4];
  if (wave == 0) {
    const bool ok = b < a.B;
    const float* xrow = a.xg + ((size_t)(ok ? b : 0) * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + jg * 8 + 4 * half;
#pragma unroll
    for (int r = 0; r < 16; ++r) xgv[r] = ok ? xrow[(r >> 2) * a.H + (r & 3)] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) cprev[u] = a.c[hb + (size_t)(jg * 8 + 4 * half + u) * a.Bpad + b];
  }
  if (recur) {
#pragma unroll
    for (int i = 0; i < kMaxQ; ++i) {
      if (wave + 4 * i < HQ) {          // wave-uniform
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].x, h4[i].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].y, h4[i].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].z, h4[i].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].w, h4[i].w, acc, 0, 0, 0);
      }
    }
    // hidden sizes beyond 4*kMaxQ*8 = 416: remaining quads, plain loop
    for (int q = wave + 4 * kMaxQ; q < HQ; q += 4) {
      const float4 w = wq[(size_t)q * 64], h = hq[(size_t)q * 64];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, h.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, h.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, h.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, h.w, acc, 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sRed[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += sRed[(w * 16 + r) * 64 + lane];

  float hv[4], cnew[4], gact[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float gi = vs_sigmoid_fast(acc[0 + u] + xgv[0 + u]);
    const float gf = vs_sigmoid_fast(acc[4 + u] + xgv[4 + u]);
    const float gg = vs_tanh_fast(acc[8 + u] + xgv[8 + u]);
    const float go = vs_sigmoid_fast(acc[12 + u] + xgv[12 + u]);
    const float cn = gf * cprev[u] + gi * gg;
    hv[u] = go * vs_tanh_fast(cn);
    cnew[u] = cn;
    gact[0][u] = gi; gact[1][u] = gf; gact[2][u] = gg; gact[3][u] = go;
    const int k = jg * 8 + 4 * half + u;
    a.c[hb + (size_t)k * a.Bpad + b] = cn;   // padded batch columns only ever hold their own garbage
    a.h_next[hfrag_index(dir, NBT, bt, HQ, k, l31)] = hv[u];
  }
  if (b < a.B) {
    float4* o = reinterpret_cast<float4*>(a.out + ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + jg * 8 + 4 * half);
    *o = make_float4(hv[0], hv[1], hv[2], hv[3]);
    if (a.c_save) {
      float4* cs = reinterpret_cast<float4*>(a.c_save + ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + jg * 8 + 4 * half);
      *cs = make_float4(cnew[0], cnew[1], cnew[2], cnew[3]);
    }
    if (a.gates_save) {
      float* grow = a.gates_save + ((size_t)b * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + jg * 8 + 4 * half;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<float4*>(grow + gq * a.H) = make_float4(gact[gq][0], gact[gq][1], gact[gq][2], gact[gq][3]);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Persistent form of the recurrence: ONE launch for all T steps (north_star: "persistent-RNN style
// kernel"; SURVEY.md 0.5: W_hh is 2.56 MB per direction, so it is partitioned over the CUs and the
// hidden state is exchanged once per step).
//
// Same decomposition and arithmetic as lstm_step_kernel (bit-identical results): workgroup (dir, bt,
// jg) owns 8 hidden units x 32 batch rows for the whole sequence.  What the launch boundary used to
// provide is now explicit:
//  * this workgroup's slice of W_hh (32 gate rows x H) is loaded ONCE and stays in registers
//    (13 float4 per lane and wave at H = 400), c stays in the registers of wave 0;
//  * h travels through L2 in MFMA fragment order: the owner stores its 1 KB quad write-through
//    (sc1), drains its store queue, then raises its own flag word to step+1; a consumer sweeps the
//    H/8 flag words of its (dir, batch tile) group with one relaxed agent-scope load per lane until
//    all carry the epoch, and then reads the quads with sc1 loads (L1-bypassing: the hand-off form
//    of cdna_hip_programming.md Guideline 16 R1 that needs no acquire fence) -- no atomics, no
//    counter serialisation, 50 producers <-> 50 consumers per group, groups never wait for each other;
//  * ping-pong buffers: a workgroup can only overwrite h_{s-1} after every member of its group has
//    published step s, i.e. has finished reading it.
// All workgroups of a launch must be co-resident (grid <= number of CUs, checked by the launcher,
// which otherwise walks the batch tiles in several launches); every spin is bounded and reports
// through *err instead of hanging.
// ---------------------------------------------------------------------------------------------
struct LstmPersistArgs {
  const float* xg;
  const float* wp;
  float* hbuf0;         // fragment order [2 dir][NBT][H/8][64 lane][4]
  float* hbuf1;
  unsigned* flags;      // [2 dir][NBT][H/8] epoch words, zeroed by the launcher before every launch
  unsigned* err;        // set to 1 when a spin gave up
  float* out;
  float* gates_save;
  float* c_save;
  int B, T, H, Bpad, bt0;
};

constexpr unsigned kSpinLimit = 1u << 22;

__global__ __launch_bounds__(256)
void lstm_persistent_kernel(LstmPersistArgs a) {
  __shared__ float sRed[3 * 16 * 64];
  __shared__ int sDead;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int HQ = a.H / 8;
  const int NBT = a.Bpad / 32;
  const int jg = blockIdx.x % HQ;
  const int bt = a.bt0 + blockIdx.x / HQ;
  const int dir = blockIdx.y;
  const int b = bt * 32 + l31;
  if (tid == 0) sDead = 0;

  // W_hh slice: resident for the whole sequence
  const float4* wq = reinterpret_cast<const float4*>(a.wp) + ((size_t)(dir * HQ + jg) * HQ) * 64 + lane;
  float4 w4[kMaxQ];
#pragma unroll
  for (int i = 0; i < kMaxQ; ++i) {
    const int q = wave + 4 * i;
    w4[i] = q < HQ ? wq[(size_t)q * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const size_t group = ((size_t)dir * NBT + bt) * HQ;                     // first quad / flag of this (dir, bt) group
  const unsigned hbytes = (unsigned)((size_t)2 * NBT * HQ * 64 * 16);
  __amdgpu_buffer_rsrc_t hrs[2] = {__builtin_amdgcn_make_buffer_rsrc(a.hbuf0, 0, hbytes, 0x00020000),
                                   __builtin_amdgcn_make_buffer_rsrc(a.hbuf1, 0, hbytes, 0x00020000)};
  unsigned* const gflags = a.flags + group;
  float cprev[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < a.T; ++s) {
    const int t = dir ? (a.T - 1 - s) : s;
    // wave 0 owns the epilogue: its xg reads do not depend on h and go out before the wait
    float xgv[16];
    if (wave == 0) {
      const bool ok = b < a.B;
      const float* xrow = a.xg + ((size_t)(ok ? b : 0) * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + jg * 8 + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) xgv[r] = ok ? xrow[(r >> 2) * a.H + (r & 3)] : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (s > 0) {
      if (wave == 0 && !sDead) {
        // every producer of this group has published h_{s-1} (epoch s)
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
          for (int j = lane; j < HQ; j += 64)
            ok = ok && (__hip_atomic_load(gflags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)s);
          if (__all(ok)) break;
          if (++spins > kSpinLimit) {
            if (lane == 0) { sDead = 1; __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      const unsigned hoff = (unsigned)((group * 64 + lane) * 16);
      float4 h4[kMaxQ];
#pragma unroll
      for (int i = 0; i < kMaxQ; ++i) {
        const int q = wave + 4 * i;
        const u32x4_t v = q < HQ ? __builtin_amdgcn_raw_buffer_load_b128(hrs[s & 1], hoff + (unsigned)q * 1024u, 0, 16 /* sc1 */)
                                 : u32x4_t{0u, 0u, 0u, 0u};
        h4[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
      }
#pragma unroll
      for (int i = 0; i < kMaxQ; ++i) {
        if (wave + 4 * i < HQ) {          // wave-uniform
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].x, h4[i].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].y, h4[i].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].z, h4[i].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].w, h4[i].w, acc, 0, 0, 0);
        }
      }
      // hidden sizes beyond 4*kMaxQ*8 = 416: remaining quads, weights re-read from L2 each step
      for (int q = wave + 4 * kMaxQ; q < HQ; q += 4) {
        const float4 w = wq[(size_t)q * 64];
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(hrs[s & 1], hoff + (unsigned)q * 1024u, 0, 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, __uint_as_float(v[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, __uint_as_float(v[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, __uint_as_float(v[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, __uint_as_float(v[3]), acc, 0, 0, 0);
      }
    }
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sRed[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += sRed[(w * 16 + r) * 64 + lane];
      float hv[4], cnew[4], gact[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float gi = vs_sigmoid_fast(acc[0 + u] + xgv[0 + u]);
        const float gf = vs_sigmoid_fast(acc[4 + u] + xgv[4 + u]);
        const float gg = vs_tanh_fast(acc[8 + u] + xgv[8 + u]);
        const float go = vs_sigmoid_fast(acc[12 + u] + xgv[12 + u]);
        const float cn = gf * cprev[u] + gi * gg;
        hv[u] = go * vs_tanh_fast(cn);
        cnew[u] = cn;
        cprev[u] = cn;
        gact[0][u] = gi; gact[1][u] = gf; gact[2][u] = gg; gact[3][u] = go;
      }
      // h quad jg in fragment order: lane (hl, b) holds units {hl, 2+hl, 4+hl, 6+hl} of this workgroup's
      // 8; this lane computed units 4*half .. 4*half+3 -> two values come from the other half-wave
      {
        const float s0 = half ? hv[0] : hv[1], s1 = half ? hv[2] : hv[3];
        const float r0 = __shfl_xor(s0, 32, 64), r1 = __shfl_xor(s1, 32, 64);
        u32x4_t v;
        v[0] = __float_as_uint(half ? r0 : hv[0]);
        v[1] = __float_as_uint(half ? r1 : hv[2]);
        v[2] = __float_as_uint(half ? hv[1] : r0);
        v[3] = __float_as_uint(half ? hv[3] : r1);
        __builtin_amdgcn_raw_buffer_store_b128(v, hrs[(s + 1) & 1], (unsigned)(((group + jg) * 64 + lane) * 16), 0, 16 /* sc1: write-through */);
      }
      // publish: the quad has left this wave's store queue, then the flag (one lane).  The
      // outputs below are off the step-to-step critical path and go out after the flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(gflags + jg, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (b < a.B) {
        float4* o = reinterpret_cast<float4*>(a.out + ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + jg * 8 + 4 * half);
        *o = make_float4(hv[0], hv[1], hv[2], hv[3]);
        if (a.c_save) {
          float4* cs = reinterpret_cast<float4*>(a.c_save + ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + jg * 8 + 4 * half);
          *cs = make_float4(cnew[0], cnew[1], cnew[2], cnew[3]);
        }
        if (a.gates_save) {
          float* grow = a.gates_save + ((size_t)b * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + jg * 8 + 4 * half;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(grow + gq * a.H) = make_float4(gact[gq][0], gact[gq][1], gact[gq][2], gact[gq][3]);
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Backward through time (the autograd of nn.LSTM, reached from train.py:110 loss.backward()).
// Saved by the training forward: activated gates i,f,g,o [B][T][8H] and cell states [B][T][2H].
// Per step (t descending for the forward direction, ascending for the reverse one):
//   dh  = dOut[t] + W_hh^T dgates_{prev step}            <- the only sequential contraction
//   do' = dh*tanh(c)*o(1-o)          dc = dh*o*(1-tanh(c)^2) + dc_carry
//   di' = dc*g*i(1-i)   df' = dc*c_prev*f(1-f)   dg' = dc*i*(1-g^2)   dc_carry = dc*f
// The gate pre-activation gradients overwrite the saved gates in place ([B][T][8H] = dxg, the
// operand of the big dW_ih / dFeat GEMMs afterwards) and are also written in MFMA B-fragment
// order for the next step's matvec.  Workgroup = 8 waves owns 32 hidden units x 32 batch rows:
// M = units (A = W_hh^T packed in fragment order), N = batch, K = 4H gate rows split over the 8
// waves and combined through LDS; the gate math then runs on 256 threads, one (4 units, batch)
// item each, with float4 accesses along the unit axis.
// ---------------------------------------------------------------------------------------------

// packed W_hh^T: [dir][ut = ceil(H/32)][q = 4H/8][lane 64][4]; element j of the float4:
//   W_hh[dir][r = 8q + 2j + (lane>>5)][unit = 32*ut + (lane&31)]   (0 for unit >= H)
__global__ void lstm_pack_whh_t_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_b,
                                       float* __restrict__ wp, int H) {
  const int NUT = (H + 31) / 32, NQ = H / 2;
  const long long total = 2LL * NUT * NQ * 256;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 3;
  const int lane = (idx >> 2) & 63;
  long long rest = idx >> 8;
  const int q = rest % NQ; rest /= NQ;
  const int ut = rest % NUT;
  const int dir = rest / NUT;
  const int r = 8 * q + 2 * j + (lane >> 5);
  const int unit = ut * 32 + (lane & 31);
  const float* w = dir ? whh_b : whh_f;
  wp[idx] = unit < H ? w[(size_t)r * H + unit] : 0.f;
}

struct LstmBwdArgs {
  const float* wpt;      // packed W_hh^T
  const float* g_prev;   // fragment-order dgates of the previous backward step: [2][NBT][NQ][64][4]
  float* g_next;
  float* gates;          // [B][T][8H]: activated gates in, pre-activation gradients out
  const float* c_all;    // [B][T][2H]
  const float* dout;     // [B][T][2H]
  float* dc;             // [2][Bpad][H] carried dc*f
  int B, T, H, Bpad, step;
};

constexpr int kBwdChunk = 13;   // K-quads per wave in flight at once

__global__ __launch_bounds__(512)
void lstm_bwd_step_kernel(LstmBwdArgs a) {
  __shared__ float sRed[8 * 16 * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NQ = a.H / 2;
  const int NUT = (a.H + 31) / 32;
  const int NBT = a.Bpad / 32;
  const int ut = blockIdx.x % NUT;
  const int bt = blockIdx.x / NUT;
  const int dir = blockIdx.y;
  const int t = dir ? a.step : (a.T - 1 - a.step);
  const int tp = dir ? t + 1 : t - 1;                 // forward-order predecessor (c_{t-1})
  const bool recur = a.step > 0;

  // gate-math operands of this thread's item (waves 0..3): issued before the matvec
  const int b31 = tid & 31, ug = (tid >> 5) & 7;
  const int b = bt * 32 + b31;
  const int u0 = ut * 32 + 4 * ug;
  const bool item = tid < 256 && b < a.B && u0 < a.H;
  float4 gi4, gf4, gg4, go4, c4, cp4, dh4, dcc4;
  gi4 = gf4 = gg4 = go4 = c4 = cp4 = dh4 = dcc4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float* grow = a.gates + ((size_t)(item ? b : 0) * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + (item ? u0 : 0);
  float* dcp = a.dc + ((size_t)dir * a.Bpad + (item ? b : 0)) * a.H + (item ? u0 : 0);
  if (item) {
    gi4 = *reinterpret_cast<const float4*>(grow);
    gf4 = *reinterpret_cast<const float4*>(grow + a.H);
    gg4 = *reinterpret_cast<const float4*>(grow + 2 * a.H);
    go4 = *reinterpret_cast<const float4*>(grow + 3 * a.H);
    const size_t so = ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + u0;
    c4 = *reinterpret_cast<const float4*>(a.c_all + so);
    dh4 = *reinterpret_cast<const float4*>(a.dout + so);
    if (tp >= 0 && tp < a.T)
      cp4 = *reinterpret_cast<const float4*>(a.c_all + ((size_t)b * a.T + tp) * (2 * a.H) + (size_t)dir * a.H + u0);
    if (recur) dcc4 = *reinterpret_cast<const float4*>(dcp);
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (recur) {
    const float4* wq = reinterpret_cast<const float4*>(a.wpt) + ((size_t)(dir * NUT + ut) * NQ) * 64 + lane;
    const float4* gq = reinterpret_cast<const float4*>(a.g_prev) + (((size_t)dir * NBT + bt) * NQ) * 64 + lane;
    for (int base = 0; wave + 8 * base < NQ; base += kBwdChunk) {
      float4 w4[kBwdChunk], g4[kBwdChunk];
#pragma unroll
      for (int i = 0; i < kBwdChunk; ++i) {
        const int q = wave + 8 * (base + i);
        const bool ok = q < NQ;
        w4[i] = ok ? wq[(size_t)q * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        g4[i] = ok ? gq[(size_t)q * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < kBwdChunk; ++i) {
        if (wave + 8 * (base + i) < NQ) {      // wave-uniform
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].x, g4[i].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].y, g4[i].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].z, g4[i].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i].w, g4[i].w, acc, 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) sRed[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (tid >= 256) return;
  // item (ug, b31): D rows 4*ug..4*ug+3 live in lane (ug&1)*32 + b31 = this thread's own lane index,
  // registers 4*(ug>>1) + u with ug>>1 == this thread's wave index
  float dh[4] = {dh4.x, dh4.y, dh4.z, dh4.w};
#pragma unroll
  for (int w = 0; w < 8; ++w)
#pragma unroll
    for (int u = 0; u < 4; ++u) dh[u] += sRed[(w * 16 + 4 * wave + u) * 64 + lane];
  if (!item) return;

  const float gi[4] = {gi4.x, gi4.y, gi4.z, gi4.w}, gf[4] = {gf4.x, gf4.y, gf4.z, gf4.w};
  const float gg[4] = {gg4.x, gg4.y, gg4.z, gg4.w}, go[4] = {go4.x, go4.y, go4.z, go4.w};
  const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
  const float dcc[4] = {dcc4.x, dcc4.y, dcc4.z, dcc4.w};
  float di[4], df[4], dg[4], dO[4], dcn[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float tc = vs_tanh_fast(cc[u]);
    dO[u] = dh[u] * tc * go[u] * (1.f - go[u]);
    const float dc = fmaf(dh[u] * go[u], 1.f - tc * tc, dcc[u]);
    di[u] = dc * gg[u] * gi[u] * (1.f - gi[u]);
    df[u] = dc * cp[u] * gf[u] * (1.f - gf[u]);
    dg[u] = dc * gi[u] * (1.f - gg[u] * gg[u]);
    dcn[u] = dc * gf[u];
  }
  *reinterpret_cast<float4*>(grow) = make_float4(di[0], di[1], di[2], di[3]);
  *reinterpret_cast<float4*>(grow + a.H) = make_float4(df[0], df[1], df[2], df[3]);
  *reinterpret_cast<float4*>(grow + 2 * a.H) = make_float4(dg[0], dg[1], dg[2], dg[3]);
  *reinterpret_cast<float4*>(grow + 3 * a.H) = make_float4(dO[0], dO[1], dO[2], dO[3]);
  *reinterpret_cast<float4*>(dcp) = make_float4(dcn[0], dcn[1], dcn[2], dcn[3]);
  float* gn = a.g_next + (((size_t)dir * NBT + bt) * NQ) * 256;
#pragma unroll
  for (int gate = 0; gate < 4; ++gate) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = gate * a.H + u0 + u;
      const float v = gate == 0 ? di[u] : gate == 1 ? df[u] : gate == 2 ? dg[u] : dO[u];
      gn[((size_t)(r >> 3) * 64 + (r & 1) * 32 + b31) * 4 + ((r & 7) >> 1)] = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Persistent BPTT: lstm_bwd_step_kernel's decomposition and arithmetic (bit-identical results) in ONE
// launch.  Workgroup (dir, bt, ut) keeps its slice of W_hh^T (32 units x 4H rows: 25 float4 per lane
// and wave at H = 400) in registers and the dc carry in the registers of the thread that owns the
// (4 units, batch row) item; the gate gradients travel in MFMA fragment order with the same
// write-through store -> drain -> flag / flag sweep -> sc1 load hand-off as the forward recurrence
// (one flag per STORING WAVE, so no workgroup barrier sits between the stores and the flags).
// ---------------------------------------------------------------------------------------------
struct LstmBwdPersistArgs {
  const float* wpt;
  float* gbuf0;          // fragment-order gate gradients [2 dir][NBT][H/2][64][4]
  float* gbuf1;
  unsigned* flags;       // [2 dir][NBT][NUT*4]
  unsigned* err;
  float* gates;
  const float* c_all;
  const float* dout;
  int B, T, H, Bpad, bt0;
};

constexpr int kBwdResident = 25;   // K-quads of W_hh^T per wave held in registers (H <= 400); the rest streams from L2

__global__ __launch_bounds__(512)
void lstm_bwd_persistent_kernel(LstmBwdPersistArgs a) {
  __shared__ float sRed[8 * 16 * 64];
  __shared__ int sDead;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int NQ = a.H / 2;
  const int NUT = (a.H + 31) / 32;
  const int NBT = a.Bpad / 32;
  const int HQ = a.H / 8;
  const int ut = blockIdx.x % NUT;
  const int bt = a.bt0 + blockIdx.x / NUT;
  const int dir = blockIdx.y;
  if (tid == 0) sDead = 0;

  const int b31 = tid & 31, ug = (tid >> 5) & 7;
  const int b = bt * 32 + b31;
  const int u0 = ut * 32 + 4 * ug;
  const bool units_ok = tid < 256 && u0 < a.H;          // wave-uniform (H % 8 == 0)
  const bool item = units_ok && b < a.B;

  const float4* wq = reinterpret_cast<const float4*>(a.wpt) + ((size_t)(dir * NUT + ut) * NQ) * 64 + lane;
  float4 w4[kBwdResident];
#pragma unroll
  for (int i = 0; i < kBwdResident; ++i) {
    const int q = wave + 8 * i;
    w4[i] = q < NQ ? wq[(size_t)q * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const size_t group = (size_t)dir * NBT + bt;
  const unsigned gbytes = (unsigned)((size_t)2 * NBT * NQ * 64 * 16);
  __amdgpu_buffer_rsrc_t grs[2] = {__builtin_amdgcn_make_buffer_rsrc(a.gbuf0, 0, gbytes, 0x00020000),
                                   __builtin_amdgcn_make_buffer_rsrc(a.gbuf1, 0, gbytes, 0x00020000)};
  const int nflag = NUT * 4;
  unsigned* const gflags = a.flags + group * nflag;
  float dcc[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < a.T; ++s) {
    const int t = dir ? s : (a.T - 1 - s);
    const int tp = dir ? t + 1 : t - 1;                 // forward-order predecessor (c_{t-1})
    // gate-math operands of this thread's item: independent of the exchange, issued before the wait
    float4 gi4, gf4, gg4, go4, c4, cp4, dh4;
    gi4 = gf4 = gg4 = go4 = c4 = cp4 = dh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float* grow = a.gates + ((size_t)(item ? b : 0) * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + (item ? u0 : 0);
    if (item) {
      gi4 = *reinterpret_cast<const float4*>(grow);
      gf4 = *reinterpret_cast<const float4*>(grow + a.H);
      gg4 = *reinterpret_cast<const float4*>(grow + 2 * a.H);
      go4 = *reinterpret_cast<const float4*>(grow + 3 * a.H);
      const size_t so = ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + u0;
      c4 = *reinterpret_cast<const float4*>(a.c_all + so);
      dh4 = *reinterpret_cast<const float4*>(a.dout + so);
      if (tp >= 0 && tp < a.T)
        cp4 = *reinterpret_cast<const float4*>(a.c_all + ((size_t)b * a.T + tp) * (2 * a.H) + (size_t)dir * a.H + u0);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (s > 0) {
      if (wave == 0 && !sDead) {
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
          for (int j = lane; j < nflag; j += 64)
            ok = ok && (__hip_atomic_load(gflags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)s);
          if (__all(ok)) break;
          if (++spins > kSpinLimit) {
            if (lane == 0) { sDead = 1; __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      const unsigned goff = (unsigned)((group * NQ * 64 + lane) * 16);
      // resident quads in register batches
      constexpr int kBatch = 9;        // loads in flight per wave (25 = 9 + 9 + 7); same MFMA order as the step kernel
#pragma unroll
      for (int base = 0; base < kBwdResident; base += kBatch) {
        u32x4_t g4[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
          const int q = wave + 8 * (base + i);
          g4[i] = (base + i < kBwdResident && q < NQ) ? __builtin_amdgcn_raw_buffer_load_b128(grs[s & 1], goff + (unsigned)q * 1024u, 0, 16 /* sc1 */)
                                                      : u32x4_t{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
          if (base + i < kBwdResident && wave + 8 * (base + i) < NQ) {      // wave-uniform
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[base + i].x, __uint_as_float(g4[i][0]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[base + i].y, __uint_as_float(g4[i][1]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[base + i].z, __uint_as_float(g4[i][2]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[base + i].w, __uint_as_float(g4[i][3]), acc, 0, 0, 0);
          }
        }
      }
      for (int q = wave + 8 * kBwdResident; q < NQ; q += 8) {        // H > 400: weights re-read from L2 each step
        const float4 w = wq[(size_t)q * 64];
        const u32x4_t g = __builtin_amdgcn_raw_buffer_load_b128(grs[s & 1], goff + (unsigned)q * 1024u, 0, 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, __uint_as_float(g[0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, __uint_as_float(g[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, __uint_as_float(g[2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, __uint_as_float(g[3]), acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sRed[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (tid < 256) {
      float dh[4] = {dh4.x, dh4.y, dh4.z, dh4.w};
#pragma unroll
      for (int w = 0; w < 8; ++w)
#pragma unroll
        for (int u = 0; u < 4; ++u) dh[u] += sRed[(w * 16 + 4 * wave + u) * 64 + lane];
      const float gi[4] = {gi4.x, gi4.y, gi4.z, gi4.w}, gf[4] = {gf4.x, gf4.y, gf4.z, gf4.w};
      const float gg[4] = {gg4.x, gg4.y, gg4.z, gg4.w}, go[4] = {go4.x, go4.y, go4.z, go4.w};
      const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
      float dg4[4][4];        // [gate i,f,g,o][unit]
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float tc = vs_tanh_fast(cc[u]);
        dg4[3][u] = dh[u] * tc * go[u] * (1.f - go[u]);
        const float dc = fmaf(dh[u] * go[u], 1.f - tc * tc, dcc[u]);
        dg4[0][u] = dc * gg[u] * gi[u] * (1.f - gi[u]);
        dg4[1][u] = dc * cp[u] * gf[u] * (1.f - gf[u]);
        dg4[2][u] = dc * gi[u] * (1.f - gg[u] * gg[u]);
        dcc[u] = dc * gf[u];
      }
      if (!item) {
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
          for (int u = 0; u < 4; ++u) dg4[gate][u] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) dcc[u] = 0.f;
      }
      // fragment-order copy for the next step: quad (gate, ut, wave) = rows gate*H + ut*32 + 8*wave .. +7,
      // lane (hl, b) holds rows {hl, 2+hl, 4+hl, 6+hl}; this lane computed 4*half .. 4*half+3
      if (units_ok) {
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
          const float s0 = half ? dg4[gate][0] : dg4[gate][1], s1 = half ? dg4[gate][2] : dg4[gate][3];
          const float r0 = __shfl_xor(s0, 32, 64), r1 = __shfl_xor(s1, 32, 64);
          u32x4_t v;
          v[0] = __float_as_uint(half ? r0 : dg4[gate][0]);
          v[1] = __float_as_uint(half ? r1 : dg4[gate][2]);
          v[2] = __float_as_uint(half ? dg4[gate][1] : r0);
          v[3] = __float_as_uint(half ? dg4[gate][3] : r1);
          const size_t quad = group * NQ + (size_t)gate * HQ + ut * 4 + wave;
          __builtin_amdgcn_raw_buffer_store_b128(v, grs[(s + 1) & 1], (unsigned)((quad * 64 + lane) * 16), 0, 16 /* sc1 */);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(gflags + ut * 4 + wave, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (item) {           // the batched GEMMs' operand, in place of the saved gates
        *reinterpret_cast<float4*>(grow) = make_float4(dg4[0][0], dg4[0][1], dg4[0][2], dg4[0][3]);
        *reinterpret_cast<float4*>(grow + a.H) = make_float4(dg4[1][0], dg4[1][1], dg4[1][2], dg4[1][3]);
        *reinterpret_cast<float4*>(grow + 2 * a.H) = make_float4(dg4[2][0], dg4[2][1], dg4[2][2], dg4[2][3]);
        *reinterpret_cast<float4*>(grow + 3 * a.H) = make_float4(dg4[3][0], dg4[3][1], dg4[3][2], dg4[3][3]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// The recurrent products on the f16 / bf16 matrix instructions (dims.math = VS_MATH_F16X3 / VS_MATH_BF16).
//
// v_mfma_f32_32x32x2_f32 retires 2 of K per 64 cycles, v_mfma_f32_32x32x16_{f16,bf16} 16 per 32: the 52 fp32 MFMAs a
// wave issues per forward step (100 per BPTT step) are 1.5 us (3 us) of a 6.8 us (12.4 us) step whose other parts are
// hand-off latency.  Same decomposition, flags and hand-off protocol as the kernels above; what changes is the operand
// form of the exchanged vector and of the resident weights:
//   * K runs in chunks of 16; lane (half, n) of a chunk holds k = 16c + 8*half + j, j = 0..7, as ONE 16-byte vector
//     (A: rows = n, B: columns = n -- the same map on both sides, so the dot product is the plain one);
//   * forward, NP = 1 (VS_MATH_BF16): h and W_hh rounded to f16 (11 significant bits; |h| < 1, no range problem) -- one
//     product per chunk.  NP = 2 (VS_MATH_F16X3): both split into f16 hi + lo planes (h scaled by 2^10, W_hh by the
//     power of two that puts max|W_hh| into [2^9, 2^10)), three products per chunk, the 2^-22 lo x lo term dropped:
//     fp32-class, as in the split-f16 convs and GEMMs;
//   * BPTT (VS_MATH_BF16 only): gate gradients and W_hh^T as bf16 -- gradients have no a-priori range, so the 8-bit
//     exponent is the right 16-bit form; one product per chunk.  The fp32-class arithmetic keeps the fp32 MFMA BPTT.
// A producer owns 8 consecutive k (8 hidden units / 8 gate rows x 32 batch columns) = exactly one half-chunk: after a
// half-wave exchange each lane holds its column's 8 values, the lower half-wave stores the hi plane (16 B per lane),
// the upper one the lo plane.  The exchange shrinks from 4 to 2 (x NP) bytes per value.
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr float kHScale = 1024.f;     // h of the split form is exchanged as f16(h * 2^10)
constexpr int kMaxC = 7;              // 16-wide K chunks per wave held in registers (H <= 448)

// f16 forms of W_hh for the forward recurrence: [dir][jg = H/8][c = ceil(H/16)][p < NP][lane 64] x 16 bytes; element j:
//   W_hh[dir][(i>>3)*H + jg*8 + (i&7)][16c + 8*(lane>>5) + j] * s,  i = lane & 31   (0 beyond H)
// hdr: [1] = s (NP = 2: from vs_scale_from_absmax_impl; NP = 1: unused) -> [0] = what the accumulators are multiplied by
template <int NP>
__global__ void lstm_pack16_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_b, u32x4_t* __restrict__ wp,
                                   float* __restrict__ hdr, int H) {
  const int HQ = H / 8, NC = (H + 15) / 16;
  const long long total = 2LL * HQ * NC * 64;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const float s = NP == 2 ? hdr[1] : 1.f;
  if (idx == 0) hdr[0] = NP == 2 ? hdr[2] * (1.f / kHScale) : 1.f;
  if (idx >= total) return;
  const int lane = idx & 63;
  long long rest = idx >> 6;
  const int c = rest % NC; rest /= NC;
  const int jg = rest % HQ;
  const int dir = rest / HQ;
  const int i = lane & 31;
  const int row = (i >> 3) * H + jg * 8 + (i & 7);
  const int k0 = 16 * c + 8 * (lane >> 5);
  const float* w = (dir ? whh_b : whh_f) + (size_t)row * H;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = k0 + j < H ? w[k0 + j] * s : 0.f;
  u32x4_t hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (NP == 2) {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x[2 * j], x[2 * j + 1]));
      hi[j] = __builtin_bit_cast(unsigned, h);
      lo[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x[2 * j] - (float)h[0], x[2 * j + 1] - (float)h[1]));
    } else {
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      const h2 h = {(_Float16)x[2 * j], (_Float16)x[2 * j + 1]};
      hi[j] = __builtin_bit_cast(unsigned, h);
    }
  }
  u32x4_t* o = wp + ((size_t)((dir * HQ + jg) * NC + c) * NP) * 64 + lane;
  o[0] = hi;
  if (NP == 2) o[64] = lo;
}

// bf16 form of W_hh^T for the BPTT: [dir][ut = ceil(H/32)][c = H/4][lane 64] x 16 bytes; element j:
//   W_hh[dir][r = 16c + 8*(lane>>5) + j][unit = 32*ut + (lane&31)]   (0 for unit >= H)
__global__ void lstm_pack16_t_kernel(const float* __restrict__ whh_f, const float* __restrict__ whh_b, u32x4_t* __restrict__ wp, int H) {
  const int NUT = (H + 31) / 32, NCt = H / 4;
  const long long total = 2LL * NUT * NCt * 64;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63;
  long long rest = idx >> 6;
  const int c = rest % NCt; rest /= NCt;
  const int ut = rest % NUT;
  const int dir = rest / NUT;
  const int unit = ut * 32 + (lane & 31);
  const int r0 = 16 * c + 8 * (lane >> 5);
  const float* w = dir ? whh_b : whh_f;
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)(unit < H ? w[(size_t)(r0 + j) * H + unit] : 0.f);
  wp[idx] = __builtin_bit_cast(u32x4_t, v);
}

struct Lstm16Args {
  const float* xg;
  const u32x4_t* wp;    // lstm_pack16_kernel<NP>
  const float* hdr;     // hdr[0]: accumulator scale
  void* hbuf0;          // [2 dir][NBT][NC][NP][64 lane] x 16 bytes
  void* hbuf1;
  unsigned* flags;      // [2 dir][NBT][H/8] epoch words
  unsigned* err;
  float* out;
  float* gates_save;
  float* c_save;
  int B, T, H, Bpad, bt0;
};

template <int NP>
__global__ __launch_bounds__(256)
void lstm16_persistent_kernel(Lstm16Args a) {
  __shared__ float sRed[3 * 16 * 64];
  __shared__ int sDead;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int HQ = a.H / 8;
  const int NC = (a.H + 15) / 16;
  const int NBT = a.Bpad / 32;
  const int jg = blockIdx.x % HQ;
  const int bt = a.bt0 + blockIdx.x / HQ;
  const int dir = blockIdx.y;
  const int b = bt * 32 + l31;
  if (tid == 0) sDead = 0;

  // W_hh slice: resident for the whole sequence
  const u32x4_t* wq = a.wp + ((size_t)(dir * HQ + jg) * NC * NP) * 64 + lane;
  f16x8 w[kMaxC][NP];
#pragma unroll
  for (int i = 0; i < kMaxC; ++i) {
    const int c = wave + 4 * i;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      w[i][p] = __builtin_bit_cast(f16x8, c < NC ? wq[(size_t)(c * NP + p) * 64] : u32x4_t{0u, 0u, 0u, 0u});
  }
  const float inv = a.hdr[0];
  const size_t group = (size_t)dir * NBT + bt;
  const unsigned hbytes = (unsigned)((size_t)2 * NBT * NC * NP * 1024);
  __amdgpu_buffer_rsrc_t hrs[2] = {__builtin_amdgcn_make_buffer_rsrc(a.hbuf0, 0, hbytes, 0x00020000),
                                   __builtin_amdgcn_make_buffer_rsrc(a.hbuf1, 0, hbytes, 0x00020000)};
  unsigned* const gflags = a.flags + group * HQ;
  float cprev[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < a.T; ++s) {
    const int t = dir ? (a.T - 1 - s) : s;
    float xgv[16];
    if (wave == 0) {
      const bool ok = b < a.B;
      const float* xrow = a.xg + ((size_t)(ok ? b : 0) * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + jg * 8 + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) xgv[r] = ok ? xrow[(r >> 2) * a.H + (r & 3)] : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (s > 0) {
      if (wave == 0 && !sDead) {
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
          for (int j = lane; j < HQ; j += 64)
            ok = ok && (__hip_atomic_load(gflags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)s);
          if (__all(ok)) break;
          if (++spins > kSpinLimit) {
            if (lane == 0) { sDead = 1; __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      const unsigned hoff = (unsigned)((group * NC * NP * 64 + lane) * 16);
      f16x8 h[kMaxC][NP];
#pragma unroll
      for (int i = 0; i < kMaxC; ++i) {
        const int c = wave + 4 * i;
#pragma unroll
        for (int p = 0; p < NP; ++p)
          h[i][p] = __builtin_bit_cast(f16x8, c < NC ? __builtin_amdgcn_raw_buffer_load_b128(hrs[s & 1], hoff + (unsigned)(c * NP + p) * 1024u, 0, 16 /* sc1 */)
                                                       : u32x4_t{0u, 0u, 0u, 0u});
      }
#pragma unroll
      for (int i = 0; i < kMaxC; ++i) {
        if (wave + 4 * i < NC) {          // wave-uniform
          if (NP == 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i][NP - 1], h[i][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i][0], h[i][NP - 1], acc, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i][0], h[i][0], acc, 0, 0, 0);
        }
      }
      // hidden sizes beyond 4*kMaxC*16 = 448: remaining chunks, weights re-read from L2 each step
      for (int c = wave + 4 * kMaxC; c < NC; c += 4) {
        f16x8 wc[NP], hc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          wc[p] = __builtin_bit_cast(f16x8, wq[(size_t)(c * NP + p) * 64]);
          hc[p] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(hrs[s & 1], hoff + (unsigned)(c * NP + p) * 1024u, 0, 16));
        }
        if (NP == 2) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[NP - 1], hc[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], hc[NP - 1], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], hc[0], acc, 0, 0, 0);
      }
    }
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sRed[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w3 = 0; w3 < 3; ++w3)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += sRed[(w3 * 16 + r) * 64 + lane];
      float hv[4], cnew[4], gact[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float gi = vs_sigmoid_fast(fmaf(acc[0 + u], inv, xgv[0 + u]));
        const float gf = vs_sigmoid_fast(fmaf(acc[4 + u], inv, xgv[4 + u]));
        const float gg = vs_tanh_fast(fmaf(acc[8 + u], inv, xgv[8 + u]));
        const float go = vs_sigmoid_fast(fmaf(acc[12 + u], inv, xgv[12 + u]));
        const float cn = gf * cprev[u] + gi * gg;
        hv[u] = go * vs_tanh_fast(cn);
        cnew[u] = cn;
        cprev[u] = cn;
        gact[0][u] = gi; gact[1][u] = gf; gact[2][u] = gg; gact[3][u] = go;
      }
      // this lane computed units 4*half .. 4*half+3 of the workgroup's 8 for batch column l31; after the half-wave
      // exchange every lane holds all 8 = the half-chunk (chunk jg>>1, k-half jg&1) of its column
      {
        float full[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float o = __shfl_xor(hv[u], 32, 64);
          full[u] = half ? o : hv[u];
          full[4 + u] = half ? hv[u] : o;
        }
        u32x4_t v;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (NP == 2) {
            const float x0 = full[2 * j] * kHScale, x1 = full[2 * j + 1] * kHScale;
            const h2 hh = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
            const unsigned lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0 - (float)hh[0], x1 - (float)hh[1]));
            v[j] = half ? lo : __builtin_bit_cast(unsigned, hh);
          } else {
            const h2 hh = {(_Float16)full[2 * j], (_Float16)full[2 * j + 1]};
            v[j] = __builtin_bit_cast(unsigned, hh);
          }
        }
        const unsigned plane = NP == 2 ? (unsigned)half : 0u;
        const unsigned off = (unsigned)((((group * NC + (jg >> 1)) * NP + plane) * 64 + (jg & 1) * 32 + l31) * 16);
        if (NP == 2 || half == 0)
          __builtin_amdgcn_raw_buffer_store_b128(v, hrs[(s + 1) & 1], off, 0, 16 /* sc1: write-through */);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(gflags + jg, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (b < a.B) {
        float4* o = reinterpret_cast<float4*>(a.out + ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + jg * 8 + 4 * half);
        *o = make_float4(hv[0], hv[1], hv[2], hv[3]);
        if (a.c_save) {
          float4* cs = reinterpret_cast<float4*>(a.c_save + ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + jg * 8 + 4 * half);
          *cs = make_float4(cnew[0], cnew[1], cnew[2], cnew[3]);
        }
        if (a.gates_save) {
          float* grow = a.gates_save + ((size_t)b * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + jg * 8 + 4 * half;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(grow + gq * a.H) = make_float4(gact[gq][0], gact[gq][1], gact[gq][2], gact[gq][3]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same recurrence with the hidden vector as its OWN flag (round 5): one agent-scope round trip per step instead
// of two.  The flag protocol above costs, per step and on the critical path: store h (write-through) -> wait for the store to
// drain -> store the flag -> [consumer] see the flag -> load h: two dependent round trips plus the drain.  Here:
//   * FOUR exchange buffers; h_s lives in buffer s % 4.  A slot that has not been written yet holds the SENTINEL 0x7FFF7FFF in
//     every dword -- two 16-bit NaNs, a pattern no finite h (hi or lo plane; bf16 gate gradient in the BPTT) produces;
//   * a consumer wave polls the chunks IT multiplies (sc1 loads, as before) until no dword of them is the sentinel: a dword is
//     written by one store instruction, so it is either old (sentinel) or complete; no flag word, no workgroup barrier in front
//     of the MFMAs;
//   * the producer of a slot re-arms it.  At step s, behind the workgroup barrier of the K reduction -- EVERY wave's poll of h_s has
//     succeeded, so every member of the group has published h_s, i.e. has finished step s - 1 and with it every read of buffer
//     (s - 1) % 4 -- it waits for its older stores (the re-arming store of the previous step among them: a whole step old, the
//     wait is free), stores the sentinel to ITS slot of buffer (s - 1) % 4 = (s + 3) % 4, does the gate arithmetic and stores
//     h_{s+1} into buffer (s + 1) % 4 with nothing behind it.  Why nobody can read stale data: buffer X % 4 is polled for h_X by a
//     consumer that has finished step X - 1, for which it needed this workgroup's h_{X-1}; that was stored at step X - 2 behind the
//     wait that acknowledged the re-arming store of step X - 3, whose target was (X - 3 + 3) % 4 = X % 4.  (Three buffers would do
//     with the acknowledgement exposed inside the step; a re-arm by a wave that has only seen ITS chunks arrive -- the first version
//     -- is a race: a slow member may still be reading the slot.)
// Every spin is bounded and reports through *err (NaN poison + vs_lstm_status as before).  H <= 448 (all chunks register-resident);
// larger hidden sizes take the flag kernel.
// ---------------------------------------------------------------------------------------------
constexpr unsigned kSentinel = 0x7FFF7FFFu;

template <int NP>
__global__ __launch_bounds__(256)
void lstm16_tagged_kernel(Lstm16Args a, void* hbuf2, void* hbuf3) {
  __shared__ float sRed[2][3 * 16 * 64];      // by step parity: no barrier separates wave 0's reads of step s from the other waves' writes of step s + 1
  __shared__ int sDead;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int HQ = a.H / 8;
  const int NC = (a.H + 15) / 16;
  const int NBT = a.Bpad / 32;
  const int jg = blockIdx.x % HQ;
  const int bt = a.bt0 + blockIdx.x / HQ;
  const int dir = blockIdx.y;
  const int b = bt * 32 + l31;
  if (tid == 0) sDead = 0;

  const u32x4_t* wq = a.wp + ((size_t)(dir * HQ + jg) * NC * NP) * 64 + lane;
  f16x8 w[kMaxC][NP];
  bool owned[kMaxC];                     // does a producer write this lane's half-chunk?  (the upper half of the last chunk when H % 16 == 8: never)
#pragma unroll
  for (int i = 0; i < kMaxC; ++i) {
    const int c = wave + 4 * i;
    owned[i] = c < NC && 16 * c + 8 * half < a.H;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      w[i][p] = __builtin_bit_cast(f16x8, c < NC ? wq[(size_t)(c * NP + p) * 64] : u32x4_t{0u, 0u, 0u, 0u});
  }
  const float inv = a.hdr[0];
  const size_t group = (size_t)dir * NBT + bt;
  const unsigned hbytes = (unsigned)((size_t)2 * NBT * NC * NP * 1024);
  __amdgpu_buffer_rsrc_t hrs[4] = {__builtin_amdgcn_make_buffer_rsrc(a.hbuf0, 0, hbytes, 0x00020000),
                                   __builtin_amdgcn_make_buffer_rsrc(a.hbuf1, 0, hbytes, 0x00020000),
                                   __builtin_amdgcn_make_buffer_rsrc(hbuf2, 0, hbytes, 0x00020000),
                                   __builtin_amdgcn_make_buffer_rsrc(hbuf3, 0, hbytes, 0x00020000)};
  // this lane's slot (wave 0 stores it): chunk jg >> 1, k-half jg & 1, plane = lane half (NP == 2) / lower half-wave only (NP == 1)
  const unsigned plane = NP == 2 ? (unsigned)half : 0u;
  const unsigned slot_off = (unsigned)((((group * NC + (jg >> 1)) * NP + plane) * 64 + (jg & 1) * 32 + l31) * 16);
  const bool storer = NP == 2 || half == 0;
  float cprev[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < a.T; ++s) {
    const int t = dir ? (a.T - 1 - s) : s;
    const int rb = s & 3, wb = (s + 1) & 3, zb = (s + 3) & 3;            // h_s is read from rb, h_{s+1} goes to wb, zb (= h_{s-1}'s) is re-armed
    float xgv[16];
    if (wave == 0) {
      const bool ok = b < a.B;
      const float* xrow = a.xg + ((size_t)(ok ? b : 0) * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + jg * 8 + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) xgv[r] = ok ? xrow[(r >> 2) * a.H + (r & 3)] : 0.f;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (s > 0) {
      const unsigned hoff = (unsigned)((group * NC * NP * 64 + lane) * 16);
      f16x8 h[kMaxC][NP];
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < kMaxC; ++i) {
          const int c = wave + 4 * i;
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            u32x4_t v = c < NC ? __builtin_amdgcn_raw_buffer_load_b128(hrs[rb], hoff + (unsigned)(c * NP + p) * 1024u, 0, 16 /* sc1 */)
                               : u32x4_t{0u, 0u, 0u, 0u};
            if (!owned[i]) v = u32x4_t{0u, 0u, 0u, 0u};
            ok = ok && v[0] != kSentinel && v[1] != kSentinel && v[2] != kSentinel && v[3] != kSentinel;
            h[i][p] = __builtin_bit_cast(f16x8, v);
          }
        }
        if (__all(ok)) break;
        if (++spins > kSpinLimit / 4 || *(volatile int*)&sDead) {      // (a poll round here is 7-13 loads, not one: the same give-up time as the flag kernels')
          if (lane == 0) { sDead = 1; __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int i = 0; i < kMaxC; ++i) {
        if (wave + 4 * i < NC) {          // wave-uniform
          if (NP == 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i][NP - 1], h[i][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i][0], h[i][NP - 1], acc, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[i][0], h[i][0], acc, 0, 0, 0);
        }
      }
    }
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sRed[s & 1][((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    // wave 1 is the re-arming wave (wave 0's store queue carries h: a sentinel store in front of it would delay it).  Its re-arming
    // store of the previous step is acknowledged HERE, in front of the barrier behind which wave 0 stores this step's h (see the header)
    if (wave == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // every wave's poll has succeeded: the group is done with buffer zb
    if (wave == 1 && s > 0 && storer)
      __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{kSentinel, kSentinel, kSentinel, kSentinel}, hrs[zb], slot_off, 0, 16);
    if (wave == 0) {
#pragma unroll
      for (int w3 = 0; w3 < 3; ++w3)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += sRed[s & 1][(w3 * 16 + r) * 64 + lane];
      float hv[4], cnew[4], gact[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float gi = vs_sigmoid_fast(fmaf(acc[0 + u], inv, xgv[0 + u]));
        const float gf = vs_sigmoid_fast(fmaf(acc[4 + u], inv, xgv[4 + u]));
        const float gg = vs_tanh_fast(fmaf(acc[8 + u], inv, xgv[8 + u]));
        const float go = vs_sigmoid_fast(fmaf(acc[12 + u], inv, xgv[12 + u]));
        const float cn = gf * cprev[u] + gi * gg;
        hv[u] = go * vs_tanh_fast(cn);
        cnew[u] = cn;
        cprev[u] = cn;
        gact[0][u] = gi; gact[1][u] = gf; gact[2][u] = gg; gact[3][u] = go;
      }
      {
        float full[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float o = __shfl_xor(hv[u], 32, 64);
          full[u] = half ? o : hv[u];
          full[4 + u] = half ? hv[u] : o;
        }
        u32x4_t v;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (NP == 2) {
            const float x0 = full[2 * j] * kHScale, x1 = full[2 * j + 1] * kHScale;
            const h2 hh = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
            const unsigned lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0 - (float)hh[0], x1 - (float)hh[1]));
            v[j] = half ? lo : __builtin_bit_cast(unsigned, hh);
          } else {
            const h2 hh = {(_Float16)full[2 * j], (_Float16)full[2 * j + 1]};
            v[j] = __builtin_bit_cast(unsigned, hh);
          }
          // A diverged step (h = NaN with an all-ones payload in both halves) must not look like an unwritten slot: the consumers
          // would spin to their bound and report "gave up" instead of carrying the NaN to the loss guard.
          v[j] = v[j] == kSentinel ? 0x7FFF7E00u : v[j];
        }
        if (storer) __builtin_amdgcn_raw_buffer_store_b128(v, hrs[wb], slot_off, 0, 16 /* sc1: write-through */);
      }
      if (b < a.B) {
        float4* o = reinterpret_cast<float4*>(a.out + ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + jg * 8 + 4 * half);
        *o = make_float4(hv[0], hv[1], hv[2], hv[3]);
        if (a.c_save) {
          float4* cs = reinterpret_cast<float4*>(a.c_save + ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + jg * 8 + 4 * half);
          *cs = make_float4(cnew[0], cnew[1], cnew[2], cnew[3]);
        }
        if (a.gates_save) {
          float* grow = a.gates_save + ((size_t)b * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + jg * 8 + 4 * half;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<float4*>(grow + gq * a.H) = make_float4(gact[gq][0], gact[gq][1], gact[gq][2], gact[gq][3]);
        }
      }
    }
  }
}

struct Lstm16BwdArgs {
  const u32x4_t* wpt;    // lstm_pack16_t_kernel
  void* gbuf0;           // bf16 gate gradients [2 dir][NBT][H/4][64 lane] x 16 bytes
  void* gbuf1;
  unsigned* flags;       // [2 dir][NBT][NUT*4]
  unsigned* err;
  float* gates;
  const float* c_all;
  const float* dout;
  int B, T, H, Bpad, bt0;
};

constexpr int kBwdRes16 = 13;   // 16-wide K chunks of W_hh^T per wave held in registers (4H <= 1664); the rest streams from L2

__global__ __launch_bounds__(512)
void lstm16_bwd_persistent_kernel(Lstm16BwdArgs a) {
  __shared__ float sRed[8 * 16 * 64];
  __shared__ int sDead;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int NCt = a.H / 4;
  const int NUT = (a.H + 31) / 32;
  const int NBT = a.Bpad / 32;
  const int ut = blockIdx.x % NUT;
  const int bt = a.bt0 + blockIdx.x / NUT;
  const int dir = blockIdx.y;
  if (tid == 0) sDead = 0;

  const int b31 = tid & 31, ug = (tid >> 5) & 7;
  const int b = bt * 32 + b31;
  const int u0 = ut * 32 + 4 * ug;
  const bool units_ok = tid < 256 && u0 < a.H;          // wave-uniform (H % 8 == 0)
  const bool item = units_ok && b < a.B;

  const u32x4_t* wq = a.wpt + ((size_t)(dir * NUT + ut) * NCt) * 64 + lane;
  bf16x8 w[kBwdRes16];
#pragma unroll
  for (int i = 0; i < kBwdRes16; ++i) {
    const int c = wave + 8 * i;
    w[i] = __builtin_bit_cast(bf16x8, c < NCt ? wq[(size_t)c * 64] : u32x4_t{0u, 0u, 0u, 0u});
  }
  const size_t group = (size_t)dir * NBT + bt;
  const unsigned gbytes = (unsigned)((size_t)2 * NBT * NCt * 1024);
  __amdgpu_buffer_rsrc_t grs[2] = {__builtin_amdgcn_make_buffer_rsrc(a.gbuf0, 0, gbytes, 0x00020000),
                                   __builtin_amdgcn_make_buffer_rsrc(a.gbuf1, 0, gbytes, 0x00020000)};
  const int nflag = NUT * 4;
  unsigned* const gflags = a.flags + group * nflag;
  float dcc[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < a.T; ++s) {
    const int t = dir ? s : (a.T - 1 - s);
    const int tp = dir ? t + 1 : t - 1;                 // forward-order predecessor (c_{t-1})
    float4 gi4, gf4, gg4, go4, c4, cp4, dh4;
    gi4 = gf4 = gg4 = go4 = c4 = cp4 = dh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float* grow = a.gates + ((size_t)(item ? b : 0) * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + (item ? u0 : 0);
    if (item) {
      gi4 = *reinterpret_cast<const float4*>(grow);
      gf4 = *reinterpret_cast<const float4*>(grow + a.H);
      gg4 = *reinterpret_cast<const float4*>(grow + 2 * a.H);
      go4 = *reinterpret_cast<const float4*>(grow + 3 * a.H);
      const size_t so = ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + u0;
      c4 = *reinterpret_cast<const float4*>(a.c_all + so);
      dh4 = *reinterpret_cast<const float4*>(a.dout + so);
      if (tp >= 0 && tp < a.T)
        cp4 = *reinterpret_cast<const float4*>(a.c_all + ((size_t)b * a.T + tp) * (2 * a.H) + (size_t)dir * a.H + u0);
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (s > 0) {
      if (wave == 0 && !sDead) {
        unsigned spins = 0;
        for (;;) {
          bool ok = true;
          for (int j = lane; j < nflag; j += 64)
            ok = ok && (__hip_atomic_load(gflags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)s);
          if (__all(ok)) break;
          if (++spins > kSpinLimit) {
            if (lane == 0) { sDead = 1; __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      const unsigned goff = (unsigned)((group * NCt * 64 + lane) * 16);
      bf16x8 g[kBwdRes16];
#pragma unroll
      for (int i = 0; i < kBwdRes16; ++i) {
        const int c = wave + 8 * i;
        g[i] = __builtin_bit_cast(bf16x8, c < NCt ? __builtin_amdgcn_raw_buffer_load_b128(grs[s & 1], goff + (unsigned)c * 1024u, 0, 16 /* sc1 */)
                                                  : u32x4_t{0u, 0u, 0u, 0u});
      }
#pragma unroll
      for (int i = 0; i < kBwdRes16; ++i) {
        if (wave + 8 * i < NCt)        // wave-uniform
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i], g[i], acc, 0, 0, 0);
      }
      for (int c = wave + 8 * kBwdRes16; c < NCt; c += 8) {        // H > 416: weights re-read from L2 each step
        const bf16x8 wc = __builtin_bit_cast(bf16x8, wq[(size_t)c * 64]);
        const bf16x8 gc = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(grs[s & 1], goff + (unsigned)c * 1024u, 0, 16));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc, gc, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sRed[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (tid < 256) {
      float dh[4] = {dh4.x, dh4.y, dh4.z, dh4.w};
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8)
#pragma unroll
        for (int u = 0; u < 4; ++u) dh[u] += sRed[(w8 * 16 + 4 * wave + u) * 64 + lane];
      const float gi[4] = {gi4.x, gi4.y, gi4.z, gi4.w}, gf[4] = {gf4.x, gf4.y, gf4.z, gf4.w};
      const float gg[4] = {gg4.x, gg4.y, gg4.z, gg4.w}, go[4] = {go4.x, go4.y, go4.z, go4.w};
      const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
      float dg4[4][4];        // [gate i,f,g,o][unit]
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float tc = vs_tanh_fast(cc[u]);
        dg4[3][u] = dh[u] * tc * go[u] * (1.f - go[u]);
        const float dc = fmaf(dh[u] * go[u], 1.f - tc * tc, dcc[u]);
        dg4[0][u] = dc * gg[u] * gi[u] * (1.f - gi[u]);
        dg4[1][u] = dc * cp[u] * gf[u] * (1.f - gf[u]);
        dg4[2][u] = dc * gi[u] * (1.f - gg[u] * gg[u]);
        dcc[u] = dc * gf[u];
      }
      if (!item) {
#pragma unroll
        for (int gate = 0; gate < 4; ++gate)
#pragma unroll
          for (int u = 0; u < 4; ++u) dg4[gate][u] = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) dcc[u] = 0.f;
      }
      // next step's operand: rows gate*H + ut*32 + 8*wave .. +7 = one half-chunk; this lane computed 4*half .. 4*half+3 of them
      if (units_ok) {
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
          bf16x8 v;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float o = __shfl_xor(dg4[gate][u], 32, 64);
            v[u] = (__bf16)(half ? o : dg4[gate][u]);
            v[4 + u] = (__bf16)(half ? dg4[gate][u] : o);
          }
          const int r0 = gate * a.H + ut * 32 + 8 * wave;
          const unsigned off = (unsigned)(((group * NCt + (r0 >> 4)) * 64 + ((r0 >> 3) & 1) * 32 + b31) * 16);
          if (half == 0)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), grs[(s + 1) & 1], off, 0, 16 /* sc1 */);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(gflags + ut * 4 + wave, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (item) {           // the batched GEMMs' operand, in place of the saved gates
        *reinterpret_cast<float4*>(grow) = make_float4(dg4[0][0], dg4[0][1], dg4[0][2], dg4[0][3]);
        *reinterpret_cast<float4*>(grow + a.H) = make_float4(dg4[1][0], dg4[1][1], dg4[1][2], dg4[1][3]);
        *reinterpret_cast<float4*>(grow + 2 * a.H) = make_float4(dg4[2][0], dg4[2][1], dg4[2][2], dg4[2][3]);
        *reinterpret_cast<float4*>(grow + 3 * a.H) = make_float4(dg4[3][0], dg4[3][1], dg4[3][2], dg4[3][3]);
      }
    }
  }
}

}  // namespace

// packed recurrent weights: the fp32 fragment form (every arithmetic: the step kernels use it), then room for the f16
// form of the math-selected persistent kernel (two planes), then a 64-float header (accumulator scale, weight scale, |max|)
static size_t lstm_packed_fp32_floats(int H) { return (size_t)2 * (H / 8) * (H / 8) * 256; }
static size_t lstm_packed_f16_floats(int H) { return (size_t)2 * (H / 8) * ((H + 15) / 16) * 2 * 256; }
extern "C" size_t vs_lstm_packed_floats(int H) { return lstm_packed_fp32_floats(H) + lstm_packed_f16_floats(H) + 64; }
// h ping, h pong, flags / c, + a fourth: four regions of 2 * Hp * Bpad floats (Hp = H rounded up to the f16 form's 16-wide K chunk; the
// tagged-data hand-off uses all four as exchange buffers) + 64: the persistent kernels' error word lives in the last 64 floats
// (never touched by the step kernels)
static size_t lstm_state_region(int B, int H) { return (size_t)2 * ((H + 15) / 16 * 16) * (((size_t)B + 31) / 32 * 32); }
extern "C" size_t vs_lstm_state_floats(int B, int H) { return 4 * lstm_state_region(B, H) + 64; }

// math: the dims.math of the call the weights are packed for (VS_MATH_CODE_*); it selects the f16 form written behind
// the fp32 one
int vs_lstm_pack_impl(const float* whh_f, const float* whh_b, float* wp, int H, hipStream_t stream, int math) {
  VS_REQUIRE(H > 0 && H % 8 == 0, "lstm: hidden size %d must be a multiple of 8", H);
  const long long total = (long long)lstm_packed_fp32_floats(H);
  hipLaunchKernelGGL(lstm_pack_whh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, whh_f, whh_b, wp, H);
  VS_LAUNCH_CHECK();
  if (math == VS_MATH_CODE_FP32) return 0;
  u32x4_t* w16 = reinterpret_cast<u32x4_t*>(wp + lstm_packed_fp32_floats(H));
  float* hdr = wp + lstm_packed_fp32_floats(H) + lstm_packed_f16_floats(H);
  const long long slots = 2LL * (H / 8) * ((H + 15) / 16) * 64;
  if (math == VS_MATH_CODE_BF16) {
    hipLaunchKernelGGL((lstm_pack16_kernel<1>), dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, whh_f, whh_b, w16, hdr, H);
  } else {
    // one power-of-two scale for both directions' W_hh: max|W_hh| * s in [2^9, 2^10)
    unsigned* amax = reinterpret_cast<unsigned*>(hdr + 8);
    VS_CHECK_HIP(hipMemsetAsync(amax, 0, sizeof(unsigned), stream));
    if (int rc = vs_absmax_accum_impl(whh_f, (long long)4 * H * H, amax, stream)) return rc;
    if (int rc = vs_absmax_accum_impl(whh_b, (long long)4 * H * H, amax, stream)) return rc;
    if (int rc = vs_scale_from_absmax_impl(amax, 1, hdr + 1, stream)) return rc;
    hipLaunchKernelGGL((lstm_pack16_kernel<2>), dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, whh_f, whh_b, w16, hdr, H);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

// which recurrence runs: 0 = persistent when its grid is resident (default), 1 = one launch per step (fp32 MFMA),
// 2 = persistent (error if it cannot be), 3 = persistent with the fp32 MFMA products whatever dims.math says (A/B of
// the f16 / bf16 products), 4 = persistent with the flag protocol of rounds 2-4 where the default is the tagged-data hand-off
// (lstm16_tagged_kernel: the f16 forward recurrence), 5 = the tagged-data hand-off in the bf16 BPTT too (correct, measured 5 % slower
// than its flag kernel: every wave polls 13 KB of gradients per round).  Test / A-B switch, process-global.
static int g_lstm_kernel = 0;
extern "C" int vs_set_lstm_kernel(int mode) {
  VS_REQUIRE(mode >= 0 && mode <= 4, "vs_set_lstm_kernel: mode %d", mode);
  g_lstm_kernel = mode;
  return 0;
}

namespace {
// A persistent launch whose spin gave up (a workgroup was not resident: the error word is 1) has produced garbage.
// The word is only read by callers that ask (vs_lstm_status), so the result itself is made unusable: NaN in the first
// 64 outputs -> NaN mask / NaN gradients -> the training loop's loss guard (train.py:112-114) and every isfinite check
// fire instead of training on wrong numbers.
__global__ void lstm_poison_kernel(const unsigned* __restrict__ err, float* __restrict__ out, int n) {
  if (*err == 0u) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __uint_as_float(0x7fc00000u);
}

// Cooperative launch: the runtime checks the grid against the kernel's occupancy and refuses (instead of queueing
// workgroups behind resident ones, where the flag protocol would spin until its bound) when it cannot be resident.
template <class Args>
hipError_t launch_resident(const void* kernel, dim3 grid, dim3 block, Args& a, hipStream_t stream) {
  void* params[] = {&a};
  return hipLaunchCooperativeKernel(kernel, grid, block, params, 0, stream);
}
}  // namespace

// state: 3 * [2][H][Bpad] floats.  Step kernels: h ping, h pong, c, zeroed here (zero initial state).
// Persistent kernel: h ping, h pong (fragment order), then the flag words + the error word.
int vs_bilstm_recurrent_impl(const float* xg, const float* wp, float* state, float* out, float* gates_save, float* c_save,
                             int B, int T, int H, hipStream_t stream, int math) {
  VS_REQUIRE(B > 0 && T > 0 && H > 0 && H % 8 == 0, "lstm: bad shape B=%d T=%d H=%d (H must be a multiple of 8)", B, T, H);
  const int Bpad = (B + 31) / 32 * 32;
  const size_t per = lstm_state_region(B, H);
  if (g_lstm_kernel == 3) math = VS_MATH_CODE_FP32;
  VS_CHECK_HIP(hipMemsetAsync(state, 0, vs_lstm_state_floats(B, H) * sizeof(float), stream));
  float* hbuf[2] = {state, state + per};
  const int HQ = H / 8, NBT = Bpad / 32;
  int dev = 0, cus = 0;
  VS_CHECK_HIP(hipGetDevice(&dev));
  VS_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // one workgroup per CU at most: every workgroup of a launch must be resident for the flag protocol
  const int bt_per_launch = cus / (2 * HQ);
  const bool persistent = g_lstm_kernel != 1 && bt_per_launch >= 1;
  VS_REQUIRE(g_lstm_kernel != 2 || persistent, "lstm: persistent recurrence needs 2*H/8 = %d workgroups <= %d CUs", 2 * HQ, cus);
  // the tagged-data hand-off of the f16 forward recurrence: three exchange buffers (the regions of h ping, h pong and of the flags),
  // the second and third armed with the sentinel
  const bool tagged = persistent && math != VS_MATH_CODE_FP32 && g_lstm_kernel != 4 && (H + 15) / 16 <= 4 * kMaxC;
  if (tagged) VS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(state + per), (int)kSentinel, 3 * per, stream));
  if (persistent) {
    unsigned* flags = reinterpret_cast<unsigned*>(state + 2 * per);      // zeroed above
    unsigned* err = reinterpret_cast<unsigned*>(state + 4 * per);         // first of the 64 trailing words
    bool launched = true;
    for (int bt0 = 0; bt0 < NBT; bt0 += bt_per_launch) {
      const int nbt = NBT - bt0 < bt_per_launch ? NBT - bt0 : bt_per_launch;
      hipError_t e;
      if (math == VS_MATH_CODE_FP32) {
        LstmPersistArgs a{xg, wp, hbuf[0], hbuf[1], flags, err, out, gates_save, c_save, B, T, H, Bpad, bt0};
        e = launch_resident(reinterpret_cast<const void*>(&lstm_persistent_kernel), dim3(HQ * nbt, 2), dim3(256), a, stream);
      } else {
        // the f16 form the weights were packed in (vs_lstm_pack_impl with the same math) and its header
        const float* w16 = wp + lstm_packed_fp32_floats(H);
        Lstm16Args a{xg, reinterpret_cast<const u32x4_t*>(w16), w16 + lstm_packed_f16_floats(H), hbuf[0], hbuf[1], flags, err, out,
                     gates_save, c_save, B, T, H, Bpad, bt0};
        if (tagged) {
          void* hb2 = state + 2 * per;
          void* hb3 = state + 3 * per;
          void* params[] = {&a, &hb2, &hb3};
          e = hipLaunchCooperativeKernel(math == VS_MATH_CODE_BF16 ? reinterpret_cast<const void*>(&lstm16_tagged_kernel<1>)
                                                                   : reinterpret_cast<const void*>(&lstm16_tagged_kernel<2>),
                                         dim3(HQ * nbt, 2), dim3(256), params, 0, stream);
        } else
        e = math == VS_MATH_CODE_BF16
                ? launch_resident(reinterpret_cast<const void*>(&lstm16_persistent_kernel<1>), dim3(HQ * nbt, 2), dim3(256), a, stream)
                : launch_resident(reinterpret_cast<const void*>(&lstm16_persistent_kernel<2>), dim3(HQ * nbt, 2), dim3(256), a, stream);
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();
        // refused before anything ran (first launch): the step kernels below do the whole job; later: an error
        VS_REQUIRE(bt0 == 0 && g_lstm_kernel != 2, "lstm: persistent recurrence could not be launched resident: %s", hipGetErrorString(e));
        launched = false;
        break;
      }
    }
    if (launched) {
      hipLaunchKernelGGL(lstm_poison_kernel, dim3(1), dim3(64), 0, stream, err, out, 64 < B * T * 2 * H ? 64 : B * T * 2 * H);
      VS_LAUNCH_CHECK();
      return 0;
    }
  }
  if (tagged) VS_CHECK_HIP(hipMemsetAsync(state, 0, vs_lstm_state_floats(B, H) * sizeof(float), stream));      // refused: un-arm the buffers
  float* c = state + 2 * per;
  dim3 grid(HQ * NBT, 2), block(256);
  for (int s = 0; s < T; ++s) {
    LstmStepArgs a{xg, wp, hbuf[s & 1], hbuf[(s + 1) & 1], c, out, gates_save, c_save, B, T, H, Bpad, s};
    hipLaunchKernelGGL(lstm_step_kernel, grid, block, 0, stream, a);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

// W_hh^T in fp32 fragment form, then room for the bf16 form (VS_MATH_BF16's BPTT), then 64 spare floats
static size_t lstm_packed_t_fp32_floats(int H) { return (size_t)2 * ((H + 31) / 32) * (H / 2) * 256; }
static size_t lstm_packed_t_bf16_floats(int H) { return (size_t)2 * ((H + 31) / 32) * (H / 4) * 256; }
extern "C" size_t vs_lstm_packed_t_floats(int H) { return lstm_packed_t_fp32_floats(H) + lstm_packed_t_bf16_floats(H) + 64; }
// backward state: dgates fragments ping/pong [2][2][NBT][H/2][256] + dc carry [2][Bpad][H]
extern "C" size_t vs_lstm_bwd_state_floats(int B, int H) {
  const size_t Bpad = ((size_t)B + 31) / 32 * 32;
  return 2 * (2 * (Bpad / 32) * (size_t)(H / 2) * 256) + 2 * Bpad * H + 64;   // + 64: error word of the persistent kernel
}

int vs_lstm_pack_t_impl(const float* whh_f, const float* whh_b, float* wp, int H, hipStream_t stream, int math) {
  VS_REQUIRE(H > 0 && H % 8 == 0, "lstm: hidden size %d must be a multiple of 8", H);
  const long long total = (long long)lstm_packed_t_fp32_floats(H);
  hipLaunchKernelGGL(lstm_pack_whh_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, whh_f, whh_b, wp, H);
  VS_LAUNCH_CHECK();
  if (math != VS_MATH_CODE_BF16) return 0;
  const long long slots = 2LL * ((H + 31) / 32) * (H / 4) * 64;
  hipLaunchKernelGGL(lstm_pack16_t_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, whh_f, whh_b,
                     reinterpret_cast<u32x4_t*>(wp + lstm_packed_t_fp32_floats(H)), H);
  VS_LAUNCH_CHECK();
  return 0;
}

// gates: activated gates from the training forward, overwritten with d(loss)/d(gate pre-activations).
// state: step kernels = dgates fragments ping/pong + dc carry; persistent kernel = the two fragment
// buffers, then (in the dc region) the flag words and the error word.
int vs_bilstm_bwd_recurrent_impl(const float* wpt, float* state, float* gates, const float* c_all, const float* dout,
                                 int B, int T, int H, hipStream_t stream, int math) {
  VS_REQUIRE(B > 0 && T > 0 && H > 0 && H % 8 == 0, "lstm_bwd: bad shape B=%d T=%d H=%d (H must be a multiple of 8)", B, T, H);
  if (g_lstm_kernel == 3) math = VS_MATH_CODE_FP32;
  const int Bpad = (B + 31) / 32 * 32;
  const size_t frag = (size_t)2 * (Bpad / 32) * (H / 2) * 256;
  VS_CHECK_HIP(hipMemsetAsync(state, 0, vs_lstm_bwd_state_floats(B, H) * sizeof(float), stream));
  float* gbuf[2] = {state, state + frag};
  const int NUT = (H + 31) / 32, NBT = Bpad / 32;
  int dev = 0, cus = 0;
  VS_CHECK_HIP(hipGetDevice(&dev));
  VS_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int bt_per_launch = cus / (2 * NUT);
  const bool persistent = g_lstm_kernel != 1 && bt_per_launch >= 1;
  VS_REQUIRE(g_lstm_kernel != 2 || persistent, "lstm_bwd: persistent recurrence needs %d workgroups <= %d CUs", 2 * NUT, cus);
  // (the tagged-data hand-off of the forward recurrence was built for this kernel too in round 5: correct and 5 % slower -- eight waves
  // poll 13 KB each per round -- so the BPTT keeps its flags: tools/attic/lstm16_bwd_tagged_kernel.hip.txt)
  if (persistent) {
    unsigned* flags = reinterpret_cast<unsigned*>(state + 2 * frag);      // 2*Bpad*H words available, 2*NBT*NUT*4 used
    unsigned* err = reinterpret_cast<unsigned*>(state + 2 * frag + (size_t)2 * Bpad * H);
    bool launched = true;
    for (int bt0 = 0; bt0 < NBT; bt0 += bt_per_launch) {
      const int nbt = NBT - bt0 < bt_per_launch ? NBT - bt0 : bt_per_launch;
      hipError_t e;
      if (math == VS_MATH_CODE_BF16) {      // gate gradients and W_hh^T as bf16 (the fragment buffers are half as large)
        Lstm16BwdArgs a{reinterpret_cast<const u32x4_t*>(wpt + lstm_packed_t_fp32_floats(H)), gbuf[0], gbuf[1], flags, err, gates, c_all, dout,
                        B, T, H, Bpad, bt0};
        e = launch_resident(reinterpret_cast<const void*>(&lstm16_bwd_persistent_kernel), dim3(NUT * nbt, 2), dim3(512), a, stream);
      } else {
        LstmBwdPersistArgs a{wpt, gbuf[0], gbuf[1], flags, err, gates, c_all, dout, B, T, H, Bpad, bt0};
        e = launch_resident(reinterpret_cast<const void*>(&lstm_bwd_persistent_kernel), dim3(NUT * nbt, 2), dim3(512), a, stream);
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();
        VS_REQUIRE(bt0 == 0 && g_lstm_kernel != 2, "lstm_bwd: persistent recurrence could not be launched resident: %s", hipGetErrorString(e));
        launched = false;
        break;
      }
    }
    if (launched) {
      hipLaunchKernelGGL(lstm_poison_kernel, dim3(1), dim3(64), 0, stream, err, gates, 64 < B * T * 8 * H ? 64 : B * T * 8 * H);
      VS_LAUNCH_CHECK();
      return 0;
    }
  }
  if (persistent) VS_CHECK_HIP(hipMemsetAsync(state, 0, vs_lstm_bwd_state_floats(B, H) * sizeof(float), stream));
  float* dc = state + 2 * frag;
  dim3 grid(NUT * NBT, 2), block(512);
  for (int s = 0; s < T; ++s) {
    LstmBwdArgs a{wpt, gbuf[s & 1], gbuf[(s + 1) & 1], gates, c_all, dout, dc, B, T, H, Bpad, s};
    hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, block, 0, stream, a);
  }
  VS_LAUNCH_CHECK();
  return 0;
}
