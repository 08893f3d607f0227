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
#include "vs_common.h"

namespace {

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
    const float gi = vs_sigmoid(acc[0 + u] + xgv[0 + u]);
    const float gf = vs_sigmoid(acc[4 + u] + xgv[4 + u]);
    const float gg = vs_tanh(acc[8 + u] + xgv[8 + u]);
    const float go = vs_sigmoid(acc[12 + u] + xgv[12 + u]);
    const float cn = gf * cprev[u] + gi * gg;
    hv[u] = go * vs_tanh(cn);
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
    const float tc = vs_tanh(cc[u]);
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

}  // namespace

extern "C" size_t vs_lstm_packed_floats(int H) { return (size_t)2 * (H / 8) * (H / 8) * 256; }
extern "C" size_t vs_lstm_state_floats(int B, int H) { return (size_t)3 * 2 * H * (((size_t)B + 31) / 32 * 32); }

int vs_lstm_pack_impl(const float* whh_f, const float* whh_b, float* wp, int H, hipStream_t stream) {
  VS_REQUIRE(H > 0 && H % 8 == 0, "lstm: hidden size %d must be a multiple of 8", H);
  const long long total = (long long)vs_lstm_packed_floats(H);
  hipLaunchKernelGGL(lstm_pack_whh_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, whh_f, whh_b, wp, H);
  VS_LAUNCH_CHECK();
  return 0;
}

// state: 3 * [2][H][Bpad] floats (h ping, h pong, c), zeroed here (zero initial state).
int vs_bilstm_recurrent_impl(const float* xg, const float* wp, float* state, float* out, float* gates_save, float* c_save,
                             int B, int T, int H, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && H > 0 && H % 8 == 0, "lstm: bad shape B=%d T=%d H=%d (H must be a multiple of 8)", B, T, H);
  const int Bpad = (B + 31) / 32 * 32;
  const size_t per = (size_t)2 * H * Bpad;
  VS_CHECK_HIP(hipMemsetAsync(state, 0, 3 * per * sizeof(float), stream));
  float* hbuf[2] = {state, state + per};
  float* c = state + 2 * per;
  dim3 grid((H / 8) * (Bpad / 32), 2), block(256);
  for (int s = 0; s < T; ++s) {
    LstmStepArgs a{xg, wp, hbuf[s & 1], hbuf[(s + 1) & 1], c, out, gates_save, c_save, B, T, H, Bpad, s};
    hipLaunchKernelGGL(lstm_step_kernel, grid, block, 0, stream, a);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t vs_lstm_packed_t_floats(int H) { return (size_t)2 * ((H + 31) / 32) * (H / 2) * 256; }
// backward state: dgates fragments ping/pong [2][2][NBT][H/2][256] + dc carry [2][Bpad][H]
extern "C" size_t vs_lstm_bwd_state_floats(int B, int H) {
  const size_t Bpad = ((size_t)B + 31) / 32 * 32;
  return 2 * (2 * (Bpad / 32) * (size_t)(H / 2) * 256) + 2 * Bpad * H;
}

int vs_lstm_pack_t_impl(const float* whh_f, const float* whh_b, float* wp, int H, hipStream_t stream) {
  VS_REQUIRE(H > 0 && H % 8 == 0, "lstm: hidden size %d must be a multiple of 8", H);
  const long long total = (long long)vs_lstm_packed_t_floats(H);
  hipLaunchKernelGGL(lstm_pack_whh_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, whh_f, whh_b, wp, H);
  VS_LAUNCH_CHECK();
  return 0;
}

// gates: activated gates from the training forward, overwritten with d(loss)/d(gate pre-activations).
int vs_bilstm_bwd_recurrent_impl(const float* wpt, float* state, float* gates, const float* c_all, const float* dout,
                                 int B, int T, int H, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && H > 0 && H % 8 == 0, "lstm_bwd: bad shape B=%d T=%d H=%d (H must be a multiple of 8)", B, T, H);
  const int Bpad = (B + 31) / 32 * 32;
  const size_t frag = (size_t)2 * (Bpad / 32) * (H / 2) * 256;
  VS_CHECK_HIP(hipMemsetAsync(state, 0, vs_lstm_bwd_state_floats(B, H) * sizeof(float), stream));
  float* gbuf[2] = {state, state + frag};
  float* dc = state + 2 * frag;
  dim3 grid(((H + 31) / 32) * (Bpad / 32), 2), block(512);
  for (int s = 0; s < T; ++s) {
    LstmBwdArgs a{wpt, gbuf[s & 1], gbuf[(s + 1) & 1], gates, c_all, dout, dc, B, T, H, Bpad, s};
    hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, block, 0, stream, a);
  }
  VS_LAUNCH_CHECK();
  return 0;
}
