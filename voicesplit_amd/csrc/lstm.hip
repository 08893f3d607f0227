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
// combined through LDS.  W_hh is pre-packed in fragment order (one coalesced dwordx4 per lane
// per 4 K-steps); h and c are kept transposed [dir][H][Bpad] so the B-operand reads and the
// state updates are coalesced along batch.
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
  const float* h_prev;  // [2][H][Bpad]
  float* h_next;        // [2][H][Bpad]
  float* c;             // [2][H][Bpad]  (updated in place: each (unit,batch) has one owner)
  float* out;           // [B][T][2H]
  int B, T, H, Bpad, step;
};

__global__ __launch_bounds__(256)
void lstm_step_kernel(LstmStepArgs a) {
  __shared__ float sRed[3 * 16 * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int HQ = a.H / 8;
  const int jg = blockIdx.x % HQ;
  const int bt = blockIdx.x / HQ;
  const int dir = blockIdx.y;
  const int t = dir ? (a.T - 1 - a.step) : a.step;
  const int b = bt * 32 + l31;
  const size_t hb = (size_t)dir * a.H * a.Bpad;

  // wave 0 owns the epilogue: start its xg / c reads before the reduction loop
  float xgv[16], cprev[4];
  if (wave == 0) {
    const bool ok = b < a.B;
    const float* xrow = a.xg + ((size_t)(ok ? b : 0) * a.T + t) * (8 * a.H) + (size_t)dir * 4 * a.H + jg * 8 + 4 * half;
#pragma unroll
    for (int r = 0; r < 16; ++r) xgv[r] = ok ? xrow[(r >> 2) * a.H + (r & 3)] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) cprev[u] = a.c[hb + (size_t)(jg * 8 + 4 * half + u) * a.Bpad + b];
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (a.step > 0) {   // h_{-1} = 0: nothing to multiply at the first step
    const float4* wq = reinterpret_cast<const float4*>(a.wp) + ((size_t)(dir * HQ + jg) * HQ) * 64 + lane;
    const float* hp = a.h_prev + hb + (size_t)half * a.Bpad + b;
    for (int q = wave; q < HQ; q += 4) {
      const float4 w4 = wq[(size_t)q * 64];
      const float* hq = hp + (size_t)(8 * q) * a.Bpad;
      const float h0 = hq[0];
      const float h1 = hq[2 * (size_t)a.Bpad];
      const float h2 = hq[4 * (size_t)a.Bpad];
      const float h3 = hq[6 * (size_t)a.Bpad];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, h0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, h1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, h2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, h3, acc, 0, 0, 0);
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

  float hv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float gi = vs_sigmoid(acc[0 + u] + xgv[0 + u]);
    const float gf = vs_sigmoid(acc[4 + u] + xgv[4 + u]);
    const float gg = vs_tanh(acc[8 + u] + xgv[8 + u]);
    const float go = vs_sigmoid(acc[12 + u] + xgv[12 + u]);
    const float cn = gf * cprev[u] + gi * gg;
    hv[u] = go * vs_tanh(cn);
    const size_t si = hb + (size_t)(jg * 8 + 4 * half + u) * a.Bpad + b;
    a.c[si] = cn;           // padded batch columns only ever hold garbage they produced themselves
    a.h_next[si] = hv[u];
  }
  if (b < a.B) {
    float4* o = reinterpret_cast<float4*>(a.out + ((size_t)b * a.T + t) * (2 * a.H) + (size_t)dir * a.H + jg * 8 + 4 * half);
    *o = make_float4(hv[0], hv[1], hv[2], hv[3]);
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
int vs_bilstm_recurrent_impl(const float* xg, const float* wp, float* state, float* out,
                             int B, int T, int H, hipStream_t stream) {
  VS_REQUIRE(B > 0 && T > 0 && H > 0 && H % 8 == 0, "lstm: bad shape B=%d T=%d H=%d (H must be a multiple of 8)", B, T, H);
  const int Bpad = (B + 31) / 32 * 32;
  const size_t per = (size_t)2 * H * Bpad;
  VS_CHECK_HIP(hipMemsetAsync(state, 0, 3 * per * sizeof(float), stream));
  float* hbuf[2] = {state, state + per};
  float* c = state + 2 * per;
  dim3 grid((H / 8) * (Bpad / 32), 2), block(256);
  for (int s = 0; s < T; ++s) {
    LstmStepArgs a{xg, wp, hbuf[s & 1], hbuf[(s + 1) & 1], c, out, B, T, H, Bpad, s};
    hipLaunchKernelGGL(lstm_step_kernel, grid, block, 0, stream, a);
  }
  VS_LAUNCH_CHECK();
  return 0;
}
