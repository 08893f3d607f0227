// Training orchestration: vs_forward_train (forward that keeps the tape) and vs_backward, plus
// the extern "C" unit-test surface of the backward kernels.  Which kernel runs on which buffer,
// in which order -- no arithmetic lives here.
//
// Reference graph being differentiated: models/voicesplit/model.py:66-89 (forward) as driven by
// train.py:94-110 (mask -> loss -> loss.backward()).
#include <mutex>
#include <stdlib.h>
#include <string.h>

#include "../../include/voicesplit_hip.h"
#include "vs_internal.h"

int vs_check_dims_impl(const vs_dims* d);   // capi.hip

namespace {

constexpr float kBnEps = 1e-5f;
constexpr float kBnMomentum = 0.1f;
struct Spec { int kt, kf, dil; };
constexpr Spec kMid[6] = {{7, 1, 1}, {5, 5, 1}, {5, 5, 2}, {5, 5, 4}, {5, 5, 8}, {5, 5, 16}};
constexpr int kSplitK = 16;    // split-K factor of the small weight-gradient GEMMs (fc1, fc2, W_hh)

inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }
template <typename T>
inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }
inline size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

int tape_layout(const vs_dims* d, vs_tape_layout* L) {
  if (int rc = vs_check_dims_impl(d)) return rc;
  memset(L, 0, sizeof(*L));
  const size_t B = d->B, T = d->T, F = d->F, H = d->H, M = B * T;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  // conv activations: fp32 [B][64][T][F], or channels-last bf16 [B][T][F][64] in the bf16 configuration (half the bytes)
  const size_t act = B * 64 * T * F * (d->math == VS_MATH_BF16 ? 2 : 4);
  // (bf16 configuration: cnn1 is recomputed from x in the backward pass -- no z[0] tensor, round 4)
  for (int l = 0; l < 7; ++l) { L->z[l] = take(d->math == VS_MATH_BF16 && l == 0 ? 256 : act); L->a[l] = take(act); }
  L->z8 = take(M * 8 * F * 4);
  L->feat = take(M * 8 * F * 4);
  L->bn_scale = take(8 * 64 * 4);
  L->bn_shift = take(8 * 64 * 4);
  L->bn_mean = take(8 * 64 * 4);
  L->bn_invstd = take(8 * 64 * 4);
  L->gates = take(M * 8 * H * 4);
  L->cstate = take(M * 2 * H * 4);
  L->lstm_out = take(M * 2 * H * 4);
  L->fc1_out = take(M * (size_t)d->FC1 * 4);
  L->dlogits = take(M * (size_t)d->FC2 * 4);
  L->dfc1 = take(M * (size_t)d->FC1 * 4);
  L->dlstm_out = take(M * 2 * H * 4);
  L->dsum = take(B * 8 * H * 4);
  L->dfeat = take(M * 8 * F * 4);
  L->grad0 = take(act);
  L->grad1 = take(act);
  L->dvbias = take(B * 8 * H * 4);
  for (int i = 0; i < 6; ++i) L->conv_packed[i] = take(vs_conv64_packed_floats(kMid[i].kt, kMid[i].kf) * 4);
  L->pack_tmp = take(vs_conv64_packed_floats(5, 5) * 4);
  L->lstm_packed = take(vs_lstm_packed_floats(d->H) * 4);
  L->lstm_packed_t = take(vs_lstm_packed_t_floats(d->H) * 4);
  L->lstm_state = take(vs_lstm_state_floats(d->B, d->H) * 4);
  L->lstm_bwd_state = take(vs_lstm_bwd_state_floats(d->B, d->H) * 4);
  L->consts = take(128 * 4);
  L->bn_stats = take((size_t)VS_BN_STAT_SLOTS * 64 * 2 * 8);
  L->bn_coef = take(3 * 64 * 4);
  L->first_acc = take((64 + VS_FIRST_BWD_SCRATCH_DOUBLES) * 8);      // [cnn1's input moments (forward -> backward)][backward scratch]
  L->colsum_tmp = take(2 * B * max3(8 * H, d->FC1, d->FC2) * 4);      // two halves: the caller's stream, the side stream's leaves
  // one scratch region, reused by the stream-ordered consumers: conv wgrad partial sums,
  // cnn8 wgrad partials, split-K partials of the fc / W_hh weight gradients
  size_t part = vs_conv64_wgrad_partial_floats(5, 5);
  part = max3(part, vs_conv64_wgrad_partial_floats(7, 1), (size_t)vs_conv_last_wgrad_blocks() * 512);
  part = max3(part, (size_t)kSplitK * d->FC2 * d->FC1, (size_t)kSplitK * d->FC1 * 2 * H);
  part = max3(part, (size_t)kSplitK * 4 * H * H, (size_t)0);
  L->partials = take(part * 4);
  L->conv_scales = take(16 * VS_SCALE_SLOT_FLOATS * 4);
  L->gemm_scales = take(32 * 4);
  // bf16 configuration: feat / W_ih / dxg as bf16 arrays shared by the forward GEMM and the two backward contractions
  L->lstm_bf16 = take(d->math == VS_MATH_BF16 ? vs_lstm_bf16_layout((long long)M, 8 * (int)F, (int)H).total : 256);
  // (behind the turn words: the per-slot sums of cnn1's backward in deterministic mode, [VS_BN_STAT_SLOTS][576] doubles)
  L->det_turn = take(VS_TURN_WORDS * 4 + (size_t)VS_BN_STAT_SLOTS * 576 * 8);
  for (int i = 0; i < 6; ++i) L->conv_packed_t[i] = take(vs_conv64_packed_floats(kMid[i].kt, kMid[i].kf) * 4);
  L->total_bytes = off;
  return 0;
}

int check_tape(const vs_dims* d, void* tape, size_t bytes, vs_tape_layout* L) {
  if (int rc = tape_layout(d, L)) return rc;
  VS_REQUIRE(tape != nullptr, "tape is NULL");
  VS_REQUIRE((reinterpret_cast<uintptr_t>(tape) & 255) == 0, "tape must be 256-byte aligned");
  VS_REQUIRE(bytes >= L->total_bytes, "tape too small: %zu < %zu bytes", bytes, L->total_bytes);
  return 0;
}

int check_params(const vs_params* p) {
  VS_REQUIRE(p != nullptr, "params is NULL");
  for (int l = 0; l < 8; ++l) {
    const vs_conv_layer& c = p->conv[l];
    VS_REQUIRE(c.weight && c.bias && c.bn_weight && c.bn_bias && c.bn_running_mean && c.bn_running_var,
               "conv layer %d has a NULL parameter", l + 1);
  }
  for (int dir = 0; dir < 2; ++dir)
    VS_REQUIRE(p->w_ih[dir] && p->w_hh[dir] && p->b_ih[dir] && p->b_hh[dir], "NULL LSTM parameter (dir %d)", dir);
  VS_REQUIRE(p->fc1_w && p->fc1_b && p->fc2_w && p->fc2_b, "NULL head parameter");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Weight gradients on a side stream.  Layer l's weight gradient needs dz_l and the tape's a_{l-1}; nothing on the
// data-gradient chain (BatchNorm backward of layer l-1 -> data gradient of layer l-1 -> ...) needs ITS result.  It
// is a matrix-pipe kernel with one workgroup per CU and an idle memory system; the two BatchNorm backward passes
// of the next layer are pure HBM streams with idle matrix pipes (2.65 ms per layer, 16 ms per step).  So the weight
// gradient of layer l is launched on a second stream right after the data gradient of layer l and runs beside the
// BatchNorm backward of layer l-1; the data gradient of layer l-1 -- which overwrites dz_l -- waits for it.
// Same kernels, same order of every sum: results are bit-identical to the one-stream schedule.
// One side stream + event pair per device, created on first use; the caller's stream is joined before vs_backward
// returns, so the fork is invisible outside (and legal under stream capture).
// ---------------------------------------------------------------------------------------------
int g_bwd_overlap = 1;
// One launch in front of the bf16 forward pass: ones[64] = 1, zeros[64] = 0, up to three scratch arrays cleared (16-byte granules).
struct ArmArgs { float* ones; void* z[3]; unsigned n16[3]; };
__global__ __launch_bounds__(256)
void forward_arm_kernel(ArmArgs a) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i < 64) { a.ones[i] = 1.f; a.ones[64 + i] = 0.f; }
#pragma unroll
  for (int r = 0; r < 3; ++r)
    if (i < a.n16[r]) reinterpret_cast<uint4*>(a.z[r])[i] = make_uint4(0u, 0u, 0u, 0u);
}

struct SideStream { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr, leaves = nullptr; };
SideStream g_side[16];
// Every exit path of vs_backward after the fork -- the error returns included -- orders the caller's stream after what
// the side stream has been given (it writes gradients and the shared partial-sum scratch), and leaves no unjoined fork
// behind in a stream capture.  The normal path joins explicitly and disarms the guard.
struct SideJoin {
  SideStream* side = nullptr;
  hipStream_t stream = nullptr;
  bool forked = false;
  ~SideJoin() {
    if (!side || !forked) return;
    if (hipEventRecord(side->join, side->s) == hipSuccess) (void)hipStreamWaitEvent(stream, side->join, 0);
  }
};
// the side stream and its two events are shared by every caller on the device: one vs_backward enqueues at a time
// (host threads driving different caller streams would otherwise re-record an event another call is about to wait on)
std::mutex g_side_mutex;
// (not std::unique_lock: its lock() / unlock() are out-of-line template members that libstdc++ marks default-visible, i.e. they would
// become exports of the library -- tests/test_abi_cpu.py compares the dynamic symbol table with the header)
struct SideLock {
  bool held = false;
  void lock() { g_side_mutex.lock(); held = true; }
  void unlock() { if (held) { g_side_mutex.unlock(); held = false; } }
  ~SideLock() { unlock(); }
};
int side_stream(SideStream** out) {
  int dev = 0;
  VS_CHECK_HIP(hipGetDevice(&dev));
  VS_REQUIRE(dev >= 0 && dev < 16, "side stream: device index out of range");
  SideStream& ss = g_side[dev];
  if (!ss.s) {
    // [r6] LOW priority -- not for the priority but for the hardware queue.  HIP maps the streams of one priority onto a pool of (by
    // default four) hardware queues; a process that has initialised RCCL holds more normal-priority streams than that, and this
    // stream, created later, then SHARES a queue with the caller's stream: the two no longer run concurrently and every kernel of the
    // backward pass runs alone (46.2 -> 49.4-49.7 ms per step with the process group merely initialised: bench.py --force-collectives,
    // profiles/r06_experiments.md section 5).  A stream of another priority draws from another pool.  At N = 1 without RCCL the three
    // priorities measure the same within 0.1 ms (46.3 / 46.4 / 46.4 ms normal / high / low, two rounds in one call); what runs here --
    // weight gradients, leaves, weight packs -- is off the critical path by construction.
    int lo = 0, hi = 0;                                  // (numerically: hi <= lo, lo = the least priority)
    VS_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    VS_CHECK_HIP(hipStreamCreateWithPriority(&ss.s, hipStreamNonBlocking, lo));
    VS_CHECK_HIP(hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming));
    VS_CHECK_HIP(hipEventCreateWithFlags(&ss.join, hipEventDisableTiming));
    VS_CHECK_HIP(hipEventCreateWithFlags(&ss.leaves, hipEventDisableTiming));
  }
  *out = &ss;
  return 0;
}

}  // namespace

extern "C" {

int vs_tape_layout_query(const vs_dims* dims, vs_tape_layout* out) {
  VS_REQUIRE(out != nullptr, "tape layout out pointer is NULL");
  return tape_layout(dims, out);
}

size_t vs_tape_bytes(const vs_dims* dims) {
  vs_tape_layout L;
  if (tape_layout(dims, &L)) return 0;
  return L.total_bytes;
}

// ---------------------------------------------------------------------------------------------
// forward with tape
// ---------------------------------------------------------------------------------------------
int vs_forward_train(const vs_dims* d, const vs_params* p, const float* x, const float* dvec, int conv_act, int bn_mode,
                     void* tape, size_t tape_bytes, float* mask, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  vs_tape_layout L;
  if (int rc = check_tape(d, tape, tape_bytes, &L)) return rc;
  if (int rc = check_params(p)) return rc;
  VS_REQUIRE(x && dvec && mask, "forward_train: NULL argument");
  VS_REQUIRE(conv_act == VS_ACT_MISH || conv_act == VS_ACT_RELU, "forward_train: conv_act must be MISH or RELU");
  VS_REQUIRE(bn_mode == VS_BN_EVAL || bn_mode == VS_BN_TRAIN, "forward_train: unknown bn_mode %d", bn_mode);
  const int B = d->B, T = d->T, F = d->F, H = d->H;
  const bool train = bn_mode == VS_BN_TRAIN;
  float* ones = at<float>(tape, L.consts);
  float* scale = at<float>(tape, L.bn_scale);
  float* shift = at<float>(tape, L.bn_shift);
  float* mean = at<float>(tape, L.bn_mean);
  float* invstd = at<float>(tape, L.bn_invstd);
  double* stats = at<double>(tape, L.bn_stats);
  // deterministic mode (bf16 configuration): the turn words of this tape, armed here; every launch that takes turns re-arms its own
  const bool det = d->math == VS_MATH_BF16 && vs_opt(VS_OPT_DETERMINISTIC) != 0;
  if (d->math == VS_MATH_BF16) {
    // [r6] the constants and every scratch array the pass wants cleared, in ONE launch (they were six runtime fill dispatches, 54 us in
    // front of the first kernel of the step: tools/dispatch_census.py)
    ArmArgs arm{ones, {at<void>(tape, L.bn_stats), at<void>(tape, L.first_acc), det ? at<void>(tape, L.det_turn) : nullptr},
                {VS_BN_STAT_SLOTS * 128 * 8 / 16, 64 * 8 / 16, (unsigned)(det ? VS_TURN_WORDS * 4 / 16 : 0)}};
    unsigned most = 8;
    for (unsigned n : arm.n16) most = n > most ? n : most;
    hipLaunchKernelGGL(forward_arm_kernel, dim3((most + 255) / 256), dim3(256), 0, stream, arm);
  } else {
  VS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ones), 0x3f800000 /* 1.0f */, 64, stream));
  VS_CHECK_HIP(hipMemsetAsync(ones + 64, 0, 64 * sizeof(float), stream));
  }
  VsTurnScope turn_scope(det ? at<unsigned>(tape, L.det_turn) : nullptr);

  // split-f16 convs: the BatchNorm+activation pass that produces a layer's input also folds its
  // |max| into that layer's scale slot (slot l = conv index l: input scale of cnn(l+1))
  float* cs = at<float>(tape, L.conv_scales);
  const bool f16 = d->math != VS_MATH_FP32;
  if (f16 && d->math != VS_MATH_BF16) VS_CHECK_HIP(hipMemsetAsync(cs, 0, 16 * VS_SCALE_SLOT_FLOATS * sizeof(float), stream));      // (bf16: no operand scales)
  // conv + bias -> z (kept), then BatchNorm + activation -> a (kept)
  auto bn = [&](int l, const float* z, float* a, int C, bool feat_layout, int stats_slots = 0) -> int {
    VsProfScope ps(VS_PROF_FWD_BN, stream);
    const vs_conv_layer& c = p->conv[l];
    unsigned* amax = (f16 && l + 1 <= 6) ? vs_amax_slot(cs + VS_SCALE_SLOT_FLOATS * (l + 1)) : nullptr;
    float *sc = scale + 64 * l, *sh = shift + 64 * l, *mu = mean + 64 * l, *is = invstd + 64 * l;
    if (train) {
      return feat_layout
                 ? vs_bn_train_feat_impl(z, a, B, T, F, c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var, kBnEps,
                                         kBnMomentum, conv_act, stats, sc, sh, mu, is, stream)
                 : vs_bn_train_impl(z, a, B, C, T * F, c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var, kBnEps,
                                    kBnMomentum, conv_act, stats, sc, sh, mu, is, amax, stream, stats_slots);
    }
    if (int rc = vs_bn_eval_consts_impl(c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var, kBnEps, C, sc, sh, mu, is, stream)) return rc;
    return feat_layout ? vs_bn_apply_feat_impl(z, a, B, T, F, conv_act, sc, sh, stream)
                       : vs_bn_apply_impl(z, a, B, C, T * F, conv_act, sc, sh, amax, stream);
  };

  const bool nhwc = d->math == VS_MATH_BF16;
  // ---- weight-only work of the step on the side stream (bf16 configuration) -----------------------------------------------------
  // Packing the six conv weights, the bf16 copy of W_ih, the d-vector fold, the recurrent and head weight images depend on nothing the
  // conv stack produces: 14 launches of 5-45 us that used to sit, with their launch gaps, in front of their consumers on the one
  // stream (~0.25 ms of a step).  They now run beside cnn1; the caller's stream joins in front of cnn2.
  SideStream* side = nullptr;
  SideLock side_lock;
  const int K = 8 * F, KE = K + d->E;
  float* dvbias = at<float>(tape, L.dvbias);
  const VsLstmBf16Layout Lpre = vs_lstm_bf16_layout((long long)B * T, K, H);
  const bool head_fused = nhwc && vs_head_fused_supported(2 * H, d->FC1, d->FC2) &&
                          L.conv_scales - L.partials >= vs_head_fused_packed_bytes(2 * H, d->FC1, d->FC2);
  bool prologue = false;
  SideJoin pro_join;
  if (nhwc && g_bwd_overlap && vs_opt(VS_OPT_FWD_PROLOGUE)) {
    side_lock.lock();
    if (int rc = side_stream(&side)) return rc;
    pro_join.side = side;
    pro_join.stream = stream;
    VS_CHECK_HIP(hipEventRecord(side->fork, stream));
    VS_CHECK_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
    pro_join.forked = true;
    hipStream_t ps = side->s;
    for (int i = 0; i < 6; ++i)
      if (int rc = vs_nhwc_pack_impl(p->conv[i + 1].weight, at<void>(tape, L.conv_packed[i]), kMid[i].kt, kMid[i].kf, 0, ps)) return rc;
    // [r6] the backward pass's weight images too (same weights, idle side stream): they were 0.1 ms on the backward's critical path
    for (int i = 0; i < 6; ++i)
      if (int rc = vs_nhwc_pack_impl(p->conv[i + 1].weight, at<void>(tape, L.conv_packed_t[i]), kMid[i].kt, kMid[i].kf, 1, ps)) return rc;
    if (int rc = vs_lstm_pack_t_impl(p->w_hh[0], p->w_hh[1], at<float>(tape, L.lstm_packed_t), H, ps, d->math)) return rc;
    for (int dir = 0; dir < 2; ++dir) {
      if (int rc = vs_gemm_nt_impl(dvec, d->E, p->w_ih[dir] + K, KE, dvbias + (size_t)dir * 4 * H, 8 * H, B, 4 * H, d->E,
                                   p->b_ih[dir], p->b_hh[dir], nullptr, 0, 1, 0, VS_ACT_NONE, ps)) return rc;
      if (int rc = vs_cvt_rows_bf16_impl(p->w_ih[dir], 4 * H, K, KE, at<char>(tape, L.lstm_bf16) + Lpre.wih + (size_t)dir * 4 * H * Lpre.Kp * 2, Lpre.Kp, ps)) return rc;
    }
    if (int rc = vs_lstm_pack_impl(p->w_hh[0], p->w_hh[1], at<float>(tape, L.lstm_packed), H, ps, d->math)) return rc;
    if (head_fused) {
      if (int rc = vs_head_fused_pack_impl(p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, 2 * H, d->FC1, d->FC2, at<void>(tape, L.partials), ps)) return rc;
    }
    VS_CHECK_HIP(hipEventRecord(side->join, side->s));
    prologue = true;
  }
  if (nhwc) {
    // BASELINE configs[2]: channels-last bf16 z / a (conv_nhwc.hip, nhwc_edge.hip); statistics from the conv epilogues
    const long long npix = (long long)B * T * F;
    // the statistics scratch is cleared ONCE here; every finalize below folds the slots and clears the scratch behind itself in the
    // same launch (vs_fold_slots): no memset kernel in front of the conv launches
    const int kStatsDoubles = VS_BN_STAT_SLOTS * 128;
    // (cleared by forward_arm_kernel above)
    auto bn16 = [&](int l) -> int {
      VsProfScope ps(VS_PROF_FWD_BN, stream);
      const vs_conv_layer& c = p->conv[l];
      float *sc = scale + 64 * l, *sh = shift + 64 * l, *mu = mean + 64 * l, *is = invstd + 64 * l;
      if (train) {
        if (int rc = vs_bn_finalize_impl(stats, VS_BN_STAT_SLOTS, (double)npix, 64, c.bn_weight, c.bn_bias, c.bn_running_mean,
                                         c.bn_running_var, kBnEps, kBnMomentum, sc, sh, mu, is, stream, kStatsDoubles)) return rc;
      } else {
        if (int rc = vs_bn_eval_consts_impl(c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var, kBnEps, 64, sc, sh, mu, is, stream)) return rc;
      }
      return vs_nhwc_bn_apply_impl(at<void>(tape, L.z[l]), at<void>(tape, L.a[l]), npix, conv_act, sc, sh, stream);
    };
    {
      // cnn1 by recomputation: the batch statistics of z1 = conv(x) + bias from the 35 moments of the input's seven shifts (one
      // pass over the 46 MB input), then ONE pass that writes a1 = act(BN(z1)): no z1 tensor, no apply pass (nhwc_edge.hip)
      VsProfScope ps(VS_PROF_CNN1, stream);
      const vs_conv_layer& c = p->conv[0];
      // (the moments stay in the tape for the backward pass: first_acc = [35 moments, padded to 64][backward scratch])
      double* mom = at<double>(tape, L.first_acc);
      // (deterministic mode: per-slot sums in the backward pass's dfeat buffer, which nothing uses before the loss)
      if (int rc = vs_nhwc_first_moments_impl(x, B, T, F, mom, stream, det ? at<double>(tape, L.dfeat) : nullptr, /*mom_is_zero=*/1)) return rc;
      if (train) {
        if (int rc = vs_nhwc_first_stats_impl(mom, c.weight, c.bias, (double)npix, stats, stream)) return rc;
        if (int rc = vs_bn_finalize_impl(stats, 1, (double)npix, 64, c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var, kBnEps,
                                         kBnMomentum, scale, shift, mean, invstd, stream, kStatsDoubles)) return rc;
      } else {
        if (int rc = vs_bn_eval_consts_impl(c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var, kBnEps, 64, scale, shift, mean, invstd, stream)) return rc;
      }
      if (int rc = vs_nhwc_conv_first_impl(x, c.weight, scale, shift, at<void>(tape, L.a[0]), B, T, F, conv_act, nullptr, stream, c.bias)) return rc;
    }
    // cnn7's BatchNorm + activation is applied by its consumer: cnn8 is an HBM-bound kernel with idle VALU (forward: on the
    // way into its matrix pipe; backward: recomputed beside the derivative), so train mode has no apply pass over z7
    // and no a7 tensor.  Everything else of bn16(6) -- finalize, running statistics, the constants -- stays.
    auto bn16_consts = [&](int l) -> int {
      VsProfScope ps(VS_PROF_FWD_BN, stream);
      const vs_conv_layer& c = p->conv[l];
      return vs_bn_finalize_impl(stats, VS_BN_STAT_SLOTS, (double)npix, 64, c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var,
                                 kBnEps, kBnMomentum, scale + 64 * l, shift + 64 * l, mean + 64 * l, invstd + 64 * l, stream, kStatsDoubles);
    };
    if (prologue) {      // the weight images are needed from here on
      VS_CHECK_HIP(hipStreamWaitEvent(stream, side->join, 0));
      pro_join.forked = false;
      side_lock.unlock();
    }
    for (int i = 0; i < 6; ++i) {
      const int l = i + 1;
      void* packed = at<void>(tape, L.conv_packed[i]);
      {
        VsProfScope ps(VS_PROF_CNN2 + i, stream);
        if (!prologue) {
          if (int rc = vs_nhwc_pack_impl(p->conv[l].weight, packed, kMid[i].kt, kMid[i].kf, 0, stream)) return rc;
          if (int rc = vs_nhwc_pack_impl(p->conv[l].weight, at<void>(tape, L.conv_packed_t[i]), kMid[i].kt, kMid[i].kf, 1, stream)) return rc;
        }
        if (train && !kStatsDoubles) VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * VS_BN_STAT_SLOTS * 128, stream));
        if (int rc = vs_nhwc_conv_impl(at<void>(tape, L.a[l - 1]), packed, ones, p->conv[l].bias, at<void>(tape, L.z[l]), B, T, F,
                                       kMid[i].kt, kMid[i].kf, kMid[i].dil, VS_ACT_NONE, train ? stats : nullptr, stream)) return rc;
      }
      if (train && l == 6) { if (int rc = bn16_consts(l)) return rc; }
      else if (int rc = bn16(l)) return rc;
    }
    {
      VsProfScope ps(VS_PROF_CNN8, stream);
      if (train) {
        if (!kStatsDoubles) VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * VS_BN_STAT_SLOTS * 16, stream));
        if (int rc = vs_nhwc_conv_last_impl(at<void>(tape, L.z[6]), p->conv[7].weight, ones, p->conv[7].bias, at<float>(tape, L.z8), B, T, F, VS_ACT_NONE, stream,
                                            stats, scale + 64 * 6, shift + 64 * 6, conv_act)) return rc;
      } else if (int rc = vs_nhwc_conv_last_impl(at<void>(tape, L.a[6]), p->conv[7].weight, ones, p->conv[7].bias, at<float>(tape, L.z8), B, T, F, VS_ACT_NONE, stream)) return rc;
    }
  } else {
  {
    VsProfScope ps(VS_PROF_CNN1, stream);
    if (int rc = vs_conv_first_fwd_impl(x, p->conv[0].weight, ones, p->conv[0].bias, at<float>(tape, L.z[0]), B, T, F, VS_ACT_NONE, nullptr, stream)) return rc;
  }
  if (int rc = bn(0, at<float>(tape, L.z[0]), at<float>(tape, L.a[0]), 64, false)) return rc;
  for (int i = 0; i < 6; ++i) {
    const int l = i + 1;
    float* packed = at<float>(tape, L.conv_packed[i]);
    // batch statistics of z accumulated by the conv epilogue itself (split-f16 / bf16 kernels): one pass less over z
    const bool fuse = train && f16;
    {
      VsProfScope ps(VS_PROF_CNN2 + i, stream);
      if (fuse) VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * VS_BN_STAT_SLOTS * 128, stream));
      if (int rc = vs_conv64_layer_impl(d->math, at<float>(tape, L.a[l - 1]), p->conv[l].weight, packed,
                                        cs + VS_SCALE_SLOT_FLOATS * l, 1, ones, p->conv[l].bias, at<float>(tape, L.z[l]),
                                        B, T, F, kMid[i].kt, kMid[i].kf, kMid[i].dil, VS_ACT_NONE, 0, nullptr, stream,
                                        fuse ? stats : nullptr)) return rc;
    }
    if (int rc = bn(l, at<float>(tape, L.z[l]), at<float>(tape, L.a[l]), 64, false, fuse ? VS_BN_STAT_SLOTS : 0)) return rc;
  }
  {
    VsProfScope ps(VS_PROF_CNN8, stream);
    if (int rc = vs_conv_last_fwd_impl(at<float>(tape, L.a[6]), p->conv[7].weight, ones, p->conv[7].bias, at<float>(tape, L.z8), B, T, F, VS_ACT_NONE, stream)) return rc;
  }
  }
  bool feat_bf16_ready = false;
  if (nhwc && train) {
    // cnn8's batch statistics came out of its own epilogue: finalize + the apply pass
    VsProfScope ps(VS_PROF_FWD_BN, stream);
    const vs_conv_layer& c = p->conv[7];
    if (int rc = vs_bn_finalize_impl(stats, VS_BN_STAT_SLOTS, (double)B * T * F, 8, c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var,
                                     kBnEps, kBnMomentum, scale + 64 * 7, shift + 64 * 7, mean + 64 * 7, invstd + 64 * 7, stream,
                                     VS_BN_STAT_SLOTS * 128)) return rc;
    // ... which also writes the bf16 row-form copy of the features the LSTM GEMMs read
    const VsLstmBf16Layout Lf = vs_lstm_bf16_layout((long long)B * T, 8 * F, H);
    if (int rc = vs_bn_apply_feat_bf16_impl(at<float>(tape, L.z8), at<float>(tape, L.feat), at<char>(tape, L.lstm_bf16) + Lf.feat, Lf.Kp, B, T, F, conv_act,
                                            scale + 64 * 7, shift + 64 * 7, stream)) return rc;
    feat_bf16_ready = true;
  } else if (int rc = bn(7, at<float>(tape, L.z8), at<float>(tape, L.feat), 8, true)) return rc;

  // BiLSTM (d-vector folded into a per-utterance row bias), gates and cell states kept
  float* xg = at<float>(tape, L.gates);
  {
    VsProfScope ps(VS_PROF_LSTM_GEMM, stream);
    if (!prologue) {
      for (int dir = 0; dir < 2; ++dir) {
        if (int rc = vs_gemm_nt_impl(dvec, d->E, p->w_ih[dir] + K, KE, dvbias + (size_t)dir * 4 * H, 8 * H, B, 4 * H, d->E,
                                     p->b_ih[dir], p->b_hh[dir], nullptr, 0, 1, 0, VS_ACT_NONE, stream)) return rc;
      }
    }
    // the backward pass's gradient buffers are idle during the forward pass.  (prologue: the bf16 W_ih is already in its place in the
    // tape -- handed over as "prepared" so that the contraction does not convert it again)
    const _Float16* wih_ready = prologue ? reinterpret_cast<const _Float16*>(at<char>(tape, L.lstm_bf16) + Lpre.wih) : nullptr;
    if (int rc = vs_lstm_input_gemm_impl(d->math, at<float>(tape, L.feat), K, p->w_ih[0], p->w_ih[1], H, KE, xg, B * T, dvbias, T,
                                         at<float>(tape, L.gemm_scales), nhwc ? at<char>(tape, L.lstm_bf16) : at<char>(tape, L.grad0),
                                         nhwc ? L.total_bytes - L.lstm_bf16 : 2 * (L.grad1 - L.grad0), stream, nullptr, wih_ready, nullptr,
                                         feat_bf16_ready)) return rc;
  }
  float* packed = at<float>(tape, L.lstm_packed);
  if (!prologue) {
    if (int rc = vs_lstm_pack_impl(p->w_hh[0], p->w_hh[1], packed, H, stream, d->math)) return rc;
    if (nhwc) { if (int rc = vs_lstm_pack_t_impl(p->w_hh[0], p->w_hh[1], at<float>(tape, L.lstm_packed_t), H, stream, d->math)) return rc; }
  }
  {
    VsProfScope ps(VS_PROF_LSTM_REC, stream);
    if (int rc = vs_bilstm_recurrent_impl(xg, packed, at<float>(tape, L.lstm_state), at<float>(tape, L.lstm_out), xg,
                                          at<float>(tape, L.cstate), B, T, H, stream, d->math)) return rc;
  }

  // head
  VsProfScope ps_head(VS_PROF_HEAD, stream);
  const int M = B * T;
  float* h1 = at<float>(tape, L.fc1_out);
  if (d->math == VS_MATH_BF16 && vs_head_fused_supported(2 * H, d->FC1, d->FC2)) {
    // one launch, h1 in registers between the two contractions and stored once for the backward pass (head_fused.hip); the weights'
    // fragment images go into the backward pass's partial-sum scratch, idle during the forward pass
    const size_t need = vs_head_fused_packed_bytes(2 * H, d->FC1, d->FC2);
    const size_t room = L.conv_scales - L.partials;
    if (room >= need) {
      void* img = at<void>(tape, L.partials);
      if (!prologue) { if (int rc = vs_head_fused_pack_impl(p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, 2 * H, d->FC1, d->FC2, img, stream)) return rc; }
      return vs_head_fused_impl(at<float>(tape, L.lstm_out), img, h1, nullptr, mask, M, 2 * H, d->FC1, d->FC2, stream);
    }
  }
  const auto vs_gemm_nt_impl = d->math == VS_MATH_BF16 ? ::vs_gemm_nt_bf16_impl : ::vs_gemm_nt_impl;
  if (int rc = vs_gemm_nt_impl(at<float>(tape, L.lstm_out), 2 * H, p->fc1_w, 2 * H, h1, d->FC1, M, d->FC1, 2 * H,
                               p->fc1_b, nullptr, nullptr, 0, 1, 1, VS_ACT_RELU, stream)) return rc;
  return vs_gemm_nt_impl(h1, d->FC1, p->fc2_w, d->FC1, mask, d->FC2, M, d->FC2, d->FC1,
                         p->fc2_b, nullptr, nullptr, 0, 1, 0, VS_ACT_SIGMOID, stream);
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
extern "C" int vs_set_backward_overlap(int on) {
  if (on != 0 && on != 1) return -1;
  g_bwd_overlap = on;
  return 0;
}

int vs_backward(const vs_dims* d, const vs_params* p, const float* x, const float* dvec, int conv_act, int bn_mode,
                void* tape, size_t tape_bytes, const float* mask, const float* dmask, const vs_grads* g, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SideLock side_lock;
  if (g_bwd_overlap) side_lock.lock();
  vs_tape_layout L;
  if (int rc = check_tape(d, tape, tape_bytes, &L)) return rc;
  if (int rc = check_params(p)) return rc;
  VS_REQUIRE(x && dvec && mask && dmask && g, "backward: NULL argument");
  VS_REQUIRE(conv_act == VS_ACT_MISH || conv_act == VS_ACT_RELU, "backward: conv_act must be MISH or RELU");
  VS_REQUIRE(bn_mode == VS_BN_EVAL || bn_mode == VS_BN_TRAIN, "backward: unknown bn_mode %d", bn_mode);
  for (int l = 0; l < 8; ++l)
    VS_REQUIRE(g->conv[l].weight && g->conv[l].bias && g->conv[l].bn_weight && g->conv[l].bn_bias,
               "backward: conv layer %d has a NULL gradient pointer", l + 1);
  for (int dir = 0; dir < 2; ++dir)
    VS_REQUIRE(g->w_ih[dir] && g->w_hh[dir] && g->b_ih[dir] && g->b_hh[dir], "backward: NULL LSTM gradient pointer (dir %d)", dir);
  VS_REQUIRE(g->fc1_w && g->fc1_b && g->fc2_w && g->fc2_b, "backward: NULL head gradient pointer");

  const int B = d->B, T = d->T, F = d->F, H = d->H, E = d->E, FC1 = d->FC1, FC2 = d->FC2;
  const int M = B * T, K8 = 8 * F, KE = K8 + E;
  const int train = bn_mode == VS_BN_TRAIN;
  const bool det = d->math == VS_MATH_BF16 && vs_opt(VS_OPT_DETERMINISTIC) != 0;      // (the word was armed by vs_forward_train and re-armed by every user)
  VsTurnScope turn_scope(det ? at<unsigned>(tape, L.det_turn) : nullptr);
  float* part = at<float>(tape, L.partials);
  float* tmp = at<float>(tape, L.colsum_tmp);
  float* ones = at<float>(tape, L.consts);
  float* zeros = ones + 64;

  // the side stream: the leaves of the backward pass (weight gradients of the head, of the LSTM, of the convs) run there
  SideStream* side = nullptr;
  if (g_bwd_overlap) { if (int rc = side_stream(&side)) return rc; }
  SideJoin side_join;
  side_join.side = side;
  side_join.stream = stream;
  // ---- head: sigmoid, fc2, relu, fc1, relu (models/voicesplit/model.py:83-87 backwards) ----
  float* dlogits = at<float>(tape, L.dlogits);
  float* h1 = at<float>(tape, L.fc1_out);
  float* dfc1 = at<float>(tape, L.dfc1);
  float* lstm_out = at<float>(tape, L.lstm_out);
  float* dlstm = at<float>(tape, L.dlstm_out);
  {
  VsProfScope ps(VS_PROF_BWD_HEAD, stream);
  // VS_MATH_BF16: the four head contractions on bf16-rounded operands (fp32 accumulate, fp32 split-K partials)
  const auto vs_gemm_general_impl = d->math == VS_MATH_BF16 ? ::vs_gemm_general_bf16_impl : ::vs_gemm_general_impl;
  // [r5] The two weight gradients of the head are leaves: with VS_OPT_HEAD_LEAF_SIDE they go to the side stream, where they run beside
  // the BPTT -- a latency-bound launch that leaves the CUs' arithmetic idle -- instead of in front of it (same kernels, same sums).
  const bool leaf_side = side != nullptr && vs_opt(VS_OPT_HEAD_LEAF_SIDE) != 0;
  hipStream_t hs = leaf_side ? side->s : stream;
  // [r5] The two data-gradient contractions of the head (the serial ones) on gemm_bf16.hip's kernel: bf16 row copies of dlogits / dfc1
  // and bf16 copies of the two weights in the idle gradient buffer of the conv stack (its backward has not begun), the relu mask in
  // the epilogue.  Same operand roundings as the generic kernel's in-flight conversion.
  auto up256 = [](size_t x) { return (x + 255) & ~size_t(255); };
  const int K2p = (FC2 + 63) / 64 * 64, K1p = (FC1 + 63) / 64 * 64, N1p = (FC1 + 7) / 8 * 8, N2p = (2 * H + 7) / 8 * 8;
  const size_t o_df = up256((size_t)M * K2p * 2), o_w2 = o_df + up256((size_t)M * K1p * 2), o_w1 = o_w2 + up256((size_t)FC2 * N1p * 2);
  const size_t head_need = o_w1 + up256((size_t)FC1 * N2p * 2);
  const bool head_bf16 = d->math == VS_MATH_BF16 && vs_opt(VS_OPT_HEAD_BWD_GEMM) != 0 && FC1 % 4 == 0 && (2 * H) % 4 == 0 &&
                         head_need <= L.grad1 - L.grad0;
  char* hb = at<char>(tape, L.grad0);
  // the bias gradients (column sums) are leaves as well: on the side stream they use the second half of the column-sum scratch
  float* tmp_leaf = leaf_side ? tmp + (size_t)B * max3(8 * H, FC1, FC2) : tmp;
  // dfc1 = (dlogits @ W2) * (h1 > 0)
  if (head_bf16) {
    if (int rc = vs_cvt_rows_bf16_impl(p->fc2_w, FC2, FC1, FC1, hb + o_w2, N1p, stream)) return rc;
    if (int rc = vs_cvt_rows_bf16_impl(p->fc1_w, FC1, 2 * H, 2 * H, hb + o_w1, N2p, stream)) return rc;
    if (int rc = vs_sigmoid_bwd_rows_impl(dmask, mask, dlogits, M, FC2, hb, K2p, stream)) return rc;
    if (int rc = vs_gemm_bf16_impl(0, 1, hb, K2p, hb + o_w2, N1p, dfc1, FC1, nullptr, 0, M, FC1, FC2, nullptr, 0, 1, 0, stream, h1, FC1)) return rc;
  } else {
    if (int rc = vs_sigmoid_bwd_impl(dmask, mask, dlogits, (long long)M * FC2, stream)) return rc;
    if (int rc = vs_gemm_general_impl(0, 1, dlogits, FC2, p->fc2_w, nullptr, 0x7fffffff, FC1, dfc1, FC1, M, FC1, FC2,
                                      nullptr, nullptr, nullptr, 0, 1, h1, FC1, 0, 0, VS_ACT_NONE, 0, 0, 0, 1, nullptr, stream)) return rc;
  }
  if (leaf_side) {
    VS_CHECK_HIP(hipEventRecord(side->fork, stream));
    VS_CHECK_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
    side_join.forked = true;
  }
  if (int rc = vs_colsum_impl(dlogits, FC2, B, T, FC2, tmp_leaf, FC2, hs)) return rc;
  if (int rc = vs_colsum_impl(tmp_leaf, FC2, 1, B, FC2, g->fc2_b, FC2, hs)) return rc;
  // dW2 = dlogits^T @ h1
  if (int rc = vs_gemm_general_impl(1, 1, dlogits, FC2, h1, nullptr, 0x7fffffff, FC1, g->fc2_w, FC1, FC2, FC1, M,
                                    nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, 0, 0, kSplitK, part, hs)) return rc;
  if (int rc = vs_colsum_impl(dfc1, FC1, B, T, FC1, tmp_leaf, FC1, hs)) return rc;
  if (int rc = vs_colsum_impl(tmp_leaf, FC1, 1, B, FC1, g->fc1_b, FC1, hs)) return rc;
  // dW1 = dfc1^T @ relu(lstm_out)
  if (int rc = vs_gemm_general_impl(1, 1, dfc1, FC1, lstm_out, nullptr, 0x7fffffff, 2 * H, g->fc1_w, 2 * H, FC1, 2 * H, M,
                                    nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 1, VS_ACT_NONE, 0, 0, 0, kSplitK, part, hs)) return rc;
  // dlstm_out = (dfc1 @ W1) * (lstm_out > 0)
  if (head_bf16) {
    if (int rc = vs_cvt_rows_bf16_impl(dfc1, M, FC1, FC1, hb + o_df, K1p, stream)) return rc;
    if (int rc = vs_gemm_bf16_impl(0, 1, hb + o_df, K1p, hb + o_w1, N2p, dlstm, 2 * H, nullptr, 0, M, 2 * H, FC1, nullptr, 0, 1, 0, stream,
                                   lstm_out, 2 * H)) return rc;
  } else {
  if (int rc = vs_gemm_general_impl(0, 1, dfc1, FC1, p->fc1_w, nullptr, 0x7fffffff, 2 * H, dlstm, 2 * H, M, 2 * H, FC1,
                                    nullptr, nullptr, nullptr, 0, 1, lstm_out, 2 * H, 0, 0, VS_ACT_NONE, 0, 0, 0, 1, nullptr, stream)) return rc;
  }
  }

  // ---- BiLSTM: BPTT, then the batched weight / input gradients ------------------------------
  float* dxg = at<float>(tape, L.gates);
  float* wpt = at<float>(tape, L.lstm_packed_t);
  // (VS_MATH_BF16: vs_forward_train left the image in the tape [r6])
  if (d->math != VS_MATH_BF16) { if (int rc = vs_lstm_pack_t_impl(p->w_hh[0], p->w_hh[1], wpt, H, stream, d->math)) return rc; }
  {
    VsProfScope ps(VS_PROF_BWD_LSTM_REC, stream);
    if (int rc = vs_bilstm_bwd_recurrent_impl(wpt, at<float>(tape, L.lstm_bwd_state), dxg, at<float>(tape, L.cstate), dlstm,
                                              B, T, H, stream, d->math)) return rc;
  }
  float* dsum = at<float>(tape, L.dsum);
  float* feat = at<float>(tape, L.feat);
  float* dfeat = at<float>(tape, L.dfeat);
  // Only dfeat continues down the conv stack; the LSTM's own parameter gradients (dW_ih, dW_hh, biases, d-vector)
  // are leaves.  They go to the side stream (see vs_set_backward_overlap above) and run beside the HBM-bound
  // cnn8 / BatchNorm backward kernels that follow the dfeat GEMMs on the caller's stream.
  float* gsc = at<float>(tape, L.gemm_scales);
  const bool f16g = d->math != VS_MATH_FP32;
  {
    VsProfScope ps(VS_PROF_BWD_LSTM_GEMM, stream);
    // split-f16 mode: the two large contractions (dW_ih feat part, dfeat) reuse the forward's scales
    // of feat / W_ih (gemm_scales[0..3]) and one new scale for the gate gradients
    if (f16g && d->math != VS_MATH_BF16) {      // the bf16 contractions need no scale
      if (int rc = vs_pow2_scale_impl(dxg, (long long)M * 8 * H, reinterpret_cast<unsigned*>(gsc + 12), gsc + 8, stream)) return rc;
    }
  }
  const bool bf16g = d->math == VS_MATH_BF16;
  const VsLstmBf16Layout Lb = vs_lstm_bf16_layout(M, K8, H);
  char* bfb = at<char>(tape, L.lstm_bf16);
  if (bf16g) {      // the gate gradients as bf16 [M][8H]: row-form A of dfeat, col-form A of dW_ih (gemm_bf16.hip)
    // (round 6: the BPTT kernel storing this form itself beside the fp32 one saved the 0.13 ms pass and cost the recurrence 0.2 ms --
    // 32 more scattered lines per store instruction in a loop bound by exactly those: profiles/r06_experiments.md section 9)
    VsProfScope ps(VS_PROF_BWD_LSTM_GEMM, stream);
    if (int rc = vs_cvt_rows_bf16_impl(dxg, M, 8 * H, 8 * H, bfb + Lb.dxg, 8 * H, stream)) return rc;
  }
  hipStream_t ls = stream;
  // [r5] The LSTM's leaf contractions start on the side stream right here, beside the dfeat contraction and the HBM-bound BatchNorm
  // backward of the features, with dW_ih -- one persistent workgroup per CU, which slows that pass down 3x -- LAST among them (started
  // behind the features' BatchNorm backward or behind cnn8's backward they collide with cnn7's data gradient: measured slower in
  // round 5, profiles/r05_experiments.md section 3; those orders are not offered any more)
  const bool wih_last = side && bf16g;
  constexpr int leaf_late = 0;
  auto fork_leaves = [&]() -> int {
    VS_CHECK_HIP(hipEventRecord(side->fork, stream));
    VS_CHECK_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
    side_join.forked = true;
    return 0;
  };
  if (side) {
    if (!leaf_late) { if (int rc = fork_leaves()) return rc; }
    ls = side->s;
  }
  if (bf16g) {
    VsProfScope ps(VS_PROF_BWD_LSTM_GEMM, stream);
    // dfeat = dxg @ [W_ih; W_ih_reverse][:, :8F]: both directions in one contraction over K = 8H
    if (int rc = vs_gemm_bf16_impl(0, 1, bfb + Lb.dxg, 8 * H, bfb + Lb.wih, Lb.Kp, dfeat, K8, nullptr, 0, M, K8, 8 * H,
                                   nullptr, 0, 1, 0, stream)) return rc;
  } else
  {
    VsProfScope ps(VS_PROF_BWD_LSTM_GEMM, stream);
    for (int dir = 0; dir < 2; ++dir) {
      const float* dxg_d = dxg + (size_t)dir * 4 * H;
      // dfeat (+)= dxg_d @ W_ih[:, :8F]
      if (f16g) {
        if (int rc = vs_gemm_f16x3_impl(0, 1, dxg_d, 8 * H, p->w_ih[dir], nullptr, 0x7fffffff, KE, dfeat, K8, M, K8, 4 * H,
                                        nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, dir, gsc + 8, gsc + 2, stream, d->math)) return rc;
      } else {
        if (int rc = vs_gemm_general_impl(0, 1, dxg_d, 8 * H, p->w_ih[dir], nullptr, 0x7fffffff, KE, dfeat, K8, M, K8, 4 * H,
                                          nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, dir, 0, 0, 1, nullptr, stream)) return rc;
      }
    }
  }
  auto lstm_leaves = [&]() -> int {
    VsProfScope ps(VS_PROF_BWD_LSTM_GEMM, ls);
    // sum_t of the gate gradients per utterance, then over the batch: bias / d-vector-column gradients, all leaves [r6: here, on the
    // leaves' stream, instead of in front of the dfeat contraction on the caller's]
    if (int rc = vs_colsum_impl(dxg, 8 * H, B, T, 8 * H, dsum, 8 * H, ls)) return rc;
    if (int rc = vs_colsum_impl(dsum, 8 * H, 1, B, 8 * H, tmp, 8 * H, ls)) return rc;
    for (int dir = 0; dir < 2; ++dir) {
      VS_CHECK_HIP(hipMemcpyAsync(g->b_ih[dir], tmp + (size_t)dir * 4 * H, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, ls));
      VS_CHECK_HIP(hipMemcpyAsync(g->b_hh[dir], tmp + (size_t)dir * 4 * H, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, ls));
      const float* dxg_d = dxg + (size_t)dir * 4 * H;
      // dW_ih[:, :8F] = dxg_d^T @ feat
      if (bf16g) {
        // both directions in one col x col contraction over K = B*T: rows < 4H -> dW_ih, the rest -> dW_ih_reverse
        if (dir == 0 && !wih_last) {
          if (int rc = vs_gemm_bf16_impl(1, 1, bfb + Lb.dxg, 8 * H, bfb + Lb.feat, Lb.Kp, g->w_ih[0], KE, g->w_ih[1], 4 * H, 8 * H, K8, M,
                                         nullptr, 0, 1, 0, ls)) return rc;
        }
      } else if (f16g) {
        if (int rc = vs_gemm_f16x3_impl(1, 1, dxg_d, 8 * H, feat, nullptr, 0x7fffffff, K8, g->w_ih[dir], KE, 4 * H, K8, M,
                                        nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, gsc + 8, gsc, ls, d->math)) return rc;
      } else {
        if (int rc = vs_gemm_general_impl(1, 1, dxg_d, 8 * H, feat, nullptr, 0x7fffffff, K8, g->w_ih[dir], KE, 4 * H, K8, M,
                                          nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, 0, 0, 1, nullptr, ls)) return rc;
      }
      // dW_ih[:, 8F:] = (sum_t dxg_d)^T @ dvec    (the repeated d-vector columns, model.py:77-81)
      if (int rc = vs_gemm_general_impl(1, 1, dsum + (size_t)dir * 4 * H, 8 * H, dvec, nullptr, 0x7fffffff, E, g->w_ih[dir] + K8, KE,
                                        4 * H, E, B, nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, 0, 0, 1, nullptr, ls)) return rc;
      // dW_hh = sum_t dgates_t^T h_{t-1}: the lstm_out rows shifted by one frame inside each utterance
      // (VS_MATH_BF16: on bf16-rounded operands like the other contractions of this configuration)
      if (int rc = (bf16g ? vs_gemm_general_bf16_impl : vs_gemm_general_impl)(1, 1, dxg_d, 8 * H, lstm_out + (size_t)dir * H, nullptr, 0x7fffffff, 2 * H, g->w_hh[dir], H,
                                        4 * H, H, M, nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0,
                                        dir ? 1 : -1, T, kSplitK, part, ls)) return rc;
      if (g->dvec) {
        if (int rc = vs_gemm_general_impl(0, 1, dsum + (size_t)dir * 4 * H, 8 * H, p->w_ih[dir] + K8, nullptr, 0x7fffffff, KE, g->dvec, E,
                                          B, E, 4 * H, nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, dir, 0, 0, 1, nullptr, ls)) return rc;
      }
    }
    if (bf16g && wih_last) {      // (3): the one-workgroup-per-CU contraction behind the small leaves
      if (int rc = vs_gemm_bf16_impl(1, 1, bfb + Lb.dxg, 8 * H, bfb + Lb.feat, Lb.Kp, g->w_ih[0], KE, g->w_ih[1], 4 * H, 8 * H, K8, M,
                                     nullptr, 0, 1, 0, ls)) return rc;
    }
    return 0;
  };
  // ABI 9: the caller's event for "head + BiLSTM gradients are final" (vs_grads.leaves_event): recorded behind the last leaf launch,
  // on the stream the leaves ran on (the head's leaves were enqueued earlier on the same stream, or on `stream` ahead of the fork)
  auto leaves_done = [&]() -> int {
    if (g->leaves_event) VS_CHECK_HIP(hipEventRecord((hipEvent_t)g->leaves_event, ls));
    if (side) VS_CHECK_HIP(hipEventRecord(side->leaves, side->s));      // (the shared partial-sum scratch `part` is free behind this)
    return 0;
  };
  if (!leaf_late) { if (int rc = lstm_leaves()) return rc; if (int rc = leaves_done()) return rc; }

  // ---- conv stack, cnn8 .. cnn1 (models/voicesplit/model.py:15-52 backwards) ------------------
  float* scale = at<float>(tape, L.bn_scale);
  float* shift = at<float>(tape, L.bn_shift);
  float* mean = at<float>(tape, L.bn_mean);
  float* invstd = at<float>(tape, L.bn_invstd);
  double* stats = at<double>(tape, L.bn_stats);
  float* coef = at<float>(tape, L.bn_coef);
  // split-f16 convs: dz of layer l is the operand of its data- and weight-gradient launches; the
  // BatchNorm backward pass that produces it folds its |max| into slot 8+l
  float* cs = at<float>(tape, L.conv_scales);
  const bool f16 = d->math != VS_MATH_FP32;
  if (f16) VS_CHECK_HIP(hipMemsetAsync(cs + 8 * VS_SCALE_SLOT_FLOATS, 0, 8 * VS_SCALE_SLOT_FLOATS * sizeof(float), stream));
  auto bn_bwd = [&](int l, const float* da, const float* z, float* dz, int C, long long R, int Lrow) -> int {
    VsProfScope ps(VS_PROF_BWD_BN, stream);
    unsigned* amax = (f16 && l >= 1 && l <= 6) ? vs_amax_slot(cs + VS_SCALE_SLOT_FLOATS * (8 + l)) : nullptr;
    return vs_bn_act_bwd_impl(da, z, dz, C, R, Lrow, conv_act, train, scale + 64 * l, shift + 64 * l, mean + 64 * l,
                              invstd + 64 * l, g->conv[l].bn_weight, g->conv[l].bn_bias, g->conv[l].bias, stats, coef, amax, stream);
  };
  // cnn8: dfeat -> dz8 (in place) -> dW8, dA7
  if (int rc = bn_bwd(7, dfeat, at<float>(tape, L.z8), dfeat, 8, (long long)M * 8, F)) return rc;
  if (d->math == VS_MATH_BF16) {
    // BASELINE configs[2]: the conv stack backward on channels-last bf16 tensors (nhwc_edge.hip, conv_nhwc.hip,
    // wgrad_nhwc.hip).  Same chain and the same side-stream schedule as below: layer l's weight gradient runs beside
    // the BatchNorm backward of layer l-1.
    const long long npix = (long long)B * T * F;
    void* gb[2] = {at<void>(tape, L.grad0), at<void>(tape, L.grad1)};
    int c = 0;
    // Every kernel that produces a layer's input gradient (cnn8's backward, the data-gradient convs) applies the
    // activation derivative of the layer below on the spot and accumulates the BatchNorm backward sums (the dy forms):
    // the BatchNorm backward proper is then finalize + one pass.
    // (the scratch is cleared once, in front of cnn8's backward; every finalize then clears it behind itself: vs_fold_slots)
    // The backward keeps the two-kernel finalize and a memset in front of every dy launch: the fused form (one launch that folds,
    // finalizes and clears) measured +1.6 ms per step beside a 2048-block BatchNorm pass and neutral beside the one-block-per-CU pass
    // (round 5: profiles/r05_experiments.md section 2 and its last paragraph).
    constexpr int kStatsDoubles = 0;      // (re-measured in round 6, call 12: 46.85 ms either way, three alternating runs)
    {
      VsProfScope ps(VS_PROF_BWD_EDGE, stream);
      VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * VS_BN_STAT_SLOTS * 128, stream));
      // partial sums in the idle second gradient buffer: `part` may still be in use by the LSTM leaf GEMMs on the side stream
      // train mode: a7 was never written (see vs_forward_train): recomputed from z7 by the kernel
      if (int rc = vs_nhwc_conv_last_bwd_impl(dfeat, p->conv[7].weight, train ? nullptr : at<void>(tape, L.a[6]), gb[c], at<float>(tape, L.grad1),
                                              g->conv[7].weight, B, T, F, at<void>(tape, L.z[6]), conv_act, scale + 64 * 6, shift + 64 * 6,
                                              mean + 64 * 6, invstd + 64 * 6, stats, stream)) return rc;
    }
    {
      // [r6] Roles of the two streams: the MATRIX kernels (data gradient, weight gradient) stay on the caller's stream, back to back; the
      // HBM-bound pass that consumes the gradient a data gradient just produced -- layer l-1's BatchNorm backward, cnn1's one-pass backward
      // behind cnn2's -- forks to the side stream and runs beside layer l's weight gradient.  Rounds 3-5 had it the other way round (weight
      // gradients on the side stream): the weight gradient, the longer of each pair, then started a cross-stream wait (~25 us) late and the
      // next data gradient waited for it across streams again; now the wait at the end of a pair is for a pass that ends ~0.1 ms before the
      // weight gradient does, and cnn1's backward (0.95 ms alone, VALU-bound) runs beside cnn2's weight gradient (0.8 ms) instead of behind it
      // (1.55 ms for the pair instead of 1.77).  45.46 -> 45.17 ms per step, three alternating runs (profiles/r06_experiments.md section 10).
      auto from_dy = [&](int l, void* gbuf, hipStream_t s, int beside) -> int {
        VsProfScope ps(VS_PROF_BWD_BN, s);
        return vs_nhwc_bn_bwd_from_dy_impl(gbuf, at<void>(tape, L.z[l]), gbuf, npix, train, scale + 64 * l, mean + 64 * l, invstd + 64 * l,
                                           g->conv[l].bn_weight, g->conv[l].bn_bias, g->conv[l].bias, stats, coef, s, kStatsDoubles, beside);
      };
      auto first_bwd = [&](void* gbuf, hipStream_t s) -> int {
        VsProfScope ps(VS_PROF_BWD_EDGE, s);
        // (deterministic mode's slot scratch: behind the turn words -- `part` belongs to the weight gradient beside it)
        return vs_nhwc_first_bwd_impl(gbuf, x, p->conv[0].weight, p->conv[0].bias, B, T, F, conv_act, train, scale, shift, mean, invstd,
                                      g->conv[0].bn_weight, g->conv[0].bn_bias, g->conv[0].bias, g->conv[0].weight,
                                      at<double>(tape, L.first_acc) + 64, s, at<double>(tape, L.first_acc),
                                      det ? at<double>(tape, L.det_turn) + VS_TURN_WORDS * 4 / 8 : nullptr);
      };
      if (int rc = from_dy(6, gb[c], stream, 0)) return rc;      // cnn7's BatchNorm backward: nothing to run beside yet
      bool part_free = side == nullptr;
      for (int i = 5; i >= 0; --i) {
        const int l = i + 1;                                       // gb[c] = dz of layer l
        {
          VsProfScope ps(VS_PROF_BWD_DGRAD + i, stream);
          const void* pack_t = at<void>(tape, L.conv_packed_t[i]);      // (written by vs_forward_train [r6])
          if (l == 1) {
            // cnn2's data gradient is the plain conv: the activation derivative of cnn1 needs z1, which cnn1's one-pass backward recomputes from x
            if (int rc = vs_nhwc_conv_impl(gb[c], pack_t, ones, zeros, gb[c ^ 1], B, T, F, kMid[i].kt, kMid[i].kf, kMid[i].dil, VS_ACT_NONE,
                                           nullptr, stream)) return rc;
          } else {
            VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * VS_BN_STAT_SLOTS * 128, stream));
            if (int rc = vs_nhwc_conv_dy_impl(gb[c], pack_t, gb[c ^ 1], at<void>(tape, L.z[l - 1]), conv_act, scale + 64 * (l - 1),
                                              shift + 64 * (l - 1), mean + 64 * (l - 1), invstd + 64 * (l - 1), stats,
                                              B, T, F, kMid[i].kt, kMid[i].kf, kMid[i].dil, stream)) return rc;
          }
        }
        hipStream_t bs = stream;
        if (side) {
          VS_CHECK_HIP(hipEventRecord(side->fork, stream));
          VS_CHECK_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
          bs = side->s;
        }
        if (l >= 2) { if (int rc = from_dy(l - 1, gb[c ^ 1], bs, side ? 1 : 0)) return rc; }
        else { if (int rc = first_bwd(gb[c ^ 1], bs)) return rc; }
        if (side) VS_CHECK_HIP(hipEventRecord(side->join, side->s));
        if (!part_free) {      // the LSTM's leaf contractions on the side stream share the partial-sum scratch: long finished by now
          VS_CHECK_HIP(hipStreamWaitEvent(stream, side->leaves, 0));
          part_free = true;
        }
        {
          VsProfScope ps(VS_PROF_BWD_WGRAD + i, stream);
          if (int rc = vs_nhwc_wgrad_impl(gb[c], at<void>(tape, L.a[l - 1]), part, g->conv[l].weight, B, T, F, kMid[i].kt, kMid[i].kf,
                                          kMid[i].dil, stream)) return rc;
        }
        if (side) VS_CHECK_HIP(hipStreamWaitEvent(stream, side->join, 0));      // (the pass beside it: done before the weight gradient is)
        c ^= 1;
      }
      side_join.forked = false;
      return 0;
    }
  }
  float* gbuf[2] = {at<float>(tape, L.grad0), at<float>(tape, L.grad1)};
  int cur = 0;
  {
    // cnn8's weight gradient: a leaf as well, and the partial-sum scratch it shares with the side stream's other
    // users is then only ever touched there, in stream order
    if (side) {
      VS_CHECK_HIP(hipEventRecord(side->fork, stream));
      VS_CHECK_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
    }
    {
      VsProfScope ps(VS_PROF_BWD_EDGE, ls);
      if (int rc = vs_conv_last_wgrad_impl(dfeat, at<float>(tape, L.a[6]), part, g->conv[7].weight, B, T, F, ls)) return rc;
    }
    VsProfScope ps(VS_PROF_BWD_EDGE, stream);
    if (int rc = vs_conv_last_dgrad_impl(dfeat, p->conv[7].weight, gbuf[cur], B, T, F, stream)) return rc;
  }
  float* pack_tmp = at<float>(tape, L.pack_tmp);
  bool wgrad_pending = false;         // a weight gradient on the side stream still reads the buffer the next data gradient writes
  for (int i = 5; i >= 0; --i) {
    const int l = i + 1;   // cnn(l+1), conv index l
    if (int rc = bn_bwd(l, gbuf[cur], at<float>(tape, L.z[l]), gbuf[cur], 64, (long long)B * 64, T * F)) return rc;
    float* sc_bwd = at<float>(tape, L.conv_scales) + VS_SCALE_SLOT_FLOATS * (8 + l);
    if (wgrad_pending) {
      VS_CHECK_HIP(hipStreamWaitEvent(stream, side->join, 0));
      wgrad_pending = false;
    }
    {
      VsProfScope ps(VS_PROF_BWD_DGRAD + i, stream);
      if (int rc = vs_conv64_layer_impl(d->math, gbuf[cur], p->conv[l].weight, pack_tmp, sc_bwd, 1,
                                        ones, zeros, gbuf[cur ^ 1], B, T, F, kMid[i].kt, kMid[i].kf, kMid[i].dil, VS_ACT_NONE, 1,
                                        nullptr, stream)) return rc;
    }
    hipStream_t ws = stream;
    if (side) {
      VS_CHECK_HIP(hipEventRecord(side->fork, stream));
      VS_CHECK_HIP(hipStreamWaitEvent(side->s, side->fork, 0));
      ws = side->s;
    }
    {
      // after the data gradient: in split-f16 mode it reuses the scale of dz that launch derived
      // (sc_bwd[0..1]) and the scale of the layer input the forward derived (slot l)
      VsProfScope ps(VS_PROF_BWD_WGRAD + i, ws);
      if (d->math != VS_MATH_FP32) {
        if (int rc = vs_conv64_wgrad_f16x3_impl(gbuf[cur], at<float>(tape, L.a[l - 1]), sc_bwd, at<float>(tape, L.conv_scales) + VS_SCALE_SLOT_FLOATS * l,
                                                part, g->conv[l].weight, B, T, F, kMid[i].kt, kMid[i].kf, kMid[i].dil, ws, d->math)) return rc;
      } else {
        if (int rc = vs_conv64_wgrad_impl(gbuf[cur], at<float>(tape, L.a[l - 1]), part, g->conv[l].weight, B, T, F,
                                          kMid[i].kt, kMid[i].kf, kMid[i].dil, ws)) return rc;
      }
    }
    if (side) {
      VS_CHECK_HIP(hipEventRecord(side->join, side->s));
      wgrad_pending = true;
    }
    cur ^= 1;
  }
  // cnn2's weight gradient reads gbuf[cur ^ 1], which cnn1's backward uses as scratch -- and the caller's stream has
  // to see everything the side stream produced: join here
  if (side) {
    VS_CHECK_HIP(hipEventRecord(side->join, side->s));
    VS_CHECK_HIP(hipStreamWaitEvent(stream, side->join, 0));
    side_join.forked = false;
  }
  if (F < 4 || (long long)T * F >= (1 << 24)) {   // packs spanning >2 frames / float frame index: unfused path
    if (int rc = bn_bwd(0, gbuf[cur], at<float>(tape, L.z[0]), gbuf[cur], 64, (long long)B * 64, T * F)) return rc;
    VsProfScope ps(VS_PROF_BWD_EDGE, stream);
    return vs_conv_first_wgrad_impl(gbuf[cur], x, at<double>(tape, L.first_acc), g->conv[0].weight, B, T, F, stream);
  }
  // cnn1: BatchNorm backward and dW1 together; dZ1 is never stored.  The idle gradient buffer holds
  // the zero-padded input rows.
  VsProfScope ps(VS_PROF_BWD_BN, stream);
  return vs_bn_act_bwd_first_impl(gbuf[cur], at<float>(tape, L.z[0]), x, gbuf[cur ^ 1], B, T, F, conv_act, train, scale, shift, mean,
                                  invstd, g->conv[0].bn_weight, g->conv[0].bn_bias, g->conv[0].bias, g->conv[0].weight, stats, coef,
                                  at<double>(tape, L.first_acc), stream);
}

// Did the persistent BiLSTM kernels of the last calls on these buffers complete?  (A launch that could not be resident
// falls back to the step kernels before it starts; one whose bounded spin gave up anyway -- CUs taken away by another
// process mid-launch -- sets a word in its state buffer and poisons its output with NaN.)  Synchronises `stream`.
int vs_lstm_status(const vs_dims* d, const void* tape, size_t tape_bytes, const void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VS_REQUIRE(tape || workspace, "lstm_status: pass a tape and / or a workspace");
  unsigned words[3] = {0u, 0u, 0u};
  if (tape) {
    vs_tape_layout L;
    if (int rc = tape_layout(d, &L)) return rc;
    VS_REQUIRE(tape_bytes >= L.total_bytes, "lstm_status: tape too small");
    const char* t = static_cast<const char*>(tape);
    VS_CHECK_HIP(hipMemcpyAsync(&words[0], t + L.lstm_state + (vs_lstm_state_floats(d->B, d->H) - 64) * sizeof(float), 4, hipMemcpyDeviceToHost, stream));
    VS_CHECK_HIP(hipMemcpyAsync(&words[1], t + L.lstm_bwd_state + (vs_lstm_bwd_state_floats(d->B, d->H) - 64) * sizeof(float), 4, hipMemcpyDeviceToHost, stream));
  }
  if (workspace) {
    vs_ws_layout W;
    if (int rc = vs_workspace_layout(d, &W)) return rc;
    VS_REQUIRE(workspace_bytes >= W.total_bytes, "lstm_status: workspace too small");
    VS_CHECK_HIP(hipMemcpyAsync(&words[2], static_cast<const char*>(workspace) + W.lstm_state + (vs_lstm_state_floats(d->B, d->H) - 64) * sizeof(float), 4,
                                hipMemcpyDeviceToHost, stream));
  }
  VS_CHECK_HIP(hipStreamSynchronize(stream));
  return (words[0] == 1u || words[1] == 1u || words[2] == 1u) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// unit-test surface
// ---------------------------------------------------------------------------------------------
int vs_conv64_pack_dgrad(const float* w, float* packed, int KT, int KF, void* stream) {
  return vs_conv64_pack_impl(w, packed, KT, KF, 1, (hipStream_t)stream);
}

int vs_conv64_wgrad(const float* dz, const float* in, float* partials, float* dw, int B, int T, int F, int KT, int KF,
                    int dil, void* stream) {
  VS_REQUIRE(dz && in && partials && dw, "conv64_wgrad: NULL argument");
  return vs_conv64_wgrad_impl(dz, in, partials, dw, B, T, F, KT, KF, dil, (hipStream_t)stream);
}

int vs_conv64_wgrad_f16x3(const float* dz, const float* in, float* partials, float* dw, float* scratch8,
                          int B, int T, int F, int KT, int KF, int dil, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VS_REQUIRE(dz && in && partials && dw && scratch8, "conv64_wgrad_f16x3: NULL argument");
  unsigned* amax = reinterpret_cast<unsigned*>(scratch8 + 4);
  if (int rc = vs_pow2_scale_impl(dz, (long long)B * 64 * T * F, amax, scratch8, stream)) return rc;
  if (int rc = vs_pow2_scale_impl(in, (long long)B * 64 * T * F, amax + 1, scratch8 + 2, stream)) return rc;
  return vs_conv64_wgrad_f16x3_impl(dz, in, scratch8, scratch8 + 2, partials, dw, B, T, F, KT, KF, dil, stream);
}

int vs_bn_act_bwd(const float* da, const float* z, float* dz, int C, long long R, int L, int act, int bn_mode,
                  const float* scale, const float* shift, const float* mean, const float* invstd,
                  float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, void* stream) {
  VS_REQUIRE(da && z && dz && scale && shift && mean && invstd && stats && coef, "bn_act_bwd: NULL argument");
  return vs_bn_act_bwd_impl(da, z, dz, C, R, L, act, bn_mode == VS_BN_TRAIN, scale, shift, mean, invstd, dgamma, dbeta, dbias,
                            stats, coef, nullptr, (hipStream_t)stream);
}

int vs_bn_act_bwd_first(const float* da, const float* z, const float* x, float* xpad, int B, int T, int F, int act, int bn_mode,
                        const float* scale, const float* shift, const float* mean, const float* invstd,
                        float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc, void* stream) {
  VS_REQUIRE(da && z && x && xpad && scale && shift && mean && invstd && dw && stats && coef && acc, "bn_act_bwd_first: NULL argument");
  return vs_bn_act_bwd_first_impl(da, z, x, xpad, B, T, F, act, bn_mode == VS_BN_TRAIN, scale, shift, mean, invstd, dgamma, dbeta, dbias,
                                  dw, stats, coef, acc, (hipStream_t)stream);
}

int vs_conv_last_dgrad(const float* dz, const float* w, float* din, int B, int T, int F, void* stream) {
  return vs_conv_last_dgrad_impl(dz, w, din, B, T, F, (hipStream_t)stream);
}

int vs_conv_last_wgrad(const float* dz, const float* in, float* partials, float* dw, int B, int T, int F, void* stream) {
  return vs_conv_last_wgrad_impl(dz, in, partials, dw, B, T, F, (hipStream_t)stream);
}

int vs_conv_first_wgrad(const float* dz, const float* x, double* acc, float* dw, int B, int T, int F, void* stream) {
  return vs_conv_first_wgrad_impl(dz, x, acc, dw, B, T, F, (hipStream_t)stream);
}

int vs_gemm(int layout_a, int layout_w, const float* A, int lda, const float* W, int ldw, float* C, int ldc,
            int M, int N, int K, const float* bias, const float* gate, int ldg, int a_relu, int w_relu, int act,
            int accumulate, int w_shift, int w_group, int splits, float* partials, void* stream) {
  VS_REQUIRE(A && W && C, "gemm: NULL argument");
  return vs_gemm_general_impl(layout_a, layout_w, A, lda, W, nullptr, 0x7fffffff, ldw, C, ldc, M, N, K, bias, nullptr, nullptr, 0, 1,
                              gate, ldg, a_relu, w_relu, act, accumulate, w_shift, w_group, splits, partials, (hipStream_t)stream);
}

int vs_gemm_f16x3(int layout_a, int layout_w, const float* A, int lda, const float* W, int ldw, float* C, int ldc,
                  int M, int N, int K, const float* bias, const float* gate, int ldg, int a_relu, int w_relu, int act,
                  int accumulate, float* scratch8, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VS_REQUIRE(A && W && C && scratch8, "gemm_f16x3: NULL argument");
  unsigned* amax = reinterpret_cast<unsigned*>(scratch8 + 4);
  // the operands are dense [rows][ld] buffers here: scale over the whole buffers
  if (int rc = vs_pow2_scale_impl(A, (long long)(layout_a ? K : M) * lda, amax, scratch8, stream)) return rc;
  if (int rc = vs_pow2_scale_impl(W, (long long)(layout_w ? K : N) * ldw, amax + 1, scratch8 + 2, stream)) return rc;
  return vs_gemm_f16x3_impl(layout_a, layout_w, A, lda, W, nullptr, 0x7fffffff, ldw, C, ldc, M, N, K, bias, nullptr, nullptr, 0, 1,
                            gate, ldg, a_relu, w_relu, act, accumulate, scratch8, scratch8 + 2, stream);
}

int vs_bilstm_recurrent_train(const float* xg, const float* packed_whh, float* state, float* out, float* gates_save,
                              float* c_save, int B, int T, int H, void* stream) {
  return vs_bilstm_recurrent_impl(xg, packed_whh, state, out, gates_save, c_save, B, T, H, (hipStream_t)stream);
}

int vs_lstm_pack_t(const float* w_hh_fwd, const float* w_hh_bwd, float* packed_t, int H, void* stream) {
  return vs_lstm_pack_t_impl(w_hh_fwd, w_hh_bwd, packed_t, H, (hipStream_t)stream);
}

int vs_bilstm_recurrent_bwd(const float* packed_t, float* state, float* gates, const float* c_all, const float* dout,
                            int B, int T, int H, void* stream) {
  return vs_bilstm_bwd_recurrent_impl(packed_t, state, gates, c_all, dout, B, T, H, (hipStream_t)stream);
}

int vs_sigmoid_bwd(const float* dmask, const float* mask, float* dlogits, long long n, void* stream) {
  return vs_sigmoid_bwd_impl(dmask, mask, dlogits, n, (hipStream_t)stream);
}

int vs_colsum(const float* x, int ld, int groups, int rows, int N, float* out, int ldo, void* stream) {
  return vs_colsum_impl(x, ld, groups, rows, N, out, ldo, (hipStream_t)stream);
}

}  // extern "C"
