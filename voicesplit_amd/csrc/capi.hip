// extern "C" surface of libvoicesplit_hip.so (declared in include/voicesplit_hip.h) and the
// orchestration of the forward pass: which kernel runs on which buffer, in which order.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/voicesplit_hip.h"
#include "vs_internal.h"

// ---- error string --------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void vs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- opt-in per-stage timing (bench.py roofline leg) -------------------------------------------
// vs_profile_begin(n) pre-creates HIP events for n forward calls; while enabled every stage of
// vs_conv_stack_fwd / vs_bilstm_fwd / vs_head_fwd is bracketed by two events recorded on the
// caller's stream (no synchronisation, ~1 us each).  vs_profile_end() synchronises the events,
// sums the elapsed time per slot and frees them.  Instrumentation only: process-global, not
// thread safe, off by default.
// defaults of vs_set_option (include/voicesplit_hip.h, enum vs_option)
int g_vs_options[VS_OPT_COUNT] = {/*FWD_PROLOGUE*/ 1, /*HEAD_LEAF_SIDE*/ 1, /*FEAT_ROWS*/ 1, /*HEAD_BWD_GEMM*/ 1, /*ABLATION*/ 0, /*DETERMINISTIC*/ 0};
thread_local unsigned* g_vs_turn = nullptr;

namespace {
struct Prof {
  bool on = false;
  int max_calls = 0;
  int calls[VS_PROF_SLOTS] = {0};
  hipEvent_t* ev = nullptr;   // [max_calls][VS_PROF_SLOTS][2]
} g_prof;

}  // namespace

// a slot may be entered several times per step (e.g. one BatchNorm scope per layer): every entry
// gets its own event pair, `calls` counts entries
VsProfScope::VsProfScope(int slot, hipStream_t s_) : s(s_), idx(-1) {
  if (!g_prof.on || g_prof.calls[slot] >= g_prof.max_calls) return;
  idx = (g_prof.calls[slot]++ * VS_PROF_SLOTS + slot) * 2;
  (void)hipEventRecord(g_prof.ev[idx], s);
}
VsProfScope::~VsProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof.ev[idx + 1], s); }
typedef VsProfScope ProfScope;

namespace {

constexpr float kBnEps = 1e-5f;       // nn.BatchNorm2d default (models/voicesplit/model.py:19)
constexpr float kBnMomentum = 0.1f;

// conv-stack table (models/voicesplit/model.py:15-52): KT, KF, time dilation
struct Spec { int kt, kf, dil; };
constexpr Spec kMid[6] = {{7, 1, 1}, {5, 5, 1}, {5, 5, 2}, {5, 5, 4}, {5, 5, 8}, {5, 5, 16}};

inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

int check_dims(const vs_dims* d) {
  VS_REQUIRE(d != nullptr, "dims is NULL");
  VS_REQUIRE(d->B > 0 && d->T > 0 && d->F > 0 && d->E > 0 && d->H > 0 && d->FC1 > 0 && d->FC2 > 0,
             "dims must be positive: B=%d T=%d F=%d E=%d H=%d FC1=%d FC2=%d", d->B, d->T, d->F, d->E, d->H, d->FC1, d->FC2);
  VS_REQUIRE(d->H % 8 == 0, "lstm_dim H=%d must be a multiple of 8", d->H);
  VS_REQUIRE(d->math == VS_MATH_FP32 || d->math == VS_MATH_F16X3 || d->math == VS_MATH_BF16, "dims.math=%d is not a VS_MATH_* code", d->math);
  VS_REQUIRE((long long)d->B * d->T < 2147483647LL / 8, "B*T too large");
  return 0;
}

// packed weights of mid layer i in whichever form the configuration uses (fp32 MFMA fragments are the largest of the NCHW
// forms; the channels-last split-f16 forward keeps its row norms / scale / per-call plan behind the packed planes)
size_t conv_packed_bytes(int i) {
  const size_t a = vs_conv64_packed_floats(kMid[i].kt, kMid[i].kf) * 4, b = vs_nhwc_f16x3_layer_scratch_bytes(kMid[i].kt, kMid[i].kf);
  return a > b ? a : b;
}

// Eval-mode forward of the fp32-class arithmetic: channels-last hi / lo planes (conv_nhwc_f16x3.hip).  (The [B][64][T][F] kernels of
// rounds 1-2 serve train mode -- its tape is fp32 NCHW -- and the strict fp32 arithmetic; as an eval route they were the A/B arm of
// round 4 (1561 against 1722-1836 utt/s) and are not offered any more, so a prepared-weights blob has ONE conv-weight format.)

int layout(const vs_dims* d, vs_ws_layout* L) {
  if (int rc = check_dims(d)) return rc;
  const size_t B = d->B, T = d->T, F = d->F, H = d->H;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  L->act0 = take(B * 64 * T * F * 4);
  L->act1 = take(B * 64 * T * F * 4);
  L->feat = take(B * T * 8 * F * 4);
  L->dvbias = take(B * 8 * H * 4);
  L->xg = take(B * T * 8 * H * 4);
  L->lstm_out = take(B * T * 2 * H * 4);
  L->fc1_out = take(B * T * (size_t)d->FC1 * 4);
  for (int i = 0; i < 6; ++i) L->conv_packed[i] = take(conv_packed_bytes(i));
  L->bn_scale = take(8 * 64 * 4);
  L->bn_shift = take(8 * 64 * 4);
  L->bn_stats = take((size_t)VS_BN_STAT_SLOTS * 64 * 2 * 8);    // partial slots of one layer at a time (stream-ordered reuse)
  L->lstm_packed = take(vs_lstm_packed_floats(d->H) * 4);
  L->lstm_state = take(vs_lstm_state_floats(d->B, d->H) * 4);
  L->conv_scales = take(8 * VS_SCALE_SLOT_FLOATS * 4);
  L->gemm_scales = take(16 * 4);
  L->total_bytes = off;
  return 0;
}

template <typename T>
inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }

int check_ws(const vs_dims* d, void* ws, size_t ws_bytes, vs_ws_layout* L) {
  if (int rc = layout(d, L)) return rc;
  VS_REQUIRE(ws != nullptr, "workspace is NULL");
  VS_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255) == 0, "workspace must be 256-byte aligned");
  VS_REQUIRE(ws_bytes >= L->total_bytes, "workspace too small: %zu < %zu bytes", ws_bytes, L->total_bytes);
  return 0;
}

// ---- prepared weights (vs_prepare_weights / vs_forward_prepared): everything an eval-mode forward derives from
// the parameters alone -- BatchNorm folded into per-channel scale/shift, conv weights in MFMA fragment order with
// their power-of-two scale, W_ih split into f16 halves with its scale, W_hh in fragment order.  Independent of B, T.
struct PrepLayout {
  size_t bn_scale, bn_shift, conv_packed[6], gemm_wscale, wih_hi, wih_lo, lstm_packed, head_packed, total_bytes;
};
struct Prep {
  float *bn_scale, *bn_shift;
  void* conv_packed[6];
  float* gemm_wscale;     // [8]: scale2 of W_ih at [0..1], |max| scratch at [4]
  _Float16 *wih_hi, *wih_lo;
  float* lstm_packed;
  void* head_packed;      // VS_MATH_BF16: fc1 / fc2 in the fused head's fragment order (head_fused.hip), else NULL
};

int prep_layout(const vs_dims* d, PrepLayout* L) {
  if (int rc = check_dims(d)) return rc;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  L->bn_scale = take(8 * 64 * 4);
  L->bn_shift = take(8 * 64 * 4);
  for (int i = 0; i < 6; ++i) L->conv_packed[i] = take(conv_packed_bytes(i));
  L->gemm_wscale = take(8 * 4);
  const size_t Kp = ((size_t)8 * d->F + VS_GEMM_KPAD - 1) / VS_GEMM_KPAD * VS_GEMM_KPAD;
  L->wih_hi = take((size_t)8 * d->H * Kp * 2);
  L->wih_lo = take((size_t)8 * d->H * Kp * 2);
  L->lstm_packed = take(vs_lstm_packed_floats(d->H) * 4);
  L->head_packed = take(d->math == VS_MATH_BF16 && vs_head_fused_supported(2 * d->H, d->FC1, d->FC2) ? vs_head_fused_packed_bytes(2 * d->H, d->FC1, d->FC2) : 0);
  L->total_bytes = off;
  return 0;
}

int prep_pointers(const vs_dims* d, const void* blob, size_t bytes, Prep* P) {
  PrepLayout L;
  if (int rc = prep_layout(d, &L)) return rc;
  VS_REQUIRE(blob != nullptr, "prepared weights: NULL buffer");
  VS_REQUIRE((reinterpret_cast<uintptr_t>(blob) & 255) == 0, "prepared weights: buffer must be 256-byte aligned");
  VS_REQUIRE(bytes >= L.total_bytes, "prepared weights: buffer too small: %zu < %zu bytes", bytes, L.total_bytes);
  void* b = const_cast<void*>(blob);
  P->bn_scale = at<float>(b, L.bn_scale);
  P->bn_shift = at<float>(b, L.bn_shift);
  for (int i = 0; i < 6; ++i) P->conv_packed[i] = at<char>(b, L.conv_packed[i]);
  P->gemm_wscale = at<float>(b, L.gemm_wscale);
  P->wih_hi = at<_Float16>(b, L.wih_hi);
  P->wih_lo = at<_Float16>(b, L.wih_lo);
  P->lstm_packed = at<float>(b, L.lstm_packed);
  P->head_packed = (d->math == VS_MATH_BF16 && vs_head_fused_supported(2 * d->H, d->FC1, d->FC2)) ? at<void>(b, L.head_packed) : nullptr;
  return 0;
}

int conv_stack_impl(const vs_dims* d, const vs_params* p, const float* x, int conv_act, int bn_mode,
                    void* ws, const vs_ws_layout& L, float* feat, hipStream_t stream, const Prep* prep, bool* feat_rows = nullptr);
int bilstm_impl(const vs_dims* d, const vs_params* p, const float* feat, const float* dvec,
                void* ws, const vs_ws_layout& L, float* lstm_out, hipStream_t stream, const Prep* prep, bool feat_rows = false);

}  // namespace

int vs_check_dims_impl(const vs_dims* d) { return check_dims(d); }

int vs_conv64_layer_impl(int math, const float* in, const float* w, void* packed, float* scales8 /* one scale slot */, int in_amax_ready,
                         const float* scale, const float* shift, float* out, int B, int T, int F, int KT, int KF,
                         int dil, int act, int transpose_flip, unsigned* amax_out, hipStream_t stream, double* bn_stats) {
  if (math != VS_MATH_FP32) {      // split-f16 or single-pass bf16: same operand plumbing (power-of-two scales, packed images)
    if (in_amax_ready) {
      if (int rc = vs_scale_from_absmax_impl(vs_amax_slot(scales8), VS_AMAX_SLOTS, scales8, stream)) return rc;
    } else {
      if (int rc = vs_pow2_scale_impl(in, (long long)B * 64 * T * F, vs_amax_slot(scales8), scales8, stream)) return rc;
    }
    if (int rc = vs_conv64_pack_f16_impl(w, static_cast<_Float16*>(packed), KT, KF, transpose_flip,
                                         reinterpret_cast<unsigned*>(scales8 + 4), scales8 + 2, stream, math)) return rc;
    return vs_conv64_f16x3_fwd_impl(in, static_cast<const _Float16*>(packed), scale, shift, scales8, scales8 + 2, out,
                                    B, T, F, KT, KF, dil, act, amax_out, stream, math, bn_stats);
  }
  VS_REQUIRE(bn_stats == nullptr, "conv64 layer: fused BatchNorm statistics are not offered by the fp32 kernels");
  if (int rc = vs_conv64_pack_impl(w, static_cast<float*>(packed), KT, KF, transpose_flip, stream)) return rc;
  return vs_conv64_fwd_impl(in, static_cast<const float*>(packed), scale, shift, out, B, T, F, KT, KF, dil, act, stream);
}

// scratch: any idle device buffer (the conv activation ping-pong in inference, a gradient buffer in
// training).  When it can hold both operands split into f16 hi/lo arrays (vs_gemm_presplit_bytes),
// the split is a pass of its own and the GEMM streams halves; otherwise (tiny batches: the split
// weights alone are 62 MB) the GEMM converts fp32 tiles while it stages them.
// the split-f16 / bf16 image of W_ih[:, :K] of both directions: scale2 (2 floats), then hi and lo halves [8H][Kp]
int vs_lstm_split_wih_impl(int math, const float* w_ih0, const float* w_ih1, int H, int K, int KE, unsigned* amax1,
                           float* w_scale2, _Float16* Wh, _Float16* Wl, hipStream_t stream) {
  const size_t Kp = (size_t)(K + VS_GEMM_KPAD - 1) / VS_GEMM_KPAD * VS_GEMM_KPAD;
  VS_CHECK_HIP(hipMemsetAsync(amax1, 0, sizeof(unsigned), stream));
  if (int rc = vs_absmax_accum_impl(w_ih0, (long long)4 * H * KE, amax1, stream)) return rc;
  if (int rc = vs_absmax_accum_impl(w_ih1, (long long)4 * H * KE, amax1, stream)) return rc;
  if (int rc = vs_scale_from_absmax_impl(amax1, 1, w_scale2, stream)) return rc;
  if (!Wh) return 0;
  if (int rc = vs_split_rows_impl(w_ih0, 4 * H, K, KE, w_scale2, Wh, Wl, 0, stream, math)) return rc;
  return vs_split_rows_impl(w_ih1, 4 * H, K, KE, w_scale2, Wh + (size_t)4 * H * Kp, Wl + (size_t)4 * H * Kp, 0, stream, math);
}

// whether the split-operand form of the LSTM input GEMM runs out of `scratch` (the condition vs_lstm_input_gemm_impl applies below):
// cnn8 may then write its output as that form's A operand (hi rows at scratch, lo rows at scratch + na)
bool vs_lstm_rows_fit(int M, int K, int H, const void* scratch, size_t scratch_bytes, bool prepared) {
  const size_t Kp = (size_t)(K + VS_GEMM_KPAD - 1) / VS_GEMM_KPAD * VS_GEMM_KPAD;
  const size_t na = ((size_t)M * Kp * 2 + 255) / 256 * 256, nw = ((size_t)8 * H * Kp * 2 + 255) / 256 * 256;
  if (!scratch || (reinterpret_cast<uintptr_t>(scratch) & 255) != 0) return false;
  if (prepared) return 2 * na <= scratch_bytes;
  return scratch_bytes >= vs_gemm_presplit_bytes(M, 8 * H, K) && 2 * na + 2 * nw <= scratch_bytes;
}

// prep_* != NULL: W_ih arrives prepared (vs_prepare_weights): its scale and split halves are read, not rebuilt
int vs_lstm_input_gemm_impl(int math, const float* feat, int K, const float* w_ih0, const float* w_ih1, int H, int KE,
                            float* xg, int M, const float* rowbias, int T, float* gs, void* scratch, size_t scratch_bytes,
                            hipStream_t stream, const float* prep_wscale2, const _Float16* prep_wh, const _Float16* prep_wl,
                            bool feat_bf16_ready, bool feat_rows_ready) {
  // the bf16 configuration's own GEMM (gemm_bf16.hip): feat and W_ih as bf16 arrays that the backward pass reuses.  It needs
  // room for the bf16 copy of feat, and for the bf16 W_ih unless that arrives prepared (vs_prepare_weights keeps it in the
  // prepared blob): a B = 1 clip of a second has room for the first but not for the 31 MB of the second.
  const VsLstmBf16Layout Lb16 = vs_lstm_bf16_layout(M, K, H);
  if (math == VS_MATH_BF16 && scratch && (reinterpret_cast<uintptr_t>(scratch) & 255) == 0 &&
      scratch_bytes >= (prep_wh ? Lb16.wih : Lb16.dxg)) {
    const VsLstmBf16Layout& Lb = Lb16;
    char* base = static_cast<char*>(scratch);
    if (!feat_bf16_ready) {      // (the training forward's BatchNorm apply of cnn8 writes it itself)
      if (int rc = vs_cvt_rows_bf16_impl(feat, M, K, K, base + Lb.feat, Lb.Kp, stream)) return rc;
    }
    if (!prep_wh) {      // (prepared weights: the bf16 W_ih lives in the prepared blob)
      if (int rc = vs_cvt_rows_bf16_impl(w_ih0, 4 * H, K, KE, static_cast<char*>(scratch) + Lb.wih, Lb.Kp, stream)) return rc;
      if (int rc = vs_cvt_rows_bf16_impl(w_ih1, 4 * H, K, KE, static_cast<char*>(scratch) + Lb.wih + (size_t)4 * H * Lb.Kp * 2, Lb.Kp, stream)) return rc;
    }
    const void* wih = prep_wh ? static_cast<const void*>(prep_wh) : static_cast<const void*>(static_cast<char*>(scratch) + Lb.wih);
    return vs_gemm_bf16_impl(0, 0, static_cast<char*>(scratch) + Lb.feat, Lb.Kp, wih, Lb.Kp, xg, 8 * H, nullptr, 0, M, 8 * H, K,
                             rowbias, 8 * H, T, 0, stream);
  }
  if (math == VS_MATH_BF16) {
    // no room for the bf16 operand copies: the split-operand GEMM below re-derives its operands from the fp32 tensors.  A
    // prepared blob of this arithmetic holds W_ih as bf16 bits (no f16 halves, no scale) -- never to be read as the split form.
    prep_wscale2 = nullptr;
    prep_wh = prep_wl = nullptr;
  }
  if (math != VS_MATH_FP32) {
    unsigned* amax = reinterpret_cast<unsigned*>(gs + 4);
    // feat_rows_ready: cnn8 wrote the split A operand and its scale (gs[0..1]) itself (conv_stack_impl, the whole-path eval forward)
    VS_REQUIRE(!feat_rows_ready || (math == VS_MATH_F16X3 && vs_lstm_rows_fit(M, K, H, scratch, scratch_bytes, prep_wscale2 != nullptr)),
               "lstm input gemm: the split feature rows were announced but do not fit");
    if (!feat_rows_ready) { if (int rc = vs_pow2_scale_impl(feat, (long long)M * K, amax, gs, stream)) return rc; }
    const size_t Kp = (size_t)(K + VS_GEMM_KPAD - 1) / VS_GEMM_KPAD * VS_GEMM_KPAD;
    const size_t na = ((size_t)M * Kp * 2 + 255) / 256 * 256, nw = ((size_t)8 * H * Kp * 2 + 255) / 256 * 256;
    const bool aligned = scratch && (reinterpret_cast<uintptr_t>(scratch) & 255) == 0;
    char* base = static_cast<char*>(scratch);
    _Float16* Ah = reinterpret_cast<_Float16*>(base);
    _Float16* Al = reinterpret_cast<_Float16*>(base + na);
    if (prep_wscale2) {
      if (aligned && 2 * na <= scratch_bytes) {
        if (!feat_rows_ready) { if (int rc = vs_split_rows_impl(feat, M, K, K, gs, Ah, Al, 0, stream, math)) return rc; }
        return vs_gemm_presplit_impl(Ah, Al, prep_wh, prep_wl, (int)Kp, xg, 8 * H, M, 8 * H, nullptr, nullptr, rowbias, 8 * H, T,
                                     VS_ACT_NONE, 0, gs, prep_wscale2, stream, math);
      }
      return vs_gemm_f16x3_impl(0, 0, feat, K, w_ih0, w_ih1, 4 * H, KE, xg, 8 * H, M, 8 * H, K, nullptr, nullptr, rowbias, 8 * H, T,
                                nullptr, 0, 0, 0, VS_ACT_NONE, 0, gs, prep_wscale2, stream, math);
    }
    const bool presplit = aligned && scratch_bytes >= vs_gemm_presplit_bytes(M, 8 * H, K) && 2 * na + 2 * nw <= scratch_bytes;
    _Float16* Wh = reinterpret_cast<_Float16*>(base + 2 * na);
    _Float16* Wl = reinterpret_cast<_Float16*>(base + 2 * na + nw);
    if (int rc = vs_lstm_split_wih_impl(math, w_ih0, w_ih1, H, K, KE, amax + 1, gs + 2, presplit ? Wh : nullptr, Wl, stream)) return rc;
    if (presplit) {
      if (!feat_rows_ready) { if (int rc = vs_split_rows_impl(feat, M, K, K, gs, Ah, Al, 0, stream, math)) return rc; }
      return vs_gemm_presplit_impl(Ah, Al, Wh, Wl, (int)Kp, xg, 8 * H, M, 8 * H, nullptr, nullptr, rowbias, 8 * H, T,
                                   VS_ACT_NONE, 0, gs, gs + 2, stream, math);
    }
    return vs_gemm_f16x3_impl(0, 0, feat, K, w_ih0, w_ih1, 4 * H, KE, xg, 8 * H, M, 8 * H, K, nullptr, nullptr, rowbias, 8 * H, T,
                              nullptr, 0, 0, 0, VS_ACT_NONE, 0, gs, gs + 2, stream, math);
  }
  return vs_gemm_nt2_impl(feat, K, w_ih0, w_ih1, 4 * H, KE, xg, 8 * H, M, 8 * H, K, nullptr, nullptr, rowbias, 8 * H, T, 0,
                          VS_ACT_NONE, stream);
}

extern "C" {

int vs_abi_version(void) { return VS_ABI_VERSION; }

int vs_set_option(int option, int value) {
  VS_REQUIRE(option >= 0 && option < VS_OPT_COUNT, "vs_set_option: unknown option %d", option);
  bool ok = true;
  switch (option) {
    case VS_OPT_FWD_PROLOGUE: case VS_OPT_HEAD_LEAF_SIDE: case VS_OPT_FEAT_ROWS: case VS_OPT_HEAD_BWD_GEMM: case VS_OPT_DETERMINISTIC:
      ok = value == 0 || value == 1; break;
    default: ok = value >= 0; break;
  }
  VS_REQUIRE(ok, "vs_set_option: value %d is outside the range of option %d", value, option);
  g_vs_options[option] = value;
  return 0;
}

int vs_get_option(int option) { return (option >= 0 && option < VS_OPT_COUNT) ? g_vs_options[option] : -1; }

int vs_profile_begin(int max_calls) {
  VS_REQUIRE(!g_prof.on, "profile: already enabled");
  VS_REQUIRE(max_calls > 0 && max_calls <= 4096, "profile: max_calls=%d out of range", max_calls);
  max_calls *= 8;   // slots entered once per layer (BatchNorm passes) use up to 8 entries per step
  const int n = max_calls * VS_PROF_SLOTS * 2;
  g_prof.ev = new hipEvent_t[n];
  for (int i = 0; i < n; ++i) VS_CHECK_HIP(hipEventCreate(&g_prof.ev[i]));
  for (int i = 0; i < VS_PROF_SLOTS; ++i) g_prof.calls[i] = 0;
  g_prof.max_calls = max_calls;
  g_prof.on = true;
  return 0;
}

int vs_profile_end(float* ms_total, int* calls) {
  VS_REQUIRE(g_prof.on, "profile: not enabled");
  g_prof.on = false;
  int rc = 0;
  for (int slot = 0; slot < VS_PROF_SLOTS; ++slot) {
    double tot = 0;
    for (int c = 0; c < g_prof.calls[slot]; ++c) {
      const int idx = (c * VS_PROF_SLOTS + slot) * 2;
      float ms = 0.f;
      hipError_t e = hipEventSynchronize(g_prof.ev[idx + 1]);
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_prof.ev[idx], g_prof.ev[idx + 1]);
      if (e != hipSuccess) { vs_set_error("profile: %s", hipGetErrorString(e)); rc = -2; }
      tot += ms;
    }
    if (ms_total) ms_total[slot] = (float)tot;
    if (calls) calls[slot] = g_prof.calls[slot];
  }
  const int n = g_prof.max_calls * VS_PROF_SLOTS * 2;
  for (int i = 0; i < n; ++i) (void)hipEventDestroy(g_prof.ev[i]);
  delete[] g_prof.ev;
  g_prof.ev = nullptr;
  g_prof.max_calls = 0;
  return rc;
}
const char* vs_last_error(void) { return g_err; }

int vs_workspace_layout(const vs_dims* dims, vs_ws_layout* out) {
  VS_REQUIRE(out != nullptr, "layout out pointer is NULL");
  memset(out, 0, sizeof(*out));
  return layout(dims, out);
}

size_t vs_workspace_bytes(const vs_dims* dims) {
  vs_ws_layout L;
  memset(&L, 0, sizeof(L));
  if (layout(dims, &L)) return 0;
  return L.total_bytes;
}

int vs_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, const float* conv_bias,
               float eps, int C, float* scale, float* shift, void* stream) {
  return vs_bn_fold_impl(gamma, beta, mean, var, conv_bias, eps, C, scale, shift, (hipStream_t)stream);
}

int vs_conv_first_fwd(const float* x, const float* w, const float* scale, const float* shift, float* out,
                      int B, int T, int F, int act, void* stream) {
  return vs_conv_first_fwd_impl(x, w, scale, shift, out, B, T, F, act, nullptr, (hipStream_t)stream);
}

int vs_conv64_pack(const float* w, float* packed, int KT, int KF, void* stream) {
  return vs_conv64_pack_impl(w, packed, KT, KF, 0, (hipStream_t)stream);
}

int vs_conv64_fwd(const float* in, const float* packed, const float* scale, const float* shift, float* out,
                  int B, int T, int F, int KT, int KF, int dil, int act, void* stream) {
  VS_REQUIRE(in != out, "conv64: in-place is not supported");
  return vs_conv64_fwd_impl(in, packed, scale, shift, out, B, T, F, KT, KF, dil, act, (hipStream_t)stream);
}

int vs_pow2_scale(const float* x, long long n, void* amax_scratch, float* scale2, void* stream) {
  VS_REQUIRE(x && amax_scratch && scale2, "pow2_scale: NULL argument");
  return vs_pow2_scale_impl(x, n, static_cast<unsigned*>(amax_scratch), scale2, (hipStream_t)stream);
}

int vs_conv64_pack_f16(const float* w, void* packed, int KT, int KF, int transpose_flip, void* amax_scratch,
                       float* w_scale2, void* stream) {
  VS_REQUIRE(w && packed && amax_scratch && w_scale2, "conv64_pack_f16: NULL argument");
  return vs_conv64_pack_f16_impl(w, static_cast<_Float16*>(packed), KT, KF, transpose_flip,
                                 static_cast<unsigned*>(amax_scratch), w_scale2, (hipStream_t)stream);
}

int vs_conv64_f16x3_fwd(const float* in, const void* packed, const float* scale, const float* shift,
                        const float* in_scale2, const float* w_scale2, float* out,
                        int B, int T, int F, int KT, int KF, int dil, int act, void* stream) {
  VS_REQUIRE(in != out, "conv64_f16x3: in-place is not supported");
  return vs_conv64_f16x3_fwd_impl(in, static_cast<const _Float16*>(packed), scale, shift, in_scale2, w_scale2, out,
                                  B, T, F, KT, KF, dil, act, nullptr, (hipStream_t)stream);
}

// ---- channels-last bf16 kernels of the VS_MATH_BF16 path (csrc/conv_nhwc.hip) -----------------------------
size_t vs_nhwc_conv_packed_bytes(int KT, int KF) { return vs_nhwc_packed_bytes(KT, KF); }

int vs_nhwc_conv_pack(const float* w, void* packed, int KT, int KF, int transpose_flip, void* stream) {
  return vs_nhwc_pack_impl(w, packed, KT, KF, transpose_flip, (hipStream_t)stream);
}

int vs_nhwc_conv(const void* in, const void* packed, const float* scale, const float* shift, void* out,
                 int B, int T, int F, int KT, int KF, int dil, int act, double* bn_stats, void* stream) {
  VS_REQUIRE(in != out, "nhwc_conv: in-place is not supported");
  return vs_nhwc_conv_impl(in, packed, scale, shift, out, B, T, F, KT, KF, dil, act, bn_stats, (hipStream_t)stream);
}

size_t vs_nhwc_conv_f16x3_scratch_bytes(int KT, int KF) { return vs_nhwc_f16x3_layer_scratch_bytes(KT, KF); }

int vs_nhwc_conv_f16x3_layer(const void* in_hi, const void* in_lo, const float* in_scale2, const unsigned* amax_in, int n_amax,
                             const float* w, const float* bn_scale, const float* bn_shift, void* scratch, int packed_ready,
                             void* out_hi, void* out_lo, float* out_scale2, unsigned* amax_out,
                             int B, int T, int F, int KT, int KF, int dil, int act, void* stream) {
  VS_REQUIRE(in_hi && in_lo && out_hi && out_lo && in_scale2 && amax_in && n_amax > 0 && bn_scale && bn_shift && out_scale2,
             "nhwc_conv_f16x3: NULL argument");
  VS_REQUIRE(in_hi != out_hi && in_lo != out_lo && in_hi != out_lo && in_lo != out_hi, "nhwc_conv_f16x3: in-place is not supported");
  VS_REQUIRE((KT == 5 && KF == 5) || (KT == 7 && KF == 1), "nhwc_conv_f16x3: kernel %dx%d is not one of the stack's (7x1, 5x5)", KT, KF);
  VS_REQUIRE(scratch != nullptr, "nhwc_conv_f16x3: NULL scratch");
  float* plan = reinterpret_cast<float*>(static_cast<char*>(scratch) + vs_nhwc_f16x3_wpart_bytes(KT, KF));
  return vs_nhwc_f16x3_layer_impl(in_hi, in_lo, in_scale2, amax_in, n_amax, w, bn_scale, bn_shift, scratch, packed_ready, plan, out_hi, out_lo,
                                  out_scale2, amax_out, B, T, F, KT, KF, dil, act, (hipStream_t)stream);
}

int vs_f16x3_split(const float* x, const float* scale2, void* hi, void* lo, long long n, void* stream) {
  return vs_f16x3_split_impl(x, scale2, hi, lo, n, (hipStream_t)stream);
}

int vs_f16x3_merge(const void* hi, const void* lo, const float* scale2, float* x, long long n, void* stream) {
  return vs_f16x3_merge_impl(hi, lo, scale2, x, n, (hipStream_t)stream);
}

int vs_cvt_rows_bf16(const float* src, long long rows, int K, int ld, void* dst, int Kp, void* stream) {
  return vs_cvt_rows_bf16_impl(src, rows, K, ld, dst, Kp, (hipStream_t)stream);
}

int vs_gemm_bf16(int a_kmajor, int b_kmajor, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                 const float* rowbias, int ldrb, int group, int accumulate, void* stream) {
  return vs_gemm_bf16_impl(a_kmajor, b_kmajor, A, lda, B, ldb, C, ldc, nullptr, 0, M, N, K, rowbias, ldrb, group, accumulate, (hipStream_t)stream);
}

int vs_gemm_bf16_split(int a_kmajor, int b_kmajor, const void* A, int lda, const void* B, int ldb, float* C, float* C2, int ldc, int split_m,
                       int M, int N, int K, int accumulate, void* stream) {
  VS_REQUIRE(C2 && split_m > 0 && split_m < M, "gemm_bf16_split: needs C2 and 0 < split_m < M (split_m %d, M %d)", split_m, M);
  return vs_gemm_bf16_impl(a_kmajor, b_kmajor, A, lda, B, ldb, C, ldc, C2, split_m, M, N, K, nullptr, 0, 1, accumulate, (hipStream_t)stream);
}

int vs_gemm_bf16_gated(int a_kmajor, int b_kmajor, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                       const float* gate, int ldg, void* stream) {
  VS_REQUIRE(gate != nullptr, "gemm_bf16_gated: gate is NULL");
  return vs_gemm_bf16_impl(a_kmajor, b_kmajor, A, lda, B, ldb, C, ldc, nullptr, 0, M, N, K, nullptr, 0, 1, 0, (hipStream_t)stream, gate, ldg);
}

int vs_nhwc_conv_first(const float* x, const float* w, const float* scale, const float* shift, void* out,
                       int B, int T, int F, int act, double* bn_stats, void* stream) {
  return vs_nhwc_conv_first_impl(x, w, scale, shift, out, B, T, F, act, bn_stats, (hipStream_t)stream);
}

int vs_nhwc_bn_apply(const void* z, void* a, long long npix, int act, const float* scale, const float* shift, void* stream) {
  return vs_nhwc_bn_apply_impl(z, a, npix, act, scale, shift, (hipStream_t)stream);
}

// cnn1 by recomputation (nhwc_edge.hip)
int vs_nhwc_first_moments(const float* x, int B, int T, int F, double* moments, void* stream) {
  return vs_nhwc_first_moments_impl(x, B, T, F, moments, (hipStream_t)stream);
}
int vs_nhwc_first_stats(const double* moments, const float* w, const float* bias, double count, double* stats, void* stream) {
  return vs_nhwc_first_stats_impl(moments, w, bias, count, stats, (hipStream_t)stream);
}
int vs_nhwc_first_bwd_scratch_doubles(void) { return VS_FIRST_BWD_SCRATCH_DOUBLES; }
int vs_nhwc_first_bwd(const void* da, const float* x, const float* w, const float* bias, int B, int T, int F, int act, int bn_mode,
                      const float* scale, const float* shift, const float* mean, const float* invstd,
                      float* dgamma, float* dbeta, float* dbias, float* dw, double* scratch, void* stream) {
  VS_REQUIRE(bn_mode == VS_BN_EVAL || bn_mode == VS_BN_TRAIN, "nhwc_first_bwd: unknown bn_mode %d", bn_mode);
  return vs_nhwc_first_bwd_impl(da, x, w, bias, B, T, F, act, bn_mode == VS_BN_TRAIN, scale, shift, mean, invstd, dgamma, dbeta, dbias, dw,
                                scratch, (hipStream_t)stream);
}

// train-mode BatchNorm2d between a conv that accumulated statistics and the apply pass (the piecewise form of what
// vs_forward_train does per layer)
int vs_bn_finalize(double* stats, int slots, double count, int C, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float eps, float momentum,
                   float* scale, float* shift, float* mean_out, float* invstd_out, void* stream) {
  VS_REQUIRE(eps >= 0.f && momentum >= 0.f && momentum <= 1.f, "bn_finalize: eps %g / momentum %g out of range", (double)eps, (double)momentum);
  VS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: give both running buffers or neither");
  return vs_bn_finalize_impl(stats, slots, count, C, gamma, beta, running_mean, running_var, eps, momentum, scale, shift, mean_out, invstd_out,
                             (hipStream_t)stream);
}

int vs_nhwc_conv_last(const void* in, const float* w, const float* scale, const float* shift, float* out,
                      int B, int T, int F, int act, void* stream) {
  return vs_nhwc_conv_last_impl(in, w, scale, shift, out, B, T, F, act, (hipStream_t)stream);
}

// cnn8 on the un-normalised output z7 of cnn7: BatchNorm + activation of cnn7 applied on the way into the matrix pipe
int vs_nhwc_conv_last_pre(const void* z7, const float* pre_scale, const float* pre_shift, int pre_act, const float* w,
                          const float* scale, const float* shift, float* out, double* bn_stats, int B, int T, int F, void* stream) {
  VS_REQUIRE(pre_scale && pre_shift, "nhwc_conv_last_pre: NULL BatchNorm constants");
  return vs_nhwc_conv_last_impl(z7, w, scale, shift, out, B, T, F, VS_ACT_NONE, (hipStream_t)stream, bn_stats, pre_scale, pre_shift, pre_act);
}

size_t vs_nhwc_conv_wgrad_partial_floats(int KT, int KF) { return vs_nhwc_wgrad_partial_floats(KT, KF); }

int vs_nhwc_conv_wgrad(const void* dz, const void* in, float* partials, float* dw, int B, int T, int F, int KT, int KF, int dil,
                       void* stream) {
  return vs_nhwc_wgrad_impl(dz, in, partials, dw, B, T, F, KT, KF, dil, (hipStream_t)stream);
}

int vs_nhwc_bn_act_bwd(const void* da, const void* z, void* dz, long long npix, int act, int bn_mode,
                       const float* scale, const float* shift, const float* mean, const float* invstd,
                       float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, void* stream) {
  return vs_nhwc_bn_act_bwd_impl(da, z, dz, npix, act, bn_mode == VS_BN_TRAIN, scale, shift, mean, invstd, dgamma, dbeta, dbias,
                                 stats, coef, (hipStream_t)stream);
}

int vs_nhwc_bn_act_bwd_first(const void* da, const void* z, const float* x, int B, int T, int F, int act, int bn_mode,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc, void* stream) {
  return vs_nhwc_bn_act_bwd_first_impl(da, z, x, B, T, F, act, bn_mode == VS_BN_TRAIN, scale, shift, mean, invstd, dgamma, dbeta, dbias,
                                       dw, stats, coef, acc, (hipStream_t)stream);
}

int vs_nhwc_conv_last_bwd_blocks(void) { return VS_NHWC_LAST_BWD_BLOCKS; }

int vs_nhwc_conv_last_bwd(const float* dz8, const float* w, const void* a7, void* din, float* partials, float* dw,
                          int B, int T, int F, void* stream) {
  return vs_nhwc_conv_last_bwd_impl(dz8, w, a7, din, partials, dw, B, T, F, nullptr, VS_ACT_NONE, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    (hipStream_t)stream);
}

// ---- the dy forms: the producer of a data gradient also does the first pass of the BatchNorm backward below it ----
int vs_nhwc_conv_dy(const void* dz, const void* packed, void* dy, const void* z, int act,
                    const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd, double* bn_stats,
                    int B, int T, int F, int KT, int KF, int dil, void* stream) {
  VS_REQUIRE(dz != dy, "nhwc_conv_dy: in-place is not supported");
  return vs_nhwc_conv_dy_impl(dz, packed, dy, z, act, bn_scale, bn_shift, bn_mean, bn_invstd, bn_stats, B, T, F, KT, KF, dil, (hipStream_t)stream);
}

int vs_nhwc_conv_last_bwd_dy(const float* dz8, const float* w, const void* a7, void* dy, float* partials, float* dw,
                             const void* z7, int act, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                             const float* bn_invstd, double* bn_stats, int B, int T, int F, void* stream) {
  VS_REQUIRE(z7, "nhwc_conv_last_bwd_dy: NULL z7");
  return vs_nhwc_conv_last_bwd_impl(dz8, w, a7, dy, partials, dw, B, T, F, z7, act, bn_scale, bn_shift, bn_mean, bn_invstd, bn_stats,
                                    (hipStream_t)stream);
}

int vs_nhwc_bn_bwd_from_dy(const void* dy, const void* z, void* dz, long long npix, int bn_mode,
                           const float* scale, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, void* stream) {
  return vs_nhwc_bn_bwd_from_dy_impl(dy, z, dz, npix, bn_mode == VS_BN_TRAIN, scale, mean, invstd, dgamma, dbeta, dbias, stats, coef,
                                     (hipStream_t)stream);
}

int vs_nhwc_bn_bwd_first_from_dy(const void* dy, const void* z, const float* x, int B, int T, int F, int bn_mode,
                                 const float* scale, const float* mean, const float* invstd,
                                 float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc, void* stream) {
  return vs_nhwc_bn_bwd_first_from_dy_impl(dy, z, x, B, T, F, bn_mode == VS_BN_TRAIN, scale, mean, invstd, dgamma, dbeta, dbias, dw,
                                           stats, coef, acc, (hipStream_t)stream);
}

int vs_conv_last_fwd(const float* in, const float* w, const float* scale, const float* shift, float* out,
                     int B, int T, int F, int act, void* stream) {
  return vs_conv_last_fwd_impl(in, w, scale, shift, out, B, T, F, act, (hipStream_t)stream);
}

int vs_gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
               const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
               int a_relu, int act, void* stream) {
  return vs_gemm_nt_impl(A, lda, W, ldw, C, ldc, M, N, K, bias1, bias2, rowbias, ldrb, group, a_relu, act, (hipStream_t)stream);
}

int vs_lstm_pack(const float* w_hh_fwd, const float* w_hh_bwd, float* packed, int H, void* stream) {
  return vs_lstm_pack_impl(w_hh_fwd, w_hh_bwd, packed, H, (hipStream_t)stream);
}

int vs_bilstm_recurrent(const float* xg, const float* packed_whh, float* state, float* out,
                        int B, int T, int H, void* stream) {
  return vs_bilstm_recurrent_impl(xg, packed_whh, state, out, nullptr, nullptr, B, T, H, (hipStream_t)stream);
}

// the same four calls with the arithmetic of the recurrent products chosen by the caller (what vs_forward* / vs_backward
// pass from dims.math): pack and recurrence must be given the same value
static int check_lstm_math(int math) {
  VS_REQUIRE(math == VS_MATH_FP32 || math == VS_MATH_F16X3 || math == VS_MATH_BF16, "lstm: unknown math %d", math);
  return 0;
}
int vs_lstm_pack_math(const float* w_hh_fwd, const float* w_hh_bwd, float* packed, int H, int math, void* stream) {
  if (int rc = check_lstm_math(math)) return rc;
  VS_REQUIRE(w_hh_fwd && w_hh_bwd && packed, "lstm_pack_math: NULL argument");
  return vs_lstm_pack_impl(w_hh_fwd, w_hh_bwd, packed, H, (hipStream_t)stream, math);
}
int vs_bilstm_recurrent_math(const float* xg, const float* packed_whh, float* state, float* out, float* gates_save, float* c_save,
                             int B, int T, int H, int math, void* stream) {
  if (int rc = check_lstm_math(math)) return rc;
  VS_REQUIRE(xg && packed_whh && state && out, "bilstm_recurrent_math: NULL argument");
  return vs_bilstm_recurrent_impl(xg, packed_whh, state, out, gates_save, c_save, B, T, H, (hipStream_t)stream, math);
}
int vs_lstm_pack_t_math(const float* w_hh_fwd, const float* w_hh_bwd, float* packed_t, int H, int math, void* stream) {
  if (int rc = check_lstm_math(math)) return rc;
  VS_REQUIRE(w_hh_fwd && w_hh_bwd && packed_t, "lstm_pack_t_math: NULL argument");
  return vs_lstm_pack_t_impl(w_hh_fwd, w_hh_bwd, packed_t, H, (hipStream_t)stream, math);
}
int vs_bilstm_recurrent_bwd_math(const float* packed_t, float* state, float* gates, const float* c_all, const float* dout,
                                 int B, int T, int H, int math, void* stream) {
  if (int rc = check_lstm_math(math)) return rc;
  VS_REQUIRE(packed_t && state && gates && c_all && dout, "bilstm_recurrent_bwd_math: NULL argument");
  return vs_bilstm_bwd_recurrent_impl(packed_t, state, gates, c_all, dout, B, T, H, (hipStream_t)stream, math);
}

// ---------------------------------------------------------------------------------------------
// stage 1: conv stack, models/voicesplit/model.py:68-74
// ---------------------------------------------------------------------------------------------
int vs_conv_stack_fwd(const vs_dims* d, const vs_params* p, const float* x, int conv_act, int bn_mode,
                      void* ws, size_t ws_bytes, float* feat, void* stream_) {
  vs_ws_layout L;
  if (int rc = check_ws(d, ws, ws_bytes, &L)) return rc;
  return conv_stack_impl(d, p, x, conv_act, bn_mode, ws, L, feat, (hipStream_t)stream_, nullptr);
}

}  // extern "C"

namespace {
int conv_stack_impl(const vs_dims* d, const vs_params* p, const float* x, int conv_act, int bn_mode,
                    void* ws, const vs_ws_layout& L, float* feat, hipStream_t stream, const Prep* prep, bool* feat_rows) {
  VS_REQUIRE(p && x, "conv_stack: NULL argument");
  if (feat_rows) *feat_rows = false;
  VS_REQUIRE(conv_act == VS_ACT_MISH || conv_act == VS_ACT_RELU, "conv_stack: conv_act must be MISH or RELU");
  VS_REQUIRE(bn_mode == VS_BN_EVAL || bn_mode == VS_BN_TRAIN, "conv_stack: unknown bn_mode %d", bn_mode);
  if (!feat) feat = at<float>(ws, L.feat);
  const int B = d->B, T = d->T, F = d->F;
  float* act[2] = {at<float>(ws, L.act0), at<float>(ws, L.act1)};
  float* scale = prep ? prep->bn_scale : at<float>(ws, L.bn_scale);
  float* shift = prep ? prep->bn_shift : at<float>(ws, L.bn_shift);
  double* stats = at<double>(ws, L.bn_stats);
  const bool train = bn_mode == VS_BN_TRAIN;
  VS_REQUIRE(!(prep && train), "conv_stack: prepared weights are an eval-mode form (BatchNorm folded)");

  for (int l = 0; l < 8; ++l) {
    const vs_conv_layer& c = p->conv[l];
    VS_REQUIRE(c.weight && c.bias && c.bn_weight && c.bn_bias && c.bn_running_mean && c.bn_running_var,
               "conv_stack: layer %d has a NULL parameter", l + 1);
  }
  const int Cl[8] = {64, 64, 64, 64, 64, 64, 64, 8};
  // Per-layer epilogue constants (slot l = scale/shift + 64*l).
  //  eval : BatchNorm folded from the running statistics, activation fused into the conv.
  //  train: the conv writes conv+bias (scale = 1, shift = bias, no activation); batch
  //         statistics, normalisation and activation follow as a second pass which then
  //         overwrites the layer's slot with the batch scale/shift.
  const int layer_act = train ? VS_ACT_NONE : conv_act;
  if (prep) {
    // folded by vs_prepare_weights
  } else if (!train) {
    for (int l = 0; l < 8; ++l) {
      const vs_conv_layer& c = p->conv[l];
      if (int rc = vs_bn_fold_impl(c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var, c.bias, kBnEps, Cl[l],
                                   scale + 64 * l, shift + 64 * l, stream)) return rc;
    }
  } else {
    VS_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(scale), 0x3f800000 /* 1.0f */, 8 * 64, stream));
    for (int l = 0; l < 8; ++l)
      VS_CHECK_HIP(hipMemcpyAsync(shift + 64 * l, p->conv[l].bias, sizeof(float) * Cl[l], hipMemcpyDeviceToDevice, stream));
  }

  if (d->math == VS_MATH_BF16) {
    // BASELINE configs[2]: channels-last bf16 activations [B][T][F][64] in the same ping-pong buffers (half their
    // size), conv_nhwc.hip / nhwc_edge.hip kernels.  eval: BatchNorm + activation in the conv epilogue; train:
    // z = conv + bias with statistics from the epilogue, then one apply pass in place.
    void* abuf[2] = {at<void>(ws, L.act0), at<void>(ws, L.act1)};
    const long long npix = (long long)B * T * F;
    int c = 0;
    auto bn_train = [&](int l, void* buf) -> int {
      const vs_conv_layer& cl = p->conv[l];
      if (int rc = vs_bn_finalize_impl(stats, VS_BN_STAT_SLOTS, (double)npix, 64, cl.bn_weight, cl.bn_bias, cl.bn_running_mean,
                                       cl.bn_running_var, kBnEps, kBnMomentum, scale + 64 * l, shift + 64 * l, nullptr, nullptr, stream)) return rc;
      return vs_nhwc_bn_apply_impl(buf, buf, npix, conv_act, scale + 64 * l, shift + 64 * l, stream);
    };
    {
      ProfScope ps(VS_PROF_CNN1, stream);
      if (train) {
        // cnn1 by recomputation (nhwc_edge.hip): statistics of z1 from the input's moments, then one pass that writes act(BN(z1))
        const vs_conv_layer& cl = p->conv[0];
        double* mom = stats + 128;                          // behind slot 0 of the statistics scratch
        if (int rc = vs_nhwc_first_moments_impl(x, B, T, F, mom, stream)) return rc;
        if (int rc = vs_nhwc_first_stats_impl(mom, cl.weight, cl.bias, (double)npix, stats, stream)) return rc;
        if (int rc = vs_bn_finalize_impl(stats, 1, (double)npix, 64, cl.bn_weight, cl.bn_bias, cl.bn_running_mean, cl.bn_running_var, kBnEps,
                                         kBnMomentum, scale, shift, nullptr, nullptr, stream)) return rc;
        if (int rc = vs_nhwc_conv_first_impl(x, cl.weight, scale, shift, abuf[c], B, T, F, conv_act, nullptr, stream, cl.bias)) return rc;
      } else if (int rc = vs_nhwc_conv_first_impl(x, p->conv[0].weight, scale, shift, abuf[c], B, T, F, layer_act, nullptr, stream)) return rc;
    }
    for (int i = 0; i < 6; ++i) {
      const int l = i + 1;
      ProfScope ps(VS_PROF_CNN2 + i, stream);
      void* packed = prep ? prep->conv_packed[i] : at<void>(ws, L.conv_packed[i]);
      if (!prep) { if (int rc = vs_nhwc_pack_impl(p->conv[l].weight, packed, kMid[i].kt, kMid[i].kf, 0, stream)) return rc; }
      if (train) VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * VS_BN_STAT_SLOTS * 128, stream));
      if (int rc = vs_nhwc_conv_impl(abuf[c], packed, scale + 64 * l, shift + 64 * l, abuf[c ^ 1], B, T, F, kMid[i].kt, kMid[i].kf,
                                     kMid[i].dil, layer_act, train ? stats : nullptr, stream)) return rc;
      c ^= 1;
      if (train) { if (int rc = bn_train(l, abuf[c])) return rc; }
    }
    ProfScope ps(VS_PROF_CNN8, stream);
    // the whole-path eval forward (feat_rows != NULL: nobody reads the fp32 features): cnn8 writes the bf16 A operand of the LSTM input
    // GEMM into the idle ping-pong buffer itself (six layers: abuf[c] is act0) -- no fp32 features, no conversion pass
    const size_t act_bytes = (size_t)B * 64 * T * F * sizeof(float);
    const VsLstmBf16Layout Lb = vs_lstm_bf16_layout((long long)B * T, 8 * F, d->H);
    if (!train && feat_rows && vs_opt(VS_OPT_FEAT_ROWS) != 0 && c == 0 && act_bytes >= (prep ? Lb.wih : Lb.dxg)) {
      *feat_rows = true;
      return vs_nhwc_conv_last_impl(abuf[c], p->conv[7].weight, scale + 64 * 7, shift + 64 * 7, nullptr, B, T, F, layer_act, stream, nullptr,
                                    nullptr, nullptr, VS_ACT_NONE, at<char>(ws, L.act1) + Lb.feat, Lb.Kp);
    }
    if (int rc = vs_nhwc_conv_last_impl(abuf[c], p->conv[7].weight, scale + 64 * 7, shift + 64 * 7, feat, B, T, F, layer_act, stream)) return rc;
    if (train) {
      if (int rc = vs_bn_train_feat_impl(feat, feat, B, T, F, p->conv[7].bn_weight, p->conv[7].bn_bias, p->conv[7].bn_running_mean,
                                         p->conv[7].bn_running_var, kBnEps, kBnMomentum, conv_act, stats,
                                         scale + 64 * 7, shift + 64 * 7, nullptr, nullptr, stream)) return rc;
    }
    return 0;
  }

  if (d->math == VS_MATH_F16X3 && !train) {
    // BASELINE configs[1]: activations as channels-last hi / lo f16 planes in the same ping-pong buffers; every layer writes its
    // output at a scale derived on the device from the tracked |max| of its input (conv_nhwc_f16x3.hip), no host round trip
    const size_t half = (size_t)B * T * F * 64 * 2;
    char* plane[2][2] = {{at<char>(ws, L.act0), at<char>(ws, L.act0) + half}, {at<char>(ws, L.act1), at<char>(ws, L.act1) + half}};
    float* cs = at<float>(ws, L.conv_scales);
    VS_CHECK_HIP(hipMemsetAsync(cs, 0, 8 * VS_SCALE_SLOT_FLOATS * sizeof(float), stream));
    auto slot = [&](int l) { return cs + VS_SCALE_SLOT_FLOATS * l; };          // [0..1]: scale pair of layer l's input; + 8: its |max|
    {
      ProfScope ps(VS_PROF_CNN1, stream);
      if (int rc = vs_absmax_any_impl(x, (long long)B * T * F, vs_amax_slot(slot(0)), stream)) return rc;
      if (int rc = vs_nhwc_first_plan_impl(vs_amax_slot(slot(0)), 1, p->conv[0].weight, scale, shift, slot(1), stream)) return rc;
      if (int rc = vs_nhwc_conv_first_split_impl(x, p->conv[0].weight, scale, shift, slot(1), plane[0][0], plane[0][1], vs_amax_slot(slot(1)),
                                                 B, T, F, layer_act, stream)) return rc;
    }
    // the whole-path forward (feat_rows != NULL: nobody reads the fp32 features): cnn8 writes the LSTM input GEMM's split A operand
    // into the idle ping-pong buffer, at a scale planned from the tracked |max| of its input -- no fp32 features, no |max| and split passes
    const size_t act_bytes = (size_t)B * 64 * T * F * sizeof(float);
    const int Kp = (8 * F + VS_GEMM_KPAD - 1) / VS_GEMM_KPAD * VS_GEMM_KPAD;
    const bool rows = feat_rows && vs_opt(VS_OPT_FEAT_ROWS) != 0 &&
                      vs_lstm_rows_fit(B * T, 8 * F, d->H, at<char>(ws, L.act1), act_bytes, prep != nullptr);
    int c = 0;
    for (int i = 0; i < 6; ++i) {
      const int l = i + 1;
      ProfScope ps(VS_PROF_CNN2 + i, stream);
      char* mine = at<char>(ws, L.conv_packed[i]);
      void* wpart = prep ? prep->conv_packed[i] : mine;
      float* plan = reinterpret_cast<float*>(mine + vs_nhwc_f16x3_wpart_bytes(kMid[i].kt, kMid[i].kf));
      if (int rc = vs_nhwc_f16x3_layer_impl(plane[c][0], plane[c][1], slot(l), vs_amax_slot(slot(l)), VS_AMAX_SLOTS, p->conv[l].weight,
                                            scale + 64 * l, shift + 64 * l, wpart, prep ? 1 : 0, plan, plane[c ^ 1][0], plane[c ^ 1][1],
                                            slot(l + 1), (l < 6 || rows) ? vs_amax_slot(slot(l + 1)) : nullptr, B, T, F, kMid[i].kt, kMid[i].kf,
                                            kMid[i].dil, layer_act, stream)) return rc;
      c ^= 1;
    }
    ProfScope ps(VS_PROF_CNN8, stream);
    if (rows) {      // six layers: c == 0, the input planes fill act0 and act1 is idle
      float* gs = at<float>(ws, L.gemm_scales);
      char* rows_hi = at<char>(ws, L.act1);
      const size_t na = ((size_t)B * T * Kp * 2 + 255) / 256 * 256;
      if (int rc = vs_nhwc_last_plan_impl(vs_amax_slot(slot(7)), VS_AMAX_SLOTS, p->conv[7].weight, scale + 64 * 7, shift + 64 * 7, gs, stream)) return rc;
      *feat_rows = true;
      return vs_nhwc_conv_last_split_impl(plane[c][0], plane[c][1], slot(7), p->conv[7].weight, scale + 64 * 7, shift + 64 * 7, nullptr,
                                          B, T, F, layer_act, stream, rows_hi, rows_hi + na, Kp, gs);
    }
    return vs_nhwc_conv_last_split_impl(plane[c][0], plane[c][1], slot(7), p->conv[7].weight, scale + 64 * 7, shift + 64 * 7, feat,
                                        B, T, F, layer_act, stream);
  }

  int cur = 0;
  // split-f16 convs: every producer of a conv operand folds its |max| into the consumer's slot
  float* cs = at<float>(ws, L.conv_scales);
  const bool f16 = d->math != VS_MATH_FP32;
  if (f16) VS_CHECK_HIP(hipMemsetAsync(cs, 0, 8 * VS_SCALE_SLOT_FLOATS * sizeof(float), stream));
  auto amax_for = [&](int consumer_layer) -> unsigned* {    // consumer_layer = conv index 1..6 (cnn2..cnn7)
    return (f16 && consumer_layer >= 1 && consumer_layer <= 6) ? vs_amax_slot(cs + VS_SCALE_SLOT_FLOATS * consumer_layer) : nullptr;
  };
  // cnn1
  {
  ProfScope ps(VS_PROF_CNN1, stream);
  if (int rc = vs_conv_first_fwd_impl(x, p->conv[0].weight, scale, shift, act[cur], B, T, F, layer_act,
                                      train ? nullptr : amax_for(1), stream)) return rc;
  if (train) {
    if (int rc = vs_bn_train_impl(act[cur], act[cur], B, 64, T * F, p->conv[0].bn_weight, p->conv[0].bn_bias, p->conv[0].bn_running_mean,
                                  p->conv[0].bn_running_var, kBnEps, kBnMomentum, conv_act, stats, scale, shift, nullptr, nullptr,
                                  amax_for(1), stream)) return rc;
  }
  }
  // cnn2..cnn7
  for (int i = 0; i < 6; ++i) {
    const int l = i + 1;
    float* packed = at<float>(ws, L.conv_packed[i]);
    ProfScope ps(VS_PROF_CNN2 + i, stream);
    const bool fuse = train && f16;        // statistics of this layer accumulated by the conv epilogue
    if (fuse) VS_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * VS_BN_STAT_SLOTS * 128, stream));
    if (prep) {          // (strict fp32 arithmetic: the other two have taken their channels-last eval routes above) weights packed once
      if (int rc = vs_conv64_fwd_impl(act[cur], static_cast<const float*>(prep->conv_packed[i]), scale + 64 * l, shift + 64 * l,
                                      act[cur ^ 1], B, T, F, kMid[i].kt, kMid[i].kf, kMid[i].dil, layer_act, stream)) return rc;
    } else
    if (int rc = vs_conv64_layer_impl(d->math, act[cur], p->conv[l].weight, packed, cs + VS_SCALE_SLOT_FLOATS * l, 1,
                                      scale + 64 * l, shift + 64 * l, act[cur ^ 1], B, T, F,
                                      kMid[i].kt, kMid[i].kf, kMid[i].dil, layer_act, 0, train ? nullptr : amax_for(l + 1), stream,
                                      fuse ? stats : nullptr)) return rc;
    cur ^= 1;
    if (train) {
      if (int rc = vs_bn_train_impl(act[cur], act[cur], B, 64, T * F, p->conv[l].bn_weight, p->conv[l].bn_bias, p->conv[l].bn_running_mean,
                                    p->conv[l].bn_running_var, kBnEps, kBnMomentum, conv_act, stats,
                                    scale + 64 * l, shift + 64 * l, nullptr, nullptr, amax_for(l + 1), stream,
                                    fuse ? VS_BN_STAT_SLOTS : 0)) return rc;
    }
  }
  // cnn8, written straight into the LSTM feature layout
  ProfScope ps(VS_PROF_CNN8, stream);
  if (int rc = vs_conv_last_fwd_impl(act[cur], p->conv[7].weight, scale + 64 * 7, shift + 64 * 7, feat, B, T, F, layer_act, stream)) return rc;
  if (train) {
    if (int rc = vs_bn_train_feat_impl(feat, feat, B, T, F, p->conv[7].bn_weight, p->conv[7].bn_bias, p->conv[7].bn_running_mean,
                                       p->conv[7].bn_running_var, kBnEps, kBnMomentum, conv_act, stats,
                                       scale + 64 * 7, shift + 64 * 7, nullptr, nullptr, stream)) return rc;
  }
  return 0;
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// stage 2: d-vector concat + BiLSTM, models/voicesplit/model.py:77-82
// ---------------------------------------------------------------------------------------------
int vs_bilstm_fwd(const vs_dims* d, const vs_params* p, const float* feat, const float* dvec,
                  void* ws, size_t ws_bytes, float* lstm_out, void* stream_) {
  vs_ws_layout L;
  if (int rc = check_ws(d, ws, ws_bytes, &L)) return rc;
  return bilstm_impl(d, p, feat, dvec, ws, L, lstm_out, (hipStream_t)stream_, nullptr);
}

}  // extern "C"

namespace {
int bilstm_impl(const vs_dims* d, const vs_params* p, const float* feat, const float* dvec,
                void* ws, const vs_ws_layout& L, float* lstm_out, hipStream_t stream, const Prep* prep, bool feat_rows) {
  VS_REQUIRE(p && dvec, "bilstm: NULL argument");
  if (!feat) feat = at<float>(ws, L.feat);
  if (!lstm_out) lstm_out = at<float>(ws, L.lstm_out);
  const int B = d->B, T = d->T, H = d->H, K = 8 * d->F, KE = K + d->E;
  float* dvbias = at<float>(ws, L.dvbias);
  float* xg = at<float>(ws, L.xg);
  {
  ProfScope ps(VS_PROF_LSTM_GEMM, stream);
  for (int dir = 0; dir < 2; ++dir) {
    VS_REQUIRE(p->w_ih[dir] && p->w_hh[dir] && p->b_ih[dir] && p->b_hh[dir], "bilstm: NULL LSTM parameter (dir %d)", dir);
    // cat((x, dvec.repeat(T))) @ W_ih^T == x @ W_ih[:, :8F]^T + (dvec @ W_ih[:, 8F:]^T): the
    // second term does not depend on t -> one [B][4H] row bias per utterance (+ b_ih + b_hh).
    if (int rc = vs_gemm_nt_impl(dvec, d->E, p->w_ih[dir] + K, KE, dvbias + (size_t)dir * 4 * H, 8 * H, B, 4 * H, d->E,
                                 p->b_ih[dir], p->b_hh[dir], nullptr, 0, 1, 0, VS_ACT_NONE, stream)) return rc;
  }
  // both directions in one launch (N = 8H): twice the workgroups, half the tail quantisation
  // the conv stack is done: its activation ping-pong is idle (feat may be the caller's own buffer)
  const size_t act_bytes = (size_t)B * 64 * T * d->F * sizeof(float);
  // feat_rows: cnn8 left the split A operand in the second ping-pong buffer (conv_stack_impl)
  if (int rc = vs_lstm_input_gemm_impl(d->math, feat, K, p->w_ih[0], p->w_ih[1], H, KE, xg, B * T, dvbias, T,
                                       at<float>(ws, L.gemm_scales), at<char>(ws, feat_rows ? L.act1 : L.act0),
                                       feat_rows ? act_bytes : (L.act1 == L.act0 + act_bytes ? 2 * act_bytes : act_bytes), stream,
                                       prep ? prep->gemm_wscale : nullptr, prep ? prep->wih_hi : nullptr,
                                       prep ? prep->wih_lo : nullptr, feat_rows && d->math == VS_MATH_BF16,
                                       feat_rows && d->math == VS_MATH_F16X3)) return rc;
  }
  float* packed = prep ? prep->lstm_packed : at<float>(ws, L.lstm_packed);
  if (!prep) { if (int rc = vs_lstm_pack_impl(p->w_hh[0], p->w_hh[1], packed, H, stream, d->math)) return rc; }
  ProfScope ps(VS_PROF_LSTM_REC, stream);
  return vs_bilstm_recurrent_impl(xg, packed, at<float>(ws, L.lstm_state), lstm_out, nullptr, nullptr, B, T, H, stream, d->math);
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// stage 3: head, models/voicesplit/model.py:83-87
// ---------------------------------------------------------------------------------------------
static int head_fwd_impl(const vs_dims* d, const vs_params* p, const float* lstm_out, void* ws, size_t ws_bytes,
                         float* logits, float* mask, hipStream_t stream, const void* head_packed) {
  vs_ws_layout L;
  if (int rc = check_ws(d, ws, ws_bytes, &L)) return rc;
  VS_REQUIRE(p && p->fc1_w && p->fc1_b && p->fc2_w && p->fc2_b, "head: NULL parameter");
  VS_REQUIRE(mask || logits, "head: no output requested");
  if (!lstm_out) lstm_out = at<float>(ws, L.lstm_out);
  float* h1 = at<float>(ws, L.fc1_out);
  const int M = d->B * d->T;
  ProfScope ps(VS_PROF_HEAD, stream);
  if (d->math == VS_MATH_BF16 && vs_head_fused_supported(2 * d->H, d->FC1, d->FC2)) {
    // one launch, h1 in registers between the two contractions (head_fused.hip).  The weights' fragment images come prepared
    // (vs_prepare_weights) or are packed here into the conv stack's first activation buffer, idle by now (stream order) -- the
    // same images either way, so the two routes stay bit-identical.  A clip of a frame or two at full width cannot hold them:
    // the two-launch form below (same roundings, fp32 summation order differs)
    const size_t need = vs_head_fused_packed_bytes(2 * d->H, d->FC1, d->FC2);
    const void* img = head_packed;
    if (!img && L.act1 - L.act0 >= need) {
      void* scratch = at<void>(ws, L.act0);
      if (int rc = vs_head_fused_pack_impl(p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, 2 * d->H, d->FC1, d->FC2, scratch, stream)) return rc;
      img = scratch;
    }
    if (img) return vs_head_fused_impl(lstm_out, img, nullptr, logits, mask, M, 2 * d->H, d->FC1, d->FC2, stream);
  }
  // VS_MATH_BF16: bf16-rounded operands on the bf16 matrix instruction, fp32 accumulate and epilogue
  const auto vs_gemm_nt_impl = d->math == VS_MATH_BF16 ? ::vs_gemm_nt_bf16_impl : ::vs_gemm_nt_impl;
  // relu(lstm) -> fc1 -> relu
  if (int rc = vs_gemm_nt_impl(lstm_out, 2 * d->H, p->fc1_w, 2 * d->H, h1, d->FC1, M, d->FC1, 2 * d->H,
                               p->fc1_b, nullptr, nullptr, 0, 1, 1, VS_ACT_RELU, stream)) return rc;
  // fc2 -> sigmoid
  if (logits) {
    if (int rc = vs_gemm_nt_impl(h1, d->FC1, p->fc2_w, d->FC1, logits, d->FC2, M, d->FC2, d->FC1,
                                 p->fc2_b, nullptr, nullptr, 0, 1, 0, VS_ACT_NONE, stream)) return rc;
  }
  if (mask) {
    if (int rc = vs_gemm_nt_impl(h1, d->FC1, p->fc2_w, d->FC1, mask, d->FC2, M, d->FC2, d->FC1,
                                 p->fc2_b, nullptr, nullptr, 0, 1, 0, VS_ACT_SIGMOID, stream)) return rc;
  }
  return 0;
}

int vs_head_fwd(const vs_dims* d, const vs_params* p, const float* lstm_out, void* ws, size_t ws_bytes,
                float* logits, float* mask, void* stream_) {
  return head_fwd_impl(d, p, lstm_out, ws, ws_bytes, logits, mask, (hipStream_t)stream_, nullptr);
}

int vs_forward(const vs_dims* d, const vs_params* p, const float* x, const float* dvec, int conv_act, int bn_mode,
               void* ws, size_t ws_bytes, float* mask, void* stream) {
  VS_REQUIRE(mask != nullptr, "forward: mask is NULL");
  vs_ws_layout L;
  if (int rc = check_ws(d, ws, ws_bytes, &L)) return rc;
  bool feat_rows = false;
  if (int rc = conv_stack_impl(d, p, x, conv_act, bn_mode, ws, L, nullptr, (hipStream_t)stream, nullptr, &feat_rows)) return rc;
  if (int rc = bilstm_impl(d, p, nullptr, dvec, ws, L, nullptr, (hipStream_t)stream, nullptr, feat_rows)) return rc;
  return vs_head_fwd(d, p, nullptr, ws, ws_bytes, nullptr, mask, stream);
}

// ---------------------------------------------------------------------------------------------
// eval-mode forward with the weight-only work done once (validation / serving: weights do not change
// between calls; utils/generic_utils.py:476-558 runs the model sample by sample at B = 1)
// ---------------------------------------------------------------------------------------------
size_t vs_prepared_bytes(const vs_dims* dims) {
  PrepLayout L;
  memset(&L, 0, sizeof(L));
  if (prep_layout(dims, &L)) return 0;
  return L.total_bytes;
}

int vs_prepare_weights(const vs_dims* d, const vs_params* p, void* prepared, size_t prepared_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  Prep P;
  if (int rc = prep_pointers(d, prepared, prepared_bytes, &P)) return rc;
  VS_REQUIRE(p != nullptr, "prepare_weights: params is NULL");
  const int Cl[8] = {64, 64, 64, 64, 64, 64, 64, 8};
  for (int l = 0; l < 8; ++l) {
    const vs_conv_layer& c = p->conv[l];
    VS_REQUIRE(c.weight && c.bias && c.bn_weight && c.bn_bias && c.bn_running_mean && c.bn_running_var,
               "prepare_weights: layer %d has a NULL parameter", l + 1);
    if (int rc = vs_bn_fold_impl(c.bn_weight, c.bn_bias, c.bn_running_mean, c.bn_running_var, c.bias, kBnEps, Cl[l],
                                 P.bn_scale + 64 * l, P.bn_shift + 64 * l, stream)) return rc;
  }
  for (int i = 0; i < 6; ++i) {
    const float* w = p->conv[i + 1].weight;
    if (d->math == VS_MATH_BF16) {
      if (int rc = vs_nhwc_pack_impl(w, P.conv_packed[i], kMid[i].kt, kMid[i].kf, 0, stream)) return rc;
    } else if (d->math == VS_MATH_F16X3) {
      if (int rc = vs_nhwc_f16x3_prepare_wpart_impl(w, P.conv_packed[i], kMid[i].kt, kMid[i].kf, stream)) return rc;
    } else {
      if (int rc = vs_conv64_pack_impl(w, static_cast<float*>(P.conv_packed[i]), kMid[i].kt, kMid[i].kf, 0, stream)) return rc;
    }
  }
  for (int dir = 0; dir < 2; ++dir)
    VS_REQUIRE(p->w_ih[dir] && p->w_hh[dir], "prepare_weights: NULL LSTM parameter (dir %d)", dir);
  if (d->math == VS_MATH_BF16) {      // [8H][Kp] bf16, both directions stacked: the B operand of gemm_bf16.hip
    const int K = 8 * d->F, KE = K + d->E, Kp = (K + 63) / 64 * 64;
    if (int rc = vs_cvt_rows_bf16_impl(p->w_ih[0], 4 * d->H, K, KE, P.wih_hi, Kp, stream)) return rc;
    if (int rc = vs_cvt_rows_bf16_impl(p->w_ih[1], 4 * d->H, K, KE, P.wih_hi + (size_t)4 * d->H * Kp, Kp, stream)) return rc;
  } else if (d->math != VS_MATH_FP32) {
    if (int rc = vs_lstm_split_wih_impl(d->math, p->w_ih[0], p->w_ih[1], d->H, 8 * d->F, 8 * d->F + d->E,
                                        reinterpret_cast<unsigned*>(P.gemm_wscale + 4), P.gemm_wscale, P.wih_hi, P.wih_lo, stream)) return rc;
  }
  if (P.head_packed) {
    VS_REQUIRE(p->fc1_w && p->fc1_b && p->fc2_w && p->fc2_b, "prepare_weights: NULL head parameter");
    if (int rc = vs_head_fused_pack_impl(p->fc1_w, p->fc1_b, p->fc2_w, p->fc2_b, 2 * d->H, d->FC1, d->FC2, P.head_packed, stream)) return rc;
  }
  return vs_lstm_pack_impl(p->w_hh[0], p->w_hh[1], P.lstm_packed, d->H, stream, d->math);
}

int vs_forward_prepared(const vs_dims* d, const vs_params* p, const void* prepared, size_t prepared_bytes,
                        const float* x, const float* dvec, int conv_act, void* ws, size_t ws_bytes, float* mask, void* stream) {
  VS_REQUIRE(mask != nullptr, "forward_prepared: mask is NULL");
  Prep P;
  if (int rc = prep_pointers(d, prepared, prepared_bytes, &P)) return rc;
  vs_ws_layout L;
  if (int rc = check_ws(d, ws, ws_bytes, &L)) return rc;
  bool feat_rows = false;
  if (int rc = conv_stack_impl(d, p, x, conv_act, VS_BN_EVAL, ws, L, nullptr, (hipStream_t)stream, &P, &feat_rows)) return rc;
  if (int rc = bilstm_impl(d, p, nullptr, dvec, ws, L, nullptr, (hipStream_t)stream, &P, feat_rows)) return rc;
  return head_fwd_impl(d, p, nullptr, ws, ws_bytes, nullptr, mask, (hipStream_t)stream, P.head_packed);
}

}  // extern "C"
