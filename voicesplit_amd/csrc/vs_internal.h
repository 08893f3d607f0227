// Prototypes of the per-kernel host launchers shared by capi.hip (inference orchestration) and
// train.hip (training forward + backward orchestration).  Internal to libvoicesplit_hip.so.
#pragma once
#include "vs_common.h"

// capi.hip: opt-in per-stage timing scope (see vs_profile_begin in the public header)
struct VsProfScope {
  hipStream_t s; int idx;
  VsProfScope(int slot, hipStream_t s);
  ~VsProfScope();
};

// capi.hip: the option table behind vs_set_option / vs_get_option (the library reads no environment variable)
extern int g_vs_options[VS_OPT_COUNT];
inline int vs_opt(int o) { return g_vs_options[o]; }
// Deterministic mode: the turn word of the tape the calling thread is working on (vs_forward_train / vs_backward set it for their
// duration when VS_OPT_DETERMINISTIC is on; NULL otherwise -- the kernel-level entry points never order their atomics)
extern thread_local unsigned* g_vs_turn;
struct VsTurnScope {
  unsigned* prev;
  explicit VsTurnScope(unsigned* t) : prev(g_vs_turn) { g_vs_turn = t; }
  ~VsTurnScope() { g_vs_turn = prev; }
};
inline int vs_det_grid(int nb, int cus = 256) { return (g_vs_turn && nb > cus) ? cus : nb; }      // deterministic mode: at most one workgroup per CU takes turns

// conv_mfma.hip
int vs_conv64_pack_impl(const float* w, float* wp, int KT, int KF, int transpose_flip, hipStream_t);
int vs_conv64_fwd_impl(const float* in, const float* wp, const float* scale, const float* shift, float* out,
                       int B, int T, int F, int KT, int KF, int dil, int act, hipStream_t);
// conv_f16x3.hip
int vs_pow2_scale_impl(const float* x, long long n, unsigned* amax_scratch, float* scale2, hipStream_t);
int vs_scale_from_absmax_impl(const unsigned* amax, int n, float* scale2, hipStream_t);
int vs_absmax_accum_impl(const float* x, long long n, unsigned* amax, hipStream_t);
// gemm_f16x3.hip
// gemm_f16x3.hip: operands split into f16 hi/lo arrays by a pass of their own
size_t vs_gemm_presplit_bytes(int M, int N, int K);
int vs_split_rows_impl(const float* x, int rows, int K, int ld, const float* scale2, _Float16* hi, _Float16* lo, int relu, hipStream_t,
                       int math = VS_MATH_CODE_F16X3);
int vs_gemm_presplit_impl(const _Float16* Ah, const _Float16* Al, const _Float16* Wh, const _Float16* Wl, int Kp,
                          float* C, int ldc, int M, int N, const float* bias1, const float* bias2,
                          const float* rowbias, int ldrb, int group, int act, int accumulate,
                          const float* a_scale2, const float* w_scale2, hipStream_t, int math = VS_MATH_CODE_F16X3);
int vs_gemm_f16x3_impl(int layout_a, int layout_w, const float* A, int lda, const float* W, const float* W_hi,
                       int n_split, int ldw, float* C, int ldc, int M, int N, int K,
                       const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                       const float* gate, int ldg, int a_relu, int w_relu, int act, int accumulate,
                       const float* a_scale2, const float* w_scale2, hipStream_t, int math = VS_MATH_CODE_F16X3);
// The LSTM input projection x @ [W_ih; W_ih_reverse]^T (+ per-utterance row bias) in either
// arithmetic.  gemm_scales: 16 floats of scratch that must survive until the backward pass in
// training: [0..1] scale of feat, [2..3] scale of the two W_ih, [4..5] uint |max| scratch.
int vs_lstm_input_gemm_impl(int math, const float* feat, int K, const float* w_ih0, const float* w_ih1, int H, int KE,
                            float* xg, int M, const float* rowbias, int T, float* gemm_scales,
                            void* scratch, size_t scratch_bytes, hipStream_t, const float* prep_wscale2 = nullptr,
                            const _Float16* prep_wh = nullptr, const _Float16* prep_wl = nullptr,
                            bool feat_bf16_ready = false /* VS_MATH_BF16: the bf16 copy of feat is already in scratch */,
                            bool feat_rows_ready = false /* VS_MATH_F16X3: cnn8 wrote the split A operand + its scale (vs_lstm_rows_fit) */);
bool vs_lstm_rows_fit(int M, int K, int H, const void* scratch, size_t scratch_bytes, bool prepared);
int vs_lstm_split_wih_impl(int math, const float* w_ih0, const float* w_ih1, int H, int K, int KE, unsigned* amax1,
                           float* w_scale2, _Float16* Wh, _Float16* Wl, hipStream_t);
int vs_conv64_pack_f16_impl(const float* w, _Float16* wp, int KT, int KF, int transpose_flip, unsigned* amax_scratch,
                            float* w_scale2, hipStream_t, int math = VS_MATH_CODE_F16X3);
int vs_conv64_f16x3_fwd_impl(const float* in, const _Float16* wp, const float* scale, const float* shift,
                             const float* in_scale2, const float* w_scale2, float* out,
                             int B, int T, int F, int KT, int KF, int dil, int act, unsigned* amax_out, hipStream_t,
                             int math = VS_MATH_CODE_F16X3, double* bn_stats = nullptr);
// conv_f16x3_pk.hip: persistent, software-pipelined form of the 5x5 kernel (same contract, bit-identical results)
int vs_conv64_f16x3_pk_impl(const float* in, const _Float16* wp, const float* scale, const float* shift,
                            const float* in_scale2, const float* w_scale2, float* out,
                            int B, int T, int F, int dil, int act, unsigned* amax_out, hipStream_t, int ablation = 0,
                            int i_end = 0x7fffffff, int math = VS_MATH_CODE_F16X3, double* bn_stats = nullptr);
// One 64->64 conv launch in either arithmetic: packs the weights (transpose_flip for the data
// gradient) into `packed`.  Split-f16 mode keeps its operand scales in one "scale slot" of
// VS_SCALE_SLOT_FLOATS floats: [0..1] input {s, 1/s}, [2..3] weight {s, 1/s}, [4] uint |max| of
// the weights, [8 .. 8+VS_AMAX_SLOTS) uint running |max| array of the input.
// in_amax_ready: the producer of `in` already folded its |max| into the slot's array
// (vs_absmax_commit); otherwise one extra pass over `in` computes it.  amax_out: the |max| array
// (of the consumer's slot) this launch folds its own output into, or NULL.
#define VS_SCALE_SLOT_FLOATS (8 + VS_AMAX_SLOTS)
int vs_conv64_layer_impl(int math, const float* in, const float* w, void* packed, float* slot, int in_amax_ready,
                         const float* scale, const float* shift, float* out, int B, int T, int F, int KT, int KF,
                         int dil, int act, int transpose_flip, unsigned* amax_out, hipStream_t,
                         double* bn_stats = nullptr);   // split-f16 / bf16 only: fused train-mode BatchNorm statistics of `out`
inline unsigned* vs_amax_slot(float* slot) { return reinterpret_cast<unsigned*>(slot + 8); }
// conv_nhwc.hip: channels-last bf16 64->64 convs (VS_MATH_BF16)
size_t vs_nhwc_packed_bytes(int KT, int KF);
int vs_nhwc_pack_impl(const float* w, void* packed, int KT, int KF, int transpose_flip, hipStream_t);
int vs_nhwc_conv_impl(const void* in, const void* packed, const float* scale, const float* shift, void* out,
                      int B, int T, int F, int KT, int KF, int dil, int act, double* bn_stats, hipStream_t);
// conv_nhwc_f16x3.hip: the same convs, forward, in the fp32-class split-f16 arithmetic on channels-last hi / lo f16 planes
size_t vs_nhwc_f16x3_packed_bytes(int KT, int KF);
int vs_nhwc_f16x3_pack_impl(const float* w, const float* w_scale2, void* packed, float* l1, int KT, int KF, hipStream_t);
int vs_nhwc_f16x3_plan_impl(const float* in_scale2, const float* w_scale2, const unsigned* amax_in, int n_amax, const float* bn_scale,
                            const float* bn_shift, const float* l1, float* eff_scale, float* eff_shift, float* out_scale2, hipStream_t);
int vs_nhwc_conv_f16x3_impl(const void* in_hi, const void* in_lo, const void* packed, const float* scale, const float* shift,
                            const float* out_scale2, void* out_hi, void* out_lo, unsigned* amax_out,
                            int B, int T, int F, int KT, int KF, int dil, int act, hipStream_t);
int vs_f16x3_split_impl(const float* x, const float* scale2, void* hi, void* lo, long long n, hipStream_t);
int vs_f16x3_merge_impl(const void* hi, const void* lo, const float* scale2, float* x, long long n, hipStream_t);
size_t vs_nhwc_f16x3_layer_scratch_bytes(int KT, int KF);
size_t vs_nhwc_f16x3_wpart_bytes(int KT, int KF);
int vs_nhwc_f16x3_prepare_wpart_impl(const float* w, void* wpart, int KT, int KF, hipStream_t);
int vs_nhwc_f16x3_layer_impl(const void* in_hi, const void* in_lo, const float* in_scale2, const unsigned* amax_in, int n_amax,
                             const float* w, const float* bn_scale, const float* bn_shift, void* wpart, int packed_ready, float* plan,
                             void* out_hi, void* out_lo, float* out_scale2, unsigned* amax_out,
                             int B, int T, int F, int KT, int KF, int dil, int act, hipStream_t);
// nhwc_edge.hip: cnn1 / cnn8 on the same planes
int vs_absmax_any_impl(const float* x, long long n, unsigned* amax, hipStream_t);
int vs_nhwc_first_plan_impl(const unsigned* amax_in, int n_amax, const float* w, const float* scale, const float* shift, float* out_scale2, hipStream_t);
int vs_nhwc_conv_first_split_impl(const float* x, const float* w, const float* scale, const float* shift, const float* out_scale2,
                                  void* out_hi, void* out_lo, unsigned* amax_out, int B, int T, int F, int act, hipStream_t);
int vs_nhwc_last_plan_impl(const unsigned* amax_in, int n_amax, const float* w, const float* scale, const float* shift, float* out_scale2, hipStream_t);
// row_hi != NULL: the output as the LSTM input GEMM's split A operand [B T][Kp] (scale out_scale2 from vs_nhwc_last_plan_impl), `out` unused
int vs_nhwc_conv_last_split_impl(const void* in_hi, const void* in_lo, const float* in_scale2, const float* w, const float* scale,
                                 const float* shift, float* out, int B, int T, int F, int act, hipStream_t,
                                 void* row_hi = nullptr, void* row_lo = nullptr, int Kp = 0, const float* out_scale2 = nullptr);
// wgrad_nhwc.hip: their weight gradient (pairs of workgroups, one partial-sum slab per pair)
#define VS_NHWC_WGRAD_MAX_PAIRS 128
size_t vs_nhwc_wgrad_partial_floats(int KT, int KF);
int vs_nhwc_wgrad_impl(const void* dz, const void* a_in, float* part, float* dw, int B, int T, int F, int KT, int KF, int dil, hipStream_t);
// nhwc_edge.hip: the HBM-bound kernels around them (cnn1, BatchNorm apply, cnn8)
int vs_nhwc_conv_first_impl(const float* x, const float* w, const float* scale, const float* shift, void* out,
                            int B, int T, int F, int act, double* bn_stats, hipStream_t,
                            const float* bias = nullptr /* scale / shift are BatchNorm constants of conv + bias */);
// cnn1 by recomputation (nhwc_edge.hip): input moments -> batch statistics of z1; one-pass backward
#define VS_FIRST_MOMENTS 35
#define VS_FIRST_BWD_SCRATCH_DOUBLES (64 * 9 + VS_FIRST_MOMENTS)
int vs_nhwc_first_moments_impl(const float* x, int B, int T, int F, double* mom, hipStream_t, double* det_slots = nullptr, int mom_is_zero = 0);
int vs_nhwc_first_stats_impl(const double* mom, const float* w, const float* bias, double count, double* stats, hipStream_t);
int vs_nhwc_first_bwd_impl(const void* da, const float* x, const float* w, const float* bias, int B, int T, int F, int act, int train,
                           const float* scale, const float* shift, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, float* dbias, float* dw, double* scratch, hipStream_t,
                           const double* moments = nullptr /* of x, when the caller still has them */, double* det_slots = nullptr);
int vs_nhwc_bn_apply_impl(const void* z, void* a, long long npix, int act, const float* scale, const float* shift, hipStream_t);
int vs_nhwc_conv_last_impl(const void* in, const float* w, const float* scale, const float* shift, float* out,
                           int B, int T, int F, int act, hipStream_t, double* bn_stats = nullptr,
                           const float* pre_scale = nullptr, const float* pre_shift = nullptr, int pre_act = VS_ACT_NONE,
                           void* rows_bf16 = nullptr /* eval: the output as bf16 rows [B T][Kp] (the LSTM input GEMM's A operand) instead of `out` */,
                           int Kp = 0);
int vs_nhwc_bn_act_bwd_impl(const void* da, const void* z, void* dz, long long npix, int act, int train,
                            const float* scale, const float* shift, const float* mean, const float* invstd,
                            float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, hipStream_t, int beside_wgrad = 0);
int vs_nhwc_bn_act_bwd_first_impl(const void* da, const void* z, const float* x, int B, int T, int F, int act, int train,
                                  const float* scale, const float* shift, const float* mean, const float* invstd,
                                  float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc, hipStream_t);
#define VS_NHWC_LAST_BWD_BLOCKS 1024
int vs_nhwc_conv_last_bwd_impl(const float* dz8, const float* w, const void* a7, void* din, float* part, float* dw,
                               int B, int T, int F, const void* z7, int act, const float* bn_scale, const float* bn_shift,
                               const float* bn_mean, const float* bn_invstd, double* bn_stats, hipStream_t stream);
int vs_nhwc_conv_dy_impl(const void* dz, const void* packed, void* dy, const void* z, int act,
                         const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd, double* bn_stats,
                         int B, int T, int F, int KT, int KF, int dil, hipStream_t stream);
int vs_nhwc_bn_bwd_from_dy_impl(const void* dy, const void* z, void* dz, long long npix, int train,
                                const float* scale, const float* mean, const float* invstd,
                                float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, hipStream_t stream, int rezero_doubles = 0,
                                int beside_wgrad = 0 /* the pass shares the CUs with a matrix-pipe kernel on another stream: a small grid */);
int vs_nhwc_bn_bwd_first_from_dy_impl(const void* dy, const void* z, const float* x, int B, int T, int F, int train,
                                      const float* scale, const float* mean, const float* invstd,
                                      float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc,
                                      hipStream_t stream);
// conv_bwd.hip: coefficients of the BatchNorm backward apply pass from the folded sums
int vs_bn_bwd_finalize_impl(double* stats, int slots, double count, int train, int C, const float* scale, const float* mean,
                            const float* invstd, float* dgamma, float* dbeta, float* dbias, float* coef, hipStream_t,
                            int rezero_doubles = 0 /* > 0: one fused launch that also clears that much of the scratch behind itself */);
// conv_edge.hip
int vs_bn_finalize_impl(double* stats, int slots, double count, int C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float eps, float momentum,
                        float* scale, float* shift, float* mean_out, float* invstd_out, hipStream_t, int rezero_doubles = 0);
int vs_bn_fold_impl(const float*, const float*, const float*, const float*, const float*, float, int, float*, float*, hipStream_t);
int vs_conv_first_fwd_impl(const float*, const float*, const float*, const float*, float*, int, int, int, int, unsigned* amax_out, hipStream_t);
int vs_conv_last_fwd_impl(const float*, const float*, const float*, const float*, float*, int, int, int, int, hipStream_t);
int vs_bn_train_impl(const float* x, float* y, int B, int C, int plane, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float eps, float momentum, int act, double* stats,
                     float* scale, float* shift, float* mean_out, float* invstd_out, unsigned* amax_out, hipStream_t,
                     int stats_slots = 0);
int vs_bn_train_feat_impl(const float* x, float* y, int B, int T, int F, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float eps, float momentum, int act, double* stats,
                          float* scale, float* shift, float* mean_out, float* invstd_out, hipStream_t);
int vs_bn_apply_impl(const float* x, float* y, int B, int C, int plane, int act, const float* scale, const float* shift, unsigned* amax_out, hipStream_t);
int vs_bn_apply_feat_impl(const float* x, float* y, int B, int T, int F, int act, const float* scale, const float* shift, hipStream_t);
int vs_bn_apply_feat_bf16_impl(const float* x, float* y, void* yb, int Kp, int B, int T, int F, int act, const float* scale, const float* shift,
                               hipStream_t);
int vs_bn_eval_consts_impl(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps, int C,
                           float* scale, float* shift, float* mean_out, float* invstd_out, hipStream_t);
// conv_bwd.hip
int vs_conv64_wgrad_impl(const float* dz, const float* in, float* part, float* dw, int B, int T, int F, int KT, int KF, int dil, hipStream_t);
int vs_conv64_wgrad_f16x3_impl(const float* dz, const float* in, const float* dz_scale2, const float* in_scale2,
                               float* part, float* dw, int B, int T, int F, int KT, int KF, int dil, hipStream_t,
                               int math = VS_MATH_CODE_F16X3);
int vs_bn_act_bwd_impl(const float* da, const float* z, float* dz, int C, long long R, int L, int act, int train,
                       const float* scale, const float* shift, const float* mean, const float* invstd,
                       float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, unsigned* amax_out, hipStream_t);
int vs_bn_act_bwd_first_impl(const float* da, const float* z, const float* x, float* xpad, int B, int T, int F, int act, int train,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc,
                             hipStream_t stream);
int vs_conv_last_dgrad_impl(const float* dz, const float* w, float* din, int B, int T, int F, hipStream_t);
int vs_conv_last_wgrad_impl(const float* dz, const float* in, float* part, float* dw, int B, int T, int F, hipStream_t);
int vs_conv_first_wgrad_impl(const float* dz, const float* x, double* acc, float* dw, int B, int T, int F, hipStream_t);
int vs_reduce_partials_impl(const float* part, int G, int n, float* out, hipStream_t);
// gemm_mfma.hip
int vs_gemm_general_impl(int layout_a, int layout_w, const float* A, int lda, const float* W, const float* W_hi,
                         int n_split, int ldw, float* C, int ldc, int M, int N, int K,
                         const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                         const float* gate, int ldg, int a_relu, int w_relu, int act, int accumulate,
                         int w_shift, int w_group, int splits, float* partials, hipStream_t);
int vs_gemm_general_bf16_impl(int layout_a, int layout_w, const float* A, int lda, const float* W, const float* W_hi,
                         int n_split, int ldw, float* C, int ldc, int M, int N, int K,
                         const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
                         const float* gate, int ldg, int a_relu, int w_relu, int act, int accumulate,
                         int w_shift, int w_group, int splits, float* partials, hipStream_t);
int vs_gemm_nt_impl(const float*, int, const float*, int, float*, int, int, int, int, const float*, const float*, const float*, int, int, int, int, hipStream_t);
// head_fused.hip: relu -> fc1 -> relu -> fc2 -> sigmoid in one launch (VS_MATH_BF16); h1 stays in registers between the two contractions
bool vs_head_fused_supported(int K1, int FC1, int FC2);
size_t vs_head_fused_packed_bytes(int K1, int FC1, int FC2);
int vs_head_fused_pack_impl(const float* w1, const float* b1, const float* w2, const float* b2, int K1, int FC1, int FC2, void* packed, hipStream_t);
int vs_head_fused_impl(const float* lstm_out, const void* packed, float* h1_out, float* logits, float* mask,
                       int M, int K1, int FC1, int FC2, hipStream_t);
int vs_gemm_nt_bf16_impl(const float*, int, const float*, int, float*, int, int, int, int, const float*, const float*, const float*, int, int, int, int, hipStream_t);
int vs_gemm_nt2_impl(const float*, int, const float*, const float*, int, int, float*, int, int, int, int, const float*, const float*, const float*, int, int, int, int, hipStream_t);
// gemm_bf16.hip: the LSTM contractions of the bf16 configuration (LDS-DMA ring, row / K-major operand forms)
int vs_cvt_rows_bf16_impl(const float* src, long long rows, int K, int ld, void* dst, int Kp, hipStream_t);
int vs_gemm_bf16_impl(int a_kmajor, int b_kmajor, const void* A, int lda, const void* B, int ldb, float* C, int ldc, float* C2, int split_m,
                      int M, int N, int K, const float* rowbias, int ldrb, int group, int accumulate, hipStream_t,
                      const float* gate = nullptr, int ldg = 0);
// layout of the three bf16 operand arrays inside one scratch region (256-byte aligned pieces): feat [M][Kp], W_ih [8H][Kp], dxg [M][8H]
struct VsLstmBf16Layout { size_t feat, wih, dxg, total; int Kp; };
inline VsLstmBf16Layout vs_lstm_bf16_layout(long long M, int K, int H) {
  VsLstmBf16Layout L;
  L.Kp = (K + 63) / 64 * 64;
  auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
  L.feat = 0;
  L.wih = up((size_t)M * L.Kp * 2);
  L.dxg = L.wih + up((size_t)8 * H * L.Kp * 2);
  L.total = L.dxg + up((size_t)M * 8 * H * 2);
  return L;
}
// lstm.hip
// math (VS_MATH_CODE_*): which products the persistent recurrences use -- fp32 MFMA, or (forward) split-f16 / f16 and
// (BPTT, VS_MATH_BF16 only) bf16; the pack calls write the matching operand form behind the fp32 one
int vs_lstm_pack_impl(const float*, const float*, float*, int, hipStream_t, int math = VS_MATH_CODE_FP32);
int vs_bilstm_recurrent_impl(const float* xg, const float* wp, float* state, float* out, float* gates_save, float* c_save,
                             int B, int T, int H, hipStream_t, int math = VS_MATH_CODE_FP32);
int vs_lstm_pack_t_impl(const float*, const float*, float*, int, hipStream_t, int math = VS_MATH_CODE_FP32);
int vs_bilstm_bwd_recurrent_impl(const float* wpt, float* state, float* gates, const float* c_all, const float* dout,
                                 int B, int T, int H, hipStream_t, int math = VS_MATH_CODE_FP32);
// reduce.hip
int vs_sigmoid_bwd_impl(const float* dmask, const float* mask, float* dlogits, long long n, hipStream_t);
int vs_sigmoid_bwd_rows_impl(const float* dmask, const float* mask, float* dlogits, long long rows, int N, void* rows_bf16, int Kp, hipStream_t);
int vs_colsum_impl(const float* x, int ld, int groups, int rows, int N, float* out, int ldo, hipStream_t);
