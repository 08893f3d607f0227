/* voicesplit_hip.h -- C ABI of libvoicesplit_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of Edresson/VoiceSplit: the speaker-conditioned
 * mask-prediction forward pass
 *     mask = model(mixed_spec[B,T,F], dvec[B,E])          train.py:94
 *                                                         utils/generic_utils.py:495,545
 * i.e. VoiceSplit.forward  (models/voicesplit/model.py:66-89) and
 *      VoiceFilter.forward (models/voicefilter/model.py:67-90).
 * The reference has no native layer at all (it dispatches to torch.nn / cuDNN), so there is no
 * upstream FFI to mirror; every entry point below names the reference lines it replaces.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer (fp32 unless noted) owned by the
 *    caller: inputs, outputs, the inference workspace and the training tape.  The library never
 *    allocates device memory and keeps no caller pointer after return.
 *  - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no hidden
 *    synchronisation except where an entry point says so (vs_profile_end, vs_lstm_status).
 *  - library-owned state (all of it): (1) ONE side stream + three events per device, created on first use: vs_backward runs the
 *    leaf gradients there and one kernel of every [weight gradient || BatchNorm backward pass] pair, vs_forward_train (VS_MATH_BF16) the weight-only launches of the
 *    step beside cnn1; both join it before they return, and the enqueue phase of concurrent calls on one device is serialised by a
 *    mutex while it is in use (vs_set_backward_overlap(0) turns it off); (2) the process-wide switches: vs_set_conv_kernel /
 *    vs_set_wgrad_kernel / vs_set_lstm_kernel / vs_set_backward_overlap and the option table behind vs_set_option (enum vs_option
 *    lists every one of them; A/B timing and cross-checks: every choice gives the same results unless the option says "ablation";
 *    the library reads NO environment variable), and the opt-in profiler vs_profile_begin / _end; (3) the error word of the
 *    persistent recurrence behind vs_lstm_status and a 64-byte zero page in device memory (static __device__ data of the bf16 GEMM:
 *    source of out-of-range operand pieces).  Calls on different streams with disjoint buffers are otherwise independent.
 *  - return 0 on success, <0 on error (-1 bad argument, -2 HIP runtime error); the message is
 *    available from vs_last_error() (thread-local).  No exceptions cross the ABI.
 *  - tensors are dense row-major with the reference's layouts: spectrogram [B][T][F]
 *    (F contiguous), conv activations [B][C][T][F] (VS_MATH_BF16: channels-last bf16 [B][T][F][C]
 *    inside the workspace / tape), LSTM features [B][T][8F] with feature index c*F+f,
 *    nn.Linear / nn.LSTM weights [out][in].
 */
#ifndef VOICESPLIT_HIP_H
#define VOICESPLIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: these are its only exports */
#endif

#define VS_ABI_VERSION 9

/* activation codes */
#define VS_ACT_RELU 0     /* VoiceFilter conv stack (models/voicefilter/model.py:21..54), head */
#define VS_ACT_MISH 1     /* VoiceSplit conv stack  (utils/generic_utils.py:395-399)           */
#define VS_ACT_NONE 2
#define VS_ACT_SIGMOID 3

/* arithmetic of the 64->64 conv layers (cnn2..cnn7 forward and data gradient) */
#define VS_MATH_FP32 0    /* v_mfma_f32_32x32x2_f32: bitwise an fp32 fmaf chain                              */
#define VS_MATH_F16X3 1   /* fp32 operands split into two f16 halves, three v_mfma_f32_32x32x16_f16 per
                             product term set, fp32 accumulate: fp32-class accuracy at 3/16 of the matrix time */
#define VS_MATH_BF16 2    /* BASELINE configs[2]: the conv stack on channels-last bf16 tensors -- activations and the
                             training tape are STORED as bf16 [B][T][F][64] (half the bytes of every pass), bf16 MFMA
                             operands, fp32 accumulate; BatchNorm statistics, LSTM recurrence, head and master weights
                             stay fp32, the LSTM GEMMs round their operands to bf16.  NOT fp32-class (8-bit mantissa):
                             opt-in only, never a default (tests/test_gpu_bf16.py states its bounds)                  */

/* BatchNorm mode */
#define VS_BN_EVAL 0      /* running statistics (model.eval(), utils/generic_utils.py:479,533) */
#define VS_BN_TRAIN 1     /* batch statistics + running-stat update (model.train(), train.py:84) */

typedef struct vs_dims {
  int B;    /* utterances in the batch                                  */
  int T;    /* frames (301 for 3 s)                                     */
  int F;    /* config.audio[backend].num_freq (601)                     */
  int E;    /* config.model.emb_dim  (256)                              */
  int H;    /* config.model.lstm_dim (400); must be a multiple of 8     */
  int FC1;  /* config.model.fc1_dim  (600)                              */
  int FC2;  /* config.model.fc2_dim  (601)                              */
  int math; /* VS_MATH_FP32, VS_MATH_F16X3 or VS_MATH_BF16 (dense contractions) */
} vs_dims;

/* One Conv2d + BatchNorm2d pair of the nn.Sequential (state_dict conv.{i}.*, conv.{i+1}.*). */
typedef struct vs_conv_layer {
  const float* weight;        /* [Cout][Cin][KT][KF]                       */
  const float* bias;          /* [Cout]                                    */
  const float* bn_weight;     /* [Cout] gamma                              */
  const float* bn_bias;       /* [Cout] beta                               */
  float* bn_running_mean;     /* [Cout] read in eval, updated in train     */
  float* bn_running_var;      /* [Cout]                                    */
} vs_conv_layer;

/* Every parameter of VoiceSplit/VoiceFilter.__init__ (models/voicesplit/model.py:10-64). */
typedef struct vs_params {
  vs_conv_layer conv[8];      /* cnn1..cnn8 = conv.{1,5,9,13,17,21,25,28}          */
  const float* w_ih[2];       /* lstm.weight_ih_l0, _reverse   [4H][8F+E]          */
  const float* w_hh[2];       /* lstm.weight_hh_l0, _reverse   [4H][H]             */
  const float* b_ih[2];       /* lstm.bias_ih_l0, _reverse     [4H]                */
  const float* b_hh[2];       /* lstm.bias_hh_l0, _reverse     [4H]                */
  const float* fc1_w;         /* [FC1][2H] */
  const float* fc1_b;         /* [FC1]     */
  const float* fc2_w;         /* [FC2][FC1]*/
  const float* fc2_b;         /* [FC2]     */
} vs_params;

/* Byte offsets of the intermediates inside the caller-provided workspace (for inspection and
 * stage-level parity tests).  Filled by vs_workspace_layout(). */
typedef struct vs_ws_layout {
  size_t total_bytes;
  size_t act0, act1;          /* ping-pong [B][64][T][F]                           */
  size_t feat;                /* cnn8 output == LSTM features [B][T][8F]           */
  size_t dvbias;              /* dvec @ W_ih[:,8F:]^T + b_ih + b_hh   [B][8H]      */
  size_t xg;                  /* gate pre-activations [B][T][2*4H]                 */
  size_t lstm_out;            /* [B][T][2H]                                        */
  size_t fc1_out;             /* relu(fc1) [B*T][FC1]                              */
  size_t conv_packed[6];      /* fragment-ordered weights of cnn2..cnn7            */
  size_t bn_scale, bn_shift;  /* [8][64] folded BatchNorm                          */
  size_t bn_stats;            /* [8][64][2] double: sum, sum of squares (train)    */
  size_t lstm_packed;         /* fragment-ordered W_hh, both directions            */
  size_t lstm_state;          /* h ping/pong + c, [3][2][H][Bpad]                  */
  size_t conv_scales;         /* [8] scale slots of the split-f16 conv launches: operand scales + running |max| arrays */
  size_t gemm_scales;         /* operand scales of the split-f16 LSTM input GEMM */
} vs_ws_layout;

int vs_abi_version(void);
const char* vs_last_error(void);

/* Opt-in per-stage GPU timing (HIP events on the caller's stream, no synchronisation while
 * enabled).  Forward slots: cnn1..cnn8 conv kernels (0-7), LSTM input GEMMs (8), LSTM recurrence
 * (9), head (10), training-forward BatchNorm passes (11).  Backward slots: head (12), BPTT (13),
 * LSTM weight/input-gradient GEMMs (14), BatchNorm+activation backward (15), weight gradient of
 * cnn2..cnn7 (16-21), data gradient of cnn2..cnn7 (22-27), cnn1/cnn8 kernels (28).  A slot may be
 * entered several times per step; vs_profile_end returns the summed time and the entry count.
 * Instrumentation for bench.py only: process-global, not thread safe. */
#define VS_PROF_CNN1 0
#define VS_PROF_CNN2 1      /* cnn2..cnn7 = 1..6 */
#define VS_PROF_CNN8 7
#define VS_PROF_LSTM_GEMM 8
#define VS_PROF_LSTM_REC 9
#define VS_PROF_HEAD 10
#define VS_PROF_FWD_BN 11
#define VS_PROF_BWD_HEAD 12
#define VS_PROF_BWD_LSTM_REC 13
#define VS_PROF_BWD_LSTM_GEMM 14
#define VS_PROF_BWD_BN 15
#define VS_PROF_BWD_WGRAD 16   /* cnn2..cnn7 = 16..21 */
#define VS_PROF_BWD_DGRAD 22   /* cnn2..cnn7 = 22..27 */
#define VS_PROF_BWD_EDGE 28
#define VS_PROF_SLOTS 29
int vs_profile_begin(int max_calls);
int vs_profile_end(float* ms_total /* [VS_PROF_SLOTS] */, int* calls /* [VS_PROF_SLOTS] */);

/* Workspace the caller must provide to the stage / whole-forward calls. */
int vs_workspace_layout(const vs_dims* dims, vs_ws_layout* out);
size_t vs_workspace_bytes(const vs_dims* dims);

/* ---- whole path: replaces VoiceSplit.forward / VoiceFilter.forward ----------------------
 * x [B][T][F], dvec [B][E] -> mask [B][T][FC2].  conv_act = VS_ACT_MISH (VoiceSplit) or
 * VS_ACT_RELU (VoiceFilter). */
int vs_forward(const vs_dims* dims, const vs_params* params, const float* x, const float* dvec,
               int conv_act, int bn_mode, void* workspace, size_t workspace_bytes,
               float* mask, void* stream);

/* ---- eval-mode forward with the weight-only work done once ------------------------------------
 * validation() / serving call the model again and again with unchanged weights
 * (utils/generic_utils.py:476-558: sample by sample, B = 1).  vs_forward re-derives on every call what
 * depends on the parameters alone (~40 small launches: BatchNorm folded into scale/shift, conv weights
 * packed in MFMA fragment order with their power-of-two scale, W_ih split into f16 halves, W_hh packed);
 * vs_prepare_weights does it once into a caller-owned buffer (vs_prepared_bytes(dims): depends on F, E,
 * H and dims.math, not on B or T; 256-byte aligned) and vs_forward_prepared reads it.  The caller
 * re-prepares whenever a parameter, a BatchNorm running statistic or dims.math changes -- the library
 * cannot see that.  Results are bit-identical to vs_forward(..., VS_BN_EVAL, ...). */
size_t vs_prepared_bytes(const vs_dims* dims);
int vs_prepare_weights(const vs_dims* dims, const vs_params* params, void* prepared, size_t prepared_bytes, void* stream);
int vs_forward_prepared(const vs_dims* dims, const vs_params* params, const void* prepared, size_t prepared_bytes,
                        const float* x, const float* dvec, int conv_act, void* workspace, size_t workspace_bytes,
                        float* mask, void* stream);

/* ---- stages (each is what vs_forward runs, in order) --------------------------------------- */
/* self.conv(x.unsqueeze(1)) + transpose/view: models/voicesplit/model.py:68-74 -> feat [B][T][8F] */
int vs_conv_stack_fwd(const vs_dims* dims, const vs_params* params, const float* x, int conv_act,
                      int bn_mode, void* workspace, size_t workspace_bytes, float* feat, void* stream);
/* repeat/cat d-vector + self.lstm: models/voicesplit/model.py:77-82 -> lstm_out [B][T][2H] */
int vs_bilstm_fwd(const vs_dims* dims, const vs_params* params, const float* feat, const float* dvec,
                  void* workspace, size_t workspace_bytes, float* lstm_out, void* stream);
/* relu/fc1/relu/fc2/sigmoid: models/voicesplit/model.py:83-87; logits may be NULL */
int vs_head_fwd(const vs_dims* dims, const vs_params* params, const float* lstm_out,
                void* workspace, size_t workspace_bytes, float* logits, float* mask, void* stream);

/* ---- kernels (unit-test surface) --------------------------------------------------------- */
/* BN(conv+bias) = conv*scale + shift */
int vs_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
               const float* conv_bias, float eps, int C, float* scale, float* shift, void* stream);
/* cnn1: [B][T][F] -> [B][64][T][F], weight [64][1][1][7] */
int vs_conv_first_fwd(const float* x, const float* w, const float* scale, const float* shift,
                      float* out, int B, int T, int F, int act, void* stream);
/* cnn2..cnn7: 64->64, (KT,KF) in {(7,1),(5,5)}, time dilation dil, "same" zero padding */
size_t vs_conv64_packed_floats(int KT, int KF);
int vs_conv64_pack(const float* w, float* packed, int KT, int KF, void* stream);
int vs_conv64_fwd(const float* in, const float* packed, const float* scale, const float* shift,
                  float* out, int B, int T, int F, int KT, int KF, int dil, int act, void* stream);
/* same layer in VS_MATH_F16X3 arithmetic: vs_pow2_scale gives {s, 1/s} for the input tensor,
 * vs_conv64_pack_f16 packs (and scales) the weights; amax_scratch = one uint32 of device scratch.
 * transpose_flip = 1 packs the data-gradient weights. */
size_t vs_conv64_packed_f16_floats(int KT, int KF);
int vs_pow2_scale(const float* x, long long n, void* amax_scratch, float* scale2, void* stream);
int vs_conv64_pack_f16(const float* w, void* packed, int KT, int KF, int transpose_flip, void* amax_scratch,
                       float* w_scale2, void* stream);
int vs_conv64_f16x3_fwd(const float* in, const void* packed, const float* scale, const float* shift,
                        const float* in_scale2, const float* w_scale2, float* out,
                        int B, int T, int F, int KT, int KF, int dil, int act, void* stream);
/* ---- channels-last bf16 form of the same layers: what VS_MATH_BF16 (BASELINE configs[2]) runs -----------------
 * Activations [B][T][F][64] bf16 (a pixel's 64 channels contiguous), fp32 accumulate, bf16 store.
 * vs_nhwc_conv_pack: weights [64][64][KT][KF] fp32 -> register-fragment order (vs_nhwc_conv_packed_bytes bytes);
 * transpose_flip = 1 packs the data-gradient weights.  vs_nhwc_conv: out = act(conv(in) * scale[co] + shift[co])
 * with 'same' zero padding and time dilation `dil`, (KT,KF) in {(7,1),(5,5)}; bn_stats (or NULL) = per-channel
 * {sum, sum of squares} of the outputs accumulated into [64 slots][64 channels][2] doubles the caller zeroed
 * (train-mode BatchNorm statistics; act must be VS_ACT_NONE then).  All buffers 16-byte aligned. */
size_t vs_nhwc_conv_packed_bytes(int KT, int KF);
int vs_nhwc_conv_pack(const float* w, void* packed, int KT, int KF, int transpose_flip, void* stream);
int vs_nhwc_conv(const void* in, const void* packed, const float* scale, const float* shift, void* out,
                 int B, int T, int F, int KT, int KF, int dil, int act, double* bn_stats, void* stream);
/* The same 64 -> 64 convs, FORWARD, in the fp32-class split-f16 arithmetic (VS_MATH_F16X3) on channels-last operands
 * (csrc/conv_nhwc_f16x3.hip; models/voicesplit/model.py:21-48 under model.eval()).  A tensor is two f16 planes [B][T][F][64]
 * (hi, lo) and a power-of-two scale pair scale2 = {s, 1/s} in device memory: x = (hi + lo) / s (vs_f16x3_split / vs_f16x3_merge
 * convert from / to fp32).  vs_nhwc_conv_f16x3_layer runs one layer from fp32 weights w [64][64][KT][KF] and folded BatchNorm
 * constants (y = act(conv(x) * bn_scale + bn_shift)): it splits and packs the weights into `scratch`
 * (vs_nhwc_conv_f16x3_scratch_bytes, 256-byte aligned; packed_ready != 0: a previous call with the same weights left them
 * there), derives the output scale from max |x| (amax_in: n_amax uints holding the bit patterns of non-negative floats, their
 * maximum = max |x|) and writes out_hi / out_lo / out_scale2 and, when amax_out is not NULL, folds max |y| into
 * amax_out[VS_AMAX_SLOTS = 1024] (zero it first; it is the next layer's amax_in with n_amax = 1024). */
size_t vs_nhwc_conv_f16x3_scratch_bytes(int KT, int KF);
int vs_nhwc_conv_f16x3_layer(const void* in_hi, const void* in_lo, const float* in_scale2, const unsigned* amax_in, int n_amax,
                             const float* w, const float* bn_scale, const float* bn_shift, void* scratch, int packed_ready,
                             void* out_hi, void* out_lo, float* out_scale2, unsigned* amax_out,
                             int B, int T, int F, int KT, int KF, int dil, int act, void* stream);
int vs_f16x3_split(const float* x, const float* scale2, void* hi, void* lo, long long n, void* stream);
int vs_f16x3_merge(const void* hi, const void* lo, const float* scale2, float* x, long long n, void* stream);
/* bf16 GEMM of the same configuration (the three LSTM contractions): C[M][N] (+)= op(A) op(B) (+ rowbias[m / group][n]);
 * A, B bf16; *_kmajor = 0: element (i, k) at i*ld + k, 1: at k*ld + i (no transposed copies: transposing LDS reads).
 * vs_cvt_rows_bf16: fp32 [rows][ld] (K valid columns) -> bf16 [rows][Kp], zero padded.  Row-form operands must be
 * padded to a multiple of 64 in k; ld multiples of 8; 16-byte aligned. */
int vs_cvt_rows_bf16(const float* src, long long rows, int K, int ld, void* dst, int Kp, void* stream);
int vs_gemm_bf16(int a_kmajor, int b_kmajor, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                 const float* rowbias, int ldrb, int group, int accumulate, void* stream);
/* The same contraction with its result rows stored into TWO matrices stacked along M: rows m < split_m go to C + m * ldc,
 * rows m >= split_m to C2 + (m - split_m) * ldc.  This is how vs_backward writes dW_ih of both LSTM directions from ONE
 * col x col contraction over K = B*T (dxg^T @ feat, M = 8H: rows < 4H -> lstm.weight_ih_l0.grad, the rest -> _reverse). */
int vs_gemm_bf16_split(int a_kmajor, int b_kmajor, const void* A, int lda, const void* B, int ldb, float* C, float* C2, int ldc, int split_m,
                       int M, int N, int K, int accumulate, void* stream);
/* The same contraction under a relu mask: C[m][n] = (op(A) op(B))[m][n] where gate[m * ldg + n] > 0, else 0.  This is how vs_backward
 * runs the head's two data gradients (models/voicesplit/model.py:83-85 backwards): dfc1 = (dlogits @ W2) * (h1 > 0) and
 * dlstm_out = (dfc1 @ W1) * (lstm_out > 0), A = the bf16 row copy of the incoming gradient, B = the bf16 weight in K-major form. */
int vs_gemm_bf16_gated(int a_kmajor, int b_kmajor, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                       const float* gate, int ldg, void* stream);
/* the memory-bound kernels around them.  cnn1: x [B][T][F] fp32 -> [B][T][F][64] bf16 (bn_stats as above);
 * BatchNorm + activation apply a = act(z * scale[c] + shift[c]) over npix pixels (a may alias z); cnn8 + transpose/view:
 * [B][T][F][64] bf16 -> [B][T][8][F] fp32. */
int vs_nhwc_conv_first(const float* x, const float* w, const float* scale, const float* shift, void* out,
                       int B, int T, int F, int act, double* bn_stats, void* stream);
/* cnn1 by recomputation (what vs_forward_train / vs_backward run in VS_MATH_BF16): cnn1 (models/voicesplit/model.py:17-19) is 7
 * multiply-adds per output and its input is 1/64 of its output, so neither its conv output z1 nor a separate BatchNorm apply
 * pass is needed.  vs_nhwc_first_moments: x [B][T][F] -> the 35 moments of its seven zero-padded shifts x_k = x[..][f + k - 3]
 * (doubles: S[k] = sum x_k, k = 0..6, then R[k][k'] = sum x_k x_k' for k <= k', row by row).  vs_nhwc_first_stats: the
 * per-channel {sum, sum of squares} of z1 = conv(x) + bias over count = B*T*F pixels that follow from them (stats [64][2],
 * one slot: feed vs_bn_finalize with slots = 1).  The forward is then ONE pass, vs_nhwc_conv_first with the BatchNorm folded
 * into its arguments: scale = bn scale, shift = bn shift + bias * bn scale.  vs_nhwc_first_bwd: the whole backward of cnn1 +
 * BatchNorm + activation in ONE pass over da1 [B][T][F][64] bf16 (z1 recomputed from x beside the derivative; the BatchNorm
 * backward is linear in dy and z, so the weight gradient follows from sum dy x_k and the moments): dgamma, dbeta, dbias [64],
 * dw [64][7] (all overwritten); scale / shift / mean / invstd = vs_bn_finalize's outputs (shift WITHOUT the bias fold);
 * scratch = vs_nhwc_first_bwd_scratch_doubles() doubles. */
int vs_nhwc_first_moments(const float* x, int B, int T, int F, double* moments /* [35] */, void* stream);
int vs_nhwc_first_stats(const double* moments, const float* w, const float* bias, double count, double* stats /* [64][2] */, void* stream);
int vs_nhwc_first_bwd_scratch_doubles(void);
int vs_nhwc_first_bwd(const void* da, const float* x, const float* w, const float* bias, int B, int T, int F, int act, int bn_mode,
                      const float* scale, const float* shift, const float* mean, const float* invstd,
                      float* dgamma, float* dbeta, float* dbias, float* dw, double* scratch, void* stream);
/* Train-mode nn.BatchNorm2d between a conv that accumulated statistics and the apply pass (models/voicesplit/model.py:19 under
 * model.train(), train.py:84): stats = [slots][C][2] doubles {sum, sum of squares} over `count` values per channel -- what
 * vs_nhwc_conv / vs_nhwc_conv_first / vs_nhwc_conv_last_pre leave in bn_stats (slots = 64, C = 64 resp. 8, count = B*T*F);
 * the slots are folded into slot 0 (stats is modified).  Out: scale[c] = gamma / sqrt(var + eps), shift[c] = beta - mean * scale
 * (biased variance: the operands of vs_nhwc_bn_apply, a = act(z * scale + shift)), mean_out / invstd_out (may be NULL: what the
 * backward kernels take as bn_mean / bn_invstd), and running_mean / running_var updated in place exactly as nn.BatchNorm2d
 * does: r = (1 - momentum) r + momentum * {mean, unbiased variance}; pass both as NULL for no update.  The reference uses
 * eps = 1e-5, momentum = 0.1 (nn.BatchNorm2d defaults).  num_batches_tracked is the caller's (host-side counter). */
int vs_bn_finalize(double* stats, int slots, double count, int C, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float eps, float momentum,
                   float* scale, float* shift, float* mean_out, float* invstd_out, void* stream);
int vs_nhwc_bn_apply(const void* z, void* a, long long npix, int act, const float* scale, const float* shift, void* stream);
int vs_nhwc_conv_last(const void* in, const float* w, const float* scale, const float* shift, float* out,
                      int B, int T, int F, int act, void* stream);
/* cnn8 straight on the un-normalised output z7 of cnn7 (train mode): a7 = pre_act(z7 * pre_scale + pre_shift) -- the
 * BatchNorm + Mish/ReLU of models/voicesplit/model.py:47-48 -- is formed in registers, rounded to bf16 exactly as
 * vs_nhwc_bn_apply would have stored it, and fed to the matrix pipe: no apply pass over cnn7's output and no a7 tensor.
 * out = conv * scale + shift, unactivated; bn_stats (may be NULL): [64 slots][8][2] doubles, zeroed by the caller,
 * receive the per-channel {sum, sum of squares} of out (cnn8's own train-mode BatchNorm). */
int vs_nhwc_conv_last_pre(const void* z7, const float* pre_scale, const float* pre_shift, int pre_act, const float* w,
                          const float* scale, const float* shift, float* out, double* bn_stats, int B, int T, int F, void* stream);
/* backward.  Weight gradient of cnn2..cnn7: dz, in [B][T][F][64] bf16 -> dw [64][64][KT][KF] fp32; partials =
 * vs_nhwc_conv_wgrad_partial_floats(KT, KF) floats of scratch.  BatchNorm + activation backward over npix pixels
 * (dz may alias da; stats = 64*64*2 doubles, coef = 192 floats of scratch), and the same fused with cnn1's 1x7 weight
 * gradient (dw [64][7], acc = 448 doubles; dz1 is never written).  cnn8 backward in one pass: dz8 [B][T][8][F] fp32,
 * a7 [B][T][F][64] bf16 -> din (bf16, layout of a7) and dw [8][64]; partials = vs_nhwc_conv_last_bwd_blocks() * 512 floats. */
size_t vs_nhwc_conv_wgrad_partial_floats(int KT, int KF);
int vs_nhwc_conv_wgrad(const void* dz, const void* in, float* partials, float* dw, int B, int T, int F, int KT, int KF, int dil,
                       void* stream);
int vs_nhwc_bn_act_bwd(const void* da, const void* z, void* dz, long long npix, int act, int bn_mode,
                       const float* scale, const float* shift, const float* mean, const float* invstd,
                       float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, void* stream);
int vs_nhwc_bn_act_bwd_first(const void* da, const void* z, const float* x, int B, int T, int F, int act, int bn_mode,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc, void* stream);
int vs_nhwc_conv_last_bwd_blocks(void);
int vs_nhwc_conv_last_bwd(const float* dz8, const float* w, const void* a7, void* din, float* partials, float* dw,
                          int B, int T, int F, void* stream);
/* The dy forms vs_backward chains: the kernel that produces a layer's input gradient da also applies the activation
 * derivative of the layer below, dy = da * act'(z * bn_scale + bn_shift) (z = that layer's conv + bias output), stores
 * dy instead of da and accumulates the per-channel {sum dy, sum dy * xhat} into bn_stats (64*64*2 doubles the caller
 * zeroed) -- the first of the two passes of vs_nhwc_bn_act_bwd, without reading the tensor again.
 * vs_nhwc_bn_bwd_from_dy / _first_from_dy are the second pass: parameter gradients + dz = cA dy + cB z + cC (dz may
 * alias dy).  act in {VS_ACT_MISH, VS_ACT_RELU}.  vs_nhwc_conv_last_bwd_dy: a7 may be NULL -- the layer input is then
 * recomputed from z7 and the BatchNorm constants (the counterpart of vs_nhwc_conv_last_pre). */
int vs_nhwc_conv_dy(const void* dz, const void* packed, void* dy, const void* z, int act,
                    const float* bn_scale, const float* bn_shift, const float* bn_mean, const float* bn_invstd, double* bn_stats,
                    int B, int T, int F, int KT, int KF, int dil, void* stream);
int vs_nhwc_conv_last_bwd_dy(const float* dz8, const float* w, const void* a7, void* dy, float* partials, float* dw,
                             const void* z7, int act, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                             const float* bn_invstd, double* bn_stats, int B, int T, int F, void* stream);
int vs_nhwc_bn_bwd_from_dy(const void* dy, const void* z, void* dz, long long npix, int bn_mode,
                           const float* scale, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, void* stream);
int vs_nhwc_bn_bwd_first_from_dy(const void* dy, const void* z, const float* x, int B, int T, int F, int bn_mode,
                                 const float* scale, const float* mean, const float* invstd,
                                 float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc, void* stream);
/* cnn8: [B][64][T][F] -> [B][T][8][F], weight [8][64][1][1] */
int vs_conv_last_fwd(const float* in, const float* w, const float* scale, const float* shift,
                     float* out, int B, int T, int F, int act, void* stream);
/* C[M][N] = act(opA(A)[M][K] @ W[N][K]^T + bias1[n] + bias2[n] + rowbias[m/group][n]) */
int vs_gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
               const float* bias1, const float* bias2, const float* rowbias, int ldrb, int group,
               int a_relu, int act, void* stream);
/* recurrence over precomputed gate inputs xg [B][T][8H] -> out [B][T][2H] */
size_t vs_lstm_packed_floats(int H);
size_t vs_lstm_state_floats(int B, int H);
int vs_lstm_pack(const float* w_hh_fwd, const float* w_hh_bwd, float* packed, int H, void* stream);
int vs_bilstm_recurrent(const float* xg, const float* packed_whh, float* state, float* out,
                        int B, int T, int H, void* stream);
/* The same with the arithmetic of the recurrent products h_{t-1} @ W_hh^T chosen by the caller -- what vs_forward* and
 * vs_backward pass from dims.math (models/voicesplit/model.py:82, nn.LSTM's recurrent GEMV): VS_MATH_FP32 = fp32 MFMA
 * (what the calls above use), VS_MATH_F16X3 = h and W_hh split into f16 hi + lo, three f16 MFMA products (fp32-class;
 * the BPTT keeps the fp32 MFMA), VS_MATH_BF16 = h / W_hh rounded to f16 in the forward, gate gradients / W_hh^T rounded
 * to bf16 in the BPTT, one product each.  Pack and recurrence must be given the same `math`; the packed buffers hold the
 * fp32 fragment form (used by the one-launch-per-step kernels in every arithmetic) followed by the 16-bit form.
 * gates_save / c_save may be NULL (inference). */
int vs_lstm_pack_math(const float* w_hh_fwd, const float* w_hh_bwd, float* packed, int H, int math, void* stream);
int vs_bilstm_recurrent_math(const float* xg, const float* packed_whh, float* state, float* out, float* gates_save, float* c_save,
                             int B, int T, int H, int math, void* stream);

/* =============================================================================================
 * Training: forward that keeps what backward needs + the backward pass itself.
 * Replaces what autograd records/replays for `mask = model(x, emb)` ... `loss.backward()`
 * (train.py:94-110) through models/voicesplit/model.py:66-89: every Conv2d / BatchNorm2d / Mish|ReLU
 * / nn.LSTM / nn.Linear / sigmoid node of the reference graph.
 * ============================================================================================= */

/* d(loss)/d(parameter), same shapes as vs_params.  Every non-NULL pointer is OVERWRITTEN (the
 * caller -- torch.autograd -- accumulates into .grad itself). */
typedef struct vs_conv_layer_grad {
  float* weight;              /* [Cout][Cin][KT][KF]                       */
  float* bias;                /* [Cout]  (exactly 0 under batch-stat BN)   */
  float* bn_weight;           /* [Cout]  d/d gamma                         */
  float* bn_bias;             /* [Cout]  d/d beta                          */
} vs_conv_layer_grad;

typedef struct vs_grads {
  vs_conv_layer_grad conv[8];
  float* w_ih[2];             /* [4H][8F+E] */
  float* w_hh[2];             /* [4H][H]    */
  float* b_ih[2];             /* [4H]       */
  float* b_hh[2];             /* [4H]       */
  float* fc1_w; float* fc1_b; float* fc2_w; float* fc2_b;
  float* dvec;                /* [B][E] d/d speaker embedding, may be NULL */
  void* leaves_event;         /* ABI 9.  NULL, or a hipEvent_t of the caller: vs_backward records it when every gradient of the head and
                                 the BiLSTM (fc1 / fc2 / w_ih / w_hh / b_ih / b_hh: 73 of the 75.5 MB at config.json's sizes) is final --
                                 about 5 ms into the backward pass at B = 64, the conv stack's 25 ms still ahead.  A data-parallel caller
                                 lets its collective stream wait for it and starts the all-reduce of that part of its gradient bucket
                                 beside the conv backward (voicesplit_amd/trainer.py).  The event is recorded on the
                                 library's side stream when the backward overlap is on, else on `stream`. */
} vs_grads;

/* Byte offsets inside the caller-provided training tape.  vs_forward_train fills the "saved"
 * part; vs_backward reads it and uses the rest as scratch.  The tape must not be touched between
 * the two calls; one tape per in-flight forward. */
typedef struct vs_tape_layout {
  size_t total_bytes;
  /* saved by the forward */
  size_t z[7];                /* conv+bias outputs of cnn1..cnn7 before BatchNorm [B][64][T][F] (VS_MATH_BF16: z[0] is not
                                 kept -- cnn1 is recomputed from x -- and a[6] only under VS_BN_EVAL) */
  size_t a[7];                /* act(BN(z)) = input of the next layer                           */
  size_t z8;                  /* cnn8 conv+bias [B][T][8][F]                                    */
  size_t feat;                /* act(BN(z8)) = LSTM features [B][T][8F]                         */
  size_t bn_scale, bn_shift, bn_mean, bn_invstd;   /* [8][64] constants the forward used       */
  size_t gates;               /* [B][T][8H]: gate pre-activations -> activated gates i,f,g,o ->
                                 (backward) gradient wrt the gate pre-activations             */
  size_t cstate;              /* [B][T][2H] cell states                                         */
  size_t lstm_out;            /* [B][T][2H]                                                     */
  size_t fc1_out;             /* relu(fc1) [B*T][FC1]                                           */
  /* backward intermediates (kept for stage-level parity tests) */
  size_t dlogits;             /* [B*T][FC2] */
  size_t dfc1;                /* [B*T][FC1] gradient wrt fc1 pre-activation                     */
  size_t dlstm_out;           /* [B*T][2H]                                                      */
  size_t dsum;                /* [B][8H] sum over t of the gate gradients                       */
  size_t dfeat;               /* [B][T][8F] gradient wrt feat, then wrt z8 (in place)           */
  size_t grad0, grad1;        /* ping-pong [B][64][T][F] activation gradients                   */
  /* scratch */
  size_t dvbias, conv_packed[6], pack_tmp, lstm_packed, lstm_packed_t, lstm_state, lstm_bwd_state;
  size_t consts;              /* ones[64], zeros[64] */
  size_t bn_stats, bn_coef, first_acc, colsum_tmp, partials;
  size_t conv_scales;         /* [16] scale slots (8 forward, 8 backward): operand scales + running |max| arrays */
  size_t gemm_scales;         /* operand scales of the split-f16 LSTM GEMMs (feat, W_ih, gate gradients) */
  size_t lstm_bf16;           /* VS_MATH_BF16: feat [B*T][Kp], [W_ih; W_ih_reverse] [8H][Kp], gate gradients [B*T][8H] as bf16 */
  size_t det_turn;            /* ABI 9: the turn words of the deterministic mode (VS_OPT_DETERMINISTIC), then its slot scratch for cnn1's backward */
  size_t conv_packed_t[6];    /* ABI 9, VS_MATH_BF16: the data-gradient weight images of cnn2..cnn7 (transposed, tap-flipped); written by
                                 vs_forward_train beside its own images (and the transposed W_hh image, lstm_packed_t), read by vs_backward */
} vs_tape_layout;

int vs_tape_layout_query(const vs_dims* dims, vs_tape_layout* out);
size_t vs_tape_bytes(const vs_dims* dims);

/* Forward with the tape.  bn_mode VS_BN_TRAIN: batch statistics, running buffers updated
 * (model.train(), train.py:84); VS_BN_EVAL: running statistics (fine-tuning with frozen BN). */
int vs_forward_train(const vs_dims* dims, const vs_params* params, const float* x, const float* dvec,
                     int conv_act, int bn_mode, void* tape, size_t tape_bytes, float* mask, void* stream);

/* Backward: dmask [B][T][FC2] = d(loss)/d(mask) -> every gradient in `grads`.  `mask` is the
 * tensor vs_forward_train returned; conv_act / bn_mode / dims / params must be those of the forward
 * call and the parameters unchanged since (VS_MATH_BF16: the tape already holds this pass's weight
 * images -- conv_packed_t, lstm_packed_t, the bf16 W_ih -- written by vs_forward_train).  The
 * gradient wrt the spectrogram x is not produced (the reference never asks for it: x is data). */
int vs_backward(const vs_dims* dims, const vs_params* params, const float* x, const float* dvec,
                int conv_act, int bn_mode, void* tape, size_t tape_bytes,
                const float* mask, const float* dmask, const vs_grads* grads, void* stream);

/* ---- backward kernels (unit-test surface) -------------------------------------------------- */
/* data gradient of cnn2..cnn7 = vs_conv64_fwd with weights packed by this (transpose + tap flip) */
int vs_conv64_pack_dgrad(const float* w, float* packed, int KT, int KF, void* stream);
/* weight gradient of cnn2..cnn7: dz, in [B][64][T][F] -> dw [64][64][KT][KF]; partials scratch */
size_t vs_conv64_wgrad_partial_floats(int KT, int KF);
int vs_conv64_wgrad(const float* dz, const float* in, float* partials, float* dw,
                    int B, int T, int F, int KT, int KF, int dil, void* stream);
/* the same in VS_MATH_F16X3 arithmetic; scratch8 = 8 floats (operand scales are derived inside) */
int vs_conv64_wgrad_f16x3(const float* dz, const float* in, float* partials, float* dw, float* scratch8,
                          int B, int T, int F, int KT, int KF, int dil, void* stream);
/* which split-f16 weight-gradient kernel vs_conv64_wgrad_f16x3 / vs_backward launch: 0 = by problem
 * size (default: a ring kernel once every workgroup gets >= 2 columns -- the four-wave form for 5x5 --
 * else the kt-split kernel), 1 = eight-wave ring, 2 = kt-split, 3 = four-wave ring (5x5; 7x1 falls back
 * to the eight-wave ring).  Process-wide; returns -1 for an unknown mode. */
int vs_set_wgrad_kernel(int mode);
/* which 5x5 split-f16 forward / data-gradient kernel vs_conv64_f16x3_fwd and the whole-path calls
 * launch: 0 = default (the persistent pipelined kernel, csrc/conv_f16x3_pk.hip), 1 = one tile per
 * workgroup (csrc/conv_f16x3.hip), 2 = persistent.  Both give bit-identical results (same K order);
 * the switch exists for A/B timing and the bitwise cross-check.  Process-wide. */
int vs_set_conv_kernel(int mode);
/* which BiLSTM recurrence / BPTT runs: 0 = default (the persistent kernels whenever all their workgroups
 * fit on the device at once, else one launch per time step), 1 = one launch per step (always the fp32 MFMA products),
 * 2 = persistent (error when the grid cannot be resident), 3 = persistent with the fp32 MFMA products whatever dims.math
 * says (A/B of the f16 / bf16 products), 4 = persistent with the flag hand-off of rounds 2-4 where the default is the tagged-data
 * hand-off (the f16 / split-f16 forward recurrence: same arithmetic).  In VS_MATH_FP32 both forms give bit-identical results.
 * Process-wide.
 * The persistent kernels keep an error word in the caller's state buffer (the first of its last 64
 * floats, both for the forward and the backward state) that becomes 1 when a bounded spin gave up
 * (a workgroup of the launch was not resident): results are then invalid. */
int vs_set_lstm_kernel(int mode);

/* Every remaining process-wide switch of the library, behind ONE call (ABI 8; the list shrank in ABI 9): the library itself reads NO
 * environment variable.  They exist for cross-checks (every on / off pair is compared by a GPU test); every value of every option gives
 * valid results unless the option says "timing ablation".  vs_set_option returns 0, or -1 for an unknown option / a value outside its range; vs_get_option returns
 * the current value (-1 for an unknown option).  Not thread-safe against concurrent calls into the library: set options before
 * work is enqueued.  (The Python package maps the environment variable named beside each option onto this call when it loads
 * the library: voicesplit_amd/_lib.py.) */
enum vs_option {
  VS_OPT_FWD_PROLOGUE = 0,    /* VS_MATH_BF16 vs_forward_train: 1 = the weight-only launches of the step (six conv weight packs, the bf16 copy of
                                 W_ih, the d-vector fold, the recurrent / head weight images) on the library's side stream beside cnn1; 0 = in
                                 line.  Default 1 (-0.15 ms per step).  Same values.  VOICESPLIT_FWD_PROLOGUE */
  VS_OPT_HEAD_LEAF_SIDE = 1,  /* 1: the head's two weight gradients and bias sums (leaves of vs_backward) on the side stream beside the BPTT;
                                 0: in front of it.  Default 1 (-0.3 ms per step).  Same values.  VOICESPLIT_HEAD_LEAF_SIDE */
  VS_OPT_FEAT_ROWS = 2,       /* whole-path eval forward (vs_forward, vs_forward_prepared): 1 = cnn8 writes the LSTM input GEMM's A operand
                                 itself (split-f16 hi / lo rows at a scale planned from the tracked |max| of its input, or bf16 rows): no fp32
                                 features in the workspace, no |max| / split passes over them; 0 = fp32 features, then the passes.  Default 1.
                                 Same values unless a lo half underflows (both scales are powers of two).  VOICESPLIT_FEAT_ROWS */
  VS_OPT_HEAD_BWD_GEMM = 3,   /* VS_MATH_BF16 vs_backward: 1 = the head's two data-gradient contractions (dfc1, dlstm_out) on the LSTM
                                 contractions' LDS-DMA kernel over bf16 copies of their operands (relu mask in the epilogue); 0 = the generic
                                 kernel (in-flight conversion).  Default 1.  Same operand roundings, fp32 summation order differs.
                                 VOICESPLIT_HEAD_BWD_GEMM */
  VS_OPT_ABLATION = 4,        /* libraries built with `make ABLATION=1` only (tools/wgrad_ablation.py, tools/split_conv_micro.py, tools/gemm_micro.py):
                                 selects a TIMING-ABLATION instance of the weight-gradient / split-f16 conv / bf16 GEMM kernels (results are
                                 meaningless).  Ignored by the product build.  VOICESPLIT_ABLATION */
  VS_OPT_DETERMINISTIC = 5,   /* VS_MATH_BF16 training step (vs_forward_train, vs_backward, vs_sisnr_loss): 1 = every partial sum that workgroups
                                 add to shared slots with fp64 atomics (BatchNorm statistics, their backward sums, cnn1's input moments,
                                 the loss head's moments) is added in WORKGROUP ORDER (the workgroups of such a launch that add to the same
                                 addresses take turns), so a rerun on the same inputs is bit-identical: masks, running statistics, loss, every gradient.
                                 0 (default) = arrival order: results agree to the last bits of fp64 sums only (SURVEY.md section 5:
                                 deterministic-rerun comparisons as the device-side sanitizer).  Costs 1.8 ms of a 46.0 ms step at B = 64
                                 (profiles/r06_experiments.md).  VOICESPLIT_DETERMINISTIC */
  VS_OPT_COUNT = 6
};
/* Round 6 (ABI 9) removed the switches whose alternative had been measured slower, with the code behind them: the NCHW eval route of the
 * fp32-class convs, the two-pass BatchNorm backward (VS_OPT_BWD_DY), the round-3 bf16 GEMM as an A/B arm and its DMA-row / band knobs,
 * the packed-fp32 epilogue builds, s_setprio / side-stream priorities, the eight-wave conv (VS_OPT_CONV8), the fused BatchNorm finalize
 * of the backward, the grid of the BatchNorm pass beside the weight gradient, the late starts of the LSTM's leaves.  The measurements
 * are in profiles/r04_*.md, profiles/r05_experiments.md and profiles/r06_experiments.md; kernels that were correct and slower are kept
 * as records under tools/attic/. */
int vs_set_option(int option, int value);
int vs_get_option(int option);

/* Did the persistent BiLSTM kernels of the last vs_forward_train / vs_backward on `tape` and / or the last vs_forward /
 * vs_bilstm_fwd on `workspace` complete?  0 = yes, 1 = a bounded spin gave up (another process took CUs away while the
 * launch ran; its output was overwritten with NaN so that no caller can use it by accident), < 0 = error.  Either
 * buffer may be NULL.  Synchronises `stream`.  The launches themselves are cooperative: a grid that cannot be resident
 * is refused by the runtime and the one-launch-per-step kernels run instead. */
int vs_lstm_status(const vs_dims* dims, const void* tape, size_t tape_bytes, const void* workspace, size_t workspace_bytes, void* stream);

/* vs_backward schedule: 1 (default) = the 64->64 weight gradients run on a second HIP stream of the library beside
 * the BatchNorm backward passes of the next layer (matrix-pipe kernel beside an HBM stream); 0 = everything in
 * order on the caller's stream.  Same kernels and summation order: results are bit-identical.  The caller's stream
 * is joined before vs_backward returns.  The side stream and its events are shared by all callers of a device, so
 * while this is on the ENQUEUE phase of concurrent vs_backward calls is serialized by a mutex (the GPU work is not).
 * Returns 0, or -1 for any other value. */
int vs_set_backward_overlap(int on);
/* BatchNorm+activation backward over rows [R][L] with channel = r % C (NCHW: R = B*C, L = T*F;
 * cnn8 feature layout: R = B*T*8, L = F).  dz may alias da.  stats: 2*C doubles, coef: 3*C floats. */
int vs_bn_act_bwd(const float* da, const float* z, float* dz, int C, long long R, int L, int act, int bn_mode,
                  const float* scale, const float* shift, const float* mean, const float* invstd,
                  float* dgamma, float* dbeta, float* dbias, double* stats, float* coef, void* stream);
/* cnn1: the same for da, z [B][64][T][F] fused with the 1x7 weight gradient dw [64][7] against the input
 * x [B][T][F] (dZ1 is consumed in registers, never written).  xpad: B*T*(F+6) floats of scratch;
 * stats: 128 doubles, coef: 192 floats, acc: 448 doubles. */
int vs_bn_act_bwd_first(const float* da, const float* z, const float* x, float* xpad, int B, int T, int F, int act, int bn_mode,
                        const float* scale, const float* shift, const float* mean, const float* invstd,
                        float* dgamma, float* dbeta, float* dbias, float* dw, double* stats, float* coef, double* acc, void* stream);
/* cnn8 */
int vs_conv_last_dgrad(const float* dz, const float* w, float* din, int B, int T, int F, void* stream);
int vs_conv_last_wgrad_blocks(void);
int vs_conv_last_wgrad(const float* dz, const float* in, float* partials /* [blocks][512] */, float* dw,
                       int B, int T, int F, void* stream);
/* cnn1: dw [64][7]; acc = 448 doubles of scratch */
int vs_conv_first_wgrad(const float* dz, const float* x, double* acc, float* dw, int B, int T, int F, void* stream);
/* general GEMM: layouts 0 = K contiguous (A[m][k], W[n][k]), 1 = K-major (A[k][m], W[k][n]);
 * gate: C = gate[m][n] > 0 ? C : 0; w_shift/w_group: K-major W row k read from row k+w_shift,
 * zero when (k % w_group)+w_shift leaves [0,w_group); splits > 1: split-K via partials [splits][M][N] */
int vs_gemm(int layout_a, int layout_w, const float* A, int lda, const float* W, int ldw, float* C, int ldc,
            int M, int N, int K, const float* bias, const float* gate, int ldg, int a_relu, int w_relu, int act,
            int accumulate, int w_shift, int w_group, int splits, float* partials, void* stream);
/* the same GEMM in VS_MATH_F16X3 arithmetic (no split-K / w_shift); scratch8 = 8 floats; A and W must
 * be dense [rows][ld] buffers, 16-byte aligned (their power-of-two scales are derived inside) */
int vs_gemm_f16x3(int layout_a, int layout_w, const float* A, int lda, const float* W, int ldw, float* C, int ldc,
                  int M, int N, int K, const float* bias, const float* gate, int ldg, int a_relu, int w_relu, int act,
                  int accumulate, float* scratch8, void* stream);
/* BiLSTM recurrence that also saves the activated gates (may alias xg) and cell states */
int vs_bilstm_recurrent_train(const float* xg, const float* packed_whh, float* state, float* out,
                              float* gates_save, float* c_save, int B, int T, int H, void* stream);
/* BPTT: gates (activated, from the call above) are overwritten with d/d(gate pre-activations) */
size_t vs_lstm_packed_t_floats(int H);
size_t vs_lstm_bwd_state_floats(int B, int H);
int vs_lstm_pack_t(const float* w_hh_fwd, const float* w_hh_bwd, float* packed_t, int H, void* stream);
int vs_bilstm_recurrent_bwd(const float* packed_t, float* state, float* gates, const float* c_all,
                            const float* dout, int B, int T, int H, void* stream);
int vs_lstm_pack_t_math(const float* w_hh_fwd, const float* w_hh_bwd, float* packed_t, int H, int math, void* stream);
int vs_bilstm_recurrent_bwd_math(const float* packed_t, float* state, float* gates, const float* c_all,
                                 const float* dout, int B, int T, int H, int math, void* stream);
int vs_sigmoid_bwd(const float* dmask, const float* mask, float* dlogits, long long n, void* stream);
/* out[g][n] = sum over the g-th block of `rows` rows of X [groups*rows][ld] */
int vs_colsum(const float* x, int ld, int groups, int rows, int N, float* out, int ldo, void* stream);

/* =============================================================================================
 * The caller side of the mask during training: train.py:95-108
 *   output = mixed*mask -> ap.torch_inv_spectrogram(output, phase)   utils/audio_processor.py:498-509
 *   loss   = SiSNR_With_Pit()(wav, target_wav, seq_len)              utils/generic_utils.py:417-474
 * in one call, together with d(loss)/d(mask).  The iSTFT runs as a GEMM against the windowed inverse
 * real-DFT basis + a gather-form overlap-add; the reference's quirks are reproduced (complex
 * spectrum = mag*exp(cos phi) + i*mag*exp(sin phi); hann(win, periodic=False) window; centre
 * trimming; means divided by seq_len; eps 1e-16; loss = 20 - mean SI-SNR).
 * ============================================================================================= */
typedef struct vs_loss_dims {
  int B, T, F;               /* spectrogram [B][T][F], F = n_fft/2 + 1                         */
  int n_fft, hop, win;       /* config.audio.voicefilter: 1200, 160, 400                        */
  float min_level_db;        /* -100  (denormalisation, utils/audio_processor.py:501-502)       */
  float ref_level_db;        /*   20                                                            */
} vs_loss_dims;

size_t vs_sisnr_workspace_bytes(const vs_loss_dims* dims);
/* mixed, mask, target, phase: [B][T][F] fp32 (target = normalised target spectrogram, phase = the
 * mixture's phase, used for both waveforms as train.py:99-100 does); seq_len: [B] int32 sample
 * counts or NULL (= hop*(T-1)); loss: one device float; dmask [B][T][F] = d(loss)/d(mask) or NULL;
 * est_wav [B][hop*(T-1)] = the estimated waveform or NULL. */
int vs_sisnr_loss(const vs_loss_dims* dims, const float* mixed, const float* mask, const float* target,
                  const float* phase, const int* seq_len, void* workspace, size_t workspace_bytes,
                  float* loss, float* dmask, float* est_wav, void* stream);

/* The other criterion of train.py:74-75: PowerLaw_Compressed_Loss (utils/generic_utils.py:353-373,
 * loss_name 'power_law_compression', the voicefilter configuration; config.json: power 0.3,
 * complex_loss_ratio 0.113).  prediction = mixed*mask (train.py:95); mixed, mask, target: n = B*T*F
 * floats; scratch: 2 doubles on the device; loss: one device float; dmask = d(loss)/d(mask) or NULL. */
int vs_powerlaw_loss(const float* mixed, const float* mask, const float* target, long long n, float power,
                     float complex_loss_ratio, double* scratch, float* loss, float* dmask, void* stream);

/* ---- audio front / back end of inference: what test.py does around the model ----
 * wav [B][hop*(T-1)] -> spec [B][T][F] (normalised dB magnitude in [0,1]) and phase [B][T][F] (may be
 * NULL): librosa.stft(n_fft, hop, win, hann, center, reflect) + amp_to_db + normalize, transposed to
 * [time, freq]  (utils/audio_processor.py:469-476, 511-514, 537-544). */
size_t vs_audio_workspace_bytes(const vs_loss_dims* dims);
int vs_wav_to_spec(const vs_loss_dims* dims, const float* wav, float* spec, float* phase,
                   void* workspace, size_t workspace_bytes, void* stream);
/* (spec * mask), phase -> wav [B][hop*(T-1)]: denormalize + db_to_amp + mag*exp(i*phase) + librosa.istft
 * (utils/audio_processor.py:478-491; est_mask*mixed_spec from utils/generic_utils.py:496).  mask may be NULL. */
int vs_spec_to_wav(const vs_loss_dims* dims, const float* spec, const float* mask, const float* phase, float* wav,
                   void* workspace, size_t workspace_bytes, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VOICESPLIT_HIP_H */
