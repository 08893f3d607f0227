// The caller side of the mask during training (train.py:95-108):
//   output = mixed * mask                                            train.py:95
//   wav    = ap.torch_inv_spectrogram(output, spec_phase)            utils/audio_processor.py:498-509
//   loss   = SiSNR_With_Pit()(wav, target_wav, seq_len)              utils/generic_utils.py:417-474
// computed on the GPU together with d(loss)/d(mask), so that the training step needs no torchaudio
// (torchaudio.functional.istft, which the reference calls, no longer exists) and no autograd graph
// over ~1 GB of waveform-domain intermediates.
//
// The reference's inverse spectrogram, quirks included:
//   S   = (clamp(spec,0,1) - 1) * (-min_level_db) + ref_level_db;   mag = 10^(S/20)
//   re  = mag * exp(cos(phase)),  im = mag * exp(sin(phase))         (":507-509" multiplies by
//         torch.exp of the stacked [cos, sin] -- not mag*cos, mag*sin; reproduced as is)
//   wav = istft(re + i*im, n_fft, hop, win, hann(win, periodic=False), center=True)
// MI355X formulation: the window is `win` <= n_fft samples wide and centred in the frame, so only
// `win` of the n_fft time samples of every frame's inverse FFT survive the windowing.  The inverse
// real FFT restricted to those samples, times the window, is a dense [2*(n_fft/2+1)] x [win] matrix:
// the iSTFT is ONE GEMM on the matrix cores (19264 x 1202 x 400 at B=64) followed by a gather-form
// overlap-add (every output sample sums its <= ceil(win/hop) frames and divides by the window
// envelope).  The backward pass is the transposed GEMM and the same gather, and the SI-SNR (one
// source: PIT is the identity) reduces to six per-utterance moments in fp64.
#include <math.h>

#include "../../include/voicesplit_hip.h"
#include "vs_internal.h"

namespace {

constexpr float kLn10Over20 = 0.11512925464970229f;

struct LossShape {
  int B, T, F, n_fft, hop, win, S;   // S = hop*(T-1) samples per utterance
  int K, ldk;                        // K = 2F (re | im), ldk = K rounded up to 4
  float min_level_db, ref_level_db;
  int periodic;      // 0: hann(win, periodic=False) (torch_spec2wav :509); 1: librosa/scipy 'hann' (fftbins=True)
  int true_phase;    // 0: mag*e^{cos}, mag*e^{sin} (torch_spec2wav :506-509); 1: mag*cos, mag*sin (spec2wav :478-481)
};

// Hann window of the training loss (torch.hamming_window(win, periodic=False, 0.5, 0.5)) or of
// librosa's stft/istft (scipy get_window('hann', win, fftbins=True))
__device__ __forceinline__ float hann_w(int j, const LossShape& s) {
  const int den = s.periodic ? s.win : s.win - 1;
  return den > 0 ? 0.5f - 0.5f * cospif(2.0f * (float)j / (float)den) : 1.0f;
}
#define hann_np(j, win) hann_w(j, s)

// basis[j][k]: contribution of spectrum entry k (k < F: Re_k, else Im_{k-F}) to windowed sample j
// of a frame (time index n = (n_fft-win)/2 + j): onesided irfft weights c_k/n_fft, c = 1 for DC and
// Nyquist, 2 otherwise.
// scale2 (or NULL): {s, 1/s} of the split-f16 contractions for this operand, from its analytic bound 2 / n_fft (the rule of
// scale_from_absmax_kernel, conv_f16x3.hip: the bound lands in [2^9, 2^10))
__global__ void istft_basis_kernel(float* __restrict__ basis, LossShape s, float* __restrict__ scale2 = nullptr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx == 0 && scale2) {
    int e = 0;
    (void)frexpf(2.0f / (float)s.n_fft, &e);
    scale2[0] = ldexpf(1.f, 10 - e);
    scale2[1] = ldexpf(1.f, e - 10);
  }
  if (idx >= s.win * s.ldk) return;
  const int j = idx / s.ldk, k = idx - j * s.ldk;
  float v = 0.f;
  if (k < s.K) {
    const int kk = k < s.F ? k : k - s.F;
    const int n = (s.n_fft - s.win) / 2 + j;
    const int r = (int)(((long long)kk * n) % s.n_fft);          // exact angle reduction
    float sn, cs;
    sincospif(2.0f * (float)r / (float)s.n_fft, &sn, &cs);
    const float c = (kk == 0 || 2 * kk == s.n_fft) ? 1.0f : 2.0f;
    v = (k < s.F ? cs : -sn) * c / (float)s.n_fft * hann_np(j, s.win);
  }
  basis[idx] = v;
}

// window envelope after centring: env[s'] = sum over frames of win^2 at that sample
__global__ void istft_envelope_kernel(float* __restrict__ env, LossShape s) {
  const int sp = blockIdx.x * blockDim.x + threadIdx.x;
  if (sp >= s.S) return;
  const int off = s.win / 2;                 // sample sp sits at frame-local j = sp - hop*t + off
  float e = 0.f;
  int t_hi = (sp + off) / s.hop;
  if (t_hi > s.T - 1) t_hi = s.T - 1;
  for (int t = t_hi; t >= 0; --t) {
    const int j = sp - s.hop * t + off;
    if (j >= s.win) break;
    const float w = hann_np(j, s.win);
    e = fmaf(w, w, e);
  }
  env[sp] = e;
}

// reim[m][k] = mag * exp(cos phi) (k < F) | mag * exp(sin phi) (k >= F);  v = a*b (mixed*mask) or a
__global__ __launch_bounds__(256)
void spec_to_reim_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ phase,
                         float* __restrict__ reim, LossShape s, unsigned* __restrict__ amax = nullptr) {
  const long long n = (long long)s.B * s.T * s.F;
  float mx = 0.f;      // max |re|, |im| of what this thread wrote: the operand scale of the split-f16 contraction (vs_absmax_commit)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long m = i / s.F;
    const int f = (int)(i - m * s.F);
    float v = a[i];
    if (b) v *= b[i];
    const float db = (fminf(fmaxf(v, 0.f), 1.f) - 1.f) * (-s.min_level_db) + s.ref_level_db;
    const float mag = expf(db * kLn10Over20);
    float sn, cs;
    sincosf(phase[i], &sn, &cs);
    const float re = s.true_phase ? mag * cs : mag * expf(cs), im = s.true_phase ? mag * sn : mag * expf(sn);
    reim[m * s.ldk + f] = re;
    reim[m * s.ldk + s.F + f] = im;
    if (f == 0) for (int c = s.K; c < s.ldk; ++c) reim[m * s.ldk + c] = 0.f;      // the row's ld padding (a contraction over ldk columns reads it)
    mx = fmaxf(mx, fmaxf(fabsf(re), fabsf(im)));
  }
  vs_absmax_commit(mx, amax);
}

// ---- analysis side: wav -> frames -> (re | im) -> normalised dB magnitude + phase ---------------
// frames[m][j] = w[j] * wav_reflect[b][hop*t - win/2 + j]     (librosa.stft: center=True, reflect padding;
// only the win samples under the centred window matter)
__global__ __launch_bounds__(256)
void stft_frames_kernel(const float* __restrict__ wav, float* __restrict__ frames, LossShape s, unsigned* __restrict__ amax = nullptr) {
  const long long n = (long long)s.B * s.T * s.win;
  float mx = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long m = i / s.win;
    const int j = (int)(i - m * s.win);
    const int b = (int)(m / s.T), t = (int)(m - (long long)b * s.T);
    int idx = s.hop * t - s.win / 2 + j;
    if (idx < 0) idx = -idx;
    if (idx >= s.S) idx = 2 * (s.S - 1) - idx;
    const float v = hann_w(j, s) * wav[(size_t)b * s.S + idx];
    frames[i] = v;
    mx = fmaxf(mx, fabsf(v));
  }
  vs_absmax_commit(mx, amax);
}

// fwd_basis[k][j]: Re_k = sum_j x_j cos(2 pi k n_j / N), Im_k = -sum_j x_j sin(...), n_j = (N-win)/2 + j
__global__ void stft_basis_kernel(float* __restrict__ basis, LossShape s, float* __restrict__ scale2 = nullptr) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx == 0 && scale2) { scale2[0] = 512.f; scale2[1] = 1.f / 512.f; }      // |cos|, |sin| <= 1 = 0.5 * 2^1 -> 2^(10 - 1) (scale_from_absmax_kernel's rule)
  if (idx >= s.K * s.win) return;
  const int k = idx / s.win, j = idx - k * s.win;
  const int kk = k < s.F ? k : k - s.F;
  const int n = (s.n_fft - s.win) / 2 + j;
  const int r = (int)(((long long)kk * n) % s.n_fft);
  float sn, cs;
  sincospif(2.0f * (float)r / (float)s.n_fft, &sn, &cs);
  basis[idx] = k < s.F ? cs : -sn;
}

// spec = clip((20 log10(max(1e-5, |D|)) - ref) / -min_db, -1, 0) + 1;  phase = angle(D)
// (wav2spec, utils/audio_processor.py:469-476 with amp_to_db :537-538 and normalize :543-544)
__global__ __launch_bounds__(256)
void reim_to_features_kernel(const float* __restrict__ reim, float* __restrict__ spec, float* __restrict__ phase, LossShape s) {
  const long long n = (long long)s.B * s.T * s.F;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long m = i / s.F;
    const int f = (int)(i - m * s.F);
    const float re = reim[m * s.ldk + f], im = reim[m * s.ldk + s.F + f];
    const float mag = sqrtf(re * re + im * im);
    const float db = 20.0f * log10f(fmaxf(1e-5f, mag)) - s.ref_level_db;
    spec[i] = fminf(fmaxf(db / (-s.min_level_db), -1.0f), 0.0f) + 1.0f;
    if (phase) phase[i] = atan2f(im, re);
  }
}

// wav[b][sp] = (sum_t frames[b*T+t][sp - hop*t + win/2]) / env[sp]
__global__ __launch_bounds__(256)
void overlap_add_kernel(const float* __restrict__ frames, const float* __restrict__ env, float* __restrict__ wav, LossShape s) {
  const int sp = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (sp >= s.S) return;
  const int off = s.win / 2;
  int t_hi = (sp + off) / s.hop;
  if (t_hi > s.T - 1) t_hi = s.T - 1;
  float acc = 0.f;
  for (int t = t_hi; t >= 0; --t) {
    const int j = sp - s.hop * t + off;
    if (j >= s.win) break;
    acc += frames[((size_t)b * s.T + t) * s.win + j];
  }
  wav[(size_t)b * s.S + sp] = acc / env[sp];
}

// dframes[m][j] = dwav[b][sp] / env[sp],  sp = hop*t - win/2 + j  (0 outside the kept samples)
__global__ __launch_bounds__(256)
void overlap_add_bwd_kernel(const float* __restrict__ dwav, const float* __restrict__ env, float* __restrict__ dframes, LossShape s,
                            unsigned* __restrict__ amax = nullptr) {
  const long long n = (long long)s.B * s.T * s.win;
  float mx = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long m = i / s.win;
    const int j = (int)(i - m * s.win);
    const int b = (int)(m / s.T), t = (int)(m - (long long)b * s.T);
    const int sp = s.hop * t - s.win / 2 + j;
    const float v = (sp >= 0 && sp < s.S) ? dwav[(size_t)b * s.S + sp] / env[sp] : 0.f;
    dframes[i] = v;
    mx = fmaxf(mx, fabsf(v));
  }
  vs_absmax_commit(mx, amax);
}

// six fp64 moments per utterance: sum_all s, and over the unmasked samples: n, e, s, e*s, e^2, s^2
__global__ __launch_bounds__(256)
void sisnr_moments_kernel(const float* __restrict__ est, const float* __restrict__ src, const int* __restrict__ len,
                          double* __restrict__ mom /* [B][8] */, LossShape s) {
  const int b = blockIdx.y;
  const int L = len ? (len[b] < s.S ? (len[b] > 0 ? len[b] : 0) : s.S) : s.S;
  double a[6] = {0, 0, 0, 0, 0, 0};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < s.S; i += gridDim.x * 256) {
    const double e = est[(size_t)b * s.S + i], r = src[(size_t)b * s.S + i];
    a[0] += r;
    if (i < L) { a[1] += e; a[2] += r; a[3] += e * r; a[4] += e * e; a[5] += r * r; }
  }
  __shared__ double red[4][6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a[q] += __shfl_down(a[q], o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 6; ++q) red[threadIdx.x >> 6][q] = a[q];
  }
  __syncthreads();
  // one addition per workgroup and moment, its four waves in a fixed order (deterministic mode launches ONE workgroup per utterance:
  // then nothing about the sum depends on arrival order)
  if (threadIdx.x < 6) atomicAdd(&mom[b * 8 + threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// per utterance: SI-SNR and the three coefficients of d(loss)/d(est[t]) = m_t*(cs*zs_t + ce*ze_t - c0);
// utils/generic_utils.py:437-473 with C = 1 (one source: the permutation search is the identity)
__global__ void sisnr_finalize_kernel(const double* __restrict__ mom, const int* __restrict__ len, float* __restrict__ coef /* [B][8] */,
                                      float* __restrict__ loss, LossShape s) {
  __shared__ double acc[64];
  const int tid = threadIdx.x;
  double part = 0.0;
  const double eps = 1e-16;
  for (int b = tid; b < s.B; b += 64) {
    const double nlen = len ? (double)len[b] : (double)s.S;          // num_samples = source_lengths (as given)
    const int Lm_i = len ? (len[b] < s.S ? (len[b] > 0 ? len[b] : 0) : s.S) : s.S;
    const double Lm = Lm_i;                                          // samples the mask keeps
    const double* m = mom + b * 8;
    const double mu_s = m[0] / nlen, mu_e = m[1] / nlen;             // means over ALL samples / num_samples
    const double Sze = m[1] - Lm * mu_e, Szs = m[2] - Lm * mu_s;
    const double dot = m[3] - mu_s * m[1] - mu_e * m[2] + Lm * mu_e * mu_s;
    const double Z = m[5] - 2 * mu_s * m[2] + Lm * mu_s * mu_s;      // sum zs^2
    const double E = m[4] - 2 * mu_e * m[1] + Lm * mu_e * mu_e;      // sum ze^2
    const double es = Z + eps;
    const double P = dot * dot * Z / (es * es);
    const double c = (es + eps) / (es * es);
    const double N = E - c * dot * dot + eps;
    const double snr = P / N;
    part += 10.0 * log10(snr + eps);
    // d l_b / d ze_t = k*( dP_t/N - P/N^2 * dN_t ),  dP_t = 2 dot Z/es^2 * zs_t,  dN_t = 2 ze_t - 2 c dot zs_t
    const double k = 10.0 / (log(10.0) * (snr + eps)) * (-1.0 / s.B);      // loss = 20 - mean_b l_b
    const double gs = k * (2 * dot * Z / (es * es) / N + P / (N * N) * 2 * c * dot);
    const double ge = k * (-P / (N * N) * 2);
    // ze_t = (e_t m_t - mu_e) m_t  ->  d/de_u = m_u ( g_u - sum_t m_t g_t / num_samples )
    const double gsum = gs * Szs + ge * Sze;
    float* o = coef + b * 8;
    o[0] = (float)gs; o[1] = (float)ge; o[2] = (float)(gsum / nlen); o[3] = (float)mu_s; o[4] = (float)mu_e;
    o[5] = (float)Lm_i;
  }
  acc[tid] = part;
  __syncthreads();
  if (tid == 0) {
    double t = 0;
    for (int i = 0; i < 64; ++i) t += acc[i];
    *loss = (float)(20.0 - t / s.B);
  }
}

__global__ __launch_bounds__(256)
void sisnr_grad_kernel(const float* __restrict__ est, const float* __restrict__ src, const float* __restrict__ coef,
                       float* __restrict__ dwav, LossShape s) {
  const int b = blockIdx.y;
  const float gs = coef[b * 8], ge = coef[b * 8 + 1], g0 = coef[b * 8 + 2], mu_s = coef[b * 8 + 3], mu_e = coef[b * 8 + 4];
  const int L = (int)coef[b * 8 + 5];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < s.S; i += gridDim.x * 256) {
    float g = 0.f;
    if (i < L) {
      const float zs = src[(size_t)b * s.S + i] - mu_s, ze = est[(size_t)b * s.S + i] - mu_e;
      g = fmaf(gs, zs, fmaf(ge, ze, -g0));
    }
    dwav[(size_t)b * s.S + i] = g;
  }
}

// dmask[i] = mixed[i] * [0 <= v <= 1] * (-min_db) * ln10/20 * mag * (dre*e^{cos} + dim*e^{sin}),  v = mixed*mask
__global__ __launch_bounds__(256)
void reim_to_dmask_kernel(const float* __restrict__ mixed, const float* __restrict__ mask, const float* __restrict__ phase,
                          const float* __restrict__ dreim, float* __restrict__ dmask, LossShape s) {
  const long long n = (long long)s.B * s.T * s.F;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long m = i / s.F;
    const int f = (int)(i - m * s.F);
    const float x = mixed[i], v = x * mask[i];
    float g = 0.f;
    if (v >= 0.f && v <= 1.f) {
      const float db = (v - 1.f) * (-s.min_level_db) + s.ref_level_db;
      const float mag = expf(db * kLn10Over20);
      float sn, cs;
      sincosf(phase[i], &sn, &cs);
      const float d = dreim[m * s.ldk + f] * expf(cs) + dreim[m * s.ldk + s.F + f] * expf(sn);
      g = x * (-s.min_level_db) * kLn10Over20 * mag * d;
    }
    dmask[i] = g;
  }
}

inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

struct LossLayout { size_t basis, fbasis, env, reim_e, reim_t, frames_e, frames_t, wav_e, wav_t, dwav, mom, coef, amax, scales, total; };

// ---- PowerLaw_Compressed_Loss (utils/generic_utils.py:353-373), the criterion train.py:74-75 picks
// for loss_name == 'power_law_compression' (the voicefilter configuration):
//   o = mixed*mask + 1e-16, t = target + 1e-16, po = o^p, pt = t^p
//   loss = mean((|pt| - |po|)^2) + ratio * mean((pt - po)^2)
// One streaming pass: both sums in fp64 (two atomics per workgroup) and, since the gradient needs
// nothing from the reduction but 1/n,
//   dmask = mixed * p*o^(p-1) * ( 2(|po| - |pt|) sign(po) + 2 ratio (po - pt) ) / n
// which is what autograd gives through mul / add / pow / abs / MSELoss.  A negative o makes po NaN,
// as torch.pow does.
__global__ __launch_bounds__(256)
void powerlaw_kernel(const float* __restrict__ mixed, const float* __restrict__ mask, const float* __restrict__ target,
                     long long n, float power, float ratio, double* __restrict__ sums, float* __restrict__ dmask) {
  constexpr float kEps = 1e-16f;
  const float inv_n = (float)(1.0 / (double)n);
  double a0 = 0.0, a1 = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float mx = mixed[i];
    const float o = mx * mask[i] + kEps;
    const float t = target[i] + kEps;
    const float po = powf(o, power), pt = powf(t, power);
    const float ds = fabsf(pt) - fabsf(po), dc = pt - po;
    a0 += (double)(ds * ds);
    a1 += (double)(dc * dc);
    if (dmask) {
      const float sgn = po > 0.f ? 1.f : (po < 0.f ? -1.f : 0.f);
      const float dpo = (-2.f * ds * sgn - 2.f * ratio * dc) * inv_n;
      dmask[i] = mx * (power * powf(o, power - 1.f)) * dpo;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_down(a0, off, 64);
    a1 += __shfl_down(a1, off, 64);
  }
  __shared__ double sh[8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[2 * w] = a0; sh[2 * w + 1] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], sh[0] + sh[2] + sh[4] + sh[6]);
    atomicAdd(&sums[1], sh[1] + sh[3] + sh[5] + sh[7]);
  }
}

__global__ void powerlaw_finalize_kernel(const double* __restrict__ sums, long long n, float ratio, float* __restrict__ loss) {
  *loss = (float)(sums[0] / (double)n + (double)ratio * (sums[1] / (double)n));
}

int make_shape(const vs_loss_dims* d, LossShape* s) {
  VS_REQUIRE(d != nullptr, "loss dims is NULL");
  VS_REQUIRE(d->B > 0 && d->T > 1 && d->F > 1 && d->hop > 0 && d->win > 0, "loss: bad dims B=%d T=%d F=%d hop=%d win=%d", d->B, d->T, d->F, d->hop, d->win);
  VS_REQUIRE(d->n_fft == 2 * (d->F - 1), "loss: num_freq F=%d must be n_fft/2+1 (n_fft=%d)", d->F, d->n_fft);
  VS_REQUIRE(d->win <= d->n_fft && d->hop <= d->win, "loss: need hop <= win <= n_fft");
  VS_REQUIRE(d->B <= 65535, "loss: B too large");
  s->B = d->B; s->T = d->T; s->F = d->F; s->n_fft = d->n_fft; s->hop = d->hop; s->win = d->win;
  s->S = d->hop * (d->T - 1);
  s->K = 2 * d->F;
  s->ldk = (s->K + 3) & ~3;
  s->min_level_db = d->min_level_db; s->ref_level_db = d->ref_level_db;
  s->periodic = 0; s->true_phase = 0;
  return 0;
}

// The overlap-add divides by the window envelope.  With little or no overlap (hop close to win) a Hann
// window leaves samples whose envelope is 0 (w[0] = 0, and w[win-1] = 0 for the non-periodic form):
// torchaudio.functional.istft asserts `window_envelop.abs().min() > 1e-11` there and librosa only
// divides where the envelope exceeds tiny -- refuse such configurations instead of emitting inf/nan.
// Same formula as istft_envelope_kernel, evaluated on the host (first / last window and one steady
// period are enough: the envelope is hop-periodic in between); cached for the last shape seen.
int check_envelope(const LossShape& s) {
  static thread_local int c_key[5] = {0, 0, 0, 0, 0};
  static thread_local double c_min = 0.0;
  const int key[5] = {s.hop, s.win, s.T, s.periodic, 1};
  bool hit = true;
  for (int i = 0; i < 5; ++i) hit = hit && key[i] == c_key[i];
  if (!hit) {
    const int off = s.win / 2, den = s.periodic ? s.win : s.win - 1;
    auto env_at = [&](int sp) {
      double e = 0.0;
      int t_hi = (sp + off) / s.hop;
      if (t_hi > s.T - 1) t_hi = s.T - 1;
      for (int t = t_hi; t >= 0; --t) {
        const int j = sp - s.hop * t + off;
        if (j >= s.win) break;
        const double w = den > 0 ? 0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * (double)j / (double)den) : 1.0;
        e += w * w;
      }
      return e;
    };
    double mn = 1e300;
    const int span = s.win + s.hop;
    for (int sp = 0; sp < s.S; ++sp) {
      if (sp >= span && sp < s.S - span) { sp = s.S - span - 1; continue; }
      const double e = env_at(sp);
      if (e < mn) mn = e;
    }
    c_min = mn;
    for (int i = 0; i < 5; ++i) c_key[i] = key[i];
  }
  VS_REQUIRE(c_min > 1e-11, "istft: window envelope reaches %.3g (hop=%d win=%d %s Hann): the overlap-add would divide by zero; "
             "use hop <= win/2", c_min, s.hop, s.win, s.periodic ? "periodic" : "symmetric");
  return 0;
}

void make_layout(const LossShape& s, LossLayout* L) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  const size_t M = (size_t)s.B * s.T;
  L->basis = take((size_t)s.win * s.ldk * 4);
  L->fbasis = take((size_t)s.K * s.win * 4);     // analysis basis [K][win] (vs_wav_to_spec)
  L->env = take((size_t)s.S * 4);
  L->reim_e = take(M * s.ldk * 4);       // later reused for d(reim)
  L->reim_t = take(M * s.ldk * 4);
  L->frames_e = take(M * s.win * 4);     // later reused for d(frames)
  L->frames_t = take(M * s.win * 4);
  L->wav_e = take((size_t)s.B * s.S * 4);
  L->wav_t = take((size_t)s.B * s.S * 4);
  L->dwav = take((size_t)s.B * s.S * 4);
  L->mom = take((size_t)s.B * 8 * 8);
  L->coef = take((size_t)s.B * 8 * 4);
  L->amax = take((size_t)3 * VS_AMAX_SLOTS * 4);      // running |max| of the three fp32 operands of the split-f16 contractions (vs_sisnr_loss)
  L->scales = take(4 * 2 * 4);                        // {s, 1/s}: reim (estimate), reim (target), d(frames), basis
  L->total = off;
}

template <typename T>
inline T* at(void* ws, size_t off) { return reinterpret_cast<T*>(static_cast<char*>(ws) + off); }

}  // namespace

extern "C" {

size_t vs_sisnr_workspace_bytes(const vs_loss_dims* d) {
  LossShape s;
  if (make_shape(d, &s)) return 0;
  LossLayout L;
  make_layout(s, &L);
  return L.total;
}

int vs_sisnr_loss(const vs_loss_dims* d, const float* mixed, const float* mask, const float* target, const float* phase,
                  const int* seq_len, void* ws, size_t ws_bytes, float* loss, float* dmask, float* est_wav, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  LossShape s;
  if (int rc = make_shape(d, &s)) return rc;
  LossLayout L;
  make_layout(s, &L);
  VS_REQUIRE(mixed && mask && target && phase && loss, "sisnr_loss: NULL argument");
  if (int rc = check_envelope(s)) return rc;
  VS_REQUIRE(ws && (reinterpret_cast<uintptr_t>(ws) & 255) == 0 && ws_bytes >= L.total, "sisnr_loss: workspace too small or misaligned (%zu < %zu)", ws_bytes, L.total);
  const int M = s.B * s.T;
  const long long nspec = (long long)M * s.F;
  const unsigned gspec = (unsigned)((nspec + 255) / 256 < 16384 ? (nspec + 255) / 256 : 16384);
  float* basis = at<float>(ws, L.basis);
  float* env = at<float>(ws, L.env);
  // [r6] The three contractions of this head (two iSTFTs, one backward) run as split-f16 products (gemm_f16x3.hip: fp32-class results from
  // the f16 matrix pipe -- 0.31 + 0.32 + 0.23 ms on the fp32 pipe at B = 64 before) whenever the operands fit its
  // 32-bit offsets; their power-of-two operand scales come from |max| values the producing kernels track themselves.
  const bool split = (size_t)M * s.ldk * 4 < (1ull << 32) - 4096;
  unsigned* amax = at<unsigned>(ws, L.amax);
  float* scales = at<float>(ws, L.scales);
  hipLaunchKernelGGL(istft_basis_kernel, dim3((s.win * s.ldk + 255) / 256), dim3(256), 0, stream, basis, s, scales + 6);
  hipLaunchKernelGGL(istft_envelope_kernel, dim3((s.S + 255) / 256), dim3(256), 0, stream, env, s);
  // both spectrograms -> (re | im) rows -> windowed frames (one GEMM each) -> waveforms
  float* reim[2] = {at<float>(ws, L.reim_e), at<float>(ws, L.reim_t)};
  float* frames[2] = {at<float>(ws, L.frames_e), at<float>(ws, L.frames_t)};
  float* wav[2] = {at<float>(ws, L.wav_e), at<float>(ws, L.wav_t)};
  // (the ld padding columns are zeroed by spec_to_reim_kernel itself [r6]: the memset that did it wrote both arrays, 186 MB, for two
  // floats per row)
  if (split) VS_CHECK_HIP(hipMemsetAsync(amax, 0, (size_t)3 * VS_AMAX_SLOTS * 4, stream));
  // with the |max| commit at the end of every wave (a load and maybe an atomic: ~3 us of latency) the grid is ONE resident round of
  // long-lived workgroups -- 16384 short ones paid that latency eight times over (56 -> 100 us per launch, call 13)
  const unsigned gamax = split && gspec > 2048 ? 2048 : gspec;
  hipLaunchKernelGGL(spec_to_reim_kernel, dim3(gamax), dim3(256), 0, stream, mixed, mask, phase, reim[0], s, split ? amax : nullptr);
  hipLaunchKernelGGL(spec_to_reim_kernel, dim3(gamax), dim3(256), 0, stream, target, (const float*)nullptr, phase, reim[1], s,
                     split ? amax + VS_AMAX_SLOTS : nullptr);
  for (int q = 0; q < 2; ++q) {
    if (split) {
      if (int rc = vs_scale_from_absmax_impl(amax + q * VS_AMAX_SLOTS, VS_AMAX_SLOTS, scales + 2 * q, stream)) return rc;
      if (int rc = vs_gemm_f16x3_impl(0, 0, reim[q], s.ldk, basis, nullptr, 0x7fffffff, s.ldk, frames[q], s.win, M, s.win, s.ldk,
                                      nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, scales + 2 * q, scales + 6, stream,
                                      VS_MATH_CODE_F16X3)) return rc;
    } else
    if (int rc = vs_gemm_general_impl(0, 0, reim[q], s.ldk, basis, nullptr, 0x7fffffff, s.ldk, frames[q], s.win, M, s.win, s.K,
                                      nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, 0, 0, 1, nullptr, stream)) return rc;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((s.S + 255) / 256, s.B), dim3(256), 0, stream, frames[q], env, wav[q], s);
  }
  if (est_wav) VS_CHECK_HIP(hipMemcpyAsync(est_wav, wav[0], (size_t)s.B * s.S * 4, hipMemcpyDeviceToDevice, stream));
  // SI-SNR
  double* mom = at<double>(ws, L.mom);
  float* coef = at<float>(ws, L.coef);
  VS_CHECK_HIP(hipMemsetAsync(mom, 0, (size_t)s.B * 8 * 8, stream));
  const int gx = vs_opt(VS_OPT_DETERMINISTIC) ? 1 : (s.S + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL(sisnr_moments_kernel, dim3(gx, s.B), dim3(256), 0, stream, wav[0], wav[1], seq_len, mom, s);
  hipLaunchKernelGGL(sisnr_finalize_kernel, dim3(1), dim3(64), 0, stream, mom, seq_len, coef, loss, s);
  if (dmask) {
    float* dwav = at<float>(ws, L.dwav);
    hipLaunchKernelGGL(sisnr_grad_kernel, dim3(gx, s.B), dim3(256), 0, stream, wav[0], wav[1], coef, dwav, s);
    float* dframes = frames[0];
    const long long nfr = (long long)M * s.win;
    hipLaunchKernelGGL(overlap_add_bwd_kernel, dim3((unsigned)((nfr + 255) / 256 < (split ? 2048 : 16384) ? (nfr + 255) / 256 : (split ? 2048 : 16384))),
                       dim3(256), 0, stream, dwav, env, dframes, s, split ? amax + 2 * VS_AMAX_SLOTS : nullptr);
    float* dreim = reim[0];
    // d(reim)[M][K] = d(frames)[M][win] @ basis[win][K]
    if (split) {
      if (int rc = vs_scale_from_absmax_impl(amax + 2 * VS_AMAX_SLOTS, VS_AMAX_SLOTS, scales + 4, stream)) return rc;
      if (int rc = vs_gemm_f16x3_impl(0, 1, dframes, s.win, basis, nullptr, 0x7fffffff, s.ldk, dreim, s.ldk, M, s.K, s.win,
                                      nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, scales + 4, scales + 6, stream,
                                      VS_MATH_CODE_F16X3)) return rc;
    } else
    if (int rc = vs_gemm_general_impl(0, 1, dframes, s.win, basis, nullptr, 0x7fffffff, s.ldk, dreim, s.ldk, M, s.K, s.win,
                                      nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, 0, 0, 1, nullptr, stream)) return rc;
    hipLaunchKernelGGL(reim_to_dmask_kernel, dim3(gspec), dim3(256), 0, stream, mixed, mask, phase, dreim, dmask, s);
  }
  VS_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// audio front / back end of inference: test.py's loop around the model
//   mixed_spec, mixed_phase = ap.get_spec_from_audio(wav)        utils/audio_processor.py:469-476
//   est_wav = ap.inv_spectrogram(est_mask*mixed_spec, mixed_phase)    :478-491, generic_utils.py:496-504
// both as one GEMM against a windowed DFT basis plus a gather (frames in / overlap-add out).
// ---------------------------------------------------------------------------------------------
int vs_powerlaw_loss(const float* mixed, const float* mask, const float* target, long long n, float power,
                     float complex_loss_ratio, double* scratch, float* loss, float* dmask, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VS_REQUIRE(mixed && mask && target && scratch && loss, "powerlaw_loss: NULL argument");
  VS_REQUIRE(n > 0, "powerlaw_loss: n=%lld", n);
  VS_CHECK_HIP(hipMemsetAsync(scratch, 0, 2 * sizeof(double), stream));
  const long long want = (n + 255) / 256;
  const int grid = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(powerlaw_kernel, dim3(grid), dim3(256), 0, stream, mixed, mask, target, n, power, complex_loss_ratio, scratch, dmask);
  hipLaunchKernelGGL(powerlaw_finalize_kernel, dim3(1), dim3(1), 0, stream, scratch, n, complex_loss_ratio, loss);
  VS_LAUNCH_CHECK();
  return 0;
}

size_t vs_audio_workspace_bytes(const vs_loss_dims* d) { return vs_sisnr_workspace_bytes(d); }

int vs_wav_to_spec(const vs_loss_dims* d, const float* wav, float* spec, float* phase, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  LossShape s;
  if (int rc = make_shape(d, &s)) return rc;
  s.periodic = 1;
  LossLayout L;
  make_layout(s, &L);
  VS_REQUIRE(wav && spec, "wav_to_spec: NULL argument");
  VS_REQUIRE(s.S > s.n_fft / 2, "wav_to_spec: clip shorter than the reflect padding");
  VS_REQUIRE(ws && (reinterpret_cast<uintptr_t>(ws) & 255) == 0 && ws_bytes >= L.total, "wav_to_spec: workspace too small or misaligned (%zu < %zu)", ws_bytes, L.total);
  const int M = s.B * s.T;
  float* frames = at<float>(ws, L.frames_e);
  float* basis = at<float>(ws, L.fbasis);
  float* reim = at<float>(ws, L.reim_e);
  const long long nfr = (long long)M * s.win;
  // (the contraction as split-f16 products, as in vs_sisnr_loss [r6]; win % 4 != 0 falls to the kernel's scalar-load instance)
  const bool split = (size_t)M * s.ldk * 4 < (1ull << 32) - 4096;
  unsigned* amax = at<unsigned>(ws, L.amax);
  float* scales = at<float>(ws, L.scales);
  const unsigned gfr = (unsigned)((nfr + 255) / 256 < (split ? 2048 : 16384) ? (nfr + 255) / 256 : (split ? 2048 : 16384));
  if (split) VS_CHECK_HIP(hipMemsetAsync(amax, 0, (size_t)VS_AMAX_SLOTS * 4, stream));
  hipLaunchKernelGGL(stft_frames_kernel, dim3(gfr), dim3(256), 0, stream, wav, frames, s, split ? amax : nullptr);
  hipLaunchKernelGGL(stft_basis_kernel, dim3((s.K * s.win + 255) / 256), dim3(256), 0, stream, basis, s, scales + 6);
  if (split) {
    if (int rc = vs_scale_from_absmax_impl(amax, VS_AMAX_SLOTS, scales, stream)) return rc;
    if (int rc = vs_gemm_f16x3_impl(0, 0, frames, s.win, basis, nullptr, 0x7fffffff, s.win, reim, s.ldk, M, s.K, s.win,
                                    nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, scales, scales + 6, stream,
                                    VS_MATH_CODE_F16X3)) return rc;
  } else
  if (int rc = vs_gemm_general_impl(0, 0, frames, s.win, basis, nullptr, 0x7fffffff, s.win, reim, s.ldk, M, s.K, s.win,
                                    nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, 0, 0, 1, nullptr, stream)) return rc;
  const long long nspec = (long long)M * s.F;
  hipLaunchKernelGGL(reim_to_features_kernel, dim3((unsigned)((nspec + 255) / 256 < 16384 ? (nspec + 255) / 256 : 16384)), dim3(256), 0, stream,
                     reim, spec, phase, s);
  VS_LAUNCH_CHECK();
  return 0;
}

int vs_spec_to_wav(const vs_loss_dims* d, const float* spec, const float* mask, const float* phase, float* wav,
                   void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  LossShape s;
  if (int rc = make_shape(d, &s)) return rc;
  s.periodic = 1;
  s.true_phase = 1;
  LossLayout L;
  make_layout(s, &L);
  VS_REQUIRE(spec && phase && wav, "spec_to_wav: NULL argument");
  if (int rc = check_envelope(s)) return rc;
  VS_REQUIRE(ws && (reinterpret_cast<uintptr_t>(ws) & 255) == 0 && ws_bytes >= L.total, "spec_to_wav: workspace too small or misaligned (%zu < %zu)", ws_bytes, L.total);
  const int M = s.B * s.T;
  const long long nspec = (long long)M * s.F;
  const unsigned gspec = (unsigned)((nspec + 255) / 256 < 16384 ? (nspec + 255) / 256 : 16384);
  float* basis = at<float>(ws, L.basis);
  float* env = at<float>(ws, L.env);
  float* reim = at<float>(ws, L.reim_e);
  float* frames = at<float>(ws, L.frames_e);
  // (the contraction as split-f16 products, as in vs_sisnr_loss [r6])
  const bool split = (size_t)M * s.ldk * 4 < (1ull << 32) - 4096;
  unsigned* amax = at<unsigned>(ws, L.amax);
  float* scales = at<float>(ws, L.scales);
  hipLaunchKernelGGL(istft_basis_kernel, dim3((s.win * s.ldk + 255) / 256), dim3(256), 0, stream, basis, s, scales + 6);
  hipLaunchKernelGGL(istft_envelope_kernel, dim3((s.S + 255) / 256), dim3(256), 0, stream, env, s);
  if (split) VS_CHECK_HIP(hipMemsetAsync(amax, 0, (size_t)VS_AMAX_SLOTS * 4, stream));
  hipLaunchKernelGGL(spec_to_reim_kernel, dim3(split && gspec > 2048 ? 2048 : gspec), dim3(256), 0, stream, spec, mask, phase, reim, s,
                     split ? amax : nullptr);
  if (split) {
    if (int rc = vs_scale_from_absmax_impl(amax, VS_AMAX_SLOTS, scales, stream)) return rc;
    if (int rc = vs_gemm_f16x3_impl(0, 0, reim, s.ldk, basis, nullptr, 0x7fffffff, s.ldk, frames, s.win, M, s.win, s.ldk,
                                    nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, scales, scales + 6, stream,
                                    VS_MATH_CODE_F16X3)) return rc;
  } else
  if (int rc = vs_gemm_general_impl(0, 0, reim, s.ldk, basis, nullptr, 0x7fffffff, s.ldk, frames, s.win, M, s.win, s.K,
                                    nullptr, nullptr, nullptr, 0, 1, nullptr, 0, 0, 0, VS_ACT_NONE, 0, 0, 0, 1, nullptr, stream)) return rc;
  hipLaunchKernelGGL(overlap_add_kernel, dim3((s.S + 255) / 256, s.B), dim3(256), 0, stream, frames, env, wav, s);
  VS_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
