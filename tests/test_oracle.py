"""CPU: pin the oracle (oracle/reference_forward.py) to the upstream reference.

1. against the committed golden tensors produced by the upstream nn.Module
   (oracle/make_golden.py) -- bit-for-bit;
2. against the live upstream module when /root/reference is mounted;
3. the explicit LSTM recurrence / Mish restatements against torch's own ops.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden
from oracle import reference_forward as R
from oracle._refimport import import_reference, reference_available


def _digest(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def _thin(name, arr, full):
    if not full:
        return arr
    if name == "cnn8":
        return arr[:, :, ::16, ::4]
    if name in ("lstm_out", "logits"):
        return arr[:, ::4]
    return arr


def case_tensors(g):
    sd = R.spread_logits(R.build_state_dict(g["dims"], g["seed"]), g["gain"])
    x, dvec = R.synthetic_inputs(g["B"], g["T"], g["dims"], g["seed"])
    return sd, x, dvec


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_reproduces_upstream_golden(name):
    g = load_golden(name)
    sd, x, dvec = case_tensors(g)
    assert _digest(sd) == g["sd_sha256"], "seeded weights differ from the run that made the fixture (torch RNG drift)"
    act = "mish" if g["model"] == "voicesplit" else "relu"
    bn_out = {}
    with torch.no_grad():
        out = R.forward(sd, x, dvec, act=act, training=g["training"], bn_out=bn_out)
    full = g["dims"]["num_freq"] > 100
    for k in ("cnn8", "lstm_out", "logits", "mask"):
        got = _thin(k, out[k].numpy(), full)
        assert got.shape == g[k].shape
        assert np.array_equal(got, g[k]), f"{name}:{k} max|d|={np.abs(got - g[k]).max()}"
    if g["training"]:
        for k, v in bn_out.items():
            assert np.array_equal(v.numpy(), g["after/" + k]), k


@pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted")
@pytest.mark.parametrize("model_name,act", [("voicesplit", "mish"), ("voicefilter", "relu")])
def test_oracle_matches_live_upstream(model_name, act):
    from oracle.make_golden import SMALL, make_config
    VoiceSplit, VoiceFilter, Mish, _lc, AttrDict = import_reference()
    cls = VoiceSplit if model_name == "voicesplit" else VoiceFilter
    torch.manual_seed(5)
    model = cls(make_config(AttrDict, SMALL)).eval()
    # constructor order / RNG consumption of build_state_dict == upstream __init__
    mine = R.build_state_dict(SMALL, seed=5, randomize_bn=False)
    up = model.state_dict()
    assert list(mine) == list(up)
    for k in up:
        assert torch.equal(mine[k], up[k]), k
    sd = R.spread_logits(R.build_state_dict(SMALL, seed=5), 4.0)
    model.load_state_dict(sd, strict=True)
    x, dvec = R.synthetic_inputs(2, 45, SMALL, seed=5)
    with torch.no_grad():
        ref = model(x, dvec)
        got = R.forward(sd, x, dvec, act=act)["mask"]
    assert torch.equal(ref, got)
    xs = torch.randn(1000) * 12
    assert torch.equal(Mish()(xs), R.mish(xs))


def test_explicit_lstm_matches_aten():
    dims = dict(num_freq=9, emb_dim=8, lstm_dim=16, fc1_dim=8, fc2_dim=9)
    sd = R.cast_state_dict(R.build_state_dict(dims, 3), torch.float64)
    xs = torch.randn(3, 17, 8 * 9 + 8, dtype=torch.float64)
    a = R.bilstm_aten(xs, sd)
    b = R.bilstm(xs, sd)
    assert (a - b).abs().max() < 1e-12


def test_mish_closed_form():
    # the HIP kernels use mish(x) = x * n/(n+2), n = e^x (e^x + 2); same function
    x = torch.linspace(-30, 30, 4001, dtype=torch.float64)
    u = torch.exp(torch.clamp(x, max=20.0))
    n = u * (u + 2)
    closed = torch.where(x > 20, x, x * n / (n + 2))
    assert (closed - R.mish(x)).abs().max() < 1e-12


def test_dvec_fold_identity():
    # cat(x, dvec) @ W_ih^T == x @ W_ih[:, :8F]^T + dvec @ W_ih[:, 8F:]^T
    dims = dict(num_freq=9, emb_dim=8, lstm_dim=16, fc1_dim=8, fc2_dim=9)
    sd = R.cast_state_dict(R.build_state_dict(dims, 4), torch.float64)
    W = sd["lstm.weight_ih_l0"]
    x = torch.randn(2, 5, 72, dtype=torch.float64)
    d = torch.randn(2, 8, dtype=torch.float64)
    full = torch.cat((x, d[:, None].repeat(1, 5, 1)), 2) @ W.t()
    fold = x @ W[:, :72].t() + (d @ W[:, 72:].t())[:, None]
    assert (full - fold).abs().max() < 1e-12


# ---------------------------------------------------------------------------------------------
# backward oracle (oracle/reference_backward.py)
# ---------------------------------------------------------------------------------------------
from conftest import GOLDEN_GRAD_CASES, load_golden_grads   # noqa: E402
from oracle import reference_backward as RB                  # noqa: E402


@pytest.mark.parametrize("name", GOLDEN_GRAD_CASES)
def test_backward_oracle_reproduces_upstream_gradients(name):
    """Same ops, same torch build as the fixture -> equal up to the summation order of the CPU
    kernels (which depends on the thread count): 2e-6 of each gradient's max magnitude."""
    g = load_golden_grads(name)
    sd, x, dvec = case_tensors(g)
    assert _digest(sd) == g["sd_sha256"]
    act = "mish" if g["model"] == "voicesplit" else "relu"
    w = RB.loss_weights(g["B"], g["T"], g["dims"]["fc2_dim"], g["seed"])
    stages = {} if "fwd/mask" in g else None
    grads = RB.gradients(sd, x, dvec, w, act=act, training=g["training"], stages=stages)
    assert sorted(grads) == sorted(g["grads"])
    if stages is not None:     # the metric configuration: forward tensors of the same training-mode call
        for nm, got in (("cnn8", stages["val/cnn8"][:, :, ::16, ::4]), ("lstm_out", stages["val/lstm_out"][:, ::8]),
                        ("logits", stages["val/logits"][:, ::8]), ("mask", stages["mask"][:, ::8])):
            ref = g["fwd/" + nm]
            assert np.abs(got.numpy() - ref).max() <= 2e-6 * np.abs(ref).max(), f"{name}:fwd/{nm}"
    for k, ref in g["grads"].items():
        got = RB.thin_grad(grads[k]).numpy()
        assert got.shape == ref.shape, k
        scale = max(g["gabs"][k], 1e-30)
        # conv biases feeding a batch-stat BatchNorm have an exactly-zero true gradient: what
        # autograd returns there is rounding noise (~1e-6 of the neighbouring gradients)
        if g["training"] and k.startswith("conv.") and k.endswith(".bias") and int(k.split(".")[1]) in (1, 5, 9, 13, 17, 21, 25, 28):
            continue
        assert np.abs(got - ref).max() <= 2e-6 * scale + 1e-12, f"{name}:{k}"


def test_backward_oracle_fp32_close_to_fp64_and_lstm_impls_agree():
    dims = dict(num_freq=21, emb_dim=8, lstm_dim=16, fc1_dim=24, fc2_dim=21)
    sd = R.spread_logits(R.build_state_dict(dims, 31), 4.0)
    x, dvec = R.synthetic_inputs(2, 23, dims, 31)
    w = RB.loss_weights(2, 23, 21, 31)
    g32 = RB.gradients(sd, x, dvec, w, act="mish", training=True)
    g64 = RB.gradients(sd, x, dvec, w, act="mish", training=True, dtype=torch.float64)
    g64l = RB.gradients(sd, x, dvec, w, act="mish", training=True, dtype=torch.float64, lstm_impl="loop", want_dvec=True)
    assert "speaker_embedding" in g64l
    for k in g32:
        scale = g64[k].abs().max().item()
        if scale < 1e-9:
            continue
        assert (g64[k] - g64l[k]).abs().max().item() <= 1e-10 * scale, k
        if not (k.startswith("conv.") and k.endswith(".bias")):
            assert (g32[k].double() - g64[k]).abs().max().item() <= 2e-4 * scale, k


def test_mish_gradient_closed_form():
    # HIP: mish'(x) = t + x*(2/(n+2))*(1+t)*sigmoid(x), t = n/(n+2), n = e^x(e^x+2); 1 for x > 20
    x = torch.linspace(-30, 30, 4001, dtype=torch.float64, requires_grad=True)
    R.mish(x).sum().backward()
    xd = x.detach()
    u = torch.exp(torch.clamp(xd, max=20.0))
    n = u * (u + 2)
    t = n / (n + 2)
    closed = torch.where(xd > 20, torch.ones_like(xd), t + xd * (2 / (n + 2)) * (1 + t) * (u / (1 + u)))
    assert (closed - x.grad).abs().max() < 1e-12


# ---------------------------------------------------------------------------------------------
# loss-side oracle (oracle/reference_loss.py)
# ---------------------------------------------------------------------------------------------
def test_sisnr_oracle_reproduces_upstream_class():
    import os
    from conftest import GOLDEN_DIR
    from oracle import reference_loss as RL
    z = np.load(os.path.join(GOLDEN_DIR, "sisnr_loss.npz"))
    est = torch.from_numpy(z["est"]).requires_grad_(True)
    loss = RL.sisnr_with_pit(est, torch.from_numpy(z["src"]), torch.from_numpy(z["lens"]))
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 1e-5
    assert np.abs(est.grad.numpy() - z["grad"]).max() <= 1e-6 * np.abs(z["grad"]).max()


def test_powerlaw_oracle_reproduces_upstream_class():
    """tests/golden/powerlaw_loss.npz was produced by the upstream PowerLaw_Compressed_Loss
    (oracle/make_golden.py --powerlaw), fp64, with exact-zero bins (the epsilon branch)."""
    import os
    from conftest import GOLDEN_DIR
    from oracle import reference_loss as RL
    z = np.load(os.path.join(GOLDEN_DIR, "powerlaw_loss.npz"))
    mixed, target = torch.from_numpy(z["mixed"]), torch.from_numpy(z["target"])
    for power, ratio in ((0.3, 0.113), (0.5, 1.0)):
        m = torch.from_numpy(z["mask"]).requires_grad_(True)
        loss = RL.training_loss_power_law(m, mixed, target, power, ratio)
        loss.backward()
        tag = f"p{power}_r{ratio}"
        assert abs(loss.item() - float(z["loss/" + tag])) < 1e-12
        assert np.abs(m.grad.numpy() - z["dmask/" + tag]).max() <= 1e-12 * np.abs(z["dmask/" + tag]).max()


def test_istft_restatement_is_a_windowed_inverse_dft():
    """torch_spec2wav's iSTFT == (windowed inverse real-DFT basis GEMM + overlap-add / envelope):
    the formulation the HIP kernels use, checked here in fp64 on the CPU."""
    import math
    from oracle import reference_loss as RL
    g = torch.Generator().manual_seed(4)
    B, T, F, n_fft, hop, win = 2, 7, 21, 40, 8, 16
    spec = torch.rand(B, T, F, generator=g, dtype=torch.float64)
    phase = (torch.rand(B, T, F, generator=g, dtype=torch.float64) - 0.5) * 6
    wav = RL.torch_spec2wav(spec, phase, n_fft, hop, win)
    S = hop * (T - 1)
    assert wav.shape == (B, S)
    j = torch.arange(win, dtype=torch.float64)
    w = 0.5 - 0.5 * torch.cos(2 * math.pi * j / (win - 1))
    n = (n_fft - win) // 2 + j
    k = torch.arange(F, dtype=torch.float64)
    c = torch.full((F,), 2.0, dtype=torch.float64)
    c[0] = 1.0
    c[-1] = 1.0
    ang = 2 * math.pi * torch.outer(n, k) / n_fft
    basis = torch.cat([torch.cos(ang) * c, -torch.sin(ang) * c], dim=1) / n_fft * w[:, None]      # [win][2F]
    mag = torch.pow(10.0, ((spec.clamp(0, 1) - 1) * 100 + 20) * 0.05)
    reim = torch.cat([mag * torch.exp(torch.cos(phase)), mag * torch.exp(torch.sin(phase))], dim=2)
    frames = reim @ basis.t()                                                                          # [B][T][win]
    out = torch.zeros(B, S, dtype=torch.float64)
    env = torch.zeros(S, dtype=torch.float64)
    for t in range(T):
        for jj in range(win):
            sp = hop * t - win // 2 + jj
            if 0 <= sp < S:
                out[:, sp] += frames[:, t, jj]
                env[sp] += w[jj] ** 2
    assert (out / env - wav).abs().max() < 1e-12


def test_audio_oracle_matches_an_independent_stft_istft():
    """oracle/reference_audio.py restates librosa.stft / librosa.istft (librosa is not installed, so
    the upstream functions cannot run here).  torch.stft / torch.istft with the periodic Hann window
    are an independent implementation of the same published algorithm (centre reflect padding,
    zero-padded window, window-sum-square normalisation, n_fft/2 trimmed): fp64 agreement to 1e-12
    on a round trip and on a masked (inconsistent) spectrogram pins the restatement to it."""
    from oracle import reference_audio as RA
    rng = np.random.default_rng(0)
    n_fft, hop, win = 1200, 160, 400
    y = rng.standard_normal(4800)
    D = RA.stft(y, n_fft, hop, win)
    w = torch.hann_window(win, periodic=True, dtype=torch.float64)
    Dt = torch.stft(torch.from_numpy(y), n_fft, hop_length=hop, win_length=win, window=w, center=True,
                    pad_mode="reflect", return_complex=True)
    assert D.shape == (601, 31) and np.abs(D - Dt.numpy()).max() < 1e-12
    back = RA.istft(D, hop, win)
    assert np.abs(back - torch.istft(Dt, n_fft, hop_length=hop, win_length=win, window=w, center=True).numpy()).max() < 1e-12
    assert np.abs(back - y).max() < 1e-12
    M = rng.random(D.shape)
    masked = torch.istft(Dt * torch.from_numpy(M), n_fft, hop_length=hop, win_length=win, window=w, center=True).numpy()
    assert np.abs(RA.istft(D * M, hop, win) - masked).max() < 1e-12
    # wav2spec: normalised dB magnitude in [0, 1] and the phase of the same transform
    S, ph = RA.wav2spec(y * 0.05)
    assert S.shape == ph.shape == (31, 601) and S.min() >= 0.0 and S.max() <= 1.0
    assert np.allclose(ph, np.angle(RA.stft(y * 0.05, n_fft, hop, win)).T)


def test_oracle_reproduces_upstream_mask_on_a_real_clip():
    """tests/golden/vs_real_clip.npz: a 3 s crop of one of the reference's demo mixtures through the
    voicefilter front end and the UPSTREAM VoiceSplit (make_golden.py --real; BASELINE configs[0]).
    The oracle recomputes the spectrogram from the stored waveform and must return the same mask."""
    from oracle import reference_audio as RA
    g = load_golden("vs_real_clip")
    sd = R.spread_logits(R.build_state_dict(g["dims"], g["seed"]), g["gain"])
    assert _digest(sd) == g["sd_sha256"], "seeded weights differ from the run that made the fixture (torch RNG drift)"
    spec, _ = RA.wav2spec(g["wav"].astype(np.float64))
    assert spec.shape == (301, 601) and 0.0 <= spec.min() and spec.max() <= 1.0 and 0.3 < spec.mean() < 0.8
    x = torch.from_numpy(spec.astype(np.float32))[None]
    with torch.no_grad():
        out = R.forward(sd, x, torch.from_numpy(g["dvec"]), act="mish", training=False)
    assert np.array_equal(out["mask"].numpy(), g["mask"])
    assert np.array_equal(out["logits"].numpy()[:, ::4], g["logits"])
    assert g["mask"].min() < 0.05 and g["mask"].max() > 0.95        # a mask that actually separates


def test_bf16_storage_model_is_the_pinned_oracle_plus_roundings():
    """oracle/bf16_model.py (the envelope tests/test_gpu_bf16.py holds the bf16 HIP path to): with the roundings off it IS
    the pinned backward oracle (1e-12); with them on, the parameter gradients of `sum(mask * w)` move by tenths of a
    tensor's maximum although every rounding is 2^-9 relative -- the loss is badly conditioned, not the arithmetic."""
    import torch
    from oracle import bf16_model
    from oracle import reference_backward as RB
    from oracle import reference_forward as R
    dims_d = dict(num_freq=53, emb_dim=16, lstm_dim=24, fc1_dim=40, fc2_dim=53)
    sd = R.build_state_dict(dims_d, 31)
    x, dvec = R.synthetic_inputs(3, 40, dims_d, 31)
    w = RB.loss_weights(3, 40, 53, 31)
    exact, mask = bf16_model.gradients(sd, x, dvec, w, bf16=False)
    ref = RB.gradients(sd, x, dvec, w, act="mish", training=True, dtype=torch.float64, lstm_impl="loop")
    zero = {f"conv.{i}.bias" for i in (1, 5, 9, 13, 17, 21, 25, 28)}
    for k, v in exact.items():
        if k not in zero:
            assert ((v - ref[k].double()).abs().max() / ref[k].abs().max()).item() < 1e-10, k
    ideal, mask16 = bf16_model.gradients(sd, x, dvec, w, bf16=True)
    worst = max(((ideal[k] - exact[k]).abs().max() / exact[k].abs().max()).item() for k in exact if k not in zero)
    assert 0.02 < worst < 1.5, worst                     # far above 2^-9, far below "wrong"
    assert float(((mask16 - mask) ** 2).mean()) < 1e-5   # while the mask itself barely moves


# ---- the audio legs against the UPSTREAM audio processor (round 4) ------------------------------------------------------------
def _audio_fixture():
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "audio_upstream.npz"))
    g = torch.Generator().manual_seed(int(z["mask_seed"]))
    mask = torch.rand(z["spec"].shape, generator=g, dtype=torch.float64).numpy()
    assert abs(mask.sum() - float(z["mask_sum"])) < 1e-9, "torch.rand drifted: the fixture's mask cannot be rebuilt"
    return z, mask


def test_audio_oracles_reproduce_the_upstream_audio_processor():
    """tests/golden/audio_upstream.npz = the reference's own openVoiceFilterAudioProcessor (utils/audio_processor.py:440-567) run
    by `oracle/make_golden.py --audio`: wav2spec, spec2wav with the mixture's phase, torch_spec2wav on 0.5 s of a demo mixture.
    (librosa / torchaudio.functional.istft are not obtainable: the upstream lines ran with torch.stft / torch.istft standing in for
    those two transforms -- oracle/_refimport.py.)  The numpy restatement (oracle/reference_audio.py) and the torch one
    (oracle/reference_loss.torch_spec2wav) must reproduce what the UPSTREAM code returned."""
    from oracle import reference_audio as RA
    from oracle import reference_loss as RL
    z, mask = _audio_fixture()
    spec, phase = RA.wav2spec(z["wav"].astype(np.float64))
    assert spec.shape == z["spec"].shape == (51, 601)
    assert np.abs(spec - z["spec"]).max() < 1e-9
    strong = z["spec"] > 0.2                                       # the phase of an empty bin is noise in both
    assert np.abs(np.angle(np.exp(1j * (phase - z["phase"]))))[strong].max() < 1e-6
    back = RA.spec2wav(z["spec"] * mask, z["phase"])
    assert np.abs(back - z["spec2wav"]).max() < 1e-12
    tw = RL.torch_spec2wav(torch.from_numpy(z["spec"] * mask)[None], torch.from_numpy(z["phase"])[None])[0].numpy()
    assert np.abs(tw - z["torch_spec2wav"]).max() <= 1e-6 * np.abs(z["torch_spec2wav"]).max()


@pytest.mark.skipif(not reference_available(), reason="/root/reference not mounted")
def test_audio_fixture_is_what_the_live_upstream_class_returns():
    from oracle._refimport import import_reference_audio
    z, mask = _audio_fixture()
    ap = import_reference_audio()(sample_rate=16000, n_fft=1200, num_freq=601, hop_length=160, win_length=400, preemphasis=0.97, power=1.5,
                                  min_level_db=-100.0, ref_level_db=20.0, num_mels=40, griffin_lim_iters=60)
    spec, phase = ap.wav2spec(z["wav"].astype(np.float64))
    assert np.array_equal(spec, z["spec"]) and np.array_equal(phase, z["phase"])
    assert np.abs(ap.spec2wav(z["spec"] * mask, z["phase"]) - z["spec2wav"]).max() < 1e-15
    tw = ap.torch_spec2wav(torch.from_numpy(z["spec"] * mask)[None], torch.from_numpy(z["phase"])[None])[0].numpy()
    assert np.abs(tw - z["torch_spec2wav"]).max() < 1e-12
    import sys
    assert "librosa" not in sys.modules and "torchaudio" not in sys.modules       # the stand-ins do not leak
