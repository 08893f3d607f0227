"""GPU: audio front / back end (vs_wav_to_spec, vs_spec_to_wav) against the numpy oracle in fp64."""
import numpy as np
import pytest
import torch

from oracle import reference_audio as RA

pytestmark = pytest.mark.gpu
AUDIO = {"n_fft": 1200, "hop_length": 160, "win_length": 400, "min_level_db": -100.0, "ref_level_db": 20.0}


def _wav(B, S, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(S) / 16000.0
    tones = sum(a * torch.sin(2 * np.pi * f * t + p) for a, f, p in [(0.05, 220.0, 0.1), (0.03, 1750.0, 1.0), (0.01, 5300.0, 2.0)])
    return (tones[None] + 0.004 * torch.randn(B, S, generator=g)).float()       # |STFT| stays below the 0 dB clip (10)


@pytest.mark.parametrize("B,S", [(1, 48000), (3, 3200)])
def test_wav_to_spec_matches_librosa_restatement(B, S):
    from voicesplit_amd import audio
    wav = _wav(B, S, 1)
    spec, phase = audio.wav_to_spec(wav.cuda(), AUDIO)
    assert spec.shape == (B, S // 160 + 1, 601)
    for b in range(B):
        rs, rp = RA.wav2spec(wav[b].double().numpy())
        assert rs.max() < 1.0                                                    # nothing clipped at the top
        # compare in the linear domain: an fp32 DFT resolves a bin to ~1e-6 of the frame's largest
        # bin, so the dB value of a deep null is only that accurate (same for librosa in complex64)
        lin = lambda s_: np.power(10.0, ((s_ - 1.0) * 100.0 + 20.0) / 20.0)
        got, ref = lin(spec[b].cpu().double().numpy()), lin(rs)
        assert (np.abs(got - ref) <= 2e-5 * ref + 3e-6 * ref.max(axis=1, keepdims=True)).all()
        assert np.median(np.abs(spec[b].cpu().double().numpy() - rs)) < 1e-6
        # the phase of a bin is only defined up to the bin's own magnitude: compare where it carries energy
        strong = ref > 1e-2 * ref.max()
        d = np.angle(np.exp(1j * (phase[b].cpu().double().numpy() - rp)))
        assert np.abs(d[strong]).max() < 1e-3


def test_spec_to_wav_and_round_trip():
    from voicesplit_amd import audio
    B, S = 2, 48000
    wav = _wav(B, S, 2)
    spec, phase = audio.wav_to_spec(wav.cuda(), AUDIO)
    g = torch.Generator().manual_seed(3)
    mask = torch.rand(B, 301, 601, generator=g).cuda()
    out = audio.spec_to_wav(spec, phase, AUDIO, mask=mask)
    assert out.shape == (B, S)
    for b in range(B):
        ref = RA.spec2wav((spec[b] * mask[b]).cpu().double().numpy(), phase[b].cpu().double().numpy())
        assert np.abs(out[b].cpu().double().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    # analysis -> synthesis with mask = 1 gives the waveform back (dB clipping at -100 dB aside)
    back = audio.spec_to_wav(spec, phase, AUDIO)
    err = (back.cpu() - wav).abs().max() / wav.abs().max()
    assert err < 2e-3, err


def test_separate_runs_end_to_end():
    import voicesplit_amd as V
    from oracle import reference_forward as R
    from voicesplit_amd import audio
    m = V.VoiceSplit(V.default_config()).cuda().eval()
    wav = _wav(2, 48000, 4).cuda()
    dvec = R.synthetic_inputs(2, 301, R.default_dims(), 4)[1].cuda()
    est = audio.separate(m, wav, dvec, AUDIO)
    assert est.shape == wav.shape and torch.isfinite(est).all()


def test_real_clip_waveform_to_mask_matches_upstream():
    """BASELINE configs[0] on the device: the stored 3 s crop of a reference demo mixture -> GPU STFT
    front end -> VoiceSplit -> mask, against the mask the UPSTREAM module produced from the oracle's
    spectrogram (tests/golden/vs_real_clip.npz).  (a) the model alone on the oracle's spectrogram at
    the contract tolerance; (b) the GPU front end against the oracle's; (c) end to end: the fp32 STFT
    moves -100 dB bins by ~1e-4 of the normalised scale, which the network amplifies a little."""
    from conftest import load_golden
    import voicesplit_amd as V
    from voicesplit_amd import audio
    from oracle import reference_forward as R
    g = load_golden("vs_real_clip")
    d = g["dims"]
    m = V.VoiceSplit(V.default_config(d["num_freq"], d["emb_dim"], d["lstm_dim"], d["fc1_dim"], d["fc2_dim"]))
    m.load_state_dict(R.spread_logits(R.build_state_dict(d, g["seed"]), g["gain"]), strict=True)
    m = m.cuda().eval()
    dvec = torch.from_numpy(g["dvec"]).cuda()
    spec_ref, phase_ref = RA.wav2spec(g["wav"].astype(np.float64))
    ref = g["mask"].astype(np.float64)
    with torch.no_grad():
        mask_a = m(torch.from_numpy(spec_ref.astype(np.float32))[None].cuda(), dvec).cpu().numpy()
    assert np.abs(mask_a - ref).max() / np.abs(ref).max() < 1e-4
    assert ((mask_a - ref) ** 2).mean() < 1e-4
    acfg = V.default_config().audio["voicefilter"]
    spec_gpu, phase_gpu = audio.wav_to_spec(torch.from_numpy(g["wav"])[None].cuda(), acfg)
    # linear-domain comparison as in test_wav_to_spec_matches_librosa_restatement (an fp32 DFT resolves
    # a bin to ~1e-6 of the frame's largest bin; real speech has bins at both dB clips)
    lin = lambda s_: np.power(10.0, ((s_ - 1.0) * 100.0 + 20.0) / 20.0)
    got, want = lin(spec_gpu[0].cpu().double().numpy()), lin(spec_ref)
    assert (np.abs(got - want) <= 2e-5 * want + 3e-6 * want.max(axis=1, keepdims=True)).all()
    assert np.median(np.abs(spec_gpu[0].cpu().double().numpy() - spec_ref)) < 1e-6
    with torch.no_grad():
        mask_c = m(spec_gpu, dvec).cpu().numpy()
    assert ((mask_c - ref) ** 2).mean() < 1e-4                                  # BASELINE: mask MSE <= 1e-4
    assert np.quantile(np.abs(mask_c - ref), 0.999) < 1e-2


def test_gpu_audio_legs_against_the_upstream_audio_processor():
    """The HIP front / back end and the training-side iSTFT against what the UPSTREAM openVoiceFilterAudioProcessor returned on
    0.5 s of a reference demo mixture (tests/golden/audio_upstream.npz, made by `oracle/make_golden.py --audio` from
    utils/audio_processor.py:469-509 with torch.stft / torch.istft standing in for librosa's and torchaudio's removed transform):
    vs_wav_to_spec vs wav2spec, vs_spec_to_wav vs spec2wav with the mixture's phase, vs_sisnr_loss's estimated waveform vs
    torch_spec2wav (the exp(cos) / exp(sin) spectrum and the non-periodic Hann window of :498-509)."""
    import os
    from conftest import GOLDEN_DIR
    from voicesplit_amd import audio, losses
    z = np.load(os.path.join(GOLDEN_DIR, "audio_upstream.npz"))
    g = torch.Generator().manual_seed(int(z["mask_seed"]))
    mask = torch.rand(z["spec"].shape, generator=g, dtype=torch.float64)
    assert abs(float(mask.sum()) - float(z["mask_sum"])) < 1e-9
    wav = torch.from_numpy(z["wav"])[None].cuda()
    spec, phase = audio.wav_to_spec(wav, AUDIO)
    lin = lambda s_: np.power(10.0, ((s_ - 1.0) * 100.0 + 20.0) / 20.0)
    got, ref = lin(spec[0].cpu().double().numpy()), lin(z["spec"])
    assert (np.abs(got - ref) <= 2e-5 * ref + 3e-6 * ref.max(axis=1, keepdims=True)).all()
    assert np.median(np.abs(spec[0].cpu().double().numpy() - z["spec"])) < 1e-6
    strong = ref > 1e-2 * ref.max()
    d = np.angle(np.exp(1j * (phase[0].cpu().double().numpy() - z["phase"])))
    assert np.abs(d[strong]).max() < 1e-3
    # back end on the UPSTREAM spectrogram / phase / mask
    s32, p32, m32 = (torch.from_numpy(a.astype(np.float32))[None].cuda() for a in (z["spec"], z["phase"], mask.numpy()))
    out = audio.spec_to_wav(s32, p32, AUDIO, mask=m32)[0].cpu().double().numpy()
    assert np.abs(out - z["spec2wav"]).max() <= 3e-5 * np.abs(z["spec2wav"]).max()
    # the loss head's waveform: mixed * mask -> torch_spec2wav (train.py:95-99)
    tgt = torch.rand(1, *z["spec"].shape, generator=torch.Generator().manual_seed(1)).cuda()
    _loss, est = losses.sisnr_loss(m32, s32, tgt, p32, None, AUDIO, return_wav=True)
    est = est[0].cpu().double().numpy()
    assert np.abs(est - z["torch_spec2wav"]).max() <= 3e-5 * np.abs(z["torch_spec2wav"]).max()
