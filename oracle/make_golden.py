"""Generate tests/golden/*.npz from the UPSTREAM reference module (build container only).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

Each case instantiates the reference ``VoiceSplit`` / ``VoiceFilter``
(models/voicesplit/model.py:9, models/voicefilter/model.py:11) from a config
``AttrDict`` (utils/generic_utils.py:560-563), loads a seeded state_dict, runs
``model(x, dvec)`` (the call made by train.py:94) and records the returned mask
plus the outputs of ``model.conv`` / ``model.lstm`` / ``model.fc2`` captured by
forward hooks.  Weights are NOT stored (5x 64x64x5x5 conv kernels are 2 MB even
for the tiny cases); they are re-derived by ``reference_forward.build_state_dict``
from the recorded seed, and the fixture carries a SHA-256 of the state_dict bytes
so RNG drift is reported as such and not as a parity failure.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import reference_forward as R            # noqa: E402
from oracle._refimport import import_reference      # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")

SMALL = dict(num_freq=37, emb_dim=16, lstm_dim=24, fc1_dim=40, fc2_dim=37)

# name, model, dims, B, T, seed, training, gain
CASES = [
    ("vs_small_eval",   "voicesplit",  SMALL, 3, 50, 11, False, 6.0),
    ("vf_small_eval",   "voicefilter", SMALL, 5, 33, 12, False, 6.0),
    ("vs_small_train",  "voicesplit",  SMALL, 4, 40, 13, True,  6.0),
    ("vs_short_T",      "voicesplit",  SMALL, 2, 20, 14, False, 6.0),   # T < dilation halo (2*16)
    ("vs_T1",           "voicesplit",  SMALL, 1, 1,  15, False, 6.0),   # single frame
    ("vs_full_b1",      "voicesplit",  R.default_dims(), 1, 301, 0, False, 8.0),
    ("vf_full_b1",      "voicefilter", R.default_dims(), 1, 301, 1, False, 8.0),
]


def make_config(AttrDict, dims):
    c = AttrDict()
    c.update({
        "model_name": "voicesplit",
        "audio": {"backend": "voicefilter", "voicefilter": {"num_freq": dims["num_freq"]}},
        "model": {k: dims[k] for k in ("lstm_dim", "fc1_dim", "fc2_dim", "emb_dim")},
    })
    return c


def state_dict_digest(sd) -> str:
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def thin(name, arr, full):
    """Full-size cases keep strided slices of the big intermediates."""
    if not full:
        return arr
    if name == "cnn8":
        return arr[:, :, ::16, ::4]
    if name in ("lstm_out", "logits"):
        return arr[:, ::4]
    return arr


def main():
    VoiceSplit, VoiceFilter, _Mish, _load_config, AttrDict = import_reference()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    for name, model_name, dims, B, T, seed, training, gain in CASES:
        sd = R.spread_logits(R.build_state_dict(dims, seed), gain)
        x, dvec = R.synthetic_inputs(B, T, dims, seed)
        cls = VoiceSplit if model_name == "voicesplit" else VoiceFilter
        model = cls(make_config(AttrDict, dims))
        model.load_state_dict(sd, strict=True)
        model.train(training)
        grabbed = {}
        hooks = [
            model.conv.register_forward_hook(lambda m, i, o: grabbed.__setitem__("cnn8", o.detach())),
            model.lstm.register_forward_hook(lambda m, i, o: grabbed.__setitem__("lstm_out", o[0].detach())),
            model.fc2.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.detach())),
        ]
        with torch.no_grad():
            mask = model(x, dvec)
        for h in hooks:
            h.remove()
        full = dims["num_freq"] > 100
        out = {
            "model": np.array(model_name), "B": B, "T": T, "seed": seed,
            "training": training, "gain": gain,
            "dims": np.array([dims[k] for k in ("num_freq", "emb_dim", "lstm_dim", "fc1_dim", "fc2_dim")]),
            "sd_sha256": np.array(state_dict_digest(sd)),
            "torch_version": np.array(torch.__version__),
            "mask": mask.numpy(),
        }
        for k, v in grabbed.items():
            out[k] = thin(k, v.numpy(), full)
        if training:   # BN buffers after the step (momentum 0.1, unbiased var)
            after = model.state_dict()
            for k in after:
                if "running_" in k or "num_batches" in k:
                    out["after/" + k] = after[k].numpy()
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez(path, **out)
        lg = grabbed["logits"]
        print(f"{name}: mask[{mask.min():.3f},{mask.max():.3f}] logits std {lg.std():.3f} "
              f"-> {os.path.getsize(path) / 1e3:.0f} kB")


# gradient fixtures: name, model, dims, B, T, seed, training (BatchNorm mode), gain
GRAD_CASES = [
    ("vs_small_train_grads", "voicesplit",  SMALL, 4, 40, 13, True,  6.0),
    ("vf_small_train_grads", "voicefilter", SMALL, 3, 35, 16, True,  6.0),
    ("vs_small_evalbn_grads", "voicesplit", SMALL, 2, 50, 17, False, 6.0),   # frozen BatchNorm (model.eval())
    ("vs_full_b1_grads",     "voicesplit",  R.default_dims(), 1, 301, 1, True, 8.0),
    # the configuration BASELINE.json's metric is quoted on (full size, several utterances, batch-stat
    # BatchNorm, train.py:84,94-110): forward intermediates, BN buffers after the step and every gradient
    ("vs_full_b8_train_grads", "voicesplit", R.default_dims(), 8, 301, 3, True, 8.0),
]


def main_grads():
    """d(loss)/d(parameters) of the UPSTREAM modules for loss = (mask * w).sum(), the graph that
    train.py:94-110 differentiates (with the audio-domain loss replaced by a fixed upstream
    gradient w, see oracle/reference_backward.py)."""
    from oracle import reference_backward as RB
    VoiceSplit, VoiceFilter, _Mish, _load_config, AttrDict = import_reference()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for name, model_name, dims, B, T, seed, training, gain in GRAD_CASES:
        if only and name not in only:
            continue
        sd = R.spread_logits(R.build_state_dict(dims, seed), gain)
        x, dvec = R.synthetic_inputs(B, T, dims, seed)
        w = RB.loss_weights(B, T, dims["fc2_dim"], seed)
        cls = VoiceSplit if model_name == "voicesplit" else VoiceFilter
        model = cls(make_config(AttrDict, dims))
        model.load_state_dict(sd, strict=True)
        model.train(training)
        full_batch = dims["num_freq"] > 100 and B > 1
        grabbed = {}
        hooks = []
        if full_batch:
            hooks = [
                model.conv.register_forward_hook(lambda m, i, o: grabbed.__setitem__("cnn8", o.detach())),
                model.lstm.register_forward_hook(lambda m, i, o: grabbed.__setitem__("lstm_out", o[0].detach())),
                model.fc2.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.detach())),
            ]
        mask = model(x, dvec)
        for h in hooks:
            h.remove()
        (mask * w).sum().backward()
        out = {
            "model": np.array(model_name), "B": B, "T": T, "seed": seed,
            "training": training, "gain": gain,
            "dims": np.array([dims[k] for k in ("num_freq", "emb_dim", "lstm_dim", "fc1_dim", "fc2_dim")]),
            "sd_sha256": np.array(state_dict_digest(sd)),
            "torch_version": np.array(torch.__version__),
            "mask_sum": np.array(float(mask.detach().double().sum())),
        }
        if full_batch:     # thinned forward tensors + the BatchNorm buffers after this one training-mode forward
            out["fwd/cnn8"] = grabbed["cnn8"].numpy()[:, :, ::16, ::4]
            out["fwd/lstm_out"] = grabbed["lstm_out"].numpy()[:, ::8]
            out["fwd/logits"] = grabbed["logits"].numpy()[:, ::8]
            out["fwd/mask"] = mask.detach().numpy()[:, ::8]
            after = model.state_dict()
            for k in after:
                if "running_" in k or "num_batches" in k:
                    out["after/" + k] = after[k].numpy()
        for k, p in model.named_parameters():
            out["grad/" + k] = RB.thin_grad(p.grad.detach()).numpy()
            out["gabs/" + k] = np.array(float(p.grad.detach().abs().max()))
        # ReLU kinks: elements of a ReLU input within 5e-5 (relative) of zero.  An implementation
        # whose forward lands on the other side of one of them differentiates a different branch;
        # the GPU test checks the signs here first (oracle/reference_backward.relu_gates).
        act = "mish" if model_name == "voicesplit" else "relu"
        with torch.no_grad():
            stages = RB.forward_with_graph(sd, x, dvec, act, training)
        assert (stages["mask"] - mask.detach()).abs().max() < 2e-5, "oracle forward differs from upstream"
        for nm in RB.relu_inputs(act):
            v = stages[nm].reshape(-1)
            idx = (v.abs() < 5e-5 * v.abs().max()).nonzero().reshape(-1)
            out["fragile_idx/" + nm] = idx.numpy()
            out["fragile_val/" + nm] = v[idx].numpy()
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez(path, **out)
        print(f"{name}: {len(list(model.parameters()))} gradients -> {os.path.getsize(path) / 1e3:.0f} kB")


def main_bf16_envelope():
    """The IDEAL-bf16 gradients of the metric configuration's fixture (vs_full_b8_train_grads: full size, 8 utterances,
    batch-statistics BatchNorm): oracle/bf16_model.py -- the reference graph with bf16 rounding injected at the storage
    points of VS_MATH_BF16 and nothing else -- on the same seeded inputs, thinned exactly like the upstream gradients of
    that fixture.  tests/test_gpu_bf16.py holds the HIP path to the envelope these define against the UPSTREAM gradients
    (how far a perfect implementation of bf16 storage is from the exact result ON THIS FIXTURE).  Needs no reference
    import; the carrier arithmetic is float32 (see bf16_model.gradients) and the run takes ~25 GB and several minutes."""
    from oracle import bf16_model
    from oracle import reference_backward as RB
    name, model_name, dims, B, T, seed, training, gain = [c for c in GRAD_CASES if c[0] == "vs_full_b8_train_grads"][0]
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    sd = R.spread_logits(R.build_state_dict(dims, seed), gain)
    up = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    assert str(up["sd_sha256"]) == state_dict_digest(sd), "RNG drift: the upstream fixture was made from other weights"
    x, dvec = R.synthetic_inputs(B, T, dims, seed)
    w = RB.loss_weights(B, T, dims["fc2_dim"], seed)
    ideal, mask = bf16_model.gradients(sd, x, dvec, w, act="mish", bf16=True, dtype=torch.float32)
    out = {"of": np.array(name), "sd_sha256": np.array(state_dict_digest(sd)), "torch_version": np.array(torch.__version__),
           "carrier": np.array("float32"), "fwd/mask": mask.numpy()[:, ::8].astype(np.float32)}
    worst_err, worst_cos = 0.0, 1.0
    for k, gk in ideal.items():
        t = RB.thin_grad(gk.detach()).double().numpy()
        out["grad/" + k] = t.astype(np.float32)
        ref = up["grad/" + k].astype(np.float64)
        if k in {f"conv.{i}.bias" for i in (1, 5, 9, 13, 17, 21, 25, 28)}:
            continue                               # conv bias in front of a batch-statistics BatchNorm: the true gradient is
                                                   # exactly zero (the upstream fp32 value is rounding noise of order 1e-9)
        err = float(np.abs(t - ref).max() / float(up["gabs/" + k]))
        cos = float((t @ ref) / max(np.linalg.norm(t) * np.linalg.norm(ref), 1e-300))
        worst_err, worst_cos = max(worst_err, err), min(worst_cos, cos)
        print(f"  {k:32s} ideal err {err:.3f} cos {cos:.4f}")
    mse = float(((out["fwd/mask"].astype(np.float64) - up["fwd/mask"]) ** 2).mean())
    path = os.path.join(GOLDEN_DIR, name.replace("_grads", "_bf16_ideal") + ".npz")
    np.savez(path, **out)
    print(f"{os.path.basename(path)}: worst ideal err {worst_err:.3f}, worst cosine {worst_cos:.4f}, mask MSE {mse:.2e} "
          f"-> {os.path.getsize(path) / 1e3:.0f} kB")


def main_audio():
    """The audio legs either side of the model, through the UPSTREAM class openVoiceFilterAudioProcessor
    (utils/audio_processor.py:440-567) executed by oracle/_refimport.import_reference_audio -- i.e. the reference's own lines for
    wav2spec (:469-476), spec2wav with the mixture's phase (:483-491) and torch_spec2wav (:498-509: denormalisation, the
    exp(cos) / exp(sin) spectrum, the non-periodic Hann window, the iSTFT arguments), with librosa.stft / librosa.istft /
    torchaudio.functional.istft (absent from this image; the last one removed from torchaudio itself) stood in by torch.stft /
    torch.istft.  Input: 0.5 s of one of the reference's demo mixtures and a seeded mask (not stored: torch.rand from the recorded seed)."""
    from scipy.io import wavfile
    from oracle._refimport import REFERENCE_ROOT, import_reference_audio
    AP = import_reference_audio()
    cfg = dict(sample_rate=16000, n_fft=1200, num_freq=601, hop_length=160, win_length=400, preemphasis=0.97, power=1.5,
               min_level_db=-100.0, ref_level_db=20.0, num_mels=40, griffin_lim_iters=60)          # config.json:83-95
    ap = AP(**cfg)
    path = os.path.join(REFERENCE_ROOT, "datasets", "LibriSpeech", "audios_demo", "2_speakers", "noisy",
                        "1701-141760-0023.251-136532-0023.wav")
    sr, wav = wavfile.read(path)
    assert sr == 16000 and wav.dtype == np.float32
    wav = np.ascontiguousarray(wav[24000:24000 + 8000])                  # 0.5 s
    spec, phase = ap.wav2spec(wav.astype(np.float64))                   # [51, 601] each
    g = torch.Generator().manual_seed(123)
    mask = torch.rand(spec.shape, generator=g, dtype=torch.float64).numpy()
    wav_np = ap.spec2wav(spec * mask, phase)                            # numpy iSTFT with the mixture's phase (test.py's path)
    wav_t = ap.torch_spec2wav(torch.from_numpy(spec * mask)[None], torch.from_numpy(phase)[None])[0].numpy()   # train.py:99
    out = {"wav": wav, "mask_seed": np.array(123), "mask_sum": np.array(mask.sum()), "spec": spec, "phase": phase, "spec2wav": wav_np,
           "torch_spec2wav": wav_t,
           "torch_version": np.array(torch.__version__),
           "source": np.array("datasets/LibriSpeech/audios_demo/2_speakers/noisy/1701-141760-0023.251-136532-0023.wav[24000:32000]")}
    p = os.path.join(GOLDEN_DIR, "audio_upstream.npz")
    np.savez_compressed(p, **out)
    print(f"audio_upstream: spec [{spec.min():.3f}, {spec.max():.3f}], |spec2wav| <= {np.abs(wav_np).max():.3f}, "
          f"|torch_spec2wav| <= {np.abs(wav_t).max():.3f} -> {os.path.getsize(p) / 1e3:.0f} kB")


def main_loss():
    """Pins oracle/reference_loss.sisnr_with_pit to the upstream SiSNR_With_Pit
    (utils/generic_utils.py:416-474): seeded waveforms in, loss and d(loss)/d(estimate) out."""
    _vs, _vf, _mish, load_config, _ad = import_reference()
    SiSNR_With_Pit = load_config.__globals__["SiSNR_With_Pit"]     # utils/generic_utils.py module namespace
    g = torch.Generator().manual_seed(99)
    B, T = 4, 2000
    est = torch.randn(B, 1, T, generator=g)
    src = 0.6 * est + 0.8 * torch.randn(B, 1, T, generator=g) + 0.05
    lens = torch.tensor([2000, 1500, 1999, 777])
    e = est.clone().requires_grad_(True)
    loss = SiSNR_With_Pit()(e * 1.0, src.clone(), lens)      # the upstream forward masks its input in place
    loss.backward()
    path = os.path.join(GOLDEN_DIR, "sisnr_loss.npz")
    np.savez(path, est=est.numpy(), src=src.numpy(), lens=lens.numpy(), loss=np.array(loss.item()), grad=e.grad.numpy(),
             torch_version=np.array(torch.__version__))
    print(f"sisnr_loss: loss {loss.item():.6f} -> {os.path.getsize(path) / 1e3:.0f} kB")


def main_powerlaw():
    """Pins oracle/reference_loss.power_law_compressed_loss to the upstream PowerLaw_Compressed_Loss
    (utils/generic_utils.py:353-373) as train.py:95,108 calls it: prediction = mixed * mask, target =
    the clean spectrogram, both normalised to [0, 1] (exact zeros included: the epsilon branch)."""
    _vs, _vf, _mish, load_config, _ad = import_reference()
    PowerLaw = load_config.__globals__["PowerLaw_Compressed_Loss"]
    g = torch.Generator().manual_seed(123)
    B, T, F = 2, 9, 31
    mixed = torch.rand(B, T, F, generator=g, dtype=torch.float64)
    target = (mixed * torch.rand(B, T, F, generator=g, dtype=torch.float64)).clamp(0, 1)
    mask = torch.sigmoid(torch.randn(B, T, F, generator=g, dtype=torch.float64))
    mixed[0, 0, :5] = 0.0            # silent bins: prediction = 0 + 1e-16
    target[0, 1, :5] = 0.0
    out = {}
    for power, ratio in ((0.3, 0.113), (0.5, 1.0)):
        m = mask.clone().requires_grad_(True)
        loss = PowerLaw(power, ratio)(mixed * m, target, None)
        loss.backward()
        tag = f"p{power}_r{ratio}"
        out["loss/" + tag] = np.array(loss.item())
        out["dmask/" + tag] = m.grad.numpy()
    path = os.path.join(GOLDEN_DIR, "powerlaw_loss.npz")
    np.savez(path, mixed=mixed.numpy(), target=target.numpy(), mask=mask.numpy(), torch_version=np.array(torch.__version__), **out)
    print(f"powerlaw_loss: {[float(v) for k, v in out.items() if k.startswith('loss/')]} -> {os.path.getsize(path) / 1e3:.0f} kB")


def main_real():
    """BASELINE configs[0] / SURVEY.md 8(c) "realistic input": a 3 s crop of one of the reference's own
    two-speaker demo mixtures (datasets/LibriSpeech/audios_demo/2_speakers/noisy, 16 kHz float32 mono)
    -> the voicefilter front end (oracle/reference_audio.wav2spec = utils/audio_processor.py:469-476)
    -> the UPSTREAM VoiceSplit in eval mode.  The fixture keeps the waveform crop (192 kB) and the mask;
    the spectrogram is recomputed from the waveform by whoever reads it."""
    from scipy.io import wavfile
    from oracle import reference_audio as RA
    from oracle._refimport import REFERENCE_ROOT
    VoiceSplit, _vf, _mish, _lc, AttrDict = import_reference()
    path = os.path.join(REFERENCE_ROOT, "datasets", "LibriSpeech", "audios_demo", "2_speakers", "noisy",
                        "1701-141760-0023.251-136532-0023.wav")
    sr, wav = wavfile.read(path)
    assert sr == 16000 and wav.dtype == np.float32 and wav.ndim == 1
    wav = np.ascontiguousarray(wav[16000:16000 + 48000])          # seconds 1..4
    spec, _phase = RA.wav2spec(wav.astype(np.float64))
    x = torch.from_numpy(spec.astype(np.float32))[None]             # [1, 301, 601]
    dims, seed, gain = R.default_dims(), 2, 8.0
    sd = R.spread_logits(R.build_state_dict(dims, seed), gain)
    g = torch.Generator().manual_seed(seed)
    dvec = torch.randn(1, dims["emb_dim"], generator=g)
    dvec = dvec / dvec.norm(dim=1, keepdim=True)                    # an L2-normalised d-vector (no GE2E encoder here)
    model = VoiceSplit(make_config(AttrDict, dims))
    model.load_state_dict(sd, strict=True)
    model.eval()
    grabbed = {}
    h = model.fc2.register_forward_hook(lambda m, i, o: grabbed.__setitem__("logits", o.detach()))
    with torch.no_grad():
        mask = model(x, dvec)
    h.remove()
    out = {"model": np.array("voicesplit"), "B": 1, "T": 301, "seed": seed, "training": False, "gain": gain,
           "dims": np.array([dims[k] for k in ("num_freq", "emb_dim", "lstm_dim", "fc1_dim", "fc2_dim")]),
           "sd_sha256": np.array(state_dict_digest(sd)), "torch_version": np.array(torch.__version__),
           "wav": wav, "dvec": dvec.numpy(), "mask": mask.numpy(), "logits": grabbed["logits"].numpy()[:, ::4],
           "source": np.array("datasets/LibriSpeech/audios_demo/2_speakers/noisy/1701-141760-0023.251-136532-0023.wav[16000:64000]")}
    p = os.path.join(GOLDEN_DIR, "vs_real_clip.npz")
    np.savez(p, **out)
    print(f"vs_real_clip: spec[{spec.min():.3f},{spec.max():.3f}] mean {spec.mean():.3f} mask[{mask.min():.3f},{mask.max():.3f}] "
          f"-> {os.path.getsize(p) / 1e3:.0f} kB")


def main_demo_clips():
    """tests/golden/demo_clips.npz: four 3 s crops of the reference's demo mixtures (datasets/LibriSpeech/audios_demo/2_speakers/
    noisy = the mixture, enhanced = the target speaker) as int16 -- data for the real-audio training test of the bf16 configuration
    (tests/test_gpu_trainer.py: overfit at the reference's own Adam lr 1e-2, config.json:23-25).  A fixture is data: only samples."""
    from scipy.io import wavfile
    from oracle._refimport import REFERENCE_ROOT
    root = os.path.join(REFERENCE_ROOT, "datasets/LibriSpeech/audios_demo/2_speakers")
    names = ["1701-141760-0023.251-136532-0023.wav", "1988-147956-0028.84-121123-0026.wav",          # (the clips of >= 4.1 s)
             "2078-142845-0028.4153-186222-0000.wav", "4831-18525-0005.3576-138058-0010.wav"]
    mixed, target = [], []
    for n in names:
        sr, m = wavfile.read(os.path.join(root, "noisy", n))
        sr2, t = wavfile.read(os.path.join(root, "enhanced", n))
        assert sr == sr2 == 16000 and m.dtype == np.float32 and t.dtype == np.float32
        lo = 16000
        mixed.append(np.clip(np.round(m[lo:lo + 48000] * 32767.0), -32768, 32767).astype(np.int16))
        target.append(np.clip(np.round(t[lo:lo + 48000] * 32767.0), -32768, 32767).astype(np.int16))
    p = os.path.join(GOLDEN_DIR, "demo_clips.npz")
    np.savez_compressed(p, mixed=np.stack(mixed), target=np.stack(target), sample_rate=np.array(16000),
                        source=np.array([f"datasets/LibriSpeech/audios_demo/2_speakers/{{noisy,enhanced}}/{n}[16000:64000]" for n in names]))
    print(f"demo_clips: {len(names)} clips -> {os.path.getsize(p) / 1e3:.0f} kB")


if __name__ == "__main__":
    if "--demo-clips" in sys.argv:
        main_demo_clips()
    elif "--bf16-envelope" in sys.argv:
        main_bf16_envelope()
    elif "--audio" in sys.argv:
        main_audio()
    elif "--real" in sys.argv:
        main_real()
    elif "--powerlaw" in sys.argv:
        main_powerlaw()
    elif "--loss" in sys.argv:
        main_loss()
    elif "--grads" in sys.argv:
        main_grads()
    else:
        main()
