"""Audio front / back end of inference on the GPU: what test.py does around the
model for the "voicefilter" audio backend,

    mixed_spec, mixed_phase = ap.get_spec_from_audio(wav)              utils/audio_processor.py:469-476
    est_mask = model(mixed_spec, emb)                                   utils/generic_utils.py:495
    est_wav  = ap.inv_spectrogram(est_mask * mixed_spec, mixed_phase)   :496-504, audio_processor.py:478-491

as calls into libvoicesplit_hip.so: the STFT / iSTFT run as one GEMM against a windowed DFT basis
plus a gather (the analysis window is 400 of the 1200 frame samples, so the dense basis is small).
"""
import ctypes

import torch

from . import _lib
from .losses import loss_dims
from .ops import _dev_check, _p, _stream

_WS = {}


def _workspace(d, device):
    lib = _lib.load()
    n = lib.vs_audio_workspace_bytes(ctypes.byref(d))
    if n == 0:
        _lib.check(-1, "vs_audio_workspace_bytes")
    key = torch.device(device).index
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        _WS.pop(key, None)
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def frames_for(n_samples: int, hop: int) -> int:
    """librosa.stft(center=True): 1 + n_samples // hop frames."""
    return 1 + n_samples // hop


def wav_to_spec(wav: torch.Tensor, audio_cfg, want_phase: bool = True):
    """wav [B, hop*(T-1)] -> (spec [B,T,F] normalised dB magnitude in [0,1], phase [B,T,F] or None)."""
    lib = _lib.load()
    _dev_check(wav, "wav")
    B, S = wav.shape
    hop = int(audio_cfg["hop_length"])
    if S % hop:
        raise ValueError(f"wav length {S} must be a multiple of hop_length {hop} (crop or pad the clip)")
    T, F = S // hop + 1, int(audio_cfg["n_fft"]) // 2 + 1
    d = loss_dims(B, T, F, audio_cfg)
    ws = _workspace(d, wav.device)
    spec = torch.empty(B, T, F, device=wav.device)
    phase = torch.empty(B, T, F, device=wav.device) if want_phase else None
    with torch.cuda.device(wav.device):
        rc = lib.vs_wav_to_spec(ctypes.byref(d), _p(wav), _p(spec), _p(phase), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "vs_wav_to_spec")
    return spec, phase


def spec_to_wav(spec: torch.Tensor, phase: torch.Tensor, audio_cfg, mask: torch.Tensor = None) -> torch.Tensor:
    """(spec * mask), phase [B,T,F] -> wav [B, hop*(T-1)] (ap.inv_spectrogram with the given phase)."""
    lib = _lib.load()
    for n, t in (("spec", spec), ("phase", phase)):
        _dev_check(t, n)
    if mask is not None:
        _dev_check(mask, "mask")
    B, T, F = spec.shape
    d = loss_dims(B, T, F, audio_cfg)
    ws = _workspace(d, spec.device)
    wav = torch.empty(B, d.hop * (T - 1), device=spec.device)
    with torch.cuda.device(spec.device):
        rc = lib.vs_spec_to_wav(ctypes.byref(d), _p(spec), _p(mask), _p(phase), _p(wav), _p(ws), ws.numel(), _stream())
    _lib.check(rc, "vs_spec_to_wav")
    return wav


def separate(model, wav: torch.Tensor, dvec: torch.Tensor, audio_cfg) -> torch.Tensor:
    """Target-speaker waveform for a batch of 3 s mixtures, all on the device:
    wav [B, hop*(T-1)], dvec [B, emb_dim] -> est_wav [B, hop*(T-1)]   (test.py's loop body)."""
    spec, phase = wav_to_spec(wav, audio_cfg)
    with torch.no_grad():
        mask = model(spec, dvec)
    return spec_to_wav(spec, phase, audio_cfg, mask=mask)
