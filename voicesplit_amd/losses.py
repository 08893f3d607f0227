"""The training-loop side of the mask: what train.py:95-108 does between
``mask = model(mixed, emb)`` and ``loss.backward()`` when ``c.loss['loss_name'] == 'si_snr'``:

    output = mixed * mask                                             train.py:95
    output = ap.torch_inv_spectrogram(output, spec_phase)             utils/audio_processor.py:498-509
    target = ap.torch_inv_spectrogram(target, spec_phase)
    loss   = SiSNR_With_Pit()(output[:, None], target[:, None], seq_len)    utils/generic_utils.py:417-474

as ONE call into libvoicesplit_hip.so (``vs_sisnr_loss``): the iSTFT as a GEMM against the windowed
inverse-DFT basis, gather overlap-add, SI-SNR moments, and the gradient w.r.t. the mask.  The
reference's ``torch_inv_spectrogram`` cannot run any more (``torchaudio.functional.istft`` was
removed from torchaudio); this is its replacement on the device, quirks included -- checked on the GPU
against the upstream lines themselves, executed with ``torch.istft`` in the removed call's place
(tests/golden/audio_upstream.npz, tests/test_gpu_audio.py).
"""
import ctypes

import torch

from . import _lib
from .ops import _dev_check, _p, _stream

_WS = {}


def loss_dims(B, T, F, audio_cfg) -> "_lib.VsLossDims":
    """audio_cfg = config.audio[config.audio['backend']] (n_fft, hop_length, win_length, min/ref_level_db)."""
    return _lib.VsLossDims(int(B), int(T), int(F), int(audio_cfg["n_fft"]), int(audio_cfg["hop_length"]),
                           int(audio_cfg["win_length"]), float(audio_cfg.get("min_level_db", -100.0)),
                           float(audio_cfg.get("ref_level_db", 20.0)))


def _workspace(d, device):
    lib = _lib.load()
    n = lib.vs_sisnr_workspace_bytes(ctypes.byref(d))
    if n == 0:
        _lib.check(-1, "vs_sisnr_workspace_bytes")
    key = torch.device(device).index
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        _WS.pop(key, None)
        ws = torch.empty(n, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


class _SiSnr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, mixed, target, phase, seq_len, d, want_wav):
        lib = _lib.load()
        for n, t in (("mask", mask), ("mixed", mixed), ("target", target), ("phase", phase)):
            _dev_check(t.detach(), n)
        if seq_len is not None:
            seq_len = seq_len.to(device=mask.device, dtype=torch.int32).contiguous()
        ws = _workspace(d, mask.device)
        loss = torch.empty((), dtype=torch.float32, device=mask.device)
        dmask = torch.empty_like(mask) if ctx.needs_input_grad[0] else None
        wav = torch.empty(d.B, d.hop * (d.T - 1), device=mask.device) if want_wav else None
        with torch.cuda.device(mask.device):
            rc = lib.vs_sisnr_loss(ctypes.byref(d), _p(mixed), _p(mask.detach()), _p(target), _p(phase), _p(seq_len), _p(ws),
                                   ws.numel(), _p(loss), _p(dmask), _p(wav), _stream())
        _lib.check(rc, "vs_sisnr_loss")
        ctx.dmask = dmask
        ctx.mark_non_differentiable(wav) if wav is not None else None
        return (loss, wav) if want_wav else loss

    @staticmethod
    def backward(ctx, grad_loss, *unused):
        return ctx.dmask * grad_loss, None, None, None, None, None, None


def sisnr_loss(mask, mixed, target, phase, seq_len, audio_cfg, return_wav: bool = False):
    """loss (0-dim tensor, differentiable w.r.t. ``mask``) of train.py:95-108 for the si_snr loss.

    mask, mixed, target, phase: [B, T, num_freq] on the GPU; seq_len: [B] sample counts or None."""
    B, T, F = mask.shape
    d = loss_dims(B, T, F, audio_cfg)
    return _SiSnr.apply(mask.contiguous(), mixed.contiguous(), target.contiguous(), phase.contiguous(), seq_len, d, return_wav)


class _PowerLaw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, mixed, target, power, ratio):
        lib = _lib.load()
        for n, t in (("mask", mask), ("mixed", mixed), ("target", target)):
            _dev_check(t.detach(), n)
        if mixed.shape != mask.shape or target.shape != mask.shape:
            raise ValueError(f"power_law_loss: shapes differ: mask {tuple(mask.shape)}, mixed {tuple(mixed.shape)}, "
                             f"target {tuple(target.shape)}")
        scratch = torch.empty(2, dtype=torch.float64, device=mask.device)
        loss = torch.empty((), dtype=torch.float32, device=mask.device)
        dmask = torch.empty_like(mask) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(mask.device):
            rc = lib.vs_powerlaw_loss(_p(mixed), _p(mask.detach()), _p(target), mask.numel(), float(power), float(ratio),
                                      _p(scratch), _p(loss), _p(dmask), _stream())
        _lib.check(rc, "vs_powerlaw_loss")
        ctx.dmask = dmask
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        return ctx.dmask * grad_loss, None, None, None, None


def power_law_loss(mask, mixed, target, power: float = 0.3, complex_loss_ratio: float = 0.113):
    """train.py:95,108 with ``criterion = PowerLaw_Compressed_Loss(power, complex_loss_ratio)``
    (utils/generic_utils.py:353-373; the voicefilter configuration): 0-dim loss, differentiable w.r.t.
    ``mask``.  mask, mixed, target: same shape, fp32, on the GPU."""
    return _PowerLaw.apply(mask.contiguous(), mixed.contiguous(), target.contiguous(), power, complex_loss_ratio)
