"""CPU oracle for the training-loop side of the mask.  TEST INFRASTRUCTURE ONLY.

Restates, in plain torch ops,
  * ``openVoiceFilterAudioProcessor.torch_spec2wav``   utils/audio_processor.py:498-509
  * ``SiSNR_With_Pit.forward`` / ``get_mask``           utils/generic_utils.py:402-474
  * ``PowerLaw_Compressed_Loss.forward``               utils/generic_utils.py:353-373
as composed by train.py:95-108.

Pinning: ``SiSNR_With_Pit`` is pinned against the upstream class (``oracle/make_golden.py --loss``
imports it from /root/reference and commits inputs/outputs to tests/golden/sisnr_loss.npz).
``power_law_compressed_loss`` is pinned the same way (tests/golden/powerlaw_loss.npz).
``torch_spec2wav`` is pinned to the upstream lines with ONE substitution (round 4): it calls
``torchaudio.functional.istft``, which no longer exists in torchaudio (and torchaudio is not in this image);
``oracle/_refimport.import_reference_audio`` executes the UPSTREAM function with that one call answered by ``torch.istft`` --
the same routine after its move into torch (irfft per frame, multiply by the zero-padded window, overlap-add, divide by the
overlap-added squared window, trim n_fft/2 on both sides) -- and ``oracle/make_golden.py --audio`` commits its output on a real
clip (tests/golden/audio_upstream.npz; tests/test_oracle.py: this restatement reproduces it to 1e-6 of the waveform's maximum,
the fp32 window of the upstream call being the difference).  Everything upstream's own -- the denormalisation, the
exp(cos) / exp(sin) spectrum, the non-periodic Hann window from ``hamming_window(win, periodic=False, alpha=.5, beta=.5)``, the
iSTFT arguments -- is therefore pinned; torchaudio's removed implementation itself is not.
"""
from itertools import permutations

import torch


def torch_spec2wav(spectrogram, phase, n_fft=1200, hop_length=160, win_length=400,
                   min_level_db=-100.0, ref_level_db=20.0):
    """utils/audio_processor.py:498-509.  spectrogram, phase: [B, T, F] -> wav [B, hop*(T-1)]."""
    spectrogram = spectrogram.transpose(2, 1)
    phase = phase.transpose(2, 1)
    S = (torch.clamp(spectrogram, 0.0, 1.0) - 1.0) * -min_level_db      # :501
    S = S + ref_level_db                                                  # :502
    mag = torch.pow(10.0, S * 0.05)                                       # :504
    ph = torch.stack([phase.cos(), phase.sin()], dim=-1).to(mag.dtype)    # :506
    m = mag.unsqueeze(-1).expand_as(ph) * torch.exp(ph)                   # :507-509  (mag*e^cos, mag*e^sin)
    window = torch.hamming_window(win_length, periodic=False, alpha=0.5, beta=0.5, dtype=mag.dtype)
    return torch.istft(torch.complex(m[..., 0], m[..., 1]), n_fft, hop_length=hop_length, win_length=win_length,
                       window=window, center=True, normalized=False, onesided=True, length=None)


def get_mask(source, source_lengths):
    """utils/generic_utils.py:402-414."""
    B, _, T = source.size()
    mask = source.new_ones((B, 1, T))
    for i in range(B):
        mask[i, :, source_lengths[i]:] = 0
    return mask


def sisnr_with_pit(estimate_source, source, source_lengths, epsilon=1e-16):
    """utils/generic_utils.py:416-474 (``estimate_source *= mask`` made out of place)."""
    assert source.size() == estimate_source.size()
    B, C, T = source.size()
    mask = get_mask(source, source_lengths)
    estimate_source = estimate_source * mask
    num_samples = source_lengths.view(-1, 1, 1).to(source.dtype)
    mean_target = torch.sum(source, dim=2, keepdim=True) / num_samples
    mean_estimate = torch.sum(estimate_source, dim=2, keepdim=True) / num_samples
    zero_mean_target = (source - mean_target) * mask
    zero_mean_estimate = (estimate_source - mean_estimate) * mask
    s_target = torch.unsqueeze(zero_mean_target, dim=1)
    s_estimate = torch.unsqueeze(zero_mean_estimate, dim=2)
    pair_wise_dot = torch.sum(s_estimate * s_target, dim=3, keepdim=True)
    s_target_energy = torch.sum(s_target ** 2, dim=3, keepdim=True) + epsilon
    pair_wise_proj = pair_wise_dot * s_target / s_target_energy
    e_noise = s_estimate - pair_wise_proj
    pair_wise_si_snr = torch.sum(pair_wise_proj ** 2, dim=3) / (torch.sum(e_noise ** 2, dim=3) + epsilon)
    pair_wise_si_snr = 10 * torch.log10(pair_wise_si_snr + epsilon)
    perms = source.new_tensor(list(permutations(range(C))), dtype=torch.long)
    index = torch.unsqueeze(perms, 2)
    perms_one_hot = source.new_zeros((*perms.size(), C)).scatter_(2, index, 1)
    snr_set = torch.einsum('bij,pij->bp', [pair_wise_si_snr, perms_one_hot])
    max_snr, _ = torch.max(snr_set, dim=1, keepdim=True)
    max_snr = max_snr / C
    return 20 - torch.mean(max_snr)


def training_loss(mask, mixed, target, phase, seq_len, **audio):
    """train.py:95-108 for loss_name == 'si_snr'."""
    output = mixed * mask                                                 # :95
    out_wav = torch_spec2wav(output, phase, **audio)                      # :99
    tgt_wav = torch_spec2wav(target, phase, **audio)                      # :100
    return sisnr_with_pit(out_wav.unsqueeze(1), tgt_wav.unsqueeze(1), seq_len), out_wav


def power_law_compressed_loss(prediction, target, power=0.3, complex_loss_ratio=0.113, epsilon=1e-16):
    """utils/generic_utils.py:353-373 (criterion of train.py:74-75 for loss_name ==
    'power_law_compression'; both MSELoss instances are the default 'mean' reduction)."""
    prediction = prediction + epsilon                                      # :363
    target = target + epsilon                                              # :364
    prediction = torch.pow(prediction, power)                              # :366
    target = torch.pow(target, power)                                      # :367
    spec_loss = torch.mean((torch.abs(target) - torch.abs(prediction)) ** 2)   # :369
    complex_loss = torch.mean((target - prediction) ** 2)                  # :370
    return spec_loss + complex_loss * complex_loss_ratio                   # :372


def training_loss_power_law(mask, mixed, target, power=0.3, complex_loss_ratio=0.113):
    """train.py:95,104-108 for loss_name == 'power_law_compression' (seq_len = None, unused)."""
    return power_law_compressed_loss(mixed * mask, target, power, complex_loss_ratio)
