"""CPU oracle for the audio front / back end of inference.  TEST INFRASTRUCTURE ONLY.

Restates, in numpy, ``openVoiceFilterAudioProcessor`` of utils/audio_processor.py:
  * ``wav2spec``          :469-476  (stft :511-514, amp_to_db :537-538, normalize :543-544)
  * ``spec2wav`` w/ phase :483-491  (denormalize :546-547, db_to_amp :540-541, istft_phase :478-481)

**Pinned to the upstream call sites, with the third-party transforms stood in** (round 4): librosa (0.6-era ``librosa.stft`` /
``librosa.istft``) is not installed in this image, so ``oracle/_refimport.import_reference_audio`` executes the UPSTREAM class
with ``torch.stft`` / ``torch.istft`` answering those two calls; ``oracle/make_golden.py --audio`` commits what the upstream lines
return on a real clip (tests/golden/audio_upstream.npz) and tests/test_oracle.py holds this restatement to it (1e-9 on the
spectrogram, 1e-12 on the waveform).  What stays outside the pin is librosa's own implementation of the two transforms; their
published algorithm is restated here and cross-checked against torch's independent implementation to 1e-12:
``stft``: reflect-pad n_fft//2, frames of n_fft at stride hop, times the periodic Hann window of
win_length zero-padded (centred) to n_fft, rfft;  ``istft``: irfft per frame, times the same padded
window, overlap-add, divide by the overlap-added squared window where it exceeds ``tiny``, trim
n_fft//2 on both sides.
"""
import numpy as np


def _padded_hann(win, n_fft):
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win) / win)       # scipy get_window('hann', win, fftbins=True)
    lpad = (n_fft - win) // 2
    return np.pad(w, (lpad, n_fft - win - lpad))


def stft(y, n_fft, hop, win):
    """librosa.stft(y, n_fft, hop_length, win_length) -> complex [1 + n_fft/2, 1 + len(y)//hop]."""
    w = _padded_hann(win, n_fft)
    yp = np.pad(y, n_fft // 2, mode="reflect")
    T = 1 + (len(yp) - n_fft) // hop
    frames = np.stack([yp[t * hop:t * hop + n_fft] * w for t in range(T)], axis=1)
    return np.fft.rfft(frames, axis=0)


def istft(D, hop, win):
    """librosa.istft(D, hop_length, win_length) -> real [hop * (T - 1)]."""
    n_fft = 2 * (D.shape[0] - 1)
    T = D.shape[1]
    w = _padded_hann(win, n_fft)
    y = np.zeros(n_fft + hop * (T - 1))
    wss = np.zeros_like(y)
    for t in range(T):
        y[t * hop:t * hop + n_fft] += w * np.fft.irfft(D[:, t], n=n_fft)
        wss[t * hop:t * hop + n_fft] += w ** 2
    nz = wss > np.finfo(np.float64).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:len(y) - n_fft // 2]


def wav2spec(y, n_fft=1200, hop=160, win=400, min_level_db=-100.0, ref_level_db=20.0):
    """:469-476 -> (S [T, F] in [0,1], phase [T, F])."""
    D = stft(y, n_fft, hop, win)                                           # :470
    S = 20.0 * np.log10(np.maximum(1e-5, np.abs(D))) - ref_level_db        # :471, :537-538
    S = np.clip(S / -min_level_db, -1.0, 0.0) + 1.0                        # :474, :543-544
    return S.T, np.angle(D).T                                              # :474-475


def spec2wav(spectrogram, phase, hop=160, win=400, min_level_db=-100.0, ref_level_db=20.0):
    """:483-491 with a phase -> wav."""
    spectrogram, phase = spectrogram.T, phase.T                            # :486
    S = (np.clip(spectrogram, 0.0, 1.0) - 1.0) * -min_level_db             # :546-547
    S = np.power(10.0, (S + ref_level_db) * 0.05)                          # :490, :540-541
    return istft(S * np.exp(1j * phase), hop, win)                         # :478-481
