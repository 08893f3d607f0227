"""Import the upstream reference (read-only at /root/reference) in the build container.

Only used by ``oracle/make_golden.py`` and by CPU tests that are skipped when the
tree is absent (it never exists on the GPU box).  ``utils/generic_utils.py:6,10,14``
imports librosa and mir_eval at module import time; neither is installed here and
neither is touched by the model forward, so empty stub modules are registered
first.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VOICESPLIT_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "voicesplit", "model.py"))


def import_reference():
    """Returns (VoiceSplit, VoiceFilter, Mish, load_config, AttrDict) from upstream."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True          # never write __pycache__ into the read-only tree
    import importlib.util as _ilu
    stub_names = ("librosa", "librosa.util", "mir_eval", "mir_eval.separation")
    # stub only what is genuinely absent, and take the stubs out again afterwards (a later
    # `import librosa` in the same process must not silently get an empty module)
    stubbed = []
    for name in stub_names:
        top = name.split(".")[0]
        if name not in sys.modules and (top in stubbed or _ilu.find_spec(top) is None):
            sys.modules[name] = types.ModuleType(name)
            stubbed.append(name)
    if "librosa.util" in stubbed:
        sys.modules["librosa"].util = sys.modules["librosa.util"]
    if "mir_eval.separation" in stubbed:
        sys.modules["mir_eval"].separation = sys.modules["mir_eval.separation"]
        sys.modules["mir_eval.separation"].bss_eval_sources = lambda *a, **k: None
    # the repo's own drop-in `models/` package would shadow upstream's: load upstream
    # under private names straight from the files.
    import importlib.util

    def load(modname, relpath):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REFERENCE_ROOT, relpath))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    saved = {k: sys.modules.get(k) for k in ("utils", "utils.generic_utils")}
    try:
        pkg = types.ModuleType("utils")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "utils")]
        sys.modules["utils"] = pkg
        gu = load("utils.generic_utils", "utils/generic_utils.py")
        vs = load("_upstream_voicesplit_model", "models/voicesplit/model.py")
        vf = load("_upstream_voicefilter_model", "models/voicefilter/model.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for name in stubbed:
            sys.modules.pop(name, None)
    return vs.VoiceSplit, vf.VoiceFilter, gu.Mish, gu.load_config, gu.AttrDict


# ---------------------------------------------------------------------------------------------
# the audio processor of the reference (utils/audio_processor.py:440-567), executed with stand-ins for the
# third-party routines that cannot be had here
# ---------------------------------------------------------------------------------------------
def _standin_librosa():
    """A module object that answers exactly the librosa calls openVoiceFilterAudioProcessor makes, through their
    documented equivalents in torch (this image has no librosa):
      librosa.stft(y, n_fft, hop_length, win_length)      -> torch.stft(window=periodic Hann(win_length) centred in n_fft,
                                                              center=True, pad_mode='reflect')   [librosa 0.6 defaults]
      librosa.istft(D, hop_length, win_length)            -> torch.istft(same window, center=True)
      librosa.filters.mel                                 -> zeros (constructor only; the mel path is not on the hot path)
    What such a run pins is every line of the REFERENCE's own arithmetic and every argument it passes; the two transforms
    themselves are torch's (tests/test_oracle.py pins oracle/reference_audio.py's restatement of them to 1e-12)."""
    import numpy as np
    import torch
    lib = types.ModuleType("librosa")

    def _win(win_length, dtype):
        return torch.hann_window(win_length, periodic=True, dtype=dtype)

    def stft(y, n_fft=2048, hop_length=None, win_length=None, window="hann", center=True, pad_mode="reflect"):
        assert window == "hann" and center and pad_mode == "reflect"
        win_length = win_length or n_fft
        hop_length = hop_length or win_length // 4
        t = torch.from_numpy(np.asarray(y, dtype=np.float64))
        return torch.stft(t, n_fft, hop_length=hop_length, win_length=win_length, window=_win(win_length, t.dtype), center=True,
                          pad_mode="reflect", return_complex=True).numpy()

    def istft(stft_matrix, hop_length=None, win_length=None, window="hann", center=True, length=None):
        assert window == "hann" and center
        n_fft = 2 * (stft_matrix.shape[0] - 1)
        win_length = win_length or n_fft
        hop_length = hop_length or win_length // 4
        t = torch.from_numpy(np.asarray(stft_matrix, dtype=np.complex128))
        return torch.istft(t, n_fft, hop_length=hop_length, win_length=win_length, window=_win(win_length, torch.float64), center=True,
                           length=length).numpy()

    lib.stft, lib.istft = stft, istft
    lib.core = types.ModuleType("librosa.core")
    lib.core.stft = stft
    lib.filters = types.ModuleType("librosa.filters")
    lib.filters.mel = lambda sr, n_fft, n_mels=128, **kw: np.zeros((n_mels, 1 + n_fft // 2))
    lib.util = types.ModuleType("librosa.util")
    return lib


def _standin_torchaudio():
    """torchaudio.functional.istft as utils/audio_processor.py:509 calls it (torchaudio <= 0.6: a REAL tensor whose last dimension
    is (re, im)) -> torch.istft, which is that function moved into torch (torchaudio 0.7 removed it in favour of it)."""
    import torch
    ta = types.ModuleType("torchaudio")
    ta.functional = types.ModuleType("torchaudio.functional")

    def istft(stft_matrix, n_fft, hop_length=None, win_length=None, window=None, center=True, pad_mode="reflect", normalized=False,
              onesided=True, length=None):
        assert stft_matrix.shape[-1] == 2
        return torch.istft(torch.view_as_complex(stft_matrix.contiguous()), n_fft, hop_length=hop_length, win_length=win_length,
                           window=window, center=center, normalized=normalized, onesided=onesided, length=length)

    ta.functional.istft = istft
    return ta


def import_reference_audio():
    """The UPSTREAM ``openVoiceFilterAudioProcessor`` class (utils/audio_processor.py:440), importable here: librosa, torchaudio and
    soundfile are absent from this image (and torchaudio.functional.istft from every current torchaudio), so the module is executed
    with the stand-ins above registered for the duration of the import.  Returns the class; the stand-ins stay reachable through
    the module's globals only."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    import importlib.util
    lib, ta = _standin_librosa(), _standin_torchaudio()
    stand = {"librosa": lib, "librosa.core": lib.core, "librosa.filters": lib.filters, "librosa.util": lib.util,
             "torchaudio": ta, "torchaudio.functional": ta.functional, "soundfile": types.ModuleType("soundfile"),
             "mir_eval": types.ModuleType("mir_eval"), "mir_eval.separation": types.ModuleType("mir_eval.separation")}
    stand["mir_eval"].separation = stand["mir_eval.separation"]
    stand["mir_eval.separation"].bss_eval_sources = lambda *a, **k: None
    ua = types.ModuleType("utils.audio")                    # utils/audio.py (the waveglow back end) is not on the path
    ua.WaveGlowSTFT = type("WaveGlowSTFT", (), {})
    keys = list(stand) + ["utils", "utils.audio", "utils.generic_utils", "utils.audio_processor"]
    saved = {k: sys.modules.get(k) for k in keys}
    try:
        for k, v in stand.items():
            sys.modules[k] = v                              # (registered for the duration of the import only: `saved` is restored below)
        pkg = types.ModuleType("utils")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "utils")]
        sys.modules["utils"] = pkg
        sys.modules["utils.audio"] = ua

        def load(modname, relpath):
            spec = importlib.util.spec_from_file_location(modname, os.path.join(REFERENCE_ROOT, relpath))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[modname] = mod
            spec.loader.exec_module(mod)
            return mod

        load("utils.generic_utils", "utils/generic_utils.py")
        apm = load("utils.audio_processor", "utils/audio_processor.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return apm.openVoiceFilterAudioProcessor
