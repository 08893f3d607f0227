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
