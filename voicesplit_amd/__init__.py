"""voicesplit_amd -- MI355X-native implementation of VoiceSplit's mask-prediction forward pass.

Public surface (mirrors the reference):

* ``VoiceSplit(config)`` / ``VoiceFilter(config)``: ``nn.Module`` classes with the reference's
  constructor, ``forward(x, speaker_embedding)`` signature and ``state_dict`` keys
  (models/voicesplit/model.py:9-89, models/voicefilter/model.py:11-90); also importable from the
  reference's own module paths ``models.voicesplit.model`` / ``models.voicefilter.model``.
* ``AttrDict`` / ``load_config``: the config object those constructors take
  (utils/generic_utils.py:560-573).
* ``ops``: stage-level entry points over the C ABI of ``libvoicesplit_hip.so``.

The compute path is the HIP library only; there is no PyTorch/CPU fallback.
"""
from .config import AttrDict, load_config, default_config  # noqa: F401
from .model import VoiceFilter, VoiceSplit  # noqa: F401

__all__ = ["VoiceSplit", "VoiceFilter", "AttrDict", "load_config", "default_config"]
