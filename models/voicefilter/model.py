"""Drop-in import path of the reference: ``from models.voicefilter.model import VoiceFilter``
(train.py:21, test.py:21) resolves to the MI355X implementation."""
from voicesplit_amd.model import VoiceFilter  # noqa: F401
