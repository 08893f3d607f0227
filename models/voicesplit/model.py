"""Drop-in import path of the reference: ``from models.voicesplit.model import VoiceSplit``
(train.py:22, test.py:22) resolves to the MI355X implementation."""
from voicesplit_amd.model import VoiceSplit  # noqa: F401
