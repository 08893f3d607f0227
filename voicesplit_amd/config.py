"""Config object consumed by the model constructors.

Mirror of the reference's ``AttrDict`` / ``load_config`` (utils/generic_utils.py:560-573): a dict
whose keys are also attributes, read from JSON that may carry ``//`` comments.  Only what the hot
path's constructor reads is required: ``config.audio[config.audio['backend']]['num_freq']`` and
``config.model['emb_dim'|'lstm_dim'|'fc1_dim'|'fc2_dim']`` (models/voicesplit/model.py:13,57-64).
"""
import json
import re


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def load_config(config_path: str) -> AttrDict:
    with open(config_path, "r") as fh:
        text = fh.read()
    text = re.sub(r"\\\n", "", text)
    text = re.sub(r"//.*\n", "\n", text)
    cfg = AttrDict()
    cfg.update(json.loads(text))
    return cfg


def default_config(num_freq: int = 601, emb_dim: int = 256, lstm_dim: int = 400,
                   fc1_dim: int = 600, fc2_dim: int = 601, model_name: str = "voicesplit") -> AttrDict:
    """The hot-path subset of the reference's config.json: model (lines 2, 37-42), the 'voicefilter'
    audio backend (44, 83-95), the loss (16-20) and the training hyper-parameters (21-32; batch_size
    is the reference's 2 -- BASELINE's B=64 is set by the caller)."""
    cfg = AttrDict()
    cfg.update({
        "model_name": model_name,
        "model": {"lstm_dim": lstm_dim, "fc1_dim": fc1_dim, "fc2_dim": fc2_dim, "emb_dim": emb_dim},
        "audio": {"backend": "voicefilter", "audio_len": 3,
                  "voicefilter": {"n_fft": 2 * (num_freq - 1), "num_freq": num_freq, "sample_rate": 16000,
                                  "hop_length": 160, "win_length": 400, "min_level_db": -100.0, "ref_level_db": 20.0}},
        "loss": {"loss_name": "si_snr" if model_name == "voicesplit" else "power_law_compression",
                 "power": 0.30, "complex_loss_ratio": 0.113},
        "train_config": AttrDict({"epochs": 1000, "learning_rate": 1e-2, "optimizer": "adam", "batch_size": 2, "seed": 42,
                                  "num_workers": 14, "logs_path": "../checkpoints/", "reinit_layers": None,
                                  "summary_interval": 2, "checkpoint_interval": 500}),
    })
    return cfg
