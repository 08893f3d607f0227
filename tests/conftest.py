import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # gpu tests never run silently on a CPU-only host
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    dims = dict(zip(("num_freq", "emb_dim", "lstm_dim", "fc1_dim", "fc2_dim"), (int(v) for v in g["dims"])))
    g["dims"] = dims
    for k in ("B", "T", "seed"):
        g[k] = int(g[k])
    g["training"] = bool(g["training"])
    g["gain"] = float(g["gain"])
    g["model"] = str(g["model"])
    g["sd_sha256"] = str(g["sd_sha256"])
    return g


GOLDEN_CASES = ["vs_small_eval", "vf_small_eval", "vs_small_train", "vs_short_T", "vs_T1",
                "vs_full_b1", "vf_full_b1"]

GOLDEN_GRAD_CASES = ["vs_small_train_grads", "vf_small_train_grads", "vs_small_evalbn_grads", "vs_full_b1_grads",
                     "vs_full_b8_train_grads"]


def load_golden_grads(name):
    """Gradient fixture made by `oracle/make_golden.py --grads` from the upstream modules."""
    g = load_golden(name)
    g["grads"] = {k[len("grad/"):]: g[k] for k in list(g) if k.startswith("grad/")}
    g["gabs"] = {k[len("gabs/"):]: float(g[k]) for k in list(g) if k.startswith("gabs/")}
    return g
