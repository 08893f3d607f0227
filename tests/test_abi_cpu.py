"""CPU: the C-ABI library loads and exports every symbol include/voicesplit_hip.h declares; the
host-side mirror of the reference interface behaves (no compute calls here -- no GPU)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from oracle import reference_forward as R


def _header_functions():
    text = open(os.path.join(ROOT, "include", "voicesplit_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from voicesplit_amd import _lib
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.vs_abi_version() == _lib.ABI_VERSION


def test_library_exports_nothing_but_the_header():
    """-fvisibility=hidden: the dynamic symbol table holds the header's functions and the compiler's own
    __hip_cuid_* / fatbin bookkeeping, no internal launcher, kernel stub or helper."""
    import subprocess
    from voicesplit_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if l.strip()]
    extra = [s for s in syms if s not in set(_header_functions()) and not s.startswith("__hip_")]
    assert not extra, f"unexpected exports: {extra[:10]}"


def test_workspace_layout_and_argument_errors():
    from voicesplit_amd import _lib, ops
    lib = _lib.load()
    d = ops.make_dims(64, 301, 601, 256, 400, 600, 601)
    lay = ops.workspace_layout(d)
    assert lay.total_bytes == lib.vs_workspace_bytes(ctypes.byref(d))
    act = 64 * 64 * 301 * 601 * 4
    assert lay.act1 - lay.act0 >= act and lay.total_bytes > 2 * act
    offs = [lay.act0, lay.act1, lay.feat, lay.dvbias, lay.xg, lay.lstm_out, lay.fc1_out, *lay.conv_packed,
            lay.bn_scale, lay.bn_shift, lay.bn_stats, lay.lstm_packed, lay.lstm_state]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    # bad arguments: error code + message, no exception across the ABI
    bad = ops.make_dims(1, 10, 37, 16, 30, 40, 37)          # H not a multiple of 8
    assert lib.vs_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"multiple of 8" in lib.vs_last_error()
    with pytest.raises(_lib.VoiceSplitHipError):
        ops.workspace_layout(bad)
    # prepared eval-mode weights: a function of the model dimensions and the arithmetic, not of B or T
    pb = lib.vs_prepared_bytes(ctypes.byref(d))
    assert pb == lib.vs_prepared_bytes(ctypes.byref(ops.make_dims(1, 17, 601, 256, 400, 600, 601))) and pb % 256 == 0
    kp = (8 * 601 + 63) // 64 * 64          # VS_GEMM_KPAD
    assert 2 * 8 * 400 * kp * 2 < pb < 2 * 8 * 400 * kp * 2 + 16 * 2 ** 20        # the split W_ih halves dominate (62 MB)
    assert lib.vs_prepared_bytes(ctypes.byref(bad)) == 0
    assert lib.vs_prepare_weights(ctypes.byref(d), None, None, 0, None) != 0 and b"NULL" in lib.vs_last_error()
    assert lib.vs_set_backward_overlap(2) == -1 and lib.vs_set_backward_overlap(1) == 0
    assert lib.vs_conv64_packed_floats(5, 5) == (8 * 25 + 4) * 512   # + 4 dummy taps for the prefetch overrun
    # fp32 fragment form + room for the two f16 planes of the math-selected recurrence + a 64-float header
    assert lib.vs_lstm_packed_floats(400) == 2 * 1600 * 400 + 2 * 1600 * 400 + 64
    assert lib.vs_lstm_packed_t_floats(400) == 2 * 13 * 200 * 256 + 2 * 13 * 100 * 256 + 64
    # the *_math entry points of the recurrence refuse an unknown arithmetic and NULL buffers before touching the device
    assert lib.vs_lstm_pack_math(None, None, None, 400, 7, None) != 0 and b"unknown math" in lib.vs_last_error()
    assert lib.vs_lstm_pack_math(None, None, None, 400, 2, None) != 0 and b"NULL" in lib.vs_last_error()
    assert lib.vs_bilstm_recurrent_math(None, None, None, None, None, None, 1, 1, 8, 1, None) != 0 and b"NULL" in lib.vs_last_error()
    assert lib.vs_bilstm_recurrent_bwd_math(None, None, None, None, None, 1, 1, 8, 9, None) != 0 and b"unknown math" in lib.vs_last_error()
    assert lib.vs_nhwc_conv_last_pre(None, None, None, 2, None, None, None, None, None, 1, 1, 8, None) != 0 and b"NULL" in lib.vs_last_error()
    # the exported BatchNorm finalize refuses NULL statistics / constants and half a pair of running buffers before any launch
    assert lib.vs_bn_finalize(None, 64, 1000.0, 64, None, None, None, None, 1e-5, 0.1, None, None, None, None, None) != 0
    assert b"bn_finalize" in lib.vs_last_error()
    one = ctypes.c_void_p(256)
    assert lib.vs_bn_finalize(one, 64, 1000.0, 64, one, one, one, None, 1e-5, 0.1, one, one, None, None, None) != 0 and b"both running" in lib.vs_last_error()
    assert lib.vs_bn_finalize(one, 64, 1000.0, 64, one, one, None, None, 1e-5, 1.5, one, one, None, None, None) != 0 and b"momentum" in lib.vs_last_error()
    # the channels-last split-f16 layer (round 4): scratch = two packed f16 planes + 1 KiB of row norms / scale / plan, which every
    # mid-layer region of the workspace has room for; NULL, in-place and foreign kernel shapes are refused before any launch
    assert lib.vs_nhwc_conv_f16x3_scratch_bytes(5, 5) == 2 * 64 * 64 * 25 * 2 + 1024
    assert all(lay.conv_packed[i + 1] - lay.conv_packed[i] >= lib.vs_nhwc_conv_f16x3_scratch_bytes(5, 5) for i in range(1, 5))
    assert lay.conv_packed[1] - lay.conv_packed[0] >= lib.vs_nhwc_conv_f16x3_scratch_bytes(7, 1)
    assert lib.vs_nhwc_conv_f16x3_layer(None, None, None, None, 1, None, None, None, None, 0, None, None, None, None,
                                        1, 4, 4, 5, 5, 1, 1, None) != 0 and b"NULL" in lib.vs_last_error()
    two = ctypes.c_void_p(512)
    assert lib.vs_nhwc_conv_f16x3_layer(one, two, one, one, 1, one, one, one, one, 0, one, two, one, None,
                                        1, 4, 4, 5, 5, 1, 1, None) != 0 and b"in-place" in lib.vs_last_error()
    three, four = ctypes.c_void_p(768), ctypes.c_void_p(1024)
    assert lib.vs_nhwc_conv_f16x3_layer(one, two, one, one, 1, one, one, one, one, 0, three, four, one, None,
                                        1, 4, 4, 3, 3, 1, 1, None) != 0 and b"7x1, 5x5" in lib.vs_last_error()
    assert lib.vs_f16x3_split(None, None, None, None, 8, None) != 0 and lib.vs_f16x3_merge(None, None, None, None, 8, None) != 0
    assert lib.vs_set_lstm_kernel(3) == 0 and lib.vs_set_lstm_kernel(4) == 0 and lib.vs_set_lstm_kernel(5) != 0 and lib.vs_set_lstm_kernel(0) == 0
    # four exchange regions (the tagged hand-off of round 5), sized for the 16-wide K chunks of the f16 form (H = 24 -> 32)
    assert lib.vs_lstm_state_floats(3, 24) == 4 * 2 * 32 * 32 + 64 and lib.vs_lstm_state_floats(64, 400) == 4 * 2 * 400 * 64 + 64


def test_missing_library_fails_loudly(tmp_path):
    from voicesplit_amd import _lib
    with pytest.raises(_lib.VoiceSplitHipError, match="no PyTorch/CPU fallback"):
        _lib.load(str(tmp_path / "libvoicesplit_hip.so"))


@pytest.mark.parametrize("cls_name", ["VoiceSplit", "VoiceFilter"])
def test_module_surface_matches_reference(cls_name):
    import voicesplit_amd
    from models.voicefilter.model import VoiceFilter   # reference import paths (train.py:21-22)
    from models.voicesplit.model import VoiceSplit
    cls = {"VoiceSplit": VoiceSplit, "VoiceFilter": VoiceFilter}[cls_name]
    assert cls is getattr(voicesplit_amd, cls_name)
    dims = R.default_dims()
    torch.manual_seed(0)
    m = cls(voicesplit_amd.default_config())
    ref = R.build_state_dict(dims, 0, randomize_bn=False)
    sd = m.state_dict()
    assert list(sd) == list(ref)                         # same keys, same order
    for k in sd:
        assert sd[k].shape == ref[k].shape and sd[k].dtype == ref[k].dtype, k
        assert torch.equal(sd[k], ref[k]), k              # same default init under the same seed
    assert sum(p.numel() for p in m.parameters()) == 18876001
    m.load_state_dict(R.build_state_dict(dims, 3), strict=True)
    assert m.training and not m.eval().training
    # CPU tensors are refused, not silently computed some other way
    from voicesplit_amd._lib import VoiceSplitHipError
    with pytest.raises(VoiceSplitHipError, match="no CPU fallback"):
        m(torch.rand(1, 4, 601), torch.rand(1, 256))


@pytest.mark.skipif(not os.path.isfile("/root/reference/config.json"), reason="/root/reference not mounted")
def test_reference_config_json_is_accepted():
    from voicesplit_amd import VoiceSplit, load_config
    c = load_config("/root/reference/config.json")
    m = VoiceSplit(c)
    assert m.lstm.input_size == 8 * 601 + 256 and m.fc2.out_features == 601


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "voicesplit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle/" not in text and "reference_forward" not in text, f


def test_tensor_index_follows_the_module_tree_and_transients_stay_out_of_copies():
    """The cached (key, owning dict, name) index behind every C-ABI call is rebuilt when the module tree changes below
    the top level too, and caches / foreign views do not travel with deepcopy or pickling."""
    import copy
    import pickle
    import voicesplit_amd as V
    m = V.VoiceSplit(V.default_config(37, 16, 24, 40, 37))
    t0 = m._tensors()
    assert list(t0) == list(m.state_dict())
    old_w = m.conv[5].weight
    m.conv[5] = torch.nn.Conv2d(64, 64, kernel_size=(7, 1))           # a child of a child replaced: Sequential.__setitem__
    assert m._tensors()["conv.5.weight"] is m.conv[5].weight and m._tensors()["conv.5.weight"] is not old_w
    m.fc1.register_buffer("extra_stat", torch.zeros(3))               # a buffer registered on a child
    assert "fc1.extra_stat" in m._tensors()
    m.fc1.register_buffer("scratch", torch.zeros(3), persistent=False)
    assert "fc1.scratch" not in m._tensors()                          # state_dict() semantics: persistent buffers only
    m.lstm.weight_hh_l0 = torch.nn.Parameter(torch.zeros_like(m.lstm.weight_hh_l0))   # a replaced tensor is read through the dict
    assert m._tensors()["lstm.weight_hh_l0"] is m.lstm.weight_hh_l0
    assert [k for k, _ in m._named_params()] == [k for k, _ in m.named_parameters()]
    m.fc1.add_module("probe", torch.nn.Linear(2, 2))                  # a NEW grandchild: no existing link changes
    assert "fc1.probe.weight" in m._tensors() and list(m._tensors()) == list(m.state_dict())
    del m.fc1._modules["probe"]
    assert "fc1.probe.weight" not in m._tensors() and list(m._tensors()) == list(m.state_dict())
    # the recurrence's error word lives in the tape: with no training forward behind the module (or its tape released) the
    # status is unknown, never "OK"
    assert V.VoiceSplit(V.default_config(37, 16, 24, 40, 37)).lstm_status() is None
    # transients
    m.__dict__["_last_tape"] = ("tape", "dims")
    m.set_gradient_sink({"fc1.weight": torch.zeros_like(m.fc1.weight)})
    m.__dict__["_prepared"] = object()
    for c in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert not any(k in c.__dict__ for k in m._TRANSIENT)
        assert list(c._tensors()) == list(m._tensors()) and c._tensors()["fc2.weight"] is c.fc2.weight
    assert "_grad_sink" in m.__dict__
    m.set_gradient_sink(None)
    assert "_grad_sink" not in m.__dict__
