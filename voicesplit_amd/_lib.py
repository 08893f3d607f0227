"""ctypes binding of libvoicesplit_hip.so (C ABI declared in include/voicesplit_hip.h).

Plain pointers and sizes only -- no torch types cross this boundary.  torch is used by callers
for device memory and streams; it must be imported (and HIP initialised) before the library is
loaded so that both resolve the same ``libamdhip64`` already mapped into the process.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvoicesplit_hip.so")

ACT_RELU, ACT_MISH, ACT_NONE, ACT_SIGMOID = 0, 1, 2, 3
BN_EVAL, BN_TRAIN = 0, 1
MATH_FP32, MATH_F16X3, MATH_BF16 = 0, 1, 2
PROF_SLOTS = 29
PROF_NAMES = ("cnn1", "cnn2", "cnn3", "cnn4", "cnn5", "cnn6", "cnn7", "cnn8", "lstm_gemm", "lstm_rec", "head",
              "fwd_bn", "bwd_head", "bwd_lstm_rec", "bwd_lstm_gemm", "bwd_bn",
              "wgrad_cnn2", "wgrad_cnn3", "wgrad_cnn4", "wgrad_cnn5", "wgrad_cnn6", "wgrad_cnn7",
              "dgrad_cnn2", "dgrad_cnn3", "dgrad_cnn4", "dgrad_cnn5", "dgrad_cnn6", "dgrad_cnn7", "bwd_edge")
ABI_VERSION = 9


class VsDims(Structure):
    _fields_ = [(n, c_int) for n in ("B", "T", "F", "E", "H", "FC1", "FC2", "math")]


class VsConvLayer(Structure):
    _fields_ = [(n, c_void_p) for n in ("weight", "bias", "bn_weight", "bn_bias",
                                        "bn_running_mean", "bn_running_var")]


class VsParams(Structure):
    _fields_ = [
        ("conv", VsConvLayer * 8),
        ("w_ih", c_void_p * 2), ("w_hh", c_void_p * 2), ("b_ih", c_void_p * 2), ("b_hh", c_void_p * 2),
        ("fc1_w", c_void_p), ("fc1_b", c_void_p), ("fc2_w", c_void_p), ("fc2_b", c_void_p),
    ]


class VsWsLayout(Structure):
    _fields_ = [
        ("total_bytes", c_size_t), ("act0", c_size_t), ("act1", c_size_t), ("feat", c_size_t),
        ("dvbias", c_size_t), ("xg", c_size_t), ("lstm_out", c_size_t), ("fc1_out", c_size_t),
        ("conv_packed", c_size_t * 6), ("bn_scale", c_size_t), ("bn_shift", c_size_t),
        ("bn_stats", c_size_t), ("lstm_packed", c_size_t), ("lstm_state", c_size_t), ("conv_scales", c_size_t),
        ("gemm_scales", c_size_t),
    ]


class VsLossDims(Structure):
    _fields_ = [("B", c_int), ("T", c_int), ("F", c_int), ("n_fft", c_int), ("hop", c_int), ("win", c_int),
                ("min_level_db", c_float), ("ref_level_db", c_float)]


class VsConvLayerGrad(Structure):
    _fields_ = [(n, c_void_p) for n in ("weight", "bias", "bn_weight", "bn_bias")]


class VsGrads(Structure):
    _fields_ = [
        ("conv", VsConvLayerGrad * 8),
        ("w_ih", c_void_p * 2), ("w_hh", c_void_p * 2), ("b_ih", c_void_p * 2), ("b_hh", c_void_p * 2),
        ("fc1_w", c_void_p), ("fc1_b", c_void_p), ("fc2_w", c_void_p), ("fc2_b", c_void_p),
        ("dvec", c_void_p),
        ("leaves_event", c_void_p),
    ]


class VsTapeLayout(Structure):
    _fields_ = [
        ("total_bytes", c_size_t), ("z", c_size_t * 7), ("a", c_size_t * 7), ("z8", c_size_t), ("feat", c_size_t),
        ("bn_scale", c_size_t), ("bn_shift", c_size_t), ("bn_mean", c_size_t), ("bn_invstd", c_size_t),
        ("gates", c_size_t), ("cstate", c_size_t), ("lstm_out", c_size_t), ("fc1_out", c_size_t),
        ("dlogits", c_size_t), ("dfc1", c_size_t), ("dlstm_out", c_size_t), ("dsum", c_size_t), ("dfeat", c_size_t),
        ("grad0", c_size_t), ("grad1", c_size_t),
        ("dvbias", c_size_t), ("conv_packed", c_size_t * 6), ("pack_tmp", c_size_t), ("lstm_packed", c_size_t),
        ("lstm_packed_t", c_size_t), ("lstm_state", c_size_t), ("lstm_bwd_state", c_size_t), ("consts", c_size_t),
        ("bn_stats", c_size_t), ("bn_coef", c_size_t), ("first_acc", c_size_t), ("colsum_tmp", c_size_t),
        ("partials", c_size_t), ("conv_scales", c_size_t), ("gemm_scales", c_size_t), ("lstm_bf16", c_size_t),
        ("det_turn", c_size_t),
        ("conv_packed_t", c_size_t * 6),
    ]


# name -> (restype, argtypes); must list every symbol include/voicesplit_hip.h declares
_P = c_void_p
SIGNATURES = {
    "vs_abi_version": (c_int, []),
    "vs_last_error": (c_char_p, []),
    "vs_profile_begin": (c_int, [c_int]),
    "vs_profile_end": (c_int, [POINTER(c_float), POINTER(c_int)]),
    "vs_workspace_layout": (c_int, [POINTER(VsDims), POINTER(VsWsLayout)]),
    "vs_workspace_bytes": (c_size_t, [POINTER(VsDims)]),
    "vs_forward": (c_int, [POINTER(VsDims), POINTER(VsParams), _P, _P, c_int, c_int, _P, c_size_t, _P, _P]),
    "vs_prepared_bytes": (c_size_t, [POINTER(VsDims)]),
    "vs_prepare_weights": (c_int, [POINTER(VsDims), POINTER(VsParams), _P, c_size_t, _P]),
    "vs_forward_prepared": (c_int, [POINTER(VsDims), POINTER(VsParams), _P, c_size_t, _P, _P, c_int, _P, c_size_t, _P, _P]),
    "vs_conv_stack_fwd": (c_int, [POINTER(VsDims), POINTER(VsParams), _P, c_int, c_int, _P, c_size_t, _P, _P]),
    "vs_bilstm_fwd": (c_int, [POINTER(VsDims), POINTER(VsParams), _P, _P, _P, c_size_t, _P, _P]),
    "vs_head_fwd": (c_int, [POINTER(VsDims), POINTER(VsParams), _P, _P, c_size_t, _P, _P, _P]),
    "vs_bn_fold": (c_int, [_P, _P, _P, _P, _P, c_float, c_int, _P, _P, _P]),
    "vs_conv_first_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vs_conv64_packed_floats": (c_size_t, [c_int, c_int]),
    "vs_conv64_pack": (c_int, [_P, _P, c_int, c_int, _P]),
    "vs_conv64_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vs_nhwc_conv_packed_bytes": (c_size_t, [c_int, c_int]),
    "vs_nhwc_conv_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "vs_nhwc_conv": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "vs_cvt_rows_bf16": (c_int, [_P, c_longlong, c_int, c_int, _P, c_int, _P]),
    "vs_gemm_bf16": (c_int, [c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P]),
    "vs_gemm_bf16_gated": (c_int, [c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "vs_gemm_bf16_split": (c_int, [c_int, c_int, _P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vs_nhwc_conv_f16x3_scratch_bytes": (c_size_t, [c_int, c_int]),
    "vs_nhwc_conv_f16x3_layer": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vs_f16x3_split": (c_int, [_P, _P, _P, _P, c_longlong, _P]),
    "vs_f16x3_merge": (c_int, [_P, _P, _P, _P, c_longlong, _P]),
    "vs_nhwc_conv_first": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "vs_nhwc_first_moments": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "vs_nhwc_first_stats": (c_int, [_P, _P, _P, ctypes.c_double, _P, _P]),
    "vs_nhwc_first_bwd_scratch_doubles": (c_int, []),
    "vs_nhwc_first_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vs_bn_finalize": (c_int, [_P, c_int, ctypes.c_double, c_int, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P]),
    "vs_nhwc_bn_apply": (c_int, [_P, _P, c_longlong, c_int, _P, _P, _P]),
    "vs_nhwc_conv_last": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vs_nhwc_conv_last_pre": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vs_nhwc_conv_wgrad_partial_floats": (c_size_t, [c_int, c_int]),
    "vs_nhwc_conv_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vs_nhwc_bn_act_bwd": (c_int, [_P, _P, _P, c_longlong, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vs_nhwc_bn_act_bwd_first": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vs_nhwc_conv_last_bwd_blocks": (c_int, []),
    "vs_nhwc_conv_last_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vs_nhwc_conv_dy": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vs_nhwc_conv_last_bwd_dy": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vs_nhwc_bn_bwd_from_dy": (c_int, [_P, _P, _P, c_longlong, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vs_nhwc_bn_bwd_first_from_dy": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vs_conv_last_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vs_conv64_packed_f16_floats": (c_size_t, [c_int, c_int]),
    "vs_pow2_scale": (c_int, [_P, c_longlong, _P, _P, _P]),
    "vs_conv64_pack_f16": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "vs_conv64_f16x3_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vs_gemm_nt": (c_int, [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, c_int,
                           c_int, c_int, _P]),
    "vs_lstm_packed_floats": (c_size_t, [c_int]),
    "vs_lstm_state_floats": (c_size_t, [c_int, c_int]),
    "vs_lstm_pack": (c_int, [_P, _P, _P, c_int, _P]),
    "vs_bilstm_recurrent": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vs_lstm_pack_math": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "vs_bilstm_recurrent_math": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    # training
    "vs_tape_layout_query": (c_int, [POINTER(VsDims), POINTER(VsTapeLayout)]),
    "vs_tape_bytes": (c_size_t, [POINTER(VsDims)]),
    "vs_forward_train": (c_int, [POINTER(VsDims), POINTER(VsParams), _P, _P, c_int, c_int, _P, c_size_t, _P, _P]),
    "vs_backward": (c_int, [POINTER(VsDims), POINTER(VsParams), _P, _P, c_int, c_int, _P, c_size_t, _P, _P,
                            POINTER(VsGrads), _P]),
    "vs_conv64_pack_dgrad": (c_int, [_P, _P, c_int, c_int, _P]),
    "vs_conv64_wgrad_partial_floats": (c_size_t, [c_int, c_int]),
    "vs_conv64_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vs_conv64_wgrad_f16x3": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "vs_set_wgrad_kernel": (c_int, [c_int]),
    "vs_set_conv_kernel": (c_int, [c_int]),
    "vs_set_lstm_kernel": (c_int, [c_int]),
    "vs_set_option": (c_int, [c_int, c_int]),
    "vs_get_option": (c_int, [c_int]),
    "vs_set_backward_overlap": (c_int, [c_int]),
    "vs_lstm_status": (c_int, [POINTER(VsDims), _P, c_size_t, _P, c_size_t, _P]),
    "vs_bn_act_bwd_first": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vs_bn_act_bwd": (c_int, [_P, _P, _P, c_int, c_longlong, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "vs_conv_last_dgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "vs_conv_last_wgrad_blocks": (c_int, []),
    "vs_conv_last_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vs_conv_first_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vs_gemm": (c_int, [c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_int,
                        c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "vs_gemm_f16x3": (c_int, [c_int, c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_int,
                              c_int, c_int, c_int, c_int, _P, _P]),
    "vs_bilstm_recurrent_train": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vs_lstm_packed_t_floats": (c_size_t, [c_int]),
    "vs_lstm_bwd_state_floats": (c_size_t, [c_int, c_int]),
    "vs_lstm_pack_t": (c_int, [_P, _P, _P, c_int, _P]),
    "vs_bilstm_recurrent_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "vs_lstm_pack_t_math": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "vs_bilstm_recurrent_bwd_math": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "vs_sisnr_workspace_bytes": (c_size_t, [POINTER(VsLossDims)]),
    "vs_sisnr_loss": (c_int, [POINTER(VsLossDims), _P, _P, _P, _P, _P, _P, c_size_t, _P, _P, _P, _P]),
    "vs_powerlaw_loss": (c_int, [_P, _P, _P, c_longlong, c_float, c_float, _P, _P, _P, _P]),
    "vs_audio_workspace_bytes": (c_size_t, [POINTER(VsLossDims)]),
    "vs_wav_to_spec": (c_int, [POINTER(VsLossDims), _P, _P, _P, _P, c_size_t, _P]),
    "vs_spec_to_wav": (c_int, [POINTER(VsLossDims), _P, _P, _P, _P, _P, c_size_t, _P]),
    "vs_sigmoid_bwd": (c_int, [_P, _P, _P, c_longlong, _P]),
    "vs_colsum": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
}

_lib = None


class VoiceSplitHipError(RuntimeError):
    pass


def load(path: str = None) -> ctypes.CDLL:
    """Load the HIP library; raises (never falls back) when it is missing or stale."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.isfile(path):
        raise VoiceSplitHipError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C voicesplit_amd/csrc`. There is no PyTorch/CPU fallback for this path.")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.vs_abi_version() != ABI_VERSION:
        raise VoiceSplitHipError(f"{path}: ABI version {lib.vs_abi_version()} != {ABI_VERSION}; rebuild")
    _apply_env_options(lib)
    _lib = lib
    return lib


# enum vs_option of include/voicesplit_hip.h.  The library reads no environment variable; for A/B timing from the shell this
# package maps the variables below onto vs_set_option ONCE, when it loads the library.
OPTIONS = {"FWD_PROLOGUE": 0, "HEAD_LEAF_SIDE": 1, "FEAT_ROWS": 2, "HEAD_BWD_GEMM": 3, "ABLATION": 4, "DETERMINISTIC": 5}
_ENV_OPTIONS = {
    "VOICESPLIT_FWD_PROLOGUE": ("FWD_PROLOGUE", int),
    "VOICESPLIT_HEAD_LEAF_SIDE": ("HEAD_LEAF_SIDE", int),
    "VOICESPLIT_FEAT_ROWS": ("FEAT_ROWS", int),
    "VOICESPLIT_HEAD_BWD_GEMM": ("HEAD_BWD_GEMM", int),
    "VOICESPLIT_ABLATION": ("ABLATION", int),
    "VOICESPLIT_DETERMINISTIC": ("DETERMINISTIC", int),
}


def _apply_env_options(lib):
    for var, (name, conv) in _ENV_OPTIONS.items():
        v = os.environ.get(var)
        if v is None or v == "":
            continue
        try:
            value = conv(v)
        except ValueError as err:
            raise VoiceSplitHipError(f"{var}={v}: not a value of this switch ({err})") from None
        if lib.vs_set_option(OPTIONS[name], value) != 0:
            msg = lib.vs_last_error()
            raise VoiceSplitHipError(f"{var}={v}: {msg.decode() if msg else 'rejected by vs_set_option'}")


def set_option(name: str, value: int):
    """vs_set_option by name (see OPTIONS); returns the previous value."""
    lib = load()
    prev = lib.vs_get_option(OPTIONS[name])
    check(lib.vs_set_option(OPTIONS[name], int(value)), f"vs_set_option({name})")
    return prev


def get_option(name: str) -> int:
    return load().vs_get_option(OPTIONS[name])


def check(rc: int, what: str):
    if rc != 0:
        msg = load().vs_last_error()
        raise VoiceSplitHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
