"""Thin torch-facing wrappers over the C ABI: torch supplies device buffers and the current HIP
stream, every FLOP happens inside libvoicesplit_hip.so.

Stage functions mirror the reference forward (models/voicesplit/model.py:66-89):
``conv_stack`` (:68-74), ``bilstm`` (:77-82), ``head`` (:83-87), ``forward`` (all of it).
The kernel-level functions (``conv64`` ...) exist for the unit tests.
"""
import ctypes
import os
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from ._lib import (ACT_MISH, ACT_NONE, ACT_RELU, ACT_SIGMOID, BN_EVAL, BN_TRAIN, VsDims, VsParams,
                   VsWsLayout, check)

ACT_CODES = {"relu": ACT_RELU, "mish": ACT_MISH, "none": ACT_NONE, "sigmoid": ACT_SIGMOID}

# conv.{idx} of the reference nn.Sequential: (Conv2d index, BatchNorm2d index) for cnn1..cnn8
CONV_INDEX = ((1, 2), (5, 6), (9, 10), (13, 14), (17, 18), (21, 22), (25, 26), (28, 29))


def _dev_check(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise _lib.VoiceSplitHipError(
            f"{name} is on {t.device}: this path only runs on an MI355X (HIP) device; there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


MATH_CODES = {"fp32": _lib.MATH_FP32, "f16x3": _lib.MATH_F16X3, "bf16": _lib.MATH_BF16}
_DEFAULT_MATH = os.environ.get("VOICESPLIT_CONV_MATH", "f16x3")


def set_conv_math(name: str):
    """Arithmetic of the dense contractions (64->64 convs forward / data / weight gradient, LSTM
    input GEMMs): "f16x3" (default) = fp32 operands split into two f16 halves, three products on
    the f16 matrix cores, fp32 accumulate -- fp32-class accuracy (same parity tolerances) at 3/16
    of the matrix-pipe time (csrc/conv_f16x3.hip); "fp32" = the f32 matrix cores, bitwise an fmaf
    chain; "bf16" (BASELINE configs[2], opt-in only) = operands rounded to bf16, ONE product on the
    bf16 matrix cores, fp32 accumulate and fp32 everywhere else -- a third of the matrix work of f16x3 at
    bf16 accuracy (not the 1e-4 contract: tests/test_gpu_bf16.py states what it holds).  Also
    selectable with VOICESPLIT_CONV_MATH."""
    global _DEFAULT_MATH
    if name not in MATH_CODES:
        raise ValueError(f"conv math must be one of {sorted(MATH_CODES)}")
    _DEFAULT_MATH = name


def get_conv_math() -> str:
    return _DEFAULT_MATH


def make_dims(B, T, F, E, H, FC1, FC2, math: Optional[str] = None) -> VsDims:
    return VsDims(int(B), int(T), int(F), int(E), int(H), int(FC1), int(FC2), MATH_CODES[math or _DEFAULT_MATH])


def workspace_layout(dims: VsDims) -> VsWsLayout:
    lay = VsWsLayout()
    check(_lib.load().vs_workspace_layout(ctypes.byref(dims), ctypes.byref(lay)), "vs_workspace_layout")
    return lay


_WS_CACHE: Dict[tuple, torch.Tensor] = {}


def get_workspace(dims: VsDims, device) -> torch.Tensor:
    """Caller-owned scratch (the library never allocates); cached per device and size."""
    nbytes = _lib.load().vs_workspace_bytes(ctypes.byref(dims))
    if nbytes == 0:
        check(-1, "vs_workspace_bytes")
    key = (torch.device(device).index, )
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        _WS_CACHE.pop(key, None)   # drop the old buffer before allocating the larger one
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _WS_CACHE[key] = ws
    return ws


def release_workspaces():
    _WS_CACHE.clear()
    _TAPE_POOL.clear()


def ws_view(ws: torch.Tensor, offset: int, shape: Sequence[int], dtype=torch.float32) -> torch.Tensor:
    n = 1
    for s in shape:
        n *= int(s)
    esz = torch.empty((), dtype=dtype).element_size()
    return ws[offset:offset + n * esz].view(dtype).view(*shape)


def pack_params(sd: Dict[str, torch.Tensor]) -> VsParams:
    """state_dict (reference key names) -> vs_params of raw device pointers."""
    p = VsParams()
    for l, (ci, bi) in enumerate(CONV_INDEX):
        names = {"weight": f"conv.{ci}.weight", "bias": f"conv.{ci}.bias",
                 "bn_weight": f"conv.{bi}.weight", "bn_bias": f"conv.{bi}.bias",
                 "bn_running_mean": f"conv.{bi}.running_mean", "bn_running_var": f"conv.{bi}.running_var"}
        for field, key in names.items():
            t = sd[key]
            _dev_check(t, key)
            setattr(p.conv[l], field, t.data_ptr())
    for d, suffix in enumerate(("", "_reverse")):
        for field, key in (("w_ih", "weight_ih_l0"), ("w_hh", "weight_hh_l0"),
                           ("b_ih", "bias_ih_l0"), ("b_hh", "bias_hh_l0")):
            t = sd[f"lstm.{key}{suffix}"]
            _dev_check(t, f"lstm.{key}{suffix}")
            getattr(p, field)[d] = t.data_ptr()
    for field, key in (("fc1_w", "fc1.weight"), ("fc1_b", "fc1.bias"), ("fc2_w", "fc2.weight"), ("fc2_b", "fc2.bias")):
        _dev_check(sd[key], key)
        setattr(p, field, sd[key].data_ptr())
    return p


# ---------------------------------------------------------------------------------------------
# whole path and stages
# ---------------------------------------------------------------------------------------------

def _check_inputs(x, dvec, dims: VsDims):
    _dev_check(x, "x")
    _dev_check(dvec, "speaker_embedding")
    if x.dim() != 3 or x.shape[2] != dims.F:
        raise ValueError(f"x must be [B, T, num_freq={dims.F}] (F contiguous), got {tuple(x.shape)}")
    if dvec.dim() != 2 or dvec.shape[0] != x.shape[0] or dvec.shape[1] != dims.E:
        raise ValueError(f"speaker_embedding must be [B={x.shape[0]}, emb_dim={dims.E}], got {tuple(dvec.shape)}")


def forward(sd, x, dvec, dims: VsDims, conv_act: str, training: bool = False,
            workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mask = model(x, dvec): the whole reference forward in one C-ABI call."""
    lib = _lib.load()
    _check_inputs(x, dvec, dims)
    params = pack_params(sd)
    ws = workspace if workspace is not None else get_workspace(dims, x.device)
    mask = torch.empty(dims.B, dims.T, dims.FC2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.vs_forward(ctypes.byref(dims), ctypes.byref(params), _p(x), _p(dvec), ACT_CODES[conv_act],
                            BN_TRAIN if training else BN_EVAL, _p(ws), ws.numel(), _p(mask), _stream())
    check(rc, "vs_forward")
    return mask


class PreparedWeights:
    """What an eval-mode forward derives from the parameters alone (vs_prepare_weights), kept on the device
    together with the identity of the tensors it was derived from: ``matches(sd, dims)`` is False as soon as
    any parameter / BatchNorm buffer was replaced or modified in place (optimizer step, load_state_dict,
    ``.to()``), or the arithmetic changed."""

    def __init__(self, sd, dims: VsDims):
        lib = _lib.load()
        some = next(iter(sd.values()))
        nbytes = lib.vs_prepared_bytes(ctypes.byref(dims))
        if nbytes == 0:
            check(-1, "vs_prepared_bytes")
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=some.device)
        self.key = self._key(sd, dims)
        params = pack_params(sd)
        with torch.cuda.device(some.device):
            rc = lib.vs_prepare_weights(ctypes.byref(dims), ctypes.byref(params), _p(self.buf), self.buf.numel(), _stream())
        check(rc, "vs_prepare_weights")

    @staticmethod
    def _key(sd, dims: VsDims):
        return (dims.F, dims.E, dims.H, dims.FC1, dims.FC2, dims.math,
                tuple((k, v.data_ptr(), v._version) for k, v in sorted(sd.items()) if not k.endswith("num_batches_tracked")))

    def matches(self, sd, dims: VsDims) -> bool:
        return self.key == self._key(sd, dims)


def forward_prepared(sd, prepared: PreparedWeights, x, dvec, dims: VsDims, conv_act: str,
                     workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Eval-mode mask = model(x, dvec) with the weight-only work read from ``prepared``; bit-identical to
    ``forward(..., training=False)``."""
    lib = _lib.load()
    _check_inputs(x, dvec, dims)
    params = pack_params(sd)
    ws = workspace if workspace is not None else get_workspace(dims, x.device)
    mask = torch.empty(dims.B, dims.T, dims.FC2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.vs_forward_prepared(ctypes.byref(dims), ctypes.byref(params), _p(prepared.buf), prepared.buf.numel(),
                                     _p(x), _p(dvec), ACT_CODES[conv_act], _p(ws), ws.numel(), _p(mask), _stream())
    check(rc, "vs_forward_prepared")
    return mask


def conv_stack(sd, x, dims: VsDims, conv_act: str, training: bool = False, workspace=None) -> torch.Tensor:
    lib = _lib.load()
    _dev_check(x, "x")
    params = pack_params(sd)
    ws = workspace if workspace is not None else get_workspace(dims, x.device)
    feat = torch.empty(dims.B, dims.T, 8 * dims.F, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.vs_conv_stack_fwd(ctypes.byref(dims), ctypes.byref(params), _p(x), ACT_CODES[conv_act],
                                   BN_TRAIN if training else BN_EVAL, _p(ws), ws.numel(), _p(feat), _stream())
    check(rc, "vs_conv_stack_fwd")
    return feat


def bilstm(sd, feat, dvec, dims: VsDims, workspace=None) -> torch.Tensor:
    lib = _lib.load()
    _dev_check(feat, "feat")
    _dev_check(dvec, "speaker_embedding")
    params = pack_params(sd)
    ws = workspace if workspace is not None else get_workspace(dims, feat.device)
    out = torch.empty(dims.B, dims.T, 2 * dims.H, dtype=torch.float32, device=feat.device)
    with torch.cuda.device(feat.device):
        rc = lib.vs_bilstm_fwd(ctypes.byref(dims), ctypes.byref(params), _p(feat), _p(dvec), _p(ws), ws.numel(),
                               _p(out), _stream())
    check(rc, "vs_bilstm_fwd")
    return out


def head(sd, lstm_out, dims: VsDims, want_logits: bool = False, workspace=None):
    lib = _lib.load()
    _dev_check(lstm_out, "lstm_out")
    params = pack_params(sd)
    ws = workspace if workspace is not None else get_workspace(dims, lstm_out.device)
    mask = torch.empty(dims.B, dims.T, dims.FC2, dtype=torch.float32, device=lstm_out.device)
    logits = torch.empty_like(mask) if want_logits else None
    with torch.cuda.device(lstm_out.device):
        rc = lib.vs_head_fwd(ctypes.byref(dims), ctypes.byref(params), _p(lstm_out), _p(ws), ws.numel(),
                             _p(logits), _p(mask), _stream())
    check(rc, "vs_head_fwd")
    return (mask, logits) if want_logits else mask


# ---------------------------------------------------------------------------------------------
# kernel-level wrappers (unit tests)
# ---------------------------------------------------------------------------------------------

def bn_fold(gamma, beta, mean, var, conv_bias, eps=1e-5):
    lib = _lib.load()
    C = gamma.numel()
    scale, shift = torch.empty_like(gamma), torch.empty_like(gamma)
    check(lib.vs_bn_fold(_p(gamma), _p(beta), _p(mean), _p(var), _p(conv_bias), eps, C, _p(scale), _p(shift), _stream()),
          "vs_bn_fold")
    return scale, shift


def conv_first(x, w, scale, shift, act: str):
    lib = _lib.load()
    for n, t in (("x", x), ("w", w), ("scale", scale), ("shift", shift)):
        _dev_check(t, n)
    B, T, F = x.shape
    out = torch.empty(B, 64, T, F, dtype=torch.float32, device=x.device)
    check(lib.vs_conv_first_fwd(_p(x), _p(w), _p(scale), _p(shift), _p(out), B, T, F, ACT_CODES[act], _stream()),
          "vs_conv_first_fwd")
    return out


def _conv64_f16x3(x, w, scale, shift, dil: int, act_code: int, transpose_flip: int):
    lib = _lib.load()
    B, C, T, F = x.shape
    KT, KF = w.shape[2], w.shape[3]
    packed = torch.empty(lib.vs_conv64_packed_f16_floats(KT, KF), dtype=torch.float32, device=x.device)
    scales = torch.zeros(8, dtype=torch.float32, device=x.device)
    amax = scales[4:].view(torch.int32)
    check(lib.vs_pow2_scale(_p(x), x.numel(), _p(amax), _p(scales), _stream()), "vs_pow2_scale")
    check(lib.vs_conv64_pack_f16(_p(w), _p(packed), KT, KF, transpose_flip, _p(amax[1:]), _p(scales[2:]), _stream()),
          "vs_conv64_pack_f16")
    out = torch.empty_like(x)
    check(lib.vs_conv64_f16x3_fwd(_p(x), _p(packed), _p(scale), _p(shift), _p(scales), _p(scales[2:]), _p(out),
                                  B, T, F, KT, KF, dil, act_code, _stream()), "vs_conv64_f16x3_fwd")
    return out


def conv64(x, w, scale, shift, dil: int, act: str, math: str = "fp32"):
    """x [B,64,T,F], w [64,64,KT,KF] -> [B,64,T,F] with 'same' zero padding and time dilation."""
    lib = _lib.load()
    for n, t in (("x", x), ("w", w), ("scale", scale), ("shift", shift)):
        _dev_check(t, n)
    if math == "f16x3":
        return _conv64_f16x3(x, w, scale, shift, dil, ACT_CODES[act], 0)
    B, C, T, F = x.shape
    KT, KF = w.shape[2], w.shape[3]
    packed = torch.empty(lib.vs_conv64_packed_floats(KT, KF), dtype=torch.float32, device=x.device)
    check(lib.vs_conv64_pack(_p(w), _p(packed), KT, KF, _stream()), "vs_conv64_pack")
    out = torch.empty_like(x)
    check(lib.vs_conv64_fwd(_p(x), _p(packed), _p(scale), _p(shift), _p(out), B, T, F, KT, KF, dil,
                            ACT_CODES[act], _stream()), "vs_conv64_fwd")
    return out


def conv_last(x, w, scale, shift, act: str):
    lib = _lib.load()
    for n, t in (("x", x), ("w", w), ("scale", scale), ("shift", shift)):
        _dev_check(t, n)
    B, C, T, F = x.shape
    out = torch.empty(B, T, 8 * F, dtype=torch.float32, device=x.device)
    check(lib.vs_conv_last_fwd(_p(x), _p(w), _p(scale), _p(shift), _p(out), B, T, F, ACT_CODES[act], _stream()),
          "vs_conv_last_fwd")
    return out


def gemm_nt(A, W, bias1=None, bias2=None, rowbias=None, group: int = 1, a_relu: bool = False, act: str = "none",
            K: Optional[int] = None):
    """act(opA(A)[:, :K] @ W[:, :K]^T + biases); A [M,lda], W [N,ldw] row-major."""
    lib = _lib.load()
    _dev_check(A, "A")
    _dev_check(W, "W")
    M, lda = A.shape
    N, ldw = W.shape
    K = K if K is not None else min(lda, ldw)
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    ldrb = rowbias.shape[1] if rowbias is not None else 0
    check(lib.vs_gemm_nt(_p(A), lda, _p(W), ldw, _p(C), N, M, N, K, _p(bias1), _p(bias2), _p(rowbias), ldrb, group,
                         int(a_relu), ACT_CODES[act], _stream()), "vs_gemm_nt")
    return C


def bilstm_recurrent(xg, w_hh_f, w_hh_b, math=None):
    """xg [B,T,8H] (bias already added) -> [B,T,2H].  math: None = the fp32-MFMA entry points, else MATH_* through the
    *_math entry points (the recurrent products on the f16 matrix instructions)."""
    lib = _lib.load()
    for n, t in (("xg", xg), ("w_hh_f", w_hh_f), ("w_hh_b", w_hh_b)):
        _dev_check(t, n)
    B, T, H8 = xg.shape
    H = H8 // 8
    packed = torch.empty(lib.vs_lstm_packed_floats(H), dtype=torch.float32, device=xg.device)
    state = torch.empty(lib.vs_lstm_state_floats(B, H), dtype=torch.float32, device=xg.device)
    out = torch.empty(B, T, 2 * H, dtype=torch.float32, device=xg.device)
    if math is None:
        check(lib.vs_lstm_pack(_p(w_hh_f), _p(w_hh_b), _p(packed), H, _stream()), "vs_lstm_pack")
        check(lib.vs_bilstm_recurrent(_p(xg), _p(packed), _p(state), _p(out), B, T, H, _stream()), "vs_bilstm_recurrent")
    else:
        check(lib.vs_lstm_pack_math(_p(w_hh_f), _p(w_hh_b), _p(packed), H, int(math), _stream()), "vs_lstm_pack_math")
        check(lib.vs_bilstm_recurrent_math(_p(xg), _p(packed), _p(state), _p(out), None, None, B, T, H, int(math), _stream()),
              "vs_bilstm_recurrent_math")
    _lstm_check_err(state, state.numel() - 64, "vs_bilstm_recurrent")
    return out


def _lstm_check_err(state: torch.Tensor, word: int, what: str):
    """The persistent recurrences report a spin that gave up (a workgroup of the launch was not
    resident) through a word in the state buffer; the step kernels leave it 0.  Unit-test helper:
    synchronises."""
    if int(state.view(torch.int32)[word].item()) != 0:
        raise _lib.VoiceSplitHipError(f"{what}: the persistent LSTM kernel gave up waiting for a peer workgroup")


# ---------------------------------------------------------------------------------------------
# training: forward with tape, backward (train.py:94-110 through models/voicesplit/model.py:66-89)
# ---------------------------------------------------------------------------------------------

# vs_grads field <- state_dict key, in the order model.parameters() yields them is NOT assumed:
# gradients are returned as a dict keyed like the state_dict.
def tape_layout(dims: VsDims) -> "_lib.VsTapeLayout":
    lay = _lib.VsTapeLayout()
    check(_lib.load().vs_tape_layout_query(ctypes.byref(dims), ctypes.byref(lay)), "vs_tape_layout_query")
    return lay


_TAPE_POOL: Dict[int, list] = {}       # device index -> free tapes
_TAPE_POOL_MAX = int(os.environ.get("VOICESPLIT_TAPE_POOL", "1"))


def new_tape(dims: VsDims, device) -> torch.Tensor:
    """Caller-owned training tape (saved activations + backward scratch, 49 GB at B=64).  Tapes
    are recycled through a small per-device pool: a forward takes one, its backward hands it back
    (``recycle_tape``), so steady-state training touches no allocator at all.  Any free tape with
    enough capacity is reused (variable B / T: the largest shape seen so far serves the smaller
    ones); at most ``VOICESPLIT_TAPE_POOL`` (default 1) free tapes are retained per device, and
    free tapes that are too small are released before a larger one is allocated."""
    nbytes = _lib.load().vs_tape_bytes(ctypes.byref(dims))
    if nbytes == 0:
        check(-1, "vs_tape_bytes")
    free = _TAPE_POOL.setdefault(torch.device(device).index, [])
    fit = [t for t in free if t.numel() >= nbytes]
    if fit:
        t = min(fit, key=lambda u: u.numel())
        free.remove(t)
        return t
    free.clear()                                              # too small: let go before the big allocation
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def recycle_tape(tape: torch.Tensor):
    pool = _TAPE_POOL.setdefault(tape.device.index, [])
    if len(pool) < _TAPE_POOL_MAX:
        pool.append(tape)


def tape_pool_bytes(device=None) -> int:
    """Bytes held by free tapes (outside torch's caching allocator statistics of live tensors)."""
    idx = None if device is None else torch.device(device).index
    return sum(t.numel() for k, v in _TAPE_POOL.items() if idx is None or k == idx for t in v)


def forward_train(sd, x, dvec, dims: VsDims, conv_act: str, training: bool, tape: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _check_inputs(x, dvec, dims)
    params = pack_params(sd)
    mask = torch.empty(dims.B, dims.T, dims.FC2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.vs_forward_train(ctypes.byref(dims), ctypes.byref(params), _p(x), _p(dvec), ACT_CODES[conv_act],
                                  BN_TRAIN if training else BN_EVAL, _p(tape), tape.numel(), _p(mask), _stream())
    check(rc, "vs_forward_train")
    return mask


GRAD_KEYS_CONV = (("weight", "weight"), ("bias", "bias"))
GRAD_KEYS_BN = (("bn_weight", "weight"), ("bn_bias", "bias"))


def backward(sd, x, dvec, dims: VsDims, conv_act: str, training: bool, tape: torch.Tensor, mask, dmask,
             want_dvec: bool = False, sink: Optional[Dict[str, torch.Tensor]] = None,
             leaves_event: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """d(loss)/d(parameters) for dmask = d(loss)/d(mask); returns {state_dict key: gradient}
    (+ 'speaker_embedding' when want_dvec).  sink: {key: preallocated tensor} the library writes those gradients into
    (overwriting, vs_backward never accumulates) instead of fresh tensors -- the trainer's flat all-reduce bucket.
    leaves_event: raw hipEvent_t handle (``torch.cuda.Event.cuda_event``) the library records when the head's and the BiLSTM's
    gradients are final (vs_grads.leaves_event, ABI 9)."""
    lib = _lib.load()
    _dev_check(dmask, "grad_mask")
    _dev_check(mask, "mask")
    params = pack_params(sd)
    grads = _lib.VsGrads()
    out: Dict[str, torch.Tensor] = {}

    def alloc(key):
        t = sink.get(key) if sink else None
        if t is None:
            t = torch.empty_like(sd[key])
        elif (t.shape != sd[key].shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != sd[key].device):
            raise ValueError(f"gradient sink for {key}: expected a contiguous float32 {tuple(sd[key].shape)} tensor on {sd[key].device}")
        out[key] = t
        return t.data_ptr()

    for l, (ci, bi) in enumerate(CONV_INDEX):
        for field, name in GRAD_KEYS_CONV:
            setattr(grads.conv[l], field, alloc(f"conv.{ci}.{name}"))
        for field, name in GRAD_KEYS_BN:
            setattr(grads.conv[l], field, alloc(f"conv.{bi}.{name}"))
    for d, suffix in enumerate(("", "_reverse")):
        for field, key in (("w_ih", "weight_ih_l0"), ("w_hh", "weight_hh_l0"),
                           ("b_ih", "bias_ih_l0"), ("b_hh", "bias_hh_l0")):
            getattr(grads, field)[d] = alloc(f"lstm.{key}{suffix}")
    for field, key in (("fc1_w", "fc1.weight"), ("fc1_b", "fc1.bias"), ("fc2_w", "fc2.weight"), ("fc2_b", "fc2.bias")):
        setattr(grads, field, alloc(key))
    if want_dvec:
        out["speaker_embedding"] = torch.empty_like(dvec)
        grads.dvec = out["speaker_embedding"].data_ptr()
    if leaves_event:
        grads.leaves_event = int(leaves_event)
    with torch.cuda.device(x.device):
        rc = lib.vs_backward(ctypes.byref(dims), ctypes.byref(params), _p(x), _p(dvec), ACT_CODES[conv_act],
                             BN_TRAIN if training else BN_EVAL, _p(tape), tape.numel(), _p(mask), _p(dmask),
                             ctypes.byref(grads), _stream())
    check(rc, "vs_backward")
    return out


# ---- backward kernels (unit tests) -------------------------------------------------------------

def conv64_dgrad(dz, w, dil: int, math: str = "fp32"):
    """dIn [B,64,T,F] = conv^T(dz, w): the forward kernel with transposed + tap-flipped weights."""
    lib = _lib.load()
    _dev_check(dz, "dz")
    _dev_check(w, "w")
    if math == "f16x3":
        ones, zeros = torch.ones(64, device=dz.device), torch.zeros(64, device=dz.device)
        return _conv64_f16x3(dz, w, ones, zeros, dil, ACT_NONE, 1)
    B, C, T, F = dz.shape
    KT, KF = w.shape[2], w.shape[3]
    packed = torch.empty(lib.vs_conv64_packed_floats(KT, KF), dtype=torch.float32, device=dz.device)
    check(lib.vs_conv64_pack_dgrad(_p(w), _p(packed), KT, KF, _stream()), "vs_conv64_pack_dgrad")
    ones, zeros = torch.ones(64, device=dz.device), torch.zeros(64, device=dz.device)
    out = torch.empty_like(dz)
    check(lib.vs_conv64_fwd(_p(dz), _p(packed), _p(ones), _p(zeros), _p(out), B, T, F, KT, KF, dil,
                            ACT_NONE, _stream()), "vs_conv64_fwd(dgrad)")
    return out


def conv64_wgrad(dz, x, KT: int, KF: int, dil: int, math: str = "fp32"):
    lib = _lib.load()
    _dev_check(dz, "dz")
    _dev_check(x, "x")
    B, C, T, F = dz.shape
    part = torch.empty(lib.vs_conv64_wgrad_partial_floats(KT, KF), dtype=torch.float32, device=dz.device)
    dw = torch.empty(64, 64, KT, KF, dtype=torch.float32, device=dz.device)
    if math == "f16x3":
        scratch = torch.zeros(8, dtype=torch.float32, device=dz.device)
        check(lib.vs_conv64_wgrad_f16x3(_p(dz), _p(x), _p(part), _p(dw), _p(scratch), B, T, F, KT, KF, dil, _stream()),
              "vs_conv64_wgrad_f16x3")
        return dw
    check(lib.vs_conv64_wgrad(_p(dz), _p(x), _p(part), _p(dw), B, T, F, KT, KF, dil, _stream()), "vs_conv64_wgrad")
    return dw


def bn_act_bwd(da, z, C: int, act: str, training: bool, scale, shift, mean, invstd):
    """rows [R][L] view of da/z (channel = r % C) -> dz, dgamma, dbeta, dbias."""
    lib = _lib.load()
    for n, t in (("da", da), ("z", z), ("scale", scale), ("shift", shift), ("mean", mean), ("invstd", invstd)):
        _dev_check(t, n)
    L = da.shape[-1]
    R = da.numel() // L
    dev = da.device
    dz = torch.empty_like(da)
    dgamma, dbeta, dbias = (torch.empty(C, device=dev) for _ in range(3))
    stats = torch.empty(2 * C, dtype=torch.float64, device=dev)
    coef = torch.empty(3 * C, device=dev)
    check(lib.vs_bn_act_bwd(_p(da), _p(z), _p(dz), C, R, L, ACT_CODES[act], BN_TRAIN if training else BN_EVAL,
                            _p(scale), _p(shift), _p(mean), _p(invstd), _p(dgamma), _p(dbeta), _p(dbias),
                            _p(stats), _p(coef), _stream()), "vs_bn_act_bwd")
    return dz, dgamma, dbeta, dbias


def bn_act_bwd_first(da, z, x, act: str, training: bool, scale, shift, mean, invstd):
    """cnn1: da, z [B,64,T,F] and the input x [B,T,F] -> dgamma, dbeta, dbias, dw [64,1,1,7]
    (BatchNorm+activation backward fused with the 1x7 weight gradient; dZ1 is not materialised)."""
    lib = _lib.load()
    for n, t in (("da", da), ("z", z), ("x", x), ("scale", scale), ("shift", shift), ("mean", mean), ("invstd", invstd)):
        _dev_check(t, n)
    B, C, T, F = da.shape
    if C != 64 or tuple(x.shape) != (B, T, F) or z.shape != da.shape:
        raise ValueError(f"bn_act_bwd_first: da {tuple(da.shape)}, z {tuple(z.shape)}, x {tuple(x.shape)}")
    dev = da.device
    dgamma, dbeta, dbias = (torch.empty(64, device=dev) for _ in range(3))
    dw = torch.empty(64, 1, 1, 7, device=dev)
    xpad = torch.empty(B * T * (F + 6), device=dev)
    stats = torch.empty(128, dtype=torch.float64, device=dev)
    acc = torch.empty(448, dtype=torch.float64, device=dev)
    coef = torch.empty(192, device=dev)
    check(lib.vs_bn_act_bwd_first(_p(da), _p(z), _p(x), _p(xpad), B, T, F, ACT_CODES[act], BN_TRAIN if training else BN_EVAL,
                                  _p(scale), _p(shift), _p(mean), _p(invstd), _p(dgamma), _p(dbeta), _p(dbias), _p(dw),
                                  _p(stats), _p(coef), _p(acc), _stream()), "vs_bn_act_bwd_first")
    return dgamma, dbeta, dbias, dw


def conv_last_dgrad(dz8, w, B, T, F):
    lib = _lib.load()
    _dev_check(dz8, "dz8")
    _dev_check(w, "w")
    out = torch.empty(B, 64, T, F, dtype=torch.float32, device=dz8.device)
    check(lib.vs_conv_last_dgrad(_p(dz8), _p(w), _p(out), B, T, F, _stream()), "vs_conv_last_dgrad")
    return out


def conv_last_wgrad(dz8, a7):
    lib = _lib.load()
    _dev_check(dz8, "dz8")
    _dev_check(a7, "a7")
    B, C, T, F = a7.shape
    part = torch.empty(lib.vs_conv_last_wgrad_blocks() * 512, dtype=torch.float32, device=a7.device)
    dw = torch.empty(8, 64, 1, 1, dtype=torch.float32, device=a7.device)
    check(lib.vs_conv_last_wgrad(_p(dz8), _p(a7), _p(part), _p(dw), B, T, F, _stream()), "vs_conv_last_wgrad")
    return dw


def conv_first_wgrad(dz1, x):
    lib = _lib.load()
    _dev_check(dz1, "dz1")
    _dev_check(x, "x")
    B, T, F = x.shape
    acc = torch.empty(448, dtype=torch.float64, device=x.device)
    dw = torch.empty(64, 1, 1, 7, dtype=torch.float32, device=x.device)
    check(lib.vs_conv_first_wgrad(_p(dz1), _p(x), _p(acc), _p(dw), B, T, F, _stream()), "vs_conv_first_wgrad")
    return dw


def gemm(A, W, M: int, N: int, K: int, layout_a: int = 0, layout_w: int = 0, bias=None, gate=None,
         a_relu=False, w_relu=False, act="none", out=None, accumulate=False, w_shift=0, w_group=0, splits=1,
         math: str = "fp32"):
    """General GEMM (see vs_gemm in the header); A/W are 2-D row-major views with their own ld."""
    lib = _lib.load()
    _dev_check(A, "A")
    _dev_check(W, "W")
    C = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=A.device)
    if math == "f16x3":
        scratch = torch.zeros(8, dtype=torch.float32, device=A.device)
        check(lib.vs_gemm_f16x3(layout_a, layout_w, _p(A), A.shape[1], _p(W), W.shape[1], _p(C), C.shape[1], M, N, K,
                                _p(bias), _p(gate), gate.shape[1] if gate is not None else 0, int(a_relu), int(w_relu),
                                ACT_CODES[act], int(accumulate), _p(scratch), _stream()), "vs_gemm_f16x3")
        return C
    part = torch.empty(splits * M * N, dtype=torch.float32, device=A.device) if splits > 1 else None
    check(lib.vs_gemm(layout_a, layout_w, _p(A), A.shape[1], _p(W), W.shape[1], _p(C), C.shape[1], M, N, K,
                      _p(bias), _p(gate), gate.shape[1] if gate is not None else 0, int(a_relu), int(w_relu),
                      ACT_CODES[act], int(accumulate), w_shift, w_group, splits, _p(part), _stream()), "vs_gemm")
    return C


def bilstm_recurrent_train(xg, w_hh_f, w_hh_b, math=None):
    """xg [B,T,8H] -> (out [B,T,2H], gates [B,T,8H] activated, c [B,T,2H]).  math: see bilstm_recurrent."""
    lib = _lib.load()
    for n, t in (("xg", xg), ("w_hh_f", w_hh_f), ("w_hh_b", w_hh_b)):
        _dev_check(t, n)
    B, T, H8 = xg.shape
    H = H8 // 8
    packed = torch.empty(lib.vs_lstm_packed_floats(H), dtype=torch.float32, device=xg.device)
    state = torch.empty(lib.vs_lstm_state_floats(B, H), dtype=torch.float32, device=xg.device)
    out = torch.empty(B, T, 2 * H, dtype=torch.float32, device=xg.device)
    gates = xg.clone()
    c = torch.empty(B, T, 2 * H, dtype=torch.float32, device=xg.device)
    if math is None:
        check(lib.vs_lstm_pack(_p(w_hh_f), _p(w_hh_b), _p(packed), H, _stream()), "vs_lstm_pack")
        check(lib.vs_bilstm_recurrent_train(_p(gates), _p(packed), _p(state), _p(out), _p(gates), _p(c), B, T, H, _stream()),
              "vs_bilstm_recurrent_train")
    else:
        check(lib.vs_lstm_pack_math(_p(w_hh_f), _p(w_hh_b), _p(packed), H, int(math), _stream()), "vs_lstm_pack_math")
        check(lib.vs_bilstm_recurrent_math(_p(gates), _p(packed), _p(state), _p(out), _p(gates), _p(c), B, T, H, int(math), _stream()),
              "vs_bilstm_recurrent_math")
    _lstm_check_err(state, state.numel() - 64, "vs_bilstm_recurrent_train")
    return out, gates, c


def bilstm_recurrent_bwd(gates, c, dout, w_hh_f, w_hh_b, math=None):
    """BPTT: returns d(loss)/d(xg) [B,T,8H] (gates is not modified: works on a copy).  math: see bilstm_recurrent."""
    lib = _lib.load()
    for n, t in (("gates", gates), ("c", c), ("dout", dout), ("w_hh_f", w_hh_f), ("w_hh_b", w_hh_b)):
        _dev_check(t, n)
    B, T, H8 = gates.shape
    H = H8 // 8
    packed_t = torch.empty(lib.vs_lstm_packed_t_floats(H), dtype=torch.float32, device=gates.device)
    state = torch.empty(lib.vs_lstm_bwd_state_floats(B, H), dtype=torch.float32, device=gates.device)
    dxg = gates.clone()
    if math is None:
        check(lib.vs_lstm_pack_t(_p(w_hh_f), _p(w_hh_b), _p(packed_t), H, _stream()), "vs_lstm_pack_t")
        check(lib.vs_bilstm_recurrent_bwd(_p(packed_t), _p(state), _p(dxg), _p(c), _p(dout), B, T, H, _stream()),
              "vs_bilstm_recurrent_bwd")
    else:
        check(lib.vs_lstm_pack_t_math(_p(w_hh_f), _p(w_hh_b), _p(packed_t), H, int(math), _stream()), "vs_lstm_pack_t_math")
        check(lib.vs_bilstm_recurrent_bwd_math(_p(packed_t), _p(state), _p(dxg), _p(c), _p(dout), B, T, H, int(math), _stream()),
              "vs_bilstm_recurrent_bwd_math")
    _lstm_check_err(state, state.numel() - 64, "vs_bilstm_recurrent_bwd")
    return dxg


def sigmoid_bwd(dmask, mask):
    lib = _lib.load()
    _dev_check(dmask, "dmask")
    _dev_check(mask, "mask")
    out = torch.empty_like(mask)
    check(lib.vs_sigmoid_bwd(_p(dmask), _p(mask), _p(out), mask.numel(), _stream()), "vs_sigmoid_bwd")
    return out


def colsum(x, groups: int, rows: int):
    lib = _lib.load()
    _dev_check(x, "x")
    N = x.shape[-1]
    out = torch.empty(groups, N, dtype=torch.float32, device=x.device)
    check(lib.vs_colsum(_p(x), N, groups, rows, N, _p(out), N, _stream()), "vs_colsum")
    return out


# ---------------------------------------------------------------------------------------------
# channels-last bf16 kernels of the VS_MATH_BF16 path (csrc/conv_nhwc.hip); unit-test surface
# ---------------------------------------------------------------------------------------------
def nhwc_conv(x, w, scale, shift, dil: int, act: str, transpose_flip: bool = False, stats=False):
    """x [B,T,F,64] bf16 (channels last), w [64,64,KT,KF] fp32 -> act(conv(x) * scale + shift) as [B,T,F,64] bf16,
    'same' zero padding, time dilation `dil`.  stats: also return the per-channel {sum, sum of squares} of the
    outputs, [64, 2] float64 (train-mode BatchNorm statistics; act must be "none"); stats="raw": the 64 partial slots
    [64 slots, 64, 2] as the kernel left them (what vs_bn_finalize takes)."""
    lib = _lib.load()
    _dev_check(x, "x", torch.bfloat16)
    for n, t in (("w", w), ("scale", scale), ("shift", shift)):
        _dev_check(t, n)
    B, T, F, C = x.shape
    if C != 64:
        raise ValueError("nhwc_conv: 64 channels expected")
    KT, KF = w.shape[2], w.shape[3]
    packed = torch.empty(lib.vs_nhwc_conv_packed_bytes(KT, KF), dtype=torch.uint8, device=x.device)
    check(lib.vs_nhwc_conv_pack(_p(w), _p(packed), KT, KF, int(transpose_flip), _stream()), "vs_nhwc_conv_pack")
    out = torch.empty_like(x)
    st = torch.zeros(64, 64, 2, dtype=torch.float64, device=x.device) if stats else None
    check(lib.vs_nhwc_conv(_p(x), _p(packed), _p(scale), _p(shift), _p(out), B, T, F, KT, KF, dil, ACT_CODES[act],
                           _p(st), _stream()), "vs_nhwc_conv")
    if stats == "raw":
        return out, st
    return (out, st.sum(0)) if stats else out


def f16x3_split(x, scale: float):
    """fp32 tensor -> (hi, lo) f16 planes of x * scale (scale a power of two) and the device scale pair {s, 1/s}."""
    lib = _lib.load()
    _dev_check(x, "x")
    s2 = torch.tensor([scale, 1.0 / scale], dtype=torch.float32, device=x.device)
    hi = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    lo = torch.empty_like(hi)
    check(lib.vs_f16x3_split(_p(x), _p(s2), _p(hi), _p(lo), x.numel(), _stream()), "vs_f16x3_split")
    return hi, lo, s2


def f16x3_merge(hi, lo, scale2):
    lib = _lib.load()
    x = torch.empty(hi.shape, dtype=torch.float32, device=hi.device)
    check(lib.vs_f16x3_merge(_p(hi), _p(lo), _p(scale2), _p(x), x.numel(), _stream()), "vs_f16x3_merge")
    return x


def nhwc_conv_f16x3(hi, lo, scale2, w, bn_scale, bn_shift, dil: int, act: str, amax_in=None, scratch=None):
    """One 64 -> 64 layer in the split-f16 arithmetic on channels-last planes (vs_nhwc_conv_f16x3_layer): hi / lo [B,T,F,64] f16,
    scale2 the device pair {s, 1/s}; -> (out_hi, out_lo, out_scale2, amax_out [1024] uint32 view as int32, scratch).
    amax_in: int32 tensor holding float bit patterns whose maximum is max |x| (default: computed here from the planes)."""
    lib = _lib.load()
    _dev_check(hi, "hi", torch.float16)
    _dev_check(lo, "lo", torch.float16)
    for n, t in (("w", w), ("bn_scale", bn_scale), ("bn_shift", bn_shift), ("scale2", scale2)):
        _dev_check(t, n)
    B, T, F, C = hi.shape
    KT, KF = w.shape[2], w.shape[3]
    if amax_in is None:
        m = ((hi.float() + lo.float()) * scale2[1]).abs().max().reshape(1)
        amax_in = m.view(torch.int32)
    ready = scratch is not None
    if scratch is None:
        scratch = torch.empty(lib.vs_nhwc_conv_f16x3_scratch_bytes(KT, KF), dtype=torch.uint8, device=hi.device)
    out_hi, out_lo = torch.empty_like(hi), torch.empty_like(lo)
    out_s2 = torch.empty(2, dtype=torch.float32, device=hi.device)
    amax_out = torch.zeros(1024, dtype=torch.int32, device=hi.device)
    check(lib.vs_nhwc_conv_f16x3_layer(_p(hi), _p(lo), _p(scale2), _p(amax_in), amax_in.numel(), _p(w), _p(bn_scale), _p(bn_shift),
                                       _p(scratch), int(ready), _p(out_hi), _p(out_lo), _p(out_s2), _p(amax_out),
                                       B, T, F, KT, KF, dil, ACT_CODES[act], _stream()), "vs_nhwc_conv_f16x3_layer")
    return out_hi, out_lo, out_s2, amax_out, scratch


def nhwc_conv_pack(w, transpose_flip: bool = False):
    """w [64,64,KT,KF] fp32 -> the register-fragment order vs_nhwc_conv / vs_nhwc_conv_dy take (uint8 tensor)."""
    lib = _lib.load()
    _dev_check(w, "w")
    KT, KF = w.shape[2], w.shape[3]
    packed = torch.empty(lib.vs_nhwc_conv_packed_bytes(KT, KF), dtype=torch.uint8, device=w.device)
    check(lib.vs_nhwc_conv_pack(_p(w), _p(packed), KT, KF, int(transpose_flip), _stream()), "vs_nhwc_conv_pack")
    return packed


def nhwc_conv_first(x, w, scale, shift, act: str, stats: bool = False):
    """cnn1: x [B,T,F] fp32, w [64,1,1,7] -> act(conv * scale + shift) as [B,T,F,64] bf16 (+ [64,2] statistics)."""
    lib = _lib.load()
    for n, t in (("x", x), ("w", w), ("scale", scale), ("shift", shift)):
        _dev_check(t, n)
    B, T, F = x.shape
    out = torch.empty(B, T, F, 64, dtype=torch.bfloat16, device=x.device)
    st = torch.zeros(64, 64, 2, dtype=torch.float64, device=x.device) if stats else None
    check(lib.vs_nhwc_conv_first(_p(x), _p(w), _p(scale), _p(shift), _p(out), B, T, F, ACT_CODES[act], _p(st), _stream()),
          "vs_nhwc_conv_first")
    return (out, st.sum(0)) if stats else out


def bn_finalize(stats, count: float, gamma, beta, running_mean=None, running_var=None, eps: float = 1e-5, momentum: float = 0.1):
    """Train-mode BatchNorm2d constants from epilogue statistics (vs_bn_finalize): stats [slots, C, 2] float64 {sum, sum of
    squares} (folded in place) -> (scale, shift, mean, invstd); running_mean / running_var are updated in place when given."""
    lib = _lib.load()
    _dev_check(stats, "stats", torch.float64)
    for n, t in (("gamma", gamma), ("beta", beta)):
        _dev_check(t, n)
    slots, C = stats.shape[0], stats.shape[1]
    scale, shift, mean, invstd = (torch.empty(C, dtype=torch.float32, device=stats.device) for _ in range(4))
    check(lib.vs_bn_finalize(_p(stats), slots, float(count), C, _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, momentum,
                             _p(scale), _p(shift), _p(mean), _p(invstd), _stream()), "vs_bn_finalize")
    return scale, shift, mean, invstd


def nhwc_first_moments(x):
    """x [B,T,F] fp32 -> the 35 float64 moments of its seven zero-padded shifts (vs_nhwc_first_moments)."""
    lib = _lib.load()
    _dev_check(x, "x")
    B, T, F = x.shape
    mom = torch.empty(35, dtype=torch.float64, device=x.device)
    check(lib.vs_nhwc_first_moments(_p(x), B, T, F, _p(mom), _stream()), "vs_nhwc_first_moments")
    return mom


def nhwc_first_stats(mom, w, bias, count: float):
    """moments -> [1, 64, 2] float64 {sum, sum of squares} of z1 = conv(x) + bias (one slot of vs_bn_finalize)."""
    lib = _lib.load()
    _dev_check(mom, "moments", torch.float64)
    stats = torch.empty(1, 64, 2, dtype=torch.float64, device=mom.device)
    check(lib.vs_nhwc_first_stats(_p(mom), _p(w), _p(bias), float(count), _p(stats), _stream()), "vs_nhwc_first_stats")
    return stats


def nhwc_first_bwd(da, x, w, bias, act: str, training: bool, scale, shift, mean, invstd):
    """cnn1 + BatchNorm + activation backward in one pass over da [B,T,F,64] bf16 -> (dw [64,7], dgamma, dbeta, dbias)."""
    lib = _lib.load()
    _dev_check(da, "da", torch.bfloat16)
    _dev_check(x, "x")
    B, T, F = x.shape
    dev = x.device
    dg, db, dbias = (torch.empty(64, dtype=torch.float32, device=dev) for _ in range(3))
    dw = torch.empty(64, 7, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.vs_nhwc_first_bwd_scratch_doubles(), dtype=torch.float64, device=dev)
    check(lib.vs_nhwc_first_bwd(_p(da), _p(x), _p(w), _p(bias), B, T, F, ACT_CODES[act], BN_TRAIN if training else BN_EVAL, _p(scale), _p(shift),
                                _p(mean), _p(invstd), _p(dg), _p(db), _p(dbias), _p(dw), _p(scratch), _stream()), "vs_nhwc_first_bwd")
    return dw, dg, db, dbias


def nhwc_bn_apply(z, scale, shift, act: str):
    lib = _lib.load()
    _dev_check(z, "z", torch.bfloat16)
    a = torch.empty_like(z)
    check(lib.vs_nhwc_bn_apply(_p(z), _p(a), z.numel() // 64, ACT_CODES[act], _p(scale), _p(shift), _stream()), "vs_nhwc_bn_apply")
    return a


def nhwc_conv_last(x, w, scale, shift, act: str):
    """cnn8 + transpose/view: x [B,T,F,64] bf16, w [8,64,1,1] -> [B,T,8F] fp32 (feature index c*F+f)."""
    lib = _lib.load()
    _dev_check(x, "x", torch.bfloat16)
    B, T, F, _ = x.shape
    out = torch.empty(B, T, 8 * F, dtype=torch.float32, device=x.device)
    check(lib.vs_nhwc_conv_last(_p(x), _p(w), _p(scale), _p(shift), _p(out), B, T, F, ACT_CODES[act], _stream()), "vs_nhwc_conv_last")
    return out


def nhwc_conv_last_pre(z7, pre_scale, pre_shift, pre_act: str, w, scale, shift, stats: bool = False):
    """cnn8 on the un-normalised cnn7 output z7 [B,T,F,64] bf16 (its BatchNorm + activation applied on the way in)
    -> [B,T,8F] fp32 unactivated (+ [8,2] float64 {sum, sum of squares} per output channel)."""
    lib = _lib.load()
    _dev_check(z7, "z7", torch.bfloat16)
    B, T, F, _ = z7.shape
    out = torch.empty(B, T, 8 * F, dtype=torch.float32, device=z7.device)
    st = torch.zeros(64, 8, 2, dtype=torch.float64, device=z7.device) if stats else None
    check(lib.vs_nhwc_conv_last_pre(_p(z7), _p(pre_scale), _p(pre_shift), ACT_CODES[pre_act], _p(w), _p(scale), _p(shift), _p(out), _p(st),
                                    B, T, F, _stream()), "vs_nhwc_conv_last_pre")
    return (out, st.sum(0)) if stats else out


def nhwc_conv_wgrad(dz, x, KT: int, KF: int, dil: int):
    """dz, x [B,T,F,64] bf16 -> dw [64,64,KT,KF] fp32."""
    lib = _lib.load()
    _dev_check(dz, "dz", torch.bfloat16)
    _dev_check(x, "x", torch.bfloat16)
    B, T, F, _ = x.shape
    part = torch.empty(lib.vs_nhwc_conv_wgrad_partial_floats(KT, KF), dtype=torch.float32, device=x.device)
    dw = torch.empty(64, 64, KT, KF, dtype=torch.float32, device=x.device)
    check(lib.vs_nhwc_conv_wgrad(_p(dz), _p(x), _p(part), _p(dw), B, T, F, KT, KF, dil, _stream()), "vs_nhwc_conv_wgrad")
    return dw


def nhwc_bn_act_bwd(da, z, act: str, training: bool, scale, shift, mean, invstd):
    """da, z [.., 64] bf16 -> (dz bf16, dgamma, dbeta, dbias)."""
    lib = _lib.load()
    _dev_check(da, "da", torch.bfloat16)
    _dev_check(z, "z", torch.bfloat16)
    dev = z.device
    dz = torch.empty_like(da)
    dg, db, dbias = (torch.empty(64, dtype=torch.float32, device=dev) for _ in range(3))
    stats = torch.empty(64 * 64 * 2, dtype=torch.float64, device=dev)
    coef = torch.empty(192, dtype=torch.float32, device=dev)
    check(lib.vs_nhwc_bn_act_bwd(_p(da), _p(z), _p(dz), z.numel() // 64, ACT_CODES[act], BN_TRAIN if training else BN_EVAL,
                                 _p(scale), _p(shift), _p(mean), _p(invstd), _p(dg), _p(db), _p(dbias), _p(stats), _p(coef), _stream()),
          "vs_nhwc_bn_act_bwd")
    return dz, dg, db, dbias


def nhwc_bn_act_bwd_first(da, z, x, act: str, training: bool, scale, shift, mean, invstd):
    """cnn1: da, z [B,T,F,64] bf16, x [B,T,F] fp32 -> (dw [64,7], dgamma, dbeta, dbias)."""
    lib = _lib.load()
    dev = z.device
    B, T, F = x.shape
    dg, db, dbias = (torch.empty(64, dtype=torch.float32, device=dev) for _ in range(3))
    dw = torch.empty(64, 7, dtype=torch.float32, device=dev)
    stats = torch.empty(64 * 64 * 2, dtype=torch.float64, device=dev)
    coef = torch.empty(192, dtype=torch.float32, device=dev)
    acc = torch.empty(448, dtype=torch.float64, device=dev)
    check(lib.vs_nhwc_bn_act_bwd_first(_p(da), _p(z), _p(x), B, T, F, ACT_CODES[act], BN_TRAIN if training else BN_EVAL,
                                       _p(scale), _p(shift), _p(mean), _p(invstd), _p(dg), _p(db), _p(dbias), _p(dw), _p(stats),
                                       _p(coef), _p(acc), _stream()), "vs_nhwc_bn_act_bwd_first")
    return dw, dg, db, dbias


def nhwc_conv_last_bwd(dz8, w, a7):
    """dz8 [B,T,8F] fp32, w [8,64,1,1], a7 [B,T,F,64] bf16 -> (din [B,T,F,64] bf16, dw [8,64])."""
    lib = _lib.load()
    _dev_check(dz8, "dz8")
    _dev_check(a7, "a7", torch.bfloat16)
    B, T, F, _ = a7.shape
    din = torch.empty_like(a7)
    part = torch.empty(lib.vs_nhwc_conv_last_bwd_blocks() * 512, dtype=torch.float32, device=a7.device)
    dw = torch.empty(8, 64, dtype=torch.float32, device=a7.device)
    check(lib.vs_nhwc_conv_last_bwd(_p(dz8), _p(w), _p(a7), _p(din), _p(part), _p(dw), B, T, F, _stream()), "vs_nhwc_conv_last_bwd")
    return din, dw


def nhwc_conv_dy(dz, packed, z, act: str, bn_scale, bn_shift, bn_mean, bn_invstd, KT: int, KF: int, dil: int):
    """Data gradient with the dy epilogue: dz [B,T,F,64] bf16, packed = transposed/flipped weights, z = the lower layer's
    conv output -> (dy bf16, stats [64 slots, 64, 2] float64 with the per-channel sums of dy and dy * xhat)."""
    lib = _lib.load()
    _dev_check(dz, "dz", torch.bfloat16)
    _dev_check(z, "z", torch.bfloat16)
    B, T, F, _ = dz.shape
    dy = torch.empty_like(dz)
    stats = torch.zeros(64, 64, 2, dtype=torch.float64, device=dz.device)
    check(lib.vs_nhwc_conv_dy(_p(dz), _p(packed), _p(dy), _p(z), ACT_CODES[act], _p(bn_scale), _p(bn_shift), _p(bn_mean), _p(bn_invstd),
                              _p(stats), B, T, F, KT, KF, dil, _stream()), "vs_nhwc_conv_dy")
    return dy, stats


def nhwc_conv_last_bwd_dy(dz8, w, a7, z7, act: str, bn_scale, bn_shift, bn_mean, bn_invstd):
    """cnn8 backward with the dy epilogue -> (dy7 bf16, dw [8,64], stats)."""
    lib = _lib.load()
    _dev_check(dz8, "dz8")
    if a7 is not None:                      # None: recomputed from z7 inside the kernel
        _dev_check(a7, "a7", torch.bfloat16)
    _dev_check(z7, "z7", torch.bfloat16)
    B, T, F, _ = z7.shape
    dy = torch.empty_like(z7)
    part = torch.empty(lib.vs_nhwc_conv_last_bwd_blocks() * 512, dtype=torch.float32, device=z7.device)
    dw = torch.empty(8, 64, dtype=torch.float32, device=z7.device)
    stats = torch.zeros(64, 64, 2, dtype=torch.float64, device=z7.device)
    check(lib.vs_nhwc_conv_last_bwd_dy(_p(dz8), _p(w), _p(a7), _p(dy), _p(part), _p(dw), _p(z7), ACT_CODES[act], _p(bn_scale), _p(bn_shift),
                                       _p(bn_mean), _p(bn_invstd), _p(stats), B, T, F, _stream()), "vs_nhwc_conv_last_bwd_dy")
    return dy, dw, stats


def nhwc_bn_bwd_from_dy(dy, z, stats, training: bool, scale, mean, invstd):
    """Second pass of the BatchNorm backward from dy and its sums -> (dz bf16, dgamma, dbeta, dbias)."""
    lib = _lib.load()
    _dev_check(dy, "dy", torch.bfloat16)
    _dev_check(z, "z", torch.bfloat16)
    dev = z.device
    dz = torch.empty_like(dy)
    dg, db, dbias = (torch.empty(64, dtype=torch.float32, device=dev) for _ in range(3))
    coef = torch.empty(192, dtype=torch.float32, device=dev)
    check(lib.vs_nhwc_bn_bwd_from_dy(_p(dy), _p(z), _p(dz), z.numel() // 64, BN_TRAIN if training else BN_EVAL, _p(scale), _p(mean),
                                     _p(invstd), _p(dg), _p(db), _p(dbias), _p(stats), _p(coef), _stream()), "vs_nhwc_bn_bwd_from_dy")
    return dz, dg, db, dbias


def nhwc_bn_bwd_first_from_dy(dy, z, x, stats, training: bool, scale, mean, invstd):
    """cnn1, from dy -> (dw [64,7], dgamma, dbeta, dbias)."""
    lib = _lib.load()
    dev = z.device
    B, T, F = x.shape
    dg, db, dbias = (torch.empty(64, dtype=torch.float32, device=dev) for _ in range(3))
    dw = torch.empty(64, 7, dtype=torch.float32, device=dev)
    coef = torch.empty(192, dtype=torch.float32, device=dev)
    acc = torch.empty(448, dtype=torch.float64, device=dev)
    check(lib.vs_nhwc_bn_bwd_first_from_dy(_p(dy), _p(z), _p(x), B, T, F, BN_TRAIN if training else BN_EVAL, _p(scale), _p(mean),
                                           _p(invstd), _p(dg), _p(db), _p(dbias), _p(dw), _p(stats), _p(coef), _p(acc), _stream()),
          "vs_nhwc_bn_bwd_first_from_dy")
    return dw, dg, db, dbias


def gemm_bf16(A, B, M: int, N: int, K: int, a_kmajor: bool = False, b_kmajor: bool = False, rowbias=None, group: int = 1,
              out=None, accumulate: bool = False):
    """C[M,N] (+)= op(A) op(B) on bf16 operands (csrc/gemm_bf16.hip); A, B 2-D bf16 tensors (row form [rows, ld >= K padded
    to 64], K-major form [K, ld]); fp32 result."""
    lib = _lib.load()
    _dev_check(A, "A", torch.bfloat16)
    _dev_check(B, "B", torch.bfloat16)
    C = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=A.device)
    check(lib.vs_gemm_bf16(int(a_kmajor), int(b_kmajor), _p(A), A.shape[1], _p(B), B.shape[1], _p(C), C.shape[1], M, N, K,
                           _p(rowbias), rowbias.shape[1] if rowbias is not None else 0, group, int(accumulate), _stream()), "vs_gemm_bf16")
    return C


def gemm_bf16_gated(A, B, gate, M: int, N: int, K: int, a_kmajor: bool = False, b_kmajor: bool = False, out=None):
    """C = op(A) op(B) where gate > 0, else 0 (vs_gemm_bf16_gated: the head's data gradients of vs_backward); gate fp32 [M, ldg >= N];
    out: a preallocated [M, ldc >= N] fp32 result (columns >= N are left untouched)."""
    lib = _lib.load()
    _dev_check(A, "A", torch.bfloat16)
    _dev_check(B, "B", torch.bfloat16)
    _dev_check(gate, "gate", torch.float32)
    C = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=A.device)
    check(lib.vs_gemm_bf16_gated(int(a_kmajor), int(b_kmajor), _p(A), A.shape[1], _p(B), B.shape[1], _p(C), C.shape[1], M, N, K,
                                 _p(gate), gate.shape[1], _stream()), "vs_gemm_bf16_gated")
    return C


def gemm_bf16_split(A, B, M: int, N: int, K: int, split_m: int, a_kmajor: bool = False, b_kmajor: bool = False, ldc: Optional[int] = None):
    """The same contraction stored into two matrices stacked along M (vs_gemm_bf16_split: the dW_ih store of vs_backward):
    returns (C [split_m, ldc], C2 [M - split_m, ldc]); columns >= N are left untouched (NaN-filled here so a test sees a
    stray store)."""
    lib = _lib.load()
    _dev_check(A, "A", torch.bfloat16)
    _dev_check(B, "B", torch.bfloat16)
    ldc = N if ldc is None else ldc
    C = torch.full((split_m, ldc), float("nan"), dtype=torch.float32, device=A.device)
    C2 = torch.full((M - split_m, ldc), float("nan"), dtype=torch.float32, device=A.device)
    check(lib.vs_gemm_bf16_split(int(a_kmajor), int(b_kmajor), _p(A), A.shape[1], _p(B), B.shape[1], _p(C), _p(C2), ldc, split_m,
                                 M, N, K, 0, _stream()), "vs_gemm_bf16_split")
    return C, C2


def cvt_rows_bf16(x, K: int, Kp: int):
    lib = _lib.load()
    _dev_check(x, "x")
    out = torch.empty(x.shape[0], Kp, dtype=torch.bfloat16, device=x.device)
    check(lib.vs_cvt_rows_bf16(_p(x), x.shape[0], K, x.shape[1], _p(out), Kp, _stream()), "vs_cvt_rows_bf16")
    return out


def lstm_status(dims: VsDims, tape: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None) -> int:
    """vs_lstm_status: 0 = the persistent BiLSTM kernels that last ran on these buffers completed, 1 = one gave up
    (output NaN-poisoned).  Synchronises the current stream."""
    lib = _lib.load()
    rc = lib.vs_lstm_status(ctypes.byref(dims), _p(tape), tape.numel() if tape is not None else 0,
                            _p(workspace), workspace.numel() if workspace is not None else 0, _stream())
    if rc < 0:
        check(rc, "vs_lstm_status")
    return rc
