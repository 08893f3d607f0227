"""CPU oracle for the VoiceSplit / VoiceFilter mask-prediction forward pass.

TEST INFRASTRUCTURE ONLY.  Nothing under ``voicesplit_amd/`` may import this
file: it is the checker that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` compare the HIP path against, never the
thing that is shipped or measured as the product.

What it restates (all citations relative to the upstream reference tree):

* layer table of the conv stack      models/voicesplit/model.py:15-52
                                     models/voicefilter/model.py:17-54
* Mish                               utils/generic_utils.py:395-399
* forward: unsqueeze / conv / transpose+view / d-vector repeat+cat /
  BiLSTM / relu / fc1 / relu / fc2 / sigmoid
                                     models/voicesplit/model.py:66-89
                                     models/voicefilter/model.py:67-90

The arithmetic itself lives in PyTorch (``torch.nn.Conv2d``, ``BatchNorm2d``,
``LSTM``, ``Linear``, ``F.softplus``), a third-party dependency of the
reference (``requirements.txt:7`` pins torch==1.0.1; this image has 2.10).  The
restatement therefore calls the same functional ops on CPU for the conv/BN/FC
stages and writes the LSTM recurrence out explicitly (gate order i,f,g,o,
``b_ih + b_hh``), which ``tests/test_oracle.py`` checks against ``nn.LSTM``.

Parity pinning: the reference ships no tests, checkpoints or golden tensors
(SURVEY.md §4), so this oracle is pinned against the reference *itself*:
``oracle/make_golden.py`` imports ``/root/reference`` in the build container,
runs the upstream ``nn.Module`` on seeded inputs and commits the outputs under
``tests/golden/``; ``tests/test_oracle.py`` requires this file to reproduce
those tensors bit-for-bit (and, when ``/root/reference`` is present, compares
against the live upstream module again).

The oracle works in whatever dtype its inputs carry (fp32 to mirror the
reference, fp64 to act as ground truth).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, NamedTuple, Optional

import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default, models/voicesplit/model.py:19
BN_MOMENTUM = 0.1


class ConvSpec(NamedTuple):
    """One row of the conv-stack table."""
    conv_idx: int      # index of the Conv2d inside the reference's nn.Sequential
    bn_idx: int        # index of the BatchNorm2d
    cin: int
    cout: int
    kt: int            # taps along time   (kernel_size[0])
    kf: int            # taps along freq   (kernel_size[1])
    dil_t: int         # dilation along time; freq dilation is always 1


# models/voicesplit/model.py:15-52 -- "same" zero padding in every layer:
# ZeroPad2d((kf//2, kf//2, dil*(kt//2), dil*(kt//2))).
CONV_TABLE: List[ConvSpec] = [
    ConvSpec(1, 2, 1, 64, 1, 7, 1),      # cnn1  :17-19
    ConvSpec(5, 6, 64, 64, 7, 1, 1),     # cnn2  :21-23
    ConvSpec(9, 10, 64, 64, 5, 5, 1),    # cnn3  :26-28
    ConvSpec(13, 14, 64, 64, 5, 5, 2),   # cnn4  :31-33
    ConvSpec(17, 18, 64, 64, 5, 5, 4),   # cnn5  :36-38
    ConvSpec(21, 22, 64, 64, 5, 5, 8),   # cnn6  :41-43
    ConvSpec(25, 26, 64, 64, 5, 5, 16),  # cnn7  :46-48
    ConvSpec(28, 29, 64, 8, 1, 1, 1),    # cnn8  :51-52 (no padding)
]


def mish(x: torch.Tensor) -> torch.Tensor:
    """utils/generic_utils.py:395-399: ``inp * tanh(softplus(inp))``.

    ``F.softplus`` defaults: beta=1, threshold=20 (softplus(x)=x for x>20).
    """
    return x * torch.tanh(F.softplus(x))


def activation(x: torch.Tensor, act: str) -> torch.Tensor:
    if act == "mish":       # VoiceSplit
        return mish(x)
    if act == "relu":       # VoiceFilter
        return torch.relu(x)
    raise ValueError(act)


def conv_layer(x, sd, spec: ConvSpec, act: str, training: bool,
               bn_out: Optional[dict] = None, pre: Optional[dict] = None, name: str = "",
               gate: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ZeroPad2d -> Conv2d -> BatchNorm2d -> activation for one table row."""
    w = sd[f"conv.{spec.conv_idx}.weight"]
    b = sd[f"conv.{spec.conv_idx}.bias"]
    pf, pt = spec.kf // 2, spec.dil_t * (spec.kt // 2)
    x = F.pad(x, (pf, pf, pt, pt))
    x = F.conv2d(x, w, b, dilation=(spec.dil_t, 1))
    if pre is not None:          # conv+bias output before BatchNorm (what the HIP tape keeps as z)
        pre[name] = x
    p = f"conv.{spec.bn_idx}."
    # nn.BatchNorm2d == F.batch_norm: eval -> running stats; train
    # (train.py:84 model.train()) -> batch statistics over (B,T,F), running
    # stats updated in place with the unbiased variance and momentum 0.1.
    rm = sd[p + "running_mean"].clone()
    rv = sd[p + "running_var"].clone()
    x = F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], training, BN_MOMENTUM, BN_EPS)
    if training and bn_out is not None:
        bn_out[p + "running_mean"] = rm
        bn_out[p + "running_var"] = rv
        bn_out[p + "num_batches_tracked"] = sd[p + "num_batches_tracked"] + 1
    if pre is not None:          # BatchNorm output = activation input (where a ReLU has its kink)
        pre["y" + name[1:]] = x
    if gate is not None and act == "relu":
        # branch-consistent ReLU for gradient parity tests: the caller fixes which side of the
        # kink every element is on (see reference_backward.relu_gates)
        return x * gate.to(x.dtype)
    return activation(x, act)


def conv_stack(x, sd, act: str, training: bool = False, taps: Optional[dict] = None,
               bn_out: Optional[dict] = None, pre: Optional[dict] = None,
               gates: Optional[dict] = None) -> torch.Tensor:
    """[B,T,F] -> [B,8,T,F]  (models/voicesplit/model.py:68-70)."""
    x = x.unsqueeze(1)
    for i, spec in enumerate(CONV_TABLE):
        x = conv_layer(x, sd, spec, act, training, bn_out, pre, f"z{i + 1}",
                       None if gates is None else gates.get(f"y{i + 1}"))
        if taps is not None:
            taps[f"cnn{i + 1}"] = x
    return x


def lstm_direction(xs, w_ih, w_hh, b_ih, b_hh, reverse: bool, taps: Optional[dict] = None) -> torch.Tensor:
    """One direction of nn.LSTM(batch_first=True), zero initial state.

    Gate order in the stacked weights is i, f, g, o;
    c' = sigmoid(f) * c + sigmoid(i) * tanh(g);  h' = sigmoid(o) * tanh(c').
    """
    B, T, _ = xs.shape
    H = w_hh.shape[1]
    h = xs.new_zeros(B, H)
    c = xs.new_zeros(B, H)
    xg = xs @ w_ih.t() + (b_ih + b_hh)          # [B,T,4H]
    if taps is not None:                        # gate pre-activation inputs (what the HIP path calls xg)
        taps["xg_reverse" if reverse else "xg"] = xg
    out = xs.new_empty(B, T, H)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = xg[:, t] + h @ w_hh.t()
        i, f, gg, o = g.split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


def bilstm_aten(xs, sd) -> torch.Tensor:
    """The same BiLSTM through ``nn.LSTM`` itself (what the reference calls,
    models/voicesplit/model.py:57-61,82).  Bit-identical to the upstream module
    on the same torch build; used for golden pinning and for the timed CPU
    baseline.  ``bilstm`` below is the explicit recurrence it must agree with."""
    import torch.nn as nn
    H = sd["lstm.weight_hh_l0"].shape[1]
    m = nn.LSTM(xs.shape[2], H, batch_first=True, bidirectional=True).to(xs.dtype)
    m.load_state_dict({k[len("lstm."):]: v for k, v in sd.items() if k.startswith("lstm.")})
    with torch.no_grad():
        y, _ = m(xs)
    return y


def bilstm(xs, sd, taps: Optional[dict] = None) -> torch.Tensor:
    """models/voicesplit/model.py:57-61,82 -> [B,T,2H] = cat(fwd, bwd)."""
    f = lstm_direction(xs, sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"],
                       sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"], False, taps)
    b = lstm_direction(xs, sd["lstm.weight_ih_l0_reverse"], sd["lstm.weight_hh_l0_reverse"],
                       sd["lstm.bias_ih_l0_reverse"], sd["lstm.bias_hh_l0_reverse"], True, taps)
    return torch.cat((f, b), dim=2)


def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, dvec: torch.Tensor,
            act: str = "mish", training: bool = False,
            bn_out: Optional[dict] = None,
            lstm_impl: str = "aten") -> "OrderedDict[str, torch.Tensor]":
    """Full forward; returns every stage so tests can localise a mismatch.

    keys: cnn1..cnn8, lstm_in [B,T,8F+E], lstm_out [B,T,2H], fc1 (post-relu),
    logits (pre-sigmoid), mask.
    """
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    y = conv_stack(x, sd, act, training, out, bn_out)      # [B,8,T,F]
    y = y.transpose(1, 2).contiguous()                    # :72
    y = y.view(y.size(0), y.size(1), -1)                  # :74  index = c*F + f
    e = dvec.unsqueeze(1).repeat(1, y.size(1), 1)         # :77-78
    y = torch.cat((y, e), dim=2)                          # :81
    out["lstm_in"] = y
    y = bilstm_aten(y, sd) if lstm_impl == "aten" else bilstm(y, sd)   # :82
    out["lstm_out"] = y
    y = torch.relu(y)                                     # :83
    y = torch.relu(F.linear(y, sd["fc1.weight"], sd["fc1.bias"]))   # :84-85
    out["fc1"] = y
    y = F.linear(y, sd["fc2.weight"], sd["fc2.bias"])     # :86
    out["logits"] = y
    out["mask"] = torch.sigmoid(y)                        # :87
    return out


# ---------------------------------------------------------------------------
# Seeded parameters and inputs (SURVEY.md §8(d) "Synthetic inputs")
# ---------------------------------------------------------------------------

def default_dims() -> dict:
    """config.json:37-42,86 defaults."""
    return dict(num_freq=601, emb_dim=256, lstm_dim=400, fc1_dim=600, fc2_dim=601)


def build_state_dict(dims: dict, seed: int = 0, randomize_bn: bool = True,
                     dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Reference default initialisation under ``torch.manual_seed(seed)``.

    Modules are created in the constructor order of
    models/voicesplit/model.py:15-64 (conv+BN x8, LSTM, fc1, fc2) so the CPU
    RNG stream is consumed exactly as the reference consumes it;
    ``tests/test_oracle.py`` asserts bit-equality with the upstream module
    when ``/root/reference`` is available.  With ``randomize_bn`` the BN
    buffers/affine are then re-drawn (running_mean~N(0,.1), running_var~U(.5,1.5),
    weight~U(.5,1.5), bias~N(0,.1)) so eval-mode BN is not an identity.
    """
    import torch.nn as nn
    torch.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for spec in CONV_TABLE:
        conv = nn.Conv2d(spec.cin, spec.cout, kernel_size=(spec.kt, spec.kf), dilation=(spec.dil_t, 1))
        bn = nn.BatchNorm2d(spec.cout)
        for k, v in conv.state_dict().items():
            sd[f"conv.{spec.conv_idx}.{k}"] = v
        for k, v in bn.state_dict().items():
            sd[f"conv.{spec.bn_idx}.{k}"] = v
    lstm = nn.LSTM(8 * dims["num_freq"] + dims["emb_dim"], dims["lstm_dim"],
                   batch_first=True, bidirectional=True)
    for k, v in lstm.state_dict().items():
        sd[f"lstm.{k}"] = v
    fc1 = nn.Linear(2 * dims["lstm_dim"], dims["fc1_dim"])
    fc2 = nn.Linear(dims["fc1_dim"], dims["fc2_dim"])
    for k, v in fc1.state_dict().items():
        sd[f"fc1.{k}"] = v
    for k, v in fc2.state_dict().items():
        sd[f"fc2.{k}"] = v
    if randomize_bn:
        g = torch.Generator().manual_seed(seed + 1000)
        for spec in CONV_TABLE:
            p = f"conv.{spec.bn_idx}."
            c = spec.cout
            sd[p + "running_mean"] = torch.randn(c, generator=g) * 0.1
            sd[p + "running_var"] = torch.rand(c, generator=g) + 0.5
            sd[p + "weight"] = torch.rand(c, generator=g) + 0.5
            sd[p + "bias"] = torch.randn(c, generator=g) * 0.1
    out = OrderedDict()
    for k, v in sd.items():
        v = v.detach().clone()
        out[k] = v.to(dtype) if v.is_floating_point() else v
    return out


def spread_logits(sd: dict, gain: float = 8.0) -> dict:
    """Default init gives mask in [0.46, 0.53]; scale the LSTM
    recurrent/FC weights so logits cover several units and parity is meaningful."""
    sd = OrderedDict(sd)
    for k in ("fc1.weight", "fc2.weight"):
        sd[k] = sd[k] * gain
    for k in list(sd):
        if k.startswith("lstm.weight_hh"):
            sd[k] = sd[k] * 2.0
    return sd


def synthetic_inputs(B: int, T: int, dims: dict, seed: int = 0, dtype=torch.float32):
    """spec ~ U[0,1] [B,T,F] (dB-normalised range, utils/audio_processor.py:537-544);
    dvec = L2-normalised randn [B,E]."""
    g = torch.Generator().manual_seed(seed + 77)
    spec = torch.rand(B, T, dims["num_freq"], generator=g)
    dvec = torch.randn(B, dims["emb_dim"], generator=g)
    dvec = dvec / dvec.norm(dim=1, keepdim=True)
    return spec.to(dtype), dvec.to(dtype)


def cast_state_dict(sd: dict, dtype) -> dict:
    return OrderedDict((k, v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items())
