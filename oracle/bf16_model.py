"""TEST INFRASTRUCTURE (oracle/): what bf16 STORAGE costs the parameter gradients of the path, independent of any
kernel.  The reference forward (models/voicesplit/model.py:66-89: conv stack with batch-statistics BatchNorm, d-vector
concat, BiLSTM, head) in fp64 autograd with bf16 rounding injected exactly where the VS_MATH_BF16 configuration rounds:
the 64->64 conv weights, z = conv + bias of cnn2..cnn7 (cnn1's is recomputed from x, never stored), a = act(BN(z)) of
cnn1..cnn7, the gradients flowing back through those, and the operands
of the LSTM input GEMM, of fc1 / fc2 and of their backward contractions.  The recurrent product is rounded as the kernels round it (h, W_hh to f16
forward; gate gradients, W_hh^T to bf16 in the BPTT).  Accumulation, statistics and the gate arithmetic are exact.
The difference between these gradients and the unrounded ones is the envelope a correct bf16 implementation lives in;
tests/test_gpu_bf16.py holds the HIP path to it.  (tools/bf16_error_sources.py prints the breakdown by rounding point.)"""
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import reference_forward as R


class _RoundSTE(torch.autograd.Function):
    """forward: round to bf16 (or pass through); backward: round the gradient to bf16 (or pass through)."""

    @staticmethod
    def forward(ctx, x, rf, rg):
        ctx.rg = rg
        return x.to(torch.bfloat16).to(x.dtype) if rf else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.rg else g), None, None


class _RecurrentProduct(torch.autograd.Function):
    """h_{t-1} @ W_hh^T as the VS_MATH_BF16 recurrence computes it (csrc/lstm.hip, lstm16_*): forward operands rounded
    to f16, the BPTT's product W_hh^T @ dgates and the dW_hh contraction on bf16-rounded operands."""

    @staticmethod
    def forward(ctx, h, w, on):
        ctx.save_for_backward(h, w)
        ctx.on = on
        if on:
            return h.to(torch.float16).to(h.dtype) @ w.to(torch.float16).to(w.dtype).t()
        return h @ w.t()

    @staticmethod
    def backward(ctx, g):
        h, w = ctx.saved_tensors
        if ctx.on:
            gb = g.to(torch.bfloat16).to(g.dtype)
            return gb @ w.to(torch.bfloat16).to(w.dtype), gb.t() @ h.to(torch.bfloat16).to(h.dtype), None
        return g @ w, g.t() @ h, None


def _rnd(x, fwd: bool, bwd: bool = False):
    return _RoundSTE.apply(x, bool(fwd), bool(bwd)) if (fwd or bwd) else x


def gradients(sd: Dict[str, torch.Tensor], x: torch.Tensor, dvec: torch.Tensor, w: torch.Tensor, act: str = "mish",
              bf16: bool = True, dtype=torch.float64) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """d(sum(mask * w))/d(parameter) in fp64, training-mode BatchNorm; bf16=True injects the storage roundings of the
    bf16 configuration, bf16=False is the exact reference.  Returns (gradients by state_dict key, mask).
    dtype: the carrier arithmetic -- float64 everywhere except the full-size B = 8 fixture of oracle/make_golden.py
    --bf16-envelope, whose autograd graph does not fit this container's memory in fp64 and runs in float32 (its 6e-8
    rounding is four orders below the bf16 roundings under study)."""
    P = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    x, dvec, w = x.to(dtype), dvec.to(dtype), w.to(dtype)
    B, T, _ = x.shape
    h = x.unsqueeze(1)
    for i, spec in enumerate(R.CONV_TABLE):
        wt = P[f"conv.{spec.conv_idx}.weight"]
        if bf16 and 1 <= i <= 6:
            wt = _rnd(wt, True)
        z = F.conv2d(h, wt, P[f"conv.{spec.conv_idx}.bias"], padding=((spec.kt // 2) * spec.dil_t, spec.kf // 2), dilation=(spec.dil_t, 1))
        if 1 <= i < 7:          # (cnn1's z is never stored: the path recomputes it from x in fp32, forward and backward -- round 4)
            z = _rnd(z, bf16, bf16)
        m = z.mean((0, 2, 3), keepdim=True)
        v = ((z - m) ** 2).mean((0, 2, 3), keepdim=True)
        y = (z - m) / torch.sqrt(v + R.BN_EPS) * P[f"conv.{spec.bn_idx}.weight"].view(1, -1, 1, 1) + P[f"conv.{spec.bn_idx}.bias"].view(1, -1, 1, 1)
        h = R.activation(y, act)
        if i < 7:
            h = _rnd(h, bf16, bf16)
    feat = h.transpose(1, 2).reshape(B, T, -1)
    K = feat.shape[2]
    outs = []
    for sfx, rev in (("", False), ("_reverse", True)):
        w_ih, w_hh = P["lstm.weight_ih_l0" + sfx], P["lstm.weight_hh_l0" + sfx]
        bias = P["lstm.bias_ih_l0" + sfx] + P["lstm.bias_hh_l0" + sfx]
        fa, wa = (_rnd(feat, True), _rnd(w_ih[:, :K], True)) if bf16 else (feat, w_ih[:, :K])
        xg = fa @ wa.t() + (dvec @ w_ih[:, K:].t()).unsqueeze(1) + bias
        xg = _rnd(xg, False, bf16)                       # the gate gradients are rounded for the two backward contractions
        H = w_hh.shape[1]
        hh, c = xg.new_zeros(B, H), xg.new_zeros(B, H)
        out = [None] * T
        for t in (range(T - 1, -1, -1) if rev else range(T)):
            g = xg[:, t] + _RecurrentProduct.apply(hh, w_hh, bool(bf16))
            i_, f_, gg, o_ = g.split(H, dim=1)
            c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(gg)
            hh = torch.sigmoid(o_) * torch.tanh(c)
            out[t] = hh
        outs.append(torch.stack(out, 1))
    lo = torch.relu(torch.cat(outs, 2))
    # head: both operands of fc1 / fc2 rounded (forward and, through the saved tensors, the two backward contractions of
    # each); the gradient arriving at each pre-activation is rounded as the operand of those contractions
    h1 = torch.relu(_rnd(F.linear(_rnd(lo, bf16), _rnd(P["fc1.weight"], bf16), P["fc1.bias"]), False, bf16))
    mask = torch.sigmoid(_rnd(F.linear(_rnd(h1, bf16), _rnd(P["fc2.weight"], bf16), P["fc2.bias"]), False, bf16))
    (mask * w).sum().backward()
    return {k: p.grad for k, p in P.items() if p.grad is not None}, mask.detach()
