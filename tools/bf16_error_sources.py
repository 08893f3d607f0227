#!/usr/bin/env python
"""What does bf16 STORAGE cost the parameter gradients of this network, independent of any kernel?  The whole path in
fp64 autograd (reference arithmetic, oracle/ tensors) with bf16 rounding injected exactly where the bf16 configuration
rounds: conv weights, z (conv + bias), a (BatchNorm + Mish), their gradients, and the operands of the LSTM input GEMM
and of its two backward contractions.  Everything else -- accumulation, statistics, recurrence, head -- is exact.
Result on the default-initialised mid-size fixture (CPU, seconds): the ideal-bf16 gradients already differ from the
unrounded ones by 0.3-0.55 of a tensor's maximum with cosine 0.94-0.99: the `sum(mask * w)` test loss sums 10^5 terms
of either sign, so its gradient is badly conditioned and amplifies the 2^-9 roundings by two orders of magnitude.
The GPU path's own numbers (tests/test_gpu_bf16.py) sit inside this envelope.  Recorded in profiles/r03_bf16_accuracy.md."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import reference_forward as R
from oracle import reference_backward as RB
torch.manual_seed(0)
dims = dict(num_freq=101, emb_dim=16, lstm_dim=24, fc1_dim=40, fc2_dim=101)
sd = {k: (v.double() if v.is_floating_point() else v) for k, v in R.build_state_dict(dims, 31).items()}
B, T, Fq = 4, 80, 101
x, dvec = R.synthetic_inputs(B, T, dims, 31); x = x.double(); dvec = dvec.double()
wl = RB.loss_weights(B, T, Fq, 31).double()

class RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rf, rg):
        ctx.rg = rg
        return x.to(torch.bfloat16).double() if rf else x.clone()
    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).double() if ctx.rg else g), None, None
def rnd(x, on, rg=False):
    return RoundSTE.apply(x, bool(on), bool(rg)) if (on or rg) else x

def run(cfg):
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and 'running' not in k}
    h = x.unsqueeze(1)
    for i, spec in enumerate(R.CONV_TABLE):
        w = P[f"conv.{spec.conv_idx}.weight"]
        if cfg.get('conv') and 1 <= i <= 6: w = rnd(w, True)
        z = F.conv2d(h, w, P[f"conv.{spec.conv_idx}.bias"], padding=((spec.kt // 2) * spec.dil_t, spec.kf // 2), dilation=(spec.dil_t, 1))
        if i < 7: z = rnd(z, cfg.get('conv'), cfg.get('conv'))
        m = z.mean((0, 2, 3), keepdim=True); v = ((z - m) ** 2).mean((0, 2, 3), keepdim=True)
        y = (z - m) / torch.sqrt(v + 1e-5) * P[f"conv.{spec.bn_idx}.weight"].view(1, -1, 1, 1) + P[f"conv.{spec.bn_idx}.bias"].view(1, -1, 1, 1)
        h = y * torch.tanh(F.softplus(y, threshold=20))
        if i < 7: h = rnd(h, cfg.get('conv'), cfg.get('conv'))
    feat = h.transpose(1, 2).reshape(B, T, -1)
    K = feat.shape[2]
    outs = []
    for sfx, rev in (("", False), ("_reverse", True)):
        w_ih = P["lstm.weight_ih_l0" + sfx]; w_hh = P["lstm.weight_hh_l0" + sfx]
        bias = P["lstm.bias_ih_l0" + sfx] + P["lstm.bias_hh_l0" + sfx]
        fa, wa = feat, w_ih[:, :K]
        if cfg.get('gemm'):
            fa, wa = rnd(feat, True, False), rnd(w_ih[:, :K], True, False)
        xg = fa @ wa.t() + (dvec @ w_ih[:, K:].t()).unsqueeze(1) + bias
        if cfg.get('gemm'):
            xg = rnd(xg, False, True)          # dxg rounded for the backward GEMMs
        H = w_hh.shape[1]
        hh = xg.new_zeros(B, H); c = xg.new_zeros(B, H); out = [None] * T
        for t in (range(T - 1, -1, -1) if rev else range(T)):
            g = xg[:, t] + hh @ w_hh.t()
            i_, f_, gg, o_ = g.split(H, dim=1)
            c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(gg)
            hh = torch.sigmoid(o_) * torch.tanh(c)
            out[t] = hh
        outs.append(torch.stack(out, 1))
    lo = torch.relu(torch.cat(outs, 2))
    h1 = torch.relu(F.linear(lo, P["fc1.weight"], P["fc1.bias"]))
    mask = torch.sigmoid(F.linear(h1, P["fc2.weight"], P["fc2.bias"]))
    (mask * wl).sum().backward()
    return {k: p.grad for k, p in P.items() if p.grad is not None}, mask.detach()

ref, mref = run({})
def report(name, cfg):
    g, m = run(cfg)
    rows = []
    for k in ref:
        if k.endswith('.bias') and k.startswith('conv') and int(k.split('.')[1]) in (1,5,9,13,17,21,25,28): continue
        a, b = g[k].reshape(-1), ref[k].reshape(-1)
        rows.append((float(a @ b / (a.norm() * b.norm())), float((a - b).abs().max() / b.abs().max()), k))
    print(f"{name:22s} min cos {min(rows)[0]:.4f} ({min(rows)[2]})  worst err {max(r[1] for r in rows):.3f}  mask max err {float((m-mref).abs().max()):.2e}")
report("conv stack bf16", dict(conv=1))
report("lstm gemms bf16", dict(gemm=1))
report("both", dict(conv=1, gemm=1))
