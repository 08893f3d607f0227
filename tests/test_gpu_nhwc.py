"""Channels-last bf16 kernels of the VS_MATH_BF16 path (csrc/conv_nhwc.hip) against fp64 torch on the same
bf16-rounded operands: the kernels accumulate in fp32 and round once on store, so the bound is one bf16 rounding
of the result (2^-8 relative) plus the fp32 accumulation noise, not a bf16-arithmetic bound."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_conv(x_nhwc, w, scale, shift, dil, act, transpose_flip=False):
    x = x_nhwc.double().permute(0, 3, 1, 2)                     # NCHW
    wd = w.to(torch.bfloat16).double()
    KT, KF = w.shape[2], w.shape[3]
    if transpose_flip:                                           # the data gradient's weights
        wd = wd.transpose(0, 1).flip(2, 3)
    y = F.conv2d(x, wd, padding=((KT // 2) * dil, KF // 2), dilation=(dil, 1))
    y = y * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if act == "relu":
        y = y.clamp_min(0)
    elif act == "mish":
        y = y * torch.tanh(F.softplus(y, threshold=20))
    return y.permute(0, 2, 3, 1).contiguous()                   # NHWC


def _close(got, ref, what):
    got = got.double().cpu()
    tol = 2.0 ** -8 * ref.abs() + 2e-3 * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} outside the bound, worst {(got - ref).abs().max():.3e} (max |ref| {ref.abs().max():.3e})"


CASES = [  # B, T, F, KT, KF, dil
    (2, 40, 37, 5, 5, 1), (1, 23, 70, 5, 5, 2), (2, 19, 33, 5, 5, 4), (1, 40, 45, 5, 5, 8), (1, 50, 20, 5, 5, 16),
    (1, 3, 601, 5, 5, 1), (1, 1, 17, 5, 5, 1), (1, 301, 40, 5, 5, 16), (2, 40, 37, 7, 1, 1), (1, 9, 601, 7, 1, 1),
    (1, 100, 64, 5, 5, 1), (3, 17, 32, 7, 1, 1),
]


@pytest.mark.parametrize("B,T,Fq,KT,KF,dil", CASES)
@pytest.mark.parametrize("act", ["none", "mish", "relu"])
def test_nhwc_conv_matches_fp64(B, T, Fq, KT, KF, dil, act):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T * 7 + Fq + dil)
    x = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    w = torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5
    scale = torch.rand(64, generator=g) + 0.5
    shift = torch.randn(64, generator=g) * 0.3
    ref = _ref_conv(x, w, scale, shift, dil, act)
    got = ops.nhwc_conv(x.cuda(), w.cuda(), scale.cuda(), shift.cuda(), dil, act)
    _close(got, ref, f"nhwc conv {KT}x{KF} dil {dil} {act}")


@pytest.mark.parametrize("B,T,Fq,KT,KF,dil", CASES[:5] + CASES[8:9])
def test_nhwc_conv_data_gradient_weights(B, T, Fq, KT, KF, dil):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(5 + T + dil)
    dz = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    w = torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5
    one, zero = torch.ones(64), torch.zeros(64)
    # autograd's data gradient of the same conv
    xin = torch.zeros(B, 64, T, Fq, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xin, w.to(torch.bfloat16).double(), padding=((KT // 2) * dil, KF // 2), dilation=(dil, 1))
    y.backward(dz.double().permute(0, 3, 1, 2))
    ref = xin.grad.permute(0, 2, 3, 1).contiguous()
    got = ops.nhwc_conv(dz.cuda(), w.cuda(), one.cuda(), zero.cuda(), dil, "none", transpose_flip=True)
    _close(got, ref, f"nhwc data gradient {KT}x{KF} dil {dil}")


@pytest.mark.parametrize("B,T,Fq,KT,KF,dil", [(2, 40, 37, 5, 5, 1), (1, 50, 45, 5, 5, 4), (2, 30, 70, 7, 1, 1)])
def test_nhwc_conv_fused_statistics(B, T, Fq, KT, KF, dil):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(11 + T)
    x = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    w = torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5
    bias = torch.randn(64, generator=g) * 0.3
    ref = _ref_conv(x, w, torch.ones(64), bias, dil, "none")
    got, st = ops.nhwc_conv(x.cuda(), w.cuda(), torch.ones(64).cuda(), bias.cuda(), dil, "none", stats=True)
    _close(got, ref, "nhwc conv with statistics")
    s1, s2 = ref.sum((0, 1, 2)), (ref * ref).sum((0, 1, 2))
    st = st.cpu()
    assert torch.allclose(st[:, 0], s1, rtol=1e-4, atol=1e-3 * s2.sqrt().max().item())
    assert torch.allclose(st[:, 1], s2, rtol=1e-4)


def test_nhwc_conv_one_hot_indexing():
    """One input pixel / channel set to 1: the output is the (flipped) weight slice, exactly (bf16 weights)."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(3)
    w = torch.randn(64, 64, 5, 5, generator=g)
    for (t0, f0, ci, dil) in ((7, 9, 3, 1), (20, 35, 63, 4), (0, 0, 17, 2), (39, 36, 40, 8)):
        x = torch.zeros(1, 40, 37, 64)
        x[0, t0, f0, ci] = 1.0
        got = ops.nhwc_conv(x.to(torch.bfloat16).cuda(), w.cuda(), torch.ones(64).cuda(), torch.zeros(64).cuda(), dil, "none").float().cpu()
        ref = _ref_conv(x.to(torch.bfloat16), w, torch.ones(64), torch.zeros(64), dil, "none").float()
        assert torch.equal(got, ref.to(torch.bfloat16).float()), (t0, f0, ci, dil)


# ---- the channels-last path inside the stage / whole-path calls (dims.math = VS_MATH_BF16) ------------------------
@pytest.mark.parametrize("act", ["mish", "relu"])
@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("B,T,Fq", [(3, 45, 53), (2, 20, 601), (1, 301, 40)])
def test_bf16_conv_stack_stage_vs_fp64_oracle(act, training, B, T, Fq):
    """vs_conv_stack_fwd in bf16: cnn1 -> channels-last bf16 -> cnn2..cnn7 (conv_nhwc.hip) -> cnn8 -> feature layout,
    eval (BatchNorm folded into the conv epilogues) and train mode (statistics from the epilogues + apply pass)."""
    from oracle import reference_forward as R
    from voicesplit_amd import ops
    dims_d = dict(num_freq=Fq, emb_dim=24, lstm_dim=32, fc1_dim=44, fc2_dim=Fq)
    sd = R.build_state_dict(dims_d, 5)
    x, _ = R.synthetic_inputs(B, T, dims_d, 5)
    bn_out = {}
    y = R.conv_stack(x.double(), {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, act, training, bn_out=bn_out)
    ref = y.transpose(1, 2).reshape(B, T, -1)
    sd_dev = {k: v.clone().cuda() for k, v in sd.items()}
    dims = ops.make_dims(B, T, Fq, 24, 32, 44, Fq, math="bf16")
    feat = ops.conv_stack(sd_dev, x.cuda(), dims, act, training=training)
    err = ((feat.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 3e-2, f"bf16 conv stack ({act}, training={training}): {err:.3e} of the output range"
    if training:      # running statistics updated from the epilogue sums (momentum 0.1)
        for k in ("conv.2.running_mean", "conv.2.running_var", "conv.6.running_mean", "conv.6.running_var",
                  "conv.26.running_mean", "conv.26.running_var"):
            r = bn_out[k]
            e = ((sd_dev[k].double().cpu() - r).abs().max() / r.abs().max().clamp_min(1e-6)).item()
            assert e < 2e-2, (k, e)


# ---- memory-bound kernels and the backward kernels of the channels-last path ------------------------------------
def _act(y, act):
    if act == "relu":
        return y.clamp_min(0)
    if act == "mish":
        return y * torch.tanh(F.softplus(y, threshold=20))
    return y


@pytest.mark.parametrize("act", ["none", "mish", "relu"])
def test_nhwc_conv_first_bn_apply_conv_last(act):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(4)
    B, T, Fq = 2, 13, 75
    x = torch.rand(B, T, Fq, generator=g)
    w1 = torch.randn(64, 1, 1, 7, generator=g) * 0.4
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    ref = F.conv2d(x.double().unsqueeze(1), w1.double(), padding=(0, 3)) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    ref = _act(ref, act).permute(0, 2, 3, 1).contiguous()
    if act == "none":
        got, st = ops.nhwc_conv_first(x.cuda(), w1.cuda(), scale.cuda(), shift.cuda(), act, stats=True)
        assert torch.allclose(st[:, 0].cpu(), ref.sum((0, 1, 2)), rtol=1e-4, atol=1e-2)
        assert torch.allclose(st[:, 1].cpu(), (ref * ref).sum((0, 1, 2)), rtol=1e-4)
    else:
        got = ops.nhwc_conv_first(x.cuda(), w1.cuda(), scale.cuda(), shift.cuda(), act)
    _close(got, ref, "nhwc cnn1")
    # BatchNorm apply on the bf16 tensor
    z = got
    a = ops.nhwc_bn_apply(z, shift.cuda() * 0 + 1.3, scale.cuda() * 0.2, act)
    _close(a, _act(z.double().cpu() * 1.3 + (scale.double() * 0.2), act), "nhwc bn apply")
    # cnn8 + transpose/view
    w8 = torch.randn(8, 64, 1, 1, generator=g) * 0.2
    s8, h8 = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.2
    feat = ops.nhwc_conv_last(z, w8.cuda(), s8.cuda(), h8.cuda(), act)
    zr = z.double().cpu()                                                  # [B,T,F,64]
    y8 = torch.einsum("btfc,oc->btof", zr, w8.to(torch.bfloat16).double().view(8, 64)) * s8.double().view(1, 1, 8, 1) + h8.double().view(1, 1, 8, 1)
    ref8 = _act(y8, act).reshape(B, T, 8 * Fq)
    err = ((feat.double().cpu() - ref8).abs().max() / ref8.abs().max()).item()
    assert err < 1e-5, err


@pytest.mark.parametrize("B,T,Fq,KT,KF,dil", CASES)
def test_nhwc_weight_gradient(B, T, Fq, KT, KF, dil):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(17 + T + dil)
    dz = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    x = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    w = torch.zeros(64, 64, KT, KF, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w, padding=((KT // 2) * dil, KF // 2), dilation=(dil, 1))
    y.backward(dz.double().permute(0, 3, 1, 2))
    got = ops.nhwc_conv_wgrad(dz.cuda(), x.cuda(), KT, KF, dil).double().cpu()
    ref = w.grad
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, f"nhwc wgrad {KT}x{KF} dil {dil}: {err:.3e}"       # exact bf16 products, fp32 accumulation


def test_nhwc_weight_gradient_one_hot():
    """dz and x one-hot: exactly one weight-gradient entry is 1."""
    from voicesplit_amd import ops
    for (t0, f0, co, t1, f1, ci, dil) in ((7, 9, 3, 7, 9, 5, 1), (20, 35, 63, 12, 36, 0, 4), (0, 0, 17, 4, 2, 40, 2), (39, 36, 40, 23, 34, 63, 8)):
        dz = torch.zeros(1, 40, 37, 64)
        x = torch.zeros(1, 40, 37, 64)
        dz[0, t0, f0, co] = 1.0
        x[0, t1, f1, ci] = 1.0
        got = ops.nhwc_conv_wgrad(dz.to(torch.bfloat16).cuda(), x.to(torch.bfloat16).cuda(), 5, 5, dil).cpu()
        ref = torch.zeros(64, 64, 5, 5)
        dt, df = (t1 - t0) // dil + 2 if (t1 - t0) % dil == 0 else -1, f1 - f0 + 2
        if 0 <= dt < 5 and 0 <= df < 5:
            ref[co, ci, dt, df] = 1.0
        assert torch.equal(got, ref), (t0, f0, co, t1, f1, ci, dil)


@pytest.mark.parametrize("act", ["mish", "relu"])
@pytest.mark.parametrize("training", [True, False])
def test_nhwc_bn_act_backward(act, training):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(23)
    B, T, Fq = 2, 21, 45
    z = (torch.randn(B, T, Fq, 64, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    da = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    zd = z.double().reshape(-1, 64).requires_grad_(True)
    if training:
        mean, var = zd.detach().mean(0), zd.detach().var(0, unbiased=False)
    else:
        mean, var = torch.randn(64, generator=g).double() * 0.1, (torch.rand(64, generator=g).double() + 0.5)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    if training:
        m_ = zd.mean(0)
        v_ = ((zd - m_) ** 2).mean(0)
        y = (zd - m_) / torch.sqrt(v_ + 1e-5) * gd + bd
    else:
        y = (zd - mean) * invstd * gd + bd
    if act == "relu":      # keep clear of the kink: bf16 z is exact, y is what the kernel recomputes in fp32
        keep = y.detach().abs() > 1e-3
    else:
        keep = torch.ones_like(y, dtype=torch.bool)
    a = _act(y, act)
    a.backward(da.double().reshape(-1, 64) * keep)
    scale = (gamma.double() * invstd).float()
    shift = (beta.double() - mean * gamma.double() * invstd).float()
    da_k = (da.double().reshape(-1, 64) * keep).reshape(B, T, Fq, 64).to(torch.bfloat16)
    dz, dg, db, dbias = ops.nhwc_bn_act_bwd(da_k.cuda(), z.cuda(), act, training, scale.cuda(), shift.cuda(),
                                            mean.float().cuda(), invstd.float().cuda())
    ref = zd.grad.reshape(B, T, Fq, 64)
    _close(dz, ref, f"nhwc bn backward dz ({act}, training={training})")
    for got, r, nm in ((dg, gd.grad, "dgamma"), (db, bd.grad, "dbeta")):
        e = ((got.double().cpu() - r).abs().max() / r.abs().max()).item()
        assert e < 1e-4, (nm, e)


def test_nhwc_cnn1_backward_and_cnn8_backward():
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(29)
    B, T, Fq = 2, 11, 53
    x = torch.rand(B, T, Fq, generator=g)
    z = (torch.randn(B, T, Fq, 64, generator=g)).to(torch.bfloat16)
    da = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    zd = z.double().reshape(-1, 64).requires_grad_(True)
    m_ = zd.mean(0)
    v_ = ((zd - m_) ** 2).mean(0)
    invstd = 1.0 / torch.sqrt(v_.detach() + 1e-5)
    y = (zd - m_) / torch.sqrt(v_ + 1e-5) * gamma.double() + beta.double()
    _act(y, "mish").backward(da.double().reshape(-1, 64))
    dz1 = zd.grad.reshape(B, T, Fq, 64)
    xp = F.pad(x.double(), (3, 3))
    ref_dw = torch.stack([torch.einsum("btfc,btf->c", dz1, xp[:, :, k:k + Fq]) for k in range(7)], dim=1)   # [64,7]
    scale = (gamma.double() * invstd).float()
    shift = (beta.double() - m_.detach() * gamma.double() * invstd).float()
    dw, dg, db, dbias = ops.nhwc_bn_act_bwd_first(da.cuda(), z.cuda(), x.cuda(), "mish", True, scale.cuda(), shift.cuda(),
                                                  m_.detach().float().cuda(), invstd.float().cuda())
    e = ((dw.double().cpu() - ref_dw).abs().max() / ref_dw.abs().max()).item()
    assert e < 2e-4, e
    assert dbias.abs().max().item() == 0.0
    # cnn8 backward
    w8 = torch.randn(8, 64, 1, 1, generator=g) * 0.2
    dz8 = torch.randn(B, T, 8 * Fq, generator=g)
    a7 = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    din, dw8 = ops.nhwc_conv_last_bwd(dz8.cuda(), w8.cuda(), a7.cuda())
    d8 = dz8.double().reshape(B, T, 8, Fq)
    ref_din = torch.einsum("btof,oc->btfc", d8, w8.double().view(8, 64))
    ref_dw8 = torch.einsum("btof,btfc->oc", d8, a7.double())
    _close(din, ref_din, "nhwc cnn8 data gradient")
    e = ((dw8.double().cpu() - ref_dw8).abs().max() / ref_dw8.abs().max()).item()
    assert e < 2e-5, e


# ---- the dy forms: the producer of a data gradient does the first BatchNorm-backward pass ---------------------------
def _bn_consts(z, gamma, beta, g, training):
    zd = z.double().reshape(-1, 64)
    if training:
        mean, var = zd.mean(0), zd.var(0, unbiased=False)
    else:
        mean, var = torch.randn(64, generator=g).double() * 0.1, torch.rand(64, generator=g).double() + 0.5
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = (gamma.double() * invstd).float()
    shift = (beta.double() - mean * gamma.double() * invstd).float()
    return mean, invstd, scale, shift


@pytest.mark.parametrize("act", ["mish", "relu"])
@pytest.mark.parametrize("B,T,Fq,KT,KF,dil", [(2, 40, 37, 5, 5, 1), (1, 23, 70, 5, 5, 2), (1, 50, 20, 5, 5, 16), (1, 301, 40, 5, 5, 8),
                                               (2, 40, 37, 7, 1, 1), (1, 9, 601, 7, 1, 1), (1, 1, 17, 5, 5, 1), (3, 17, 32, 7, 1, 1)])
def test_nhwc_conv_dy_epilogue(B, T, Fq, KT, KF, dil, act):
    """dy = dgrad(dz) * act'(z * scale + shift) and its BatchNorm-backward sums, against fp64 on the same operands."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(41 + T + dil + KT)
    dz = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    z = (torch.randn(B, T, Fq, 64, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    w = torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    mean, invstd, scale, shift = _bn_consts(z, gamma, beta, g, True)
    da = _ref_conv(dz, w, torch.ones(64), torch.zeros(64), dil, "none", transpose_flip=True)
    y = (z.double() * scale.double() + shift.double()).requires_grad_(True)
    _act(y, act).backward(da)
    ref = y.grad
    xhat = (z.double() - mean) * invstd
    packed = ops.nhwc_conv_pack(w.cuda(), transpose_flip=True)
    dy, st = ops.nhwc_conv_dy(dz.cuda(), packed, z.cuda(), act, scale.cuda(), shift.cuda(), mean.float().cuda(), invstd.float().cuda(),
                              KT, KF, dil)
    _close(dy, ref, f"dy epilogue {KT}x{KF} dil {dil} {act}")
    st = st.sum(0).cpu()
    s1, s2 = ref.sum((0, 1, 2)), (ref * xhat).sum((0, 1, 2))
    n = (ref * ref).sum((0, 1, 2)).sqrt()          # the scale of a sum of that many terms
    assert ((st[:, 0] - s1).abs() <= 1e-3 * n + 1e-4 * s1.abs()).all(), (st[:, 0] - s1).abs().max()
    assert ((st[:, 1] - s2).abs() <= 2e-3 * n + 1e-4 * s2.abs()).all(), (st[:, 1] - s2).abs().max()


@pytest.mark.parametrize("act", ["mish", "relu"])
@pytest.mark.parametrize("training", [True, False])
def test_nhwc_cnn8_backward_dy_then_bn_backward(act, training):
    """cnn8 backward (dy form) + the second BatchNorm pass = autograd through a7 = act(BN(z7)), out = cnn8(a7)."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(31)
    B, T, Fq = 2, 13, 53
    z = (torch.randn(B, T, Fq, 64, generator=g) * 1.2 + 0.2).to(torch.bfloat16)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    w8 = torch.randn(8, 64, 1, 1, generator=g) * 0.2
    dz8 = torch.randn(B, T, 8 * Fq, generator=g)
    mean, invstd, scale, shift = _bn_consts(z, gamma, beta, g, training)
    zd = z.double().reshape(-1, 64).requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    if training:
        m_ = zd.mean(0)
        y = (zd - m_) / torch.sqrt(((zd - m_) ** 2).mean(0) + 1e-5) * gd + bd
    else:
        y = (zd - mean) * invstd * gd + bd
    a7 = _act(y, act)
    a7_bf = a7.detach().reshape(B, T, Fq, 64).to(torch.bfloat16)
    d8 = dz8.double().reshape(B, T, 8, Fq)
    out = torch.einsum("btfc,oc->btof", a7.reshape(B, T, Fq, 64), w8.double().view(8, 64))
    out.backward(d8)
    dy, dw8, st = ops.nhwc_conv_last_bwd_dy(dz8.cuda(), w8.cuda(), a7_bf.cuda(), z.cuda(), act, scale.cuda(), shift.cuda(),
                                            mean.float().cuda(), invstd.float().cuda())
    dz, dg, db, dbias = ops.nhwc_bn_bwd_from_dy(dy, z.cuda(), st, training, scale.cuda(), mean.float().cuda(), invstd.float().cuda())
    ref = zd.grad.reshape(B, T, Fq, 64)
    if act == "relu":                               # fp32 vs fp64 may disagree on the side of the kink for |y| ~ 0
        ref = ref * (y.detach().abs() > 1e-4).reshape(B, T, Fq, 64)
        dz = dz.double().cpu() * (y.detach().abs() > 1e-4).reshape(B, T, Fq, 64)
    _close(dz, ref, f"cnn8 backward dy + bn pass ({act}, training={training})")
    for got, r, nm in ((dg, gd.grad, "dgamma"), (db, bd.grad, "dbeta")):
        e = ((got.double().cpu() - r).abs().max() / r.abs().max()).item()
        assert e < 2e-3, (nm, e)                    # the sums are of bf16-rounded a7-side products: 2^-9 relative, averaged
    ref_dw8 = torch.einsum("btof,btfc->oc", d8, a7_bf.double())
    e = ((dw8.double().cpu() - ref_dw8).abs().max() / ref_dw8.abs().max()).item()
    assert e < 2e-5, e


def test_nhwc_cnn1_backward_from_dy():
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(37)
    B, T, Fq = 2, 11, 53
    x = torch.rand(B, T, Fq, generator=g)
    z = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    dy = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    mean, invstd, scale, shift = _bn_consts(z, gamma, beta, g, True)
    zd = z.double().reshape(-1, 64).requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    m_ = zd.mean(0)
    y = (zd - m_) / torch.sqrt(((zd - m_) ** 2).mean(0) + 1e-5) * gd + bd
    y.backward(dy.double().reshape(-1, 64))
    dz1 = zd.grad.reshape(B, T, Fq, 64)
    xp = F.pad(x.double(), (3, 3))
    ref_dw = torch.stack([torch.einsum("btfc,btf->c", dz1, xp[:, :, k:k + Fq]) for k in range(7)], dim=1)
    xhat = (z.double() - mean) * invstd
    st = torch.zeros(64, 64, 2, dtype=torch.float64)
    st[5, :, 0] = dy.double().sum((0, 1, 2))        # any slot: the finalize folds them
    st[9, :, 1] = (dy.double() * xhat).sum((0, 1, 2))
    dw, dg, db, dbias = ops.nhwc_bn_bwd_first_from_dy(dy.cuda(), z.cuda(), x.cuda(), st.cuda(), True, scale.cuda(), mean.float().cuda(),
                                                      invstd.float().cuda())
    e = ((dw.double().cpu() - ref_dw).abs().max() / ref_dw.abs().max()).item()
    assert e < 2e-4, e
    for got, r, nm in ((dg, gd.grad, "dgamma"), (db, bd.grad, "dbeta")):
        e = ((got.double().cpu() - r).abs().max() / r.abs().max()).item()
        assert e < 1e-4, (nm, e)
    assert dbias.abs().max().item() == 0.0


# ---- bf16 GEMM of the LSTM contractions (csrc/gemm_bf16.hip) --------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(200, 300, 130), (128, 256, 64), (1, 17, 8), (777, 520, 1000), (3010, 3200, 424)])
def test_gemm_bf16_all_operand_forms(M, N, K):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    Kp = (K + 63) // 64 * 64
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    rb = torch.randn(5, N, generator=g)
    group = -(-M // 5)
    Ab = ops.cvt_rows_bf16(A.cuda(), K, Kp)                  # [M, Kp], zero padded
    Bb = ops.cvt_rows_bf16(B.cuda(), K, Kp)
    assert torch.equal(Ab[:, :K].float().cpu(), A.to(torch.bfloat16).float()) and Ab[:, K:].abs().max().item() == 0 if Kp > K else True
    Ad, Bd = A.to(torch.bfloat16).double(), B.to(torch.bfloat16).double()
    ref = Ad @ Bd.t()
    # row x row, with the per-row-group bias of the LSTM input projection
    got = ops.gemm_bf16(Ab, Bb, M, N, K, rowbias=rb.cuda(), group=group).double().cpu()
    want = ref + rb.double()[torch.arange(M) // group]
    assert ((got - want).abs().max() / want.abs().max()).item() < 2e-5
    # row x col: B given as [K, ldb] (K-major)
    Np = (N + 7) // 8 * 8
    Bk = torch.zeros(K, Np, dtype=torch.bfloat16)
    Bk[:, :N] = B.to(torch.bfloat16).t()
    got = ops.gemm_bf16(Ab, Bk.cuda(), M, N, K, b_kmajor=True).double().cpu()
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 2e-5
    # col x col, accumulating on top of an existing C
    Mp = (M + 7) // 8 * 8
    Ak = torch.zeros(K, Mp, dtype=torch.bfloat16)
    Ak[:, :M] = A.to(torch.bfloat16).t()
    C0 = torch.randn(M, N, generator=g)
    got = ops.gemm_bf16(Ak.cuda(), Bk.cuda(), M, N, K, a_kmajor=True, b_kmajor=True, out=C0.clone().cuda(), accumulate=True).double().cpu()
    assert ((got - (ref + C0.double())).abs().max() / ref.abs().max()).item() < 2e-5


# ---- cnn7's BatchNorm + activation applied by its consumer cnn8 (train mode: no apply pass over z7, no a7 tensor) --------------
@pytest.mark.parametrize("act", ["mish", "relu"])
def test_cnn8_applies_the_batchnorm_of_cnn7_itself(act):
    """vs_nhwc_conv_last_pre(z7) == vs_nhwc_conv_last(vs_nhwc_bn_apply(z7)) bit for bit (the same formulas and the same bf16
    rounding of a7, in registers instead of through memory), its statistics are those of what it wrote, and the backward
    with a7 = NULL (recomputed from z7) == the backward that reads the stored a7, bit for bit."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(77)
    B, T, Fq = 3, 9, 75                       # 75 = 4 * 16 + 11: a partly filled last pixel block
    z7 = (torch.randn(B, T, Fq, 64, generator=g) * 1.3 + 0.1).to(torch.bfloat16).cuda()
    psc, psh = (torch.rand(64, generator=g) + 0.5).cuda(), (torch.randn(64, generator=g) * 0.3).cuda()
    w8 = (torch.randn(8, 64, 1, 1, generator=g) * 0.2).cuda()
    ones, bias = torch.ones(8).cuda(), (torch.randn(8, generator=g) * 0.2).cuda()
    a7 = ops.nhwc_bn_apply(z7, psc, psh, act)
    ref = ops.nhwc_conv_last(a7, w8, ones, bias, "none")
    got, st = ops.nhwc_conv_last_pre(z7, psc, psh, act, w8, ones, bias, stats=True)
    assert torch.equal(got, ref)
    assert torch.equal(ops.nhwc_conv_last_pre(z7, psc, psh, act, w8, ones, bias), ref)
    r = ref.double().reshape(B, T, 8, Fq)
    assert torch.allclose(st[:, 0], r.sum((0, 1, 3)), rtol=1e-5, atol=1e-4)
    assert torch.allclose(st[:, 1], (r * r).sum((0, 1, 3)), rtol=1e-5)
    # backward
    dz8 = torch.randn(B, T, 8 * Fq, generator=g).cuda()
    mean, invstd = (torch.randn(64, generator=g) * 0.1).cuda(), (torch.rand(64, generator=g) + 0.5).cuda()
    dy0, dw0, st0 = ops.nhwc_conv_last_bwd_dy(dz8, w8, a7, z7, act, psc, psh, mean, invstd)
    dy1, dw1, st1 = ops.nhwc_conv_last_bwd_dy(dz8, w8, None, z7, act, psc, psh, mean, invstd)
    assert torch.equal(dy0, dy1) and torch.equal(dw0, dw1) and torch.equal(st0.sum(0), st1.sum(0))


# ---- cnn1 by recomputation (round 4): statistics from the input's moments, one-pass forward, one-pass backward -----------------
def _shifts(x):
    """x [B,T,F] -> [B,T,F,7]: x[..., f + k - 3], zero outside the row (ZeroPad2d((3, 3, 0, 0)))."""
    xp = F.pad(x, (3, 3))
    return torch.stack([xp[..., k:k + x.shape[-1]] for k in range(7)], -1)


@pytest.mark.parametrize("B,T,Fq", [(2, 13, 75), (1, 5, 4), (3, 40, 601)])
def test_cnn1_statistics_from_input_moments(B, T, Fq):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(B + T + Fq)
    x = torch.rand(B, T, Fq, generator=g)
    w = torch.randn(64, 1, 1, 7, generator=g) * 0.4
    bias = torch.randn(64, generator=g) * 0.3
    mom = ops.nhwc_first_moments(x.cuda())
    xs = _shifts(x.double())                                                    # [B,T,F,7]
    S = xs.sum((0, 1, 2))
    R = torch.einsum("btfk,btfl->kl", xs, xs)
    ref = torch.cat([S] + [R[k, k:] for k in range(7)])
    assert mom.shape == (35,)
    assert torch.allclose(mom.cpu(), ref, rtol=2e-6, atol=1e-9 * ref.abs().max().item())
    st = ops.nhwc_first_stats(mom, w.cuda(), bias.cuda(), B * T * Fq).cpu()
    z = F.conv2d(x.double()[:, None], w.double(), bias.double(), padding=(0, 3))
    assert torch.allclose(st[0, :, 0], z.sum((0, 2, 3)), rtol=1e-5, atol=1e-6 * z.abs().sum((0, 2, 3)).max().item())
    assert torch.allclose(st[0, :, 1], (z * z).sum((0, 2, 3)), rtol=1e-5)
    # ... and through vs_bn_finalize (one slot) they give the BatchNorm constants of train mode
    n = B * T * Fq
    gamma, beta = (torch.rand(64, generator=g) + 0.5), torch.randn(64, generator=g) * 0.2
    scale, shift, mean, invstd = ops.bn_finalize(st.cuda(), n, gamma.cuda(), beta.cuda())
    m, v = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
    assert torch.allclose(mean.double().cpu(), m, rtol=1e-5, atol=1e-6)
    assert torch.allclose(invstd.double().cpu(), 1 / torch.sqrt(v + 1e-5), rtol=2e-5)
    # the one-pass forward: a1 = act(BN(z1)) with the BatchNorm folded into the kernel's arguments
    for act in ("mish", "relu"):
        a1 = ops.nhwc_conv_first(x.cuda(), w.cuda(), scale, shift + bias.cuda() * scale, act).double().cpu()
        ref_a = _act((z - m.view(1, -1, 1, 1)) / torch.sqrt(v.view(1, -1, 1, 1) + 1e-5) * gamma.double().view(1, -1, 1, 1)
                     + beta.double().view(1, -1, 1, 1), act).permute(0, 2, 3, 1)
        err = (a1 - ref_a).abs()
        assert (err <= 2.0 ** -8 * ref_a.abs() + 1e-4).all(), (act, err.max().item())


@pytest.mark.parametrize("act", ["mish", "relu"])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,T,Fq", [(2, 13, 75), (3, 40, 301)])
def test_cnn1_one_pass_backward(act, training, B, T, Fq):
    """vs_nhwc_first_bwd: d/d{conv.1.weight, conv.1.bias, conv.2.weight, conv.2.bias} of sum(da * act(BN(conv(x) + bias))) for a
    bf16 upstream gradient da -- against fp64 autograd through exactly that graph (batch statistics or frozen ones)."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(17 + T)
    x = torch.rand(B, T, Fq, generator=g)
    w = (torch.randn(64, 1, 1, 7, generator=g) * 0.4)
    bias = torch.randn(64, generator=g) * 0.3
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    rmean, rvar = torch.randn(64, generator=g) * 0.2 + 0.8, torch.rand(64, generator=g) * 0.2 + 0.1
    da = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16)
    wd, bd, gd, btd = (t.double().clone().requires_grad_(True) for t in (w, bias, gamma, beta))
    z = F.conv2d(x.double()[:, None], wd, bd, padding=(0, 3))
    if training:
        m, v = z.mean((0, 2, 3), keepdim=True), z.var((0, 2, 3), unbiased=False, keepdim=True)
    else:
        m, v = rmean.double().view(1, -1, 1, 1), rvar.double().view(1, -1, 1, 1)
    y = (z - m) / torch.sqrt(v + 1e-5) * gd.view(1, -1, 1, 1) + btd.view(1, -1, 1, 1)
    if act == "relu":                                   # keep clear of the kink (the kernel recomputes y in fp32)
        da = (da.float() * (y.detach().abs() > 1e-4).permute(0, 2, 3, 1).float()).to(torch.bfloat16)
    (_act(y, act) * da.double().permute(0, 3, 1, 2)).sum().backward()
    mean = m.detach().reshape(-1).float()
    invstd = (1 / torch.sqrt(v.detach() + 1e-5)).reshape(-1).float()
    scale = (gamma.double() * invstd.double()).float()
    shift = (beta.double() - mean.double() * scale.double()).float()
    dw, dg, db, dbias = ops.nhwc_first_bwd(da.cuda(), x.cuda(), w.cuda(), bias.cuda(), act, training, scale.cuda(), shift.cuda(),
                                           mean.cuda(), invstd.cuda())
    for got, ref, nm in ((dw, wd.grad.reshape(64, 7), "dw"), (dg, gd.grad, "dgamma"), (db, btd.grad, "dbeta")):
        e = ((got.double().cpu() - ref).abs().max() / ref.abs().max()).item()
        assert e < 2e-4, (nm, e)
    if training:
        assert dbias.abs().max().item() == 0.0 and bd.grad.abs().max().item() < 1e-6 * db.abs().max().item() + 1e-9
    else:
        e = ((dbias.double().cpu() - bd.grad).abs().max() / bd.grad.abs().max()).item()
        assert e < 2e-4, ("dbias", e)
