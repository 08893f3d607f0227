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
