"""GPU: the 64 -> 64 convs of models/voicesplit/model.py:21-48 (eval-mode BatchNorm folded, activation fused) in the
fp32-class split-f16 arithmetic on channels-last hi / lo planes (csrc/conv_nhwc_f16x3.hip), against fp64.

Two comparisons per case: against fp64 on the operands the kernel was GIVEN (x reconstructed from its planes: what is left is the
weights' split, the dropped lo x lo product, fp32 accumulation and the output split -- 1e-6 of the tensor's range), and against
fp64 on the original fp32 tensor (adds the input split: the path's contract, REL_TOL = 1e-4 with two orders of margin)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _act(y, act):
    if act == "relu":
        return y.clamp_min(0)
    if act == "mish":
        return y * torch.tanh(F.softplus(y, threshold=20))
    return y


def _merge(hi, lo, s2):
    return (hi.double() + lo.double()) * s2[1].double()


@pytest.mark.parametrize("act", ["mish", "relu", "none"])
@pytest.mark.parametrize("B,T,Fq,KT,KF,dil", [
    (2, 37, 53, 5, 5, 1), (1, 61, 40, 5, 5, 2), (3, 30, 17, 5, 5, 16), (2, 33, 70, 7, 1, 1), (1, 5, 16, 5, 5, 1), (1, 2, 3, 7, 1, 1),
    (1, 1, 1, 5, 5, 4), (5, 7, 15, 5, 5, 8), (1, 19, 33, 7, 1, 1), (2, 100, 16, 5, 5, 4),
])
def test_layer_matches_fp64(act, B, T, Fq, KT, KF, dil):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(T * 10 + dil)
    x = torch.randn(B, T, Fq, 64, generator=g) * torch.rand(1, 1, 1, 64, generator=g) * 3.0
    w = torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5
    sc = (torch.rand(64, generator=g) + 0.5) * torch.where(torch.rand(64, generator=g) < 0.2, -1.0, 1.0)
    sh = torch.randn(64, generator=g) * 0.5
    hi, lo, s2 = ops.f16x3_split(x.cuda(), 2.0 ** 7)
    oh, ol, os2, amax, _ = ops.nhwc_conv_f16x3(hi, lo, s2, w.cuda(), sc.cuda(), sh.cuda(), dil, act)
    got = _merge(oh, ol, os2).cpu()

    def ref_of(xin):
        z = F.conv2d(xin.permute(0, 3, 1, 2), w.double(), None, padding=((KT // 2) * dil, KF // 2), dilation=(dil, 1))
        return _act(z * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), act).permute(0, 2, 3, 1)

    given = ref_of(_merge(hi, lo, s2).cpu())
    scale = given.abs().max()
    assert torch.isfinite(got).all()
    assert ((got - given).abs().max() / scale).item() < 2e-6
    assert ((got - ref_of(x.double())).abs().max() / scale).item() < 3e-6
    # the output scale is a power of two that keeps the planes inside f16; the tracked |max| covers the tensor and stays under the
    # plan's bound (it also sees the outputs the kernel computes and drops: columns / rows of the last tile beyond the image,
    # whose taps still reach real pixels -- values the same bound holds for)
    s = os2[0].item()
    assert s == 2.0 ** round(torch.log2(torch.tensor(s)).item()) and (oh.float().abs().max() < 32768.0)
    tracked = amax.view(torch.float32).max().item()
    assert tracked >= scale.item() * (1 - 1e-6)
    xmax = _merge(hi, lo, s2).abs().max().item()
    bound = (sc.double().abs() * w.double().abs().sum((1, 2, 3)) * xmax + sh.double().abs()).max().item()
    assert tracked <= max(bound, 0.3125) * (1 + 1e-5) and tracked * s < 32768.0


def test_layers_chain_on_their_own_scales_and_reuse_packed_weights():
    """Three layers back to back: each consumes the planes, scale pair and tracked |max| the previous one produced; the second
    call with packed_ready reuses the scratch."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(4)
    B, T, Fq = 2, 40, 37
    x = torch.randn(B, T, Fq, 64, generator=g)
    ws = [torch.randn(64, 64, kt, kf, generator=g) / (64 * kt * kf) ** 0.5 * 1.7 for kt, kf in ((7, 1), (5, 5), (5, 5))]
    dils = [1, 1, 2]
    sc = [torch.rand(64, generator=g) + 0.5 for _ in ws]
    sh = [torch.randn(64, generator=g) * 0.2 for _ in ws]
    hi, lo, s2 = ops.f16x3_split(x.cuda(), 2.0 ** 9)
    amax = None
    ref = x.double()
    scratches = []
    for w, d, a, b in zip(ws, dils, sc, sh):
        hi, lo, s2, amax, scr = ops.nhwc_conv_f16x3(hi, lo, s2, w.cuda(), a.cuda(), b.cuda(), d, "mish", amax_in=amax)
        scratches.append(scr)
        kt, kf = w.shape[2], w.shape[3]
        z = F.conv2d(ref.permute(0, 3, 1, 2), w.double(), None, padding=((kt // 2) * d, kf // 2), dilation=(d, 1))
        ref = _act(z * a.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1), "mish").permute(0, 2, 3, 1)
        got = _merge(hi, lo, s2).cpu()
        assert ((got - ref).abs().max() / ref.abs().max()).item() < 1e-5
    # packed_ready: same result, bit for bit
    h0, l0, s0 = ops.f16x3_split(x.cuda(), 2.0 ** 9)
    a1 = ops.nhwc_conv_f16x3(h0, l0, s0, ws[0].cuda(), sc[0].cuda(), sh[0].cuda(), 1, "mish")
    a2 = ops.nhwc_conv_f16x3(h0, l0, s0, ws[0].cuda(), sc[0].cuda(), sh[0].cuda(), 1, "mish", scratch=scratches[0])
    assert torch.equal(a1[0], a2[0]) and torch.equal(a1[1], a2[1])


def test_full_width_timing_shapes_run():
    """The metric configuration's layer shape at a reduced batch (B = 4): finite, deterministic, batch-independent."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 301, 601, 64, generator=g)
    w = torch.randn(64, 64, 5, 5, generator=g) / 40.0
    sc, sh = torch.ones(64), torch.zeros(64)
    hi, lo, s2 = ops.f16x3_split(x.cuda(), 2.0 ** 8)
    a = ops.nhwc_conv_f16x3(hi, lo, s2, w.cuda(), sc.cuda(), sh.cuda(), 4, "mish")
    b = ops.nhwc_conv_f16x3(hi, lo, s2, w.cuda(), sc.cuda(), sh.cuda(), 4, "mish")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    one = ops.nhwc_conv_f16x3(hi[2:3].contiguous(), lo[2:3].contiguous(), s2, w.cuda(), sc.cuda(), sh.cuda(), 4, "mish",
                              amax_in=((hi.float() + lo.float()) * s2[1]).abs().max().reshape(1).view(torch.int32))
    assert torch.equal(one[0][0], a[0][2]) and torch.equal(one[1][0], a[1][2])


@pytest.mark.parametrize("amp", [1e-6, 1.0, 3e4])
def test_scale_plan_follows_the_input_range(amp):
    """The output scale comes from a bound built on the tracked max |x|: inputs six decades below or four above unit range (and
    BatchNorm scales of both signs, shifts that dominate the conv term) keep the planes inside f16 and the result at the same
    relative accuracy; an all-zero input gives act(shift) everywhere."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(3)
    B, T, Fq, KT, KF, dil = 2, 25, 21, 5, 5, 2
    x = torch.randn(B, T, Fq, 64, generator=g) * amp
    w = torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5
    sc = (torch.rand(64, generator=g) + 0.5) * torch.where(torch.rand(64, generator=g) < 0.5, -1.0, 1.0) / amp
    sh = torch.randn(64, generator=g) * 2.0
    s_in = 2.0 ** (9 - int(torch.log2(x.abs().max()).ceil().item()))
    hi, lo, s2 = ops.f16x3_split(x.cuda(), s_in)
    oh, ol, os2, amax, _ = ops.nhwc_conv_f16x3(hi, lo, s2, w.cuda(), sc.cuda(), sh.cuda(), dil, "mish")
    got = _merge(oh, ol, os2).cpu()
    z = F.conv2d(_merge(hi, lo, s2).cpu().permute(0, 3, 1, 2), w.double(), None, padding=((KT // 2) * dil, KF // 2), dilation=(dil, 1))
    ref = _act(z * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), "mish").permute(0, 2, 3, 1)
    assert torch.isfinite(got).all() and oh.float().abs().max() < 32768.0
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 3e-6
    # zero input: the plan's floor keeps the scale finite, every pixel is act(shift)
    zh, zl = torch.zeros_like(hi), torch.zeros_like(lo)
    oh, ol, os2, _, _ = ops.nhwc_conv_f16x3(zh, zl, s2, w.cuda(), sc.cuda(), sh.cuda(), dil, "mish")
    got0 = _merge(oh, ol, os2).cpu()
    want0 = _act(sh.double(), "mish").view(1, 1, 1, 64).expand_as(got0)
    assert (got0 - want0).abs().max().item() < 1e-5


# ---- full width: the metric configuration's layer shapes, against fp64 on a sample of pixels (parity item 1) ----
def _sample_pixels(T, Fq):
    """Every row for a few columns (the image's left edge, a strip boundary, the middle, the last -- ragged -- strip of F = 601:
    columns 592..600) and every column for a few rows (the first / last rows, rows around the first segment boundaries): the strip
    / segment decode of the launch (nseg, seg_rows, the 38th strip) takes its full-size values only here."""
    cols = sorted(({0, 1, 2, 15, 16, 17, 31, 32, 47, 48, 63, 64, Fq // 2, Fq // 2 + 1} & set(range(Fq))) | set(range(max(0, Fq - 11), Fq)))
    rows = sorted({0, 1, 2, 3, 4, 5, 6, 7, 17, 18, 23, 24, 25, 47, 48, 95, 96, 150, 200, 201, T - 5, T - 4, T - 3, T - 2, T - 1} & set(range(T)))
    tt = torch.cat([torch.arange(T).repeat_interleave(len(cols)), torch.tensor(rows).repeat_interleave(Fq)])
    ff = torch.cat([torch.tensor(cols).repeat(T), torch.arange(Fq).repeat(len(rows))])
    return tt, ff


def sampled_conv_fp64(x, w, dil, tt, ff):
    """fp64 conv output [B, n, 64] at the pixels (tt, ff) of a channels-last input x [B, T, F, 64] (any float dtype, on the
    device), zero padding, time dilation `dil`: the patches are gathered from a padded fp64 copy, one einsum."""
    B, T, Fq, C = x.shape
    KT, KF = w.shape[2], w.shape[3]
    P, PF = (KT // 2) * dil, KF // 2
    xp = torch.zeros(B, T + 2 * P, Fq + 2 * PF, C, dtype=torch.float64, device=x.device)
    xp[:, P:P + T, PF:PF + Fq] = x.double()
    tt, ff = tt.to(x.device), ff.to(x.device)
    out = torch.zeros(B, tt.numel(), w.shape[0], dtype=torch.float64, device=x.device)
    w64 = w.double().to(x.device)
    for dt in range(KT):
        for df in range(KF):
            patch = xp[:, tt + dt * dil, ff + df]                     # [B, n, ci]
            out += torch.einsum("bnc,oc->bno", patch, w64[:, :, dt, df])
    return out


@pytest.mark.parametrize("KT,KF,dil", [(5, 5, 1), (5, 5, 2), (5, 5, 4), (5, 5, 8), (5, 5, 16), (7, 1, 1)])
def test_full_width_layer_matches_fp64_on_a_pixel_sample(KT, KF, dil):
    """B = 2 x 301 x 601 (the metric configuration's image), every dilation of the stack + the 7x1 layer: ~29 000 pixels x 64
    output channels (both channel halves = both workgroups of a pair) per utterance against fp64, same bounds as the small cases."""
    from voicesplit_amd import ops
    B, T, Fq = 2, 301, 601
    g = torch.Generator().manual_seed(100 * KT + dil)
    x = torch.randn(B, T, Fq, 64, generator=g) * (torch.rand(1, 1, 1, 64, generator=g) * 2.0 + 0.1)
    w = torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5
    sc = (torch.rand(64, generator=g) + 0.5) * torch.where(torch.rand(64, generator=g) < 0.2, -1.0, 1.0)
    sh = torch.randn(64, generator=g) * 0.5
    hi, lo, s2 = ops.f16x3_split(x.cuda(), 2.0 ** 6)
    oh, ol, os2, amax, _ = ops.nhwc_conv_f16x3(hi, lo, s2, w.cuda(), sc.cuda(), sh.cuda(), dil, "mish")
    got_full = _merge(oh, ol, os2)
    assert torch.isfinite(got_full).all()
    tt, ff = _sample_pixels(T, Fq)
    assert tt.numel() >= 20000
    got = got_full[:, tt.cuda(), ff.cuda()]

    def ref_of(xin):
        z = sampled_conv_fp64(xin, w, dil, tt, ff)
        return _act(z * sc.double().cuda() + sh.double().cuda(), "mish")

    given = ref_of(_merge(hi, lo, s2))
    scale = given.abs().max()
    assert ((got - given).abs().max() / scale).item() < 2e-6
    assert ((got - ref_of(x.cuda())).abs().max() / scale).item() < 3e-6
    assert amax.view(torch.float32).max().item() >= scale.item() * (1 - 1e-6)
