"""GPU: the boundary a C host stands on when it runs the bf16 path PIECEWISE, and the eval-mode
entry points at the small shapes validation() / serving use (utils/generic_utils.py:476-558: one clip at a time).

* one train-mode layer of models/voicesplit/model.py:26-28 (Conv2d -> BatchNorm2d in model.train() -> Mish) as three C-ABI
  calls -- vs_nhwc_conv (statistics from the epilogue) -> vs_bn_finalize -> vs_nhwc_bn_apply -- against fp64 torch,
  including the running-statistic update of nn.BatchNorm2d (momentum 0.1, unbiased variance);
* model.eval() in VS_MATH_BF16 on clips so short that the activation scratch cannot hold a bf16 copy of W_ih (the prepared
  blob holds it: ADVICE round 3, medium) -- mask against the fp64 oracle.
* the production shapes of the bf16 LSTM contractions (the three operand forms at B = 8 of the metric configuration and
  the two-matrix store of dW_ih), tight against fp64 on the same bf16-rounded operands."""
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


@pytest.mark.parametrize("act", ["mish", "relu"])
@pytest.mark.parametrize("B,T,Fq,KT,KF,dil", [(3, 40, 37, 5, 5, 2), (2, 30, 70, 7, 1, 1)])
def test_train_mode_layer_piecewise_through_the_c_abi(act, B, T, Fq, KT, KF, dil):
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(100 + T + dil)
    x = (torch.randn(B, T, Fq, 64, generator=g) * 0.8).to(torch.bfloat16)
    w = torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5
    bias = torch.randn(64, generator=g) * 0.3
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    rmean, rvar = torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5
    # --- the three calls a C host makes
    z, slots = ops.nhwc_conv(x.cuda(), w.cuda(), torch.ones(64).cuda(), bias.cuda(), dil, "none", stats="raw")
    assert slots.shape == (64, 64, 2)
    rm_dev, rv_dev = rmean.clone().cuda(), rvar.clone().cuda()
    n = B * T * Fq
    scale, shift, mean, invstd = ops.bn_finalize(slots, n, gamma.cuda(), beta.cuda(), rm_dev, rv_dev)
    a = ops.nhwc_bn_apply(z, scale, shift, act)
    # --- fp64: conv on the same bf16-rounded operands, nn.BatchNorm2d's training-mode arithmetic
    zr = F.conv2d(x.double().permute(0, 3, 1, 2), w.to(torch.bfloat16).double(), bias.double(), padding=((KT // 2) * dil, KF // 2),
                  dilation=(dil, 1))
    m, v = zr.mean((0, 2, 3)), zr.var((0, 2, 3), unbiased=False)
    assert torch.allclose(mean.double().cpu(), m, rtol=1e-5, atol=1e-5)
    assert torch.allclose(invstd.double().cpu(), 1 / torch.sqrt(v + 1e-5), rtol=1e-5)
    assert torch.allclose(scale.double().cpu(), gamma.double() / torch.sqrt(v + 1e-5), rtol=1e-5)
    assert torch.allclose(shift.double().cpu(), beta.double() - m * gamma.double() / torch.sqrt(v + 1e-5), rtol=1e-4, atol=1e-5)
    assert torch.allclose(rm_dev.double().cpu(), 0.9 * rmean.double() + 0.1 * m, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv_dev.double().cpu(), 0.9 * rvar.double() + 0.1 * zr.var((0, 2, 3), unbiased=True), rtol=1e-5)
    # the activation output: BatchNorm + activation of the z the conv STORED (bf16), rounded once more on store
    zs = z.double().cpu()
    ar = _act(zs * scale.double().cpu() + shift.double().cpu(), act)
    err = (a.double().cpu() - ar).abs()
    assert (err <= 2.0 ** -8 * ar.abs() + 1e-6).all(), err.max()
    # ... and it is the layer: against BatchNorm of the unrounded conv output, one bf16 rounding of z propagated
    full = _act((zr - m.view(1, -1, 1, 1)) / torch.sqrt(v.view(1, -1, 1, 1) + 1e-5) * gamma.double().view(1, -1, 1, 1)
                + beta.double().view(1, -1, 1, 1), act).permute(0, 2, 3, 1)
    assert ((a.double().cpu() - full).abs().max() / full.abs().max()).item() < 2e-2
    # no running buffers: constants only, nothing else touched
    _, slots2 = ops.nhwc_conv(x.cuda(), w.cuda(), torch.ones(64).cuda(), bias.cuda(), dil, "none", stats="raw")
    sc2, sh2, _, _ = ops.bn_finalize(slots2, n, gamma.cuda(), beta.cuda())
    assert torch.equal(sc2, scale) and torch.equal(sh2, shift)


@pytest.mark.parametrize("dims_d,B,T", [
    (dict(num_freq=53, emb_dim=24, lstm_dim=32, fc1_dim=44, fc2_dim=53), 1, 5),     # scratch < bf16 W_ih: the fall-through
    (dict(num_freq=53, emb_dim=24, lstm_dim=32, fc1_dim=44, fc2_dim=53), 3, 45),
    (None, 1, 20),                                                                   # config.json sizes, a 0.2 s clip
    (None, 1, 101),                                                                  # ... a 1 s clip
])
def test_bf16_eval_forward_on_short_clips(dims_d, B, T):
    """model.eval() in VS_MATH_BF16 goes through vs_prepare_weights / vs_forward_prepared: the prepared blob holds W_ih as
    bf16 bits.  Whatever the clip length, the LSTM input GEMM must either take the bf16 kernel (room for the bf16 copy of
    feat is all it needs then) or rebuild its operands from the fp32 weights -- never read the blob as split-f16 halves."""
    import voicesplit_amd as V
    from oracle import reference_forward as R
    from voicesplit_amd import ops
    d = dims_d or R.default_dims()
    sd = R.spread_logits(R.build_state_dict(d, 41), 6.0)
    x, dvec = R.synthetic_inputs(B, T, d, 41)
    m = V.VoiceSplit(V.default_config(d["num_freq"], d["emb_dim"], d["lstm_dim"], d["fc1_dim"], d["fc2_dim"]))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    prev = ops.get_conv_math()
    ops.set_conv_math("bf16")
    try:
        with torch.no_grad():
            got = m(x.cuda(), dvec.cuda())
            again = m(x.cuda(), dvec.cuda())              # second call: the cached prepared weights
    finally:
        ops.set_conv_math(prev)
    ref = R.forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x.double(), dvec.double(), act="mish")["mask"]
    assert torch.equal(got, again)
    err = (got.double().cpu() - ref).abs()
    assert torch.isfinite(got).all()
    assert err.max().item() < 3e-2 and (err ** 2).mean().item() < 1e-5, (err.max().item(), (err ** 2).mean().item())


# ---- bf16 GEMM at the shapes vs_forward_train / vs_backward launch it with (csrc/gemm_bf16.hip) -------------------------
def _rel(got, ref):
    return ((got - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize("rows,K,ld,Kp,offset", [
    (37, 64, 72, 128, 0),        # vector path (K % 8 == 0, ld % 4 == 0, aligned): eight pieces of data, eight of padding per row
    (301, 3200, 3200, 3200, 0),  # the gate gradients' shape class: no padding
    (37, 61, 72, 64, 0),         # K % 8 != 0: element-wise path, the last piece half data half padding
    (37, 64, 70, 128, 0),        # ld % 4 != 0: element-wise path
    (37, 64, 72, 128, 1),        # source not 16-byte aligned: element-wise path
])
def test_cvt_rows_bf16_is_round_to_nearest_even_with_zero_padding(rows, K, ld, Kp, offset):
    """vs_cvt_rows_bf16 (the operand conversion of every bf16 contraction; round 6 gave it a vector path): bit for bit torch's
    float -> bfloat16 conversion on the K data columns, zeros in the Kp - K padding columns, on both kernel paths."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(rows + K + ld)
    base = torch.randn(rows * ld + 8, generator=g).cuda()
    base[5] = float("inf")
    x = base[offset:offset + rows * ld].view(rows, ld)
    out = ops.cvt_rows_bf16(x, K, Kp)
    assert out.shape == (rows, Kp)
    ref = x[:, :K].to(torch.bfloat16)
    assert torch.equal(out[:, :K].view(torch.int16), ref.view(torch.int16))
    assert (out[:, K:].view(torch.int16) == 0).all()


def test_gemm_bf16_production_shapes():
    """B = 8 of the metric configuration: M = 8 * 301 = 2408 rows, N = 8H = 3200, K = 8F = 4808 (padded to 4864) --
    xg = feat @ W_ih^T (row x row, with the d-vector row bias), dfeat = dxg @ W_ih (row x col), dW_ih = dxg^T @ feat
    (col x col) -- against fp64 on the same bf16-rounded operands.  Bound: fp32 accumulation over K products."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(8)
    M, N, K = 2408, 3200, 4808
    Kp = (K + 63) // 64 * 64
    feat = torch.randn(M, K, generator=g).cuda()
    wih = (torch.randn(N, K, generator=g) * 0.02).cuda()
    dxg = torch.randn(M, N, generator=g).cuda()
    rb = torch.randn(8, N, generator=g).cuda()
    fb, wb = ops.cvt_rows_bf16(feat, K, Kp), ops.cvt_rows_bf16(wih, K, Kp)
    db = dxg.to(torch.bfloat16).contiguous()
    fd, wd, dd = fb[:, :K].double(), wb[:, :K].double(), db.double()
    xg = ops.gemm_bf16(fb, wb, M, N, K, rowbias=rb, group=301)
    ref = fd @ wd.t() + rb.double().repeat_interleave(301, 0)
    assert _rel(xg.double(), ref) < 2e-5
    dfeat = ops.gemm_bf16(db, wb, M, K, N, b_kmajor=True)                    # [M, K]: sum over n of dxg[m, n] * W_ih[n, k]
    assert _rel(dfeat.double(), dd @ wd) < 2e-5
    dwih = ops.gemm_bf16(db, fb, N, K, M, a_kmajor=True, b_kmajor=True)      # [N, K]: sum over m of dxg[m, n] * feat[m, k]
    ref_dw = dd.t() @ fd
    assert _rel(dwih.double(), ref_dw) < 2e-5
    # the store vs_backward uses: both directions' dW_ih from one contraction, rows < 4H -> C, the rest -> C2, each with the
    # leading dimension of lstm.weight_ih_l0 (8F + E: the d-vector columns are written by another kernel and stay untouched)
    ldc = K + 256
    c0, c1 = ops.gemm_bf16_split(db, fb, N, K, M, N // 2, a_kmajor=True, b_kmajor=True, ldc=ldc)
    assert _rel(c0[:, :K].double(), ref_dw[:N // 2]) < 2e-5 and _rel(c1[:, :K].double(), ref_dw[N // 2:]) < 2e-5
    assert torch.isnan(c0[:, K:]).all() and torch.isnan(c1[:, K:]).all()
    assert torch.equal(c0[:, :K], dwih[:N // 2]) and torch.equal(c1[:, :K], dwih[N // 2:])


@pytest.mark.parametrize("M,N,K,split", [(300, 200, 130, 100), (512, 256, 64, 256), (520, 77, 1000, 257), (64, 40, 8, 1)])
def test_gemm_bf16_split_store(M, N, K, split):
    """Tile rows that straddle split_m, a split inside the first tile, N / M that do not fill a tile."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(M + K)
    Kp = (K + 63) // 64 * 64
    A, Bm = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    Ab, Bb = ops.cvt_rows_bf16(A.cuda(), K, Kp), ops.cvt_rows_bf16(Bm.cuda(), K, Kp)
    ref = Ab[:, :K].double() @ Bb[:, :K].double().t()
    c0, c1 = ops.gemm_bf16_split(Ab, Bb, M, N, K, split, ldc=N + 8)
    assert c0.shape == (split, N + 8) and c1.shape == (M - split, N + 8)
    assert _rel(torch.cat((c0[:, :N], c1[:, :N])).double(), ref) < 2e-5
    assert torch.isnan(c0[:, N:]).all() and torch.isnan(c1[:, N:]).all()
    from voicesplit_amd import _lib
    with pytest.raises(_lib.VoiceSplitHipError, match="split_m"):
        ops.gemm_bf16_split(Ab, Bb, M, N, K, M)


@pytest.mark.parametrize("M,N,K", [(2408, 600, 601), (2408, 800, 600), (135, 44, 53), (135, 64, 44), (300, 601, 600)])
def test_gemm_bf16_gated_head_shapes(M, N, K):
    """vs_gemm_bf16_gated = the head's two data gradients of vs_backward: row-form gradient (K = 601 / 600: a K tail inside a 64-step,
    zero padded) x K-major weight [K][N padded to 8], relu mask in the epilogue; N = 601 (not a multiple of 4) takes the element-wise
    kernel.  Against fp64 on the same bf16-rounded operands."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    Kp, Np = (K + 63) // 64 * 64, (N + 7) // 8 * 8
    dy = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(K, N, generator=g) * 0.05).cuda()                     # fc weight [out = K of this contraction][in = N]
    gate = torch.randn(M, N, generator=g).cuda()
    gate[::7, ::3] = 0.0                                                    # (relu'(0) = 0)
    db, wb = ops.cvt_rows_bf16(dy, K, Kp), ops.cvt_rows_bf16(w, N, Np)
    got = ops.gemm_bf16_gated(db, wb, gate, M, N, K, b_kmajor=True)
    ref = (db[:, :K].double() @ wb[:, :N].double()) * (gate > 0).double()
    assert _rel(got.double(), ref) < 2e-5
    assert torch.equal(got == 0, ~(gate > 0) | (ref == 0))
    plain = ops.gemm_bf16(db, wb, M, N, K, b_kmajor=True)
    assert torch.equal(got, torch.where(gate > 0, plain, torch.zeros_like(plain)))


@pytest.mark.parametrize("M,N,K", [(300, 601, 600), (257, 77, 130), (520, 254, 64)])
def test_gemm_bf16_n_not_a_multiple_of_4_with_an_aligned_leading_dimension_leaves_the_padding_alone(M, N, K):
    """with N % 4 != 0 but ldc % 4 == 0 the interleaved kernel's partial-tile epilogue (16 bytes per lane,
    checked by the group's first column) wrote up to three columns past N and read rowbias / gate there.  Such shapes now take the
    element-wise epilogue: the columns >= N of C stay as they were (NaN here), in the plain, the row-bias, the split and the gated entry."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(M * 3 + N)
    Kp, ldc = (K + 63) // 64 * 64, (N + 3) // 4 * 4 + 4
    A, Bm = torch.randn(M, K, generator=g).cuda(), torch.randn(N, K, generator=g).cuda()
    Ab, Bb = ops.cvt_rows_bf16(A, K, Kp), ops.cvt_rows_bf16(Bm, K, Kp)
    ref = Ab[:, :K].double() @ Bb[:, :K].double().t()
    rb = torch.full((2, ldc), float("nan"), device="cuda")
    rb[:, :N] = torch.randn(2, N, generator=g).cuda()
    nan = lambda: torch.full((M, ldc), float("nan"), dtype=torch.float32, device="cuda")
    out = ops.gemm_bf16(Ab, Bb, M, N, K, out=nan())
    assert _rel(out[:, :N].double(), ref) < 2e-5 and torch.isnan(out[:, N:]).all()
    group = (M + 1) // 2
    out = ops.gemm_bf16(Ab, Bb, M, N, K, rowbias=rb, group=group, out=nan())
    want = ref + rb[:, :N].double().repeat_interleave(group, 0)[:M]
    assert _rel(out[:, :N].double(), want) < 2e-5 and torch.isnan(out[:, N:]).all()
    c0, c1 = ops.gemm_bf16_split(Ab, Bb, M, N, K, M // 3, ldc=ldc)
    assert _rel(torch.cat((c0[:, :N], c1[:, :N])).double(), ref) < 2e-5
    assert torch.isnan(c0[:, N:]).all() and torch.isnan(c1[:, N:]).all()
    gate = torch.full((M, ldc), float("nan"), device="cuda")
    gate[:, :N] = torch.randn(M, N, generator=g).cuda()
    out = ops.gemm_bf16_gated(Ab, Bb, gate, M, N, K, out=nan())
    assert _rel(out[:, :N].double(), ref * (gate[:, :N] > 0).double()) < 2e-5 and torch.isnan(out[:, N:]).all()
