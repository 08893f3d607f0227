"""GPU: the backward pass (vs_backward and every backward kernel) through the C ABI against
(1) torch autograd in fp64 on the same seeded inputs (kernel level),
(2) the backward oracle in fp64 and the committed upstream gradients (module level),
(3) size-independent properties at BASELINE.json's B=64.
Tolerance: 1e-4 of each gradient tensor's max magnitude at module level (the path's fp32
contract), 3e-5 for single kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN_GRAD_CASES, load_golden_grads
from oracle import reference_backward as RB
from oracle import reference_forward as R

pytestmark = pytest.mark.gpu
KTOL = 3e-5
MTOL = 1e-4


def dev():
    return torch.device("cuda:0")


def rel_err(got, ref):
    ref = ref.detach().to(torch.float64).cpu()
    got = got.detach().to(torch.float64).cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


CONV_BWD_CASES = [
    # KT, KF, dil, B, T, F
    (7, 1, 1, 2, 19, 37),
    (5, 5, 1, 2, 19, 37),
    (5, 5, 2, 1, 23, 70),
    (5, 5, 4, 2, 21, 133),
    (5, 5, 8, 1, 50, 64),
    (5, 5, 16, 2, 40, 31),
    (5, 5, 16, 1, 20, 37),     # T shorter than the dilation halo: some taps never see the image
    (5, 5, 1, 1, 1, 5),
    (7, 1, 1, 1, 8, 601),
    (5, 5, 2, 3, 9, 601),
    (5, 5, 1, 1, 70, 37),      # ring kernel: the column is cut into 2 chunks of 35 steps (rows k0-2, k0-1 pre-loaded)
    (7, 1, 1, 2, 100, 64),     # ring kernel: 3 chunks, 3 pre-loaded rows each
]
WGRAD_MATH = ["fp32", "f16x3:ring", "f16x3:ktsplit", "f16x3:ring4"]     # every split-f16 weight-gradient kernel on every shape


class _wgrad_kernel:
    """Pin the split-f16 weight-gradient kernel (vs_set_wgrad_kernel) for the duration of a test."""

    def __init__(self, math):
        self.mode = {"ring": 1, "ktsplit": 2, "ring4": 3}.get(math.partition(":")[2], 0)

    def __enter__(self):
        from voicesplit_amd import _lib
        assert _lib.load().vs_set_wgrad_kernel(self.mode) == 0

    def __exit__(self, *exc):
        from voicesplit_amd import _lib
        _lib.load().vs_set_wgrad_kernel(0)


def _conv_ref(x, w, dil):
    KT, KF = w.shape[2], w.shape[3]
    return F.conv2d(F.pad(x, (KF // 2, KF // 2, dil * (KT // 2), dil * (KT // 2))), w, dilation=(dil, 1))


@pytest.mark.parametrize("KT,KF,dil,B,T,Fq", CONV_BWD_CASES)
@pytest.mark.parametrize("math", WGRAD_MATH)
def test_conv64_dgrad_and_wgrad(KT, KF, dil, B, T, Fq, math):
    from voicesplit_amd import ops
    pin, math = _wgrad_kernel(math), math.partition(":")[0]
    g = torch.Generator().manual_seed(KT * 1000 + dil * 10 + B)
    x = torch.randn(B, 64, T, Fq, generator=g)
    w = torch.randn(64, 64, KT, KF, generator=g) * (1.0 / (64 * KT * KF) ** 0.5)
    dz = torch.randn(B, 64, T, Fq, generator=g)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    (_conv_ref(xd, wd, dil) * dz.double()).sum().backward()
    d = dev()
    dx = ops.conv64_dgrad(dz.to(d), w.to(d), dil, math=math)
    with pin:
        dw = ops.conv64_wgrad(dz.to(d), x.to(d), KT, KF, dil, math=math)
    assert rel_err(dx, xd.grad) < KTOL
    assert rel_err(dw, wd.grad) < KTOL


@pytest.mark.parametrize("math", WGRAD_MATH)
def test_conv64_wgrad_one_hot_indices(math):
    """dz one-hot at (b,co,t,f), input one-hot at (b,ci,t',f'): exactly one weight-gradient entry,
    at (co, ci, kt, kf) with t' = t+(kt-2)*dil, f' = f+kf-2 -- catches transposed / flipped taps."""
    from voicesplit_amd import ops
    pin, math = _wgrad_kernel(math), math.partition(":")[0]
    d = dev()
    B, T, Fq, dil = 2, 14, 75, 2
    for (b, co, t, f, ci, kt, kf) in [(0, 5, 6, 10, 11, 0, 4), (1, 63, 3, 70, 0, 4, 0), (1, 33, 7, 63, 62, 2, 3),
                                      (0, 0, 13, 0, 40, 1, 2)]:
        tp, fp = t + (kt - 2) * dil, f + kf - 2
        assert 0 <= tp < T and 0 <= fp < Fq
        dz = torch.zeros(B, 64, T, Fq)
        x = torch.zeros(B, 64, T, Fq)
        dz[b, co, t, f] = 1.0
        x[b, ci, tp, fp] = 3.0
        with pin:
            dw = ops.conv64_wgrad(dz.to(d), x.to(d), 5, 5, dil, math=math).cpu()
        ref = torch.zeros(64, 64, 5, 5)
        ref[co, ci, kt, kf] = 3.0
        assert torch.equal(dw, ref), (b, co, t, f, ci, kt, kf, dw.nonzero().tolist())


@pytest.mark.parametrize("act", ["mish", "relu"])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("layout", ["nchw", "feat"])
@pytest.mark.parametrize("shape", [(3, 11, 37, 0), (2, 33, 301, 0), (2, 33, 301, 1)])
def test_bn_act_backward(act, training, layout, shape):
    """shape = (B, T, F, skew): 33*301 = 9933 elements per row crosses the 8192-element chunk of the
    row walk with every 16-byte phase; skew = 1 puts dA one float off z's phase (scalar walk)."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(7)
    B, T, Fq, skew = shape
    C = 64 if layout == "nchw" else 8
    z = torch.randn(B, C, T, Fq, generator=g) * 1.5 + 0.3
    da = torch.randn(B, C, T, Fq, generator=g)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    rmean = torch.randn(C, generator=g) * 0.1
    rvar = torch.rand(C, generator=g) + 0.5
    zd = z.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    y = F.batch_norm(zd, rmean.double().clone(), rvar.double().clone(), gd, bd, training, 0.1, 1e-5)
    (R.activation(y, act) * da.double()).sum().backward()
    if training:
        mean = z.double().mean(dim=(0, 2, 3))
        var = z.double().var(dim=(0, 2, 3), unbiased=False)
    else:
        mean, var = rmean.double(), rvar.double()
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.double() * invstd
    shift = beta.double() - mean * scale
    d = dev()
    if layout == "feat":      # [B][T][8][F]
        zz, dd = z.permute(0, 2, 1, 3).contiguous(), da.permute(0, 2, 1, 3).contiguous()
    else:
        zz, dd = z, da
    dd_dev = dd.to(d)
    if skew:
        buf = torch.empty(dd.numel() + skew, device=d)
        buf[skew:].copy_(dd_dev.reshape(-1))
        dd_dev = buf[skew:].view(dd.shape)
    dz, dgamma, dbeta, dbias = ops.bn_act_bwd(dd_dev.reshape(-1, Fq if layout == "feat" else T * Fq),
                                              zz.to(d).reshape(-1, Fq if layout == "feat" else T * Fq), C, act, training,
                                              scale.float().to(d), shift.float().to(d), mean.float().to(d), invstd.float().to(d))
    dz = dz.reshape(zz.shape)
    if layout == "feat":
        dz = dz.permute(0, 2, 1, 3)
    assert rel_err(dz, zd.grad) < KTOL
    assert rel_err(dgamma, gd.grad) < KTOL
    assert rel_err(dbeta, bd.grad) < KTOL
    ref_dbias = zd.grad.sum(dim=(0, 2, 3))
    assert (dbias.double().cpu() - ref_dbias).abs().max() < 1e-4 * zd.grad.abs().sum(dim=(0, 2, 3)).max()


@pytest.mark.parametrize("act", ["mish", "relu"])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(3, 11, 37), (2, 33, 301), (1, 2, 5)])
def test_bn_act_backward_fused_with_first_layer_wgrad(act, training, shape):
    """cnn1's BatchNorm backward with dW1 folded in (dZ1 never written): rows of 37 / 301 / 5 bins put
    frame boundaries inside the float4 packs, 33*301 crosses the 8192-element chunk."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(17)
    B, T, Fq = shape
    z = torch.randn(B, 64, T, Fq, generator=g) * 1.5 + 0.3
    da = torch.randn(B, 64, T, Fq, generator=g)
    x = torch.rand(B, T, Fq, generator=g)
    gamma = torch.rand(64, generator=g) + 0.5
    beta = torch.randn(64, generator=g) * 0.1
    rmean = torch.randn(64, generator=g) * 0.1
    rvar = torch.rand(64, generator=g) + 0.5
    zd = z.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    y = F.batch_norm(zd, rmean.double().clone(), rvar.double().clone(), gd, bd, training, 0.1, 1e-5)
    (R.activation(y, act) * da.double()).sum().backward()
    xp = F.pad(x.double(), (3, 3))
    dw_ref = torch.stack([(zd.grad * xp[:, None, :, k:k + Fq]).sum(dim=(0, 2, 3)) for k in range(7)], dim=1)
    if training:
        mean = z.double().mean(dim=(0, 2, 3))
        var = z.double().var(dim=(0, 2, 3), unbiased=False)
    else:
        mean, var = rmean.double(), rvar.double()
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.double() * invstd
    shift = beta.double() - mean * scale
    d = dev()
    dgamma, dbeta, dbias, dw = ops.bn_act_bwd_first(da.to(d), z.to(d), x.to(d), act, training, scale.float().to(d),
                                                    shift.float().to(d), mean.float().to(d), invstd.float().to(d))
    assert rel_err(dw.reshape(64, 7), dw_ref) < KTOL
    assert rel_err(dgamma, gd.grad) < KTOL
    assert rel_err(dbeta, bd.grad) < KTOL
    ref_dbias = zd.grad.sum(dim=(0, 2, 3))
    assert (dbias.double().cpu() - ref_dbias).abs().max() < 1e-4 * zd.grad.abs().sum(dim=(0, 2, 3)).max()


def test_conv_edge_layers_backward():
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(9)
    B, T, Fq = 2, 7, 83
    d = dev()
    # cnn8: 1x1, 64->8, output in [B][T][8][F]
    a7 = torch.randn(B, 64, T, Fq, generator=g)
    w8 = torch.randn(8, 64, 1, 1, generator=g) * 0.2
    dz8 = torch.randn(B, T, 8, Fq, generator=g)
    ad, wd = a7.double().requires_grad_(True), w8.double().requires_grad_(True)
    (F.conv2d(ad, wd).transpose(1, 2) * dz8.double()).sum().backward()
    assert rel_err(ops.conv_last_dgrad(dz8.to(d), w8.to(d), B, T, Fq), ad.grad) < KTOL
    assert rel_err(ops.conv_last_wgrad(dz8.to(d), a7.to(d)), wd.grad) < KTOL
    # cnn1: 1x7, 1->64
    x = torch.rand(B, T, Fq, generator=g)
    w1 = torch.randn(64, 1, 1, 7, generator=g)
    dz1 = torch.randn(B, 64, T, Fq, generator=g)
    w1d = w1.double().requires_grad_(True)
    (F.conv2d(F.pad(x.double().unsqueeze(1), (3, 3, 0, 0)), w1d) * dz1.double()).sum().backward()
    assert rel_err(ops.conv_first_wgrad(dz1.to(d), x.to(d)), w1d.grad) < KTOL


GEMM_BWD_CASES = [
    # la, lw, M, N, K, pad (extra leading-dim elements: 0 -> vector loads, 1 -> scalar path), splits
    (0, 0, 70, 90, 200, 0, 1),
    (0, 1, 301, 150, 77, 0, 1),
    (0, 1, 130, 96, 64, 1, 1),
    (1, 1, 96, 200, 301, 0, 1),
    (1, 1, 37, 53, 1000, 0, 8),
    (1, 1, 64, 40, 555, 1, 4),
    (1, 0, 100, 60, 130, 0, 1),
    (0, 0, 128, 128, 32, 0, 1),
]


@pytest.mark.parametrize("la,lw,M,N,K,pad,splits", GEMM_BWD_CASES)
@pytest.mark.parametrize("math", ["fp32", "f16x3"])
def test_gemm_layouts(la, lw, M, N, K, pad, splits, math):
    if math == "f16x3" and splits > 1:
        pytest.skip("split-K is only needed (and only offered) by the fp32 kernel")
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    ref = A.double() @ W.double().t()
    d = dev()

    def store(mat, layout):     # [rows][K] logical -> device 2-D buffer in the requested layout, ld padded
        m = mat.t().contiguous() if layout else mat
        buf = torch.zeros(m.shape[0], m.shape[1] + pad)
        buf[:, :m.shape[1]] = m
        return buf.to(d)

    got = ops.gemm(store(A, la), store(W, lw), M, N, K, layout_a=la, layout_w=lw, splits=splits, math=math)
    assert rel_err(got, ref) < KTOL


def test_gemm_epilogue_gate_accumulate_relu_and_shift():
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(3)
    d = dev()
    M, N, K = 90, 70, 120
    A, W = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    gate = torch.randn(M, N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    out = c0.clone().to(d)
    ops.gemm(A.to(d), W.to(d), M, N, K, layout_a=0, layout_w=1, gate=gate.to(d), out=out, accumulate=True)
    ref = c0.double() + (A.double() @ W.double()) * (gate > 0)
    assert rel_err(out, ref) < KTOL
    out = c0.clone().to(d)
    ops.gemm(A.to(d), W.to(d), M, N, K, layout_a=0, layout_w=1, gate=gate.to(d), out=out, accumulate=True, math="f16x3")
    assert rel_err(out, ref) < KTOL
    # relu on the K-major operand + bias + sigmoid
    bias = torch.randn(N, generator=g)
    got = ops.gemm(A.to(d), W.to(d), M, N, K, layout_a=0, layout_w=1, w_relu=True, bias=bias.to(d), act="sigmoid")
    assert rel_err(got, torch.sigmoid(A.double() @ W.double().clamp_min(0) + bias.double())) < KTOL
    # dW_hh pattern: sum_k A[k][m] * W[k + shift][n] inside groups of T rows
    T, Bn, H4, H = 13, 3, 40, 24
    Kk = T * Bn
    G = torch.randn(Kk, H4, generator=g)
    Hh = torch.randn(Kk, H, generator=g)
    for shift in (-1, 1):
        ref = torch.zeros(H4, H, dtype=torch.float64)
        for k in range(Kk):
            t = k % T
            if 0 <= t + shift < T:
                ref += torch.outer(G[k].double(), Hh[k + shift].double())
        got = ops.gemm(G.to(d), Hh.to(d), H4, H, Kk, layout_a=1, layout_w=1, w_shift=shift, w_group=T, splits=4)
        assert rel_err(got, ref) < KTOL


@pytest.mark.parametrize("B,T,H", [(2, 9, 24), (5, 17, 32), (33, 6, 40), (3, 12, 400), (2, 301, 40), (1, 301, 400)])
def test_bilstm_train_forward_and_bptt(B, T, H):
    """Saved gates / cell states of the training forward and the gate gradients of the BPTT kernel
    against autograd through the explicit recurrence (fp64)."""
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(B * 100 + T)
    xg = torch.randn(B, T, 8 * H, generator=g)
    whh = [torch.randn(4 * H, H, generator=g) * (1.5 / H ** 0.5) for _ in range(2)]
    dout = torch.randn(B, T, 2 * H, generator=g)
    xgd = xg.double().requires_grad_(True)
    outs, gates_ref, c_ref = [], [], []
    for dirn in range(2):
        h = torch.zeros(B, H, dtype=torch.float64)
        c = torch.zeros(B, H, dtype=torch.float64)
        out = [None] * T
        gs, cs = [None] * T, [None] * T
        for t in (range(T - 1, -1, -1) if dirn else range(T)):
            pre = xgd[:, t, dirn * 4 * H:(dirn + 1) * 4 * H] + h @ whh[dirn].double().t()
            i, f, gg, o = pre.split(H, dim=1)
            i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
            c = f * c + i * gg
            h = o * torch.tanh(c)
            out[t], gs[t], cs[t] = h, torch.cat((i, f, gg, o), 1), c
        outs.append(torch.stack(out, 1))
        gates_ref.append(torch.stack(gs, 1))
        c_ref.append(torch.stack(cs, 1))
    out_ref = torch.cat(outs, 2)
    (out_ref * dout.double()).sum().backward()
    d = dev()
    out, gates, c = ops.bilstm_recurrent_train(xg.to(d), whh[0].to(d), whh[1].to(d))
    assert rel_err(out, out_ref) < KTOL
    assert rel_err(gates, torch.cat(gates_ref, 2)) < KTOL
    assert rel_err(c, torch.cat(c_ref, 2)) < KTOL
    dxg = ops.bilstm_recurrent_bwd(gates, c, dout.to(d), whh[0].to(d), whh[1].to(d))
    err_t = (dxg.double().cpu() - xgd.grad).abs().amax(dim=(0, 2)) / xgd.grad.abs().max()
    worst = torch.topk(err_t, min(5, T))
    assert rel_err(dxg, xgd.grad) < KTOL, (worst.indices.tolist(), worst.values.tolist())


@pytest.mark.parametrize("B,T,H", [(3, 11, 24), (33, 5, 16), (2, 40, 400), (64, 37, 400), (100, 4, 400), (130, 6, 16), (5, 9, 424)])
def test_persistent_lstm_matches_the_step_kernels(B, T, H):
    """The persistent recurrence / BPTT (one launch, flag hand-off between workgroups) against one launch per
    time step: same decomposition, same reduction order, so every output (h, saved gates, cell states, gate
    gradients) agrees to rounding -- the two kernels are separate compilations, and fma contraction of the gate
    arithmetic may differ by an ulp, which the recurrence carries along: 2e-6 of each tensor's range, an order
    below the 3e-5 both hold against the fp64 reference.  B = 64 at H = 400 is the bench shape (2 batch tiles x 50 workgroups x 2
    directions resident at once), B = 100 needs two launches of the persistent kernel, H = 424 exceeds the
    register-resident part of W_hh."""
    from voicesplit_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 31 + T + H)
    xg = torch.randn(B, T, 8 * H, generator=g)
    whh = [torch.randn(4 * H, H, generator=g) * (1.5 / H ** 0.5) for _ in range(2)]
    dout = torch.randn(B, T, 2 * H, generator=g)
    d = dev()
    res = {}
    try:
        for mode in (1, 2):
            assert lib.vs_set_lstm_kernel(mode) == 0
            o_inf = ops.bilstm_recurrent(xg.to(d), whh[0].to(d), whh[1].to(d))
            out, gates, c = ops.bilstm_recurrent_train(xg.to(d), whh[0].to(d), whh[1].to(d))
            dxg = ops.bilstm_recurrent_bwd(gates, c, dout.to(d), whh[0].to(d), whh[1].to(d))
            res[mode] = (o_inf.cpu(), out.cpu(), gates.cpu(), c.cpu(), dxg.cpu())
    finally:
        lib.vs_set_lstm_kernel(0)
    worst = {name: rel_err(y, x) for name, x, y in zip(("out", "out_train", "gates", "c", "dxg"), res[1], res[2])}
    assert max(worst.values()) < 2e-6, worst
    assert torch.isfinite(res[2][4]).all()


def test_sigmoid_bwd_and_colsum():
    from voicesplit_amd import ops
    g = torch.Generator().manual_seed(1)
    d = dev()
    m = torch.rand(7, 13, 53, generator=g)
    dm = torch.randn(7, 13, 53, generator=g)
    assert rel_err(ops.sigmoid_bwd(dm.to(d), m.to(d)), dm.double() * m.double() * (1 - m.double())) < 1e-6
    x = torch.randn(6 * 11, 301, generator=g)
    assert rel_err(ops.colsum(x.to(d), 6, 11), x.double().reshape(6, 11, 301).sum(1)) < 1e-6


# ---------------------------------------------------------------------------------------------
# module level: model(x, emb) -> loss -> loss.backward()   (train.py:94-110)
# ---------------------------------------------------------------------------------------------

def _module(cls_name, dims_d, sd):
    import voicesplit_amd as V
    m = getattr(V, cls_name)(V.default_config(dims_d["num_freq"], dims_d["emb_dim"], dims_d["lstm_dim"],
                                              dims_d["fc1_dim"], dims_d["fc2_dim"]))
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def _our_relu_signs(tape, dims, act, B, T):
    """'was the ReLU input positive' for every ReLU of the path, read from the HIP tape."""
    from voicesplit_amd import ops
    lay = ops.tape_layout(dims)
    Fq, H = dims.F, dims.H
    pos = {"lstm_out": ops.ws_view(tape, lay.lstm_out, (B, T, 2 * H)).cpu() > 0,
           "fc1_pre": ops.ws_view(tape, lay.fc1_out, (B, T, dims.FC1)).cpu() > 0}      # relu(fc1) > 0 <=> fc1 > 0
    if act == "relu":
        for l in range(7):
            pos[f"y{l + 1}"] = ops.ws_view(tape, lay.a[l], (B, 64, T, Fq)).cpu() > 0
        pos["y8"] = ops.ws_view(tape, lay.feat, (B, T, 8, Fq)).cpu().permute(0, 2, 1, 3) > 0
    return pos


def _branch_consistent_oracle(sd, x, dvec, w, act, training, tape, dims, **kw):
    """fp64 oracle gradients with every ReLU pinned to the branch the HIP forward took for the
    (few) elements within 1e-4 of a kink -- see oracle/reference_backward.relu_gates."""
    B, T = x.shape[0], x.shape[1]
    sd64 = R.cast_state_dict(sd, torch.float64)
    with torch.no_grad():
        fw = RB.forward_with_graph(sd64, x.double(), dvec.double(), act, training, kw.get("lstm_impl", "aten"))
    ref_in = {k: fw[k] for k in RB.relu_inputs(act)}
    gates, n_over, n_tot = RB.relu_gates(ref_in, _our_relu_signs(tape, dims, act, B, T))
    assert n_over <= 5 + 2e-5 * n_tot, f"{n_over} of {n_tot} ReLU decisions differ from the oracle's"
    return RB.gradients(sd, x, dvec, w, act=act, training=training, dtype=torch.float64, gates=gates, **kw)


def _zero_bias_keys(training):
    # conv biases in front of a batch-stat BatchNorm: the true gradient is exactly 0
    return {f"conv.{i}.bias" for i in (1, 5, 9, 13, 17, 21, 25, 28)} if training else set()


@pytest.mark.parametrize("cls_name,act", [("VoiceSplit", "mish"), ("VoiceFilter", "relu")])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("B,T", [(2, 45), (5, 17)])
def test_module_backward_matches_fp64_oracle(cls_name, act, training, B, T):
    from voicesplit_amd import ops
    dims_d = dict(num_freq=53, emb_dim=24, lstm_dim=32, fc1_dim=44, fc2_dim=53)
    sd = R.spread_logits(R.build_state_dict(dims_d, 21), 6.0)
    x, dvec = R.synthetic_inputs(B, T, dims_d, 21)
    w = RB.loss_weights(B, T, 53, 21)
    m = _module(cls_name, dims_d, sd)
    m.train(training)
    emb = dvec.cuda().requires_grad_(True)
    mask = m(x.cuda(), emb)
    assert mask.requires_grad
    # keep the tape alive past backward for the stage-level comparison
    tape = mask.grad_fn.tape
    (mask * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    dims = ops.make_dims(B, T, 53, 24, 32, 44, 53)
    lay = ops.tape_layout(dims)
    stages = {}
    ref = _branch_consistent_oracle(sd, x, dvec, w, act, training, tape, dims, lstm_impl="loop", want_dvec=True,
                                    stages=stages)
    assert rel_err(mask, stages["mask"]) < MTOL
    assert rel_err(ops.ws_view(tape, lay.dlogits, (B, T, 53)), stages["logits"]) < MTOL
    assert rel_err(ops.ws_view(tape, lay.dfc1, (B, T, 44)), stages["fc1_pre"]) < MTOL
    assert rel_err(ops.ws_view(tape, lay.dlstm_out, (B, T, 64)), stages["lstm_out"]) < MTOL
    zero = _zero_bias_keys(training)
    worst = {}
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        if k in zero:
            assert p.grad.abs().max().item() == 0.0
            continue
        worst[k] = rel_err(p.grad, ref[k])
    _dump(f"oracle64_{cls_name}_{training}_{B}_{T}", worst)
    bad = {k: v for k, v in worst.items() if v >= MTOL}
    assert not bad, bad
    assert rel_err(emb.grad, ref["speaker_embedding"]) < MTOL


def _align_fragile_relus(tape, dims, act, g):
    """Put the HIP tape on the upstream's side of every near-kink ReLU input the fixture lists.

    d(relu)/dx jumps at 0: an implementation whose forward differs from the upstream's by rounding
    can sit on the other side of a kink for the few elements with |x| ~ 1e-7 and then differentiates
    a different, equally valid branch.  The fixture (oracle/make_golden.py --grads) records every
    upstream ReLU input within 5e-5 (relative) of zero; before the backward pass those elements of the
    tape are overwritten with the upstream's values (a perturbation <= 5e-5 of the tensor's range on
    a handful of elements), so the comparison is deterministic and never skipped.  Returns the
    number of elements whose branch actually differed."""
    from voicesplit_amd import ops
    lay = ops.tape_layout(dims)
    B, T, Fq, H = dims.B, dims.T, dims.F, dims.H
    dev_ = tape.device
    n_flip = 0

    def fragile(nm):
        idx = torch.from_numpy(np.asarray(g["fragile_idx/" + nm])).long().to(dev_)
        val = torch.from_numpy(np.asarray(g["fragile_val/" + nm])).float().to(dev_)
        return idx, val

    idx, val = fragile("lstm_out")          # stored raw; relu is applied by the consumers
    if idx.numel():
        v = ops.ws_view(tape, lay.lstm_out, (B * T * 2 * H,))
        n_flip += int(((v[idx] > 0) != (val > 0)).sum())
        v[idx] = val
    idx, val = fragile("fc1_pre")           # stored after the ReLU: gate = (h1 > 0)
    if idx.numel():
        v = ops.ws_view(tape, lay.fc1_out, (B * T * dims.FC1,))
        n_flip += int(((v[idx] > 0) != (val > 0)).sum())
        v[idx] = val.clamp_min(0)
    if act == "relu":                        # VoiceFilter: y_l = z_l*scale_l + shift_l feeds a ReLU after every BatchNorm
        for l in range(8):
            idx, val = fragile(f"y{l + 1}")
            if not idx.numel():
                continue
            C = 64 if l < 7 else 8
            sc = ops.ws_view(tape, lay.bn_scale + 64 * 4 * l, (C,))
            sh = ops.ws_view(tape, lay.bn_shift + 64 * 4 * l, (C,))
            c = (idx // (T * Fq)) % C                     # the oracle's y is [B][C][T][F]
            if l < 7:
                z = ops.ws_view(tape, lay.z[l], (B * 64 * T * Fq,))
                a = ops.ws_view(tape, lay.a[l], (B * 64 * T * Fq,))
                k = idx
            else:                                         # cnn8 lives in the LSTM feature layout [B][T][8][F]
                z = ops.ws_view(tape, lay.z8, (B * T * 8 * Fq,))
                a = ops.ws_view(tape, lay.feat, (B * T * 8 * Fq,))
                b_, t_, f_ = idx // (8 * T * Fq), (idx // Fq) % T, idx % Fq
                k = ((b_ * T + t_) * 8 + c) * Fq + f_
            y_now = z[k] * sc[c] + sh[c]
            n_flip += int(((y_now > 0) != (val > 0)).sum())
            # far enough from zero that the fp32 fma of the backward kernel cannot round across it
            tgt = torch.where(val > 0, 1.0, -1.0) * torch.maximum(val.abs(), 1e-5 * sh[c].abs().clamp_min(1.0))
            z[k] = ((tgt.double() - sh[c].double()) / sc[c].double()).float()
            assert bool((((z[k] * sc[c] + sh[c]) > 0) == (val > 0)).all())
            a[k] = tgt.clamp_min(0)
    return n_flip


def _golden_grad_params():
    # the metric configuration (full size, several utterances, batch-stat BatchNorm) in both arithmetics
    return [(n, "f16x3") for n in GOLDEN_GRAD_CASES] + [("vs_full_b8_train_grads", "fp32")]


@pytest.mark.parametrize("name,math", _golden_grad_params())
def test_module_backward_matches_upstream_golden_gradients(name, math):
    from voicesplit_amd import ops
    g = load_golden_grads(name)
    d = g["dims"]
    sd = R.spread_logits(R.build_state_dict(d, g["seed"]), g["gain"])
    x, dvec = R.synthetic_inputs(g["B"], g["T"], d, g["seed"])
    w = RB.loss_weights(g["B"], g["T"], d["fc2_dim"], g["seed"])
    m = _module("VoiceSplit" if g["model"] == "voicesplit" else "VoiceFilter", d, sd)
    m.train(g["training"])
    prev = ops.get_conv_math()
    ops.set_conv_math(math)
    try:
        mask = m(x.cuda(), dvec.cuda())
        assert abs(float(mask.detach().double().sum()) - float(g["mask_sum"])) < 1e-4 * abs(float(g["mask_sum"]))
        tape = mask.grad_fn.tape
        dims = ops.make_dims(g["B"], g["T"], d["num_freq"], d["emb_dim"], d["lstm_dim"], d["fc1_dim"], d["fc2_dim"])
        B, T, Fq, H = g["B"], g["T"], d["num_freq"], d["lstm_dim"]
        if "fwd/mask" in g:      # forward intermediates + BatchNorm buffers of the same training-mode call
            lay = ops.tape_layout(dims)
            feat = ops.ws_view(tape, lay.feat, (B, T, 8, Fq)).cpu().permute(0, 2, 1, 3)
            assert rel_err(feat[:, :, ::16, ::4], torch.from_numpy(g["fwd/cnn8"])) < MTOL
            assert rel_err(ops.ws_view(tape, lay.lstm_out, (B, T, 2 * H))[:, ::8], torch.from_numpy(g["fwd/lstm_out"])) < MTOL
            assert rel_err(mask[:, ::8], torch.from_numpy(g["fwd/mask"])) < MTOL
            sdc = {k: v.detach() for k, v in m.state_dict().items()}
            _, lg = ops.head(sdc, ops.ws_view(tape, lay.lstm_out, (B, T, 2 * H)).clone(), dims, want_logits=True)
            assert rel_err(lg[:, ::8], torch.from_numpy(g["fwd/logits"])) < MTOL
            after = m.state_dict()
            for k in after:
                if "running_" in k:
                    assert rel_err(after[k], torch.from_numpy(g["after/" + k])) < MTOL, k
                if "num_batches" in k:
                    assert int(after[k]) == int(g["after/" + k])
        act = "mish" if g["model"] == "voicesplit" else "relu"
        n_frag = sum(len(g["fragile_idx/" + nm]) for nm in RB.relu_inputs(act))
        n_flip = _align_fragile_relus(tape, dims, act, g)
        assert n_flip <= 5 + 0.05 * n_frag, f"{n_flip} of {n_frag} near-kink ReLU inputs on the other side"
        (mask * w.cuda()).sum().backward()
    finally:
        ops.set_conv_math(prev)
    zero = _zero_bias_keys(g["training"])
    bad, table = {}, {}
    for k, p in m.named_parameters():
        if k in zero:
            continue
        got = RB.thin_grad(p.grad.detach().cpu()).double().numpy()
        err = np.abs(got - g["grads"][k]).max() / max(g["gabs"][k], 1e-30)
        table[k] = float(err)
        if err >= MTOL:
            bad[k] = err
    _dump(f"{name}_{math}", table)
    assert not bad, bad


def test_training_step_semantics():
    """What train.py relies on around the call: the mask is an autograd tensor, BatchNorm running
    statistics and num_batches_tracked move exactly once per forward, gradients accumulate into
    .grad across two backward calls, x with requires_grad is refused loudly, a second backward
    through the same graph is refused."""
    dims_d = dict(num_freq=37, emb_dim=16, lstm_dim=24, fc1_dim=40, fc2_dim=37)
    sd = R.spread_logits(R.build_state_dict(dims_d, 2), 6.0)
    m = _module("VoiceSplit", dims_d, sd).train()
    x, dvec = R.synthetic_inputs(3, 21, dims_d, 2)
    xc, dc = x.cuda(), dvec.cuda()
    with torch.no_grad():
        ref_mask = m(xc, dc)                       # inference-path kernels, batch-stat BN
    rm1 = m.conv[2].running_mean.clone()
    mask = m(xc, dc)                               # tape path
    assert torch.allclose(mask, ref_mask, rtol=0, atol=1e-5)    # two ways of summing the batch statistics
    assert int(m.conv[2].num_batches_tracked) == 2
    assert not torch.equal(m.conv[2].running_mean, rm1)
    mask.sum().backward()
    g1 = m.fc2.weight.grad.clone()
    with pytest.raises(RuntimeError):
        mask.sum().backward()
    m(xc, dc).sum().backward()
    assert torch.allclose(m.fc2.weight.grad, 2 * g1, rtol=1e-4, atol=1e-7)
    with pytest.raises(NotImplementedError):
        m(xc.clone().requires_grad_(True), dc)
    # an optimizer step runs on the produced gradients (train.py:34,111)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    before = m.fc1.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, m.fc1.weight.detach())


def test_side_stream_weight_gradients_are_bit_identical():
    """vs_set_backward_overlap: the 64->64 weight gradients on the library's second stream (beside the BatchNorm
    backward passes of the next layer) vs everything in order on the caller's stream -- same kernels, same
    summation order, so every parameter gradient must be bit-identical; twice, to catch a missing join."""
    from voicesplit_amd import _lib
    dims_d = dict(num_freq=601, emb_dim=256, lstm_dim=32, fc1_dim=48, fc2_dim=601)
    sd = R.spread_logits(R.build_state_dict(dims_d, 5), 6.0)
    x, dvec = R.synthetic_inputs(3, 70, dims_d, 5)
    xc, dc = x.cuda(), dvec.cuda()
    w = torch.randn(3, 70, 601, generator=torch.Generator().manual_seed(9)).cuda()
    lib = _lib.load()

    def grads(overlap):
        assert lib.vs_set_backward_overlap(overlap) == 0
        m = _module("VoiceSplit", dims_d, sd).train()
        out = []
        for _ in range(2):
            m.zero_grad(set_to_none=True)
            (m(xc, dc) * w).sum().backward()
            torch.cuda.synchronize()
            out.append({k: v.grad.clone() for k, v in m.named_parameters()})
        return out

    try:
        serial = grads(0)
        overlapped = grads(1)
    finally:
        lib.vs_set_backward_overlap(1)
    assert lib.vs_set_backward_overlap(2) == -1
    for a, b in zip(serial, overlapped):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_full_batch_backward_properties():
    """BASELINE size (B=64, 301x601): finite gradients; with frozen BatchNorm utterances are
    independent, so a loss that only touches utterance 5 must give the gradients of that utterance
    run alone; scaling the upstream gradient by 2 scales every gradient by 2."""
    import voicesplit_amd as V
    dims_d = R.default_dims()
    sd = R.spread_logits(R.build_state_dict(dims_d, 0), 8.0)
    m = V.VoiceSplit(V.default_config()).eval()
    m.load_state_dict(sd)
    m = m.cuda()
    x, dvec = R.synthetic_inputs(64, 301, dims_d, 0)
    xc, dc = x.cuda(), dvec.cuda()
    w = RB.loss_weights(1, 301, 601, 5).cuda()

    def grads_of(xb, db, sel, scale):
        m.zero_grad(set_to_none=True)
        mask = m(xb, db)
        (mask[sel] * w[0] * scale).sum().backward()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters()}

    big = grads_of(xc, dc, 5, 1.0)
    one = grads_of(xc[5:6].contiguous(), dc[5:6].contiguous(), 0, 1.0)
    twice = grads_of(xc, dc, 5, 2.0)
    torch.cuda.synchronize()
    bad = {}
    for k in big:
        assert torch.isfinite(big[k]).all(), k
        sc = one[k].abs().max().clamp_min(1e-30)
        e1 = ((big[k] - one[k]).abs().max() / sc).item()
        e2 = ((twice[k] - 2 * big[k]).abs().max() / sc).item()
        if e1 >= MTOL or e2 >= 1e-5:
            bad[k] = (e1, e2)
    assert not bad, bad


def _dump(name, table):
    """Keep the per-tensor error table of the big cases (merged back from the GPU box)."""
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, f"errors_{name}.json"), "w") as f:
            json.dump(table, f, indent=1)
    except OSError:
        pass


def test_full_size_layerwise_backward_vs_fp64_oracle():
    """One full-size utterance (301x601, Mish, batch-stat BN): every conv-stack backward kernel
    fed with the fp64 oracle's own tensors (cast to fp32) and compared with the oracle's result,
    layer by layer -- long reductions (181k pixels) and the real dilation/size combinations."""
    from voicesplit_amd import ops
    dims_d = R.default_dims()
    sd = R.spread_logits(R.build_state_dict(dims_d, 0), 8.0)
    x, dvec = R.synthetic_inputs(1, 301, dims_d, 0)
    w = RB.loss_weights(1, 301, 601, 0)
    st = {}
    ref = RB.gradients(sd, x, dvec, w, act="mish", training=True, dtype=torch.float64, stages=st)
    d = dev()
    f32 = lambda t: t.float().contiguous().to(d)
    table = {}
    T, Fq = 301, 601
    for l, spec in enumerate(R.CONV_TABLE):
        if l == 0 or l == 7:
            continue
        z, da, dz = st[f"val/z{l + 1}"], st[f"cnn{l + 1}"], st[f"z{l + 1}"]
        a_in = st[f"val/cnn{l}"]
        mean = z.mean(dim=(0, 2, 3))
        invstd = 1.0 / torch.sqrt(z.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
        gamma, beta = sd[f"conv.{spec.bn_idx}.weight"].double(), sd[f"conv.{spec.bn_idx}.bias"].double()
        scale = gamma * invstd
        shift = beta - mean * scale
        gdz, dgamma, dbeta, _ = ops.bn_act_bwd(f32(da).reshape(64, T * Fq), f32(z).reshape(64, T * Fq), 64, "mish", True,
                                               f32(scale), f32(shift), f32(mean), f32(invstd))
        table[f"cnn{l + 1}.bn_dz"] = rel_err(gdz.reshape(1, 64, T, Fq), dz)
        table[f"cnn{l + 1}.dgamma"] = rel_err(dgamma, ref[f"conv.{spec.bn_idx}.weight"])
        table[f"cnn{l + 1}.dbeta"] = rel_err(dbeta, ref[f"conv.{spec.bn_idx}.bias"])
        for math in ("fp32", "f16x3"):
            dw = ops.conv64_wgrad(f32(dz), f32(a_in), spec.kt, spec.kf, spec.dil_t, math=math)
            table[f"cnn{l + 1}.wgrad.{math}"] = rel_err(dw, ref[f"conv.{spec.conv_idx}.weight"])
        for math in ("fp32", "f16x3"):
            din = ops.conv64_dgrad(f32(dz), f32(sd[f"conv.{spec.conv_idx}.weight"]), spec.dil_t, math=math)
            table[f"cnn{l + 1}.dgrad.{math}"] = rel_err(din, st[f"cnn{l}"])
        # and the forward kernel on every pixel of the full-size layer (the golden fixtures only
        # keep a strided subset of cnn8): y = (conv + bias)*scale + shift
        bias = sd[f"conv.{spec.conv_idx}.bias"].double()
        for math in ("fp32", "f16x3"):
            out = ops.conv64(f32(a_in), f32(sd[f"conv.{spec.conv_idx}.weight"]), f32(scale), f32(shift + bias * scale),
                             spec.dil_t, "mish", math=math)
            table[f"cnn{l + 1}.fwd.{math}"] = rel_err(out, st[f"val/cnn{l + 1}"])
    _dump("layerwise_full", table)
    bad = {k: v for k, v in table.items() if v >= KTOL}
    assert not bad, bad


def test_full_size_module_gradients_vs_fp64_oracle():
    """Full-size utterance (301x601, H=400) through the module: every parameter gradient and the
    gradients the tape keeps (gate pre-activations per direction and frame, lstm output, fc1,
    logits, cnn8) against the branch-consistent fp64 oracle."""
    from voicesplit_amd import ops
    dims_d = R.default_dims()
    sd = R.spread_logits(R.build_state_dict(dims_d, 0), 8.0)
    x, dvec = R.synthetic_inputs(1, 301, dims_d, 0)
    w = RB.loss_weights(1, 301, 601, 0)
    m = _module("VoiceSplit", dims_d, sd).train()
    mask = m(x.cuda(), dvec.cuda())
    tape = mask.grad_fn.tape
    (mask * w.cuda()).sum().backward()
    torch.cuda.synchronize()
    dims = ops.make_dims(1, 301, 601, 256, 400, 600, 601)
    lay = ops.tape_layout(dims)
    st = {}
    ref = _branch_consistent_oracle(sd, x, dvec, w, "mish", True, tape, dims, lstm_impl="loop", stages=st)
    table = {}
    zero = _zero_bias_keys(True)
    for k, p in m.named_parameters():
        if k not in zero:
            table["grad/" + k] = rel_err(p.grad, ref[k])
    table["mask"] = rel_err(mask, st["mask"])
    table["dlogits"] = rel_err(ops.ws_view(tape, lay.dlogits, (1, 301, 601)), st["logits"])
    table["dfc1"] = rel_err(ops.ws_view(tape, lay.dfc1, (1, 301, 600)), st["fc1_pre"])
    table["dlstm_out"] = rel_err(ops.ws_view(tape, lay.dlstm_out, (1, 301, 800)), st["lstm_out"])
    table["lstm_out_fwd_err"] = rel_err(ops.ws_view(tape, lay.lstm_out, (1, 301, 800)), st["val/lstm_out"])
    dxg = ops.ws_view(tape, lay.gates, (1, 301, 3200)).double().cpu()
    for name, sl in (("xg", slice(0, 1600)), ("xg_reverse", slice(1600, 3200))):
        ref = st[name]
        err_t = (dxg[:, :, sl] - ref).abs().amax(dim=(0, 2)) / ref.abs().max()
        worst = torch.topk(err_t, 5)
        table["d" + name] = float(err_t.max())
        table["d" + name + "_worst_t"] = worst.indices.tolist()
        table["d" + name + "_worst_err"] = [float(v) for v in worst.values]
    table["dfeat->dz8 (cnn8)"] = rel_err(ops.ws_view(tape, lay.dfeat, (1, 301, 8, 601)).permute(0, 2, 1, 3), st["z8"])
    _dump("stages_full", table)
    bad = {k: v for k, v in table.items() if isinstance(v, float) and v >= MTOL}
    assert not bad, (bad, table)
