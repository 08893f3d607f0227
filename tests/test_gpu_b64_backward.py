"""GPU: the BACKWARD of the metric batch (B = 64 x 301 x 601, bf16 configuration, batch-statistics BatchNorm: what bench.py times)
against fp64 -- VERDICT round 5, "parity first" item 1.  tests/test_gpu_b64.py holds the forward of this batch; the backward was
compared with references at B = 8 at most.  What B = 64 adds is schedule (item decode of the persistent conv / weight-gradient
kernels at 64 x dil x 19 items, the one-block-per-CU BatchNorm pass, 75.25 tile rows of the LSTM contractions, two batch tiles of
the BPTT, the side-stream forks), so every stage is checked ON ITS OWN OPERANDS, the tensors the previous stage left on the device,
and -- unlike a pixel sample of four utterances -- IN FULL: a skipped or doubly visited item of any utterance is a wrong element.

  1. One forward + backward of the module at B = 64 (the product path: vs_forward_train, vs_backward with its side stream).  From the
     tape as the backward left it, in fp64 on the device:
       head      dlogits = dmask mask (1 - mask); fc2 / fc1 weight and bias gradients as contractions over all 19 264 rows of the
                 bf16-rounded operands; dfc1 and dlstm_out (the gated GEMMs) on a row sample;
       BPTT      the gate gradients of four scattered utterances (both batch tiles) against the fp64 recurrence run backwards over
                 the activated gates / cell states the forward saved (cloned before the backward overwrote them);
       LSTM      dW_ih (3200 x 4808 over K = 19 264, both directions, d-vector columns), dW_hh, the biases, from the tape's gate
                 gradients;
       cnn8      dz8 (BatchNorm + Mish backward of the features, every element), its dgamma / dbeta, conv.28's weight gradient.
  2. The conv stack cnn7 .. cnn1 again through the C ABI's kernel entries (the same launches vs_backward makes, one at a time, each
     result kept): the dy-form data gradient of every layer against an fp64 convolution of the whole dz tensor (every element + the
     two BatchNorm-backward sums), the BatchNorm-backward pass dz = cA dy + cB z + cC in full, EVERY weight gradient of cnn2 .. cnn7
     in full (64 x 64 x 25 numbers from fp64 contractions over all 11.6 M pixels), cnn1's one-pass backward.
  3. The parameter gradients vs_backward itself produced (side-stream schedule) against that chain, and against the serial
     schedule (vs_set_backward_overlap(0)).
  4. configs[4] at its own batch size: a 256-window + a 4-window eval batch through streaming.separate_long_many, four scattered
     windows against the CPU oracle, both arithmetics.
Bounds: one bf16 rounding (`_bf16_close`) for stored bf16 tensors, 2e-5 of the maximum for fp32-accumulated contractions of
bf16-rounded operands, 1e-4 where fp32 BatchNorm coefficients enter.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import reference_forward as R
from test_gpu_b64 import IDX, _bf16_close, _math, _mish, _model

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
CONV_I = [1, 5, 9, 13, 17, 21, 25, 28]                       # Sequential indices of the conv modules; BatchNorm = + 1
SPEC = [(7, 1, 1), (5, 5, 1), (5, 5, 2), (5, 5, 4), (5, 5, 8), (5, 5, 16)]      # cnn2 .. cnn7


def _dmish(y):
    sp = F.softplus(y, threshold=20)
    t = torch.tanh(sp)
    return t + y * (1 - t * t) * torch.sigmoid(y)


def _r(t):
    """the operand a bf16 contraction multiplies: the fp32 value rounded to bf16, as fp64"""
    return t.to(BF).double()


def _rel(got, ref):
    return ((got.double() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-300)).item()


def _conv_full_fp64(x, w, dil, chunk=4):
    """fp64 'same' convolution of a whole channels-last tensor x [B, T, F, 64] (bf16, on the device) with w [64 out, 64 in, KT, KF]
    (fp32: the bf16-rounded weights), time dilation dil: yields (b0, out [chunk, T, F, 64]) per batch chunk -- one fp64 GEMM per tap."""
    B, T, Fq, C = x.shape
    KT, KF = w.shape[2], w.shape[3]
    P, PF = (KT // 2) * dil, KF // 2
    w64 = w.double()
    for b0 in range(0, B, chunk):
        xb = x[b0:b0 + chunk]
        n = xb.shape[0]
        xp = torch.zeros(n, T + 2 * P, Fq + 2 * PF, C, dtype=torch.float64, device=x.device)
        xp[:, P:P + T, PF:PF + Fq] = xb.double()
        out = torch.zeros(n * T * Fq, 64, dtype=torch.float64, device=x.device)
        for dt in range(KT):
            for df in range(KF):
                out.addmm_(xp[:, dt * dil:dt * dil + T, df:df + Fq].reshape(-1, C), w64[:, :, dt, df].t())
        del xp
        yield b0, out.view(n, T, Fq, 64)


def _wgrad_full_fp64(dz, a, KT, KF, dil, chunk=4):
    """dw[co][ci][dt][df] = sum_{b,t,f} dz[b][t][f][co] a[b][t + (dt - KT/2) dil][f + df - KF/2][ci] in fp64 over the whole batch."""
    B, T, Fq, C = a.shape
    P, PF = (KT // 2) * dil, KF // 2
    dw = torch.zeros(64, 64, KT, KF, dtype=torch.float64, device=a.device)
    for b0 in range(0, B, chunk):
        d2 = dz[b0:b0 + chunk].double().reshape(-1, 64).t().contiguous()          # [co, n]
        n = dz[b0:b0 + chunk].shape[0]
        ap = torch.zeros(n, T + 2 * P, Fq + 2 * PF, C, dtype=torch.float64, device=a.device)
        ap[:, P:P + T, PF:PF + Fq] = a[b0:b0 + chunk].double()
        for dt in range(KT):
            for df in range(KF):
                dw[:, :, dt, df] += d2 @ ap[:, dt * dil:dt * dil + T, df:df + Fq].reshape(-1, C)
        del ap, d2
    return dw


def _bn_bwd_check(dy, z, dz_got, scale, mean, invstd, s1, s2, n, what, chunk=4):
    """dz = scale (dy - s1 / n - xhat s2 / n) for the whole tensor (chunked), against the bf16 tensor the pass wrote"""
    sc, mu, iv = scale.double(), mean.double(), invstd.double()
    for b0 in range(0, dy.shape[0], chunk):
        xhat = (z[b0:b0 + chunk].double() - mu) * iv
        ref = sc * (dy[b0:b0 + chunk].double() - s1 / n - xhat * (s2 / n))
        _bf16_close(dz_got[b0:b0 + chunk], ref, (what, b0))
        del xhat, ref


def _bptt_fp64(gates, c, dout, w_hh):
    """Gate gradients [n, T, 8H] of the BiLSTM from its saved activated gates [n, T, 2, 4, H] (i, f, g, o), cell states [n, T, 2H]
    and the gradient of its output dout [n, T, 2H]; exact fp64 (models/voicesplit/model.py:82 backwards, zero initial state)."""
    n, T = gates.shape[0], gates.shape[1]
    H = c.shape[2] // 2
    g = gates.double().view(n, T, 2, 4, H)
    c = c.double().view(n, T, 2, H)
    dout = dout.double().view(n, T, 2, H)
    dxg = torch.zeros(n, T, 2, 4, H, dtype=torch.float64, device=gates.device)
    for d in range(2):
        W = w_hh[d].double()                                                   # [4H, H]
        dh_rec = torch.zeros(n, H, dtype=torch.float64, device=gates.device)
        dc_next = torch.zeros(n, H, dtype=torch.float64, device=gates.device)    # dc_{t'} f_{t'} of the step processed before
        order = range(T - 1, -1, -1) if d == 0 else range(T)                   # reverse of the direction's forward order
        for t in order:
            tp = t - 1 if d == 0 else t + 1                                    # the step whose state this one read
            i, f, gg, o = g[:, t, d, 0], g[:, t, d, 1], g[:, t, d, 2], g[:, t, d, 3]
            ct = c[:, t, d]
            cprev = c[:, tp, d] if 0 <= tp < T else torch.zeros_like(ct)
            dh = dout[:, t, d] + dh_rec
            tc = torch.tanh(ct)
            do = dh * tc * o * (1 - o)
            dc = dh * o * (1 - tc * tc) + dc_next
            di = dc * gg * i * (1 - i)
            df_ = dc * cprev * f * (1 - f)
            dg = dc * i * (1 - gg * gg)
            dc_next = dc * f
            dgates = torch.cat((di, df_, dg, do), dim=1)                       # [n, 4H]
            dxg[:, t, d] = dgates.view(n, 4, H)
            dh_rec = dgates @ W
    return dxg.view(n, T, 8 * H)


def test_b64_train_backward_bf16_stagewise_vs_fp64_on_the_tapes_operands():
    from voicesplit_amd import _lib, ops
    m, sd, dims_d = _model(5, 6.0)
    m.train()
    B, T, Fq, H = 64, 301, 601, dims_d["lstm_dim"]
    E, FC1, FC2 = dims_d["emb_dim"], dims_d["fc1_dim"], dims_d["fc2_dim"]
    M, K8 = B * T, 8 * Fq
    x, dvec = R.synthetic_inputs(B, T, dims_d, 13)
    xc, dc = x.cuda(), dvec.cuda()
    dmask = (torch.randn(B, T, FC2, generator=torch.Generator().manual_seed(21)) * 1e-3).cuda()
    sdc = {k: v.cuda() for k, v in sd.items()}
    lib = _lib.load()

    def step(keep_tape):
        m.zero_grad(set_to_none=True)
        mask = m(xc, dc)
        tape = mask.grad_fn.tape
        saved = None
        if keep_tape:
            dims = ops.make_dims(B, T, Fq, E, H, FC1, FC2)
            lay = ops.tape_layout(dims)
            saved = (ops.ws_view(tape, lay.gates, (B, T, 8 * H))[IDX].clone(), ops.ws_view(tape, lay.cstate, (B, T, 2 * H))[IDX].clone(), lay)
        mask.backward(dmask)
        torch.cuda.synchronize()
        assert m.lstm_status() == 0
        return {k: p.grad.detach().clone() for k, p in m.named_parameters()}, mask.detach(), tape, saved

    with _math("bf16"):
        G, mask, tape, (gates_f, c_f, lay) = step(True)
        for k, v in G.items():
            assert torch.isfinite(v).all(), k

        act_shape = (B, T, Fq, 64)
        z = {l: ops.ws_view(tape, lay.z[l], act_shape, BF) for l in range(1, 7)}
        a = {l: ops.ws_view(tape, lay.a[l], act_shape, BF) for l in range(0, 6)}
        t_scale = ops.ws_view(tape, lay.bn_scale, (8, 64))
        t_shift = ops.ws_view(tape, lay.bn_shift, (8, 64))
        t_mean = ops.ws_view(tape, lay.bn_mean, (8, 64))
        t_invstd = ops.ws_view(tape, lay.bn_invstd, (8, 64))
        z8 = ops.ws_view(tape, lay.z8, (B, T, 8, Fq))
        feat = ops.ws_view(tape, lay.feat, (M, K8))
        lstm_out = ops.ws_view(tape, lay.lstm_out, (M, 2 * H))
        h1 = ops.ws_view(tape, lay.fc1_out, (M, FC1))
        dlogits = ops.ws_view(tape, lay.dlogits, (M, FC2))
        dfc1 = ops.ws_view(tape, lay.dfc1, (M, FC1))
        dlstm = ops.ws_view(tape, lay.dlstm_out, (M, 2 * H))
        dxg = ops.ws_view(tape, lay.gates, (M, 8 * H))                          # the gate gradients, in place of the gates
        dz8 = ops.ws_view(tape, lay.dfeat, (B, T, 8, Fq))                       # the features' gradient after their BatchNorm backward
        npix = float(B * T * Fq)

        # ---- 1a. head ---------------------------------------------------------------------------------------------------------
        mk = mask.reshape(M, FC2).double()
        want = dmask.reshape(M, FC2).double() * mk * (1 - mk)
        assert _rel(dlogits, want) < 1e-5
        del mk, want
        assert _rel(G["fc2.bias"], dlogits.double().sum(0)) < 2e-5
        assert _rel(G["fc2.weight"], _r(dlogits).t() @ _r(h1)) < 2e-5
        assert _rel(G["fc1.bias"], dfc1.double().sum(0)) < 2e-5
        assert _rel(G["fc1.weight"], _r(dfc1).t() @ _r(lstm_out.clamp_min(0))) < 2e-5
        rows = torch.cat([torch.arange(0, M, 41), torch.arange(M - 300, M)]).cuda()          # incl. the ragged last tile rows
        want = (_r(dlogits[rows]) @ _r(sdc["fc2.weight"])) * (h1[rows] > 0)
        assert _rel(dfc1[rows], want) < 2e-5
        want = (_r(dfc1[rows]) @ _r(sdc["fc1.weight"])) * (lstm_out[rows] > 0)
        assert _rel(dlstm[rows], want) < 2e-5

        # ---- 1b. BPTT on four utterances (both batch tiles of the persistent kernel) ----------------------------------------------
        w_hh = [sdc["lstm.weight_hh_l0"], sdc["lstm.weight_hh_l0_reverse"]]
        want = _bptt_fp64(gates_f, c_f, dlstm.view(B, T, 2 * H)[IDX], w_hh)
        got = dxg.view(B, T, 8 * H)[IDX]
        e = _rel(got, want)
        cosv = F.cosine_similarity(got.double().flatten(), want.flatten(), dim=0).item()
        assert e < 4e-3 and cosv > 0.99999, ("bptt", e, cosv)                 # tests/test_gpu_lstm16.py's bound for the bf16 recurrent products
        del want, got, gates_f, c_f

        # ---- 1c. the LSTM's parameter gradients from the tape's gate gradients ------------------------------------------------------
        dxg_r, feat_r = _r(dxg), _r(feat)
        dwih = dxg_r.t() @ feat_r                                              # [8H, 8F]
        dsum = dxg.double().view(B, T, 8 * H).sum(1)                            # [B, 8H]
        dvcols = dsum.t() @ dc.double()                                        # [8H, E]: the repeated d-vector columns
        bsum = dsum.sum(0)
        lo3 = lstm_out.view(B, T, 2 * H)
        for d, sfx in enumerate(("", "_reverse")):
            rows_d = slice(d * 4 * H, (d + 1) * 4 * H)
            gw = G[f"lstm.weight_ih_l0{sfx}"]
            assert _rel(gw[:, :K8], dwih[rows_d]) < 2e-5, ("dW_ih", d)
            assert _rel(gw[:, K8:], dvcols[rows_d]) < 2e-5, ("dW_ih d-vector columns", d)
            assert _rel(G[f"lstm.bias_ih_l0{sfx}"], bsum[rows_d]) < 2e-5 and torch.equal(G[f"lstm.bias_ih_l0{sfx}"], G[f"lstm.bias_hh_l0{sfx}"])
            hprev = torch.zeros(B, T, H, dtype=torch.float64, device="cuda")    # h_{t-1} (forward) / h_{t+1} (reverse), zero at the sequence end
            if d == 0:
                hprev[:, 1:] = _r(lo3[:, :-1, :H])
            else:
                hprev[:, :-1] = _r(lo3[:, 1:, H:])
            want = dxg_r[:, rows_d].t() @ hprev.view(M, H)
            assert _rel(G[f"lstm.weight_hh_l0{sfx}"], want) < 2e-5, ("dW_hh", d)
        del dwih, feat_r, hprev, want

        # ---- 1d. cnn8: BatchNorm + Mish backward of the features (every element), its parameter gradients -------------------------
        wih8 = torch.cat([_r(sdc["lstm.weight_ih_l0"][:, :K8]), _r(sdc["lstm.weight_ih_l0_reverse"][:, :K8])], 0)     # [8H, 8F]
        dfeat = (dxg_r @ wih8).view(B, T, 8, Fq)
        del dxg_r, wih8
        sc8, sh8 = t_scale[7, :8].double().view(1, 1, 8, 1), t_shift[7, :8].double().view(1, 1, 8, 1)
        mu8, iv8 = t_mean[7, :8].double().view(1, 1, 8, 1), t_invstd[7, :8].double().view(1, 1, 8, 1)
        z8d = z8.double()
        dy8 = dfeat * _dmish(z8d * sc8 + sh8)
        xh8 = (z8d - mu8) * iv8
        s1, s2 = dy8.sum((0, 1, 3)), (dy8 * xh8).sum((0, 1, 3))
        want = sc8 * (dy8 - s1.view(1, 1, 8, 1) / npix - xh8 * s2.view(1, 1, 8, 1) / npix)
        assert _rel(dz8, want) < 1e-4
        assert _rel(G["conv.29.weight"], s2) < 1e-4 and _rel(G["conv.29.bias"], s1) < 1e-4
        assert G["conv.28.bias"].abs().max().item() == 0.0                      # a bias in front of a batch-statistics BatchNorm
        del dfeat, z8d, dy8, xh8, want
        w8 = sdc["conv.28.weight"].view(8, 64)
        dw8 = torch.zeros(8, 64, dtype=torch.float64, device="cuda")
        for b0 in range(0, B, 8):
            a7 = _mish(z[6][b0:b0 + 8].double() * t_scale[6].double() + t_shift[6].double()).to(BF).double()      # [8, T, F, 64]
            dw8 += torch.einsum("btof,btfc->oc", dz8[b0:b0 + 8].double(), a7)
            del a7
        assert _rel(G["conv.28.weight"].view(8, 64), dw8) < 1e-4

        # ---- 2. the conv stack again through the kernel entries, every result against fp64 in full ----------------------------------
        chain = {}                                                             # parameter gradients of the piecewise chain
        # cnn8's backward in dy form: the gradient wrt a7, times Mish'(BN(z7)), with the two BatchNorm-backward sums of cnn7
        dy, dw8_u, st = ops.nhwc_conv_last_bwd_dy(dz8.reshape(B, T, 8 * Fq), sdc["conv.28.weight"], None, z[6], "mish",
                                                  t_scale[6], t_shift[6], t_mean[6], t_invstd[6])
        chain["conv.28.weight"] = dw8_u.view(8, 64, 1, 1)
        s1 = torch.zeros(64, dtype=torch.float64, device="cuda")
        s2 = torch.zeros(64, dtype=torch.float64, device="cuda")
        n2 = torch.zeros(64, dtype=torch.float64, device="cuda")
        w8r = w8.double()                                                      # this kernel multiplies in fp32 (VALU): no operand rounding
        for b0 in range(0, B, 4):
            da = torch.einsum("btof,oc->btfc", dz8[b0:b0 + 4].double(), w8r)
            zz = z[6][b0:b0 + 4].double()
            ref = da * _dmish(zz * t_scale[6].double() + t_shift[6].double())
            _bf16_close(dy[b0:b0 + 4], ref, ("dy7", b0), floor=3e-5)
            s1 += ref.sum((0, 1, 2)); s2 += (ref * (zz - t_mean[6].double()) * t_invstd[6].double()).sum((0, 1, 2)); n2 += (ref * ref).sum((0, 1, 2))
            del da, zz, ref

        def check_sums(st, s1, s2, n2, what):
            got = st.sum(0)
            nn = n2.sqrt()
            assert ((got[:, 0] - s1).abs() <= 1e-3 * nn + 1e-4 * s1.abs()).all(), (what, "sum dy")
            assert ((got[:, 1] - s2).abs() <= 2e-3 * nn + 1e-4 * s2.abs()).all(), (what, "sum dy xhat")

        check_sums(st, s1, s2, n2, "cnn7")
        for l in range(6, 0, -1):                                              # conv index l = cnn(l+1): 6 = cnn7 ... 1 = cnn2
            KT, KF, dil = SPEC[l - 1]
            ci = CONV_I[l]
            # dy (of this layer's BatchNorm output) -> dz: the BatchNorm backward pass, with the sums the producer of dy left
            dz, dg, db, dbias = ops.nhwc_bn_bwd_from_dy(dy, z[l], st, True, t_scale[l], t_mean[l], t_invstd[l])
            _bn_bwd_check(dy, z[l], dz, t_scale[l], t_mean[l], t_invstd[l], s1, s2, npix, ("dz", l + 1))
            assert _rel(dg, s2) < 1e-4 and _rel(db, s1) < 1e-4 and dbias.abs().max().item() == 0.0
            chain[f"conv.{ci + 1}.weight"], chain[f"conv.{ci + 1}.bias"] = dg, db
            del dy
            # the weight gradient, every one of its 64 x 64 x KT x KF numbers
            dw = ops.nhwc_conv_wgrad(dz, a[l - 1], KT, KF, dil)
            want = _wgrad_full_fp64(dz, a[l - 1], KT, KF, dil)
            assert _rel(dw, want) < 2e-5, ("weight gradient", l + 1, _rel(dw, want))
            chain[f"conv.{ci}.weight"] = dw
            del want
            # the data gradient: dy form for cnn3 .. cnn7 (the activation derivative and BatchNorm-backward sums of the layer below),
            # the plain conv for cnn2 (cnn1's backward recomputes z1 from x)
            wt = sdc[f"conv.{ci}.weight"].to(BF).float().transpose(0, 1).flip(2, 3).contiguous()          # [ci, co, KT-1-dt, KF-1-df]
            packed = ops.nhwc_conv_pack(sdc[f"conv.{ci}.weight"], transpose_flip=True)
            s1.zero_(); s2.zero_(); n2.zero_()
            if l > 1:
                dy, st = ops.nhwc_conv_dy(dz, packed, z[l - 1], "mish", t_scale[l - 1], t_shift[l - 1], t_mean[l - 1], t_invstd[l - 1], KT, KF, dil)
                for b0, da in _conv_full_fp64(dz, wt, dil):
                    zz = z[l - 1][b0:b0 + da.shape[0]].double()
                    ref = da * _dmish(zz * t_scale[l - 1].double() + t_shift[l - 1].double())
                    _bf16_close(dy[b0:b0 + da.shape[0]], ref, ("dy", l, b0), floor=3e-5)
                    s1 += ref.sum((0, 1, 2)); s2 += (ref * (zz - t_mean[l - 1].double()) * t_invstd[l - 1].double()).sum((0, 1, 2))
                    n2 += (ref * ref).sum((0, 1, 2))
                    del da, zz, ref
                check_sums(st, s1, s2, n2, f"cnn{l}")
            else:
                ones, zeros = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
                da1 = torch.empty_like(dz)
                _lib.check(lib.vs_nhwc_conv(ops._p(dz), ops._p(packed), ops._p(ones), ops._p(zeros), ops._p(da1), B, T, Fq, KT, KF, dil,
                                            _lib.ACT_NONE, ops._p(None), ops._stream()), "vs_nhwc_conv")
                for b0, da in _conv_full_fp64(dz, wt, dil):
                    _bf16_close(da1[b0:b0 + da.shape[0]], da, ("da1", b0), floor=3e-5)
                    del da
            del dz
        # cnn1: one pass over da1 with z1 recomputed from x -> its four parameter gradients
        w1, b1 = sdc["conv.1.weight"], sdc["conv.1.bias"]
        dw1, dg1, db1, dbias1 = ops.nhwc_first_bwd(da1, xc, w1, b1, "mish", True, t_scale[0], t_shift[0], t_mean[0], t_invstd[0])
        chain["conv.1.weight"], chain["conv.2.weight"], chain["conv.2.bias"] = dw1.view(64, 1, 1, 7), dg1, db1
        w1d, b1d = w1.double().view(64, 7), b1.double()
        sc, sh, mu, iv = t_scale[0].double(), t_shift[0].double(), t_mean[0].double(), t_invstd[0].double()

        def z1_chunk(b0, n):
            xp = F.pad(xc[b0:b0 + n].double(), (3, 3))                          # [n, T, F + 6]
            shifts = torch.stack([xp[:, :, k:k + Fq] for k in range(7)], dim=-1)      # [n, T, F, 7]
            return shifts @ w1d.t() + b1d, shifts                              # [n, T, F, 64]

        s1.zero_(); s2.zero_()
        for b0 in range(0, B, 4):
            z1, _ = z1_chunk(b0, 4)
            ref = da1[b0:b0 + 4].double() * _dmish(z1 * sc + sh)
            s1 += ref.sum((0, 1, 2)); s2 += (ref * (z1 - mu) * iv).sum((0, 1, 2))
            del z1, ref
        want_dw1 = torch.zeros(64, 7, dtype=torch.float64, device="cuda")
        for b0 in range(0, B, 4):
            z1, shifts = z1_chunk(b0, 4)
            dyv = da1[b0:b0 + 4].double() * _dmish(z1 * sc + sh)
            dz1 = sc * (dyv - s1 / npix - (z1 - mu) * iv * (s2 / npix))
            want_dw1 += torch.einsum("btfc,btfk->ck", dz1, shifts)
            del z1, shifts, dyv, dz1
        assert _rel(dw1, want_dw1) < 1e-4 and _rel(dg1, s2) < 1e-4 and _rel(db1, s1) < 1e-4 and dbias1.abs().max().item() == 0.0
        del da1

        # ---- 3. vs_backward's own conv gradients (side-stream schedule) against the chain, and against the serial schedule ----------
        # the same launches on the same operands; the BatchNorm sums are fp64 atomics in arrival order, so a last bit of a coefficient --
        # and with it single bf16 roundings of dz -- may differ between two runs
        for k, v in chain.items():
            assert _rel(G[k], v) < 1e-5, ("vs_backward vs the kernel chain", k, _rel(G[k], v))
        try:
            assert lib.vs_set_backward_overlap(0) == 0
            G0 = step(False)[0]
        finally:
            lib.vs_set_backward_overlap(1)
        G1 = step(False)[0]
        identical = all(torch.equal(G0[k], G1[k]) for k in G0)
        for k in G0:
            assert _rel(G1[k], G0[k]) < 1e-5, ("side-stream vs serial schedule", k, _rel(G1[k], G0[k]))
            assert _rel(G1[k], G[k]) < 1e-5, ("run to run", k)
        print("b64 backward: serial and side-stream schedules bit-identical:", identical)


@pytest.mark.parametrize("math", ["f16x3", "bf16"])
def test_long_form_batch_of_256_windows_scattered_windows_vs_oracle(math):
    """BASELINE configs[4] at its own batch size: 26 clips of 3001 frames = 260 windows of 301 frames (the last of a clip zero
    padded) through streaming.separate_long_many = one forward batch of 256 windows + one of 4; four scattered windows -- the
    first, one in the middle, a zero-padded last window, the last of the 256-batch and one of the 4-batch -- against the CPU oracle
    run on that window alone."""
    from voicesplit_amd import streaming
    m, sd, dims_d = _model(6, 8.0)
    m.eval()
    N, TL, Fq = 26, 3001, 601
    g = torch.Generator().manual_seed(17)
    specs = torch.rand(N, TL, Fq, generator=g)
    dvecs = F.normalize(torch.randn(N, dims_d["emb_dim"], generator=g), dim=1)
    with _math(math):
        masks = streaming.separate_long_many(m, specs.cuda(), dvecs.cuda(), window=301, max_batch=256)
    torch.cuda.synchronize()
    assert masks.shape == (N, TL, dims_d["fc2_dim"]) and torch.isfinite(masks).all()
    worst_rel, worst_abs, se, cnt = 0.0, 0.0, 0.0, 0
    for clip, w in ((0, 0), (7, 3), (13, 9), (25, 5), (25, 9)):                # window index clip * 10 + w: 0, 73, 139, 255 | 259
        lo, hi = w * 301, min((w + 1) * 301, TL)
        xw = torch.zeros(1, 301, Fq)
        xw[0, :hi - lo] = specs[clip, lo:hi]
        with torch.no_grad():
            ref = R.forward(sd, xw, dvecs[clip:clip + 1], act="mish")["mask"][0, :hi - lo]
        err = (masks[clip, lo:hi].cpu().double() - ref.double()).abs()
        worst_rel = max(worst_rel, float(err.max() / ref.abs().max()))
        worst_abs = max(worst_abs, float(err.max()))
        se += float((err ** 2).sum()); cnt += err.numel()
    assert se / cnt <= 1e-4
    if math == "f16x3":
        assert worst_rel <= 1e-4, worst_rel
    else:
        assert worst_abs <= 6e-2, worst_abs


@pytest.mark.parametrize("B", [64, 5])
def test_deterministic_mode_reruns_bit_identically(B):
    """vs_set_option(VS_OPT_DETERMINISTIC, 1) -- SURVEY.md section 5 lists deterministic-rerun comparisons as the device-side sanitizer, and
    the reference's PyTorch BatchNorm is run-to-run reproducible: the workgroups that add partial sums to shared slots (BatchNorm
    statistics and their backward sums, cnn1's moments, the features' BatchNorm backward, the loss head's moments) take turns in
    workgroup order.  Three training steps of the metric configuration (bf16, SI-SNR criterion through the GPU iSTFT, Adam), run twice from
    the same initialisation: masks, loss values, every gradient of every step, the final weights and the BatchNorm running
    statistics must agree bit for bit.  (In the default mode they agree to the last bits of fp64 sums only; whether this pair of
    runs happened to is printed, not asserted.)"""
    import voicesplit_amd as V
    from voicesplit_amd import _lib
    from voicesplit_amd.trainer import Trainer, synthetic_batches
    c = V.default_config()
    c.train_config["learning_rate"] = 1e-3
    acfg = c.audio[c.audio["backend"]]
    batch = next(iter(synthetic_batches(1, B, 301, 601, 256, acfg["hop_length"], torch.device("cuda"), seed=31)))

    def run():
        torch.manual_seed(17)
        tr = Trainer(V.VoiceSplit(c).cuda(), c)
        out = {}
        for s in range(3):
            out[f"loss{s}"] = torch.tensor(tr.train_step(batch))
            for k, p in tr.model.named_parameters():
                out[f"grad{s}/{k}"] = p.grad.detach().clone()
        torch.cuda.synchronize()
        assert tr.model.lstm_status() == 0
        out.update({"w/" + k: v.detach().clone() for k, v in tr.model.state_dict().items()})
        return out

    prev = _lib.get_option("DETERMINISTIC")
    try:
        with _math("bf16"):
            _lib.set_option("DETERMINISTIC", 0)
            a, b = run(), run()
            same_default = all(torch.equal(a[k], b[k]) for k in a)
            _lib.set_option("DETERMINISTIC", 1)
            d1, d2 = run(), run()
    finally:
        _lib.set_option("DETERMINISTIC", prev)
    print(f"B = {B}: default mode, two runs bit-identical: {same_default}")
    for k in d1:
        assert torch.equal(d1[k], d2[k]), k
    # the mode changes the ORDER of fp64 additions, nothing else: the results agree with the default mode's to fp32 rounding of the
    # BatchNorm coefficients (and what a flipped bf16 rounding downstream of one amounts to)
    for k in d1:
        if k.startswith("loss"):
            assert abs(float(d1[k]) - float(a[k])) <= 1e-3 * max(1.0, abs(float(a[k]))), (k, float(d1[k]), float(a[k]))
