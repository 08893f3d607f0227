"""GPU: VS_MATH_BF16 (BASELINE configs[2]: bf16 forward + backward, the configuration bench.py's headline is quoted on).

What this arithmetic is (round 3 onwards): conv activations and the training tape are STORED as
channels-last bf16 [B][T][F][64] (z = conv + bias and a = act(BN(z)) of cnn1..cnn7, the gradients between layers), every
dense contraction -- the 64->64 convs forward / data gradient / weight gradient, the three LSTM GEMMs, fc1 / fc2 and their
backward contractions, dW_hh -- issues ONE bf16 MFMA product per product on bf16-rounded operands with fp32 accumulation;
the recurrent product h @ W_hh^T runs on f16 operands forward and on bf16 operands in the BPTT.  fp32 stay: accumulators,
BatchNorm statistics (fp64 across workgroups), gate arithmetic and cell state, the loss head, master weights / state_dict.

bf16 keeps 8 significant bits (2^-9 = 2e-3 relative rounding per stored value), so this mode does NOT meet the path's fp32
contract (1e-4) and never runs unless asked for.  It is held to the bounds below against the same oracles as the default
arithmetic, on stress fixtures (randomised BatchNorm statistics, recurrent / head weights scaled up so that logits spread --
chosen to make errors visible) and on the pinned metric-configuration fixture of the UPSTREAM module.
Measured on MI355X this round (round 4, after cnn1 became a recomputed fp32 layer; every value is dumped to
gpurun_out/errors_bf16_*.json, errors_trainer_*.json):

  small stress fixtures            frozen BatchNorm       batch-statistics BatchNorm          bound asserted
                                   VoiceSplit/VoiceFilter VoiceSplit (Mish) VoiceFilter (ReLU)
  conv stack output (rel. range)   9e-4 / 1.2e-3          1.1e-2            1.8e-2             3e-2
  LSTM output                      2.9e-3 / 5.1e-3        4.7e-2 .. 5.8e-2  4.8e-2             8e-2
  mask, max abs                    3.1e-3 / 5.0e-3        4.0e-2            4.0e-2             6e-2
  mask MSE (BASELINE: <= 1e-4)     1e-6                   2e-5              4e-5               1e-4
  gradients, max / tensor max      6e-2 / 9e-2            0.28 .. 0.52      0.46               0.6
  gradients, cosine vs fp64        0.998                  0.977 .. 0.981    0.936 (*)          0.93 (Mish) / 0.90 (ReLU)
  (ranges: three runs on three boxes this round -- the fp64 atomics of the BatchNorm sums are unordered, and these fixtures amplify an ulp)

  metric configuration, full size, B = 8, upstream fixture:  mask max abs 0.043, mask MSE 4.52e-5 (ideal bf16 storage: 4.53e-5),
  worst gradient 0.28 of its tensor's maximum / cosine 0.980 (ideal: 0.23 / 0.982), the pooled small vectors 0.19 relative L2 /
  cosine 0.983 (ideal: 0.17 / 0.986); bounds 0.45 / 0.95 and the envelope test.  Trainer + gradient sink + SI-SNR head
  (tests/test_gpu_trainer.py): loss 27.8116 vs 27.8121, worst gradient 0.59 / cosine 0.979 (full dims, B = 4, T = 31), 0.18 / 0.994
  (small); 40-step trajectory 27.5 -> 15.6 vs 27.5 -> 15.9 fp32-class.

Bounds and why.  Forward quantities and the gradient maximum are held to the bounds of round 2 (3e-2 / 8e-2 / 6e-2 / 0.6) for BOTH
models again: round 3 had widened them for the ReLU model (0.12 / 0.1) when tape and inter-layer gradients moved to bf16 storage; with
cnn1 recomputed from x in fp32 (no bf16 z1, a1 rounded once) the measurements are back inside.  (*) The cosine bound of the ReLU
model stays at round 3's 0.90: its worst tensor is conv.29.bias, EIGHT numbers that are each a sum of 10^5 terms of either sign --
0.936 here, 0.927 in round 3, 0.995 for the Mish model on the same inputs: sampling noise of an 8-element vector, which the
full-size tests below therefore judge pooled (see _judged).  The mask MSE bound (BASELINE's 1e-4) holds with 2.2x margin at worst.
(Batch statistics subtract each channel's mean, so a rounding error that is 2e-3 of |z| becomes a larger fraction of the normalised
value whenever the mean dominates the spread; the recurrence with scaled-up W_hh amplifies what reaches it.)  How much of the
gradient error is the arithmetic's own is measured, not argued: oracle/bf16_model.py (fp64 with bf16 rounding injected at this
path's storage points and nothing else) gives the error of a PERFECT implementation of bf16 storage; the envelope tests hold the
kernels to it on the small fixture AND on the full-size metric fixture."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden_grads
from oracle import reference_backward as RB
from oracle import reference_forward as R

pytestmark = pytest.mark.gpu
FEAT_TOL = 3e-2
LSTM_TOL = {"mish": 8e-2, "relu": 8e-2}
MASK_ABS_TOL = {"mish": 6e-2, "relu": 6e-2}
GRAD_TOL = 0.6
COS_MIN = {"mish": 0.93, "relu": 0.90}
FULL_GRAD_TOL = 0.45       # the full-size metric-configuration fixture (vs_full_b8_train_grads)
FULL_COS_MIN = 0.95


def _rel(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def _cos(got, ref):
    got = got.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    return float((got @ ref) / (got.norm() * ref.norm()).clamp_min(1e-300))


def _dump(name, table):
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"errors_bf16_{name}.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)


class _math:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        from voicesplit_amd import ops
        self.prev = ops.get_conv_math()
        ops.set_conv_math(self.name)

    def __exit__(self, *exc):
        from voicesplit_amd import ops
        ops.set_conv_math(self.prev)


@pytest.mark.parametrize("cls_name,act", [("VoiceSplit", "mish"), ("VoiceFilter", "relu")])
@pytest.mark.parametrize("training", [True, False])
def test_bf16_module_forward_and_backward_vs_fp64_oracle(cls_name, act, training):
    import voicesplit_amd as V
    from voicesplit_amd import ops
    dims_d = dict(num_freq=53, emb_dim=24, lstm_dim=32, fc1_dim=44, fc2_dim=53)
    B, T = 3, 45
    sd = R.spread_logits(R.build_state_dict(dims_d, 21), 6.0)
    x, dvec = R.synthetic_inputs(B, T, dims_d, 21)
    w = RB.loss_weights(B, T, 53, 21)
    m = getattr(V, cls_name)(V.default_config(53, 24, 32, 44, 53))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train(training)
    with _math("bf16"):
        mask = m(x.cuda(), dvec.cuda())
        tape = mask.grad_fn.tape
        dims = ops.make_dims(B, T, 53, 24, 32, 44, 53)
        lay = ops.tape_layout(dims)
        feat = ops.ws_view(tape, lay.feat, (B, T, 8 * 53)).clone()
        lstm_out = ops.ws_view(tape, lay.lstm_out, (B, T, 64)).clone()
        (mask * w.cuda()).sum().backward()
    stages = {}
    ref = RB.gradients(sd, x, dvec, w, act=act, training=training, dtype=torch.float64, lstm_impl="loop", stages=stages)
    table = {"fwd/feat": _rel(feat, stages["val/feat"]), "fwd/lstm_out": _rel(lstm_out, stages["val/lstm_out"]),
             "fwd/mask_abs": float((mask.detach().double().cpu() - stages["mask"]).abs().max()),
             "fwd/mask_mse": float(((mask.detach().double().cpu() - stages["mask"]) ** 2).mean())}
    zero = {f"conv.{i}.bias" for i in (1, 5, 9, 13, 17, 21, 25, 28)} if training else set()
    for k, p in m.named_parameters():
        if k in zero:
            assert p.grad.abs().max().item() == 0.0
            continue
        table["grad/" + k] = _rel(p.grad, ref[k])
        table["cos/" + k] = _cos(p.grad, ref[k])
    _dump(f"{cls_name}_{training}", table)
    assert table["fwd/feat"] < FEAT_TOL and table["fwd/lstm_out"] < LSTM_TOL[act], table
    assert table["fwd/mask_abs"] < MASK_ABS_TOL[act] and table["fwd/mask_mse"] < 1e-4, table
    bad = {k: v for k, v in table.items() if (k.startswith("grad/") and not v < GRAD_TOL) or (k.startswith("cos/") and not v >= COS_MIN[act])}
    assert not bad, bad            # `not <`: a NaN gradient is a failure, not a pass


_FULL_B8 = {}
SMALL = 256          # tensors below this many elements (the 64- / 8-element BatchNorm and conv-bias vectors) are judged POOLED


def _judged(vectors, g):
    """{name: vector} as the full-size tests judge them.  The gradient of this fixture's loss is a sum of ~10^7 terms of either
    sign per parameter; for an 8- or 64-element vector the cosine / max error of bf16 storage is then sampling noise (conv.29.bias,
    8 elements: cosine 0.971 in round 3, 0.879 after cnn1 became MORE accurate; the ideal-bf16 model moves the same way) --
    so every vector below SMALL elements is scaled by its tensor's maximum and concatenated into one vector `small*`, judged with
    the same bounds as the large tensors (every element stays under test; the noise of ~1500 elements averages)."""
    out, pool_v, pool_r = {}, [], []
    for k, v in vectors.items():
        ref = g["grads"][k].astype(np.float64)
        if v.size >= SMALL:
            out[k] = (v, ref, max(g["gabs"][k], 1e-30))
        else:
            pool_v.append(v / max(g["gabs"][k], 1e-30))
            pool_r.append(ref / max(g["gabs"][k], 1e-30))
    out["small*"] = (np.concatenate(pool_v), np.concatenate(pool_r), None)
    return out


def _err_cos(v, ref, scale):
    """(error, cosine): error = max |v - ref| / tensor maximum; for the pooled vector (scale None) the relative L2 error -- a maximum
    over the pool would again be the one noisiest element."""
    err = np.abs(v - ref).max() / scale if scale is not None else np.linalg.norm(v - ref) / max(np.linalg.norm(ref), 1e-300)
    return float(err), float((v @ ref) / max(np.linalg.norm(v) * np.linalg.norm(ref), 1e-300))


def _full_b8_run():
    """One bf16 training-mode forward + backward of the pinned metric configuration (full size, 8 utterances), shared by
    the two tests below: (table of errors vs the UPSTREAM fixture, thinned HIP gradients)."""
    if _FULL_B8:
        return _FULL_B8["table"], _FULL_B8["grads"], _FULL_B8["g"]
    import voicesplit_amd as V
    g = load_golden_grads("vs_full_b8_train_grads")
    d = g["dims"]
    sd = R.spread_logits(R.build_state_dict(d, g["seed"]), g["gain"])
    x, dvec = R.synthetic_inputs(g["B"], g["T"], d, g["seed"])
    w = RB.loss_weights(g["B"], g["T"], d["fc2_dim"], g["seed"])
    m = V.VoiceSplit(V.default_config())
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train(True)
    with _math("bf16"):
        mask = m(x.cuda(), dvec.cuda())
        (mask * w.cuda()).sum().backward()
    table = {"fwd/mask_abs": float(np.abs(mask.detach().cpu().numpy()[:, ::8] - g["fwd/mask"]).max()),
             "fwd/mask_mse": float(((mask.detach().cpu().numpy()[:, ::8] - g["fwd/mask"]) ** 2).mean())}
    zero = {f"conv.{i}.bias" for i in (1, 5, 9, 13, 17, 21, 25, 28)}
    grads = {}
    for k, p in m.named_parameters():
        if k in zero:
            continue
        grads[k] = RB.thin_grad(p.grad.detach().cpu()).double().numpy()
    for k, (v, ref, scale) in _judged(grads, g).items():
        table["grad/" + k], table["cos/" + k] = _err_cos(v, ref, scale)
    for k, v in grads.items():                  # the raw per-tensor figures of the small vectors, for the record only
        if v.size < SMALL:
            e, c = _err_cos(v, g["grads"][k].astype(np.float64), max(g["gabs"][k], 1e-30))
            table["unpooled/grad/" + k], table["unpooled/cos/" + k] = e, c
    _dump("vs_full_b8_train", table)
    _FULL_B8.update(table=table, grads=grads, g=g)
    del m, mask
    from voicesplit_amd import ops
    ops.release_workspaces()
    torch.cuda.empty_cache()
    return table, grads, g


def test_bf16_metric_configuration_vs_upstream_golden():
    """Full size, 8 utterances, batch-statistics BatchNorm (the pinned metric configuration) in bf16 arithmetic
    against the UPSTREAM module's forward tensors and gradients."""
    table, _, _ = _full_b8_run()
    assert table["fwd/mask_abs"] < MASK_ABS_TOL["mish"] and table["fwd/mask_mse"] < 1e-4, table      # 0.043 measured over 301 x 601 x 8 values
    # this fixture (default-scale activations at full size) is milder than the small stress fixtures above: its own, tighter
    # bounds -- measured 0.28 of a tensor's maximum / cosine 0.980 at worst (ideal bf16 storage: 0.23 / 0.982), small vectors pooled
    bad = {k: v for k, v in table.items() if (k.startswith("grad/") and not v < FULL_GRAD_TOL) or (k.startswith("cos/") and not v >= FULL_COS_MIN)}
    assert not bad, bad            # `not <`: a NaN gradient is a failure, not a pass


def test_bf16_metric_configuration_inside_the_ideal_envelope_at_full_size():
    """The envelope argument ON the fixture it is used to excuse (VERDICT round 3, next #1a): tests/golden/
    vs_full_b8_train_bf16_ideal.npz holds the gradients of oracle/bf16_model.py -- the reference graph with bf16 rounding
    injected at this path's storage points and nothing else -- on the metric configuration's fixture (made by
    `python -m oracle.make_golden --bf16-envelope`).  Against the UPSTREAM gradients of that fixture the HIP path may be
    worse than that ideal by a rounding-order margin only: worst tensor error <= 1.5 x ideal + 0.05, worst cosine >=
    ideal - 0.03, and the same per tensor with the margins of one more rounding (2 x + 0.05, - 0.05)."""
    table, grads, g = _full_b8_run()
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vs_full_b8_train_bf16_ideal.npz"))
    assert str(z["sd_sha256"]) == g["sd_sha256"]
    env = {}
    ideal_j = _judged({k: z["grad/" + k].astype(np.float64) for k in grads}, g)
    hip_j = _judged(grads, g)
    for k, (ideal, ref, scale) in ideal_j.items():
        env["ideal_err/" + k], env["ideal_cos/" + k] = _err_cos(ideal, ref, scale)
        env["hip_err/" + k], env["hip_cos/" + k] = table["grad/" + k], table["cos/" + k]
        got = hip_j[k][0]
        env["hip_vs_ideal_cos/" + k] = float((got @ ideal) / max(np.linalg.norm(got) * np.linalg.norm(ideal), 1e-300))
    env["ideal_mask_mse"] = float(((z["fwd/mask"].astype(np.float64) - g["fwd/mask"]) ** 2).mean())
    env["hip_mask_mse"] = table["fwd/mask_mse"]
    _dump("vs_full_b8_envelope", env)
    hip_err = max(v for k, v in env.items() if k.startswith("hip_err/"))
    ideal_err = max(v for k, v in env.items() if k.startswith("ideal_err/"))
    hip_cos = min(v for k, v in env.items() if k.startswith("hip_cos/"))
    ideal_cos = min(v for k, v in env.items() if k.startswith("ideal_cos/"))
    assert hip_err == hip_err and hip_cos == hip_cos
    assert hip_err <= 1.5 * ideal_err + 0.05, (hip_err, ideal_err)
    assert hip_cos >= ideal_cos - 0.03, (hip_cos, ideal_cos)
    bad = {k: (env["hip_err/" + k], env["ideal_err/" + k], env["hip_cos/" + k], env["ideal_cos/" + k]) for k in ideal_j
           if not (env["hip_err/" + k] <= 2.0 * env["ideal_err/" + k] + 0.05 and env["hip_cos/" + k] >= env["ideal_cos/" + k] - 0.05)}
    assert not bad, bad
    assert env["hip_mask_mse"] <= 2.0 * env["ideal_mask_mse"] + 2e-5, (env["hip_mask_mse"], env["ideal_mask_mse"])


def test_bf16_gradients_sit_inside_the_ideal_bf16_envelope():
    """How much of the gradient error above is the arithmetic's and how much the kernels'?  oracle/bf16_model.py is the
    reference in fp64 with bf16 rounding injected at the storage points of this configuration and NOTHING else changed:
    its gradients already differ from the exact ones by 0.3-0.6 of a tensor's maximum at cosine 0.94-0.99 on this
    (default-initialised, batch-statistics) fixture -- `sum(mask * w)` adds 10^5 terms of either sign, so its gradient is
    badly conditioned.  The HIP path must not be worse than that ideal by more than a rounding-order margin."""
    import voicesplit_amd as V
    from oracle import bf16_model
    dims_d = dict(num_freq=101, emb_dim=16, lstm_dim=24, fc1_dim=40, fc2_dim=101)
    B, T = 4, 80
    sd = R.build_state_dict(dims_d, 31)
    x, dvec = R.synthetic_inputs(B, T, dims_d, 31)
    w = RB.loss_weights(B, T, 101, 31)
    exact, mask_exact = bf16_model.gradients(sd, x, dvec, w, bf16=False)
    ideal, _ = bf16_model.gradients(sd, x, dvec, w, bf16=True)
    m = V.VoiceSplit(V.default_config(101, 16, 24, 40, 101))
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train(True)
    with _math("bf16"):
        mask = m(x.cuda(), dvec.cuda())
        (mask * w.cuda()).sum().backward()
    zero = {f"conv.{i}.bias" for i in (1, 5, 9, 13, 17, 21, 25, 28)}
    table = {"fwd/mask_mse": float(((mask.detach().double().cpu() - mask_exact) ** 2).mean())}
    for k, p in m.named_parameters():
        if k in zero:
            continue
        table["hip_err/" + k], table["hip_cos/" + k] = _rel(p.grad, exact[k]), _cos(p.grad, exact[k])
        table["ideal_err/" + k], table["ideal_cos/" + k] = _rel(ideal[k], exact[k]), _cos(ideal[k], exact[k])
    _dump("ideal_envelope", table)
    hip_err = max(v for k, v in table.items() if k.startswith("hip_err/"))
    ideal_err = max(v for k, v in table.items() if k.startswith("ideal_err/"))
    hip_cos = min(v for k, v in table.items() if k.startswith("hip_cos/"))
    ideal_cos = min(v for k, v in table.items() if k.startswith("ideal_cos/"))
    assert table["fwd/mask_mse"] < 1e-4
    assert hip_err == hip_err and hip_cos == hip_cos                      # no NaN
    assert hip_err <= 1.5 * ideal_err + 0.05, (hip_err, ideal_err)
    assert hip_cos >= ideal_cos - 0.03, (hip_cos, ideal_cos)


def test_bf16_is_never_the_default():
    from voicesplit_amd import ops
    assert ops.get_conv_math() in ("f16x3", "fp32")


def test_head_data_gradients_on_the_lstm_gemm_kernel_agree_with_the_generic_kernel():
    """vs_set_option(VS_OPT_HEAD_BWD_GEMM): dfc1 = (dlogits @ W2) * (h1 > 0) and dlstm_out = (dfc1 @ W1) * (lstm_out > 0) on
    gemm_bf16.hip's kernel (bf16 copies of the operands, relu mask in the epilogue) against the generic kernel that converts the
    same operands to bf16 while it stages them: same products, another fp32 summation order -- the head's own gradients agree to
    fp32 rounding, everything below the BPTT to the bf16 storage rounding a changed last bit can flip."""
    import voicesplit_amd as V
    from voicesplit_amd import _lib
    dims_d = dict(num_freq=601, emb_dim=256, lstm_dim=32, fc1_dim=48, fc2_dim=601)
    sd = R.spread_logits(R.build_state_dict(dims_d, 5), 6.0)
    x, dvec = R.synthetic_inputs(3, 70, dims_d, 5)
    w = torch.randn(3, 70, 601, generator=torch.Generator().manual_seed(9)).cuda()
    prev = _lib.get_option("HEAD_BWD_GEMM")
    got = {}
    try:
        with _math("bf16"):
            for mode in (1, 0):
                _lib.set_option("HEAD_BWD_GEMM", mode)
                m = V.VoiceSplit(V.default_config(601, 256, 32, 48, 601))
                m.load_state_dict(sd, strict=True)
                m = m.cuda().train(True)
                (m(x.cuda(), dvec.cuda()) * w).sum().backward()
                torch.cuda.synchronize()
                got[mode] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    finally:
        _lib.set_option("HEAD_BWD_GEMM", prev)
    for k in got[1]:
        assert torch.isfinite(got[1][k]).all(), k
        tol = 1e-5 if k.startswith("fc2") else (2e-4 if k.startswith("fc1") else 2e-2)
        assert _rel(got[1][k], got[0][k]) < tol, (k, _rel(got[1][k], got[0][k]))

def test_every_schedule_of_the_bf16_step_gives_the_same_bits():
    """the side-stream schedule rests on buffer-idle invariants that live in comments (the second half of the column-sum
    scratch, `part` touched on the side stream only, the conv gradient buffers idle while the head's bf16 operands sit there).  Every
    combination of vs_set_backward_overlap x VS_OPT_HEAD_LEAF_SIDE x VS_OPT_FWD_PROLOGUE runs the same launches on the same operands in
    another stream order: all sixteen runs (each combination twice, to catch a missing fork or join) must give bit-identical masks,
    running statistics and gradients."""
    import itertools
    import voicesplit_amd as V
    from voicesplit_amd import _lib
    dims_d = dict(num_freq=601, emb_dim=256, lstm_dim=32, fc1_dim=48, fc2_dim=601)
    sd = R.spread_logits(R.build_state_dict(dims_d, 5), 6.0)
    x, dvec = R.synthetic_inputs(3, 70, dims_d, 5)
    w = torch.randn(3, 70, 601, generator=torch.Generator().manual_seed(9)).cuda()
    lib = _lib.load()
    prev = {k: _lib.get_option(k) for k in ("HEAD_LEAF_SIDE", "FWD_PROLOGUE")}
    runs = {}
    try:
        with _math("bf16"):
            for overlap, leaf, pro in itertools.product((1, 0), (1, 0), (1, 0)):
                assert lib.vs_set_backward_overlap(overlap) == 0
                _lib.set_option("HEAD_LEAF_SIDE", leaf)
                _lib.set_option("FWD_PROLOGUE", pro)
                m = V.VoiceSplit(V.default_config(601, 256, 32, 48, 601))
                m.load_state_dict(sd, strict=True)
                m = m.cuda().train(True)
                out = []
                for _ in range(2):
                    m.zero_grad(set_to_none=True)
                    mask = m(x.cuda(), dvec.cuda())
                    (mask * w).sum().backward()
                    torch.cuda.synchronize()
                    st = {"mask": mask.detach().clone(), **{"grad/" + k: p.grad.detach().clone() for k, p in m.named_parameters()},
                          **{"buf/" + k: b.detach().clone() for k, b in m.named_buffers()}}
                    out.append(st)
                runs[(overlap, leaf, pro)] = out
    finally:
        lib.vs_set_backward_overlap(1)
        for k, v in prev.items():
            _lib.set_option(k, v)
    base = runs[(1, 1, 1)]
    for key, out in runs.items():
        for a, b in zip(base, out):
            for k in a:
                assert torch.equal(a[k], b[k]), (key, k)
