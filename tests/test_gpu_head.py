"""GPU: the mask head of models/voicesplit/model.py:83-87 as ONE kernel in VS_MATH_BF16 (csrc/head_fused.hip):
mask = sigmoid(fc2(relu(fc1(relu(lstm_out))))), h1 chained through registers between the two contractions.

Against fp64 on the same bf16-rounded operands (relu(lstm_out), fc1.weight, fc2.weight rounded to bf16; h1 rounded to bf16 before
fc2; fp32 biases) -- the roundings of the two-launch form it replaces.  What is left is fp32 accumulation order (1e-7 of the
logits: the MEAN bound), plus the rare h1 element that lands on the other side of a bf16 rounding boundary because of it: one
bf16 ulp of one of FC1 terms, |h1| 2^-8 |w2| ~ 3e-3 absolute at these scales (measured 3.1e-3 of logits up to 10: the MAX bound)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3        # max |logits - ref| / max |ref|: one h1 element rounded the other way
LOGIT_MEAN_TOL = 2e-6   # mean |logits - ref| / max |ref|: fp32 accumulation order
MASK_TOL = 1.5e-3       # absolute, on sigmoid outputs (slope <= 1/4)


def _bf(t):
    return t.to(torch.bfloat16).double()


def _head_ref(lo, w1, b1, w2, b2):
    h1 = (_bf(lo.clamp_min(0)) @ _bf(w1).t() + b1.double()).clamp_min(0)
    logits = _bf(h1.float()) @ _bf(w2).t() + b2.double()
    return h1, logits, torch.sigmoid(logits)


def _state(H, FC1, FC2, seed):
    """A whole state dict (ops.head packs every parameter pointer); the head's own entries re-drawn with logits of a few units."""
    from oracle import reference_forward as R
    sd = R.build_state_dict(dict(num_freq=53, emb_dim=24, lstm_dim=H, fc1_dim=FC1, fc2_dim=FC2), seed)
    g = torch.Generator().manual_seed(seed)
    sd.update({"fc1.weight": torch.randn(FC1, 2 * H, generator=g) / (2 * H) ** 0.5 * 1.5, "fc1.bias": torch.randn(FC1, generator=g) * 0.3,
               "fc2.weight": torch.randn(FC2, FC1, generator=g) / FC1 ** 0.5 * 3.0, "fc2.bias": torch.randn(FC2, generator=g) * 0.5})
    return sd


@pytest.mark.parametrize("B,T,H,FC1,FC2", [
    (3, 301, 400, 600, 601),      # config.json sizes: the <19, 38> instance; 903 rows = 11 workgroups + 23 rows
    (2, 500, 400, 640, 630),      # the <20, 40> instance
    (2, 400, 296, 601, 607),      # FC1 not a multiple of 4: element-wise h1 stores; ragged last tiles
    (7, 45, 32, 44, 53),          # the small instance, the sizes the CPU-sized parity cases use
    (5, 33, 16, 128, 128),        # ... filled to its edge, K1 = 32 = one chunk
    (1, 1000, 40, 100, 17),       # K1 = 80: a ragged last k chunk (16 of 32)
])
def test_fused_head_matches_fp64_on_rounded_operands(B, T, H, FC1, FC2):
    from voicesplit_amd import ops
    sd = _state(H, FC1, FC2, B * 1000 + FC1)
    g = torch.Generator().manual_seed(T)
    lo = torch.randn(B, T, 2 * H, generator=g)
    dims = ops.make_dims(B, T, 53, 24, H, FC1, FC2, math="bf16")
    sdc = {k: v.cuda() for k, v in sd.items()}
    mask, logits = ops.head(sdc, lo.cuda(), dims, want_logits=True)
    _, ref_logits, ref_mask = _head_ref(lo.view(-1, 2 * H), sd["fc1.weight"], sd["fc1.bias"], sd["fc2.weight"], sd["fc2.bias"])
    got = logits.double().cpu().view(-1, FC2)
    assert torch.isfinite(got).all()
    assert ((got - ref_logits).abs().max() / ref_logits.abs().max()).item() < LOGIT_TOL
    assert ((got - ref_logits).abs().mean() / ref_logits.abs().max()).item() < LOGIT_MEAN_TOL
    assert (mask.double().cpu().view(-1, FC2) - ref_mask).abs().max().item() < MASK_TOL
    only_mask = ops.head(sdc, lo.cuda(), dims)
    assert torch.equal(only_mask, mask)


def test_fused_head_agrees_with_the_two_launch_form():
    """Without prepared weights the images are packed into the conv stack's first activation buffer; a clip of a few frames has
    no room for them there and takes the two bf16 GEMM launches.  The same rows inside a longer batch take the fused kernel:
    same roundings, so the two agree to fp32 accumulation order (and the odd h1 element rounded the other way)."""
    from voicesplit_amd import ops
    H, FC1, FC2 = 400, 600, 601
    sd = {k: v.cuda() for k, v in _state(H, FC1, FC2, 5).items()}
    lo = torch.randn(4, 301, 2 * H, generator=torch.Generator().manual_seed(9)).cuda()
    big, big_l = ops.head(sd, lo, ops.make_dims(4, 301, 53, 24, H, FC1, FC2, math="bf16"), want_logits=True)
    few = lo[1:2, :6].contiguous()                          # 6 frames x 53 bins x 64 channels x 4 B = 81 KB < the 2.7 MB of images
    one, one_l = ops.head(sd, few, ops.make_dims(1, 6, 53, 24, H, FC1, FC2, math="bf16"), want_logits=True)
    assert ((big_l[1, :6] - one_l[0]).abs().max() / one_l.abs().max()).item() < LOGIT_TOL
    assert (big[1, :6] - one[0]).abs().max().item() < MASK_TOL


@pytest.mark.parametrize("dims_d,B,T", [(dict(num_freq=53, emb_dim=24, lstm_dim=32, fc1_dim=44, fc2_dim=53), 3, 37), (None, 2, 60)])
def test_train_forward_keeps_h1_for_the_backward_pass(dims_d, B, T):
    """vs_forward_train in VS_MATH_BF16: the fused head stores fc1's activation (fp32, before the bf16 rounding fc2 sees) in the
    tape, where vs_backward reads it."""
    from oracle import reference_forward as R
    from voicesplit_amd import ops
    d = dims_d or R.default_dims()
    sd = R.spread_logits(R.build_state_dict(d, 3), 6.0)
    x, dvec = R.synthetic_inputs(B, T, d, 3)
    sdc = {k: v.cuda() for k, v in sd.items()}
    H = d["lstm_dim"]
    dims = ops.make_dims(B, T, d["num_freq"], d["emb_dim"], H, d["fc1_dim"], d["fc2_dim"], math="bf16")
    tape = ops.new_tape(dims, "cuda")
    mask = ops.forward_train(sdc, x.cuda(), dvec.cuda(), dims, "mish", True, tape)
    lay = ops.tape_layout(dims)
    lo = ops.ws_view(tape, lay.lstm_out, (B * T, 2 * H)).cpu()
    h1 = ops.ws_view(tape, lay.fc1_out, (B * T, d["fc1_dim"])).double().cpu()
    ref_h1, _, ref_mask = _head_ref(lo, sd["fc1.weight"], sd["fc1.bias"], sd["fc2.weight"], sd["fc2.bias"])
    assert ((h1 - ref_h1).abs().max() / ref_h1.abs().max()).item() < 2e-5
    assert (mask.double().cpu().view(-1, d["fc2_dim"]) - ref_mask).abs().max().item() < MASK_TOL
