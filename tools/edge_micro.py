#!/usr/bin/env python
"""Launch time and effective HBM rate of the memory-side kernels of the channels-last bf16 path (csrc/nhwc_edge.hip) at BASELINE
size (B = 64, 301 x 601): cnn1 (one-pass forward, input moments, one-pass backward), the BatchNorm apply / backward passes,
cnn8 forward (on z7, BatchNorm + activation of cnn7 on the way in) and backward (dy form, a7 recomputed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from voicesplit_amd import _lib, ops  # noqa: E402

if os.environ.get("VS_MICRO_LIB"):          # a variant build of the library for A/B timing
    _lib.load(os.path.join(ROOT, os.environ["VS_MICRO_LIB"]))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    B, T, F = int(os.environ.get("VS_B", 64)), 301, 601
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.rand(B, T, F, generator=g).to(dev)
    act16 = torch.randn(B, T, F, 64, device=dev).to(torch.bfloat16)
    da16 = torch.randn(B, T, F, 64, device=dev).to(torch.bfloat16)
    one, zero = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    sc = torch.rand(64, device=dev) + 0.5
    sh = torch.randn(64, device=dev) * 0.2
    w1 = torch.randn(64, 1, 1, 7, device=dev) * 0.4
    b1 = torch.randn(64, device=dev) * 0.3
    w8 = torch.randn(8, 64, 1, 1, device=dev) * 0.2
    dz8 = torch.randn(B, T, 8 * F, device=dev)
    big = B * T * F * 64 * 2 / 1e9                       # one channels-last bf16 tensor, GB
    res = {}

    def rec(name, ms, gb):
        res[name] = {"ms": round(ms, 3), "GB": round(gb, 2), "TB/s": round(gb / ms, 2)}

    rec("cnn1 one pass: x -> a1 = mish(BN(conv + bias))", timed(lambda: ops.nhwc_conv_first(x, w1, sc, sh, "mish")), big)
    rec("cnn1 conv only + statistics (round 3's first pass)", timed(lambda: ops.nhwc_conv_first(x, w1, one, b1, "none", stats=True)), big)
    rec("cnn1 input moments", timed(lambda: ops.nhwc_first_moments(x)), B * T * F * 4 / 1e9)
    rec("cnn1 one-pass backward (mish)", timed(lambda: ops.nhwc_first_bwd(da16, x, w1, b1, "mish", True, sc, sh, zero, one)), big)
    rec("BatchNorm + mish apply", timed(lambda: ops.nhwc_bn_apply(act16, sc, sh, "mish")), 2 * big)
    # the same pass without the activation's arithmetic (what the memory system alone allows this access pattern) and with relu
    rec("BatchNorm apply, no activation", timed(lambda: ops.nhwc_bn_apply(act16, sc, sh, "none")), 2 * big)
    rec("BatchNorm + relu apply", timed(lambda: ops.nhwc_bn_apply(act16, sc, sh, "relu")), 2 * big)
    st = torch.zeros(64, 64, 2, dtype=torch.float64, device=dev)
    rec("BatchNorm backward from dy (one pass)", timed(lambda: ops.nhwc_bn_bwd_from_dy(da16, act16, st, True, sc, zero, one)), 3 * big)
    rec("cnn8 forward on z7 (BatchNorm + mish of cnn7 applied on the way in) + statistics",
        timed(lambda: ops.nhwc_conv_last_pre(act16, sc, sh, "mish", w8, torch.ones(8, device=dev), torch.zeros(8, device=dev), stats=True)),
        big + B * T * F * 8 * 4 / 1e9)
    rec("cnn8 backward, dy form, a7 recomputed from z7", timed(lambda: ops.nhwc_conv_last_bwd_dy(dz8, w8, None, act16, "mish", sc, sh, zero, one)),
        2 * big + B * T * F * 8 * 4 / 1e9)
    print(json.dumps(res, indent=1), flush=True)


if __name__ == "__main__":
    main()
