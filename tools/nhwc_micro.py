#!/usr/bin/env python
"""Launch time of the channels-last bf16 64->64 conv (csrc/conv_nhwc.hip) at BASELINE size (B=64, 301x601), per layer
shape, next to the fp32-layout kernels of the same layers in bf16 / f16x3 arithmetic."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from voicesplit_amd import _lib  # noqa: E402
if os.environ.get("VS_MICRO_LIB"):          # an experimental build of the library (tools only: the product loads its own)
    _lib.load(os.path.join(ROOT, "voicesplit_amd", os.environ["VS_MICRO_LIB"]))
from voicesplit_amd import ops  # noqa: E402


def timed(fn, reps=3):
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
    print(json.dumps(run(), indent=1), flush=True)


def run():
    B, T, F = int(os.environ.get("VS_B", 64)), 301, 601
    dev = torch.device("cuda:0")
    x = torch.randn(B, T, F, 64, device=dev).to(torch.bfloat16)
    one, zero = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    res = {}
    for (KT, KF, dil) in ((5, 5, 1), (5, 5, 2), (5, 5, 4), (5, 5, 8), (5, 5, 16), (7, 1, 1)):
        w = torch.randn(64, 64, KT, KF, device=dev) / (64 * KT * KF) ** 0.5
        gflop = 2.0 * B * 64 * 64 * KT * KF * T * F / 1e9
        for act, stats in (("none", False), ("none", True), ("mish", False)):
            ms = timed(lambda: ops.nhwc_conv(x, w, one, zero, dil, act, stats=stats))
            res[f"nhwc {KT}x{KF} dil{dil} act={act} stats={int(stats)}"] = {"ms": round(ms, 3), "tflops": round(gflop / ms, 1)}
        if dil == 4 or KF == 1:      # the same launch on all-zero activations: what the clock does without operand toggling
            xz = torch.zeros_like(x)
            ms = timed(lambda: ops.nhwc_conv(xz, w, one, zero, dil, "none"))
            res[f"nhwc {KT}x{KF} dil{dil} act=none, zero activations"] = {"ms": round(ms, 3), "tflops": round(gflop / ms, 1)}
        if os.environ.get("VS_MICRO_DY", "1") != "0" and dil in (1, 4):
            packed = ops.nhwc_conv_pack(w, transpose_flip=True)
            mean, invstd = zero, one
            for act in ("mish", "relu"):
                ms = timed(lambda: ops.nhwc_conv_dy(x, packed, x, act, one, zero, mean, invstd, KT, KF, dil))
                res[f"nhwc {KT}x{KF} dil{dil} dy epilogue act'={act}"] = {"ms": round(ms, 3), "tflops": round(gflop / ms, 1)}
        if os.environ.get("VS_MICRO_WGRAD", "1") == "0":
            continue
        ms = timed(lambda: ops.nhwc_conv_wgrad(x, x, KT, KF, dil))
        res[f"nhwc wgrad {KT}x{KF} dil{dil}"] = {"ms": round(ms, 3), "tflops": round(gflop / ms, 1)}
    return res


if __name__ == "__main__":
    main()
