#!/usr/bin/env python
"""Micro-benchmark of the 64->64 weight-gradient kernels at BASELINE size (B=64, 301x601)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from voicesplit_amd import _lib, ops  # noqa: E402


def main():
    B, T, F = int(os.environ.get("VS_MICRO_B", "64")), 301, 601
    reps = int(os.environ.get("VS_MICRO_REPS", "3"))
    lib = _lib.load()
    lib.vs_set_wgrad_kernel(int(os.environ.get("VS_MICRO_WGRAD_KERNEL", "0")))     # 0 auto, 1 ring, 2 kt-split
    dev = torch.device("cuda:0")
    x = torch.randn(B, 64, T, F, device=dev)
    dz = torch.randn(B, 64, T, F, device=dev) * 1e-3
    res = {"B": B}
    for (KT, KF, dil) in [(5, 5, 1), (5, 5, 4), (5, 5, 16), (7, 1, 1)]:
        part = torch.empty(lib.vs_conv64_wgrad_partial_floats(KT, KF), dtype=torch.float32, device=dev)
        dw = torch.empty(64, 64, KT, KF, device=dev)
        scratch = torch.zeros(8, device=dev)
        for math in ("fp32", "f16x3"):
            def run():
                if math == "fp32":
                    _lib.check(lib.vs_conv64_wgrad(ops._p(dz), ops._p(x), ops._p(part), ops._p(dw), B, T, F, KT, KF, dil, ops._stream()), "wgrad")
                else:
                    _lib.check(lib.vs_conv64_wgrad_f16x3(ops._p(dz), ops._p(x), ops._p(part), ops._p(dw), ops._p(scratch), B, T, F, KT, KF, dil, ops._stream()), "wgrad16")
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            res[f"{KT}x{KF}_d{dil}_{math}"] = {"ms": round(ms, 3), "tflops": round(2.0 * 64 * 64 * KT * KF * T * F * B / ms / 1e9, 1)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
