#!/usr/bin/env python
"""Micro-benchmark of the 64->64 conv kernels (fp32 MFMA and split-f16) at BASELINE size
(B=64, 301x601).  Prints one JSON line."""
import ctypes
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
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 64, T, F, device=dev)
    out = torch.empty_like(x)
    scale = (torch.rand(64, generator=g) + 0.5).to(dev)
    shift = (torch.randn(64, generator=g) * 0.1).to(dev)
    res = {"B": B}
    only = os.environ.get("VS_MICRO_ONLY")            # e.g. "7x1": that kernel shape, split-f16 kernel only
    for (KT, KF, dil) in [(5, 5, 1), (5, 5, 2), (5, 5, 4), (5, 5, 8), (5, 5, 16), (7, 1, 1)]:
        if only and only != f"{KT}x{KF}":
            continue
        w = (torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5).to(dev)
        packed = torch.empty(lib.vs_conv64_packed_floats(KT, KF), dtype=torch.float32, device=dev)
        _lib.check(lib.vs_conv64_pack(ops._p(w), ops._p(packed), KT, KF, ops._stream()), "pack")
        packed16 = torch.empty(lib.vs_conv64_packed_f16_floats(KT, KF), dtype=torch.float32, device=dev)
        scales = torch.zeros(8, dtype=torch.float32, device=dev)
        amax = scales[4:].view(torch.int32)
        _lib.check(lib.vs_pow2_scale(ops._p(x), x.numel(), ops._p(amax), ops._p(scales), ops._stream()), "scale")
        _lib.check(lib.vs_conv64_pack_f16(ops._p(w), ops._p(packed16), KT, KF, 0, ops._p(amax[1:]), ops._p(scales[2:]), ops._stream()), "pack16")
        for act in (("mish_f16x3",) if only else ("mish", "none", "mish_f16x3", "scale_pass")):
            def run():
                if act == "scale_pass":
                    _lib.check(lib.vs_pow2_scale(ops._p(x), x.numel(), ops._p(amax), ops._p(scales), ops._stream()), "scale")
                elif act == "mish_f16x3":
                    _lib.check(lib.vs_conv64_f16x3_fwd(ops._p(x), ops._p(packed16), ops._p(scale), ops._p(shift), ops._p(scales),
                                                       ops._p(scales[2:]), ops._p(out), B, T, F, KT, KF, dil,
                                                       ops.ACT_CODES["mish"], ops._stream()), "conv16")
                else:
                    _lib.check(lib.vs_conv64_fwd(ops._p(x), ops._p(packed), ops._p(scale), ops._p(shift), ops._p(out),
                                                 B, T, F, KT, KF, dil, ops.ACT_CODES[act], ops._stream()), "conv")
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tf = 2.0 * 64 * 64 * KT * KF * T * F * B / ms / 1e9
            res[f"{KT}x{KF}_d{dil}_{act}"] = {"ms": round(ms, 3), "tflops": round(tf, 1)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
