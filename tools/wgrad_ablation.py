#!/usr/bin/env python
"""Timing ablations of the channels-last weight gradient (csrc/wgrad_nhwc.hip) on an ABLATION=1 build of the library
(make -C voicesplit_amd/csrc ABLATION=1 LIB=../libvoicesplit_hip_abl.so, built out of tree): which of DMA, LDS fragment
reads and MFMAs the launch time follows.  Results of the ablated launches are wrong by construction."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from voicesplit_amd import _lib  # noqa: E402

_lib.load(os.path.join(ROOT, "voicesplit_amd", "libvoicesplit_hip_abl.so"))
from voicesplit_amd import ops  # noqa: E402


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
    B, T, F = 64, 301, 601
    dev = torch.device("cuda:0")
    x = torch.randn(B, T, F, 64, device=dev).to(torch.bfloat16)
    z = torch.zeros_like(x)
    res = {}
    for (KT, KF, dil) in ((5, 5, 1), (5, 5, 4), (7, 1, 1)):
        for abl, name in ((0, "full"), (1, "no DMA"), (2, "no LDS reads"), (4, "no MFMA"), (3, "MFMA only"), (6, "DMA only"), (5, "LDS reads only")):
            _lib.set_option("ABLATION", abl)
            ms = timed(lambda: ops.nhwc_conv_wgrad(x, x, KT, KF, dil))
            res[f"wgrad {KT}x{KF} dil{dil} {name}"] = round(ms, 3)
        _lib.set_option("ABLATION", 0)
        res[f"wgrad {KT}x{KF} dil{dil} full, zero operands"] = round(timed(lambda: ops.nhwc_conv_wgrad(z, z, KT, KF, dil)), 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
