#!/usr/bin/env python
"""Micro-benchmark of the three large LSTM contractions at BASELINE size (B=64: M = 19264 rows)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from voicesplit_amd import _lib, ops  # noqa: E402

if os.environ.get("VS_MICRO_LIB"):          # e.g. an ablation build (voicesplit_amd/libvoicesplit_hip_abl.so)
    _lib.load(os.path.join(ROOT, os.environ["VS_MICRO_LIB"]))


def main():
    dev = torch.device("cuda:0")
    M, H4, K8, KE = 64 * 301, 1600, 4808, 5064
    feat = torch.randn(M, K8, device=dev)
    w = torch.randn(2 * H4, KE, device=dev) * 0.01
    dxg = torch.randn(M, 2 * H4, device=dev) * 1e-3
    res = {}
    cases = {
        "xg = feat @ W_ih^T (NT, N=3200)": lambda math: ops.gemm(feat, w, M, 2 * H4, K8, 0, 0, math=math),
        "dfeat = dxg_d @ W_ih (NN, K=1600)": lambda math: ops.gemm(dxg, w, M, K8, H4, 0, 1, math=math),
        "dW_ih = dxg_d^T @ feat (TN, K=19264)": lambda math: ops.gemm(dxg, feat, H4, K8, M, 1, 1, math=math),
    }
    flops = {"xg": 2.0 * M * 2 * H4 * K8, "dfeat": 2.0 * M * K8 * H4, "dW_ih": 2.0 * H4 * K8 * M}
    only = os.environ.get("VS_MICRO_ONLY")          # e.g. "xg:f16x3"
    for name, fn in cases.items():
        for math in ("fp32", "f16x3"):
            if only and only != f"{name.split()[0]}:{math}":
                continue
            fn(math)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn(math)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            res[f"{name} [{math}]"] = {"ms": round(ms, 3), "tflops": round(flops[name.split()[0]] / ms / 1e9, 1)}
    # the bf16 configuration's own GEMM (csrc/gemm_bf16.hip) on the same three contractions
    Kp = (K8 + 63) // 64 * 64
    feat_bf = ops.cvt_rows_bf16(feat, K8, Kp)
    wih_bf = ops.cvt_rows_bf16(w[:, :K8].contiguous(), K8, Kp)
    dxg_bf = ops.cvt_rows_bf16(dxg, 2 * H4, 2 * H4)
    bf_cases = {
        "xg (row x row) [bf16 gemm]": (lambda: ops.gemm_bf16(feat_bf, wih_bf, M, 2 * H4, K8), 2.0 * M * 2 * H4 * K8),
        "dfeat (row x col, K=3200) [bf16 gemm]": (lambda: ops.gemm_bf16(dxg_bf, wih_bf, M, K8, 2 * H4, b_kmajor=True), 2.0 * M * K8 * 2 * H4),
        "dW_ih (col x col, K=19264) [bf16 gemm]": (lambda: ops.gemm_bf16(dxg_bf, feat_bf, 2 * H4, K8, M, a_kmajor=True, b_kmajor=True), 2.0 * 2 * H4 * K8 * M),
        "cvt feat -> bf16": (lambda: ops.cvt_rows_bf16(feat, K8, Kp), 0.0),
    }
    for name, (fn, fl) in bf_cases.items():
        if only and only not in ("bf16", name.split()[0]):
            continue
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res[name] = {"ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1)}
    print(json.dumps(res, indent=1), flush=True)


if __name__ == "__main__":
    main()
