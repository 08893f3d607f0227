#!/usr/bin/env python
"""Round 4: where does the bf16 GEMM's time go beside its MFMAs?  Same kernel, same shapes, operands varied:
random vs all-zero operands (the chip's clock follows the operand toggle rate), with / without the row bias, K doubled
(T(2K) - T(K) = the K loop of one K; 2 T(K) - T(2K) = everything that does not scale with K: epilogue, cold start, tail)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from voicesplit_amd import ops  # noqa: E402


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n, 3)


def main():
    dev = torch.device("cuda:0")
    M, N, K = 64 * 301, 3200, 4808
    res = {}
    for fill in ("randn", "zeros"):
        mk = (lambda *s: torch.randn(*s, device=dev)) if fill == "randn" else (lambda *s: torch.zeros(*s, device=dev))
        for kk in (K, 2 * K):
            Kp = (kk + 63) // 64 * 64
            feat = ops.cvt_rows_bf16(mk(M, kk), kk, Kp)
            w = ops.cvt_rows_bf16(mk(N, kk) * 0.01, kk, Kp)
            rb = mk(64, N)
            out = torch.empty(M, N, device=dev)
            res[f"xg row x row K={kk} {fill}"] = timeit(lambda: ops.gemm_bf16(feat, w, M, N, kk, out=out))
            res[f"xg row x row K={kk} {fill} + rowbias"] = timeit(lambda: ops.gemm_bf16(feat, w, M, N, kk, rowbias=rb, group=301, out=out))
            del feat, w
        dxg = ops.cvt_rows_bf16(mk(M, N) * 1e-3, N, N)
        feat = ops.cvt_rows_bf16(mk(M, K), K, 4864)
        wih = ops.cvt_rows_bf16(mk(N, K) * 0.01, K, 4864)
        o2 = torch.empty(M, K, device=dev)
        res[f"dfeat row x col {fill}"] = timeit(lambda: ops.gemm_bf16(dxg, wih, M, K, N, b_kmajor=True, out=o2))
        o3 = torch.empty(N, K, device=dev)
        res[f"dW_ih col x col {fill}"] = timeit(lambda: ops.gemm_bf16(dxg, feat, N, K, M, a_kmajor=True, b_kmajor=True, out=o3))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
