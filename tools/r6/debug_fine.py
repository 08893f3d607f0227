"""Where do the fine-epilogue (VS_OPT_CONV_EPILOGUE = 1) conv outputs differ from round 3's epilogue?  (round 6 debugging)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
EPX = int(os.environ.get("VS_DEBUG_EP", "1"))
from voicesplit_amd import _lib, ops

def run(B, T, Fq, KT, KF, dil, act, stats, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, Fq, 64, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(64, 64, KT, KF, generator=g) / (64 * KT * KF) ** 0.5).cuda()
    scale = (torch.rand(64, generator=g) + 0.5).cuda()
    shift = (torch.randn(64, generator=g) * 0.3).cuda()
    outs = {}
    for ep in (0, EPX, EPX, EPX):
        _lib.set_option("CONV_EPILOGUE", ep)
        o = ops.nhwc_conv(x, w, scale, shift, dil, act, stats=stats)
        if stats:
            o = o[0]
        torch.cuda.synchronize()
        outs.setdefault(ep, []).append(o.float().cpu())
    _lib.set_option("CONV_EPILOGUE", 0)
    ref = outs[0][0]
    for k, o in enumerate(outs[EPX]):
        d = (o - ref).abs()
        bad = d > (2.0 ** -7 * ref.abs() + 1e-3 * ref.abs().max())
        n = int(bad.sum())
        msg = f"{B}x{T}x{Fq} {KT}x{KF} dil{dil} act={act} stats={stats} run{k}: {n} bad, max diff {d.max():.3e}"
        if n:
            idx = bad.nonzero()
            msg += f" | b {sorted(set(idx[:,0].tolist()))} t {sorted(set(idx[:,1].tolist()))[:20]} f {sorted(set(idx[:,2].tolist()))[:40]} c {sorted(set(idx[:,3].tolist()))[:64]}"
            msg += f" | first {idx[:6].tolist()} got {o[bad][:6].tolist()} want {ref[bad][:6].tolist()}"
        print(msg, flush=True)
    same = all(torch.equal(outs[EPX][0], o) for o in outs[EPX][1:])
    print("   fine runs identical to each other:", same, flush=True)

for act in ("mish", "relu", "none"):
    for (B, T, Fq, KT, KF, dil) in [(2, 40, 37, 5, 5, 1), (1, 23, 70, 5, 5, 2), (2, 19, 33, 5, 5, 4), (1, 50, 20, 5, 5, 16), (1, 100, 64, 5, 5, 1), (2, 40, 37, 7, 1, 1), (4, 301, 601, 5, 5, 1)]:
        run(B, T, Fq, KT, KF, dil, act, False, 1000 * B + T)
run(2, 40, 37, 5, 5, 1, "none", True, 5)
