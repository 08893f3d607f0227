"""Eval-mode forward latency, weights re-derived per call (vs_forward) vs prepared once (vs_forward_prepared):
python tools/eval_latency.py"""
import sys, torch
sys.path.insert(0, ".")
import voicesplit_amd as V
from voicesplit_amd import ops
m = V.VoiceSplit(V.default_config(601, 256, 400, 600, 601)).cuda().eval()
sd = {k: v.detach() for k, v in m._tensors().items()}
def timeit(fn, n):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B, T, n in ((1, 301, 50), (1, 1000, 20), (8, 301, 20), (64, 301, 5)):
    x = torch.rand(B, T, 601, device="cuda"); d = torch.randn(B, 256, device="cuda")
    dims = m._dims(B, T)
    prep = ops.PreparedWeights(sd, dims)
    a = timeit(lambda: ops.forward(sd, x, d, dims, "mish", training=False), n)
    b = timeit(lambda: ops.forward_prepared(sd, prep, x, d, dims, "mish"), n)
    with torch.no_grad():
        c = timeit(lambda: m(x, d), n)
    print(f"B={B:3d} T={T:5d}: vs_forward {a:7.3f} ms   vs_forward_prepared {b:7.3f} ms   module call (cache check + prepared) {c:7.3f} ms")
