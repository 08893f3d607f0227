"""Times the BiLSTM recurrence (forward, training forward, BPTT) per kernel variant: python tools/lstm_time.py [B]"""
import sys, torch
sys.path.insert(0, ".")
from voicesplit_amd import _lib, ops
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, H = 301, 400
d = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
xg = torch.randn(B, T, 8 * H, generator=g).to(d)
whh = [(torch.randn(4 * H, H, generator=g) * (1.5 / H ** 0.5)).to(d) for _ in range(2)]
dout = torch.randn(B, T, 2 * H, generator=g).to(d)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mode, math, name in ((1, None, "one launch per step"), (2, None, "persistent, fp32 MFMA products"),
                         (4, _lib.MATH_F16X3, "flag hand-off, split-f16 fwd / fp32 BPTT"), (4, _lib.MATH_BF16, "flag hand-off, f16 fwd / bf16 BPTT"),
                         (2, _lib.MATH_F16X3, "tagged fwd, split-f16 fwd / fp32 BPTT"), (2, _lib.MATH_BF16, "tagged fwd, f16 fwd / bf16 BPTT")):
    assert lib.vs_set_lstm_kernel(mode) == 0
    out, gates, c = ops.bilstm_recurrent_train(xg, whh[0], whh[1], math=math)
    f = timeit(lambda: ops.bilstm_recurrent(xg, whh[0], whh[1], math=math))
    ft = timeit(lambda: ops.bilstm_recurrent_train(xg, whh[0], whh[1], math=math))
    def bwd():
        gg = gates.clone()
        return ops.bilstm_recurrent_bwd(gg, c, dout, whh[0], whh[1], math=math)
    clone = timeit(lambda: gates.clone())
    b = timeit(bwd) - clone
    print(f"B={B} {name:36s} forward {f:6.3f} ms ({f / T * 1e3:5.2f} us/step)  train fwd {ft:6.3f}  BPTT {b:6.3f} ms ({b / T * 1e3:5.2f} us/step)")
lib.vs_set_lstm_kernel(0)
