#!/usr/bin/env python
"""Wall-clock of one training step of the hot path on one MI355X (fwd with tape, backward),
B=64 synthetic [64,301,601] + [64,256], fp32.  Prints one JSON line; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel breakdown."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import voicesplit_amd as V  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--eval-bn", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = V.VoiceSplit(V.default_config()).to(dev)
    m.train(not args.eval_bn)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(args.batch, 301, 601, generator=g).to(dev)
    d = torch.randn(args.batch, 256, generator=g)
    d = (d / d.norm(dim=1, keepdim=True)).to(dev)
    w = torch.randn(args.batch, 301, 601, generator=g).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fwd = bwd = 0.0
    for it in range(args.warmup + args.steps):
        m.zero_grad(set_to_none=True)
        ev[0].record()
        mask = m(x, d)
        ev[1].record()
        mask.backward(w)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= args.warmup:
            fwd += ev[0].elapsed_time(ev[1])
            bwd += ev[1].elapsed_time(ev[2])
    # host-side view of one bench.py-style step: wall clock with a synchronize after each phase
    from voicesplit_amd.sharding import GradientBucket
    bucket = GradientBucket(m.parameters()).attach()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    phases = {}

    def lap(name, t0):
        torch.cuda.synchronize()
        phases[name] = phases.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()

    for it in range(3):
        if it == 1:
            phases = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bucket.zero(); t0 = lap("bucket_zero", t0)
        mask = m(x, d); t0 = lap("forward", t0)
        mask.backward(w); t0 = lap("backward", t0)
        bucket.all_reduce(1); t0 = lap("bucket_allreduce", t0)
        opt.step(); t0 = lap("adam", t0)
    phases = {k: round(v / 2, 2) for k, v in phases.items()}
    t0 = time.perf_counter()
    for it in range(2):
        bucket.zero()
        mask = m(x, d)
        mask.backward(w)
        bucket.all_reduce(1)
        opt.step()
    torch.cuda.synchronize()
    phases["unsynced_step_wall"] = round((time.perf_counter() - t0) * 1e3 / 2, 2)
    n = args.steps
    ok = all(torch.isfinite(p.grad).all().item() for p in m.parameters())
    print(json.dumps({"batch": args.batch, "fwd_train_ms": round(fwd / n, 3), "bwd_ms": round(bwd / n, 3),
                      "utt_per_s_fwd_bwd": round(args.batch / ((fwd + bwd) / n / 1e3), 2), "finite": ok,
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2), "host_phases_ms": phases}), flush=True)


if __name__ == "__main__":
    main()
