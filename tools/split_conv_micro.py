"""(VOICESPLIT_ABLATION=N selects a timing ablation / the in-kernel probes: needs a library built with
`make -C voicesplit_amd/csrc ABLATION=1`; the production build ignores it.)
Times vs_nhwc_conv_f16x3_layer at the metric configuration's layer shapes (B = 64, 301 x 601): the gate of the channels-last
split-f16 forward (the NCHW kernel it replaces: 6.3-6.5 ms per 5x5 layer, 3.4 ms for the 7x1).  Writes gpurun_out/<name>.json."""
import json
import sys

import torch

from voicesplit_amd import ops


def main(name):
    B, T, Fq = 64, 301, 601
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, Fq, 64, generator=g).cuda()
    hi, lo, s2 = ops.f16x3_split(x, 2.0 ** 8)
    del x
    amax_in = torch.tensor([4.0], dtype=torch.float32).cuda().view(torch.int32)
    sc, sh = torch.ones(64).cuda(), torch.zeros(64).cuda()
    out = {}
    import os
    from voicesplit_amd import _lib
    shapes = ((5, 5, 1), (5, 5, 4), (5, 5, 16), (7, 1, 1)) if not os.environ.get("VOICESPLIT_ABLATION") else ((5, 5, 1), (7, 1, 1))
    for kt, kf, dil in shapes:
        w = (torch.randn(64, 64, kt, kf, generator=g) / (64 * kt * kf) ** 0.5).cuda()
        r = ops.nhwc_conv_f16x3(hi, lo, s2, w, sc, sh, dil, "mish", amax_in=amax_in)
        scr = r[4]
        del r
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            r = ops.nhwc_conv_f16x3(hi, lo, s2, w, sc, sh, dil, "mish", amax_in=amax_in, scratch=scr)
            del r
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 5
        fl = 2.0 * 64 * 64 * kt * kf * B * T * Fq
        out[f"{kt}x{kf}_dil{dil}"] = {"ms": round(ms, 3), "tflops_fp32_equiv": round(fl / ms / 1e9, 1)}
        if int(os.environ.get("VOICESPLIT_ABLATION", "0")) & 32:
            r = ops.nhwc_conv_f16x3(hi, lo, s2, w, sc, sh, dil, "mish", amax_in=amax_in, scratch=scr)
            probe = r[3].view(torch.int64)[:4].tolist()          # summed over the launch's waves (1024 at B = 64)
            waves = 1024
            out[f"{kt}x{kf}_dil{dil}"]["probe_cycles_per_wave"] = {"total": probe[0] / waves, "vmcnt_wait": probe[1] / waves,
                                                                     "lgkm_barrier_wait": probe[2] / waves, "groups": probe[3] / waves}
            print(out[f"{kt}x{kf}_dil{dil}"], flush=True)
        print(kt, kf, dil, ms, flush=True)
    json.dump(out, open(f"gpurun_out/{name}.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "split_conv_micro")
