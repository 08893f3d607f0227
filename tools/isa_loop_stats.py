#!/usr/bin/env python
"""Static instruction mix of the hottest basic block (the one with the most MFMAs) of every kernel in a gfx950 assembly
listing -- the budget DESIGN.md 6.0 says matters for a one-wave-per-SIMD kernel (one issue per ~4 cycles, any class).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s voicesplit_amd/csrc/gemm_bf16.hip
    python tools/isa_loop_stats.py /tmp/k.s [kernel-name-substring]
"""
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):\s*; @", text, flags=re.M)]
    for i, (pos, name) in enumerate(starts):
        if want not in name:
            continue
        end = text.find(".section", pos)
        body = text[pos:end if end > 0 else None]
        blocks, cur = {}, None
        for l in body.split("\n"):
            l = l.strip()
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if m:
                cur = m.group(1)
                blocks[cur] = []
                continue
            if cur and l and not l.startswith((".", ";")):
                blocks[cur].append(l.split(";")[0].strip())
        if not blocks:
            continue
        lab, b = max(blocks.items(), key=lambda kv: sum("v_mfma" in x for x in kv[1]))

        def cnt(pred):
            return sum(1 for x in b if pred(x))
        print(name[-44:], lab, "instr", len(b), "mfma", cnt(lambda x: "v_mfma" in x),
              "salu", cnt(lambda x: x.startswith("s_") and not x.startswith(("s_waitcnt", "s_nop", "s_barrier"))),
              "valu", cnt(lambda x: x.startswith("v_") and "mfma" not in x and "accvgpr" not in x),
              "accvgpr", cnt(lambda x: "accvgpr" in x), "ds", cnt(lambda x: x.startswith("ds_")),
              "vmem", cnt(lambda x: x.startswith(("buffer_", "global_"))), "scratch", cnt(lambda x: x.startswith("scratch_")),
              "waitcnt", cnt(lambda x: x.startswith("s_waitcnt")), "nop", cnt(lambda x: x.startswith("s_nop")))


if __name__ == "__main__":
    main()
