#!/usr/bin/env python
"""Static check of a gfx950 assembly listing for reads of a matrix-pipe result that come too soon.

conv_nhwc.hip's round-6 kernels issue their MFMAs from volatile inline assembly (so that the compiler cannot move them relative to
the epilogue's micro-ops) with the accumulators in VGPRs.  The compiler cannot see an MFMA inside an assembly statement, so it
inserts none of the wait states the hardware wants between a matrix-pipe write of a VGPR and a VALU / store read of it
(LLVM's GCNHazardRecognizer: 7 wait states behind a 4-pass instruction, 11 behind an 8-pass one); the kernel's structure provides
them -- >= 4 MFMAs between the last write of a row's accumulators and the first micro-op that reads them, an explicit s_nop in
front of a group's last row -- and THIS script verifies it on what the compiler actually emitted:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude -o /tmp/conv_nhwc.s voicesplit_amd/csrc/conv_nhwc.hip
    python tools/mfma_hazard_scan.py /tmp/conv_nhwc.s [kernel-name-substring]

or, on what the build produced (tests/test_hazard_scan_cpu.py does this: seconds instead of a second compilation):

    llvm-objdump --offloading conv_nhwc.o        # writes conv_nhwc.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 beside its input
    llvm-objdump -d <that file> > /tmp/conv_nhwc.dis && python tools/mfma_hazard_scan.py /tmp/conv_nhwc.dis

For every v_mfma whose destination is a VGPR tuple it follows the instruction stream (fall-through order; the hot blocks are straight
line) and counts wait states -- 4 per MFMA (it holds the issue port for its passes), N + 1 per s_nop N, 1 per other instruction --
until the first non-MFMA instruction that reads one of the registers.  Fewer than MIN_WAIT (12) is reported; exit code 1 then."""
import re
import sys

MIN_WAIT = 12


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(name, lines):
    ins = []
    for l in lines:
        l = l.strip()
        if not l or l.startswith((".", ";", "//")) or re.match(r"\.?LBB", l) or re.match(r"[0-9a-f]+ <", l):
            continue
        l = l.split("//")[0].split(";")[0].strip()      # (assembly listings comment with ';', llvm-objdump with '//')
        if l:
            ins.append(l)
    pending = {}                                   # vgpr -> wait states since the MFMA that wrote it
    bad = []
    n_mfma = 0
    for i, x in enumerate(ins):
        parts = x.replace(",", " ").split()
        op, ops = parts[0], parts[1:]
        if op.startswith("v_mfma"):
            for r in pending:
                pending[r] += 4
            d = regs(ops[0])
            if d:
                n_mfma += 1
            for r in d:
                pending[r] = 0
            continue
        srcs = set()
        if op.startswith("v_"):
            for t in ops[1:]:
                srcs |= regs(t)
        elif op.startswith(("buffer_store", "global_store", "ds_write", "ds_store")):
            for t in ops:
                srcs |= regs(t)
        for r in srcs:
            if r in pending and pending[r] < MIN_WAIT:
                bad.append((pending[r], i, x))
        step = 1
        if op == "s_nop":
            step = int(ops[0], 0) + 1
        for r in list(pending):
            pending[r] += step
            if pending[r] > 64:
                del pending[r]
        if op.startswith("v_") and ops:                 # overwritten by a VALU result: no longer a matrix-pipe value
            for r in regs(ops[0]):
                pending.pop(r, None)
        if op.startswith(("ds_read", "buffer_load", "global_load")) and ops:
            for r in regs(ops[0]):
                pending.pop(r, None)
    return n_mfma, bad


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    objdump = "Disassembly of section" in text      # llvm-objdump -d of the code object instead of the compiler's -S listing
    if objdump:
        starts = [(m.start(), m.group(1)) for m in re.finditer(r"^[0-9a-f]+ <(_Z\w+)>:", text, flags=re.M)]
    else:
        starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):\s*; @", text, flags=re.M)]
    rc = 0
    for pos, name in starts:
        if want not in name:
            continue
        end = min([p for p, _ in starts if p > pos], default=-1) if objdump else text.find(".section", pos)
        n, bad = scan(name, text[pos:end if end > 0 else None].split("\n"))
        if n == 0:
            continue
        print(f"{name[-70:]}: {n} MFMAs with VGPR results, {len(bad)} reads sooner than {MIN_WAIT} wait states")
        for b in bad[:8]:
            print("    wait states %d at instruction %d: %s" % b)
        rc |= 1 if bad else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
