"""The wait states between conv_nhwc.hip's assembly-issued MFMAs and the first read of their results, checked on the BUILT library.

The kernels issue their MFMAs from volatile inline assembly with the accumulators in VGPRs: the compiler's hazard recognizer cannot
see them, so the kernel's structure has to provide the wait states.  tools/mfma_hazard_scan.py counts
them in the disassembly of the code objects inside libvoicesplit_hip.so -- what ships, not a second compilation; a toolchain or source
change that moves a read too close fails HERE, on the CPU, not as one wrong value in 3 500 on the GPU."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIB = os.path.join(ROOT, "voicesplit_amd", "libvoicesplit_hip.so")


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="needs ROCm's llvm-objdump")
def test_no_read_of_a_matrix_pipe_result_comes_too_soon(tmp_path):
    assert os.path.exists(LIB), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = shutil.copy(LIB, tmp_path / "lib.so")                      # (--offloading writes the code objects beside its input)
    subprocess.run([OBJDUMP, "--offloading", str(lib)], check=True, capture_output=True, timeout=300)
    found = []
    for co in sorted(glob.glob(str(tmp_path / "lib.so.*gfx950"))):
        dis = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True, timeout=300).stdout
        if "nhwc_conv_kernel" not in dis:
            continue
        listing = tmp_path / "conv.dis"
        listing.write_text(dis)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_hazard_scan.py"), str(listing), "nhwc_conv_kernel"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        found += [l for l in r.stdout.splitlines() if "MFMAs with VGPR results" in l]
    assert len(found) >= 12, found          # every (kernel shape, activation, statistics, dy) instance of the library
    assert all(", 0 reads sooner" in l for l in found), found


def test_the_scanner_reports_a_read_that_comes_too_soon(tmp_path):
    """(the checker checked: four instructions behind an MFMA are 4 wait states, not 12)"""
    bad = tmp_path / "bad.s"
    bad.write_text("_Zbad:                                  ; @_Zbad\n"
                   "\tv_mfma_f32_16x16x32_bf16 v[0:3], a[0:3], v[8:11], 0\n"
                   "\ts_nop 2\n"
                   "\tv_add_f32_e32 v20, v1, v1\n"
                   "\t.section .rodata\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_hazard_scan.py"), str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "1 reads sooner" in r.stdout, r.stdout
    good = tmp_path / "good.s"
    good.write_text(bad.read_text().replace("s_nop 2", "s_nop 15"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_hazard_scan.py"), str(good)], capture_output=True, text=True)
    assert r.returncode == 0 and ", 0 reads sooner" in r.stdout, r.stdout
