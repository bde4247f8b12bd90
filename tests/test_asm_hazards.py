"""CPU (cross-compile only): the inline-asm carry producers of cuhe_amd/csrc/modp.cuh keep to the patterns that need no
software wait states on gfx950 -- tools/asm_hazard_check.py compiles the device code to assembly and inspects every
asm site (about 13 000).  A VALU instruction must not read an SGPR pair written by the VALU instruction just before it:
LLVM pads its own code for that, nothing pads an asm statement."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_sgpr_read_after_valu_write_around_inline_asm():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_hazard_check.py")], capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:]
    assert "findings: 0" in r.stdout
