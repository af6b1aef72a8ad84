"""CPU: the three-term bf16 kernels issue their MFMAs as asm statements, so the compiler pads no wait states for them.  tools/check_asm_mfma_hazards.py
reads the gfx950 ISA of those files and fails when a VALU write sits right in front of an MFMA that reads it, or a reader right behind an MFMA result
(the scheduling bug the LN form of the row-GEMM had before its waits were tied to the operand registers)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc to emit the ISA")
def test_no_unpadded_hazard_around_asm_mfmas():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_mfma_hazards.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MFMAs checked" in r.stdout
