"""The counted waits of the LDS-ring pair kernels, re-counted in the ISA of the last build (scripts/check_ring_waits.py).

hipcc does not see the LDS-DMA copies (inline asm), so the `s_waitcnt vmcnt(N)` that orders a chunk's LDS reads behind its
copies is the generator's arithmetic alone: N vector-memory operations must really sit between a chunk's copies and its
evaluation, and hipcc must not have drawn a wait of its own into the loop (spill reloads, late loads: the ring would be
drained every iteration).  No GPU needed: the objects of `python -m nequip_amd.csrc.build` are disassembled."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "nequip_amd", "csrc", "build")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.isdir(BUILD) or not os.path.exists(OBJDUMP), reason="no build objects / no llvm-objdump here")
def test_ring_kernel_waits_hold_in_the_isa():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_ring_waits.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 problem(s)" in r.stdout and " ring kernel instantiations checked" in r.stdout
