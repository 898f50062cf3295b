"""Build-time lint for the toolchain defect of DESIGN.md 4.2 (CPU suite: it only disassembles the built objects).

Every kernel of the shipped build keeps the VGPRs that carry its spilled SGPRs in registers; the three wrong-code incidents of rounds
3-4 were all builds in which such a carrier register was itself spilled to scratch (tools/check_spill_carriers.py).  A source change that
pushes a kernel over that edge fails here, before it reaches the GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_spills_an_sgpr_spill_carrier():
    build = os.path.join(ROOT, "pbrt-v4_amd", "_build")
    assert os.path.exists(os.path.join(build, "wf_backend.o")), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_spill_carriers.py"), build], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "0 of them also spill a carrier register" in p.stdout
