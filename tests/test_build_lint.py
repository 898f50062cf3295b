"""Build-time lint for the toolchain defect of DESIGN.md 4.6 (CPU suite: it only disassembles the built objects).

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


def test_scratch_allocations_cover_call_chains_and_lane_saves_are_whole_wave():
    """tools/stack_audit.py: no kernel's deepest direct call chain needs more scratch than the kernel is allocated, and no kernel can reach a
    recursive device function (which would make it a dynamically-sized-stack kernel: Log2IntF was one until round 5).
    tools/carrier_audit.py: every save / reload of a VGPR whose lanes hold spilled SGPRs is done with all lanes enabled."""
    build = os.path.join(ROOT, "pbrt-v4_amd", "_build")
    assert os.path.exists(os.path.join(build, "wf_backend.o")), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", t)] + a, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for t, a in (("stack_audit.py", [build]), ("carrier_audit.py", [build, "-q"]))]
    outs = [p.communicate(timeout=1200)[0] for p in procs]
    assert procs[0].returncode == 0 and ": 0 whose deepest direct call chain needs more scratch" in outs[0], outs[0][-3000:]
    assert "recursion" not in outs[0], outs[0][-3000:]
    assert procs[1].returncode == 0 and "\n0 save / reload of live lanes under a partial EXEC" in outs[1], outs[1][-3000:]
