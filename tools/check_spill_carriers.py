#!/usr/bin/env python3
"""tools/check_spill_carriers.py [build dir] — a build-time lint for the toolchain defect of DESIGN.md 4.6.

ROCm 7.2's compiler spills SGPRs through the lanes of "carrier" VGPRs (v_writelane_b32 / v_readlane_b32).  When a kernel is so short of
VGPRs that a carrier is ITSELF spilled to scratch and reloaded (whole-wave mode), this tree has seen wrong code three times (a diffuse
material kernel at 168 VGPRs: a deterministic memory fault, and two nondeterministic renders); every kernel that has been correct keeps its
carriers in registers.  This script disassembles the gfx950 code object of every unit under the build directory (default
pbrt-v4_amd/_build) and lists, per kernel / device function, the carrier VGPRs and how often each is stored to / loaded from scratch.

Exit status 1 when a function spills a carrier, unless it is named in ALLOWED below (empty at the end of round 4).  `make -C pbrt-v4_amd` does not run it; tests/test_build_lint.py does (CPU suite), so that a change that pushes a
kernel over the edge is seen before it reaches the GPU.
"""
import collections
import glob
import os
import re
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernels that are known to spill a carrier and are accepted (regular expressions on the demangled name)
ALLOWED = []   # (round 4: the curve kernels, GEN = 3, were here until their occupancy target was lowered to 4 waves)


def code_object(obj, out):
    d = open(obj, "rb").read()
    i = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if i < 0:
        return False
    n = struct.unpack_from("<Q", d, i + 24)[0]
    p = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", d, p)
        p += 24
        triple = d[p:p + tl].decode()
        p += tl
        if "gfx950" in triple:
            open(out, "wb").write(d[i + off:i + off + size])
            return True
    return False


def scan(co):
    notes = subprocess.run([OBJDUMP.replace("llvm-objdump", "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    kernels = set(re.findall(r"\.name:\s+(\S+)", notes))
    txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    funcs, cur = collections.OrderedDict(), None
    for l in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", l)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is not None and l.startswith("\t") or (cur is not None and re.match(r"^\s+[sv]_|^\s+scratch_|^\s+global_|^\s+flat_|^\s+ds_|^\s+buffer_", l)):
            funcs[cur].append(l.strip())
    out = []
    for name, ins in funcs.items():
        carriers = collections.Counter()
        for l in ins:
            m = re.match(r"v_writelane_b32 (v\d+),", l)
            if m:
                carriers[m.group(1)] += 1
        if not carriers:
            continue
        spilled = {}
        for v in carriers:
            st = sum(1 for l in ins if re.match(r"scratch_store_dword\w* off, %s\b" % v, l))
            ld = sum(1 for l in ins if re.match(r"scratch_load_dword\w* %s\b" % v, l))
            # a device FUNCTION saves a callee-saved carrier once in its prologue and restores it once in its epilogue: that is the calling
            # convention, not a spill; a kernel has no such pair
            if (st or ld) and (name in kernels or st > 1 or ld > 1):
                spilled[v] = (st, ld)
        out.append((name, dict(carriers), spilled))
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [re.sub(r"\(.*", "", s).replace("void ", "") for s in p.stdout.split("\n")]


def main():
    build = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pbrt-v4_amd", "_build")
    bad, total = [], 0
    for obj in sorted(glob.glob(os.path.join(build, "*.o"))):
        co = "/tmp/_carriers_%d.co" % os.getpid()
        if not code_object(obj, co):
            continue
        rows = scan(co)
        os.remove(co)
        names = demangle([r[0] for r in rows])
        for (mangled, carriers, spilled), name in zip(rows, names):
            total += 1
            if spilled:
                ok = any(re.search(a, name) for a in ALLOWED)
                print("%-18s %-46s carriers %s  SPILLED (stores, loads): %s%s" % (os.path.basename(obj), name[:46], len(carriers), spilled, "  [allowed]" if ok else ""))
                if not ok:
                    bad.append((obj, name))
    print("%d functions spill SGPRs through VGPR lanes; %d of them also spill a carrier register%s" % (total, len(bad), "" if not bad else " and are not on the allowed list"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
