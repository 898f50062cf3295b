#!/usr/bin/env python3
"""Prints VGPR / SGPR / scratch / LDS / code size per kernel of pbrt-v4_amd/_build/libwfhip.so (all code objects), or of a .o / .so given as
the argument.  kernel_rows(path) returns the same as a list of dicts (bench.py's `occupancy` block reads it)."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
KEYS = ("name", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "agpr_count", "max_flat_workgroup_size")


def kernel_rows(so):
    rows = []
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run(["objcopy", "-O", "binary", "-j", ".hip_fatbin", so, fat], check=True)
        data = open(fat, "rb").read()
        # the section concatenates one bundle per translation unit
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        offs = [m.start() for m in re.finditer(magic, data)] + [len(data)]
        for i in range(len(offs) - 1):
            b = os.path.join(td, "b%d.bin" % i)
            open(b, "wb").write(data[offs[i]:offs[i + 1]])
            co = os.path.join(td, "b%d.co" % i)
            r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + b, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode: continue
            notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            syms = subprocess.run([LLVM + "llvm-readelf", "-s", "-W", co], capture_output=True, text=True).stdout
            size = {}
            for l in syms.splitlines():
                f = l.split()
                if len(f) >= 8 and f[3] == "FUNC": size[f[7]] = int(f[2])
            unit = []
            cur = {}
            for l in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", l)
                if not m: continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "agpr_count" and cur.get("name"): unit.append(cur); cur = {}
                if k in KEYS: cur[k] = v
            if cur.get("name"): unit.append(cur)
            for r_ in unit: r_.setdefault("code", size.get(r_.get("name"), 0))
            rows += unit
    names = [r_["name"] for r_ in rows]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
    for r_, d in zip(rows, dem):
        r_["demangled"] = re.sub(r"\(.*", "", d.strip())
    return rows


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "pbrt-v4_amd", "_build", "libwfhip.so")
    print("%-72s %5s %5s %5s %7s %7s %6s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "spill", "code B"))
    for r_ in sorted(kernel_rows(so), key=lambda r: r.get("name", "")):
        print("%-72s %5s %5s %5s %7s %7s %6s %8s" % (r_["demangled"][-72:], r_.get("vgpr_count"), r_.get("agpr_count"), r_.get("sgpr_count"), r_.get("private_segment_fixed_size"),
                                               r_.get("group_segment_fixed_size"), r_.get("vgpr_spill_count"), r_.get("code")))
