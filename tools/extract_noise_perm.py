#!/usr/bin/env python3
"""pbrt-v4_amd/data/noise_perm.txt: Ken Perlin's 256-entry permutation (doubled to 512), read out of the reference's
util/noise.cpp the same way tools/extract_spectral_tables.py reads the CIE tables — a constant table, loaded by the host at
scene-build time and uploaded with the scene (wf_scene_desc.noise_perm).  Needs /root/reference."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/pbrt/util/noise.cpp").read()
m = re.search(r"NoisePerm\[2 \* NoisePermSize\] = \{(.*?)\};", src, re.S)
body = re.sub(r"//[^\n]*", "", m.group(1))
vals = [int(v) for v in re.findall(r"\d+", body)]
assert len(vals) == 512 and vals[:256] == vals[256:] and sorted(vals[:256]) == list(range(256))
out = os.path.join(ROOT, "pbrt-v4_amd", "data", "noise_perm.txt")
with open(out, "w") as f:
    f.write("# Perlin noise permutation, 2 x 256 entries (util/noise.cpp:19-56)\n")
    for i in range(0, 512, 32):
        f.write(" ".join(str(v) for v in vals[i:i + 32]) + "\n")
print(out, len(vals))
