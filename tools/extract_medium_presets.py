#!/usr/bin/env python3
"""pbrt-v4_amd/data/medium_presets.txt: the measured scattering coefficients behind `"string preset"` of the media and `"string name"`
of the subsurface material (GetMediumScatteringProperties, media.cpp:79-150: Jensen et al. 2001, Narasimhan et al. 2006; reduced
scattering sigma'_s and absorption sigma_a in mm^-1, RGB), read out of the reference source the way tools/extract_spectral_tables.py
reads the CIE tables.  One line per medium: name | sigma'_s r g b | sigma_a r g b (the decimal literals as written, which the host
converts double -> float exactly as the reference's RGB(double, double, double) constructor calls do).  Needs /root/reference."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/pbrt/media.cpp").read()
rows = re.findall(r'\{"([^"]+)",\s*RGB\(([^)]*)\),\s*RGB\(([^)]*)\)\}', src)
assert len(rows) >= 40, len(rows)
out = os.path.join(ROOT, "pbrt-v4_amd", "data", "medium_presets.txt")
with open(out, "w") as f:
    f.write("# name | sigma'_s (mm^-1, RGB) | sigma_a (mm^-1, RGB)   (media.cpp:81-140)\n")
    for name, s, a in rows:
        f.write("%s | %s | %s\n" % (name, " ".join(v.strip() for v in s.split(",")), " ".join(v.strip() for v in a.split(","))))
print(out, len(rows))
