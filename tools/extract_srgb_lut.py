#!/usr/bin/env python3
"""pbrt-v4_amd/data/srgb_to_linear_lut.txt: the 256-entry table 8-bit sRGB texels are decoded with (SRGB8ToLinear,
util/color.h:536; the literals of util/color.cpp:281 — they are not reproduced by the SRGBToLinear polynomial, so the table
is data, like the CIE tables).  The host reads each literal as a double and narrows it to float, as the compiler does for
`Float x = 0.0003035270;`.  Needs /root/reference."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/pbrt/util/color.cpp").read()
m = re.search(r"SRGBToLinearLUT\[256\] = \{(.*?)\};", src, re.S)
vals = re.findall(r"\d+\.\d+", m.group(1))
assert len(vals) == 256 and float(vals[0]) == 0 and float(vals[255]) == 1
out = os.path.join(ROOT, "pbrt-v4_amd", "data", "srgb_to_linear_lut.txt")
with open(out, "w") as f:
    f.write("# SRGBToLinearLUT, 256 entries (util/color.cpp:281-325): decimal literals, double -> float\n")
    for i in range(0, 256, 8):
        f.write(" ".join(vals[i:i + 8]) + "\n")
print(out, len(vals))
