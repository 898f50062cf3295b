#!/usr/bin/env python3
"""Summarise `-Rpass-analysis=kernel-resource-usage` remarks (make ... EXTRA=-Rpass-analysis=kernel-resource-usage 2> log): one line per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2:] or [""]
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    g = lambda k: int(m.group(1)) if (m := re.search(k + r": (\d+)", b)) else -1
    n = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if any(p in n for p in pat):
        print("%-60s vgpr %3d agpr %3d scratch %4d occ %d lds %6d sgpr %3d" % (n.split("(")[0][-60:], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), g("SGPRs")))
