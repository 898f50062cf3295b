#!/usr/bin/env python3
"""pbrt-v4_amd/data/sobol_matrices.bin: the Sobol' generator matrices the reference's SobolSampler reads (util/sobolmatrices.cpp:
SobolMatrices32[1024 x 52] — Joe & Kuo's direction numbers as tabulated by L. Gruenschloss — and the VdCSobolMatrices /
VdCSobolMatricesInv tables of SobolIntervalToIndex), read out of the reference source the way tools/extract_spectral_tables.py reads
the CIE tables.  Layout: uint32 magic 'SOBM', uint32 nDims, uint32 matrixSize, uint32 nVdC, uint32 nVdCInv, then the uint32
matrices, the uint64 VdC rows, the uint64 VdCInv rows.  Needs /root/reference.

Data: S. Joe and F. Y. Kuo, "Constructing Sobol sequences with better two-dimensional projections", SIAM J. Sci. Comput. 30,
2635-2654 (2008), http://web.maths.unsw.edu.au/~fkuo/sobol/ ; tabulation (c) 2012 Leonhard Gruenschloss, distributed under the
permissive licence reproduced in pbrt-v4_amd/data/sobol_matrices.LICENSE."""
import os, re, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/pbrt/util/sobolmatrices.cpp").read()

def table(name):
    m = re.search(name + r"\[[^\]]*\](?:\[[^\]]*\])?\s*=\s*\{(.*?)\n\};", src, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    return [int(v.rstrip("ULul"), 16) for v in re.findall(r"0x[0-9a-fA-F]+(?:ULL|ull|UL|ul|U|u)?", body)]

m32 = table("SobolMatrices32")
def rows(name):
    # rows may list fewer than 52 initialisers (the rest are zero)
    m = re.search(name + r"\[\]\[SobolMatrixSize\]\s*=\s*\{(.*?)\}\};", src, re.S)
    out = []
    for r in re.findall(r"\{//\s*m = \d+(.*?)(?=\}\s*,\s*\{//|$)", m.group(1), re.S):
        vals = [int(v.rstrip("ULul"), 16) for v in re.findall(r"0x[0-9a-fA-F]+(?:ULL|ull)?", r)]
        assert len(vals) <= 52
        out += vals + [0] * (52 - len(vals))
    return out

vdc = rows("VdCSobolMatrices")
inv = rows("VdCSobolMatricesInv")
assert len(m32) == 1024 * 52 and len(vdc) % 52 == 0 and len(inv) % 52 == 0, (len(m32), len(vdc), len(inv))
out = os.path.join(ROOT, "pbrt-v4_amd", "data", "sobol_matrices.bin")
with open(out, "wb") as f:
    f.write(struct.pack("<4sIIII", b"SOBM", 1024, 52, len(vdc) // 52, len(inv) // 52))
    f.write(struct.pack("<%dI" % len(m32), *m32))
    f.write(struct.pack("<%dQ" % len(vdc), *vdc))
    f.write(struct.pack("<%dQ" % len(inv), *inv))
lic = re.search(r"// Copyright \(c\) 2012 Leonhard Gruenschloss.*?SOFTWARE\.", src, re.S).group(0)
open(os.path.join(ROOT, "pbrt-v4_amd", "data", "sobol_matrices.LICENSE"), "w").write(re.sub(r"(?m)^// ?", "", lic) + "\n")
print(out, len(m32), len(vdc) // 52, len(inv) // 52, os.path.getsize(out))
