#!/usr/bin/env python3
"""Prints the constant tables of glibc 2.35's float routines (x86-64) as C initialisers.

The device restatement of glibc's sinf/cosf/expf/logf (pbrt-v4_amd/csrc/common/wf_libm.h) needs the
exact table constants of the installed libm: __sincosf_table / __inv_pio4 (s_sincosf_data.o),
__exp2f_data (e_exp2f_data.o) and __logf_data (e_logf_data.o).  They are read from the static archive
/lib/x86_64-linux-gnu/libm-2.35.a of this image and printed as hex literals; the output was pasted into
wf_libm.h once.  tests/test_libm_restatement.py checks the restated functions against the live libm.
"""
import struct, subprocess, sys, tempfile, os

AR = "/lib/x86_64-linux-gnu/libm-2.35.a"

def rodata(member):
    with tempfile.TemporaryDirectory() as td:
        subprocess.run(["ar", "x", AR, member], cwd=td, check=True)
        subprocess.run(["objcopy", "-O", "binary", "-j", ".rodata", member, "out.bin"], cwd=td, check=True)
        return open(os.path.join(td, "out.bin"), "rb").read()

def u64s(b): return struct.unpack("<%dQ" % (len(b) // 8), b)
def u32s(b): return struct.unpack("<%dI" % (len(b) // 4), b)

def dbl(u): return struct.unpack("<d", struct.pack("<Q", u))[0].hex()

d = rodata("s_sincosf_data.o")
print("// __inv_pio4[24]"); print(", ".join("0x%08xu" % v for v in u32s(d[:0x60])))
t = u64s(d[0x60:0x60 + 2 * 0x70])
for e in range(2):
    print("// __sincosf_table[%d]: sign[4], hpi_inv, hpi, c0, c1, s1?, ... by offset" % e)
    print(", ".join(dbl(v) for v in t[e * 14:(e + 1) * 14]))
d = rodata("e_exp2f_data.o")
v = u64s(d)
print("// __exp2f_data.tab[32] (u64)"); print(", ".join("0x%016xull" % x for x in v[:32]))
print("// shift_scaled, poly[3], shift, invln2_scaled, poly_scaled[3]"); print(", ".join(dbl(x) for x in v[32:]))
d = rodata("e_logf_data.o")
v = u64s(d)
print("// __logf_data.tab[16] {invc, logc}"); print(", ".join("{%s, %s}" % (dbl(v[2 * i]), dbl(v[2 * i + 1])) for i in range(16)))
print("// ln2, poly[3]"); print(", ".join(dbl(x) for x in v[32:]))
