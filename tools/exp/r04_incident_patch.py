#!/usr/bin/env python3
"""tools/exp/r04_incident_patch.py in.so out.so — the causal test of DESIGN.md 4.6's mechanism.

In the round-4 incident kernel (k_eval_material<1, 0> of commit f38bd9e, default flags) the compiler emits, twice:
    s_or_saveexec_b64 s[100:101], -1          ; all lanes on, for the whole-wave copy of an SGPR-spill carrier
    v_mov_b64_e32 v[120:121], v[154:155]      ; ordinary live-range-split copies of the register allocator ...
    v_mov_b32_e32 v155, v153                  ; ... executed with ALL lanes enabled: v155 (the high half of a light pointer in the lanes
    v_mov_b32_e32 v106, 0x260                 ;     that are inactive in this divergent region) is overwritten in every lane
    v_bfrev_b32_e32 v73, 1
    v_mov_b32_e32 v164, v166                  ; the carrier copy the bracket was opened for
    s_mov_b64 exec, s[100:101]
and restores v[154:155] from v[120:121] later under the region's EXEC — the active lanes only.  This moves the s_or_saveexec behind the four
ordinary instructions (same bytes, reordered; no branch target inside the span)."""
import struct
import sys


def dw(*ws):
    return b"".join(struct.pack("<I", w) for w in ws)


d = bytearray(open(sys.argv[1], "rb").read())
old = dw(0xBEE421C1, 0x7EF0719A, 0x7F360399, 0x7ED402FF, 0x00000260, 0x7E925881, 0x7F4803A6, 0xBEFE0164)
new = dw(0x7EF0719A, 0x7F360399, 0x7ED402FF, 0x00000260, 0x7E925881, 0xBEE421C1, 0x7F4803A6, 0xBEFE0164)
n, i = 0, d.find(old)
while i >= 0:
    d[i:i + len(old)] = new
    n += 1
    i = d.find(old, i + 1)
print("%d site(s) patched" % n)
open(sys.argv[2], "wb").write(d)
