#!/usr/bin/env python3
"""tools/carrier_slots.py unit.o|dir ... — do the scratch slots that hold whole-wave images of SGPR-spill carriers have other tenants?

StackSlotColoring lets spill slots with disjoint live ranges share their bytes.  A carrier VGPR (lanes = spilled SGPRs) is saved with ALL
lanes enabled; an ordinary spill writes the ACTIVE lanes only.  When the two share a slot, what a carrier reload returns for an inactive
lane depends on whole-wave liveness, which is not what thread-level live ranges describe.  This lists, per kernel, every kernel-frame slot
that a carrier is saved to and every other register stored to overlapping bytes (the round-4 incident kernel has two).
Build-time analysis only."""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_spill_carriers as C
import carrier_audit as A


def main():
    objs = []
    for a in [x for x in sys.argv[1:] if not x.startswith("-")] or [os.path.join(C.ROOT, "pbrt-v4_amd", "_build")]:
        objs += sorted(os.path.join(a, f) for f in os.listdir(a) if f.endswith(".o")) if os.path.isdir(a) else [a]
    shared = ncar = 0
    for obj in objs:
        co = "/tmp/carrier_slots.co"
        if not C.code_object(obj, co):
            continue
        for name, ins in A.functions(co).items():
            carriers = {m.group(1) for l in ins for m in [re.match(r"v_writelane_b32 (v\d+),", l)] if m}
            if not carriers:
                continue
            acc = collections.defaultdict(collections.Counter)   # (offset, dwords, base) -> {(kind, reg, wwm): n}
            for i, l in enumerate(ins):
                m = re.match(r"scratch_(store|load)_dword(x\d)? (.*)$", l)
                if not m:
                    continue
                toks = [t.strip() for t in m.group(3).split(",")]
                reg = toks[1] if m.group(1) == "store" else toks[0]
                base = toks[-1].split()[0]
                addr = toks[0] if m.group(1) == "store" else toks[1]
                if addr != "off":
                    continue     # lane-addressed private arrays, not spill slots
                o = re.search(r"offset:(\d+)", l)
                wwm = bool(re.match(r"s_(or|xor)_saveexec_b64 \S+ -1|s_mov_b64 exec, -1", ins[i - 1])) if i else False
                acc[(int(o.group(1)) if o else 0, int((m.group(2) or "x1")[1:]), base)][(m.group(1), reg, wwm)] += 1
            cslots = {k for k, c in acc.items() if any(r in carriers and w for (_, r, w) in c)}
            ncar += len(cslots)
            for (o, w, b) in sorted(cslots):
                for (o2, w2, b2), c in sorted(acc.items()):
                    others = {k: n for k, n in c.items() if not (k[1] in carriers and k[2])}
                    if b2 == b and o2 < o + 4 * w and o < o2 + 4 * w2 and others:
                        shared += 1
                        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                        print("%s %s: carrier slot %s+%d also holds %s" % (os.path.basename(obj), re.sub(r"\(.*", "", d)[:50], b, o,
                              ", ".join("%s %s x%d%s" % (k[0], k[1], n, "" if not k[2] else " (whole-wave)") for k, n in sorted(others.items()))))
    print("%d carrier save slot(s) in %d unit(s); %d shared with another tenant" % (ncar, len(objs), shared))


if __name__ == "__main__":
    main()
