#!/usr/bin/env python3
"""tools/stack_audit.py [build dir | unit.o ...] — is every kernel's scratch allocation as deep as its deepest call chain?

A gfx950 kernel gets `.private_segment_fixed_size` bytes of scratch per lane; its own frame sits at the bottom, it sets the stack pointer
(s32) above it and every device function it calls bumps s32 by its own frame (`s_addk_i32 s32, N` / `s_add_i32 s32, s32, N`).  A call chain
deeper than the allocation writes into the scratch of other lanes / waves: silent corruption, unrepeatable images, memory faults — the
symptoms of DESIGN.md 4.6's toolchain defect.  This walks the call graph of every code object (direct calls: s_getpc_b64 + s_add_u32 +
s_swappc_b64; an indirect call is reported), sums the frames along the deepest chain and compares with the allocation.
Build-time analysis only (llvm-objdump / llvm-readelf); nothing is executed."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_spill_carriers as C

READELF = C.OBJDUMP.replace("llvm-objdump", "llvm-readelf")


def parse(co):
    txt = subprocess.run([C.OBJDUMP, "-d", co], capture_output=True, text=True).stdout
    funcs, cur = {}, None
    for l in txt.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", l)
        if m:
            cur = m.group(2)
            funcs[cur] = {"addr": int(m.group(1), 16), "ins": []}
            continue
        m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+):", l)
        if m and cur:
            funcs[cur]["ins"].append((int(m.group(2), 16), m.group(1).strip()))
    return funcs


def analyse(co):
    funcs = parse(co)
    by_addr = {f["addr"]: n for n, f in funcs.items()}
    notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
    kern = {}
    for blk in notes.split("- .agpr_count")[1:]:
        n = re.search(r"\.name:\s+(\S+)", blk)
        p = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        d = re.search(r"\.uses_dynamic_stack:\s+(\w+)", blk)
        if n and p:
            kern[n.group(1)] = (int(p.group(1)), d.group(1) if d else "?")
    info = {}
    for name, f in funcs.items():
        ins = f["ins"]
        frame, sp0, calls, indirect = 0, None, set(), 0
        pcreg = {}
        for k, (a, l) in enumerate(ins):
            m = re.match(r"s_addk_i32 s32, (0x[0-9a-f]+|\d+)", l) or re.match(r"s_add_i32 s32, s32, (0x[0-9a-f]+|\d+)", l)
            if m:
                v = int(m.group(1), 0)
                if v < 0x8000 or "s_add_i32" in l:
                    frame = max(frame, v if v < 2**31 else 0)
            m = re.match(r"s_mov(?:k_i32|_b32) s32, (0x[0-9a-f]+|\d+)", l)
            if m and sp0 is None:
                sp0 = int(m.group(1), 0)
            m = re.match(r"s_getpc_b64 s\[(\d+):(\d+)\]", l)
            if m:
                pcreg[int(m.group(1))] = ins[k + 1][0] if k + 1 < len(ins) else a + 4
            m = re.match(r"s_add_u32 s(\d+), s(\d+), (0x[0-9a-f]+|-?\d+)", l)
            if m and int(m.group(1)) == int(m.group(2)) and int(m.group(1)) in pcreg and isinstance(pcreg[int(m.group(1))], int):
                imm = int(m.group(3), 0)
                if imm >= 2**31:
                    imm -= 2**32
                pcreg[int(m.group(1))] = ("T", pcreg[int(m.group(1))] + imm)
            m = re.match(r"s_swappc_b64 s\[30:31\], s\[(\d+):(\d+)\]", l)
            if m:
                t = pcreg.get(int(m.group(1)))
                if isinstance(t, tuple) and t[1] in by_addr:
                    calls.add(by_addr[t[1]])
                else:
                    indirect += 1
        info[name] = {"frame": frame, "sp0": sp0, "calls": calls, "indirect": indirect}
    depth_memo = {}

    def depth(n, seen=()):
        if n in depth_memo:
            return depth_memo[n]
        if n in seen:
            return (float("inf"), [n + " (recursion)"])
        best, chain = 0, []
        for c in info[n]["calls"]:
            d, ch = depth(c, seen + (n,))
            if d > best:
                best, chain = d, ch
        r = (info[n]["frame"] + best, [n] + chain)
        depth_memo[n] = r
        return r

    rows = []
    for k, (alloc, dyn) in kern.items():
        if k not in info:
            continue
        i = info[k]
        below = 0
        chain = []
        for c in i["calls"]:
            d, ch = depth(c)
            if d > below:
                below, chain = d, ch
        need = (i["sp0"] or 0) + below
        rows.append((k, alloc, i["sp0"], below, need, dyn, sum(info[c]["indirect"] for c in info if c == k or c in chain) , chain))
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")] or [os.path.join(C.ROOT, "pbrt-v4_amd", "_build")]
    objs = []
    for a in args:
        objs += sorted(os.path.join(a, f) for f in os.listdir(a) if f.endswith(".o")) if os.path.isdir(a) else [a]
    bad = n = withcalls = 0
    for obj in objs:
        co = "/tmp/stack_audit.co"
        if not C.code_object(obj, co):
            continue
        for k, alloc, sp0, below, need, dyn, indirect, chain in analyse(co):
            n += 1
            withcalls += below > 0
            short = need > alloc
            if short or indirect or "-v" in sys.argv:
                dk = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                print("%-22s %-60s allocation %5d  own frame (s32 at entry) %5s  deepest chain below %5s  needed %5s %s%s" % (
                    os.path.basename(obj), re.sub(r"\(.*", "", dk)[:60], alloc, sp0, below, need, "  ** SHORT by %s **" % (need - alloc) if short else "",
                    "  (%d indirect call(s): not followed)" % indirect if indirect else ""))
                if short:
                    print("      chain: " + " -> ".join(re.sub(r"\(.*", "", c) for c in subprocess.run(["c++filt"], input="\n".join(chain), capture_output=True, text=True).stdout.split("\n") if c))
            bad += short
    print("%d kernels (%d with calls): %d whose deepest direct call chain needs more scratch than the kernel is allocated" % (n, withcalls, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
