#!/usr/bin/env python3
"""tools/carrier_audit.py unit.o [function-regex] — what happens to the SGPR-spill carrier VGPRs of a gfx950 code object, in program order.

The defect class of DESIGN.md 4.6: a kernel whose carrier VGPR (the target of v_writelane_b32: SGPRs spilled into its lanes) is itself
saved to scratch.  v_writelane / v_readlane ignore EXEC, an ordinary scratch store does not: a carrier saved or reloaded under a partial
EXEC loses the lanes of the inactive threads — and with them spilled SGPRs (pointers, saved EXEC masks).  For every carrier this prints the
sequence of events (W writelane, R readlane, S / s scratch store whole-wave / under the current EXEC, L / l scratch load likewise, C / c
whole-wave copy into / out of it, D other definition, U other use; * = a writelane target, the others hold whole-wave copies) and flags
  * a store / load of the carrier that is NOT bracketed by `s_or_saveexec_b64 sX, -1` ... `s_mov_b64 exec, sX`   (partial-EXEC save)
  * a readlane from the carrier while an ordinary definition has replaced it and no reload has been seen (linear order: a hint, the code
    is not a straight line)
Build-time analysis only (llvm-objdump); nothing is executed."""
import re
import subprocess
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_spill_carriers as C


def functions(co):
    txt = subprocess.run([C.OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    cur, out = None, {}
    for l in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", l)
        if m:
            cur = m.group(1)
            out[cur] = []
        elif cur and l.strip() and not l.startswith("Disassembly"):
            out[cur].append(l.strip().split("//")[0].strip())
    return out


def regs_of(tok):
    """v167 -> {167}; v[10:13] -> {10..13}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def audit(name, ins, verbose=True):
    # whole-wave regions in program order: s_or_saveexec_b64 sX, -1 ... s_mov_b64 exec, sX
    # idioms: s_or_saveexec_b64 sX, -1 | s_mov_b64 sX, exec ; s_mov_b64 exec, -1 (keeps SCC) | s_xor_saveexec_b64 sX, -1 (a callee's
    # prologue / epilogue: the lanes that are INACTIVE in the caller — the active ones are the caller's to save) | s_mov_b64 exec, <2^n - 1>
    # (an SGPR spilled to memory through lanes 0..n-1 of a borrowed VGPR)
    wwm, inw = [], False
    for l in ins:
        f = l.replace(",", "")
        m = re.match(r"s_mov_b64 exec (\S+)$", f)
        if re.match(r"s_(or|xor)_saveexec_b64 \S+ -1", f) or (m and re.fullmatch(r"-1|\d+|0x[0-9a-f]+", m.group(1))):
            inw = True
        elif m:                      # exec restored from an SGPR pair
            wwm.append(inw)
            inw = False
            continue
        wwm.append(inw)
    carriers = {int(m.group(1)) for l in ins for m in [re.match(r"v_writelane_b32 v(\d+),", l)] if m}
    carriers_r = {int(m.group(1)) for l in ins for m in [re.match(r"v_readlane_b32 s\d+, v(\d+),", l)] if m}   # reload targets of a carrier's slot
    # the compiler splits the live range of a carrier with whole-wave copies (v_mov under EXEC = -1): their other side holds lanes too
    group = set(carriers) | carriers_r
    changed = True
    while changed:
        changed = False
        for k, l in enumerate(ins):
            m = re.match(r"v_mov_b32_e32 v(\d+), v(\d+)$", l)
            if m and wwm[k]:
                a, b = int(m.group(1)), int(m.group(2))
                if (a in group) != (b in group):
                    group |= {a, b}
                    changed = True
    findings = []
    # (round 5, the mechanism of the round-4 incident) an ORDINARY vector instruction inside a whole-wave bracket writes the lanes of threads
    # that are inactive at that point — lanes whose values thread-level liveness (and the restore that follows under the region's EXEC) does
    # not cover.  Only the `s_or_saveexec_b64 sX, -1` / `s_mov_b64 exec, -1` brackets count: a callee's prologue (s_xor_saveexec) and the
    # SGPR-to-memory idiom (exec = 2^n - 1) enable lanes on purpose.
    full, inb = [], False
    for l in ins:
        f = l.replace(",", "")
        if re.match(r"s_or_saveexec_b64 \S+ -1", f) or re.match(r"s_mov_b64 exec -1$", f):
            inb = True
            full.append(False)
            continue
        if re.match(r"s_mov_b64 exec ", f) or re.match(r"s_(and|or|xor|andn2)\w*_b64 exec ", f):
            inb = False
        full.append(inb)
    for k, l in enumerate(ins):
        if not full[k]:
            continue
        op, _, rest = l.partition(" ")
        toks = [t.strip() for t in rest.split(",")]
        if op.startswith("scratch_store") and len(toks) > 1 and toks[0] == "off":
            # an ordinary SPILL STORE with all lanes enabled overwrites the slot's bytes of the inactive threads (synthetic seed 5)
            src = regs_of(toks[1])
            if src and not (src & group):
                findings.append((name, min(src), k, "ordinary spill store with ALL lanes enabled (inside a whole-wave bracket): " + l))
            continue
        if not op.startswith(("v_", "scratch_load", "global_load", "flat_load", "buffer_load", "ds_read")) or op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_writelane", "v_nop")):
            continue
        dst = regs_of(toks[0]) if toks else set()
        if dst and not (dst & group):
            findings.append((name, min(dst), k, "ordinary vector instruction with ALL lanes enabled (inside a whole-wave bracket): " + l))
    for v in sorted(group):
        # per-lane bookkeeping in program order: which lanes hold a spilled SGPR.  None = unknown (a reload of a slot not seen stored)
        ev, valid, slots = [], set(), {}
        for k, l in enumerate(ins):
            op, _, rest = l.partition(" ")
            toks = [t.strip() for t in rest.split(",")]
            w = wwm[k]
            off = re.search(r"offset:(\d+)", l)
            base = toks[-1].split() if toks else []
            off = (base[0] if base else "") + ":" + (off.group(1) if off else "0")
            if op == "v_writelane_b32" and regs_of(toks[0]) == {v}:
                ev.append("W")
                if valid is not None:
                    valid.add(int(toks[2]) if toks[2].isdigit() else -1)
            elif op == "v_readlane_b32" and len(toks) > 1 and regs_of(toks[1]) == {v}:
                ev.append("R")
                lane = int(toks[2]) if toks[2].isdigit() else -1
                if valid is not None and lane not in valid:
                    findings.append((name, v, k, "readlane of lane %d, which holds no spilled SGPR at this point of the listing" % lane))
            elif op.startswith("scratch_store") and len(toks) > 1 and v in regs_of(toks[1]):
                ev.append("S" if w else "s")
                if not w and valid:
                    findings.append((name, v, k, "%d live lanes stored to scratch under the current EXEC (not whole-wave): %s" % (len(valid), l)))
                slots[off] = None if valid is None else set(valid)
            elif op.startswith("scratch_load") and v in regs_of(toks[0]):
                ev.append("L" if w else "l")
                if w:
                    valid = slots.get(off, None)
                    valid = None if valid is None else set(valid)
                else:
                    if valid:
                        findings.append((name, v, k, "ordinary reload over %d live lanes: %s" % (len(valid), l)))
                    valid = set()
            else:
                isdef = toks and op.startswith(("v_", "flat_load", "global_load", "ds_read", "buffer_load")) and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane"))
                dst = regs_of(toks[0]) if isdef else set()
                if v in dst:
                    if w and op == "v_mov_b32_e32" and regs_of(toks[1]):
                        ev.append("C")       # whole-wave copy INTO the register: the lanes of the source (not tracked across registers)
                        valid = None
                    else:
                        ev.append("D")
                        if valid:
                            # (legal when the lanes are dead; a later readlane of them is what gets reported)
                            pass
                        valid = set()
                elif any(v in regs_of(t) for t in toks[1:]):
                    ev.append("c" if (w and op == "v_mov_b32_e32") else "U")
        comp, last, n = [], None, 0
        for e in ev + [None]:
            if e == last:
                n += 1
            else:
                if last:
                    comp.append(last + (str(n) if n > 1 else ""))
                last, n = e, 1
        if verbose:
            print("%-50s v%-3d%s %s" % (name[:50], v, "*" if v in carriers else " ", " ".join(comp)[:300]))
    return findings, len(group)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    verbose = "-q" not in sys.argv
    objs = []
    for a in args[:1]:
        objs = sorted(os.path.join(a, f) for f in os.listdir(a) if f.endswith(".o")) if os.path.isdir(a) else [a]
    pat = re.compile(args[1]) if len(args) > 1 else None
    allf, nfun, nreg = [], 0, 0
    for obj in objs:
        co = "/tmp/carrier_audit.co"
        if not C.code_object(obj, co):
            continue
        fs = functions(co)
        names = list(fs)
        dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")))
        for n in names:
            d = dem.get(n, n)
            if pat and not pat.search(d):
                continue
            f, g = audit(os.path.basename(obj) + " " + re.sub(r"\(.*", "", d), fs[n], verbose)
            allf += f
            nfun += g > 0
            nreg += g
    hard = [f for f in allf if "readlane of lane" not in f[3] and "ALL lanes enabled" not in f[3]]
    wave = [f for f in allf if "ALL lanes enabled" in f[3]]
    soft = [f for f in allf if "readlane of lane" in f[3]]
    print("\n%d unit(s), %d function(s) with lane-carrying VGPRs, %d such registers" % (len(objs), nfun, nreg))
    print("%d save / reload of live lanes under a partial EXEC (the defect class)" % len(hard))
    for f in hard:
        print("  %s v%d @%d: %s" % f)
    print("%d ordinary vector instruction(s) executed inside a whole-wave bracket (the mechanism of the round-4 incident: DESIGN.md 4.6)" % len(wave))
    for f in wave[:40]:
        print("  %s v%d @%d: %s" % f)
    print("%d readlane(s) that precede the writelane of their lane IN LISTING ORDER (loop-carried SGPRs look like this: a hint, not a verdict)" % len(soft))
    if verbose:
        for f in soft:
            print("  %s v%d @%d: %s" % f)
    sys.exit(1 if hard or wave else 0)


if __name__ == "__main__":
    main()
