#!/usr/bin/env python3
"""tools/reduce_scene.py scene.pbrt — delta-debugging of a scene on which oracle/_ref/pbrt_ref --wavefront and oracle/_build/wf_cpu give
different images (tools/diff_fuzz_scenes.py finding): removes blocks (AttributeBegin .. AttributeEnd, ObjectBegin .. ObjectEnd) and single
lines, then single parameters, while the two renders still differ; writes <scene>.min.pbrt."""
import os, re, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import read_pfm
REF = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref"); CPU = os.path.join(ROOT, "oracle", "_build", "wf_cpu")
td = tempfile.mkdtemp()

def differs(text):
    p = os.path.join(td, "t.pbrt"); open(p, "w").write(text)
    ro, co = os.path.join(td, "r.pfm"), os.path.join(td, "c.pfm")
    for q in (ro, co):
        if os.path.exists(q): os.unlink(q)
    try:
        if os.environ.get("REDUCE_SEQ"):
            # REDUCE_SEQ=1: the sequential pair (one reference thread against the checker's emulation of the reference's unwritten
            # MediumSampleWorkItem::depth) — for findings that are NOT that order dependence, in scenes with media: without it the
            # reduction drifts to a scene that differs through the stale depth only
            a = subprocess.run([REF, "--wavefront", "--quiet", "--seed", "0", "--nthreads", "1", "--outfile", ro, p], capture_output=True, timeout=20)
            b = subprocess.run([CPU, "--quiet", "--emulate-stale-medium-depth", "--outfile", co, p], capture_output=True, timeout=20)
        else:
            a = subprocess.run([REF, "--wavefront", "--quiet", "--seed", "0", "--nthreads", "4", "--outfile", ro, p], capture_output=True, timeout=8)
            b = subprocess.run([CPU, "--quiet", "--nthreads", "4", "--outfile", co, p], capture_output=True, timeout=8)
    except subprocess.TimeoutExpired:
        return False
    if a.returncode or b.returncode or not os.path.exists(ro) or not os.path.exists(co): return False
    r, c = read_pfm(ro), read_pfm(co)
    if r.shape == c.shape and (r.view(np.uint32) == c.view(np.uint32)).all(): return False
    if os.environ.get("REDUCE_SEQ"):
        # ... and the reference must not be reading memory it never wrote (the unwritten depth of a never-used slot: heap contents):
        # its image has to survive glibc's allocation fill pattern, or the reduction drifts to THAT difference
        rp = os.path.join(td, "rp.pfm")
        if os.path.exists(rp): os.unlink(rp)
        try:
            e = subprocess.run([REF, "--wavefront", "--quiet", "--seed", "0", "--nthreads", "1", "--outfile", rp, p], capture_output=True, timeout=20,
                               env=dict(os.environ, MALLOC_PERTURB_="85"))
        except subprocess.TimeoutExpired:
            return False
        if e.returncode or not os.path.exists(rp): return False
        q = read_pfm(rp)
        if q.shape != r.shape or not (q.view(np.uint32) == r.view(np.uint32)).all(): return False
    return True

def units(lines):
    out, i = [], 0
    while i < len(lines):
        if lines[i].startswith(("AttributeBegin", "ObjectBegin")):
            end = "AttributeEnd" if lines[i].startswith("AttributeBegin") else "ObjectEnd"
            j = i
            while j < len(lines) and not lines[j].startswith(end): j += 1
            out.append(lines[i:j + 1]); i = j + 1
        else:
            out.append([lines[i]]); i += 1
    return out

text = open(sys.argv[1]).read()
assert differs(text), "the scene does not reproduce a difference"
changed = True
while changed:
    changed = False
    us = units(text.splitlines())
    k = 0
    while k < len(us):
        if us[k] == ["WorldBegin"]: k += 1; continue
        cand = "\n".join(l for u in us[:k] + us[k + 1:] for l in u) + "\n"
        if differs(cand):
            us.pop(k); text = cand; changed = True
        else:
            k += 1
    # inside blocks: single lines
    lines = text.splitlines(); k = 0
    while k < len(lines):
        if lines[k].startswith(("AttributeBegin", "AttributeEnd", "ObjectBegin", "ObjectEnd", "WorldBegin")): k += 1; continue
        cand = "\n".join(lines[:k] + lines[k + 1:]) + "\n"
        if differs(cand): lines.pop(k); text = cand; changed = True
        else: k += 1
# single parameters:  "type name" [ ... ]  or  "type name" "value" / value
lines = text.splitlines()
pat = re.compile(r'\s"(?:float|integer|rgb|string|bool|texture|spectrum|point3|point2|vector3|normal)\s[A-Za-z0-9_.]+"\s(?:\[[^\]]*\]|"[^"]*"|\S+)')
for k in range(len(lines)):
    while True:
        ms = list(pat.finditer(lines[k])); done = True
        for m in ms:
            cand_line = lines[k][:m.start()] + lines[k][m.end():]
            cand = "\n".join(lines[:k] + [cand_line] + lines[k + 1:]) + "\n"
            if differs(cand):
                lines[k] = cand_line; text = cand; done = False; break
        if done: break
out = sys.argv[1].replace(".pbrt", ".min.pbrt")
open(out, "w").write(text)
print(text)
