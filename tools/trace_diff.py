#!/usr/bin/env python3
"""tools/trace_diff.py — where does a pixel of a scene first diverge between the reference and the CPU port?

    python tools/trace_diff.py scene.pbrt [--pixel x,y | --auto] [--spp N]

Renders the scene with oracle/_ref/pbrt_ref --wavefront and oracle/_build/wf_cpu, picks a differing pixel (--auto, default) and the first
sample index at which that pixel differs, then prints the diff of the two path traces (oracle/_ref/ref_trace, wf_cpu --trace-path:
the path state after every stage of every depth, hex floats).  Test infrastructure; needs /root/reference's build."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import read_pfm  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref")
TRACE = os.path.join(ROOT, "oracle", "_ref", "ref_trace")
CPU = os.path.join(ROOT, "oracle", "_build", "wf_cpu")


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--pixel")
    ap.add_argument("--sample", type=int)
    ap.add_argument("--context", type=int, default=0)
    a = ap.parse_args()
    scene = os.path.abspath(a.scene)
    cwd = os.path.dirname(scene)
    td = tempfile.mkdtemp(prefix="wf_trace_")
    r, c = os.path.join(td, "r.pfm"), os.path.join(td, "c.pfm")
    if a.pixel:
        px, py = (int(v) for v in a.pixel.split(","))
    else:
        run([REF, "--wavefront", "--quiet", "--seed", "0", "--nthreads", "4", "--outfile", r, scene], cwd=cwd)
        run([CPU, "--quiet", "--nthreads", "4", "--outfile", c, scene], cwd=cwd)
        ri, ci = read_pfm(r), read_pfm(c)
        d = (ri.view(np.uint32) != ci.view(np.uint32)).any(axis=2)
        print("%d of %d pixels differ" % (d.sum(), d.size))
        if not d.any():
            return
        ys, xs = np.nonzero(d)
        py, px = int(ys[0]), int(xs[0])
        print("pixel %d,%d: ref %s  cpu %s" % (px, py, ri[py, px], ci[py, px]))
    text = open(scene).read()
    spp = 64
    m = re.search(r'"integer pixelsamples"\s*\[?\s*(\d+)', text)
    if m:
        spp = int(m.group(1))
    # the film's pixel (0, 0) is the crop window's first pixel: --pixel takes image coordinates
    samples = [a.sample] if a.sample is not None else range(spp)
    for s in samples:
        run([REF, "--wavefront", "--quiet", "--seed", "0", "--nthreads", "1", "--pixel", "%d,%d" % (px, py), "--debugstart", "%d,1" % s, "--outfile", r, scene], cwd=cwd)
        run([CPU, "--quiet", "--nthreads", "1", "--pixel", "%d,%d" % (px, py), "--debugstart", "%d,1" % s, "--outfile", c, scene], cwd=cwd)
        if not os.path.exists(r) or not os.path.exists(c):
            continue
        if a.sample is None and (read_pfm(r).view(np.uint32) == read_pfm(c).view(np.uint32)).all():
            continue
        print("sample %d differs" % s)
        # a one-pixel copy of the scene: the Film directive gets "integer pixelbounds"
        one = os.path.join(cwd, "_trace_one_pixel.pbrt")
        film = re.sub(r'(Film\s+"\w+")', r'\1 "integer pixelbounds" [ %d %d %d %d ]' % (px, px + 1, py, py + 1), text, count=1)
        film = re.sub(r'"float cropwindow"\s*\[[^\]]*\]', "", film)
        open(one, "w").write(film)
        try:
            tr = run([TRACE, one, str(s)], cwd=cwd).stdout.splitlines()
            tc = [l for l in run([CPU, "--quiet", "--nthreads", "1", "--trace-path", "--samples", str(s), str(s + 1), "1", "--outfile", c, one], cwd=cwd).stdout.splitlines() if l.startswith("d")]
        finally:
            os.unlink(one)
        tr = [l for l in tr if l.startswith("d")]
        n = 0
        for i in range(max(len(tr), len(tc))):
            lr = tr[i] if i < len(tr) else "<none>"
            lc = tc[i] if i < len(tc) else "<none>"
            if lr != lc:
                print("ref: " + lr)
                print("cpu: " + lc)
                n += 1
                if n > a.context:
                    break
            elif n:
                print("     " + lr)
        if n == 0:
            print("traces equal (%d lines): the difference is outside the traced stages" % len(tr))
        return
    print("no single sample differs at that pixel")


if __name__ == "__main__":
    main()
