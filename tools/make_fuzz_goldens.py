#!/usr/bin/env python3
"""tools/make_fuzz_goldens.py — the committed corpus of the differential fuzzer's scenes for the GPU leg (tests/test_gpu_parity.py::
test_fuzz_corpus_vs_reference).

    python tools/make_fuzz_goldens.py [--n 48] [--seed 4]

Generates scenes with tools/diff_fuzz_scenes.py's generator (seed base --seed), keeps the first --n that the reference renders, renders
deterministically (1, 2 and 4 threads agree) and the CPU port reproduces bit for bit, and writes them — file names made relative to the
corpus directory — with the reference's image to tests/golden/fuzz/s<seed>.pbrt / s<seed>_ref.pfm.  The reduced scenes of the findings
fixed in round 4 (tools/open_findings) join the corpus as regression cases.  Needs /root/reference's build (oracle/_ref/pbrt_ref); the GPU box only
reads the committed files."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import diff_fuzz_scenes as fz  # noqa: E402
from conftest import read_pfm  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "fuzz")


def relativise(text):
    return text.replace(fz.GOLDEN + "/", "../").replace("/root/repo/tests/golden/", "../")


def try_scene(name, text, work):
    """-> reference image path or None"""
    path = os.path.join(OUT, name + ".pbrt")
    open(path, "w").write(relativise(text))
    imgs = []
    for n in (4, 2, 1):
        o = os.path.join(work, "r%d.pfm" % n)
        if os.path.exists(o):
            os.unlink(o)
        st, _ = fz.render(fz.REF, ["--wavefront", "--quiet", "--seed", "0", "--nthreads", str(n)], path, o)
        if st != "ok":
            os.unlink(path)
            return None
        imgs.append(read_pfm(o))
    co = os.path.join(work, "c.pfm")
    st, _ = fz.render(fz.CPU, ["--quiet", "--nthreads", "4"], path, co)
    same = st == "ok" and all((imgs[0].view(np.uint32) == i.view(np.uint32)).all() for i in imgs[1:]) and (read_pfm(co).view(np.uint32) == imgs[0].view(np.uint32)).all()
    if not same or not np.isfinite(imgs[0]).all():
        os.unlink(path)
        return None
    shutil.copy(os.path.join(work, "r4.pfm"), os.path.join(OUT, name + "_ref.pfm"))
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--append", action="store_true", help="keep the committed corpus and add --n scenes of a NEW seed base to it (round 6: 54 -> 300+ scenes for the GPU leg)")
    ap.add_argument("--stress", action="store_true", help="diff_fuzz_scenes.py --stress: the grammar's rare features two to three times as often")
    ap.add_argument("--options", action="store_true", help="diff_fuzz_scenes.py --options: also draw the extended Option directives")
    a = ap.parse_args()
    fz.STRESS, fz.EXTENDED_OPTIONS = a.stress, a.options
    kept = []
    if a.append:
        kept = open(os.path.join(OUT, "CORPUS.txt")).read().split()
    else:
        shutil.rmtree(OUT, ignore_errors=True)
        os.makedirs(OUT)
    n_target = len(kept) + a.n
    work = tempfile.mkdtemp(prefix="wf_fuzzgold_")
    i = 0
    while a.append and len(kept) < n_target and i < 40 * a.n:
        seed = a.seed * 100000 + i
        i += 1
        if "s%d" % seed in kept:
            continue
        if try_scene("s%d" % seed, fz.Gen(seed).scene(), work):
            kept.append("s%d" % seed)
            print("kept s%d (%d / %d)" % (seed, len(kept), n_target), flush=True)
            open(os.path.join(OUT, "CORPUS.txt"), "w").write("\n".join(kept) + "\n")
    if a.append:
        shutil.rmtree(work, ignore_errors=True)
        print("%d scenes in %s" % (len(kept), OUT))
        return
    i = 0
    while len(kept) < a.n and i < 40 * a.n:
        seed = a.seed * 100000 + i
        i += 1
        if try_scene("s%d" % seed, fz.Gen(seed).scene(), work):
            kept.append("s%d" % seed)
            print("kept s%d (%d / %d)" % (seed, len(kept), a.n), flush=True)
    # regression cases: the reduced scenes of round 3's open findings
    for f in sorted(os.listdir(os.path.join(ROOT, "tools", "open_findings"))):
        if f.endswith(".pbrt"):
            name = "finding_" + f[:-5].replace(".", "_")
            if try_scene(name, open(os.path.join(ROOT, "tools", "open_findings", f)).read(), work):
                kept.append(name)
                print("kept", name, flush=True)
    open(os.path.join(OUT, "CORPUS.txt"), "w").write("\n".join(kept) + "\n")
    shutil.rmtree(work, ignore_errors=True)
    print("%d scenes in %s" % (len(kept), OUT))


if __name__ == "__main__":
    main()
