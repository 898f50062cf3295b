#!/usr/bin/env python3
"""tools/make_volpath_goldens.py — the PHYSICAL oracle of the north_star: pbrt's CPU VolPathIntegrator (cpu/integrators.cpp:953-1390).

The wavefront path is compared with `pbrt --wavefront` bit for bit everywhere else in tests/ (same sampler dimensions, same random
decisions: SURVEY 8(c) caveat 1 explains why a per-pixel comparison needs that).  VolPath draws its samples in a different order (and seeds
the media's RNG differently, wavefront/media.cpp:44 vs cpu/integrators.cpp:975-977), so the two agree IN EXPECTATION only.  This script
renders the downscaled benchmark stand-ins (tools/make_scenes.py bench_small: the generators of bench.py's workloads = BASELINE configs
2, 4, 3) and a low-resolution killeroo-like scene (configs[1]) with the reference's VolPath integrator — `pbrt_ref` WITHOUT --wavefront: the
scene files name "volpath" — twice, seeds 0 and 1, SPP samples per pixel each, and writes the two images' means over a GRID x GRID block
grid into tests/golden/volpath/<name>.json (a | b; their difference is the golden's own noise estimate).
tests/test_gpu_parity.py::test_volpath_in_expectation compares the GPU's image of the same scene with them.

    python tools/make_volpath_goldens.py [--spp 2048] [--grid 8]

Needs oracle/_ref/pbrt_ref (the reference compiled here); about 25 minutes on 8 cores.  Test infrastructure only."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conftest import read_pfm  # noqa: E402
import make_scenes  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref")
OUT = os.path.join(ROOT, "tests", "golden", "volpath")
KILLEROO_SMALL = dict(res=(192, 108), spp=4)


def scene_for(name, td):
    """(scene path) of the stand-in `name`, generated into td — the SAME generator call the GPU test makes"""
    if name == "killeroo_like_small":
        p = os.path.join(td, name + ".pbrt")
        make_scenes.killeroo_like(p, KILLEROO_SMALL["res"], KILLEROO_SMALL["spp"])
        return p
    return make_scenes.bench_small(name, td)


def block_means(img, grid):
    h, w, _ = img.shape
    ys = np.linspace(0, h, grid + 1).astype(int)
    xs = np.linspace(0, w, grid + 1).astype(int)
    return np.array([[img[ys[j]:ys[j + 1], xs[i]:xs[i + 1]].mean(axis=(0, 1)) for i in range(grid)] for j in range(grid)])   # [grid][grid][3]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=2048)
    ap.add_argument("--grid", type=int, default=8)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    names = [n for n in list(make_scenes.BENCH_SMALL) + ["killeroo_like_small"] if not a.only or n in a.only.split(",")]
    for name in names:
        with tempfile.TemporaryDirectory() as td:
            path = scene_for(name, td)
            tree = make_scenes.tree_hash(td) if name != "killeroo_like_small" else None   # (before the renders land in td)
            imgs = []
            for seed in (0, 1):
                out = os.path.join(td, "vp%d.pfm" % seed)
                subprocess.run([REF, "--quiet", "--seed", str(seed), "--spp", str(a.spp), "--nthreads", str(os.cpu_count()), "--outfile", out, path],
                               check=True, cwd=td)
                imgs.append(read_pfm(out).astype(np.float64))
            rec = {"scene": name, "integrator": "volpath (pbrt_ref without --wavefront; cpu/integrators.cpp:953-1390)", "spp_each": a.spp, "seeds": [0, 1], "grid": a.grid,
                   "resolution": [imgs[0].shape[1], imgs[0].shape[0]], "tree_hash": tree,
                   "mean_a": imgs[0].mean(axis=(0, 1)).tolist(), "mean_b": imgs[1].mean(axis=(0, 1)).tolist(),
                   "blocks_a": block_means(imgs[0], a.grid).tolist(), "blocks_b": block_means(imgs[1], a.grid).tolist()}
            json.dump(rec, open(os.path.join(OUT, name + ".json"), "w"))
            print(name, "mean a", rec["mean_a"], "b", rec["mean_b"], flush=True)


if __name__ == "__main__":
    main()
