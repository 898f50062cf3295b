#!/usr/bin/env python3
"""tests/golden/bvh_stats.json: what `pbrt --wavefront --stats` (the shimmed reference build oracle/_ref/pbrt_ref)
reports for its BVHAggregate on the parity scenes and on small renders of the bench stand-ins: interior / leaf node counts,
primitives in leaves, total "Nodes visited" and "Ray-Triangle intersection tests" over the render.  tests/test_host.py
compares oracle/wf_cpu's own counters with these: the restated SAH builder produces the reference's tree, and the
reference-order walk visits the reference's nodes — the N_nodes / N_tris of the roofline's algorithmic bytes
 are the reference's numbers, not a private tree's."""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_scenes
REF = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref")
G = os.path.join(ROOT, "tests", "golden")


def ref_stats(path, spp):
    with tempfile.TemporaryDirectory() as td:
        p = subprocess.run([REF, "--wavefront", "--seed", "0", "--spp", str(spp), "--stats", "--outfile", os.path.join(td, "o.pfm"), path],
                           capture_output=True, text=True, check=True, cwd=os.path.dirname(path))
    t = p.stdout + p.stderr
    def one(pat, default=None):
        m = re.search(pat, t)
        return int(m.group(1)) if m else default
    tri = re.search(r"Ray-Triangle intersection tests\s+(\d+) /\s+(\d+)", t)
    return {"bvh_interior_nodes": one(r"Interior nodes\s+(\d+)", 0), "bvh_leaf_nodes": one(r"Leaf nodes\s+(\d+)"),
            "bvh_leaf_prims": int(re.search(r"Primitives per leaf node\s+(\d+) /", t).group(1)),
            "bvh_nodes_visited": one(r"Nodes visited\s+(\d+)"), "tri_tests": int(tri.group(2)) if tri else 0}


def scenes(td):
    out = [(n, os.path.join(G, n + ".pbrt"), 4) for n in ("cornell64", "blobs_small", "materials_lights", "alpha_normalmap", "instances", "envmap", "blobs_hlbvh")]
    # SplitMethod::Middle / EqualCounts: blobs_small with the Accelerator directive in front of WorldBegin
    for m in ("middle", "equal"):
        v = os.path.join(td, "blobs_%s.pbrt" % m)
        open(v, "w").write(open(os.path.join(G, "blobs_small.pbrt")).read().replace("\nWorldBegin", '\nAccelerator "bvh" "string splitmethod" "%s"\nWorldBegin' % m, 1))
        out.append(("blobs_" + m, v, 4))
    k = os.path.join(td, "killeroo_like_240.pbrt")
    make_scenes.killeroo_like(k, (240, 135), 1)
    out.append(("killeroo_like_240x135_1spp", k, 1))
    s = os.path.join(td, "sanmiguel_like_small.pbrt")
    make_scenes.sanmiguel_like(s, (240, 135), 1, n_meshes=100, n_defs=10, tex_res=64, sky_res=64)
    out.append(("sanmiguel_like_100meshes_240x135_1spp", s, 1))
    return out


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as td:
        res = {name: dict(ref_stats(path, spp), spp=spp) for name, path, spp in scenes(td)}
    json.dump(res, open(os.path.join(G, "bvh_stats.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))
