"""The production traversal layout, checked on the HOST (no GPU): wf_debug_fastbvh_check (include/wf_abi.h) builds the QNode / LeafTri /
instance-entry arrays wf_scene_upload would upload — since round 6 with the top-level tree rebuilt over partially re-braided instances
(wf_traverse.h, SubEntry) — and walks them with random rays in double arithmetic.  The property the tree owes: every triangle a ray
really hits (brute force over the top-level triangles and every (instance, triangle) pair) is among the triangles the walk tests.
The hits themselves are decided by the exact triangle test on the device and pinned bit for bit by the GPU suite's goldens."""
import os

import pytest

from conftest import GOLDEN

INSTANCE_SCENES = ["instances", "instances_quadrics", "media_instances", "animated"]


def check(wfpt, path, braid, n_rays=48, seed=3, **env):
    old = {k: os.environ.get(k) for k in ["WF_BRAID"] + list(env)}
    os.environ["WF_BRAID"] = str(braid)
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        s = wfpt.Scene(path=path)
        try:
            return s.fastbvh_check(n_rays, seed)
        finally:
            s.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("name", INSTANCE_SCENES)
@pytest.mark.parametrize("braid", [0, 2, 8, 64])
def test_golden_instance_scenes_covered(wfpt, name, braid):
    path = os.path.join(GOLDEN, name + ".pbrt")
    if not os.path.exists(path):
        pytest.skip("no such golden scene")
    r = check(wfpt, path, braid)
    assert r["missed"] == 0, r
    assert r["qnodes"] > 0 and r["leaf_records"] > 0


def test_rebraiding_opens_instances_and_stays_a_superset(wfpt, tmp_path):
    """a two-level scene with rotated, scaled clusters (the bench stand-in's generator at 1/20 size): more entries than instances, nothing missed,
    and the same with one entry per instance (WF_BRAID=0: the reference's own top-level tree)"""
    import make_scenes
    p = str(tmp_path / "sm.pbrt")
    make_scenes.sanmiguel_like(p, (64, 36), 1, n_meshes=100, n_defs=10, n_emitters=10, tex_res=16, sky_res=16)
    r0 = check(wfpt, p, 0, n_rays=64)
    r8 = check(wfpt, p, 8, n_rays=64)
    r64 = check(wfpt, p, 64, n_rays=64, WF_BRAID_MIN_FRAC=0)
    assert r0["entries"] == 95 and r8["entries"] > 4 * 95 and r64["entries"] > r8["entries"], (r0, r8, r64)
    for r in (r0, r8, r64):
        assert r["true_hits"] > 20 and r["missed"] == 0, r
    # the same rays, the same triangles really hit
    assert r0["true_hits"] == r8["true_hits"] == r64["true_hits"]


def test_scene_without_instances_keeps_the_reference_topology(wfpt):
    """no instances: nothing to re-braid — the production tree is the reference's tree collapsed four-wide, whatever WF_BRAID says"""
    path = os.path.join(GOLDEN, "cornell64.pbrt")
    a, b = check(wfpt, path, 0), check(wfpt, path, 8)
    assert a == b and a["entries"] == 0 and a["missed"] == 0 and a["true_hits"] > 0
