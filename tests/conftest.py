import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
WF_CPU = os.path.join(ROOT, "oracle", "_build", "wf_cpu")
WF_PROBE = os.path.join(ROOT, "oracle", "_build", "wf_probe")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    spec = importlib.util.spec_from_file_location("wfpt", os.path.join(ROOT, "pbrt-v4_amd", "wfpt.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="session")
def built():
    """In-tree build of the product libraries and of the CPU checker (no GPU needed: hipcc cross-compiles)."""
    need = [os.path.join(ROOT, "pbrt-v4_amd", "_build", "libwfhip.so"), os.path.join(ROOT, "pbrt-v4_amd", "_build", "libwfhost.so"), WF_CPU, WF_PROBE]
    if not all(os.path.exists(p) for p in need):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def wfpt(built):
    m = load_pkg()
    m.libs()
    return m


def run_wf_cpu(scene_path, out_pfm, spp=None, extra=()):
    cmd = [WF_CPU, "--quiet", "--outfile", out_pfm]
    if spp:
        cmd += ["--spp", str(spp)]
    cmd += list(extra) + [scene_path]
    p = subprocess.run(cmd, check=True, capture_output=True, text=True)
    import json
    return json.loads(p.stdout.strip().splitlines()[-1])


def read_pfm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        assert magic in (b"PF", b"Pf")
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        nc = 3 if magic == b"PF" else 1
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, nc)
    return data[::-1].astype(np.float32)


def image_error(a, b, floor=1e-2):
    """relative error per value, relative to max(|b|, floor)"""
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return d / np.maximum(np.abs(b), floor)


def bench_small_scene(name, tmp_dir):
    """Regenerate a downscaled benchmark stand-in (tools/make_scenes.py bench_small: the generators of bench.py's workloads with
    fewer meshes) into tmp_dir and check the tree against the hash recorded when its golden render was made."""
    import json
    import make_scenes
    path = make_scenes.bench_small(name, str(tmp_dir))
    want = json.load(open(os.path.join(GOLDEN, "bench_small_hashes.json")))[name]
    assert make_scenes.tree_hash(str(tmp_dir)) == want, "the generator of %s drifted from the one its golden was rendered from" % name
    return path, make_scenes.BENCH_SMALL[name]["spp"]
