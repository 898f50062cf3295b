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
WF_PROPS = os.path.join(ROOT, "oracle", "_build", "wf_props")
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
    need = [os.path.join(ROOT, "pbrt-v4_amd", "_build", "libwfhip.so"), os.path.join(ROOT, "pbrt-v4_amd", "_build", "libwfhost.so"), WF_CPU, WF_PROBE, WF_PROPS]
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


def read_exr_channels(path):
    """Minimal OpenEXR reader for the tests: single-part scan-line files WITHOUT compression (what the oracle build's OpenEXR stand-in
    and the product's multi-channel writer produce) -> {channel name: float32 array [h][w]} (half channels are widened)."""
    import struct
    b = open(path, "rb").read()
    assert struct.unpack_from("<I", b, 0)[0] == 20000630, "not an OpenEXR file"
    pos = 8
    chans, dw, comp = [], None, None
    while b[pos] != 0:
        e = b.index(b"\0", pos); name = b[pos:e].decode(); pos = e + 1
        e = b.index(b"\0", pos); pos = e + 1
        size = struct.unpack_from("<i", b, pos)[0]; pos += 4
        v = b[pos:pos + size]; pos += size
        if name == "channels":
            p = 0
            while v[p] != 0:
                e = v.index(b"\0", p); cn = v[p:e].decode(); p = e + 1
                chans.append((cn, struct.unpack_from("<i", v, p)[0])); p += 16
        elif name == "dataWindow":
            dw = struct.unpack("<4i", v)
        elif name == "compression":
            comp = v[0]
    pos += 1
    assert comp == 0 and dw is not None, "only uncompressed files"
    w, h = dw[2] - dw[0] + 1, dw[3] - dw[1] + 1
    out = {cn: np.zeros((h, w), np.float32) for cn, _ in chans}
    offs = struct.unpack_from("<%dQ" % h, b, pos)
    for off in offs:
        y, nbytes = struct.unpack_from("<ii", b, off)
        p = off + 8
        for cn, ty in chans:
            if ty == 1:
                out[cn][y - dw[1]] = np.frombuffer(b, dtype="<f2", count=w, offset=p).astype(np.float32); p += 2 * w
            else:
                out[cn][y - dw[1]] = np.frombuffer(b, dtype="<f4", count=w, offset=p); p += 4 * w
    return out
