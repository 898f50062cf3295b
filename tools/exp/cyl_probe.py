import sys, os, numpy as np
sys.path.insert(0, "/root/repo/tests")
from conftest import load_pkg
wfpt = load_pkg(); wfpt.libs()
s = wfpt.Scene(path="/root/repo/tests/golden/fuzz/finding_s200010_min.pbrt", spp=1)
s.create_renderer(0)
rng = np.random.default_rng(1)
n = 400000
phi = rng.random(n) * 2 * np.pi
z = rng.random(n) * 2.4 - 1.2
r = 1 + (rng.random(n) - 0.5) * 4e-6 * (rng.random(n) < 0.7)
o = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
# the scene's render space is camera-world: translate by the camera position? use bounds to check
print("bounds", s.bounds())
b0, b1 = s.bounds()
c = (b0 + b1) / 2
o = (o + c * np.array([1, 1, 0], np.float32) + np.array([0, 0, 0], np.float32)).astype(np.float32)
tmax = np.full(n, np.inf, np.float32)
ref = s.trace_closest(o, d, tmax, reference_order=True)
fast = s.trace_closest(o, d, tmax, reference_order=False)
bad = np.nonzero((ref["prim"] != fast["prim"]) | (ref["t"].view(np.uint32) != fast["t"].view(np.uint32)))[0]
print("retraced", (fast["nodes_visited"] == 7).sum(), "of mismatches retraced", (fast["nodes_visited"][bad] == 7).sum())
print("hits", (ref["prim"] >= 0).mean(), "mismatches", len(bad))
for i in bad[:8]:
    print(i, o[i], d[i], "ref", ref["prim"][i], ref["t"][i], "fast", fast["prim"][i], fast["t"][i], "nv", fast["nodes_visited"][i])
