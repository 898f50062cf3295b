import os, sys, subprocess
import numpy as np
sys.path.insert(0, "tests")
from conftest import load_pkg, GOLDEN, WF_CPU
from test_gpu_parity import _random_rays
wfpt = load_pkg(); wfpt.libs()
name = sys.argv[1] if len(sys.argv) > 1 else "instances"
path = os.path.join(GOLDEN, name + ".pbrt")
s = wfpt.Scene(path=path, spp=4); s.create_renderer(0)
n = 20000
lo, hi = s.bounds(); pad = 0.1 * (hi - lo)
o, d, tmax = _random_rays(n, lo - pad, hi + pad, 7)
ref = s.trace_closest(o, d, tmax)
fast = s.trace_closest(o, d, tmax, reference_order=False)
bad = np.where((fast["prim"] != ref["prim"]) | (fast["instance"] != ref["instance"]))[0]
print("mismatches", len(bad), "of", n, "bounds", lo, hi)
for i in bad[:12]:
    print(i, "ref prim %d inst %d t %.9g | fast prim %d inst %d t %.9g  dt/t %.3g tmax %.4g" % (ref["prim"][i], ref["instance"][i], ref["t"][i], fast["prim"][i], fast["instance"][i], fast["t"][i],
          (fast["t"][i] - ref["t"][i]) / max(ref["t"][i], 1e-9), tmax[i]), "o", o[i], "d", d[i])
