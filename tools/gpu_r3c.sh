#!/bin/bash
# round 3: A/B of the LDS budget of the production walk (ring-stack entries per lane x LDS-cached top nodes) and of the any-hit refill
# threshold, on the spec scene and the killeroo-like scene; debug counters of the big-tree tests; which goldens have near-ties
mkdir -p gpurun_out
export TMPDIR=/tmp
GREP="Intersect|Route" bash tools/gpu_sm16.sh > gpurun_out/r3c_ab_sm16.txt 2>&1
cat gpurun_out/r3c_ab_sm16.txt
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
for b in pbrt-v4_amd/_build pbrt-v4_amd/_exp*; do
  echo "== killeroo $b"
  timeout 120 $b/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Intersect|Route"
done 2>&1 | tee gpurun_out/r3c_ab_killeroo.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "big_two_level or full_wavefront" 2>&1 | grep -E "debug counters|passed|failed|Error" | tee gpurun_out/r3c_pytest_big.txt
python - <<'PY' 2>&1 | tee gpurun_out/r3c_ties.txt
import glob, os, sys
sys.path.insert(0, "tests")
from conftest import load_pkg
wfpt = load_pkg()
for p in sorted(glob.glob("tests/golden/*.pbrt")):
    try:
        s = wfpt.Scene(path=p, spp=4); s.create_renderer(0); s.debug_counters(True); s.render(); d = s.debug_counters(True); s.close()
        if d["inline_retraces"] or d["spilled_entries"]: print(os.path.basename(p), d)
    except Exception as e:
        print(os.path.basename(p), "ERR", str(e)[:80])
PY
