#!/bin/bash
# quick GPU check: parity tests + per-kernel profile of the killeroo-like scene
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -s 2>&1 | grep -E "passed|failed|frac over|Error|assert" | tee gpurun_out/pytest_gpu.txt
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt > gpurun_out/killeroo_stats.txt 2>&1
grep -E "Rendering|launches|Total" gpurun_out/killeroo_stats.txt
python tools/make_scenes.py materials-lights /tmp/ml.pbrt --res 1920 1080 --spp 8
timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/ml.pfm /tmp/ml.pbrt 2>&1 | grep -E "Rendering|launches|Total"
for e in "$@"; do echo "== $e"; env $e timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Intersect"; done
