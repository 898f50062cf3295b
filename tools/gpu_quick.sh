#!/bin/bash
# quick GPU check: parity tests + per-kernel profile of the killeroo-like scene
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.txt
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt > gpurun_out/killeroo_stats.txt 2>&1
grep -E "Rendering|launches|Total" gpurun_out/killeroo_stats.txt
for e in "$@"; do echo "== $e"; env $e timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Intersect"; done
