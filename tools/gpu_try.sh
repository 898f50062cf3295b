#!/bin/bash
# quick correctness + perf check of a kernel change: GPU parity suite, then env-variant A/B on killeroo-like 16 spp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
for e in "$@"; do
  echo "== $e"
  env $e timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|launches|Total GPU"
done
if [ -n "$SM" ]; then
  timeout 600 python bench.py --workload sanmiguel-like --meshes 1600 --steps 8 --warmup 1 --cpu-spp 0 2>/dev/null | tail -1
fi
