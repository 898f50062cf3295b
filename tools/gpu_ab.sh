#!/bin/bash
# A/B of traversal variants (env knobs) on the same box: killeroo-like 16 spp, per-kernel profile
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
for e in "$@"; do
  echo "== $e"
  env $e timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Intersect|Route"
done
