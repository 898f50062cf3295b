#!/bin/bash
# per-kernel times of the killeroo-like scene at 16 spp (quick A/B of traversal changes)
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
for i in 1 2; do timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Intersect|Total GPU"; done
