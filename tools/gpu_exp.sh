#!/bin/bash
# timing experiments: variant builds under pbrt-v4_amd/_exp*/ (not products; parts of a kernel compiled out)
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
for d in pbrt-v4_amd/_build pbrt-v4_amd/_exp*; do
  echo "== $d"
  timeout 120 $d/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Diffuse|Intersect shadow|Intersect closest"
done
