#!/bin/bash
# timing experiments: variant builds under pbrt-v4_amd/_exp*/ (not products) beside the product build, same box,
# interleaved twice so that box drift shows
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
[ -n "$SM" ] && python tools/make_scenes.py sanmiguel-like /tmp/sm.pbrt --spp 8 --meshes 1600
for rep in 1 2; do
for d in pbrt-v4_amd/_build pbrt-v4_amd/_exp*; do
  echo "== $d"
  timeout 100 $d/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|${GREP:-Intersect}"
  [ -n "$SM" ] && timeout 300 $d/pbrt_amd --stats --outfile /tmp/sm.pfm /tmp/sm.pbrt 2>&1 | grep -E "Rendering|${GREP:-Intersect}"
done
done
