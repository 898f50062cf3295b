#!/bin/bash
# per-kernel times of the sanmiguel-like scene at spec, 16 spp, one run per environment setting given as arguments
export TMPDIR=/tmp
d=/tmp/wfbench_sm
mkdir -p $d
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
for e in "$@"; do
  echo "== $e"
  env $e timeout 150 pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/sm.pfm $d/sm.pbrt 2>&1 | grep -E "Rendering|${GREP:-Intersect|Total GPU}"
done
