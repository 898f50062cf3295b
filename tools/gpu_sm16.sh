#!/bin/bash
# per-kernel times of the sanmiguel-like scene at spec, 16 spp, for every build under pbrt-v4_amd/_build and _exp*
export TMPDIR=/tmp
d=/tmp/wfbench_sm
mkdir -p $d
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
# (the first run on a fresh box is 5-10 % slow — clocks, page faults of the scene files —: one untimed run first)
pbrt-v4_amd/_build/pbrt_amd --quiet --spp 4 --outfile /tmp/sm.pfm $d/sm.pbrt > /dev/null 2>&1
for b in pbrt-v4_amd/_build pbrt-v4_amd/_exp*; do
  [ -x $b/pbrt_amd ] || continue
  echo "== $b"
  timeout 150 $b/pbrt_amd --stats --spp 16 --outfile /tmp/sm.pfm $d/sm.pbrt 2>&1 | grep -E "Rendering|${GREP:-Intersect|Total GPU}"
done
