#!/bin/bash
# material-kernel timing experiments on the spec scene: product build vs variant builds under pbrt-v4_amd/_exp*/
mkdir -p /tmp/sm /tmp/tab
python tools/make_scenes.py sanmiguel-like /tmp/sm/sm.pbrt --spp 8 > /dev/null
export WF_TABLE_CACHE=/tmp/tab
for d in pbrt-v4_amd/_build pbrt-v4_amd/_exp*; do
  echo "== $d"
  timeout 200 $d/pbrt_amd --stats --outfile /tmp/sm.pfm /tmp/sm/sm.pbrt 2>&1 | grep -E "Rendering|Material|Total GPU|Intersect shadow  "
done
