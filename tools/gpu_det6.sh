#!/bin/bash
export TMPDIR=/tmp
d=/tmp/wfbench_sm
mkdir -p $d gpurun_out
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
{
for cfg in "WF_X=0" "WF_HOST_BVH_BUILD=1" "WF_PIXEL_MAJOR=0"; do
  bad=0
  for k in 1 2 3 4 5 6 7 8 9 10; do
    env $cfg timeout 150 pbrt-v4_amd/_build/pbrt_amd --quiet --spp 16 --outfile /tmp/det.pfm $d/sm.pbrt > /tmp/out.txt 2>&1 || bad=$((bad+1))
  done
  echo "$cfg: $bad of 10 runs ended with an unresolved re-walk"
done
} 2>&1 | tee gpurun_out/det6.txt
