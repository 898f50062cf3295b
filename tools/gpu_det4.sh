#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/wfbench_sm
mkdir -p $d
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
for k in 1 2 3 4 5 6; do
  rm -f /tmp/det.pfm
  WF_DEBUG_DRAIN=1 timeout 150 pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/det.pfm $d/sm.pbrt > /tmp/out_$k.txt 2>&1; echo "rc=$?"
  grep -E "rror|Indirect rays, depth 1 |Rendering" /tmp/out_$k.txt | tr -s ' ' | cut -c1-200; grep "drain" /tmp/out_$k.txt | tail -1 | cut -c1-150
  sha1sum /tmp/det.pfm | cut -c1-8
done 2>&1 | tee gpurun_out/det4_sm16.txt
