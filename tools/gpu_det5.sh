#!/bin/bash
export TMPDIR=/tmp
d=/tmp/wfbench_sm
mkdir -p $d gpurun_out
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
for k in 1 2 3 4 5 6 7 8; do
  WF_DEBUG_DRAIN=1 timeout 150 pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/det.pfm $d/sm.pbrt > /tmp/out_$k.txt 2>&1; rc=$?
  if [ $rc != 0 ]; then echo "== run $k rc=$rc"; grep -v "^\[load\]" /tmp/out_$k.txt | head -30 | cut -c1-300; break; fi
done 2>&1 | tee gpurun_out/det5.txt
