#!/bin/bash
# near-tie queue diagnostics: pushes vs re-walks vs re-walks that found nothing, next to the ray counts (spec scene, 16 spp, 8 renders)
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/wfbench_sm
mkdir -p $d
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
for k in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  WF_DEBUG_DRAIN=1 timeout 150 pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/det.pfm $d/sm.pbrt 2>&1 | grep -E "Indirect rays, depth 1 |\[drain\]" | tr -s ' ' | cut -c1-260 | tail -2 | tr '\n' ';'
  sha1sum /tmp/det.pfm | cut -c1-8
done 2>&1 | tee gpurun_out/det3_sm16.txt
