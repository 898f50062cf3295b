#!/bin/bash
export TMPDIR=/tmp
d=/tmp/wfbench_sm
mkdir -p $d gpurun_out
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
{
bad=0
for k in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  WF_DEBUG_DRAIN=1 timeout 150 pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/det.pfm $d/sm.pbrt > /tmp/out.txt 2>&1 || bad=$((bad+1))
  echo "$(grep 'Indirect rays, depth 1 ' /tmp/out.txt | tr -s ' ') $(grep drain /tmp/out.txt | tail -1 | cut -c1-110) $(sha1sum /tmp/det.pfm | cut -c1-8)"
done
echo "failed runs: $bad of 14"
} 2>&1 | tee gpurun_out/det7.txt
