#!/bin/bash
# which part of the closest-hit launch loses rays?  variants: no service workgroups (workers drain the near-tie queue), no refill (batch loop + inline re-walk)
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/wfbench_sm
mkdir -p $d
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
{
for b in _exp_inl _exp_nosvc; do
  echo "== $b"
  for k in 1 2 3 4 5 6 7 8; do
    timeout 150 pbrt-v4_amd/$b/pbrt_amd --stats --spp 16 --outfile /tmp/det.pfm $d/sm.pbrt 2>&1 | grep -E "Indirect rays, depth 1 |Intersect closest" | tr -s ' ' | cut -c1-90 | tr '\n' ';'
    sha1sum /tmp/det.pfm | cut -c1-8
  done
done
} 2>&1 | tee gpurun_out/det2_sm16.txt
