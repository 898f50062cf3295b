#!/bin/bash
# A/B: pixel-major vs sample-major item order on the spec scene (16 spp), per-kernel times; then the GPU suite on the default (pixel-major)
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/wfbench_sm
mkdir -p $d
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
pbrt-v4_amd/_build/pbrt_amd --quiet --spp 4 --outfile /tmp/sm.pfm $d/sm.pbrt > /dev/null 2>&1
for pm in 0 1 0 1; do
  echo "== WF_PIXEL_MAJOR=$pm"
  WF_PIXEL_MAJOR=$pm timeout 150 pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/sm_$pm.pfm $d/sm.pbrt 2>&1 | grep -E "Rendering|Intersect|Total GPU|Evaluate|Generate|Handle|Update"
done 2>&1 | tee gpurun_out/pm_ab_sm16.txt
cmp /tmp/sm_0.pfm /tmp/sm_1.pfm && echo "images identical" | tee -a gpurun_out/pm_ab_sm16.txt
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/pm_pytest_gpu.txt 2>&1; tail -4 gpurun_out/pm_pytest_gpu.txt
