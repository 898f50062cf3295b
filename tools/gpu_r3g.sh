#!/bin/bash
# round 3: A/B of v_rcp_f32 for the slab constants (_build) against IEEE divisions (_exp_norcp) and round 2 (_exp_old): spec scene and
# killeroo-like scene, 16 spp; then the whole GPU test suite
mkdir -p gpurun_out
export TMPDIR=/tmp
GREP="Intersect|Route" bash tools/gpu_sm16.sh > gpurun_out/r3g_ab_sm16.txt 2>&1
cat gpurun_out/r3g_ab_sm16.txt
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
for b in pbrt-v4_amd/_build pbrt-v4_amd/_build pbrt-v4_amd/_exp*; do
  echo "== killeroo $b"
  timeout 120 $b/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Intersect|Route"
done 2>&1 | tee gpurun_out/r3g_ab_killeroo.txt
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r3g_pytest_gpu.txt
