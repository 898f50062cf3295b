#!/bin/bash
# round 3: A/B of the wave-level ray replacement (BatchTraceRefill) — _build = refill at <= 24 active lanes, _exp_r12 / _exp_r40 other
# thresholds, _exp_r0 = no refill (ring stack + inline near-tie resolution only), _exp_old = round-2 build — on the spec scene and on the
# cache-resident killeroo-like scene; then the parity tests
mkdir -p gpurun_out
export TMPDIR=/tmp
GREP="Intersect|Route|Total GPU" bash tools/gpu_sm16.sh > gpurun_out/r3b_ab_sm16.txt 2>&1
cat gpurun_out/r3b_ab_sm16.txt
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
for b in pbrt-v4_amd/_build pbrt-v4_amd/_exp*; do
  echo "== killeroo $b"
  timeout 120 $b/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Intersect|Route"
done 2>&1 | tee gpurun_out/r3b_ab_killeroo.txt
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r3b_pytest_gpu.txt
