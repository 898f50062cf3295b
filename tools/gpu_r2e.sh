#!/bin/bash
# A/B of the product build against pbrt-v4_amd/_exp_old (a build of an earlier commit) on the spec scene, full per-kernel tables,
# then the GPU parity tests of the goldens added last
mkdir -p gpurun_out
export TMPDIR=/tmp
GREP="launches|Total GPU" bash tools/gpu_sm16.sh > gpurun_out/ab_full.txt 2>&1
cat gpurun_out/ab_full.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "realistic or curves or ewa or png" 2>&1 | tail -4 | tee gpurun_out/pytest_new.txt
