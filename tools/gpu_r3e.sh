#!/bin/bash
# round 3: A/B of the LDS-resident render-space ray (_exp_nopre) and the conservative instance pre-test on top of it (_build) against the
# build before both (_exp_prev) and the round-2 build (_exp_old), spec scene 16 spp; parity tests of the instanced scenes
mkdir -p gpurun_out
export TMPDIR=/tmp
GREP="Intersect|Route" bash tools/gpu_sm16.sh > gpurun_out/r3e_ab_sm16.txt 2>&1
cat gpurun_out/r3e_ab_sm16.txt
timeout 900 python -m pytest tests -q -m gpu -k "instances or big_two or full_wavefront or benchmark_standins or curves or strip" 2>&1 | tail -5 | tee gpurun_out/r3e_pytest_gpu.txt
