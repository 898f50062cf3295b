#!/bin/bash
# round 3: A/B of the parked instance transitions (LeafPhase: _build 12 lanes, _exp_tb6, _exp_tb24, _exp_tb0 = run at once) on the spec
# scene 16 spp, then parity of the instanced scenes
mkdir -p gpurun_out
export TMPDIR=/tmp
GREP="Intersect|Route" bash tools/gpu_sm16.sh > gpurun_out/r3f_ab_sm16.txt 2>&1
cat gpurun_out/r3f_ab_sm16.txt
timeout 900 python -m pytest tests -q -m gpu -k "instances or big_two or full_wavefront or benchmark_standins or curves or strip" 2>&1 | tail -5 | tee gpurun_out/r3f_pytest_gpu.txt
