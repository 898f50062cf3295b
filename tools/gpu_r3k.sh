#!/bin/bash
# round 3: occupancy A/B of the two-level traversal kernels on the spec scene, 16 spp: closest-hit at 3 / 4 (= _build) / 5 waves per SIMD
# (_exp_tw3, _exp_tw5), any-hit at 4 instead of 5 (_exp_ts4); media + instances parity
mkdir -p gpurun_out
export TMPDIR=/tmp
GREP="Intersect|Route" bash tools/gpu_sm16.sh > gpurun_out/r3k_ab_sm16.txt 2>&1
cat gpurun_out/r3k_ab_sm16.txt
timeout 900 python -m pytest tests -q -m gpu -k "media" 2>&1 | tail -3 | tee gpurun_out/r3k_pytest_media.txt
