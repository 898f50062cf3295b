#!/bin/bash
# round 3: load time of the spec scene with the parallel host build on the GPU box's cores; A/B of the material kernels with
# out-of-line libm entry points (_exp_nilibm) on the spec scene, 16 spp
mkdir -p gpurun_out
export TMPDIR=/tmp
GREP="BxDF|Rendering" bash tools/gpu_sm16.sh > gpurun_out/r3n_ab_sm16.txt 2>&1
cat gpurun_out/r3n_ab_sm16.txt
WF_LOAD_TIMING=1 pbrt-v4_amd/_build/pbrt_amd --quiet --spp 1 --outfile /tmp/sm.pfm /tmp/wfbench_sm/sm.pbrt 2>&1 | grep "\[load\]" | tee gpurun_out/r3n_load_timing.txt
nproc
