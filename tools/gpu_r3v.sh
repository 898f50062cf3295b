#!/bin/bash
# round 3: NEE requests regrouped by light class (material kernels): parity subset, then A/B against the build without it
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "materials_lights or cornell64 or subsurface or hair or measured or media_box or portal or sanmiguel or gbuffer or mix" > gpurun_out/r3v_pytest.txt 2>&1; grep -v "^  File\|^Extension" gpurun_out/r3v_pytest.txt | tail -8
GREP="Intersect|Total GPU|Material" timeout 400 bash tools/gpu_sm16.sh 2>&1 | tee gpurun_out/r3v_ab_sm16.txt
