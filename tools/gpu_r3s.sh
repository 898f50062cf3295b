#!/bin/bash
# round 3: the closest-hit walk with refill + in-kernel near-tie drain: parity subset first (a hang must not eat the budget), then the A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "envmap or big or blobs_small or instances or cornell64 or alpha_normalmap or sanmiguel or stage or fused" > gpurun_out/r3s_pytest.txt 2>&1; grep -v "^  File\|^Extension" gpurun_out/r3s_pytest.txt | tail -8
timeout 400 bash tools/gpu_sm16.sh 2>&1 | tee gpurun_out/r3s_ab_sm16.txt
