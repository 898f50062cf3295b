#!/bin/bash
# round 3 (re-entry): the new goldens on the GPU (camera motion blur: k_gen_camera_rays<true>, material variant 2 with the moving camera)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -k "camera_motion or film_sensor or displacement or plymesh_mixed or abi or cornell64" > gpurun_out/r3y_pytest_gpu.txt 2>&1; tail -15 gpurun_out/r3y_pytest_gpu.txt
