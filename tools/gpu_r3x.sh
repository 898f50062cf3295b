#!/bin/bash
# round 3 (re-entry): whole GPU suite on the rebuilt head (new goldens: film_sensor, displacement, plymesh_mixed; adapter with quadrics / patches / curves)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r3x_pytest_gpu.txt 2>&1; tail -15 gpurun_out/r3x_pytest_gpu.txt
