#!/bin/bash
# round 3: the boundary — the reference's integrator over HipAggregate on instanced / alpha / media scenes, device-pointer entry points —
# then the whole GPU suite
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "hip_aggregate or device_pointer" 2>&1 | tail -25 | tee gpurun_out/r3l_pytest_boundary.txt
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r3l_pytest_gpu.txt
