#!/bin/bash
# round 3: whole GPU suite on the head build, then the driver's bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r3w_pytest_gpu.txt 2>&1; tail -6 gpurun_out/r3w_pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3w_bench_err.txt | tee gpurun_out/r3w_bench_k20.json | cut -c1-1500
