#!/bin/bash
# round 3: whole GPU suite on the head build (NanoVDB medium, ABI 6, parallel host build), then the driver's bench command (load_s)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r3m_pytest_gpu.txt
WF_LOAD_TIMING=1 timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3m_bench_err.txt | tee gpurun_out/r3m_bench_k20.json | cut -c1-600
grep "\[load\]" gpurun_out/r3m_bench_err.txt | head -20
