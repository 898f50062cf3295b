#!/bin/bash
# round 3: whole GPU suite (SpectralFilm, GBufferFilm, NanoVDB, boundary, ...) on the head build, then the driver's bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r3o_pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3o_bench_err.txt | tee gpurun_out/r3o_bench_k20.json | cut -c1-500
