#!/bin/bash
# one GPU-box session after the coverage additions of round 2 (PNG / EXR, EWA, curves, alpha-masked emitters): the full parity
# suite, smoke, and the default bench (the spec-level san-miguel-like scene) to check that the hot kernels did not move
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/smoke.txt
timeout 900 python bench.py --steps ${STEPS:-64} --warmup 1 2>gpurun_out/bench_err.txt | tee gpurun_out/bench.json
tail -3 gpurun_out/bench_err.txt
