#!/bin/bash
# round 3 (re-entry): bench.py with the live PMC traffic measurement (short run)
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python bench.py --steps 8 --warmup 2 --cpu-spp 0 2>gpurun_out/r3z_bench_err.txt ) 2>&1 | tee gpurun_out/r3z_bench.json | cut -c1-3000
tail -5 gpurun_out/r3z_bench_err.txt
