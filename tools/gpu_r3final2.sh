#!/bin/bash
# round 3 final: bench lines (with parity blocks) of the other three stand-ins on the head build
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python bench.py --workload killeroo-like --steps 64 --warmup 2 --pmc-spp 0 2>/dev/null | tee gpurun_out/r3f_bench_killeroo.json | cut -c1-200
timeout 300 python bench.py --workload cloud-like --steps 16 --warmup 2 --pmc-spp 0 2>/dev/null | tee gpurun_out/r3f_bench_cloud.json | cut -c1-200
timeout 400 python bench.py --workload tm-like --steps 16 --warmup 1 --pmc-spp 0 2>/dev/null | tee gpurun_out/r3f_bench_tm.json | cut -c1-200
