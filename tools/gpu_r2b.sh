#!/bin/bash
# round-2 GPU session: strict parity tests, smoke, bench on the north_star config (sanmiguel-like at spec) and on configs[1]
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "passed|failed|max rel|FAILED|Error|assert" | grep -v "max rel 0.0 bit" | tee gpurun_out/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.txt
timeout 1500 python bench.py --steps ${STEPS:-32} --warmup 1 --cpu-spp ${CPUSPP:-0} 2>gpurun_out/bench_err.txt | tee gpurun_out/bench.json
tail -3 gpurun_out/bench_err.txt
timeout 600 python bench.py --workload killeroo-like --steps 64 --warmup 2 --cpu-spp 0 2>gpurun_out/bench_k_err.txt | tee gpurun_out/bench_killeroo.json
d=$(ls -d /tmp/wfbench_sanmiguel-like_* | head -1)
timeout 600 pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/sm.pfm $d/sanmiguel-like.pbrt > gpurun_out/sanmiguel_stats.txt 2>&1
grep -E "Rendering|launches|Total" gpurun_out/sanmiguel_stats.txt | head -40
