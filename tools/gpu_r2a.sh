#!/bin/bash
# round-2 first GPU session: strict parity tests (no -x: list every failure), smoke, short bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "passed|failed|max rel|FAILED|Error|assert" | tee gpurun_out/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.txt
timeout 600 python bench.py --steps 64 --warmup 2 2>gpurun_out/bench_err.txt | tee gpurun_out/bench.json
tail -3 gpurun_out/bench_err.txt
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt > gpurun_out/killeroo_stats.txt 2>&1
grep -E "Rendering|launches|Total|ms" gpurun_out/killeroo_stats.txt | head -40
