#!/bin/bash
# tools/gpu_round.sh — one GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace of the bench.
# Outputs go to gpurun_out/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.txt
timeout 900 python bench.py --steps ${STEPS:-64} --warmup 2 2>gpurun_out/bench_err.txt | tee gpurun_out/bench.json
tail -5 gpurun_out/bench_err.txt
if [ "${PROF:-1}" = "1" ]; then
  rm -rf /tmp/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 1 --cpu-spp 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/rocprof_err.txt)
  find /tmp/prof -name "*stats*" | head
  for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/rocprof_kernel_stats.csv; done
  head -30 gpurun_out/rocprof_kernel_stats.csv
fi
