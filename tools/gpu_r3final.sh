#!/bin/bash
# round 3 final: whole GPU suite, the driver's bench command (with the live PMC traffic and the parity block), the 64-step bench under rocprofv3 --kernel-trace --stats
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu > gpurun_out/r3f_pytest_gpu.txt 2>&1; tail -5 gpurun_out/r3f_pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3f_bench_err.txt | tee gpurun_out/r3f_bench_k20.json | cut -c1-600
rm -rf /tmp/prof
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 2 --cpu-spp 0 --pmc-spp 0 > $GRAFT_REPO_ROOT/gpurun_out/r3f_bench_k64_under_rocprof.json 2> /tmp/rocprof_err.txt)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/r3f_bench_k64_rocprofv3_kernel_stats.csv; done
head -8 gpurun_out/r3f_bench_k64_rocprofv3_kernel_stats.csv | cut -c1-200
cut -c1-300 gpurun_out/r3f_bench_k64_under_rocprof.json
