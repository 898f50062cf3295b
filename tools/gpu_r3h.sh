#!/bin/bash
# round 3: BASELINE configs[3] at the SURVEY 8(d) spec — the cloud-like stand-in with a 512^3 density grid: bench line with the per-stage
# breakdown, and the rocprofv3 kernel statistics of the same command
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python bench.py --workload cloud-like --steps 16 --warmup 2 --breakdown 2>gpurun_out/r3h_bench_cloud_err.txt | tee gpurun_out/r3h_bench_cloud_512.json
tail -3 gpurun_out/r3h_bench_cloud_err.txt
rm -rf /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload cloud-like --steps 16 --warmup 2 --cpu-spp 0 > $GRAFT_REPO_ROOT/gpurun_out/r3h_bench_cloud_under_rocprof.json 2> /tmp/rocprof_err.txt)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/r3h_cloud_rocprofv3_kernel_stats.csv; done
head -14 gpurun_out/r3h_cloud_rocprofv3_kernel_stats.csv
