#!/bin/bash
# round 3 head build: the driver's bench command, the 64-step bench under rocprofv3 (kernel statistics), FETCH_SIZE / WRITE_SIZE of the
# traversal kernels on the spec scene at 16 spp, bench lines of the killeroo-like and tm-like stand-ins
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3i_bench_err.txt | tee gpurun_out/r3i_bench_k20.json
rm -rf /tmp/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 2 --cpu-spp 0 > $GRAFT_REPO_ROOT/gpurun_out/r3i_bench_k64_under_rocprof.json 2> /tmp/rocprof_err.txt)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/r3i_bench_k64_rocprofv3_kernel_stats.csv; done
head -14 gpurun_out/r3i_bench_k64_rocprofv3_kernel_stats.csv
timeout 600 python bench.py --steps 64 --warmup 2 --cpu-spp 0 2>/dev/null | tee gpurun_out/r3i_bench_k64.json | cut -c1-400
SCN=$(ls -d /tmp/wfbench_sanmiguel-like_*/sanmiguel-like.pbrt | head -1)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o k -- $GRAFT_REPO_ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --spp 16 --outfile /tmp/k.pfm $SCN > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { tail -3 /tmp/pmc_$c.log | cut -c1-200; continue; }
  python3 - "$f" $c <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/r3i_pmc_fetch_write_16spp.txt
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in sorted(agg, key=lambda k: -agg[k])[:14]:
    print("%s %-60s dispatches %4d  mean per dispatch %12.1f KiB  total %14.1f KiB" % (sys.argv[2], k[-60:], len(cnt[k]), agg[k] / len(cnt[k]), agg[k]))
PY
done
cd $GRAFT_REPO_ROOT
pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/sm.pfm $SCN 2>&1 | grep -E "Camera rays|Indirect rays|Shadow rays" | tee -a gpurun_out/r3i_pmc_fetch_write_16spp.txt
timeout 600 python bench.py --workload killeroo-like --steps 64 --warmup 2 2>/dev/null | tee gpurun_out/r3i_bench_killeroo.json | cut -c1-300
timeout 900 python bench.py --workload tm-like --steps 16 --warmup 1 2>/dev/null | tee gpurun_out/r3i_bench_tm.json | cut -c1-300
