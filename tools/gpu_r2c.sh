#!/bin/bash
# round-2 measurement session on the north_star config: bench line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python bench.py --steps ${STEPS:-64} --warmup 1 --cpu-spp ${CPUSPP:-0} 2>gpurun_out/bench_err.txt | tee gpurun_out/bench.json
tail -2 gpurun_out/bench_err.txt
rm -rf /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-64} --warmup 1 --cpu-spp 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> /tmp/rocprof_err.txt)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/rocprof_kernel_stats.csv; done
head -14 gpurun_out/rocprof_kernel_stats.csv | cut -c1-200
d=$(ls -d /tmp/wfbench_sanmiguel-like_* | head -1)
rm -f gpurun_out/pmc_traffic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o k -- $GRAFT_REPO_ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --spp 16 --stats --outfile /tmp/k4.pfm $d/sanmiguel-like.pbrt > /tmp/pmc_$c.log 2>&1)
  grep -E "Total rays|Camera rays|Indirect rays|Shadow rays" /tmp/pmc_$c.log | head -14 > gpurun_out/pmc_rays_$c.txt
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY' | tee -a gpurun_out/pmc_traffic.txt
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in sorted(agg, key=lambda k: -agg[k])[:12]:
    print("%s %-60s dispatches %4d  mean per dispatch %12.1f KiB  total %14.1f KiB" % (sys.argv[2], k[-60:], len(cnt[k]), agg[k] / len(cnt[k]), agg[k]))
PY
done
cat gpurun_out/pmc_rays_FETCH_SIZE.txt
