#!/bin/bash
# round 3, first GPU session: parity tests (incl. the big-tree and benchmark-stand-in tests), A/B of the ring stack + inline
# near-tie resolution against the round-2 build (pbrt-v4_amd/_exp_old) on the spec scene, FETCH_SIZE / WRITE_SIZE of the new
# traversal kernels, and the driver's bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r3a_pytest_gpu.txt
GREP="Intersect|Route|Total GPU|BxDF|launches" bash tools/gpu_sm16.sh > gpurun_out/r3a_ab_sm16.txt 2>&1
cat gpurun_out/r3a_ab_sm16.txt
SCN=/tmp/wfbench_sm/sm.pbrt
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o k -- $GRAFT_REPO_ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --spp 16 --outfile /tmp/k.pfm $SCN > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { tail -3 /tmp/pmc_$c.log | cut -c1-200; continue; }
  python3 - "$f" $c <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/r3a_pmc_fetch_write_16spp.txt
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
pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/sm.pfm $SCN 2>&1 | grep -E "Camera rays|Indirect rays|Shadow rays" | tee -a gpurun_out/r3a_pmc_fetch_write_16spp.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>gpurun_out/r3a_bench_err.txt | tee gpurun_out/r3a_bench.json
tail -5 gpurun_out/r3a_bench_err.txt
