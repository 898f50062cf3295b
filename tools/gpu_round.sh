#!/bin/bash
# tools/gpu_round.sh — one GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace of the bench,
# HBM-traffic PMC passes.  Outputs go to gpurun_out/ (merged back by gpurun); the summaries worth keeping
# are copied to profiles/ by hand.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/smoke.txt
timeout 600 python bench.py --steps ${STEPS:-64} --warmup 2 2>gpurun_out/bench_err.txt | tee gpurun_out/bench.json
tail -3 gpurun_out/bench_err.txt
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt > gpurun_out/killeroo_stats.txt 2>&1
grep -E "Rendering|launches|Total" gpurun_out/killeroo_stats.txt
if [ "${BIG:-1}" = "1" ]; then
  timeout 900 python bench.py --workload sanmiguel-like --meshes ${MESHES:-1600} --steps 16 --warmup 1 --cpu-spp 0 2>gpurun_out/bench_sm_err.txt | tee gpurun_out/bench_sanmiguel.json
  tail -3 gpurun_out/bench_sm_err.txt
fi
if [ "${PROF:-1}" = "1" ]; then
  rm -rf /tmp/prof
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-64} --warmup 2 --cpu-spp 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> /tmp/rocprof_err.txt)
  for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/rocprof_kernel_stats.csv; done
  head -16 gpurun_out/rocprof_kernel_stats.csv
  python tools/make_scenes.py killeroo-like /tmp/k4.pbrt --spp 4
  rm -f gpurun_out/pmc_traffic.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    (cd /tmp && timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o k -- $GRAFT_REPO_ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --outfile /tmp/k4.pfm /tmp/k4.pbrt > /tmp/pmc_$c.log 2>&1)
    f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" $c <<'PY' | tee -a gpurun_out/pmc_traffic.txt
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    agg[k] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in sorted(agg, key=lambda k: -agg[k]):
    print("%s %-60s dispatches %4d  mean per dispatch %12.1f KiB" % (sys.argv[2], k[-60:], len(cnt[k]), agg[k] / len(cnt[k])))
PY
  done
fi
