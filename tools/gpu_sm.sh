#!/bin/bash
# san-miguel-like stand-in: bench at 16 sample indices per pass + SQ counters of the traversal kernels (2.5 M-triangle variant)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --workload sanmiguel-like --meshes 1600 --steps 16 --warmup 1 --cpu-spp 0 2>/dev/null | tail -1 | tee gpurun_out/bench_sanmiguel16.json
python tools/make_scenes.py sanmiguel-like /tmp/sm.pbrt --spp 8 --meshes 400
cd /tmp
for pass in "sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  set -- $pass; name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o k -- $GRAFT_REPO_ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --outfile /tmp/sm.pfm /tmp/sm.pbrt > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  echo "== $name"
  [ -z "$f" ] && { tail -3 /tmp/pmc_$name.log | cut -c1-200; continue; }
  python3 - "$f" <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/pmc_sm.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in agg:
    if "closest_fast" in k or "shadow_fast" in k:
        print(k, len(cnt[k]), {c: "%.4g" % (v / len(cnt[k])) for c, v in agg[k].items()})
PY
done
