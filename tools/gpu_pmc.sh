#!/bin/bash
# PMC counters for the traversal kernels on the killeroo-like scene (separate passes, no tracing domains
# other than --kernel-trace)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 4
cd /tmp
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  WF_REFILL_BOUNCE=1 WF_REFILL_SHADOW=1 timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o k -- $GRAFT_REPO_ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --outfile /tmp/k.pfm /tmp/k.pbrt > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  echo "== $name: $f"
  python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r.get("Dispatch_Id"))
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in agg:
    if "closest" in k or "shadow" in k or "eval_material" in k:
        print(k, cnt[k], {c: "%.4g" % (v / cnt[k]) for c, v in agg[k].items()})
PY
  cp "$f" $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.csv 2>/dev/null
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
