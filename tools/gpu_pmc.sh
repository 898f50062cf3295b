#!/bin/bash
# PMC counters for the traversal kernels on the killeroo-like scene (separate passes; --kernel-trace only)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
if [ "${SCENE:-killeroo}" = "sanmiguel" ]; then
  mkdir -p /tmp/wfbench_sm
  [ -f /tmp/wfbench_sm/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like /tmp/wfbench_sm/sm.pbrt --spp 16 > /dev/null
  SCN=/tmp/wfbench_sm/sm.pbrt
  EXTRA_ARGS="--spp ${SPP:-4}"
else
  python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp ${SPP:-16}
  SCN=/tmp/k.pbrt
  EXTRA_ARGS=""
fi
cd /tmp
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o k -- $GRAFT_REPO_ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet $EXTRA_ARGS --outfile /tmp/k.pfm $SCN > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  echo "== $name"
  [ -z "$f" ] && { tail -3 /tmp/pmc_$name.log | cut -c1-200; return; }
  python3 - "$f" <<'PY' | tee -a $GRAFT_REPO_ROOT/gpurun_out/pmc/summary.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-46:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in agg:
    if "closest" in k or "shadow" in k or "route" in k or "material" in k or "gen_" in k:
        print(k, len(cnt[k]), {c: "%.4g" % (v / len(cnt[k])) for c, v in agg[k].items()})
PY
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
pass sq3 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_ACTIVE_INST_SCA
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
