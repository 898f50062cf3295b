#!/bin/bash
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 4
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|TD_[A-Z_0-9a-z]+)\b" | sort -u | tr '\n' ' ' | head -c 6000; echo
cd /tmp
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  WF_REFILL_BOUNCE=1 WF_REFILL_SHADOW=1 timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o k -- $GRAFT_REPO_ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --outfile /tmp/k.pfm /tmp/k.pbrt > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  echo "== $name: $f"; tail -3 /tmp/pmc_$name.log | cut -c1-300
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
    if "closest" in k or "shadow" in k:
        print(k, cnt[k], {c: "%.4g" % (v / cnt[k]) for c, v in agg[k].items()})
PY
}
pass ta1 TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp2 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
