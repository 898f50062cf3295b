#!/bin/bash
# round 3: the transmittance wavefront — media parity tests, A/B on the cloud-like spec scene (WF_TR_WAVEFRONT=0: the per-lane loop),
# then the 64-step bench under rocprofv3 with the full-size production warm-up
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "media or cloud or grid or benchmark_standins or subsurface or edge" 2>&1 | tail -5 | tee gpurun_out/r3j_pytest_media.txt
for v in 0 1; do
  echo "== WF_TR_WAVEFRONT=$v"
  WF_TR_WAVEFRONT=$v timeout 900 python bench.py --workload cloud-like --steps 16 --warmup 2 --breakdown --cpu-spp 0 2>/dev/null | tee gpurun_out/r3j_bench_cloud_tr$v.json | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step']); print({k:v['total_ms'] for k,v in j['stage_ms'].items() if 'shadow' in k or 'medium' in k.lower() or 'Tr' in k})"
done
rm -rf /tmp/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 2 --cpu-spp 0 > $GRAFT_REPO_ROOT/gpurun_out/r3j_bench_k64_under_rocprof.json 2> /tmp/rocprof_err.txt)
for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f gpurun_out/r3j_bench_k64_rocprofv3_kernel_stats.csv; done
head -8 gpurun_out/r3j_bench_k64_rocprofv3_kernel_stats.csv | cut -c1-160
cut -c1-300 gpurun_out/r3j_bench_k64_under_rocprof.json
