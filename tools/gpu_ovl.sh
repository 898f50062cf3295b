#!/bin/bash
# the re-trace launch beside the routing pass / sample generation (second stream) against everything on one stream: bench.py twice
export TMPDIR=/tmp
for v in 1 0 1 0; do
  echo "== WF_OVERLAP_RETRACE=$v"
  WF_OVERLAP_RETRACE=$v timeout 600 python bench.py --steps 64 --warmup 1 --cpu-spp 0 2>/dev/null | python3 -c "import sys,json; b=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'])"
done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
