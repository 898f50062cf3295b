#!/bin/bash
# samples-per-pass sweep on the bench workload + the GPU parity suite
mkdir -p gpurun_out
for s in 1 2 4 8 16; do
  echo "== samples_per_pass $s"
  timeout 200 python bench.py --steps 32 --warmup 2 --cpu-spp 0 --samples-per-pass $s 2>&1 | tail -1 | tee gpurun_out/spp_$s.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['mray_per_s'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_ms'), d.get('roofline',{}).get('rays_per_launch'))"
done
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
