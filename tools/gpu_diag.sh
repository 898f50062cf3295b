#!/bin/bash
# GPU parity suite with the complete log kept (gpurun_out/pytest_gpu_full.txt), then the killeroo-like perf check
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -x -v -m gpu > gpurun_out/pytest_gpu_full.txt 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed|Fatal|fault|Memory" gpurun_out/pytest_gpu_full.txt | tail -40
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Intersect|Diffuse|Total GPU"
# general-primitive traversal variants at full size: alpha cutouts + normal maps, spheres
for sc in alpha-normalmap spheres; do
  python tools/make_scenes.py $sc tests/golden/_big_$sc.pbrt --spp 16
  echo "== $sc 1080p 16 spp"
  timeout 120 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/big.pfm tests/golden/_big_$sc.pbrt 2>&1 | grep -E "Rendering|Intersect|Total GPU"
  rm -f tests/golden/_big_$sc.pbrt
done
