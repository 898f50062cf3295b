#!/bin/bash
# round 3: MeasuredMaterial parity, the device SAH builder against the host builder, and the spec scene's load with either
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "measured or device_sah or device_morton" > gpurun_out/r3p_pytest.txt 2>&1; grep -v "^  File\|^Extension" gpurun_out/r3p_pytest.txt | tail -25
d=/tmp/wfbench_sm
mkdir -p $d
python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
for mode in device host device; do
  echo "== $mode"
  if [ $mode = host ]; then export WF_HOST_BVH_BUILD=1; else unset WF_HOST_BVH_BUILD; fi
  WF_LOAD_TIMING=1 timeout 200 pbrt-v4_amd/_build/pbrt_amd --stats --spp 4 --outfile /tmp/sm_$mode.pfm $d/sm.pbrt 2>&1 | grep -E "\[load\]|Rendering|Intersect closest"
done 2>&1 | tee gpurun_out/r3p_load.txt
cmp /tmp/sm_device.pfm /tmp/sm_host.pfm && echo "images identical" | tee -a gpurun_out/r3p_load.txt
