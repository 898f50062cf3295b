#!/bin/bash
# round 3: the spec scene's load with the device SAH builder against the host builder (top level: 4 M primitives)
mkdir -p gpurun_out
export TMPDIR=/tmp
d=/tmp/wfbench_sm
mkdir -p $d
python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
for mode in device host device host; do
  echo "== $mode"
  unset WF_HOST_BVH_BUILD WF_DEVICE_BVH_MIN
  if [ $mode = host ]; then export WF_HOST_BVH_BUILD=1; fi
  if [ $mode = devdefs ]; then export WF_DEVICE_BVH_MIN=20000; fi
  WF_LOAD_TIMING=1 timeout 200 pbrt-v4_amd/_build/pbrt_amd --stats --spp 4 --outfile /tmp/sm_$mode.pfm $d/sm.pbrt 2>&1 | grep -E "\[load\]|Rendering"
done 2>&1 | tee gpurun_out/r3q_load.txt
cmp /tmp/sm_device.pfm /tmp/sm_host.pfm && echo "images identical" | tee -a gpurun_out/r3q_load.txt
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "device_sah or instances or big" > gpurun_out/r3q_pytest.txt 2>&1; grep -v "^  File\|^Extension" gpurun_out/r3q_pytest.txt | tail -8
