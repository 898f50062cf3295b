#!/bin/bash
# ray-coherence pass: parity under sorting, then timings on the sanmiguel-like scene (spec) for several key layouts
export TMPDIR=/tmp
WF_RAY_SORT=7 WF_SORT_MIN=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "image_vs_oracle or mix_material or media_many or per_stage" 2>&1 | tail -3
mkdir -p /tmp/sm /tmp/tab
python tools/make_scenes.py sanmiguel-like /tmp/sm/sm.pbrt --spp ${SPP:-8} > /dev/null
export WF_TABLE_CACHE=/tmp/tab
run() {
  echo "== $*"
  env "$@" timeout 200 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/sm.pfm /tmp/sm/sm.pbrt 2>&1 | grep -E "Rendering|Intersect|Sort|Total GPU|Material"
}
run WF_RAY_SORT=0
run WF_RAY_SORT=1
run WF_RAY_SORT=3
run WF_RAY_SORT=7
run WF_RAY_SORT=3 WF_SORT_OBITS=5 WF_SORT_DBITS=3
run WF_RAY_SORT=3 WF_SORT_OBITS=7 WF_SORT_DBITS=5
run WF_RAY_SORT=3 WF_SORT_OBITS=8 WF_SORT_DBITS=0
run WF_RAY_SORT=3 WF_SORT_OBITS=4 WF_SORT_DBITS=6
run WF_RAY_SORT=0
