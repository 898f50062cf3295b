#!/bin/bash
# split-route traversal (WF_SPLIT_ROUTE): parity, then timings against the block-routed walk
export TMPDIR=/tmp
WF_SPLIT_ROUTE=2 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "image_vs_oracle or mix_material or media_many or per_stage or closest_hit or edge or determin" 2>&1 | tail -3
mkdir -p /tmp/sm /tmp/tab
python tools/make_scenes.py sanmiguel-like /tmp/sm/sm.pbrt --spp ${SPP:-16} > /dev/null
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
export WF_TABLE_CACHE=/tmp/tab
run() {
  echo "== $*"
  env "$@" timeout 200 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/sm.pfm /tmp/sm/sm.pbrt 2>&1 | grep -E "Rendering|Intersect closest  |Intersect shadow  |Route|Total GPU"
  env "$@" timeout 200 pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Rendering|Intersect closest  |Route|Total GPU"
}
run WF_SPLIT_ROUTE=1
run WF_SPLIT_ROUTE=2 WF_CURSOR_CHUNK=1
run WF_SPLIT_ROUTE=2 WF_CURSOR_CHUNK=2
run WF_SPLIT_ROUTE=2 WF_CURSOR_CHUNK=4
run WF_SPLIT_ROUTE=2 WF_CURSOR_CHUNK=8
run WF_SPLIT_ROUTE=2 WF_CURSOR_CHUNK=16
