#!/bin/bash
# traversal tuning sweep on the killeroo-like scene (env knobs read by wf_scene_upload)
mkdir -p gpurun_out
python tools/make_scenes.py killeroo-like /tmp/k.pbrt --spp 16
run() { echo "== $*"; env "$@" pbrt-v4_amd/_build/pbrt_amd --stats --outfile /tmp/k.pfm /tmp/k.pbrt 2>&1 | grep -E "Intersect|Rendering finished"; }
run WF_NO_FAST=1
run WF_REFILL_PRIMARY=1 WF_REFILL_BOUNCE=40 WF_REFILL_SHADOW=40
run WF_REFILL_PRIMARY=1 WF_REFILL_BOUNCE=1 WF_REFILL_SHADOW=1
run WF_REFILL_PRIMARY=1 WF_REFILL_BOUNCE=20 WF_REFILL_SHADOW=20
run WF_REFILL_PRIMARY=1 WF_REFILL_BOUNCE=56 WF_REFILL_SHADOW=56
run WF_REFILL_PRIMARY=32 WF_REFILL_BOUNCE=32 WF_REFILL_SHADOW=32
run WF_PGRID_MULT=1.6 WF_REFILL_PRIMARY=1 WF_REFILL_BOUNCE=40 WF_REFILL_SHADOW=40
run WF_PGRID_MULT=0.6 WF_REFILL_PRIMARY=1 WF_REFILL_BOUNCE=40 WF_REFILL_SHADOW=40
