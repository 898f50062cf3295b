#!/bin/bash
# tools/build_variant.sh NAME "EXTRA hipcc flags" — a variant build of the HIP back end under pbrt-v4_amd/_exp_NAME for same-box A/B
# timing (tools/gpu_sm16.sh runs every _exp* beside _build; WF_BUILD_DIR selects one from Python).  Only wf_backend.o is recompiled;
# the material objects are copied from _build.
set -e
cd "$(dirname "$0")/../pbrt-v4_amd"
name=$1; shift
out=_exp_$name
mkdir -p $out
cp -u _build/*.o $out/
rm -f $out/wf_backend.o
make OUT=$out EXTRA="$*" -j4 2>&1 | grep -E "error|warning: unused" | head -5 || true
ls -la $out/libwfhip.so $out/pbrt_amd
