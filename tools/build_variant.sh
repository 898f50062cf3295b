#!/bin/bash
# tools/build_variant.sh NAME "EXTRA hipcc flags" ["MATEXTRA hipcc flags"] — a variant build of the HIP back end under
# pbrt-v4_amd/_exp_NAME for same-box A/B timing (`tools/gpu_session.sh sm16` runs every _exp* beside _build; WF_BUILD_DIR selects one
# from Python).  wf_backend.o is recompiled with EXTRA; the material objects are copied from _build unless MATEXTRA is given (then all
# thirty are recompiled with it: about a minute on eight cores).
set -e
cd "$(dirname "$0")/../pbrt-v4_amd"
name=$1; extra=$2; matextra=$3
out=_exp_$name
mkdir -p $out
cp -u _build/*.o $out/ 2>/dev/null || true
rm -f $out/wf_backend.o
[ -n "$matextra" ] && rm -f $out/wf_mat_*.o
make OUT=$out EXTRA="$extra" MATEXTRA="$matextra" -j8 2>&1 | grep -E "error|warning: unused" | head -5 || true
ls -la $out/libwfhip.so $out/pbrt_amd
