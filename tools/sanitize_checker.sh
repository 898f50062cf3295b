#!/bin/bash
# tools/sanitize_checker.sh — the CPU checker (oracle/wf_cpu: the per-item bodies the HIP kernels run, compiled for the host) built with
# AddressSanitizer + UndefinedBehaviorSanitizer and run over every golden and fuzz scene.  Round 5: three wrong-code
# incidents of the material kernels were put down to a toolchain defect; an out-of-bounds private-array index or an uninitialised read
# in the shared source would look the same.  This rules the shared source out on these inputs (device-only code — the LDS regroup,
# the queue allocation — is not covered).  About ten minutes on 8 cores.  Test infrastructure only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/asan_build}
mkdir -p $OUT
make -C $ROOT/oracle OUT=$OUT CXXFLAGS="-std=c++17 -O1 -g -mfma -ffp-contract=off -pthread -I ../include -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer" $OUT/wf_cpu > /dev/null
cd $ROOT/tests/golden
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
n=0; bad=0
for f in *.pbrt fuzz/*.pbrt; do
  d=$(dirname $f)
  out=$( (cd $d && timeout 600 $OUT/wf_cpu --quiet --nthreads 2 --outfile /tmp/asan_out.pfm $(basename $f)) 2>&1 ) || true
  if echo "$out" | grep -qE "runtime error|AddressSanitizer"; then bad=$((bad+1)); echo "=== $f"; echo "$out" | grep -E "runtime error|AddressSanitizer|#[0-4] " | head -12; fi
  n=$((n+1))
done
echo "scenes $n, with sanitizer findings $bad"
[ $bad -eq 0 ]
