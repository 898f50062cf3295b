#!/bin/bash
# tools/exp/wwm_bracket/run.sh — on the GPU box: every built case through the harness
cd "$(dirname "$0")/_build/${SUB:-.}"
for d in *_default.co; do c=${d%_default.co}; echo "== $c ($( (grep " $c.o" candidates.txt 2>/dev/null || echo 0) | awk '{print $1}') ordinary instructions inside whole-wave brackets in the default build)"; timeout 120 ./harness ${c}_default.co ${c}_safe.co ./${c}_ref.so; echo "   exit status $?"; done
