#!/bin/bash
# tools/exp/wwm_bracket/build.sh [first_seed last_seed] — generates random kernels, keeps the ones in which the compiler (default flags) puts ordinary
# vector instructions inside a whole-wave bracket (tools/carrier_audit.py), and builds for each: the default code object, the
# -amdgpu-spill-sgpr-to-vgpr=0 one, the host reference; and the harness.  Everything under tools/exp/wwm_bracket/_build (git-ignored).
cd "$(dirname "$0")"
B=_build; mkdir -p $B/scan
A=${1:-1}; Z=${2:-60}
for s in $(seq $A $Z); do
  NV=$((140 + (s % 5) * 25)); W=$((3 + s % 2))
  python3 gen.py $s 112 80 $NV $W $B/scan/s$s
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -c $B/scan/s$s.hip -o $B/scan/s$s.o 2>/dev/null ) &
  (( s % 8 == 0 )) && wait
done; wait
python3 ../../carrier_audit.py $B/scan -q | grep "ALL lanes" | awk '{print $1}' | sort | uniq -c | sort -rn > $B/candidates.txt
cat $B/candidates.txt
for o in $(awk '{print $2}' $B/candidates.txt); do
  c=${o%.o}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --genco $B/scan/$c.hip -o $B/${c}_default.co &
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-spill-sgpr-to-vgpr=0 --genco $B/scan/$c.hip -o $B/${c}_safe.co &
  g++ -O2 -std=c++17 -mfma -ffp-contract=off -shared -fPIC $B/scan/${c}_ref.cpp -o $B/${c}_ref.so &
  wait
done
hipcc -O2 -std=c++17 harness.cpp -o $B/harness -ldl
ls $B | head -40
