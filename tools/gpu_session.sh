#!/bin/bash
# tools/gpu_session.sh — the GPU-box sessions of this repository as ONE parameterised script (it replaces the sixty one-off gpu_*.sh of
# rounds 1-3).  Run through gpurun from the repository root:
#     gpurun --timeout 1500 -- 'bash tools/gpu_session.sh tests bench prof sm16'
# Steps (any subset, in the order given):
#   tests            pytest -m gpu + smoke()                                    -> gpurun_out/<TAG>_pytest_gpu.txt
#   bench            the driver's command, bench.py --steps $STEPS --warmup 5   -> gpurun_out/<TAG>_bench_k$STEPS.json
#   prof             rocprofv3 --kernel-trace --stats of the same command       -> gpurun_out/<TAG>_bench_k$STEPS_rocprofv3_kernel_stats.csv
#   workloads        bench.py --workload killeroo-like / cloud-like / tm-like   -> gpurun_out/<TAG>_bench_<workload>.json
#   sm16             per-kernel times (pbrt_amd --stats) of the spec scene at $SPP spp for _build and every _exp* build (same-box A/B)
#   pmc              rocprofv3 PMC passes (SQ / TCP / TCC / FETCH / WRITE, one pass each) over the spec scene at $PMC_SPP spp
#   soak             $SOAK renders of the spec scene at 4 spp: every image must be the first one (the near-tie queue's guard)
#   ab               pbrt_amd --stats under each environment given in $AB (";"-separated) — knob A/B on one box
#   goldens          every tests/golden scene with a reference render through the native binary, pixel payloads compared bit for bit (no torch import: about a minute)
# Environment: TAG (default r04), STEPS (20), SPP (16), PMC_SPP (4), SCENE (sanmiguel | sanmiguel_sphere | sanmiguel_marble | killeroo | cloud | tm), GREP (kernel-name filter of sm16).
export TMPDIR=/tmp
TAG=${TAG:-r05}; STEPS=${STEPS:-20}; SPP=${SPP:-16}; PMC_SPP=${PMC_SPP:-4}; SCENE=${SCENE:-sanmiguel}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
scene_file() {   # the benchmarked stand-in as a scene file under /tmp (generated once per box)
  case $SCENE in
    sanmiguel) d=/tmp/wfbench_sm; f=$d/sm.pbrt; gen="sanmiguel-like";;
    sanmiguel_sphere) d=/tmp/wfbench_sms; f=$d/sm.pbrt; gen="sanmiguel-like";;   # + one sphere: the GEN = 2 kernels on the headline geometry
    sanmiguel_marble) d=/tmp/wfbench_smm; f=$d/sm.pbrt; gen="sanmiguel-like";;   # + one marble-textured quad: a texture graph on the commonest material type
    killeroo)  d=/tmp/wfbench_k;  f=$d/k.pbrt;  gen="killeroo-like";;
    cloud)     d=/tmp/wfbench_c;  f=$d/c.pbrt;  gen="cloud-like";;
    tm)        d=/tmp/wfbench_tm; f=$d/tm.pbrt; gen="tm-like";;
  esac
  mkdir -p $d
  if [ ! -f $f ]; then
    python $ROOT/tools/make_scenes.py $gen $f --spp 16 > /dev/null
    [ $SCENE = sanmiguel_marble ] && printf 'AttributeBegin\n  Texture "marb" "spectrum" "marble" "float scale" 3\n  Material "diffuse" "texture reflectance" "marb"\n  Translate 0 -6 0.02\n  Shape "trianglemesh" "point3 P" [ -1.5 -1.5 0  1.5 -1.5 0  1.5 1.5 0  -1.5 1.5 0 ] "integer indices" [ 0 1 2 0 2 3 ]\nAttributeEnd\n' >> $f
    [ $SCENE = sanmiguel_sphere ] && printf 'AttributeBegin\n  Material "conductor" "float roughness" 0.1\n  Translate 0 1.5 0\n  Shape "sphere" "float radius" 0.75\nAttributeEnd\n' >> $f
  fi
  echo $f
}
for step in "$@"; do
  echo "=== $step"
  case $step in
    tests)
      timeout 1500 python -m pytest tests -x -q -s -m gpu ${PYTEST_ARGS:-} > $OUT/${TAG}_pytest_gpu_full.txt 2>&1
      grep -E "passed|failed|FAILED|ERROR|error" $OUT/${TAG}_pytest_gpu_full.txt | tail -8 | tee $OUT/${TAG}_pytest_gpu.txt
      timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/${TAG}_smoke.txt;;
    bench)
      # the box itself (the pool's boxes differ by up to 17 % on the same binary: clocks / power cap recorded beside every bench line)
      (rocm-smi --showclocks --showpower --showmaxpower --showperflevel 2>/dev/null | grep -E "clk|Power|Perf" | head -12) > $OUT/${TAG}_box.txt
      timeout 900 python bench.py --steps $STEPS --warmup 5 2> $OUT/${TAG}_bench_err.txt | tee $OUT/${TAG}_bench_k$STEPS.json
      tail -3 $OUT/${TAG}_bench_err.txt;;
    prof)
      rm -rf /tmp/prof
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $ROOT/bench.py --steps $STEPS --warmup 5 --cpu-spp 0 \
          > $OUT/${TAG}_bench_k${STEPS}_under_rocprofv3.json 2> /tmp/rocprof_err.txt)
      for f in $(find /tmp/prof -name "*kernel_stats.csv"); do cp $f $OUT/${TAG}_bench_k${STEPS}_rocprofv3_kernel_stats.csv; done
      head -14 $OUT/${TAG}_bench_k${STEPS}_rocprofv3_kernel_stats.csv | cut -c1-200;;
    workloads)
      for w in ${WORKLOADS:-killeroo-like cloud-like}; do
        timeout 900 python bench.py --workload $w --steps ${WSTEPS:-16} --warmup 2 2>> $OUT/${TAG}_bench_err.txt | tee $OUT/${TAG}_bench_$w.json
      done;;
    sm16)
      f=$(scene_file)
      $ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --spp 4 --outfile /tmp/x.pfm $f > /dev/null 2>&1   # (first run on a fresh box: clocks, page faults)
      for b in $ROOT/pbrt-v4_amd/_build $ROOT/pbrt-v4_amd/_exp*; do
        [ -x $b/pbrt_amd ] || continue
        echo "== $b" | tee -a $OUT/${TAG}_${SCENE}_${SPP}spp_stats.txt
        timeout 200 $b/pbrt_amd --stats --spp $SPP --outfile /tmp/x.pfm $f 2>&1 | grep -E "Rendering|${GREP:-Intersect|Material|Medium|medium|Total GPU|film|Generate|escaped|emitters|Route|ransmittance}" | tee -a $OUT/${TAG}_${SCENE}_${SPP}spp_stats.txt
      done;;
    wall)
      # whole-render wall time (no per-stage profile: the material stage's parallel streams are on) of every build, under each environment of $AB
      f=$(scene_file)
      $ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --spp 4 --outfile /tmp/x.pfm $f > /dev/null 2>&1
      IFS=';' read -ra envs <<< "${AB:-WF_NONE=1}"
      for b in $ROOT/pbrt-v4_amd/_build $ROOT/pbrt-v4_amd/_exp*; do
        [ -x $b/pbrt_amd ] || continue
        for e in "${envs[@]}"; do
          for rep in 1 2; do
            echo "== $b $e: $(env $e timeout 200 $b/pbrt_amd --spp $SPP --outfile /tmp/x.pfm $f 2>&1 | grep -E 'Rendering finished')" | tee -a $OUT/${TAG}_${SCENE}_${SPP}spp_wall.txt
          done
        done
      done;;
    pmc)
      f=$(scene_file)
      pass() {
        name=$1; shift
        rm -rf /tmp/pmc_$name
        (cd /tmp && timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -o k -- $ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --spp $PMC_SPP --outfile /tmp/x.pfm $f > /tmp/pmc_$name.log 2>&1)
        c=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
        echo "== $name" | tee -a $OUT/${TAG}_${SCENE}_pmc_${PMC_SPP}spp.txt
        [ -z "$c" ] && { tail -3 /tmp/pmc_$name.log | cut -c1-200; return; }
        python3 - "$c" <<'PY' | tee -a $OUT/${TAG}_${SCENE}_pmc_${PMC_SPP}spp.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-46:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in sorted(agg):
    if any(s in k for s in ("closest", "shadow", "route", "material", "k_mat_", "gen_", "medium", "tr_", "film")):
        print(k, len(cnt[k]), {c: "%.4g" % v for c, v in agg[k].items()})   # totals over the dispatches
PY
      }
      pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
      pass sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU
      pass fetch FETCH_SIZE
      pass write WRITE_SIZE
      if [ "${PMC_FULL:-0}" = "1" ]; then
        pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum
        pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
      fi
      # the ray / item counts of the same scene at the same spp (the denominators)
      $ROOT/pbrt-v4_amd/_build/pbrt_amd --stats --spp $PMC_SPP --outfile /tmp/x.pfm $f 2>&1 | grep -E "rays|launches" | tee -a $OUT/${TAG}_${SCENE}_pmc_${PMC_SPP}spp.txt;;
    soak)
      f=$(scene_file)
      for i in $(seq 1 ${SOAK:-8}); do
        $ROOT/pbrt-v4_amd/_build/pbrt_amd --quiet --spp 4 --outfile /tmp/soak_$i.pfm $f > /dev/null 2>&1 || echo "render $i failed"
        cmp -s /tmp/soak_1.pfm /tmp/soak_$i.pfm && echo "render $i identical" || echo "render $i DIFFERS"
      done | tee $OUT/${TAG}_${SCENE}_soak.txt;;
    goldens)
      # every golden scene that has a reference render (tests/golden/<name>_ref.pfm, written by pbrt_ref --wavefront) through the NATIVE binary
      # (no Python / torch import: a fresh box answers within a minute), compared byte for byte: the quick whole-corpus identity check
      # of a freshly built tree.  Scenes whose reference render was made with other options than "--spp 4" are left to the pytest suite.
      for p in $ROOT/tests/golden/*.pbrt; do
        n=$(basename $p .pbrt); r=$ROOT/tests/golden/${n}_ref.pfm
        [ -f $r ] || continue
        case $n in cornell64_*) spp="";; cornell400|gbuffer_film*|spectral_film*|*_small|volpath_*|mix_materials) continue;; *) spp="--spp 4";; esac
        (cd $ROOT/tests/golden && timeout 60 ${GOLDEN_BIN:-$ROOT/pbrt-v4_amd/_build/pbrt_amd} --quiet $spp --outfile /tmp/g_$n.pfm $p > /tmp/g_$n.log 2>&1)
      done
      # ... and the committed corpus of the differential fuzzer (tests/golden/fuzz: generated scenes + the reduced findings) at the scenes' own spp
      for p in $ROOT/tests/golden/fuzz/*.pbrt; do
        n=$(basename $p .pbrt); r=$ROOT/tests/golden/fuzz/${n}_ref.pfm
        [ -f $r ] || continue
        case $n in stale_depth_*) continue;; esac   # (the reference's order-dependent image: checked under the checker's sequential emulation only)
        (cd $ROOT/tests/golden/fuzz && timeout 60 ${GOLDEN_BIN:-$ROOT/pbrt-v4_amd/_build/pbrt_amd} --quiet --outfile /tmp/g_fuzz__$n.pfm $p > /tmp/g_fuzz__$n.log 2>&1)
      done
      # the pixel payloads bit for bit (the two writers' header lines differ in how they print the scale)
      python3 - $ROOT/tests/golden <<'PY' | tee $OUT/${TAG}_goldens_native.txt
import glob, os, sys
import numpy as np
def pixels(path):
    with open(path, "rb") as f:
        magic = f.readline().strip(); w, h = map(int, f.readline().split()); f.readline()
        return magic, w, h, np.frombuffer(f.read(), dtype=np.uint32)
n = bad = 0
for g in sorted(glob.glob("/tmp/g_*.pfm")):
    name = os.path.basename(g)[2:-4]
    ref = os.path.join(sys.argv[1], "fuzz", name[6:] + "_ref.pfm") if name.startswith("fuzz__") else os.path.join(sys.argv[1], name + "_ref.pfm")
    a, b = pixels(g), pixels(ref)
    n += 1
    same = a[:3] == b[:3] and a[3].size == b[3].size and bool((a[3] == b[3]).all())
    if not same:
        bad += 1
        print("DIFFERS:", name, "identical fraction", float((a[3] == b[3]).mean()) if a[3].size == b[3].size else "size")
print("goldens through the native binary: %d rendered, %d differ from the reference's render (bit for bit)" % (n, bad))
PY
      ;;
    ab)
      f=$(scene_file)
      IFS=';' read -ra envs <<< "${AB:-WF_NONE=1}"
      for e in "${envs[@]}"; do
        echo "== $e" | tee -a $OUT/${TAG}_${SCENE}_ab.txt
        env $e timeout 200 $ROOT/pbrt-v4_amd/_build/pbrt_amd --stats --spp $SPP --outfile /tmp/x.pfm $f 2>&1 | grep -E "Rendering|${GREP:-Intersect|Material|Total GPU}" | tee -a $OUT/${TAG}_${SCENE}_ab.txt
      done;;
    *) echo "unknown step $step";;
  esac
done
