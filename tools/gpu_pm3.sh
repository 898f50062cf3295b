#!/bin/bash
# pixel-major with the coalesced film update: A/B on the spec scene at 16 spp (same box), --stats vs --quiet images, GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/wfbench_sm
mkdir -p $d
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
pbrt-v4_amd/_build/pbrt_amd --quiet --spp 4 --outfile /tmp/sm.pfm $d/sm.pbrt > /dev/null 2>&1
for pm in 0 1 0 1; do
  echo "== WF_PIXEL_MAJOR=$pm"
  WF_PIXEL_MAJOR=$pm timeout 150 pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/sm_s$pm.pfm $d/sm.pbrt 2>&1 | grep -E "Rendering|Intersect|Total GPU|aterial|Generate|Handle|Update film"
done 2>&1 | tee gpurun_out/pm3_ab_sm16.txt
for pm in 0 1; do WF_PIXEL_MAJOR=$pm timeout 150 pbrt-v4_amd/_build/pbrt_amd --quiet --spp 16 --outfile /tmp/sm_q$pm.pfm $d/sm.pbrt > /dev/null 2>&1; done
python - <<'PY' | tee -a gpurun_out/pm3_ab_sm16.txt
import numpy as np
def rd(p):
    f=open(p,'rb'); f.readline(); w,h=map(int,f.readline().split()); f.readline(); return np.frombuffer(f.read(),'<f4').reshape(h,w,3)
im={k:rd('/tmp/sm_%s.pfm'%k) for k in ('s0','s1','q0','q1')}
for a,b in (('q0','q1'),('s0','q0'),('s1','q1'),('s0','s1')):
    d=(im[a].view(np.uint32)!=im[b].view(np.uint32))
    print(a,b,'differing values',int(d.sum()),'max rel',float((np.abs(im[a]-im[b])/np.maximum(np.abs(im[a]),1e-3)).max()))
PY
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pm3_pytest_gpu.txt 2>&1; tail -4 gpurun_out/pm3_pytest_gpu.txt
