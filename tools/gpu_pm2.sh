#!/bin/bash
# determinism check: the spec scene at 16 spp twice per item order; differing values counted
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/wfbench_sm
mkdir -p $d
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
for run in a b; do for pm in 0 1; do
  WF_PIXEL_MAJOR=$pm timeout 150 pbrt-v4_amd/_build/pbrt_amd --quiet --spp 16 --outfile /tmp/sm_${pm}${run}.pfm $d/sm.pbrt > /dev/null 2>&1
done; done
python - <<'PY' | tee gpurun_out/pm2_determinism.txt
import numpy as np
def rd(p):
    f=open(p,'rb'); f.readline(); w,h=map(int,f.readline().split()); f.readline(); return np.frombuffer(f.read(),'<f4').reshape(h,w,3)
im={k:rd('/tmp/sm_%s.pfm'%k) for k in ('0a','0b','1a','1b')}
for a,b in (('0a','0b'),('1a','1b'),('0a','1a')):
    d=(im[a].view(np.uint32)!=im[b].view(np.uint32))
    ys,xs=np.nonzero(d.any(axis=2))
    print(a,b,'differing values',int(d.sum()),'pixels',len(ys),'max rel',float((np.abs(im[a]-im[b])/np.maximum(np.abs(im[a]),1e-3)).max()), list(zip(ys[:5].tolist(),xs[:5].tolist())))
PY
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pm2_pytest_gpu.txt 2>&1; tail -4 gpurun_out/pm2_pytest_gpu.txt
