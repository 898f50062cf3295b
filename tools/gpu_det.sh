#!/bin/bash
# where does the rare run-to-run difference come from?  the spec scene at 16 spp, 4 renders per configuration: distinct images counted
export TMPDIR=/tmp
mkdir -p gpurun_out
d=/tmp/wfbench_sm
mkdir -p $d
[ -f $d/sm.pbrt ] || python tools/make_scenes.py sanmiguel-like $d/sm.pbrt --spp 16 > /dev/null
run() {  # name, env...
  name=$1; shift
  for k in 1 2 3 4; do
    env "$@" timeout 150 pbrt-v4_amd/_build/pbrt_amd --stats --spp 16 --outfile /tmp/det_${name}_$k.pfm $d/sm.pbrt 2>&1 | grep -E "Camera rays|Indirect rays, depth 1 |Shadow rays, depth 0 " | tr -s ' ' | tr '\n' ';'
    echo
  done
}
{
echo "== default"; run default WF_X=0
echo "== WF_SPLIT_ROUTE=0 (workgroup-routed walk: no refill, no service workgroups)"; run noroute WF_SPLIT_ROUTE=0
echo "== WF_HOST_BVH_BUILD=1"; run hostbvh WF_HOST_BVH_BUILD=1
echo "== WF_PIXEL_MAJOR=0"; run smajor WF_PIXEL_MAJOR=0
python - <<'PY'
import numpy as np, hashlib, glob
def rd(p):
    f=open(p,'rb'); f.readline(); w,h=map(int,f.readline().split()); f.readline(); return np.frombuffer(f.read(),'<f4').reshape(h,w,3)
ref=None
for name in ('default','noroute','hostbvh','smajor'):
    ims=[rd('/tmp/det_%s_%d.pfm'%(name,k)) for k in (1,2,3,4)]
    hs=[hashlib.sha1(i.tobytes()).hexdigest()[:8] for i in ims]
    if ref is None: ref=ims[0]
    diffs=[int((i.view(np.uint32)!=ref.view(np.uint32)).sum()) for i in ims]
    where=[]
    for i in ims:
        ys,xs=np.nonzero((i.view(np.uint32)!=ref.view(np.uint32)).any(axis=2))
        where.append(list(zip(ys[:4].tolist(),xs[:4].tolist())))
    print(name, hs, 'values differing from default run 1:', diffs, where)
PY
} 2>&1 | tee gpurun_out/det_sm16.txt
