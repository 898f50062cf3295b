cd $GRAFT_REPO_ROOT/tests/golden/fuzz
B=$GRAFT_REPO_ROOT/pbrt-v4_amd/_build/pbrt_amd
for e in "WF_NONE=1" "WF_ANIM_FAST=0" "WF_BRAID=0" "WF_BRAID=0 WF_TIGHT_INSTANCES=0" "WF_FUSE=1"; do
  env $e $B --quiet --outfile /tmp/o.pfm s6300008.pbrt > /dev/null 2>&1
  python3 - "$e" <<'PY'
import sys, numpy as np
def px(p):
    f=open(p,'rb'); f.readline(); w,h=map(int,f.readline().split()); f.readline(); return np.frombuffer(f.read(),dtype=np.uint32)
a,b=px('/tmp/o.pfm'),px('s6300008_ref.pfm')
print(sys.argv[1], 'identical fraction', float((a==b).mean()))
PY
done
