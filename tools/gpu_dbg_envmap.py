import os, sys, subprocess, json, tempfile
import numpy as np
sys.path.insert(0, "tests")
from conftest import load_pkg, GOLDEN, run_wf_cpu, read_pfm, image_error
wfpt = load_pkg(); wfpt.libs()
base = open(os.path.join(GOLDEN, "envmap.pbrt")).read()
variants = {
    "base": base,
    "maxdepth1": base.replace('"integer maxdepth" [ 5 ]', '"integer maxdepth" [ 1 ]'),
    "maxdepth2": base.replace('"integer maxdepth" [ 5 ]', '"integer maxdepth" [ 2 ]'),
    "maxdepth3": base.replace('"integer maxdepth" [ 5 ]', '"integer maxdepth" [ 3 ]'),
    "floorlow": base.replace("[-8 0 -8 -8 0 8 8 0 8 8 0 -8]", "[-8 -0.01 -8 -8 -0.01 8 8 -0.01 8 8 -0.01 -8]"),
    "uniform": base.replace('LightSource "infinite" "string filename" [ "sky.pfm" ] "float scale" [ 0.8 ] "float illuminance" [ 3.0 ]', 'LightSource "infinite" "rgb L" [0.5 0.6 0.7]'),
    "norotate": base.replace("Rotate -90 1 0 0", "").replace("Rotate 40 0 0 1", ""),
    "noillum": base.replace('"float illuminance" [ 3.0 ]', ''),
}
os.chdir(GOLDEN)
for name, text in variants.items():
    path = os.path.join(GOLDEN, "_dbg_%s.pbrt" % name)
    open(path, "w").write(text)
    s = wfpt.Scene(path=path, spp=4); s.create_renderer(0)
    s.render(0, 4, 1)
    st = s.stats()
    img = s.image()
    tot = s.total_rays()
    s.close()
    j = run_wf_cpu(path, "/tmp/cpu.pfm", 4)
    cpu = read_pfm("/tmp/cpu.pfm")
    d = (img.view(np.uint32) != cpu.view(np.uint32)).any(axis=2)
    print(name, "gpu ind", st["indirect_rays"][:6], "sh", st["shadow_rays"][:6], "cpu ind", j["indirect_rays"][:6], "sh", j["shadow_rays"][:6])
    print(name, "rays gpu", tot, "cpu", j["rays"], "differing pixels", d.sum(), "max rel", image_error(img, cpu).max(), np.argwhere(d)[:4].tolist())
    os.remove(path)
