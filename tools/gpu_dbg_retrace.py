import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
from conftest import load_pkg
import make_scenes
wfpt = load_pkg(); host, hip = wfpt.libs()
if len(sys.argv) > 1:
    scene_path = sys.argv[1]
else:
    make_scenes.killeroo_like("/tmp/k.pbrt", (1920, 1080), 16)
    scene_path = "/tmp/k.pbrt"
s = wfpt.Scene(path=scene_path, spp=16); s.create_renderer(0)
ctx = s.ctx
for f in ("wf_reset_ray_queue", "wf_gen_camera_rays", "wf_reset_stage_queues", "wf_gen_ray_samples", "wf_intersect_closest", "wf_eval_material", "wf_intersect_shadow", "wf_handle_escaped", "wf_handle_emissive"):
    getattr(hip, f).argtypes = [C.c_void_p] + [C.c_int] * (2 if f in ("wf_gen_camera_rays", "wf_gen_ray_samples", "wf_eval_material") else 1)
hip.wf_queue_size.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
def qs(name):
    n = C.c_int(0); assert hip.wf_queue_size(ctx, name.encode(), C.byref(n)) == 0; return n.value
hip.wf_reset_ray_queue(ctx, 0)
hip.wf_gen_camera_rays(ctx, 0, 0)
for depth in range(0, 4):
    hip.wf_reset_stage_queues(ctx, depth)
    hip.wf_gen_ray_samples(ctx, depth, 0)
    hip.wf_intersect_closest(ctx, depth)
    print("depth", depth, "rays", qs("ray%d" % (depth & 1)), "retrace", qs("retrace"))
    hip.wf_handle_escaped(ctx, depth); hip.wf_handle_emissive(ctx, depth)
    for m in (1, 2, 3, 6):
        hip.wf_eval_material(ctx, m, depth)
    hip.wf_intersect_shadow(ctx, depth)
