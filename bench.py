#!/usr/bin/env python3
"""bench.py — whole-job throughput of the MI355X wavefront path tracer on BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[1], "killeroo-simple 1080p 64spp on 1xMI355X", on its
synthetic stand-in (tools/make_scenes.py killeroo-like: ~30k triangles, diffuse + smooth dielectric, two
quad area lights, closed room; the pbrt-v4-scenes asset is not available offline).  A *step* is one pass
of the hot path over one batch: one sample index over the whole 1920x1080 image (both 540-scanline
wavefront passes, wavefront/integrator.cpp:336-442) = 2 073 600 pixel samples.  The default K = 64 steps is
the configuration's full 64 spp.

N > 1 (launched by torch.distributed.run, one process per GPU): the scene is replicated, rank r renders
the step indices r, r+N, ... (identical per-pixel sample sets to the 1-GPU render because the sampler is
keyed on (pixel, sampleIndex, dimension)), and the double-precision film accumulators are summed with one
RCCL all-reduce inside the timed region.  Total work is fixed as N grows: "scaling": "strong".

The JSON line also carries
  roofline      achieved algorithmic GB/s of the dominant kernel ("Intersect closest") vs the 8 TB/s HBM peak:
                bytes per ray from SURVEY.md §8(d)'s formula with the kernel's own node/triangle visit
                counters (collected during warm-up steps with the counting kernel variant), times the rays
                the timed launches traced, divided by the launches' HIP-event durations on the render stream
  cpu_baseline  the reference's own CPU wavefront path (oracle/_ref/pbrt_ref --wavefront, built from the
                unmodified reference sources) on the same scene at a bounded spp, all host cores
"""
import argparse
import importlib.util
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
W, H = 1920, 1080
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def load_pkg():
    spec = importlib.util.spec_from_file_location("wfpt", os.path.join(ROOT, "pbrt-v4_amd", "wfpt.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def make_scene(path, spp, workload="killeroo-like", meshes=1600):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_scenes
    if workload == "sanmiguel-like":
        make_scenes.sanmiguel_like(path, (W, H), spp, n_meshes=meshes)
    elif workload == "cloud-like":
        make_scenes.cloud_like(path, (W, H), spp)
    else:
        make_scenes.killeroo_like(path, (W, H), spp)


def closest_bytes(c, stats_hits_emitter=0):
    """B_closest = 184/ray + 32/node + 48/triangle test + 252 per hit with a material
    (+192 per emitter hit, not counted here: < 1% of hits on this scene)."""
    return 184 * c["closest_rays"] + 32 * c["closest_nodes"] + 48 * c["closest_tris"] + 252 * c["closest_hits"]


def cpu_baseline(scene_path, spp):
    ref = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref")
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "cpu.pfm")
        if os.path.exists(ref):
            t0 = time.time()
            p = subprocess.run([ref, "--wavefront", "--seed", "0", "--spp", str(spp), "--nthreads", str(cores), "--outfile", out, scene_path],
                               capture_output=True, text=True, timeout=900)
            wall = time.time() - t0
            if p.returncode == 0:
                m = re.findall(r"\((\d+\.\d+)s\)", p.stdout + p.stderr)
                secs = float(m[-1]) if m else wall
                return {"value": W * H * spp / secs / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "reference",
                        "sample": "same scene, 1920x1080, %d spp, pbrt --wavefront (CPU WavefrontPathIntegrator), %.1f s render" % (spp, secs)}
        port = os.path.join(ROOT, "oracle", "_build", "wf_cpu")
        if os.path.exists(port):
            p = subprocess.run([port, "--spp", str(spp), "--nthreads", str(cores), "--quiet", "--outfile", out, scene_path],
                               capture_output=True, text=True, timeout=900)
            if p.returncode == 0:
                j = json.loads(p.stdout.strip().splitlines()[-1])
                return {"value": W * H * spp / j["seconds"] / 1e6, "unit": "Msamples/s", "cores": j["threads"], "kind": "port",
                        "sample": "same scene, 1920x1080, %d spp, oracle/wf_cpu" % spp}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-spp", type=int, default=1, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--samples-per-pass", type=int, default=0, help="sample indices carried per pass (0 = automatic, ~64 M rays in flight)")
    ap.add_argument("--workload", choices=["killeroo-like", "sanmiguel-like", "cloud-like"], default="killeroo-like",
                    help="killeroo-like = BASELINE configs[1] stand-in (default, the metric's config); sanmiguel-like = configs[2] stand-in")
    ap.add_argument("--meshes", type=int, default=1600, help="sanmiguel-like: number of 6272-triangle meshes (1600 = 10 M triangles)")
    a = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU path is benchmarked as the product)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    wfpt = load_pkg()
    K, Wm = a.steps, a.warmup
    spp_total = 1
    while spp_total < max(K, Wm):
        spp_total *= 2
    td = tempfile.mkdtemp(prefix="wfbench_")
    scene_path = os.path.join(td, "killeroo-like.pbrt")
    make_scene(scene_path, spp_total, a.workload, a.meshes)
    scene = wfpt.Scene(path=scene_path, spp=spp_total)
    scene.create_renderer(local_rank, samples_per_pass=a.samples_per_pass)
    info = scene.info

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    film_t = torch.zeros((info.height, info.width, 4), dtype=torch.float64, device="cuda") if world > 1 else None

    # warm-up steps (sample indices 0..Wm-1, every rank renders all of them) with the counting kernels:
    # gives the per-ray node/triangle visit averages for the roofline
    counters = None
    if Wm > 0:
        scene.enable_counters(True)
        scene.render(0, Wm, 1)
        counters = scene.counters()
        scene.enable_counters(False)
    scene.clear_film()
    if dist is not None:
        dist.all_reduce(film_t)  # warm the communicator
        film_t.zero_()
    rays_before = scene.total_rays()
    stats_before = scene.stats()
    scene.enable_profile(0 if a.no_roofline else 2)

    barrier()
    t0 = time.perf_counter()
    # timed region: K steps = sample indices 0 .. K-1, dealt round-robin to the ranks
    scene.render(rank, K, world)
    if dist is not None:
        scene.film_to_tensor(film_t)
        dist.all_reduce(film_t)
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    rays_t = torch.tensor([scene.total_rays() - rays_before], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_t)
    T = float(elapsed.item())
    total_rays = float(rays_t.item())

    if rank == 0:
        samples = float(info.width) * info.height * K
        out = {
            "metric": "Msamples/sec (whole node), 1920x1080 at fixed spp",
            "value": samples / T / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": 1e3 * T / K,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "mray_per_s": total_rays / T / 1e6,
            "config": {"workload": ("killeroo-simple 1080p 64spp (BASELINE.json configs[1]) on the killeroo-like stand-in: %d triangles, "
                                    "diffuse + dielectric, maxdepth %d, zsobol; step = 1 sample index x 1920x1080"
                                    if a.workload == "killeroo-like" else
                                    "Disney cloud 1080p (BASELINE.json configs[3]) on the cloud-like stand-in (64^3 uniformgrid medium, g = 0.877, "
                                    "%d triangles, maxdepth %d, zsobol; step = 1 sample index x 1920x1080"
                                    if a.workload == "cloud-like" else
                                    "San Miguel 1080p (BASELINE.json configs[2]) on the sanmiguel-like stand-in: %d triangles, diffuse/coated "
                                    "diffuse/dielectric/conductor, sun + sky + 400 emitters, maxdepth %d, zsobol; step = 1 sample index x 1920x1080")
                                   % (info.n_triangles, info.max_depth),
                       "resolution": [info.width, info.height], "spp": K, "samples_per_pass": scene.samples_per_pass, "partition": "sample-index round-robin x%d + RCCL film all-reduce" % world
                       if world > 1 else "single GPU"},
        }
        if not a.no_roofline and counters and counters["closest_rays"] > 0:
            _, hip = wfpt.libs()
            import ctypes as C
            ms = C.c_double(0)
            n = C.c_int(0)
            hip.wf_kernel_time_ms(scene.ctx, b"Intersect closest", C.byref(ms), C.byref(n))
            st = scene.stats()
            rays_closest = (st["camera_rays"] - stats_before["camera_rays"]) + sum(st["indirect_rays"][1:]) - sum(stats_before["indirect_rays"][1:])
            bytes_per_ray = closest_bytes(counters) / counters["closest_rays"]
            if n.value > 0 and ms.value > 0:
                launches = n.value
                avg_ms = ms.value / launches
                bytes_per_launch = bytes_per_ray * rays_closest / launches
                gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9
                # measured HBM bytes per launch: PMC passes cannot run inside this process; the committed rocprofv3
                # FETCH_SIZE/WRITE_SIZE summary of the same kernel on the same workload (profiles/pmc_traffic.json,
                # bytes per ray, corrected as MI355X_MICROARCH.md prescribes) scaled to this run's rays per launch
                traffic = None
                tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
                if a.workload == "killeroo-like" and os.path.exists(tp):
                    traffic = json.load(open(tp))["hbm_bytes_per_ray"] * rays_closest / launches
                out["roofline"] = {
                    "bound": "hbm", "kernel": "Intersect closest (k_closest_fast)", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": bytes_per_launch,
                    "launches": launches, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_ray": bytes_per_ray,
                    "nodes_per_ray": counters["closest_nodes"] / counters["closest_rays"],
                    "tris_per_ray": counters["closest_tris"] / counters["closest_rays"],
                    "rays_per_launch": rays_closest / launches,
                }
        if world == 1 and a.cpu_spp > 0:
            out["cpu_baseline"] = cpu_baseline(scene_path, a.cpu_spp)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
