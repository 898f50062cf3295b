#!/usr/bin/env python3
"""bench.py — whole-job throughput of the MI355X wavefront path tracer on BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2], "San Miguel 1080p 256spp on 1xMI355X" — the configuration the
north_star's target is quoted on — on its synthetic stand-in at the SURVEY.md 8(d) specification (tools/make_scenes.py
sanmiguel-like: 2000 meshes x 5000 triangles = 10 M unique triangles, three decades of sizes, 60 % of the meshes in 200
object-instance definitions instanced 1..50x (two-level BVH), 10 % alpha-cut, 1k^2 image textures on the diffuse 40 %,
coated diffuse / dielectric / conductor, a sun + a 2k^2 equal-area image sky + 500 emitters; the pbrt-v4-scenes asset is
not available offline).  A *step* is one pass of the hot path over one batch: one sample index over the whole 1920x1080
image (both 540-scanline wavefront passes, wavefront/integrator.cpp:336-442) = 2 073 600 pixel samples; 256 steps are the
configuration's 256 spp (throughput does not depend on K: every step repeats the same passes on new sample indices).
--workload killeroo-like selects configs[1] (round 1's default), cloud-like configs[3].

N > 1 (launched by torch.distributed.run, one process per GPU): the scene is replicated (built once: the first rank to
arrive writes the table cache, the others load it), the IMAGE is partitioned — rank r renders the interleaved 16-line
strips r, r+N, ... for every step (pbrt-v4_amd/multigpu.py, wf_set_strips) — and the double-precision film accumulators
go to rank 0 with one RCCL reduce inside the timed region (disjoint strips: a gather, bit-identical to the 1-GPU film).
Total work is fixed as N grows: "scaling": "strong".  --partition samples selects the sample-index partition instead.

The JSON line also carries
  roofline      achieved algorithmic GB/s of the dominant kernel ("Intersect closest") vs the 8 TB/s HBM peak:
                bytes per ray from SURVEY.md §8(d)'s formula with the kernel's own node/triangle visit
                counters (collected during warm-up steps with the counting kernel variant), times the rays
                the timed launches traced, divided by the launches' HIP-event durations on the render stream
                (frac = the walk kernel alone; stage_frac = the whole closest-hit stage the bytes are charged to: walk +
                "Route hits" (EnqueueWorkAfterIntersection) + the near-tie re-trace launch where a scene still has one)
  roofline_shadow  the same for the any-hit stage ("Intersect shadow"): 124 B/ray + 32 B/node + 48 B/triangle test + 32 B per
                unoccluded ray
  cpu_baseline  the reference's own CPU wavefront path (oracle/_ref/pbrt_ref --wavefront, built from the
                unmodified reference sources) on the same scene at a bounded spp, all host cores
  parity        the image the GPU renders of the SAME scene at the cpu_baseline's spp (same sampler state) against the image
                pbrt_ref --wavefront just rendered: max relative error (relative to max(|ref|, 1e-2)) and the fraction of
                bit-identical values; above 1e-3 the run fails
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


def make_scene(path, spp, workload="sanmiguel-like", meshes=2000):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_scenes
    if workload == "sanmiguel-like":
        return make_scenes.sanmiguel_like(path, (W, H), spp, n_meshes=meshes, n_defs=max(1, meshes // 10))
    elif workload == "tm-like":
        return make_scenes.tm_like(path, (W, H), spp)
    elif workload == "cloud-like":
        make_scenes.cloud_like(path, (W, H), spp, n=512)   # a 512^3 density grid
    else:
        make_scenes.killeroo_like(path, (W, H), spp)


def closest_bytes(c, stats_hits_emitter=0):
    """B_closest = 184/ray + 32/node + 48/triangle test + 252 per hit with a material
    (+192 per emitter hit, not counted here: < 1% of hits on this scene)."""
    return 184 * c["closest_rays"] + 32 * c["closest_nodes"] + 48 * c["closest_tris"] + 252 * c["closest_hits"]


def cpu_baseline(scene_path, spp, read_pfm):
    """(cpu_baseline dict, the image the CPU path rendered) — the image is what the `parity` block compares the GPU with"""
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
                return ({"value": W * H * spp / secs / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "reference",
                         "sample": "same scene, %dx%d, %d spp, pbrt --wavefront (CPU WavefrontPathIntegrator), %.1f s render" % (W, H, spp, secs)},
                        read_pfm(out).copy())
        port = os.path.join(ROOT, "oracle", "_build", "wf_cpu")
        if os.path.exists(port):
            p = subprocess.run([port, "--spp", str(spp), "--nthreads", str(cores), "--quiet", "--outfile", out, scene_path],
                               capture_output=True, text=True, timeout=900)
            if p.returncode == 0:
                j = json.loads(p.stdout.strip().splitlines()[-1])
                return ({"value": W * H * spp / j["seconds"] / 1e6, "unit": "Msamples/s", "cores": j["threads"], "kind": "port",
                         "sample": "same scene, %dx%d, %d spp, oracle/wf_cpu" % (W, H, spp)}, read_pfm(out).copy())
    return None, None


def measure_traffic(scene_path, spp):
    """HBM bytes per ray of the two traversal kernels, MEASURED in this run: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — they do
    not fit one pass; --kernel-trace only beside --pmc) over child processes that render the benchmarked scene file at `spp` with the
    product's CLI.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950; ray counts from the child's --stats output.
    Returns None when rocprofv3 or the CLI is unavailable (the bench line then falls back to the committed profile and says so)."""
    import csv
    import glob
    import shutil
    exe = os.path.join(ROOT, "pbrt-v4_amd", "_build", "pbrt_amd")
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or not os.path.exists(prof):
        return None
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    totals = {}
    kernels = {}   # kind -> {kernel name: weight}: which kernels the stage's dispatches were (the heaviest one is reported)
    rays = None
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        # (third pass, round 5: the SQ counters of "wave occupancy under divergence" — lanes active per VALU instruction, the share of a
        # wave's lifetime spent waiting / issuing)
        # (fourth and fifth pass, round 6: the instruction mix and the L2's request count, for the DESIGN ceilings of roofline.design)
        for counter in ("FETCH_SIZE", "WRITE_SIZE", SQ_PASS, INST_PASS, L2_PASS):
            out = os.path.join(td, counter.split()[0])
            try:
                pr = subprocess.run([prof, "--pmc"] + counter.split() + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "k", "--",
                                     exe, "--stats", "--spp", str(spp), "--outfile", os.path.join(td, "k.pfm"), scene_path],
                                    capture_output=True, text=True, timeout=400, cwd="/tmp", env=env)
            except Exception:
                return None
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if pr.returncode != 0 or not files:
                if counter in (SQ_PASS, INST_PASS, L2_PASS):
                    continue   # (these passes are extras: the traffic figures stand without them)
                return None
            for r in csv.DictReader(open(files[0])):
                k = r["Kernel_Name"].split("(")[0]
                kind = kernel_kind(k)
                cname = r.get("Counter_Name", counter)
                if kind and cname in counter.split():
                    totals[(kind, cname)] = totals.get((kind, cname), 0.0) + float(r["Counter_Value"])
                    half = "mat_shade" if "k_mat_shade" in k else "mat_nee" if "k_mat_nee" in k else None
                    if half:   # (the two halves of the material stage separately: VERDICT r5 weak 9)
                        totals[(half, cname)] = totals.get((half, cname), 0.0) + float(r["Counter_Value"])
                    kernels.setdefault(kind, {})[k.strip()] = kernels.setdefault(kind, {}).get(k.strip(), 0.0) + (float(r["Counter_Value"]) if cname in ("FETCH_SIZE", "SQ_WAVE_CYCLES") else 0.0)
            txt = pr.stdout + pr.stderr
            cam = re.findall(r"Camera rays\s+(\d+)", txt)
            ind = re.findall(r"Indirect rays, depth\s+\d+\s+(\d+)", txt)
            sh = re.findall(r"Shadow rays, depth\s+\d+\s+(\d+)", txt)
            if cam:
                rays = {"closest": int(cam[-1]) + sum(int(v) for v in ind), "shadow": sum(int(v) for v in sh)}
                mi, me = re.findall(r"Material items\s+(\d+)", txt), re.findall(r"Medium-sample items\s+(\d+)", txt)
                rays["material"] = int(mi[-1]) if mi else 0
                rays["medium"] = int(me[-1]) if me else 0
    if not rays or rays["closest"] <= 0:
        return None
    res = {"spp": spp}
    for kind in ("closest", "shadow", "material", "medium"):   # (material / medium: bytes per ITEM of the stage's kernels)
        f, w = totals.get((kind, "FETCH_SIZE")), totals.get((kind, "WRITE_SIZE"))
        if f is None or w is None or rays.get(kind, 0) <= 0:
            continue
        # the counters are in KiB
        res[kind] = {"hbm_bytes_per_ray": (2 * f + w) * 1024.0 / rays[kind], "write_bytes_per_ray": w * 1024.0 / rays[kind], "rays": rays[kind]}
        wc, wa, ia = totals.get((kind, "SQ_WAVE_CYCLES")), totals.get((kind, "SQ_WAIT_ANY")), totals.get((kind, "SQ_ACTIVE_INST_ANY"))
        tc, av = totals.get((kind, "SQ_THREAD_CYCLES_VALU")), totals.get((kind, "SQ_ACTIVE_INST_VALU"))
        if wc and av:
            # lanes active per VALU instruction = thread-cycles / (64 x instruction-cycles); wait / issue = share of the waves' lifetime
            res[kind]["sq"] = {"lanes_active": tc / (64.0 * av), "wait_frac": wa / wc, "issue_frac": ia / wc}
        iv = totals.get((kind, "SQ_INSTS_VALU"))
        if iv:
            res[kind]["insts_per_ray"] = {c[9:].lower(): totals[(kind, c)] / rays[kind] for c in INST_PASS.split() if (kind, c) in totals}
        rq = totals.get((kind, "TCC_REQ_sum"))
        if rq:
            res[kind]["l2_requests_per_ray"] = rq / rays[kind]
            if totals.get((kind, "TCC_HIT_sum")) is not None and totals.get((kind, "TCC_MISS_sum")) is not None:
                hm = totals[(kind, "TCC_HIT_sum")] + totals[(kind, "TCC_MISS_sum")]
                res[kind]["l2_hit_rate"] = totals[(kind, "TCC_HIT_sum")] / hm if hm else None
        if kind in kernels:
            res[kind]["kernels"] = sorted(kernels[kind], key=lambda n: -kernels[kind][n])
    if rays.get("material", 0) > 0:
        for half in ("mat_shade", "mat_nee"):
            f, w = totals.get((half, "FETCH_SIZE")), totals.get((half, "WRITE_SIZE"))
            if f is not None and w is not None:
                res.setdefault("material", {}).setdefault("by_half", {})[half] = {"hbm_bytes_per_item": (2 * f + w) * 1024.0 / rays["material"], "write_bytes_per_item": w * 1024.0 / rays["material"]}
    return res if "closest" in res else None


SQ_PASS = "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"
INST_PASS = "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_FLAT"
L2_PASS = "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
# MI355X_MICROARCH.md: 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD's 16 lanes for at least 4 cycles; peak engine clock 2.4 GHz;
# L2 34.5 TB/s aggregate in 128-byte lines
N_SIMD, CLOCK_HZ, L2_PEAK_GBS, L2_LINE = 1024, 2.4e9, 34500.0, 128


def design_ceilings(live_kind, rays_per_launch, avg_launch_ms, hbm_bytes_per_ray):
    """roofline.design: what the kernel AS BUILT asks of the chip, against ceilings that cannot be exceeded —
    VALU issue slots (instructions x 4 cycles / (SIMDs x clock x time)), the L2's request rate (requests x line size against its aggregate
    bandwidth: an upper bound on the bytes, a request may be narrower than a line) and the measured HBM traffic.  The SURVEY 8(d) number
    in roofline.frac prices the REFERENCE's visit counts and can exceed 1 on a cache-resident scene; these cannot."""
    if not live_kind or avg_launch_ms <= 0:
        return None
    t = avg_launch_ms * 1e-3
    d = {"source": "this run's rocprofv3 PMC child passes (SQ_INSTS_*, TCC_REQ_sum; per ray) x this run's rays per launch / the HIP-event launch time"}
    fr = {}
    ipr = live_kind.get("insts_per_ray")
    if ipr and ipr.get("valu"):
        d["valu_insts_per_ray"] = ipr["valu"] * 64.0   # wave instructions per ray x 64 lanes = lane slots issued per ray
        d["wave_insts_per_ray"] = ipr
        fr["valu_issue"] = ipr["valu"] * rays_per_launch * 4.0 / (N_SIMD * CLOCK_HZ * t)
        la = (live_kind.get("sq") or {}).get("lanes_active")
        if la:
            d["lanes_active"] = la
            d["useful_valu_frac"] = fr["valu_issue"] * la   # issue slots x lanes that did work
    if live_kind.get("l2_requests_per_ray"):
        d["l2_request_bytes_per_ray"] = live_kind["l2_requests_per_ray"] * L2_LINE
        d["l2_hit_rate"] = live_kind.get("l2_hit_rate")
        fr["l2"] = d["l2_request_bytes_per_ray"] * rays_per_launch / t / 1e9 / L2_PEAK_GBS
    if hbm_bytes_per_ray:
        fr["hbm"] = hbm_bytes_per_ray * rays_per_launch / t / 1e9 / HBM_PEAK_GBS
    if not fr:
        return None
    d["frac_of_ceiling"] = fr
    d["binds"] = max(fr, key=lambda k: fr[k])
    d["frac"] = fr[d["binds"]]
    return d


def kernel_kind(k):
    return ("closest" if "k_closest_fast" in k else "shadow" if "k_shadow_fast" in k
            else "material" if ("k_eval_material" in k or "k_mat_shade" in k or "k_mat_nee" in k)
            else "medium" if "k_medium_sample" in k else None)


def code_object_occupancy(names=None):
    """waves per SIMD each kernel of libwfhip.so can hold, from its code object's metadata (tools/kernel_resources.py): 512 unified
    registers per SIMD lane in granules of 8 (VGPRs + AGPRs), at most 8 waves per SIMD, 160 KB of LDS per CU shared by the workgroups
    resident on its 4 SIMDs (MI355X_MICROARCH.md).  names: substrings to keep."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import kernel_resources
        rows = kernel_resources.kernel_rows(os.path.join(ROOT, "pbrt-v4_amd", "_build", "libwfhip.so"))
    except Exception:
        return None
    out = {}
    for r in rows:
        name = r["demangled"]
        if names and not any(n in name for n in names):
            continue
        regs = int(r.get("vgpr_count") or 0) + int(r.get("agpr_count") or 0)
        regs = max(8, (regs + 7) // 8 * 8)
        wg = int(r.get("max_flat_workgroup_size") or 256)
        lds = int(r.get("group_segment_fixed_size") or 0)
        by_regs = min(8, 512 // regs)
        waves_per_wg = max(1, (wg + 63) // 64)
        by_lds = 8 if lds == 0 else min(8.0, (160 * 1024 // lds) * waves_per_wg / 4.0)
        out[name] = {"vgpr": int(r.get("vgpr_count") or 0), "agpr": int(r.get("agpr_count") or 0), "scratch_bytes": int(r.get("private_segment_fixed_size") or 0),
                     "spilled_vgprs": int(r.get("vgpr_spill_count") or 0), "lds_bytes": lds, "workgroup": wg,
                     "waves_per_simd": min(by_regs, by_lds), "limited_by": "registers" if by_regs <= by_lds else "lds"}
    return out


def shadow_bytes(c):
    """B_shadow = 124/ray + 32/node + 48/triangle test + 32 per unoccluded ray (the L read-modify-write)"""
    return 124 * c["shadow_rays"] + 32 * c["shadow_nodes"] + 48 * c["shadow_tris"] + 32 * c["shadow_unoccluded"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-spp", type=int, default=1, help="spp of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--pmc-spp", type=int, default=4, help="spp of the rocprofv3 PMC passes that measure roofline.traffic in this run (0 = use the committed profile)")
    ap.add_argument("--breakdown", action="store_true", help="time EVERY launch of the timed region with HIP events and add the per-stage totals "
                    "(\"stage_ms\") to the JSON line; the near-tie re-trace of general-primitive scenes then runs on the main stream (no overlap)")
    ap.add_argument("--samples-per-pass", type=int, default=0, help="sample indices carried per pass (0 = automatic, ~64 M rays in flight)")
    ap.add_argument("--workload", choices=["killeroo-like", "sanmiguel-like", "cloud-like", "tm-like"], default="sanmiguel-like",
                    help="sanmiguel-like = BASELINE configs[2] stand-in at the SURVEY 8(d) spec (default: the north_star target config); "
                         "killeroo-like = configs[1] stand-in; cloud-like = configs[3] stand-in; tm-like = configs[4] stand-in (3840x2160, maxdepth 50)")
    ap.add_argument("--partition", choices=["strips", "samples"], default="strips", help="N > 1: interleaved scanline strips (default) or sample indices")
    ap.add_argument("--meshes", type=int, default=2000, help="sanmiguel-like: number of 5000-triangle meshes (2000 = 10 M unique triangles)")
    a = ap.parse_args()
    global W, H
    if a.workload == "tm-like":
        W, H = 3840, 2160

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
    spec = importlib.util.spec_from_file_location("multigpu", os.path.join(ROOT, "pbrt-v4_amd", "multigpu.py"))
    multigpu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(multigpu)
    K, Wm = a.steps, a.warmup
    spp_total = 1
    while spp_total < max(K, Wm):
        spp_total *= 2
    # the generated scene files are kept per box under /tmp (keyed by the generator's source and parameters): the driver's
    # N = 1, 2, 4, 8 runs and the N ranks of one run share them; rank 0 generates, the others wait for its marker file
    import hashlib
    gen_src = open(os.path.join(ROOT, "tools", "make_scenes.py"), "rb").read()
    key = hashlib.sha1(gen_src + ("%s %d %d" % (a.workload, a.meshes, spp_total)).encode()).hexdigest()[:16]
    td = os.path.join(tempfile.gettempdir(), "wfbench_%s_%s" % (a.workload, key))
    scene_path = os.path.join(td, a.workload + ".pbrt")
    marker = os.path.join(td, "_complete")
    t_gen0 = time.perf_counter()
    if not os.path.exists(marker):
        if rank == 0:
            os.makedirs(td, exist_ok=True)
            make_scene(scene_path, spp_total, a.workload, a.meshes)
            open(marker, "w").write("ok\n")
        else:
            while not os.path.exists(marker):
                time.sleep(0.5)
    t_gen = time.perf_counter() - t_gen0
    # the built scene tables are shared through the on-disk cache (WF_TABLE_CACHE): rank 0 builds and writes, the others load
    os.environ.setdefault("WF_TABLE_CACHE", td)
    if dist is not None and rank != 0:
        dist.barrier()
    t_load0 = time.perf_counter()
    scene = wfpt.Scene(path=scene_path, spp=spp_total)         # parse + flat tables + host SAH BVH builds (or the cache)
    t_parse = time.perf_counter() - t_load0
    if dist is not None and rank == 0:
        dist.barrier()
    # upload + production BVH layout + queues.  A rank of a strip partition sizes its queues for ITS rows: a pass then carries `world`
    # times the sample indices of 1/world of the pixels — launches as large as a single GPU's, as far as K allows
    scene.create_renderer(local_rank, samples_per_pass=a.samples_per_pass,
                          strips=(rank, world, multigpu.STRIP_HEIGHT) if (world > 1 and a.partition == "strips") else None)
    torch.cuda.synchronize()
    t_upload = time.perf_counter() - t_load0 - t_parse
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
        # ... and one untimed run of the SAME K steps through the PRODUCTION kernels (the counting pass above runs the reference-order
        # variants): code objects, scratch, RCCL state and queue pages are touched before the timed region, not inside it — and every
        # production launch of the process has the timed region's size, so that a rocprofv3 --stats average of this command agrees
        # with roofline.avg_launch_ms
        multigpu.render_partition(scene, rank, world, 0, K, a.partition)
    scene.clear_film()
    if dist is not None:
        dist.all_reduce(film_t)  # warm the communicator
        film_t.zero_()
    rays_before = scene.total_rays()
    stats_before = scene.stats()
    scene.enable_profile(1 if a.breakdown else (0 if a.no_roofline else 2))

    barrier()
    t0 = time.perf_counter()
    # timed region: K steps = sample indices 0 .. K-1 of this rank's part of the image (or this rank's sample indices), then the
    # film reduce to rank 0
    multigpu.render_partition(scene, rank, world, 0, K, a.partition)
    t_render = time.perf_counter() - t0          # (render_partition returns after the context's stream is synchronised)
    t_copy = t_reduce = 0.0
    if dist is not None:
        scene.film_to_tensor(film_t)
        t_copy = time.perf_counter() - t0 - t_render
        if a.partition == "strips":
            multigpu.gather_film(film_t, dist, rank, world, 0)   # every rank sends only its own scanlines: 1 / world of the film
        else:
            multigpu.reduce_film(film_t, dist, 0)
        torch.cuda.synchronize()
        t_reduce = time.perf_counter() - t0 - t_render - t_copy
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    rays_t = torch.tensor([scene.total_rays() - rays_before], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_t)
    T = float(elapsed.item())
    total_rays = float(rays_t.item())
    # per-rank phases of the timed region (ms): render of the rank's strips, film copy into the RCCL buffer, reduce to rank 0
    phases = torch.tensor([1e3 * t_render, 1e3 * t_copy, 1e3 * t_reduce], dtype=torch.float64, device="cuda")
    all_phases = [torch.zeros_like(phases) for _ in range(world)]
    if dist is not None:
        dist.all_gather(all_phases, phases)
    else:
        all_phases = [phases]

    if rank == 0:
        samples = float(info.width) * info.height * K
        out = {
            "metric": "Msamples/sec (whole node), %dx%d at fixed spp" % (W, H),
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
            "per_rank_ms": [{"render": round(float(p[0]), 3), "film_copy": round(float(p[1]), 3), "reduce": round(float(p[2]), 3)} for p in all_phases],
            "load_s": {"generate_scene_files": round(t_gen, 2), "parse_and_host_bvh_build": round(t_parse, 2), "upload_and_device_layout": round(t_upload, 2)},
            "config": {"workload": ("Transparent Machines 4K 1024spp (BASELINE.json configs[4]) on the tm-like stand-in (SURVEY 8(d) row 5: nested dielectric shells "
                                    "eta 1.3-1.7, a quarter of them rough, coated-conductor frames, 2000 small area lights, image infinite light): %d triangles, "
                                    "maxdepth %d, zsobol; step = 1 sample index x 3840x2160"
                                    if a.workload == "tm-like" else
                                    "killeroo-simple 1080p 64spp (BASELINE.json configs[1]) on the killeroo-like stand-in: %d triangles, "
                                    "diffuse + dielectric, maxdepth %d, zsobol; step = 1 sample index x 1920x1080"
                                    if a.workload == "killeroo-like" else
                                    "Disney cloud 1080p 128spp (BASELINE.json configs[3]) on the cloud-like stand-in at the SURVEY 8(d) spec (512^3 fBm density grid, uniformgrid medium, g = 0.877, sigma_s = 12, inside an interface box, distant light + uniform sky): "
                                    "%d triangles, maxdepth %d, zsobol; step = 1 sample index x 1920x1080"
                                    if a.workload == "cloud-like" else
                                    "San Miguel 1080p 256spp (BASELINE.json configs[2]) on the sanmiguel-like stand-in at the SURVEY 8(d) spec: %d unique "
                                    "triangles in " + str(a.meshes) + " meshes, 60 %% of them in " + str(max(1, a.meshes // 10)) + " object-instance definitions "
                                    "(two-level BVH), 10 %% alpha-cut, 1k^2 image textures, diffuse / coated diffuse / dielectric / conductor, sun + 2k^2 image "
                                    "sky + 500 emitters, maxdepth %d, zsobol; step = 1 sample index x 1920x1080")
                                   % (info.n_triangles, info.max_depth),
                       "resolution": [info.width, info.height], "spp": K, "samples_per_pass": scene.samples_per_pass, "partition": (("interleaved 16-line scanline strips x%d" if a.partition == "strips" else "sample-index round-robin x%d") % world) + " + RCCL gather of the owned scanlines to rank 0"
                       if world > 1 else "single GPU"},
        }
        if not a.no_roofline and counters and counters["closest_rays"] > 0:
            _, hip = wfpt.libs()
            import ctypes as C

            def kernel_ms(name):
                ms, n = C.c_double(0), C.c_int(0)
                hip.wf_kernel_time_ms(scene.ctx, name.encode(), C.byref(ms), C.byref(n))
                return ms.value, n.value

            st = scene.stats()
            rays_closest = (st["camera_rays"] - stats_before["camera_rays"]) + sum(st["indirect_rays"][1:]) - sum(stats_before["indirect_rays"][1:])
            rays_shadow = sum(st["shadow_rays"]) - sum(stats_before["shadow_rays"])
            bytes_per_ray = closest_bytes(counters) / counters["closest_rays"]
            walk_ms, launches = kernel_ms("Intersect closest")
            route_ms, _ = kernel_ms("Route hits")
            retrace_ms, retrace_launches = kernel_ms("Intersect closest: near-tie re-trace")
            # measured HBM bytes per launch: PMC passes cannot run inside this process (rocprofv3 wraps the process); the committed
            # rocprofv3 FETCH_SIZE / WRITE_SIZE summary of the same kernel on the same workload (profiles/pmc_traffic.json: bytes per
            # ray, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; its `source` names the profile file) scaled to
            # this run's rays per launch
            pj = {}
            tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tp):
                pj = json.load(open(tp)).get(a.workload, {})
                if not isinstance(pj, dict):
                    pj = {}
            # ... unless this run can measure it itself: two rocprofv3 PMC child processes on the same scene file (N = 1 only)
            live = None
            if world == 1 and a.pmc_spp > 0 and a.workload != "tm-like":
                t_pmc0 = time.perf_counter()
                live = measure_traffic(scene_path, a.pmc_spp)
                if live:
                    pj = dict(pj)
                    pj["hbm_bytes_per_ray"] = live["closest"]["hbm_bytes_per_ray"]
                    pj["write_bytes_per_ray"] = live["closest"]["write_bytes_per_ray"]
                    if "shadow" in live:
                        pj["shadow_hbm_bytes_per_ray"] = live["shadow"]["hbm_bytes_per_ray"]
                        pj["shadow_write_bytes_per_ray"] = live["shadow"]["write_bytes_per_ray"]
                    pj["source"] = ("MEASURED IN THIS RUN: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) over child processes "
                                    "rendering this scene file at %d spp with pbrt_amd (%.0f s); 2 x FETCH_SIZE + WRITE_SIZE, bytes per ray x this run's rays per launch"
                                    % (a.pmc_spp, time.perf_counter() - t_pmc0))
            if launches > 0 and walk_ms > 0:
                avg_ms = walk_ms / launches
                bytes_per_launch = bytes_per_ray * rays_closest / launches
                gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9
                stage_ms = (walk_ms + route_ms + retrace_ms) / launches
                per_ray = pj.get("hbm_bytes_per_ray")
                traffic = per_ray * rays_closest / launches if per_ray else None
                out["roofline"] = {
                    "bound": "hbm", "kernel": "Intersect closest (k_closest_fast)", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": (pj.get("source", "profiles/pmc_traffic.json") + ("" if live else " (rocprofv3 PMC of the same kernel and workload, bytes per ray x this run's rays per launch; not measured in this process)")) if traffic else None,
                    "traffic_bytes_per_ray": per_ray, "write_bytes_per_ray": pj.get("write_bytes_per_ray"),
                    # the whole stage the bytes are charged to: walk + routing pass (+ the near-tie re-trace launch of scenes that have one)
                    "stage_frac": bytes_per_launch / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "stage_ms_per_launch": {"walk": avg_ms, "route_hits": route_ms / launches, "retrace": retrace_ms / launches, "retrace_launches": retrace_launches},
                    # the same algorithmic rate against the L2 ceiling (34.5 TB/s aggregate), and the measured HBM rate of the kernel
                    "l2_frac": gbs / 34500.0,
                    "traffic_gbs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
                    "traffic_frac_of_hbm_peak": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "launches": launches, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_ray": bytes_per_ray,
                    "nodes_per_ray": counters["closest_nodes"] / counters["closest_rays"],
                    "tris_per_ray": counters["closest_tris"] / counters["closest_rays"],
                    "rays_per_launch": rays_closest / launches,
                }
                if live:
                    dz = design_ceilings(live.get("closest"), rays_closest / launches, avg_ms, per_ray)
                    if dz:
                        out["roofline"]["design"] = dz
            sh_ms, sh_launches = kernel_ms("Intersect shadow")
            if sh_launches > 0 and sh_ms > 0 and counters["shadow_rays"] > 0:
                sb = shadow_bytes(counters) / counters["shadow_rays"]
                avg = sh_ms / sh_launches
                per_launch = sb * rays_shadow / sh_launches
                per_ray = pj.get("shadow_hbm_bytes_per_ray")
                out["roofline_shadow"] = {
                    "bound": "hbm", "kernel": "Intersect shadow (k_shadow_fast)", "achieved": per_launch / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": per_launch / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": per_ray * rays_shadow / sh_launches if per_ray else None,
                    "launches": sh_launches, "avg_launch_ms": avg, "algorithmic_bytes_per_ray": sb, "rays_per_launch": rays_shadow / sh_launches,
                    "nodes_per_ray": counters["shadow_nodes"] / counters["shadow_rays"], "tris_per_ray": counters["shadow_tris"] / counters["shadow_rays"],
                    "mray_per_s": rays_shadow / (sh_ms * 1e-3) / 1e6,
                    "closest_mray_per_s": (rays_closest / (walk_ms * 1e-3) / 1e6) if walk_ms > 0 else None,
                }
                if live:
                    dz = design_ceilings(live.get("shadow"), rays_shadow / sh_launches, avg, per_ray)
                    if dz:
                        out["roofline_shadow"]["design"] = dz
            # ---- the material stage (K9) against the same roofline: B_mat = 252 + 48 bytes read per item (the reference's
            # MaterialEvalWorkItem + its RaySamples) + 184 per spawned ray + 124 per shadow ray written; items from the queue
            # counters (wf_material_items_download), time = the HIP-event durations of every "...Material + BxDF eval" launch of the timed region
            try:
                items = scene.material_items()
                n_items = sum(v for k, v in items.items() if k != "medium_sample")
                rep = [e for e in scene.profile_report()]
                mat_ms = sum(e["total_ms"] for e in rep if "Material" in e["name"])
                mat_launches = sum(e["launches"] for e in rep if "Material" in e["name"])
                spawned = sum(st["indirect_rays"][1:]) - sum(stats_before["indirect_rays"][1:])
                if n_items > 0 and mat_ms > 0:
                    b = 300.0 * n_items + 184.0 * spawned + 124.0 * rays_shadow
                    out["roofline_material"] = {
                        "bound": "hbm", "kernel": "EvaluateMaterialAndBSDF (k_mat_shade<type, variant> + k_mat_nee<type, rare>, all types)", "achieved": b / (mat_ms * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b / (mat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "items": n_items, "items_by_type": {str(k): v for k, v in items.items() if v and k != "medium_sample"}, "spawned_rays": spawned, "shadow_rays": rays_shadow,
                        "algorithmic_bytes": b, "algorithmic_bytes_per_item": b / n_items, "total_ms": mat_ms, "launches": mat_launches,
                        "mitems_per_s": n_items / (mat_ms * 1e-3) / 1e6,
                    }
                    # what the split stage moves BY DESIGN on top of the SURVEY 8(d) bytes: k_mat_shade stores the item's NeeItem,
                    # k_mat_nee loads it — (10 + NBX(type)) planes of 16 B each way (wf_mat.hip NeeIO; NBX = ceil(sizeof(BxDF) / 16)).  Reported
                    # beside `frac` (which stays the contract's 300 B/item model), not folded into it.
                    nbx = {1: 1, 2: 3, 3: 2, 4: 1, 5: 2, 6: 5, 7: 7, 8: 2, 9: 5, 10: 2}
                    nee_b = sum(2.0 * 16.0 * (10 + nbx.get(int(k), 2)) * v for k, v in items.items() if k != "medium_sample" and v)
                    out["roofline_material"]["as_built"] = {
                        "nee_record_bytes_per_item": nee_b / n_items, "bytes_per_item": (b + nee_b) / n_items,
                        "achieved": (b + nee_b) / (mat_ms * 1e-3) / 1e9, "frac": (b + nee_b) / (mat_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "note": "SURVEY 8(d) bytes + the NeeItem the two material kernels hand over through HBM (store + load)"}
                    if live and "material" in live:   # HBM bytes per item of the k_eval_material kernels, from the same PMC child passes
                        rm = out["roofline_material"]
                        rm["traffic_bytes_per_item"] = live["material"]["hbm_bytes_per_ray"]
                        rm["write_bytes_per_item"] = live["material"]["write_bytes_per_ray"]
                        rm["traffic"] = rm["traffic_bytes_per_item"] * n_items / mat_launches
                        rm["algorithmic_bytes_per_launch"] = b / mat_launches
                        rm["traffic_source"] = "this run's rocprofv3 PMC child passes (2 x FETCH_SIZE + WRITE_SIZE of every k_mat_shade / k_mat_nee dispatch / the items the child's --stats reports) x this run's items per launch"
                        if "by_half" in live["material"]:
                            rm["traffic_by_half"] = live["material"]["by_half"]
                        dz = design_ceilings(live["material"], n_items / mat_launches, mat_ms / mat_launches, rm["traffic_bytes_per_item"])
                        if dz:
                            rm["design"] = dz
                med_ms = sum(e["total_ms"] for e in rep if e["name"].startswith("Sample medium"))
                med_launches = sum(e["launches"] for e in rep if e["name"].startswith("Sample medium"))
                if items.get("medium_sample", 0) > 0 and med_ms > 0:
                    # K5 (a14): 364 bytes read per MediumSampleWorkItem, 120 written per scattering event (a MediumScatterWorkItem; the
                    # scattered items of a surface-free medium scene are the rays its scattering stage spawns)
                    b = 364.0 * items["medium_sample"] + 120.0 * spawned
                    out["roofline_medium"] = {
                        "bound": "hbm", "kernel": "SampleMediumInteraction (k_medium_sample)", "achieved": b / (med_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": b / (med_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "items": items["medium_sample"], "algorithmic_bytes": b,
                        "total_ms": med_ms, "launches": med_launches, "mitems_per_s": items["medium_sample"] / (med_ms * 1e-3) / 1e6,
                    }
                    if live and "medium" in live:
                        rm = out["roofline_medium"]
                        rm["traffic_bytes_per_item"] = live["medium"]["hbm_bytes_per_ray"]
                        rm["write_bytes_per_item"] = live["medium"]["write_bytes_per_ray"]
                        rm["traffic"] = rm["traffic_bytes_per_item"] * items["medium_sample"] / med_launches
                        rm["algorithmic_bytes_per_launch"] = b / med_launches
            except Exception as ex:   # (a measurement block must not take the bench line down)
                out["roofline_material_error"] = str(ex)
        # wave occupancy under divergence (north_star): the dominant kernels' waves per SIMD from the code object, and — when this run's PMC
        # child passes ran — the lanes active per VALU instruction and the share of the waves' lifetime spent waiting / issuing
        try:
            if not a.no_roofline and counters and counters["closest_rays"] > 0:
                occ = code_object_occupancy(["k_closest_fast", "k_shadow_fast", "k_mat_shade", "k_mat_nee", "k_medium_sample"]) or {}
                ran = {}
                for kind in ("closest", "shadow", "material", "medium"):
                    if live and kind in live:
                        ks = [k for k in live[kind].get("kernels", []) if any(k.replace("void ", "") in o or o in k for o in occ)]
                        ran[kind] = {"kernels": {k: next((occ[o] for o in occ if o in k or k.replace("void ", "") in o), None) for k in ks[:4]}}
                        if "sq" in live[kind]:
                            ran[kind].update(live[kind]["sq"])
                if ran:
                    out["occupancy"] = ran
                    for kind, key in (("closest", "roofline"), ("shadow", "roofline_shadow"), ("material", "roofline_material"), ("medium", "roofline_medium")):
                        if kind in ran and key in out:
                            for f in ("lanes_active", "wait_frac", "issue_frac"):
                                if f in ran[kind]:
                                    out[key][f] = ran[kind][f]
                            w = [v["waves_per_simd"] for v in ran[kind]["kernels"].values() if v]
                            if w:
                                out[key]["waves_per_simd"] = w[0]
                elif occ:
                    out["occupancy"] = {"code_object": {k: v for k, v in occ.items() if "<1, true, true>" in k or "<1, true>" in k or "<1, 1>" in k or "<1, false>" in k}}
        except Exception as ex:
            out["occupancy_error"] = str(ex)
        if a.breakdown:
            out["stage_ms"] = {e["name"]: {"launches": e["launches"], "total_ms": round(e["total_ms"], 3)} for e in scene.profile_report()}
        parity_fail = None
        if world == 1 and a.cpu_spp > 0:
            out["cpu_baseline"], ref_img = cpu_baseline(scene_path, a.cpu_spp, wfpt.read_pfm)
            if ref_img is not None:
                # parity on the benchmarked workload: the same scene file at the baseline's spp (the sampler's scrambling depends on
                # the sample count, so the scene is loaded once more with it) rendered by the product, against the image the
                # reference's CPU wavefront path just wrote
                import numpy as np
                scene.close()
                os.environ.pop("WF_TABLE_CACHE", None)
                s1 = wfpt.Scene(path=scene_path, spp=a.cpu_spp)
                s1.create_renderer(local_rank)
                s1.render(0, a.cpu_spp, 1)
                img = s1.image()
                s1.close()
                rel = np.abs(img.astype(np.float64) - ref_img) / np.maximum(np.abs(ref_img), 1e-2)
                out["parity"] = {"against": "oracle/_ref/pbrt_ref --wavefront (the cpu_baseline render), same scene file, %d spp, seed 0" % a.cpu_spp,
                                 "max_rel": float(rel.max()), "bit_identical_fraction": float((img.view(np.uint32) == ref_img.view(np.uint32)).mean()),
                                 "tolerance": 1e-3, "values": int(img.size)}
                if not (rel.max() <= 1e-3) or not np.isfinite(img).all():
                    parity_fail = "parity FAILED on the benchmarked workload: max relative error %g > 1e-3" % rel.max()
        print(json.dumps(out), flush=True)
        if parity_fail:
            raise SystemExit(parity_fail)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
