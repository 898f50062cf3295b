#!/usr/bin/env python3
"""tools/predict_scaling.py — the 1 -> 8 GPU curve of bench.py, PREDICTED from one GPU (no multi-GPU node has been
available to any round).

The multi-GPU path partitions the image into interleaved 16-line strips (pbrt-v4_amd/multigpu.py: rank r renders strips r, r + N, ...),
replicates the scene, and exchanges nothing until one gather of each rank's own scanlines to rank 0.  A rank's work therefore does not
depend on the other ranks being there: this script renders, on ONE device and one after the other, what each rank of an N-GPU job would
render — through the same calls bench.py makes (Scene.create_renderer(strips=(r, N, 16)), multigpu.render_partition, K steps after a
warm-up of the same K steps) — for N = 1, 2, 4, 8, and records every rank's render time.  The job's time at N is the SLOWEST rank's
(the barrier + max-over-ranks of the bench contract) plus the fixed costs measured here at N = 1 with RCCL (the film copy into the
communicator's buffer and a one-rank gather) plus the wire time of the gather priced on one xGMI link (rank 0 receives (N - 1) / N of
the film: MI355X_MICROARCH.md, ~153 GB/s per link, the peers' links in parallel, so one piece's time; 2x for protocol overhead).

    python tools/predict_scaling.py [--steps 20] [--workload sanmiguel-like] > profiles/r06_scale_prediction.json

PREDICTED, UNMEASURED ON HARDWARE: what it establishes is the load balance of the strip partition (max / mean of the ranks' times) and
the fixed costs; what it cannot see is contention a real node adds (host threads, PCIe during load, xGMI arbitration)."""
import argparse
import importlib.util
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XGMI_LINK_GBS = 153.0


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--workload", default="sanmiguel-like")
    ap.add_argument("--meshes", type=int, default=2000)
    ap.add_argument("--ranks", default="1,2,4,8")
    a = ap.parse_args()
    import torch
    assert torch.cuda.is_available(), "needs a MI355X"
    sys.path.insert(0, ROOT)
    import bench
    wfpt = bench.load_pkg()
    multigpu = load("multigpu", os.path.join(ROOT, "pbrt-v4_amd", "multigpu.py"))
    K = a.steps
    spp_total = 1
    while spp_total < K:
        spp_total *= 2
    td = os.path.join(tempfile.gettempdir(), "wfscale_%s_%d" % (a.workload, a.meshes))
    scene_path = os.path.join(td, a.workload + ".pbrt")
    if not os.path.exists(os.path.join(td, "_complete")):
        os.makedirs(td, exist_ok=True)
        bench.make_scene(scene_path, spp_total, a.workload, a.meshes)
        open(os.path.join(td, "_complete"), "w").write("ok\n")
    os.environ.setdefault("WF_TABLE_CACHE", td)
    out = {"what": "bench.py --gpus N --steps %d on %s, predicted from one GPU: each rank's strip set rendered alone on the same device" % (K, a.workload),
           "status": "PREDICTED, UNMEASURED ON HARDWARE", "steps": K, "strip_height": multigpu.STRIP_HEIGHT, "per_n": {}}
    film_bytes = None
    for N in [int(v) for v in a.ranks.split(",")]:
        ranks = []
        for r in range(N):
            s = wfpt.Scene(path=scene_path, spp=spp_total)
            s.create_renderer(0, strips=(r, N, multigpu.STRIP_HEIGHT) if N > 1 else None)
            torch.cuda.synchronize()
            multigpu.render_partition(s, r, N, 0, K, "strips")   # warm-up: the same K steps (code objects, scratch, queue pages)
            s.clear_film()
            rays0 = s.total_rays()
            t0 = time.perf_counter()
            multigpu.render_partition(s, r, N, 0, K, "strips")
            ms = 1e3 * (time.perf_counter() - t0)
            film_bytes = s.info.width * s.info.height * 32
            ranks.append({"rank": r, "render_ms": round(ms, 3), "rays": int(s.total_rays() - rays0), "rows": int(len(multigpu.strip_rows(r, N, s.info.height)))})
            s.close()
        out["per_n"][str(N)] = {"ranks": ranks}
    # fixed costs at N = 1 with the library path: film copy into a tensor + a one-rank RCCL gather
    fixed_ms = None
    try:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        s = wfpt.Scene(path=scene_path, spp=spp_total)
        s.create_renderer(0)
        film_t = torch.zeros((s.info.height, s.info.width, 4), dtype=torch.float64, device="cuda")
        dist.all_reduce(film_t)
        film_t.zero_()
        s.render(0, 1, 1)
        reps = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.film_to_tensor(film_t)
            multigpu.gather_film(film_t, dist, 0, 1, 0)
            torch.cuda.synchronize()
            reps.append(1e3 * (time.perf_counter() - t0))
        fixed_ms = sorted(reps)[len(reps) // 2]
        s.close()
        dist.destroy_process_group()
    except Exception as ex:   # (the prediction stands without the RCCL leg; say so)
        out["fixed_cost_error"] = str(ex)
    out["fixed_ms_film_copy_and_one_rank_gather"] = fixed_ms
    t1 = max(x["render_ms"] for x in out["per_n"]["1"]["ranks"]) if "1" in out["per_n"] else None
    for n, rec in out["per_n"].items():
        N = int(n)
        times = [x["render_ms"] for x in rec["ranks"]]
        rec["max_ms"], rec["mean_ms"] = max(times), sum(times) / len(times)
        rec["imbalance_max_over_mean"] = rec["max_ms"] / rec["mean_ms"]
        wire_ms = 0.0 if N == 1 else 2.0 * (film_bytes / N) / (XGMI_LINK_GBS * 1e9) * 1e3
        rec["gather_wire_ms_one_xgmi_link_x2"] = wire_ms
        rec["predicted_ms"] = rec["max_ms"] + (fixed_ms or 0.0) + wire_ms
        if t1:
            rec["predicted_speedup_vs_1"] = (t1 + (fixed_ms or 0.0)) / rec["predicted_ms"]
            rec["predicted_msamples_per_s"] = 1920.0 * 1080.0 * K / (rec["predicted_ms"] * 1e-3) / 1e6 if a.workload != "tm-like" else None
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
