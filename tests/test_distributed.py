"""N > 1 path on CPU: two gloo ranks render disjoint sample-index subsets with the CPU checker, the
double-precision film accumulators are all-reduced (as bench.py does with RCCL) and must reproduce the
single-process film.  world_size = 2, 127.0.0.1 rendezvous."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import GOLDEN, ROOT, WF_CPU

WORKER = textwrap.dedent("""
    import os, subprocess, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    wf_cpu, scene, outdir = sys.argv[1:4]
    spp = 4
    film_path = os.path.join(outdir, "film_%d.bin" % rank)
    # rank r renders sample indices r, r + world, ... (same partition as bench.py --gpus N)
    subprocess.run([wf_cpu, "--quiet", "--spp", str(spp), "--nthreads", "2", "--samples", str(rank), str(spp), str(world),
                    "--dump-film", film_path, "--outfile", os.path.join(outdir, "img_%d.pfm" % rank), scene], check=True, stdout=subprocess.DEVNULL)
    film = torch.from_numpy(np.fromfile(film_path, dtype=np.float64))
    dist.all_reduce(film)
    if rank == 0:
        film.numpy().tofile(os.path.join(outdir, "film_sum.bin"))
    dist.barrier()
    dist.destroy_process_group()
""")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sample_partition_film_reduce_gloo(built, tmp_path):
    scene = os.path.join(GOLDEN, "cornell64.pbrt")
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(worker), WF_CPU, scene, str(tmp_path)]
    subprocess.run(cmd, check=True, timeout=600, cwd=ROOT)
    single = tmp_path / "film_single.bin"
    subprocess.run([WF_CPU, "--quiet", "--spp", "4", "--dump-film", str(single), "--outfile", str(tmp_path / "s.pfm"), scene], check=True,
                   stdout=subprocess.DEVNULL)
    a = np.fromfile(tmp_path / "film_sum.bin", dtype=np.float64)
    b = np.fromfile(single, dtype=np.float64)
    assert a.shape == b.shape == (64 * 64 * 4,)
    # identical sample sets; only the order of the double-precision additions differs
    assert np.allclose(a, b, rtol=1e-12, atol=0)
    assert (a.reshape(-1, 4)[:, 3] > 0).all()
