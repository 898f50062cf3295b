"""N > 1 path on CPU (world_size = 2, gloo, 127.0.0.1 rendezvous).  The product's partition / reduce module
(pbrt-v4_amd/multigpu.py) runs as it does under bench.py --gpus N; the per-rank rendering — which needs a GPU in the product —
is stood in for by the CPU checker, which takes the same partition parameters as wf_set_strips (`--strips rank count height`).
Checked: the strips the kernels' BandScanline assigns to a rank are exactly multigpu.strip_rows(), the ranks' strips are
disjoint and cover the image, and the film reduced to rank 0 is BIT-IDENTICAL to the single-process film (strip partition);
the sample-index partition agrees to double rounding."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import GOLDEN, ROOT, WF_CPU

WORKER = textwrap.dedent("""
    import importlib.util, os, subprocess, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    root, wf_cpu, scene, outdir, partition = sys.argv[1:6]
    spec = importlib.util.spec_from_file_location("multigpu", os.path.join(root, "pbrt-v4_amd", "multigpu.py"))
    multigpu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(multigpu)
    spp, H, W = 4, 64, 96

    class CheckerScene:   # the two calls render_partition makes, answered by the CPU checker
        def set_strips(self, r, n, h):
            self.strips = (r, n, h)
        def render(self, begin, end, step):
            film_path = os.path.join(outdir, "film_%s_%d.bin" % (partition, rank))
            subprocess.run([wf_cpu, "--quiet", "--spp", str(spp), "--nthreads", "2", "--samples", str(begin), str(end), str(step),
                            "--strips"] + [str(v) for v in self.strips] +
                           ["--dump-film", film_path, "--outfile", os.path.join(outdir, "img_%d.pfm" % rank), scene], check=True, stdout=subprocess.DEVNULL)
            self.film = torch.from_numpy(np.fromfile(film_path, dtype=np.float64).reshape(H, W, 4))
            return 0.0

    s = CheckerScene()
    multigpu.render_partition(s, rank, world, 0, spp, partition)
    if partition == "strips":
        owned = np.where(s.film[..., 3].numpy().sum(axis=1) > 0)[0]
        assert (owned == multigpu.strip_rows(rank, world, H)).all(), (rank, owned)
    # strips: every rank sends only its own scanlines (gather_film); samples: a true sum (reduce_film)
    film = multigpu.gather_film(s.film, dist, rank, world, 0) if partition == "strips" else multigpu.reduce_film(s.film, dist, 0)
    if rank == 0:
        film.numpy().tofile(os.path.join(outdir, "film_%s_sum.bin" % partition))
    dist.barrier()
    dist.destroy_process_group()
""")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(tmp_path, partition, scene):
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(worker), ROOT, WF_CPU, scene, str(tmp_path), partition]
    subprocess.run(cmd, check=True, timeout=600, cwd=ROOT)
    single = tmp_path / "film_single.bin"
    subprocess.run([WF_CPU, "--quiet", "--spp", "4", "--dump-film", str(single), "--outfile", str(tmp_path / "s.pfm"), scene], check=True,
                   stdout=subprocess.DEVNULL)
    return np.fromfile(tmp_path / ("film_%s_sum.bin" % partition), dtype=np.float64), np.fromfile(single, dtype=np.float64)


def test_strip_rows_partition_the_image():
    import importlib.util
    spec = importlib.util.spec_from_file_location("multigpu", os.path.join(ROOT, "pbrt-v4_amd", "multigpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for H in (64, 1080, 2160, 45):
        for world in (1, 2, 3, 8):
            rows = [m.strip_rows(r, world, H) for r in range(world)]
            allrows = np.sort(np.concatenate(rows))
            assert (allrows == np.arange(H)).all()
            assert max(len(r) for r in rows) - min(len(r) for r in rows) <= m.STRIP_HEIGHT


def test_strip_partition_film_reduce_gloo(built, tmp_path):
    a, b = _run(tmp_path, "strips", os.path.join(GOLDEN, "instances.pbrt"))
    assert a.shape == b.shape == (64 * 96 * 4,)
    assert (a.view(np.uint64) == b.view(np.uint64)).all()   # disjoint strips, rows copied into place: no arithmetic at all
    assert (a.reshape(-1, 4)[:, 3] > 0).all()


def test_sample_partition_film_reduce_gloo(built, tmp_path):
    a, b = _run(tmp_path, "samples", os.path.join(GOLDEN, "instances.pbrt"))
    # identical sample sets; only the order of the double-precision additions differs
    assert np.allclose(a, b, rtol=1e-12, atol=0)
    assert (a.reshape(-1, 4)[:, 3] > 0).all()


# ---------------------------------------------------------------------------------------------------------------------
# The same N > 1 path with the PRODUCT renderer: two processes on the one GPU of the box — each with a context of its
# own on cuda:0, its strip set, its queues sized for its rows (create_renderer(strips=...)), multigpu.render_partition — and the strip
# gather over gloo (RCCL refuses two ranks on one device; the collective's payload and the row bookkeeping are the same).  The film on
# rank 0 must be bit-identical with a single-context render of the same scene.
GPU_WORKER = textwrap.dedent("""
    import importlib.util, os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    root, scene_path, outdir = sys.argv[1:4]
    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    wfpt = load("wfpt", os.path.join(root, "pbrt-v4_amd", "wfpt.py"))
    multigpu = load("multigpu", os.path.join(root, "pbrt-v4_amd", "multigpu.py"))
    spp = 4
    s = wfpt.Scene(path=scene_path, spp=spp)
    s.create_renderer(0, strips=(rank, world, multigpu.STRIP_HEIGHT))
    s.clear_film()
    multigpu.render_partition(s, rank, world, 0, spp, "strips")
    film = torch.from_numpy(np.ascontiguousarray(s.film()))
    H = film.shape[0]
    owned = np.where(film[..., 3].numpy().sum(axis=1) > 0)[0]
    assert (owned == multigpu.strip_rows(rank, world, H)).all(), (rank, owned)
    film = multigpu.gather_film(film, dist, rank, world, 0)
    if rank == 0:
        film.numpy().tofile(os.path.join(outdir, "film_product_gathered.bin"))
    s.close()
    dist.barrier()
    dist.destroy_process_group()
""")


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["instances", "cornell64"])
def test_strip_partition_product_renderer_two_ranks_one_gpu(wfpt, tmp_path, scene):
    worker = tmp_path / "gpu_worker.py"
    worker.write_text(GPU_WORKER)
    path = os.path.join(GOLDEN, scene + ".pbrt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(worker), ROOT, path, str(tmp_path)]
    subprocess.run(cmd, check=True, timeout=900, cwd=ROOT)
    s = wfpt.Scene(path=path, spp=4)
    s.create_renderer(0)
    s.clear_film()
    s.render(0, 4, 1)
    single = np.ascontiguousarray(s.film()).astype(np.float64)
    s.close()
    got = np.fromfile(tmp_path / "film_product_gathered.bin", dtype=np.float64)
    assert got.size == single.size
    assert (got.view(np.uint64) == single.reshape(-1).view(np.uint64)).all()
    assert (single[..., 3] > 0).all()
