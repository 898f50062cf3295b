"""multigpu.py — the image partition and film reduce of multi-GPU rendering (north_star: "the image is tiled
across the 8 GPUs of one node; RCCL over xGMI only for the final film reduce").

One process per GPU (torch.distributed, backend "nccl" = RCCL).  The scene is replicated; rank r renders the scanline strips
r, r + N, r + 2N, ... of STRIP_HEIGHT lines each — interleaved, so that sky and foliage are dealt evenly — for EVERY sample
index, in full-size wavefronts (a pass carries many sample indices of the rank's lines).  Each rank keeps a full-size film
whose foreign lines stay zero; one `reduce(SUM)` to rank 0 is therefore a gather, and the result is bit-identical to the
single-GPU film (every pixel's double-precision sums are formed on one rank, in the same order).  No collective runs
during rendering.  The kernels' side of the partition is wf_set_strips (include/wf_abi.h).

The alternative partition (sample indices r, r + N, ...: `partition="samples"`) needs a true sum and changes the order of
the double additions (1e-16 relative); it is kept for comparison."""
import numpy as np

STRIP_HEIGHT = 16


def strip_rows(rank, world, height, strip=STRIP_HEIGHT):
    """image scanlines (0-based, relative to the film's pixel bounds) owned by `rank` of `world`"""
    y = np.arange(height)
    return y[(y // strip) % world == rank]


def render_partition(scene, rank, world, sample_begin, sample_end, partition="strips"):
    """render this rank's share of sample indices [sample_begin, sample_end) into the scene's film; returns seconds"""
    if partition == "strips":
        scene.set_strips(rank, world, STRIP_HEIGHT)
        return scene.render(sample_begin, sample_end, 1)
    if partition == "samples":
        scene.set_strips(0, 1, STRIP_HEIGHT)
        return scene.render(sample_begin + rank, sample_end, world)
    raise ValueError("unknown partition %r" % partition)


def gather_film(film, dist, rank, world, dst=0, strip=STRIP_HEIGHT):
    """The exchange of the strip partition: every rank sends ONLY the scanlines it owns (H / world of them: 1 / world of the film —
    the reduce(SUM) of full films this replaces moved `world` films of mostly zeros) and rank `dst` puts them in place.
    film: the rank's [H, W, 4] float64 accumulators as a torch tensor (CUDA for nccl = RCCL, CPU for gloo); after the call rank `dst`
    holds the whole image's film, bit-identical to a single-GPU render (no arithmetic: rows are copied).  The collective is issued
    even at world size 1 (a one-rank RCCL gather), so that a single-GPU box exercises the library path."""
    import torch
    if dist is None:
        return film
    H = film.shape[0]
    rows = [torch.as_tensor(strip_rows(r, world, H, strip), device=film.device) for r in range(world)]
    most = max(len(r) for r in rows)
    mine = torch.zeros((most,) + tuple(film.shape[1:]), dtype=film.dtype, device=film.device)   # equal-sized pieces: ranks own 67 or 68 strips' worth
    mine[:len(rows[rank])] = film.index_select(0, rows[rank])
    pieces = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, pieces, dst=dst)
    if rank == dst:
        for r in range(world):
            if r != dst:
                film.index_copy_(0, rows[r], pieces[r][:len(rows[r])])
    return film


def reduce_film(film, dist, dst=0):
    """film: the rank's [H, W, 4] float64 accumulators as a torch tensor (CUDA for nccl, CPU for gloo).  After the call rank
    `dst` holds the whole image's film.  With the strip partition the addends of every element are zero on all ranks but one."""
    if dist is not None and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film
