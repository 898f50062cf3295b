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


def reduce_film(film, dist, dst=0):
    """film: the rank's [H, W, 4] float64 accumulators as a torch tensor (CUDA for nccl, CPU for gloo).  After the call rank
    `dst` holds the whole image's film.  With the strip partition the addends of every element are zero on all ranks but one."""
    if dist is not None and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film
