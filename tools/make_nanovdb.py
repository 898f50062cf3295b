#!/usr/bin/env python3
"""tools/make_nanovdb.py — writes NanoVDB files (.nvdb) of float grids from dense numpy arrays: the fixtures of the NanoVDB-medium tests.

The layout is the one pbrt-v4_amd/csrc/host/nanovdb_io.cpp reads (NanoVDB 32.x ABI as restated there: FileHeader, FileMetaData,
GridData 672 B, TreeData 64 B, RootData + 32-byte tiles, InternalNode<5> / InternalNode<4> with value and child masks and 8-byte
table entries holding child offsets relative to the node, LeafNode<float> with 512 values).  PARITY UNPINNED: NanoVDB itself is not
available in this environment, so neither this writer nor the reader has been checked against a file NanoVDB produced.

    write_nvdb(path, [("density", array[nz][ny][nx], origin_index(x, y, z), voxel_size, translation)], codec="none" | "zip")

Voxels whose value differs from the background (0) are active; leaves are created for every 8^3 block with an active voxel, every
other region is background (no constant tiles are written)."""
import struct
import zlib

import numpy as np

MAGIC = 0x304244566F6E614E
VERSION = (32 << 21) | (3 << 10) | 3


def _internal_header(log2dim):
    return (24 + 8 + 2 * (1 << (3 * log2dim)) // 8 + 16 + 31) // 32 * 32


def _grid_blob(name, vals, origin, voxel, trans, grid_class=2):
    """vals[z][y][x] float32; voxel (i, j, k) of the array = index coordinate origin + (i, j, k)."""
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    nz, ny, nx = vals.shape
    ox, oy, oz = origin
    active = vals != 0
    if not active.any():
        raise ValueError("grid %s has no active voxel" % name)
    zz, yy, xx = np.nonzero(active)
    bmin = (int(xx.min()) + ox, int(yy.min()) + oy, int(zz.min()) + oz)
    bmax = (int(xx.max()) + ox, int(yy.max()) + oy, int(zz.max()) + oz)

    def value(x, y, z):   # index coordinates -> value (0 outside the array)
        i, j, k = x - ox, y - oy, z - oz
        if 0 <= i < nx and 0 <= j < ny and 0 <= k < nz:
            return float(vals[k, j, i])
        return 0.0

    # leaves keyed by their origin; then lower nodes (128^3), upper nodes (4096^3)
    leaves = {}
    for x, y, z in zip(xx + ox, yy + oy, zz + oz):
        leaves.setdefault((int(x) & ~7, int(y) & ~7, int(z) & ~7), None)
    lowers, uppers = {}, {}
    for lo in leaves:
        lowers.setdefault(tuple(c & ~127 for c in lo), []).append(lo)
    for lo in lowers:
        uppers.setdefault(tuple(c & ~4095 for c in lo), []).append(lo)
    upper_keys = sorted(uppers)
    lower_keys = [k for u in upper_keys for k in sorted(uppers[u])]
    leaf_keys = [k for l in lower_keys for k in sorted(lowers[l])]
    UP, LOW, LEAF = _internal_header(5) + 8 * 32768, _internal_header(4) + 8 * 4096, 96 + 4 * 512
    root_size = 64 + 32 * len(upper_keys)
    # buffer order (as NanoVDB lays a grid out): grid, tree, root, upper nodes, lower nodes, leaves
    off_root = 672 + 64
    off_upper = off_root + root_size
    off_lower = off_upper + UP * len(upper_keys)
    off_leaf = off_lower + LOW * len(lower_keys)
    total = off_leaf + LEAF * len(leaf_keys)
    buf = bytearray(total)
    upper_at = {k: off_upper + UP * i for i, k in enumerate(upper_keys)}
    lower_at = {k: off_lower + LOW * i for i, k in enumerate(lower_keys)}
    leaf_at = {k: off_leaf + LEAF * i for i, k in enumerate(leaf_keys)}
    vmin, vmax = float(vals[active].min()), float(vals[active].max())

    for k, at in leaf_at.items():
        block = np.zeros((8, 8, 8), np.float32)   # [x][y][z]: LeafNode::CoordToOffset = x << 6 | y << 3 | z
        mask = 0
        for a in range(8):
            for b in range(8):
                for c in range(8):
                    v = value(k[0] + a, k[1] + b, k[2] + c)
                    block[a, b, c] = v
                    if v != 0:
                        mask |= 1 << ((a << 6) | (b << 3) | c)
        struct.pack_into("<3i3BB", buf, at, k[0], k[1], k[2], 7, 7, 7, 0)
        buf[at + 16:at + 80] = mask.to_bytes(64, "little")
        struct.pack_into("<4f", buf, at + 80, float(block.min()), float(block.max()), float(block.mean()), float(block.std()))
        buf[at + 96:at + 96 + 2048] = block.tobytes()

    def internal(at, key, log2dim, children, child_at, child_size):
        n_entries = 1 << (3 * log2dim)
        mask_bytes = n_entries // 8
        size = child_size * (1 << log2dim)
        struct.pack_into("<6i", buf, at, key[0], key[1], key[2], key[0] + size - 1, key[1] + size - 1, key[2] + size - 1)
        struct.pack_into("<Q", buf, at + 24, 0)
        child_mask = 0
        table = at + _internal_header(log2dim)
        for c in children:
            n = (((c[0] - key[0]) // child_size) << (2 * log2dim)) | (((c[1] - key[1]) // child_size) << log2dim) | ((c[2] - key[2]) // child_size)
            child_mask |= 1 << n
            struct.pack_into("<q", buf, table + 8 * n, child_at[c] - at)
        buf[at + 32:at + 32 + mask_bytes] = (0).to_bytes(mask_bytes, "little")                                     # value mask: no active tiles
        buf[at + 32 + mask_bytes:at + 32 + 2 * mask_bytes] = child_mask.to_bytes(mask_bytes, "little")
        struct.pack_into("<4f", buf, at + 32 + 2 * mask_bytes, vmin, vmax, 0.0, 0.0)

    for k, at in lower_at.items():
        internal(at, k, 4, lowers[k], leaf_at, 8)
    for k, at in upper_at.items():
        internal(at, k, 5, uppers[k], lower_at, 128)
    # root: bbox, table size, background / min / max / average / deviation, then the tiles
    struct.pack_into("<6iI5f", buf, off_root, *bmin, *bmax, len(upper_keys), 0.0, vmin, vmax, 0.0, 0.0)
    for i, k in enumerate(upper_keys):
        key = ((((k[2] & 0xFFFFFFFF) >> 12)) | (((k[1] & 0xFFFFFFFF) >> 12) << 21) | (((k[0] & 0xFFFFFFFF) >> 12) << 42))
        struct.pack_into("<QqIf", buf, off_root + 64 + 32 * i, key, upper_at[k] - off_root, 0, 0.0)
    # tree: byte offsets from the TreeData to the first leaf / lower / upper node and the root; node and tile counts; active voxels
    struct.pack_into("<4Q3I3IQ", buf, 672, off_leaf - 672, off_lower - 672, off_upper - 672, off_root - 672, len(leaf_keys), len(lower_keys), len(upper_keys),
                     0, 0, 0, int(active.sum()))
    # grid: Map = uniform scale `voxel` + translation `trans` (world = voxel * index + trans)
    mat = [voxel, 0, 0, 0, voxel, 0, 0, 0, voxel]
    inv = [1.0 / voxel, 0, 0, 0, 1.0 / voxel, 0, 0, 0, 1.0 / voxel]
    wmin = [voxel * b + t for b, t in zip(bmin, trans)]
    wmax = [voxel * (b + 1) + t for b, t in zip(bmax, trans)]
    nm = name.encode()[:255]
    struct.pack_into("<QQIIIIQ", buf, 0, MAGIC, 0, VERSION, 0, 0, 1, total)
    buf[40:40 + len(nm)] = nm
    struct.pack_into("<9f9f3ff9d9d3dd", buf, 296, *[float(np.float32(m)) for m in mat], *[float(np.float32(m)) for m in inv], *[float(np.float32(t)) for t in trans], 1.0,
                     *mat, *inv, *trans, 1.0)
    struct.pack_into("<6d3dIIqIIQQ", buf, 296 + 264, *wmin, *wmax, voxel, voxel, voxel, grid_class, 1, 0, 0, 0, 0, 0)
    meta = dict(gridSize=total, voxelCount=int(active.sum()), worldBBox=wmin + wmax, indexBBox=list(bmin) + list(bmax), voxelSize=[voxel] * 3,
                nodeCount=[len(leaf_keys), len(lower_keys), len(upper_keys), 1], gridClass=grid_class)
    return bytes(buf), meta


def write_nvdb(path, grids, codec="none"):
    """grids: [(name, values[z][y][x], origin (x, y, z) in index space, voxel size, translation (x, y, z))]"""
    blobs = [_grid_blob(*g) for g in grids]
    code = {"none": 0, "zip": 1}[codec]
    with open(path, "wb") as f:
        f.write(struct.pack("<QIHH", MAGIC, VERSION, len(grids), code))
        payload = []
        for (name, *_), (blob, m) in zip(grids, blobs):
            data = blob if code == 0 else struct.pack("<Q", len(zlib.compress(blob))) + zlib.compress(blob)
            payload.append(data)
            nm = name.encode() + b"\0"
            f.write(struct.pack("<4QII6d6i3dI4I3IHHI", m["gridSize"], len(data), 0, m["voxelCount"], 1, m["gridClass"], *m["worldBBox"], *m["indexBBox"], *m["voxelSize"],
                                len(nm), *m["nodeCount"], 0, 0, 0, code, 0, VERSION))
            f.write(nm)
        for data in payload:
            f.write(data)


def smoke_grid(n=40, seed=9):
    """a small puff of smoke (density) and a hot core (temperature) for the parity fixtures; index origin (-20, -6, -20)"""
    rng = np.random.RandomState(seed)
    z, y, x = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n), np.linspace(-1, 1, n), indexing="ij")
    r2 = x * x + (y * 1.3) ** 2 + z * z
    noise = rng.rand(n, n, n).astype(np.float32)
    dens = np.clip(1.2 * (1 - r2) * (0.6 + 0.4 * noise), 0, 1).astype(np.float32)
    dens[dens < 0.05] = 0
    temp = np.where(r2 < 0.25, 1800 + 1500 * (1 - 4 * r2), 0).astype(np.float32)
    return dens, temp


if __name__ == "__main__":
    import sys
    d, t = smoke_grid()
    write_nvdb(sys.argv[1] if len(sys.argv) > 1 else "smoke.nvdb", [("density", d, (-20, -6, -20), 0.05, (0.0, 0.3, 0.0)), ("temperature", t, (-20, -6, -20), 0.05, (0.0, 0.3, 0.0))],
               codec=sys.argv[2] if len(sys.argv) > 2 else "none")
