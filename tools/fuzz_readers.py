#!/usr/bin/env python3
"""tools/fuzz_readers.py — mutation fuzzing of the host library's file readers (images, PLY, NanoVDB, measured BRDFs, scene
files): a malformed file must come back as an error (WfError / SceneError), never as a crash of the embedding process.

    python tools/fuzz_readers.py [--iters N] [--seed S] [--only png,exr,...]

Seeds are the fixtures under tests/golden and files the image tests' own encoders write.  Every mutated file is read in a
forked child (the libraries are loaded once in the parent); a child killed by a signal is a finding: the offending file is kept
under /tmp/wf_fuzz_findings/ and the script exits 1.  Mutations: bit flips, byte splats (0x00 / 0xff / 0x7f / 0x80), 32-bit
little- and big-endian extreme values at random offsets, truncation, duplication of a random slice, and (text formats) token
replacement with extreme numbers."""
import argparse
import os
import random
import shutil
import signal
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
EXTREME = [0, 1, 0x7fffffff, 0x80000000, 0xffffffff, 0xfffffffe, 0x7ffffffe, 65535, 65536, 0x10000000, 0x40000000]
TEXT_EXTREME = ["0", "-1", "1e38", "-1e38", "1e-45", "nan", "inf", "2147483647", "-2147483648", "4294967296", "1e308", "", "[", "]", '"', "#"]


def mutate(data, rng, text):
    b = bytearray(data)
    if not b:
        return bytes(b)
    for _ in range(rng.choice([1, 1, 2, 3, 8])):
        k = rng.randrange(8 if text else 7)
        pos = rng.randrange(len(b))
        if k == 0:
            b[pos] ^= 1 << rng.randrange(8)
        elif k == 1:
            b[pos] = rng.choice([0, 0xff, 0x7f, 0x80, rng.randrange(256)])
        elif k == 2 and len(b) >= 4:
            pos = rng.randrange(len(b) - 3)
            b[pos:pos + 4] = rng.choice(EXTREME).to_bytes(4, rng.choice(["little", "big"]))
        elif k == 3:
            b = b[:pos]
            if not b:
                break
        elif k == 4:
            a = rng.randrange(len(b))
            n = rng.randrange(1, min(64, len(b) - a) + 1)
            b[pos:pos] = b[a:a + n]
        elif k == 5 and len(b) >= 8:
            pos = rng.randrange(len(b) - 7)
            b[pos:pos + 8] = rng.choice([0, 0xffffffffffffffff, 0x7fffffffffffffff, 1 << 40, 1 << 33]).to_bytes(8, "little")
        elif k == 6:
            n = rng.randrange(1, min(32, len(b) - pos) + 1)
            del b[pos:pos + n]
            if not b:
                break
        elif k == 7:
            toks = bytes(b).split(b" ")
            if len(toks) > 1:
                toks[rng.randrange(len(toks))] = rng.choice(TEXT_EXTREME).encode()
                b = bytearray(b" ".join(toks))
    return bytes(b)


def run_child(fn, path):
    """0 = read or clean error, else the signal that killed the child"""
    pid = os.fork()
    if pid == 0:
        try:
            devnull = os.open(os.devnull, os.O_WRONLY)
            os.dup2(devnull, 1); os.dup2(devnull, 2)
            signal.alarm(20)
            fn(path)
        except BaseException:
            pass
        os._exit(0)
    _, status = os.waitpid(pid, 0)
    return os.WTERMSIG(status) if os.WIFSIGNALED(status) else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    import numpy as np
    from conftest import load_pkg
    import test_image_io as tio
    wfpt = load_pkg()
    host, _ = wfpt.libs()
    work = tempfile.mkdtemp(prefix="wf_fuzz_")
    rng = random.Random(a.seed)
    nrng = np.random.default_rng(a.seed)

    seeds = []   # (kind, path, reader, is_text)

    def img(path):
        wfpt.read_image(path)

    def scene_file(path):
        s = wfpt.Scene(path=path, spp=1)
        s.close()

    for f in sorted(os.listdir(GOLDEN)):
        p = os.path.join(GOLDEN, f)
        if f.endswith(".png"):
            seeds.append(("png", p, img, False))
        elif f in ("alpha.pfm", "normal.pfm"):
            seeds.append(("pfm", p, img, False))
    # EXR (scan-line and tiled, every compression the reader has), QOI, HDR, TGA from the tests' own encoders
    px = nrng.random((9, 13, 3)).astype(np.float32)
    chans = {"R": px[..., 0], "G": px[..., 1], "B": px[..., 2]}
    for comp in (0, 1, 2, 3):
        for half in (False, True):
            p = os.path.join(work, "s_%d_%d.exr" % (comp, half))
            tio._write_exr(p, chans, comp, half)
            seeds.append(("exr", p, img, False))
            p = os.path.join(work, "t_%d_%d.exr" % (comp, half))
            tio._write_tiled_exr(p, chans, comp, half, (4, 5), level_mode=1 if comp == 3 else 0)
            seeds.append(("exr", p, img, False))
    p = os.path.join(work, "a.qoi")
    open(p, "wb").write(tio._qoi_encode(tio._smooth_image(nrng, 9, 13, 4), 0))
    seeds.append(("qoi", p, img, False))
    p = os.path.join(work, "a.hdr")
    open(p, "wb").write(tio._hdr_encode(nrng.integers(0, 256, (6, 40, 4)).astype(np.uint8), True))
    seeds.append(("hdr", p, img, False))
    for bpp, rle, pal in ((24, True, False), (32, False, False), (8, True, True)):
        p = os.path.join(work, "a_%d_%d.tga" % (bpp, rle))
        open(p, "wb").write(tio._tga_encode(nrng.integers(0, 256, (7, 9, 4 if bpp == 32 else 3)).astype(np.uint8), bpp, rle, True, palette=pal))
        seeds.append(("tga", p, img, False))
    # PLY (binary + ascii), read through a scene that names the file
    def ply_scene(path):
        s = wfpt.Scene(text='Film "rgb" "integer xresolution" 4 "integer yresolution" 4 "string filename" "x.pfm"\nWorldBegin\nLightSource "infinite"\n'
                            'Shape "plymesh" "string filename" "%s"\n' % path, spp=1)
        s.close()
    for f in ("bilinear_quads.ply", "displace_cone.ply", "ball.ply.gz"):
        seeds.append(("ply", os.path.join(GOLDEN, f), ply_scene, f.endswith(".ply") and b"ascii" in open(os.path.join(GOLDEN, f), "rb").read(64)))
    # measured BRDF tensor files
    def bsdf_scene(path):
        s = wfpt.Scene(text='Film "rgb" "integer xresolution" 4 "integer yresolution" 4 "string filename" "x.pfm"\nWorldBegin\nLightSource "infinite"\n'
                            'Material "measured" "string filename" "%s"\nShape "sphere"\n' % path, spp=1)
        s.close()
    for f in ("measured_iso.bsdf", "measured_aniso.bsdf"):
        seeds.append(("bsdf", os.path.join(GOLDEN, f), bsdf_scene, False))
    # NanoVDB
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import make_nanovdb
        p = os.path.join(work, "a.nvdb")
        vals = nrng.random((12, 10, 9)).astype(np.float32)
        make_nanovdb.write_nanovdb(p, "density", vals, (0, 0, 0)) if hasattr(make_nanovdb, "write_nanovdb") else None
        if os.path.exists(p):
            seeds.append(("nvdb", p, lambda q: wfpt.read_nanovdb(q, "density"), False))
    except Exception as e:   # the generator's entry point differs: skip the format, say so
        print("nanovdb seeds skipped:", e)
    # scene files (text)
    for f in ("parser_torture.pbrt", "materials_lights.pbrt", "quadrics.pbrt", "curves.pbrt", "media_box.pbrt", "instances.pbrt", "realistic_camera.pbrt",
              "film_sensor.pbrt", "textures_extra.pbrt", "loopsubdiv.pbrt", "bilinear_lights.pbrt", "hair.pbrt", "subsurface.pbrt"):
        seeds.append(("pbrt", os.path.join(GOLDEN, f), scene_file, True))

    only = set(filter(None, a.only.split(",")))
    findings = 0
    counts = {}
    for kind, path, fn, text in seeds:
        if only and kind not in only:
            continue
        data = open(path, "rb").read()
        ext = "".join(os.path.splitext(path)[1:]) if not path.endswith(".gz") else ".ply.gz"
        # a scene file is mutated next to its original so that relative file names keep resolving
        out_dir = os.path.dirname(path) if kind == "pbrt" else work
        for i in range(a.iters):
            m = mutate(data, rng, text)
            q = os.path.join(out_dir, "_fuzz_%d%s" % (os.getpid(), ext))
            open(q, "wb").write(m)
            sig = run_child(fn, q)
            counts[kind] = counts.get(kind, 0) + 1
            if sig:
                findings += 1
                os.makedirs("/tmp/wf_fuzz_findings", exist_ok=True)
                keep = "/tmp/wf_fuzz_findings/%s_%d_sig%d%s" % (kind, findings, sig, ext)
                shutil.copy(q, keep)
                print("CRASH (signal %d): %s mutated from %s" % (sig, keep, path), flush=True)
            os.unlink(q)
    shutil.rmtree(work, ignore_errors=True)
    print("fuzzed:", counts, "findings:", findings)
    sys.exit(1 if findings else 0)


if __name__ == "__main__":
    main()
