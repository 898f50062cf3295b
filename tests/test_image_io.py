"""Image::Read of the host library (wfh_read_image: .png and .exr decoders, ColorEncoding) against pixel values known in
Python.  The PNG fixtures are the committed tests/golden/png_*.png (tools/make_png_fixtures.py); the EXR files are written
here with zlib (half / float, NONE / RLE / ZIPS / ZIP), so a decode must reproduce the arrays bit for bit."""
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import GOLDEN


def _png_pixels(path):
    """Decode a PNG with Python only (zlib + the five filters): (array [h][w][samples], colour type, bit depth, palette)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, plte = 8, b"", None
    while pos < len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if t == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
        elif t == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif t == b"IDAT":
            idat += body
        pos += 12 + n
    nc = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    stride = (w * nc * depth + 7) // 8
    bpp = max(1, nc * depth // 8)
    raw = zlib.decompress(idat)
    rows, prev = [], bytearray(stride)
    for y in range(h):
        ft = raw[(stride + 1) * y]
        cur = bytearray(raw[(stride + 1) * y + 1:(stride + 1) * (y + 1)])
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 4:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            else:
                pred = [0, a, b, (a + b) >> 1][ft]
            cur[i] = (cur[i] + pred) & 255
        rows.append(bytes(cur))
        prev = cur
    if depth == 16:
        px = np.frombuffer(b"".join(rows), ">u2").reshape(h, w, nc).astype(np.uint32)
    elif depth == 8:
        px = np.frombuffer(b"".join(rows), np.uint8).reshape(h, w, nc).astype(np.uint32)
    else:
        bits = np.unpackbits(np.frombuffer(b"".join(rows), np.uint8).reshape(h, stride), axis=1)[:, :w * nc * depth]
        px = bits.reshape(h, w * nc, depth).dot(1 << np.arange(depth - 1, -1, -1)).reshape(h, w, nc).astype(np.uint32)
    return px, ctype, depth, plte


def _srgb_lut():
    vals = []
    for line in open(os.path.join(os.path.dirname(GOLDEN), "..", "pbrt-v4_amd", "data", "srgb_to_linear_lut.txt")):
        if not line.startswith("#"):
            vals += [np.float32(float(v)) for v in line.split()]
    return np.array(vals, np.float32)


@pytest.mark.parametrize("name", ["png_rgb8", "png_rgba8", "png_rgba8_opaque", "png_grey8", "png_greya8", "png_pal8", "png_pal4", "png_grey2"])
def test_png_8bit_decodes_to_the_encoded_texels(wfpt, name):
    px, ctype, depth, plte = _png_pixels(os.path.join(GOLDEN, name + ".png"))
    if ctype == 3:
        px = plte[px[..., 0]].astype(np.uint32)
    elif depth < 8:
        px = px * 255 // ((1 << depth) - 1)
    if ctype == 4:
        px = px[..., :1]   # ReadPNG keeps Y of grey + alpha
    lin, fmt = wfpt.read_image(os.path.join(GOLDEN, name + ".png"), "linear")
    assert fmt == 0 and lin.shape == px.shape
    assert (lin == (px.astype(np.float32) / np.float32(255))).all()
    srgb, _ = wfpt.read_image(os.path.join(GOLDEN, name + ".png"))
    assert (srgb == _srgb_lut()[px]).all()
    g, _ = wfpt.read_image(os.path.join(GOLDEN, name + ".png"), "gamma 2.2")
    assert np.allclose(g, (px / 255.0) ** 2.2, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("name", ["png_rgb16", "png_grey16"])
def test_png_16bit_becomes_half(wfpt, name):
    px, ctype, depth, _ = _png_pixels(os.path.join(GOLDEN, name + ".png"))
    assert depth == 16
    lin, fmt = wfpt.read_image(os.path.join(GOLDEN, name + ".png"), "linear")
    assert fmt == 1
    want = (px.astype(np.float32) / np.float32(65535)).astype(np.float16).astype(np.float32)   # Half(v / 65535.f), round to nearest even
    assert (lin == want).all()


def _write_exr(path, chans, compression, half):
    """chans: dict name -> [h][w] array.  Scan-line file, increasing y, one pixel type."""
    names = sorted(chans)
    h, w = chans[names[0]].shape
    dt = np.float16 if half else np.float32
    hdr = struct.pack("<II", 20000630, 2)
    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(val)) + val
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iBxxxii", 1 if half else 2, 0, 1, 1) for n in names) + b"\0"
    hdr += attr("channels", "chlist", chl)
    hdr += attr("compression", "compression", bytes([compression]))
    hdr += attr("dataWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    hdr += attr("displayWindow", "box2i", struct.pack("<iiii", 0, 0, w - 1, h - 1))
    hdr += attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    hdr += attr("screenWindowWidth", "float", struct.pack("<f", 1))
    hdr += b"\0"
    lines = {0: 1, 1: 1, 2: 1, 3: 16}[compression]
    chunks = []
    for y0 in range(0, h, lines):
        raw = b"".join(chans[n][y].astype(dt).tobytes() for y in range(y0, min(h, y0 + lines)) for n in names)
        if compression == 0:
            body = raw
        else:
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]])                       # de-interleave
            d = t.astype(np.int32)
            p = np.concatenate([d[:1], (d[1:] - d[:-1] + 128 + 256) % 256]).astype(np.uint8)   # predictor
            if compression == 1:
                out, i = bytearray(), 0
                pb = p.tobytes()
                while i < len(pb):   # simple RLE: runs of >= 3 equal bytes, literals otherwise
                    j = i
                    while j + 1 < len(pb) and pb[j + 1] == pb[i] and j - i < 126:
                        j += 1
                    if j - i >= 2:
                        out += bytes([j - i, pb[i]]); i = j + 1
                    else:
                        k = i
                        while k < len(pb) and k - i < 127 and not (k + 2 < len(pb) and pb[k] == pb[k + 1] == pb[k + 2]):
                            k += 1
                        out += bytes([(256 - (k - i)) & 255]) + pb[i:k]; i = k
                body = bytes(out)
            else:
                body = zlib.compress(p.tobytes())
            if len(body) >= len(raw):
                body = raw
        chunks.append(struct.pack("<ii", y0, len(body)) + body)
    off = len(hdr) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    open(path, "wb").write(hdr + table + b"".join(chunks))


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
@pytest.mark.parametrize("half", [True, False])
def test_exr_scanline_decode(wfpt, tmp_path, compression, half):
    rng = np.random.default_rng(100 * compression + half)
    h, w = 37, 29
    base = rng.random((h, w)).astype(np.float32)
    chans = {"R": base * 3, "G": np.round(base * 4) / 4, "B": base ** 2, "A": (base > 0.3).astype(np.float32)}
    dt = np.float16 if half else np.float32
    path = str(tmp_path / "t.exr")
    _write_exr(path, chans, compression, half)
    px, fmt = wfpt.read_image(path)
    assert fmt == (1 if half else 2) and px.shape == (h, w, 4)
    for i, n in enumerate("RGBA"):
        assert (px[..., i] == chans[n].astype(dt).astype(np.float32)).all(), n
    _write_exr(path, {"Y": chans["R"]}, compression, half)
    px, _ = wfpt.read_image(path)
    assert px.shape == (h, w, 1) and (px[..., 0] == chans["R"].astype(dt).astype(np.float32)).all()


def test_exr_written_by_the_film_reads_back(wfpt, tmp_path):
    rgb = np.random.default_rng(3).random((9, 13, 3)).astype(np.float32)
    path = str(tmp_path / "o.exr")
    wfpt.write_pfm(path, rgb)   # by extension: .exr
    px, fmt = wfpt.read_image(path)
    assert fmt == 2 and (px == rgb).all()


def test_unsupported_exr_compression_is_an_error_not_a_crash(wfpt, tmp_path):
    path = str(tmp_path / "p.exr")
    _write_exr(path, {"Y": np.zeros((4, 4), np.float32)}, 0, False)
    data = bytearray(open(path, "rb").read())
    i = data.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    data[i] = 4   # PIZ
    open(path, "wb").write(bytes(data))
    with pytest.raises(wfpt.WfError, match="PIZ"):
        wfpt.read_image(path)


def test_scene_with_exr_environment_map_renders_like_the_pfm_one(wfpt, tmp_path):
    """The envmap golden with its sky image converted to a (float, uncompressed) .exr: same pixels, so the CPU checker's
    image must equal the reference's golden render bit for bit."""
    import shutil
    from conftest import read_pfm, run_wf_cpu
    sky, _ = wfpt.read_image(os.path.join(GOLDEN, "sky.pfm"))
    wfpt.write_pfm(str(tmp_path / "sky.exr"), sky)
    text = open(os.path.join(GOLDEN, "envmap.pbrt")).read()
    assert "sky.pfm" in text
    open(tmp_path / "envmap_exr.pbrt", "w").write(text.replace("sky.pfm", "sky.exr"))
    for f in os.listdir(GOLDEN):   # whatever else the scene includes
        if f.endswith(".pfm") and f != "sky.pfm" and f in text:
            shutil.copy(os.path.join(GOLDEN, f), tmp_path / f)
    out = str(tmp_path / "o.pfm")
    run_wf_cpu(str(tmp_path / "envmap_exr.pbrt"), out, 4)
    ref = read_pfm(os.path.join(GOLDEN, "envmap_ref.pfm"))
    assert (read_pfm(out).view(np.uint32) == ref.view(np.uint32)).all()


def test_ply_ascii_little_big_endian_and_gzip_load_the_same_mesh(wfpt, tmp_path):
    """One mesh written as ASCII, binary little-endian, binary big-endian and gzipped PLY: the CPU checker's renders of the four
    scenes are bit-identical (the readers feed the same vertex and index arrays to the scene tables)."""
    import gzip
    from conftest import read_pfm, run_wf_cpu
    rng = np.random.default_rng(5)
    n = 6
    P = (rng.random((n * n, 3)).astype(np.float32) * 0.1 + np.stack(np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n)), -1).reshape(-1, 2).astype(np.float32) @ np.array([[1, 0, 0], [0, 1, 0]], np.float32)).astype(np.float32)
    F = [(r * n + c, r * n + c + 1, (r + 1) * n + c) for r in range(n - 1) for c in range(n - 1)] + [(r * n + c + 1, (r + 1) * n + c + 1, (r + 1) * n + c) for r in range(n - 1) for c in range(n - 1)]
    hdr = "ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
    def binary(end):
        return (hdr % ("binary_%s_endian" % ("little" if end == "<" else "big"), len(P), len(F))).encode() + \
            b"".join(struct.pack(end + "3f", *p) for p in P) + b"".join(struct.pack(end + "B3i", 3, *f) for f in F)
    files = {"a.ply": (hdr % ("ascii", len(P), len(F))).encode() + "".join("%.9g %.9g %.9g\n" % tuple(p) for p in P).encode() + "".join("3 %d %d %d\n" % f for f in F).encode(),
             "l.ply": binary("<"), "b.ply": binary(">"), "g.ply.gz": gzip.compress(binary("<"))}
    imgs = []
    for name, data in files.items():
        open(tmp_path / name, "wb").write(data)
        scene = ('LookAt 0 0 4  0 0 0  0 1 0\nCamera "perspective" "float fov" 40\nSampler "independent" "integer pixelsamples" 2\n'
                 'Film "rgb" "integer xresolution" 32 "integer yresolution" 32 "string filename" "o.pfm"\nWorldBegin\n'
                 'LightSource "distant" "point3 from" [1 2 4] "point3 to" [0 0 0] "rgb L" [3 3 3]\n'
                 'Material "diffuse" "rgb reflectance" [0.6 0.5 0.4]\nShape "plymesh" "string filename" "%s"\n' % name)
        open(tmp_path / (name + ".pbrt"), "w").write(scene)
        out = str(tmp_path / (name + ".pfm"))
        run_wf_cpu(str(tmp_path / (name + ".pbrt")), out, 2)
        imgs.append(read_pfm(out))
    assert imgs[0].max() > 0
    for im in imgs[1:]:
        assert (im.view(np.uint32) == imgs[0].view(np.uint32)).all()
