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


# ---- .qoi / .hdr / .tga (csrc/host/image_formats.cpp): files written here from known pixels; the decoders of the reference (ext/qoi,
# stb_image) are absent submodules, so these known-answer tests are what pins the decode -------------------------------------------
def _qoi_encode(px, colorspace):
    """QOI 1.0 encoder (all six chunk types) for an [h][w][3|4] uint8 array."""
    h, w, nc = px.shape
    out = bytearray(b"qoif" + struct.pack(">IIBB", w, h, nc, colorspace))
    index = [(0, 0, 0, 0)] * 64
    prev = (0, 0, 0, 255)
    run = 0
    flat = px.reshape(-1, nc)
    for k in range(flat.shape[0]):
        r, g, b = (int(v) for v in flat[k][:3])
        a = int(flat[k][3]) if nc == 4 else 255
        cur = (r, g, b, a)
        if cur == prev:
            run += 1
            if run == 62 or k == flat.shape[0] - 1:
                out.append(0xc0 | (run - 1))
                run = 0
            continue
        if run:
            out.append(0xc0 | (run - 1))
            run = 0
        ip = (r * 3 + g * 5 + b * 7 + a * 11) % 64
        if index[ip] == cur:
            out.append(ip)
        else:
            index[ip] = cur
            if a == prev[3]:
                vr, vg, vb = ((r - prev[0] + 128) % 256) - 128, ((g - prev[1] + 128) % 256) - 128, ((b - prev[2] + 128) % 256) - 128
                if -2 <= vr <= 1 and -2 <= vg <= 1 and -2 <= vb <= 1:
                    out.append(0x40 | (vr + 2) << 4 | (vg + 2) << 2 | (vb + 2))
                elif -32 <= vg <= 31 and -8 <= vr - vg <= 7 and -8 <= vb - vg <= 7:
                    out += bytes([0x80 | (vg + 32), (vr - vg + 8) << 4 | (vb - vg + 8)])
                else:
                    out += bytes([0xfe, r, g, b])
            else:
                out += bytes([0xff, r, g, b, a])
        prev = cur
    return bytes(out) + b"\0" * 7 + b"\1"


def _smooth_image(rng, h, w, nc):
    """an image with runs, small differences, repeated colours and noise: every QOI chunk type / RLE packet kind occurs"""
    base = np.cumsum(rng.integers(-3, 4, size=(h, w, nc)), axis=1) + rng.integers(0, 256, size=(h, 1, nc))
    px = (base % 256).astype(np.uint8)
    px[:, w // 3: w // 3 + 9] = px[:, w // 3: w // 3 + 1]          # runs
    px[h // 2:, : w // 4] = rng.integers(0, 256, size=(h - h // 2, w // 4, nc))   # noise
    px[1::2, w // 2:] = px[0:-1:2, w // 2:] if h % 2 == 0 else px[1::2, w // 2:]  # repeated colours
    return px


@pytest.mark.parametrize("nc,colorspace", [(3, 0), (4, 0), (3, 1), (4, 1)])
def test_qoi_decodes_to_the_encoded_texels(wfpt, tmp_path, nc, colorspace):
    rng = np.random.default_rng(7 + nc + colorspace)
    px = _smooth_image(rng, 24, 70, nc)
    if nc == 4:
        px[..., 3] = np.where(rng.random((24, 70)) < 0.8, 255, px[..., 3])
    path = str(tmp_path / "t.qoi")
    open(path, "wb").write(_qoi_encode(px, colorspace))
    img, fmt = wfpt.read_image(path, "gamma 2.2")   # (the file's colour-space byte decides, not the caller's encoding: ReadQOI)
    assert fmt == 0 and img.shape == px.shape
    want = px.astype(np.float32) / np.float32(255) if colorspace == 1 else _srgb_lut()[px]
    assert (img == want).all()


def _hdr_encode(rgbe, rle):
    h, w, _ = rgbe.shape
    out = bytearray(b"#?RADIANCE\n# written by the test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1\n\n" + ("-Y %d +X %d\n" % (h, w)).encode())
    for y in range(h):
        if not rle:
            out += rgbe[y].tobytes()
            continue
        out += bytes([2, 2, w >> 8, w & 255])
        for k in range(4):
            row = rgbe[y, :, k].tobytes()
            i = 0
            while i < w:
                j = i
                while j + 1 < w and row[j + 1] == row[i] and j - i < 126:
                    j += 1
                if j - i >= 3:
                    out += bytes([128 + (j - i + 1), row[i]])
                    i = j + 1
                else:
                    j = i
                    while j < w and j - i < 127 and not (j + 3 < w and row[j] == row[j + 1] == row[j + 2] == row[j + 3]):
                        j += 1
                    j = max(j, i + 1)
                    out += bytes([j - i]) + row[i:j]
                    i = j
    return bytes(out)


@pytest.mark.parametrize("w,rle", [(40, True), (40, False), (6, False)])
def test_hdr_decodes_like_stb(wfpt, tmp_path, w, rle):
    rng = np.random.default_rng(11 + w)
    h = 9
    rgbe = rng.integers(0, 256, size=(h, w, 4)).astype(np.uint8)
    rgbe[..., 3] = rng.integers(100, 150, size=(h, w))
    rgbe[2, 3:20] = rgbe[2, 3]          # runs in every component plane
    rgbe[4, :, 3] = 0                   # exponent 0: black
    if not rle:
        rgbe[0, 0, 0] = 200             # (a flat file whose first bytes looked like an RLE header would be read as one)
    path = str(tmp_path / "t.hdr")
    open(path, "wb").write(_hdr_encode(rgbe, rle))
    img, fmt = wfpt.read_image(path)
    assert fmt == 2 and img.shape == (h, w, 3)
    f1 = np.ldexp(np.float32(1), rgbe[..., 3].astype(np.int32) - 136).astype(np.float32)
    want = np.where(rgbe[..., 3:4] != 0, rgbe[..., :3].astype(np.float32) * f1[..., None], np.float32(0))
    assert (img == want).all()


def _tga_encode(px, bpp, rle, top_down, grey=False, palette=False):
    """px: [h][w][nc] uint8 R G B (A) or Y (A); 16 bpp = 5-5-5 colour (or grey + alpha when grey)."""
    h, w, nc = px.shape
    rows = px if top_down else px[::-1]
    def pixel(v):
        if bpp == 8:
            return bytes([int(v[0])])
        if bpp == 16 and grey:
            return bytes([int(v[0]), int(v[1])])
        if bpp == 16:
            r, g, b = int(v[0]) >> 3, int(v[1]) >> 3, int(v[2]) >> 3
            return struct.pack("<H", r << 10 | g << 5 | b)
        return bytes([int(v[2]), int(v[1]), int(v[0])] + ([int(v[3])] if bpp == 32 else []))
    body = bytearray()
    pal = b""
    pal_len = 0
    flat = rows.reshape(-1, nc)
    if palette:
        colours, inv = np.unique(flat, axis=0, return_inverse=True)
        pal_len = len(colours)
        pal = b"".join(bytes([int(c[2]), int(c[1]), int(c[0])]) for c in colours)
        codes = [bytes([int(i)]) for i in np.asarray(inv).reshape(-1)]
    else:
        codes = [pixel(v) for v in flat]
    if not rle:
        body = b"".join(codes)
    else:
        i = 0
        while i < len(codes):
            j = i
            while j + 1 < len(codes) and codes[j + 1] == codes[i] and j - i < 127:
                j += 1
            if j > i:
                body += bytes([0x80 | (j - i)]) + codes[i]
                i = j + 1
            else:
                j = i
                while j + 1 < len(codes) and codes[j + 1] != codes[j] and j - i < 127:
                    j += 1
                body += bytes([j - i]) + b"".join(codes[i:j + 1])
                i = j + 1
    itype = (1 if palette else 3 if grey else 2) + (8 if rle else 0)
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 1 if palette else 0, itype, 0, pal_len, 24 if palette else 0, 0, 0, w, h, 8 if palette else bpp,
                      (0x20 if top_down else 0) | (8 if bpp == 32 else 0))
    return hdr + pal + bytes(body)


@pytest.mark.parametrize("bpp,rle,top_down,grey,palette", [(24, False, False, False, False), (24, True, True, False, False), (32, True, False, False, False),
                                                            (16, False, False, False, False), (16, True, True, False, False), (8, True, False, True, False),
                                                            (16, False, True, True, False), (24, True, False, False, True)])
def test_tga_decodes_like_stb(wfpt, tmp_path, bpp, rle, top_down, grey, palette):
    rng = np.random.default_rng(bpp + 2 * rle + 4 * top_down)
    nc = (2 if bpp == 16 else 1) if grey else (4 if bpp == 32 else 3)
    px = _smooth_image(rng, 13, 37, nc)
    if palette:
        px = (px // 64 * 64).astype(np.uint8)   # few colours
    path = str(tmp_path / "t.tga")
    open(path, "wb").write(_tga_encode(px, bpp, rle, top_down, grey, palette))
    img, fmt = wfpt.read_image(path, "linear")   # (stb's 8-bit loads are taken as sRGB whatever the caller asks for: util/image.cpp:888-916)
    want = px
    if bpp == 16 and not grey:
        want = ((px >> 3).astype(np.uint32) * 255 // 31).astype(np.uint8)
    if grey:
        want = want[..., :1]
    elif nc == 4:
        want = want[..., :3]
    assert fmt == 0 and img.shape == want.shape
    assert (img == _srgb_lut()[want]).all()


@pytest.mark.parametrize("ext", ["qoi", "hdr", "tga"])
def test_malformed_image_files_raise(wfpt, tmp_path, ext):
    rng = np.random.default_rng(3)
    px = _smooth_image(rng, 8, 16, 3)
    good = {"qoi": _qoi_encode(px, 0), "hdr": _hdr_encode(np.concatenate([px, px[..., :1]], axis=2), True), "tga": _tga_encode(px, 24, True, False)}[ext]
    cases = [good[:10], b"", bytes(rng.integers(0, 256, size=200, dtype=np.uint8))]
    if ext != "qoi":
        cases.append(good[: len(good) // 2])   # (qoi_decode itself tolerates a short chunk stream: the last pixel repeats — kept)
    for k, bad in enumerate(cases):
        path = str(tmp_path / ("bad%d.%s" % (k, ext)))
        open(path, "wb").write(bad)
        with pytest.raises(Exception):
            wfpt.read_image(path)


def test_qoi_and_tga_textures_render_like_the_same_png(tmp_path):
    """The 8-bit texels of a .qoi / .tga file enter the texture pipeline exactly as a PNG's do (U256 + sRGB encoding, the MIP pyramid's
    per-level re-quantisation included): one scene rendered with the same pixels from the three containers gives one image.  The PNG path
    is pinned to the reference by the png_textures golden."""
    import shutil
    from conftest import read_pfm, run_wf_cpu
    px, ctype, depth, _ = _png_pixels(os.path.join(GOLDEN, "png_rgb8.png"))
    assert ctype == 2 and depth == 8
    px = px.astype(np.uint8)
    shutil.copy(os.path.join(GOLDEN, "png_rgb8.png"), str(tmp_path / "tex.png"))
    open(str(tmp_path / "tex.qoi"), "wb").write(_qoi_encode(px, 0))
    open(str(tmp_path / "tex.tga"), "wb").write(_tga_encode(px, 24, True, False))
    imgs = {}
    for ext in ("png", "qoi", "tga"):
        scene = str(tmp_path / ("s_%s.pbrt" % ext))
        open(scene, "w").write('''LookAt 0 2.5 -4  0 0 0  0 1 0
Camera "perspective" "float fov" [ 40 ]
Sampler "zsobol" "integer pixelsamples" [ 4 ]
Film "rgb" "integer xresolution" [ 48 ] "integer yresolution" [ 48 ] "string filename" [ "o.pfm" ] "bool savefp16" [ false ]
WorldBegin
LightSource "distant" "point3 from" [ 1 3 -2 ] "point3 to" [ 0 0 0 ] "rgb L" [ 3 3 3 ]
Texture "t" "spectrum" "imagemap" "string filename" [ "tex.%s" ] "float uscale" [ 3 ] "float vscale" [ 3 ]
Material "diffuse" "texture reflectance" "t"
Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point3 P" [-3 0 -3 -3 0 3 3 0 3 3 0 -3] "point2 uv" [0 0 0 1 1 1 1 0]
''' % ext)
        out = str(tmp_path / ("o_%s.pfm" % ext))
        run_wf_cpu(scene, out, spp=4)
        imgs[ext] = read_pfm(out)
    assert imgs["png"].mean() > 0.01
    assert (imgs["qoi"].view(np.uint32) == imgs["png"].view(np.uint32)).all()
    assert (imgs["tga"].view(np.uint32) == imgs["png"].view(np.uint32)).all()


def _exr_block(raw, compression):
    """one block's bytes as the file stores them: zip / rle over the de-interleaved, predicted bytes (kept raw when that is not smaller)"""
    if compression == 0:
        return raw
    t = np.frombuffer(raw, np.uint8)
    t = np.concatenate([t[0::2], t[1::2]])
    d = t.astype(np.int32)
    p = np.concatenate([d[:1], (d[1:] - d[:-1] + 128 + 256) % 256]).astype(np.uint8)
    body = zlib.compress(p.tobytes())
    return body if len(body) < len(raw) else raw


def _write_tiled_exr(path, chans, compression, half, tile, level_mode=0):
    """Single-part TILED file (version flag 0x200, `tiles` attribute): the tiles of level (0, 0) — and, for level_mode 1, one coarser MIP level
    behind them in the offset table, which a reader of the finest level must ignore."""
    names = sorted(chans)
    h, w = chans[names[0]].shape
    dt = np.float16 if half else np.float32
    hdr = struct.pack("<II", 20000630, 2 | 0x200)
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
    hdr += attr("tiles", "tiledesc", struct.pack("<IIB", tile[0], tile[1], level_mode))
    hdr += b"\0"
    def level_chunks(level, lchans):
        lh, lw = lchans[names[0]].shape
        out = []
        for ty in range((lh + tile[1] - 1) // tile[1]):
            for tx in range((lw + tile[0] - 1) // tile[0]):
                x0, y0 = tx * tile[0], ty * tile[1]
                x1, y1 = min(lw, x0 + tile[0]), min(lh, y0 + tile[1])
                raw = b"".join(lchans[n][y, x0:x1].astype(dt).tobytes() for y in range(y0, y1) for n in names)
                body = _exr_block(raw, compression)
                out.append(struct.pack("<iiiii", tx, ty, level, level, len(body)) + body)
        return out
    chunks = level_chunks(0, chans)
    if level_mode == 1:
        chunks += level_chunks(1, {n: chans[n][::2, ::2] * 0 + 7 for n in names})
    off = len(hdr) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    open(path, "wb").write(hdr + table + b"".join(chunks))


@pytest.mark.parametrize("compression,half,tile,level_mode", [(0, False, (16, 16), 0), (3, True, (16, 8), 0), (3, False, (64, 64), 0), (2, True, (8, 8), 1)])
def test_exr_tiled_decode(wfpt, tmp_path, compression, half, tile, level_mode):
    rng = np.random.default_rng(compression + 10 * half + tile[0])
    h, w = 37, 29   # (ragged right and bottom tiles)
    base = rng.random((h, w)).astype(np.float32)
    chans = {"R": base * 3, "G": np.round(base * 4) / 4, "B": base ** 2, "A": (base > 0.3).astype(np.float32)}
    dt = np.float16 if half else np.float32
    path = str(tmp_path / "t.exr")
    _write_tiled_exr(path, chans, compression, half, tile, level_mode)
    px, fmt = wfpt.read_image(path)
    assert fmt == (1 if half else 2) and px.shape == (h, w, 4)
    for i, n in enumerate("RGBA"):
        assert (px[..., i] == chans[n].astype(dt).astype(np.float32)).all(), n
    # malformed: a tile header that names a tile twice / truncated file
    data = open(path, "rb").read()
    for bad in ((data[: len(data) - 40],) if level_mode == 0 else ()) + (data[: len(data) // 2],):   # (the coarser MIP level at the end is never read)
        open(path, "wb").write(bad)
        with pytest.raises(wfpt.WfError):
            wfpt.read_image(path)


def _write_png(path, px, ctype, depth, interlace, plte=None):
    """PNG writer for the tests: filter 0 or 2 (alternating) rows, optional Adam7 interlacing.  px: [h][w][samples] integer array."""
    h, w, nc = px.shape
    def pack_rows(a):
        rows = b""
        for y in range(a.shape[0]):
            if depth == 16:
                line = a[y].astype(">u2").tobytes()
            elif depth == 8:
                line = a[y].astype(np.uint8).tobytes()
            else:
                bits = ((a[y].reshape(-1)[:, None] >> np.arange(depth - 1, -1, -1)) & 1).astype(np.uint8).reshape(-1)
                line = np.packbits(bits).tobytes()
            if y % 2 and y > 0:   # filter type 2 (Up) on odd rows
                prev = rows[-len(line):]
                line = bytes((line[i] - prev_u[i]) & 255 for i in range(len(line)))
                rows += b"\2" + line
                prev_u = bytes((line[i] + prev_u[i]) & 255 for i in range(len(line)))
            else:
                rows += b"\0" + line
                prev_u = line
        return rows
    if not interlace:
        raw = pack_rows(px)
    else:
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = px[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += pack_rows(sub)
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        out += chunk(b"PLTE", plte.astype(np.uint8).tobytes())
    out += chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    open(path, "wb").write(out)


@pytest.mark.parametrize("ctype,depth,shape", [(2, 8, (21, 13)), (6, 8, (9, 33)), (0, 16, (17, 17)), (0, 2, (11, 19)), (3, 4, (8, 8)), (2, 8, (1, 1)), (0, 8, (3, 2))])
def test_png_interlaced_equals_non_interlaced(wfpt, tmp_path, ctype, depth, shape):
    """Adam7 files decode to what the same pixels give without interlacing (all seven passes, sub-byte depths, images smaller than the 8 x 8
    pattern), and the writer used here is checked against the independent Python decoder above."""
    rng = np.random.default_rng(ctype * 10 + depth)
    nc = {0: 1, 2: 3, 3: 1, 6: 4}[ctype]
    h, w = shape
    px = rng.integers(0, 1 << min(depth, 4 if ctype == 3 else 16), size=(h, w, nc)).astype(np.uint32)
    plte = rng.integers(0, 256, size=(16, 3)) if ctype == 3 else None
    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    _write_png(a, px, ctype, depth, False, plte)
    _write_png(b, px, ctype, depth, True, plte)
    back, ct, dp, _ = _png_pixels(a)
    assert ct == ctype and dp == depth and (back == px).all()
    ia, fa = wfpt.read_image(a, "linear")
    ib, fb = wfpt.read_image(b, "linear")
    assert fa == fb and ia.shape == ib.shape and (ia == ib).all()
