#!/usr/bin/env python3
"""tests/golden/png_*.png: small PNG textures for the png_textures golden scene, written with zlib only (no image
library in this image).  Every colour type the reader handles appears once, rows cycle through the five PNG filter types
so the un-filtering code is exercised, and one file has a non-power-of-two resolution (FloatResizeUp path).
Deterministic: running it again rewrites identical files."""
import os, struct, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")

def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)

def write_png(path, rows, w, h, ctype, depth, bpp, plte=None, cycle=True, idat_split=1):
    """rows: list of bytes objects (one unfiltered scanline each)."""
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = (y % 5) if cycle else 0
        out = bytearray(len(row))
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
            out[i] = (x - pred) & 255
        raw.append(ft)
        raw += out
        prev = row
    z = zlib.compress(bytes(raw), 9)
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if plte is not None:
        png += chunk(b"PLTE", bytes(plte))
    n = len(z)
    for k in range(idat_split):
        png += chunk(b"IDAT", z[k * n // idat_split:(k + 1) * n // idat_split])
    png += chunk(b"IEND", b"")
    open(path, "wb").write(png)
    print(path, len(png))

rng = np.random.default_rng(20250923)

def pattern(w, h):
    y, x = np.mgrid[0:h, 0:w]
    r = 128 + 100 * np.sin(x * 0.45) * np.cos(y * 0.3)
    g = 40 + 190 * ((x // 6 + y // 5) % 2)
    b = 255 * x / max(1, w - 1) * (0.4 + 0.6 * y / max(1, h - 1))
    img = np.stack([r, g, b], -1) + rng.integers(-12, 13, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)

# 8-bit RGB, 48 x 40 (not a power of two), two IDAT chunks
img = pattern(48, 40)
write_png(os.path.join(G, "png_rgb8.png"), [img[y].tobytes() for y in range(40)], 48, 40, 2, 8, 3, idat_split=2)

# 8-bit RGBA 64 x 64: a leaf-like cut-out with a soft edge
y, x = np.mgrid[0:64, 0:64]
d = np.hypot((x - 31.5) / 30.0, (y - 31.5) / 18.0)
alpha = np.clip((1.05 - d) * 6, 0, 1)
alpha[(np.abs(x - 31.5) < 1.5)] *= 0.35
rgb = pattern(64, 64)
rgb[..., 1] = np.clip(rgb[..., 1] * 0.5 + 110, 0, 255)
rgba = np.concatenate([rgb, (alpha * 255 + 0.5).astype(np.uint8)[..., None]], -1).astype(np.uint8)
write_png(os.path.join(G, "png_rgba8.png"), [rgba[r].tobytes() for r in range(64)], 64, 64, 6, 8, 4)

# 8-bit RGBA whose alpha is 255 everywhere (MIPMap::CreateFromFile drops the channel)
opaque = np.concatenate([pattern(16, 16), np.full((16, 16, 1), 255, np.uint8)], -1)
write_png(os.path.join(G, "png_rgba8_opaque.png"), [opaque[r].tobytes() for r in range(16)], 16, 16, 6, 8, 4)

# 8-bit grey 32 x 32
grey = np.clip(30 + 200 * (0.5 + 0.5 * np.sin(np.mgrid[0:32, 0:32][1] * 0.7 + np.mgrid[0:32, 0:32][0] * 0.2)), 0, 255).astype(np.uint8)
write_png(os.path.join(G, "png_grey8.png"), [grey[r].tobytes() for r in range(32)], 32, 32, 0, 8, 1)

# 8-bit grey + alpha 16 x 16 (the reader keeps Y only)
ga = np.stack([grey[:16, :16], 255 - grey[:16, :16]], -1)
write_png(os.path.join(G, "png_greya8.png"), [ga[r].tobytes() for r in range(16)], 16, 16, 4, 8, 2)

# 16-bit RGB 16 x 16 (big-endian samples)
v16 = (pattern(16, 16).astype(np.uint32) * 257 + rng.integers(0, 200, (16, 16, 3))).clip(0, 65535).astype(">u2")
write_png(os.path.join(G, "png_rgb16.png"), [v16[r].tobytes() for r in range(16)], 16, 16, 2, 16, 6)

# 16-bit grey 8 x 8
g16 = (np.mgrid[0:8, 0:8][0] * 8 + np.mgrid[0:8, 0:8][1]).astype(np.uint32) * 1000 + 500
write_png(os.path.join(G, "png_grey16.png"), [g16.astype(">u2")[r].tobytes() for r in range(8)], 8, 8, 0, 16, 2)

# 8-bit palette 32 x 16 and a 4-bit palette 16 x 8
pal = [(int(40 + 13 * i) % 256, int(200 - 11 * i) % 256, int(17 * i) % 256) for i in range(16)]
idx = ((np.mgrid[0:16, 0:32][1] // 3 + np.mgrid[0:16, 0:32][0] // 2) % 16).astype(np.uint8)
write_png(os.path.join(G, "png_pal8.png"), [idx[r].tobytes() for r in range(16)], 32, 16, 3, 8, 1, plte=[c for p in pal for c in p])
idx4 = ((np.mgrid[0:8, 0:16][1] + np.mgrid[0:8, 0:16][0] * 3) % 16).astype(np.uint8)
rows4 = [bytes((int(idx4[r, 2 * k]) << 4) | int(idx4[r, 2 * k + 1]) for k in range(8)) for r in range(8)]
write_png(os.path.join(G, "png_pal4.png"), rows4, 16, 8, 3, 4, 1, plte=[c for p in pal for c in p])

# 2-bit grey 16 x 16
g2 = ((np.mgrid[0:16, 0:16][1] // 2 + np.mgrid[0:16, 0:16][0] // 3) % 4).astype(np.uint8)
rows2 = [bytes((int(g2[r, 4 * k]) << 6) | (int(g2[r, 4 * k + 1]) << 4) | (int(g2[r, 4 * k + 2]) << 2) | int(g2[r, 4 * k + 3]) for k in range(4)) for r in range(16)]
write_png(os.path.join(G, "png_grey2.png"), rows2, 16, 16, 0, 2, 1)

# 8-bit RGB normal map 64 x 64: bumps, encoded (n * 0.5 + 0.5) * 255
yy, xx = np.mgrid[0:64, 0:64]
hx = 0.6 * np.cos(xx * np.pi / 8) * np.sin(yy * np.pi / 8)
hy = 0.6 * np.sin(xx * np.pi / 8) * np.cos(yy * np.pi / 8)
n = np.stack([-hx, -hy, np.ones_like(hx)], -1)
n /= np.linalg.norm(n, axis=-1, keepdims=True)
nm = np.clip((n * 0.5 + 0.5) * 255 + 0.5, 0, 255).astype(np.uint8)
write_png(os.path.join(G, "png_normal.png"), [nm[r].tobytes() for r in range(64)], 64, 64, 2, 8, 3)
