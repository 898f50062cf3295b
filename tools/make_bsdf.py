#!/usr/bin/env python3
"""tools/make_bsdf.py — writes small measured-BRDF tensor files (.bsdf) for the MeasuredMaterial tests.

The container is the reference's "tensor_file" format (bxdfs.cpp:730-812: 12-byte magic, version 1.0, field count, then per field a
name, rank, dtype, byte offset and shape); the fields and their shapes are the ones MeasuredBxDFData::Create checks (bxdfs.cpp:886-924):
theta_i [T], phi_i [P], ndf [a, b], sigma [a, b], vndf [P, T, r, r], luminance [P, T, r, r], spectra [P, T, W, r, r], wavelengths [W],
description (uint8 string), jacobian (uint8 [1]).

The tables are analytic stand-ins for a measurement — a GGX-like normal distribution, its projected area, visible-normal densities
that lean toward the mirror direction of each incident angle, and reflectance spectra tinted per file — smooth and strictly positive,
so renders are finite; they are NOT a fit to any real material.  P = 1 makes the isotropic layout (phi_i = [0]), P >= 3 the anisotropic
one (phi_i spans -pi .. pi, the only reduction the reader accepts)."""
import struct
import sys

import numpy as np

DT = {np.dtype("uint8"): 1, np.dtype("float32"): 10}


def write_tensor(path, fields):
    """fields: [(name, ndarray)] — uint8 or float32"""
    header = 12 + 2 + 4
    for name, a in fields:
        header += 2 + len(name) + 2 + 1 + 8 + 8 * a.ndim
    blobs, off = [], header
    for name, a in fields:
        off = (off + 7) // 8 * 8
        blobs.append(off)
        off += a.nbytes
    with open(path, "wb") as f:
        f.write(b"tensor_file\0" + bytes([1, 0]) + struct.pack("<I", len(fields)))
        for (name, a), o in zip(fields, blobs):
            f.write(struct.pack("<H", len(name)) + name.encode() + struct.pack("<HBQ", a.ndim, DT[a.dtype], o))
            f.write(struct.pack("<%dQ" % a.ndim, *a.shape))
        for (name, a), o in zip(fields, blobs):
            f.write(b"\0" * (o - f.tell()))
            f.write(np.ascontiguousarray(a).tobytes())


def synth(path, n_phi=1, n_theta=6, res=8, ndf_res=(4, 16), n_wl=10, alpha=0.35, tint=(0.8, 0.5, 0.3), seed=3):
    rng = np.random.RandomState(seed)
    theta_i = (np.linspace(0, 1, n_theta) ** 2 * (np.pi / 2)).astype(np.float32)          # u2theta of a uniform grid
    phi_i = np.array([0.0], np.float32) if n_phi == 1 else np.linspace(-np.pi, np.pi, n_phi).astype(np.float32)
    wavelengths = np.linspace(360, 830, n_wl).astype(np.float32)

    def D(theta_m, a=alpha):
        c = np.cos(theta_m)
        return a * a / (np.pi * ((a * a - 1) * c * c + 1) ** 2)

    # ndf / sigma over the unit square: x = theta2u (sqrt(2 theta / pi)), y = phi2u
    ny, nx = ndf_res
    ux = np.linspace(0, 1, nx)[None, :] * np.ones((ny, 1))
    uy = np.linspace(0, 1, ny)[:, None] * np.ones((1, nx))
    th = ux ** 2 * (np.pi / 2)
    aniso = 1 + (0.3 * np.cos(2 * (2 * uy - 1) * np.pi) if n_phi > 1 else 0)
    ndf = (D(th) * aniso + 0.02).astype(np.float32)
    sigma = (0.25 + 0.75 * np.cos(th) * (1 + 0.1 * (aniso - 1))).astype(np.float32)

    g = np.linspace(0, 1, res)
    gx, gy = g[None, :] * np.ones((res, 1)), g[:, None] * np.ones((1, res))
    vndf = np.zeros((n_phi, n_theta, res, res), np.float32)
    lum = np.zeros((n_phi, n_theta, res, res), np.float32)
    spectra = np.zeros((n_phi, n_theta, n_wl, res, res), np.float32)
    base = np.interp(wavelengths, [360, 450, 550, 650, 830], [tint[2] * 0.6, tint[2], tint[1], tint[0], tint[0] * 0.9])
    for p in range(n_phi):
        for t in range(n_theta):
            thm = gx ** 2 * (np.pi / 2)
            # visible normals: the distribution itself times a lobe toward half the incident angle (in the theta2u coordinate)
            centre = np.sqrt(0.5 * theta_i[t] * 2 / np.pi)
            lobe = np.exp(-((gx - centre) / 0.45) ** 2) * (1 + 0.25 * np.cos((2 * gy - 1) * np.pi - (phi_i[p] if n_phi > 1 else 0)))
            vndf[p, t] = 0.15 + D(thm) * lobe * (0.2 + np.sin(thm))
            lum[p, t] = 0.6 + 0.4 * np.cos(np.pi * (gx - 0.3 * t / n_theta)) * np.cos(0.5 * np.pi * (gy - 0.5)) + 0.05 * rng.rand(res, res)
            for w in range(n_wl):
                spectra[p, t, w] = base[w] * lum[p, t] * (0.9 + 0.1 * np.cos(theta_i[t])) + 0.01 * rng.rand(res, res)
    fields = [("description", np.frombuffer(b"synthetic stand-in (tools/make_bsdf.py)", np.uint8).copy()),
              ("jacobian", np.array([1], np.uint8)),
              ("phi_i", phi_i), ("theta_i", theta_i), ("wavelengths", wavelengths),
              ("ndf", ndf), ("sigma", sigma), ("vndf", vndf), ("luminance", lum), ("spectra", spectra.astype(np.float32))]
    write_tensor(path, fields)


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "tests/golden"
    synth(out + "/measured_iso.bsdf", n_phi=1, tint=(0.8, 0.5, 0.3), seed=3)
    synth(out + "/measured_aniso.bsdf", n_phi=5, n_theta=5, res=6, alpha=0.25, tint=(0.3, 0.55, 0.8), seed=5)
