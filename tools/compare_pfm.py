#!/usr/bin/env python3
"""Compare two PFM images: prints max abs / max rel error, fraction of bit-exact values, and how many
pixel values exceed the relative tolerance.  usage: compare_pfm.py a.pfm b.pfm [rel_tol] [floor]"""
import sys
import numpy as np


def read_pfm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        assert magic in (b"PF", b"Pf"), magic
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        nc = 3 if magic == b"PF" else 1
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, nc)
    return data[::-1].astype(np.float32)


def compare(a, b, floor=1e-3):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    rel = d / np.maximum(np.abs(b), floor)
    return {
        "max_abs": float(d.max()),
        "max_rel": float(rel.max()),
        "mean_rel": float(rel.mean()),
        "frac_exact": float((a == b).mean()),
        "mean_a": float(a.mean()),
        "mean_b": float(b.mean()),
    }


if __name__ == "__main__":
    a, b = read_pfm(sys.argv[1]), read_pfm(sys.argv[2])
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
    floor = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
    r = compare(a, b, floor)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    rel = d / np.maximum(np.abs(b), floor)
    r["frac_over_tol"] = float((rel > tol).mean())
    print(r)
