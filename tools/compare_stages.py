#!/usr/bin/env python3
"""Compare stage dumps of oracle/_ref/ref_stages (the reference) and oracle/_build/wf_cpu --dump-stages.
usage: compare_stages.py ref_dir cpu_dir"""
import sys
import numpy as np


def load(d, name, w):
    a = np.fromfile("%s/%s.bin" % (d, name), dtype=np.float32).reshape(-1, w)
    return a[np.argsort(a[:, 1 if name == "mat_items" else 0], kind="stable")]


names = {"camera_rays": (8, ["pix", "ox", "oy", "oz", "dx", "dy", "dz", "time"]),
         "samples": (8, ["pix", "d.uc", "d.ux", "d.uy", "i.uc", "i.ux", "i.uy", "rr"]),
         "mat_items": (28, ["type", "pix", "pilo.x", "pilo.y", "pilo.z", "pihi.x", "pihi.y", "pihi.z", "n.x", "n.y", "n.z", "ns.x", "ns.y", "ns.z",
                            "dpdus.x", "dpdus.y", "dpdus.z", "wo.x", "wo.y", "wo.z", "u", "v", "dpdu.x", "dpdu.y", "dpdu.z", "dpdv.x", "dpdv.y", "dpdv.z"])}
for name, (w, cols) in names.items():
    a, b = load(sys.argv[1], name, w), load(sys.argv[2], name, w)
    print(name, a.shape, b.shape)
    if a.shape != b.shape:
        print("  SHAPE MISMATCH")
        continue
    same = a.view(np.uint32) == b.view(np.uint32)
    for c in range(w):
        if not same[:, c].all():
            bad = np.where(~same[:, c])[0]
            i = bad[0]
            print("  %-8s differs in %6d rows (%.4f); e.g. row %d pix %d: ref %.9g  cpu %.9g" % (cols[c], len(bad), len(bad) / len(a), i, int(a[i, 1 if name == "mat_items" else 0]), a[i, c], b[i, c]))
