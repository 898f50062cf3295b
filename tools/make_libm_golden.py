#!/usr/bin/env python3
"""tests/golden/libm_<fn>_{in,out}.bin: known-answer vectors of this image's glibc 2.35 float routines.

Inputs: 16 384 float32 per function (pairs for atan2) = the range-reduction / branch edges of each routine, the
special values, and seeded draws over the domain the path uses.  Outputs: the LIVE libm, through
oracle/_build/libm_check eval.  The device restatement (csrc/common/wf_libm.h) must reproduce them bit for bit
(tests/test_gpu_parity.py::test_device_libm_golden); oracle/_build/libm_check exhaustive is the 2^32-argument proof.
"""
import os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
CHECK = os.path.join(ROOT, "oracle", "_build", "libm_check")
N = 16384

def f32(bits): return np.array(bits, dtype=np.uint32).view(np.float32)

def around(bits, k=4):
    out = []
    for b in bits:
        out += [b + d for d in range(-k, k + 1)] + [(b + d) | 0x80000000 for d in range(-k, k + 1)]
    return f32([x & 0xffffffff for x in out])

SPECIAL = f32([0, 0x80000000, 1, 0x80000001, 0x007fffff, 0x00800000, 0x7f7fffff, 0xff7fffff, 0x7f800000, 0xff800000, 0x7fc00000, 0x3f800000, 0xbf800000])
EDGES = {
    "sin": [0x39800000, 0x3f490fdb, 0x42f00000, 0x3fc90fdb, 0x40490fdb, 0x4b000000, 0x5f000000, 0x7f000000],
    "cos": [0x39800000, 0x3f490fdb, 0x42f00000, 0x3fc90fdb, 0x40490fdb, 0x4b000000, 0x5f000000, 0x7f000000],
    "exp": [0x42b00000, 0x42b17218, 0xc2cff1b4, 0xc2ce8ecf, 0x3f317218, 0x33000000],
    "log": [0x3f800000, 0x3f330000, 0x3f7f0000, 0x00800000, 0x00000100, 0x7f7fffff],
    "atan": [0x4c000000, 0x3ee00000, 0x31000000, 0x3f980000, 0x3f300000, 0x401c0000],
    "asin": [0x3f800000, 0x3f000000, 0x32000000, 0x3f79999a],
    "acos": [0x3f800000, 0x3f000000, 0x32800000],
    "cosh": [0x41b00000, 0x3eb17218, 0x24000000, 0x42b1717f, 0x42b2d4fc, 0x33000000],
    "atanh": [0x3f000000, 0x31800000, 0x3f800000, 0x3ed413d7, 0x3e95f61f],
    "tan": [0x3f490fda, 0x3f490fdb, 0x3fc90fdb, 0x3fc90fda, 0x39000000, 0x3f2ca140, 0x42f00000, 0x4016cbe4, 0x5f000000, 0x7f000000, 0x40490fdb],
    "sinh": [0x41b00000, 0x31800000, 0x3f800000, 0x42b1717f, 0x42b2d4fc, 0x3eb17218, 0x3f851592, 0x4195b844, 0x33000000],
}
RANGE = {"sin": (-8, 8), "cos": (-8, 8), "exp": (-90, 90), "log": (0, 100), "atan": (-20, 20), "asin": (-1, 1), "acos": (-1, 1),
         "cosh": (-3, 3), "atanh": (-1, 1), "sinh": (-12, 12), "tan": (-1.6, 1.6)}

def inputs(fn, rng):
    if fn == "atan2":
        y = rng.standard_normal(N).astype(np.float32)
        x = rng.standard_normal(N).astype(np.float32)
        sp = np.concatenate([SPECIAL, f32([0x3f800000, 0x5f000000, 0x1f000000])])
        k = 0
        for a in sp:
            for b in sp:
                y[k], x[k] = a, b
                k += 1
        y[k:k + 2000] *= np.float32(2.0) ** rng.integers(-70, 70, 2000).astype(np.float32)
        return np.stack([y, x], axis=1).astype(np.float32)
    lo, hi = RANGE[fn]
    v = rng.uniform(lo, hi, N).astype(np.float32)
    e = np.concatenate([SPECIAL, around(EDGES[fn])])
    v[:e.size] = e
    # every exponent once more, random mantissas, both signs
    bits = (rng.integers(0, 1 << 32, 2048, dtype=np.uint64)).astype(np.uint32)
    v[e.size:e.size + 2048] = bits.view(np.float32)
    return v

def main():
    rng = np.random.default_rng(2035)
    for fn in ("sin", "cos", "exp", "log", "atan", "asin", "acos", "cosh", "atanh", "atan2", "sinh", "tan"):
        x = inputs(fn, rng)
        out = subprocess.run([CHECK, "eval", fn], input=x.tobytes(), capture_output=True, check=True).stdout
        y = np.frombuffer(out, dtype=np.float32)
        assert y.size == N
        x.tofile(os.path.join(G, "libm_%s_in.bin" % fn))
        y.tofile(os.path.join(G, "libm_%s_out.bin" % fn))
        print(fn, "finite", np.isfinite(y).mean())

if __name__ == "__main__":
    main()
