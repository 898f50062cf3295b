"""The device's elementary functions (csrc/common/wf_libm.h) restate the glibc 2.35 float routines the reference is
linked against.  CPU side of the proof: the restatement compiled for the host reproduces the committed known-answer
vectors (tests/golden/libm_*.bin, made from the live libm by tools/make_libm_golden.py) bit for bit, the live libm of the
machine running the tests still produces those vectors (i.e. the goldens describe *this* libm), and a 2^24-argument
strided sweep per function finds no difference (`libm_check exhaustive` is the full 2^32 sweep: 90 s on 8 cores)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CHECK = os.path.join(ROOT, "oracle", "_build", "libm_check")
FNS = ("sin", "cos", "exp", "log", "atan", "asin", "acos", "cosh", "atanh", "atan2", "sinh", "tan")


def _eval(mode, fn, x):
    out = subprocess.run([CHECK, mode, fn], input=x.tobytes(), capture_output=True, check=True).stdout
    return np.frombuffer(out, dtype=np.float32)


def same_bits(a, b):
    nan = np.isnan(a) & np.isnan(b)
    return nan | (a.view(np.uint32) == b.view(np.uint32))


@pytest.mark.parametrize("fn", FNS)
def test_restatement_and_live_libm_vs_golden(built, fn):
    x = np.fromfile(os.path.join(GOLDEN, "libm_%s_in.bin" % fn), dtype=np.float32)
    y = np.fromfile(os.path.join(GOLDEN, "libm_%s_out.bin" % fn), dtype=np.float32)
    assert same_bits(_eval("evalmine", fn, x), y).all()
    assert same_bits(_eval("eval", fn, x), y).all()


def test_strided_sweep(built):
    r = subprocess.run([CHECK, "quick"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
