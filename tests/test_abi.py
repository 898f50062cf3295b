"""The C-ABI libraries load and export every entry point the headers declare (no GPU, no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(wfh?_[a-z0-9_]+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_headers_declare_expected_symbols(wfpt):
    abi = declared("wf_abi.h")
    host = declared("wf_host.h")
    assert len(abi) >= 30 and len(host) >= 10
    assert sorted(wfpt.ABI_SYMBOLS) == abi
    assert sorted(wfpt.HOST_SYMBOLS) == host


def test_libraries_export_every_declared_symbol(wfpt):
    host, hip = wfpt.libs()
    for name in declared("wf_abi.h"):
        assert hasattr(hip, name), name
    for name in declared("wf_host.h"):
        assert hasattr(host, name), name
    hip.wf_abi_version.restype = ctypes.c_int
    assert hip.wf_abi_version() == 12   # include/wf_abi.h WF_ABI_VERSION (12: wf_instance.anim_plus1, wf_scene_desc.animated)


def test_no_cpu_fallback_in_product():
    """libwfhip/libwfhost must not link or reference the oracle."""
    for lib in ("libwfhip.so", "libwfhost.so", "pbrt_amd"):
        data = open(os.path.join(ROOT, "pbrt-v4_amd", "_build", lib), "rb").read()
        assert b"wf_cpu" not in data and b"oracle/" not in data and b"pbrt_ref" not in data


def test_ctx_create_fails_loudly_without_gpu(wfpt):
    import torch
    if torch.cuda.is_available():
        return
    host, hip = wfpt.libs()
    ctx = ctypes.c_void_p()
    rc = hip.wf_ctx_create(0, ctypes.byref(ctx))
    assert rc != 0
    assert hip.wf_last_error()
