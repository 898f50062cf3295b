"""wfpt — ctypes binding of the MI355X wavefront path tracer's two shared libraries:

  _build/libwfhost.so  (include/wf_host.h)  scene parser, flat-table builder, render loop
  _build/libwfhip.so   (include/wf_abi.h)   HIP/CDNA4 kernels behind the C ABI

Python is plumbing only (tests, bench.py, torch.distributed for the multi-GPU film reduce).  There is no
CPU fallback: if the libraries are missing, or no gfx950 device is visible when a renderer is created,
the calls raise.

The package directory is called ``pbrt-v4_amd`` (not an importable identifier); load this module with
``importlib`` — see ``load()`` in tests/conftest.py, bench.py and __graft_entry__.py.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.environ.get("WF_BUILD_DIR") or os.path.join(HERE, "_build")  # WF_BUILD_DIR: timing / validation of variant builds (pbrt-v4_amd/_exp*)
DATA = os.path.join(HERE, "data")


class WfError(RuntimeError):
    pass


class Info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "spp", "max_queue_size", "n_passes", "scanlines_per_pass",
        "n_triangles", "n_bvh_nodes", "n_lights", "max_depth", "save_fp16", "y0")]


class RenderStats(C.Structure):
    _fields_ = [("camera_rays", C.c_uint64), ("indirect_rays", C.c_uint64 * 64), ("shadow_rays", C.c_uint64 * 64)]


class HitRecord(C.Structure):
    _fields_ = [("prim", C.c_int32), ("t", C.c_float), ("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float),
                ("nodes_visited", C.c_int32), ("tris_tested", C.c_int32), ("instance", C.c_int32)]


class TraversalCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "closest_rays", "closest_nodes", "closest_tris", "closest_hits",
        "shadow_rays", "shadow_nodes", "shadow_tris", "shadow_unoccluded")]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int32), ("total_ms", C.c_float), ("min_ms", C.c_float), ("max_ms", C.c_float)]


# every symbol include/wf_abi.h and include/wf_host.h declare (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "wf_last_error", "wf_abi_version", "wf_ctx_create", "wf_ctx_destroy", "wf_sync", "wf_stream", "wf_scene_upload",
    "wf_medium_sample", "wf_intersect_shadow_tr", "wf_subsurface_probe", "wf_intersect_one_random", "wf_subsurface_scatter", "wf_trace_one_random_host", "wf_morton_sort", "wf_build_bvh_sah", "wf_aggregate_bounds", "wf_queues_alloc", "wf_set_pass_samples", "wf_set_strips", "wf_film_clear", "wf_reset_ray_queue", "wf_reset_stage_queues",
    "wf_gen_camera_rays", "wf_gen_ray_samples", "wf_intersect_closest", "wf_handle_escaped", "wf_handle_emissive",
    "wf_eval_material", "wf_intersect_shadow", "wf_update_film", "wf_render_pass", "wf_film_download",
    "wf_film_device_ptr", "wf_film_upload", "wf_film_spectral_download", "wf_film_gbuffer_download", "wf_film_copy_to_device", "wf_film_copy_from_device", "wf_film_gather_strips", "wf_stats_add", "wf_material_items_download", "wf_stats_download",
    "wf_profile_report", "wf_profile_enable",
    "wf_trace_closest_host", "wf_trace_any_host", "wf_sampler_probe", "wf_libm_probe", "wf_kat_probe", "wf_queue_size", "wf_queue_download",
    "wf_counters_enable", "wf_counters_download", "wf_kernel_time_ms", "wf_debug_counters", "wf_debug_fastbvh_check", "wf_trace_closest_host_t", "wf_trace_any_host_t", "wf_ctx_query",
    "wf_trace_closest_device", "wf_trace_any_device", "wf_device_alloc", "wf_device_free", "wf_device_upload", "wf_device_download", "wf_trace_shadow_tr_host",
]
HOST_SYMBOLS = [
    "wfh_init", "wfh_last_error", "wfh_scene_load", "wfh_scene_load_string", "wfh_scene_free", "wfh_scene_desc", "wfh_scene_info",
    "wfh_renderer_create", "wfh_renderer_create_strips", "wfh_renderer_set_strips", "wfh_renderer_samples_per_pass", "wfh_renderer_ctx", "wfh_render", "wfh_clear_film", "wfh_download_film", "wfh_stats",
    "wfh_film_to_rgb", "wfh_write_image", "wfh_read_image", "wfh_read_nanovdb", "wfh_build_bvh_host", "wfh_film_channels", "wfh_write_film_image",
]

_hip = None
_host = None


def libs():
    """Load (once) and return (libwfhost, libwfhip).  Raises if the in-tree build is missing."""
    global _hip, _host
    if _host is not None:
        return _host, _hip
    hip_path = os.path.join(BUILD, "libwfhip.so")
    host_path = os.path.join(BUILD, "libwfhost.so")
    for p in (hip_path, host_path):
        if not os.path.exists(p):
            raise WfError("%s is missing: run __graft_entry__.build() (make -C pbrt-v4_amd)" % p)
    _hip = C.CDLL(hip_path, mode=C.RTLD_GLOBAL)
    _host = C.CDLL(host_path, mode=C.RTLD_GLOBAL)
    _hip.wf_last_error.restype = C.c_char_p
    _hip.wf_stream.restype = C.c_void_p
    _hip.wf_stream.argtypes = [C.c_void_p]
    _host.wfh_last_error.restype = C.c_char_p
    _host.wfh_scene_load.restype = C.c_void_p
    _host.wfh_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
    _host.wfh_scene_load_string.restype = C.c_void_p
    _host.wfh_scene_load_string.argtypes = [C.c_char_p, C.c_int, C.c_int]
    _host.wfh_scene_free.argtypes = [C.c_void_p]
    _host.wfh_scene_desc.restype = C.c_void_p
    _host.wfh_scene_desc.argtypes = [C.c_void_p]
    _host.wfh_scene_info.argtypes = [C.c_void_p, C.POINTER(Info)]
    _host.wfh_renderer_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
    _host.wfh_renderer_create_strips.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    _host.wfh_renderer_samples_per_pass.argtypes = [C.c_void_p]
    _host.wfh_renderer_set_strips.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    _host.wfh_renderer_ctx.restype = C.c_void_p
    _host.wfh_renderer_ctx.argtypes = [C.c_void_p]
    _host.wfh_render.restype = C.c_double
    _host.wfh_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    _host.wfh_clear_film.argtypes = [C.c_void_p]
    _host.wfh_download_film.argtypes = [C.c_void_p, C.c_void_p]
    _host.wfh_stats.argtypes = [C.c_void_p, C.POINTER(RenderStats)]
    _host.wfh_film_to_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    _host.wfh_write_image.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int]
    _host.wfh_read_image.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]
    for name in ("wf_sync", "wf_film_clear", "wf_ctx_destroy"):
        getattr(_hip, name).argtypes = [C.c_void_p]
    _hip.wf_profile_enable.argtypes = [C.c_void_p, C.c_int]
    _hip.wf_counters_enable.argtypes = [C.c_void_p, C.c_int]
    _hip.wf_counters_download.argtypes = [C.c_void_p, C.POINTER(TraversalCounters)]
    _hip.wf_profile_report.argtypes = [C.c_void_p, C.POINTER(ProfileEntry), C.c_int, C.POINTER(C.c_int)]
    _hip.wf_film_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    _hip.wf_film_copy_to_device.argtypes = [C.c_void_p, C.c_void_p]
    _hip.wf_film_copy_from_device.argtypes = [C.c_void_p, C.c_void_p]
    _hip.wf_trace_closest_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    _hip.wf_trace_any_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _hip.wf_sampler_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    _hip.wf_libm_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    _hip.wf_kat_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _hip.wf_material_items_download.argtypes = [C.c_void_p, C.c_void_p]
    _hip.wf_kernel_time_ms.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    _hip.wf_aggregate_bounds.argtypes = [C.c_void_p, C.c_void_p]
    _hip.wf_render_pass.argtypes = [C.c_void_p, C.c_int, C.c_int]
    if _host.wfh_init(DATA.encode()) != 0:
        raise WfError("wfh_init failed (data dir %s)" % DATA)
    return _host, _hip


def _check(rc, what):
    if rc != 0:
        _, hip = libs()
        raise WfError("%s failed: %s" % (what, hip.wf_last_error().decode(errors="replace")))


class Scene:
    """A parsed .pbrt scene flattened into the ABI's tables (BasicScene + CreateAggregate equivalent)."""

    def __init__(self, path=None, text=None, spp=0, seed=0):
        host, _ = libs()
        if path is not None:
            self.h = host.wfh_scene_load(os.fsencode(path), spp, seed)
        else:
            self.h = host.wfh_scene_load_string(text.encode(), spp, seed)
        if not self.h:
            host.wfh_last_error.restype = C.c_char_p
            raise WfError("scene load failed: " + (host.wfh_last_error() or b"").decode(errors="replace"))
        self.info = Info()
        host.wfh_scene_info(self.h, C.byref(self.info))
        self._renderer = False

    @property
    def width(self):
        return self.info.width

    @property
    def height(self):
        return self.info.height

    @property
    def spp(self):
        return self.info.spp

    def create_renderer(self, device=0, samples_per_pass=0, strips=None):
        """WavefrontPathIntegrator ctor: upload tables to HIP device `device`, allocate queues.
        samples_per_pass: sample indices one pass carries (0 = automatic); the film is bit-identical for any value.
        strips = (rank, count[, height]): this renderer is one rank of a multi-GPU image partition from the start — its queues are
        sized for its own rows (wfh_renderer_create_strips)."""
        host, _ = libs()
        if strips is not None and strips[1] > 1:
            rc = host.wfh_renderer_create_strips(self.h, device, samples_per_pass, strips[0], strips[1], strips[2] if len(strips) > 2 else 16)
        else:
            rc = host.wfh_renderer_create(self.h, device, samples_per_pass)
        if rc != 0:
            raise WfError("renderer creation failed: " + (host.wfh_last_error() or b"").decode(errors="replace"))
        self._renderer = True
        self.samples_per_pass = host.wfh_renderer_samples_per_pass(self.h)
        self.ctx = host.wfh_renderer_ctx(self.h)
        return self

    def set_strips(self, rank, count, height=16):
        """multi-GPU image partition (wf_set_strips): this renderer owns the scanline strips rank, rank + count, ..."""
        host, _ = libs()
        if host.wfh_renderer_set_strips(self.h, rank, count, height) != 0:
            raise WfError("set_strips failed")

    def render(self, sample_begin=0, sample_end=None, sample_step=1, fused=True):
        """Render(): returns wall seconds."""
        host, _ = libs()
        if not self._renderer:
            raise WfError("create_renderer() first")
        if sample_end is None:
            sample_end = self.info.spp
        s = host.wfh_render(self.h, sample_begin, sample_end, sample_step, 1 if fused else 0)
        if s < 0:
            raise WfError("render failed")
        return s

    def clear_film(self):
        host, _ = libs()
        host.wfh_clear_film(self.h)

    def film(self):
        """[H, W, 4] float64: rgbSum[3], weightSum per pixel (film.h:302-307)."""
        host, _ = libs()
        a = np.empty((self.info.height, self.info.width, 4), dtype=np.float64)
        if host.wfh_download_film(self.h, a.ctypes.data) != 0:
            raise WfError("film download failed")
        return a

    def film_to_rgb(self, film):
        host, _ = libs()
        film = np.ascontiguousarray(film, dtype=np.float64)
        rgb = np.empty((self.info.height, self.info.width, 3), dtype=np.float32)
        host.wfh_film_to_rgb(self.h, film.ctypes.data, rgb.ctypes.data)
        return rgb

    def image(self):
        return self.film_to_rgb(self.film())

    def film_channels(self):
        """SpectralFilm / GBufferFilm::GetImage of a scene with `Film "spectral"` or `Film "gbuffer"`: (channel names, float32 array [H][W][n])"""
        host, _ = libs()
        host.wfh_film_channels.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_char_p, C.c_void_p]
        nc = C.c_int32(0)
        if host.wfh_film_channels(self.h, C.byref(nc), None, None) != 0:
            raise WfError(host.wfh_last_error().decode(errors="replace"))
        names = C.create_string_buffer(32 * nc.value)
        px = np.empty((self.info.height, self.info.width, nc.value), np.float32)
        if host.wfh_film_channels(self.h, C.byref(nc), names, px.ctypes.data) != 0:
            raise WfError(host.wfh_last_error().decode(errors="replace"))
        return [names.raw[32 * i:32 * (i + 1)].split(b"\0")[0].decode() for i in range(nc.value)], px

    def write_film_image(self, path):
        host, _ = libs()
        host.wfh_write_film_image.argtypes = [C.c_void_p, C.c_char_p]
        if host.wfh_write_film_image(self.h, os.fsencode(path)) != 0:
            raise WfError("could not write %s: %s" % (path, host.wfh_last_error().decode(errors="replace")))

    def stats(self):
        host, _ = libs()
        st = RenderStats()
        host.wfh_stats(self.h, C.byref(st))
        return {"camera_rays": int(st.camera_rays), "indirect_rays": [int(v) for v in st.indirect_rays],
                "shadow_rays": [int(v) for v in st.shadow_rays]}

    def total_rays(self):
        st = self.stats()
        return st["camera_rays"] + sum(st["indirect_rays"][1:]) + sum(st["shadow_rays"])

    # ---- direct C-ABI access used by the parity tests and bench.py ----
    def trace_closest(self, o, d, tmax, reference_order=True):
        """reference_order=True: the reference-order walk (fills nodes_visited / tris_tested);
        False: the production traversal kernel (wf_traverse.h)"""
        _, hip = libs()
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        n = o.shape[0]
        out = (HitRecord * n)()
        _check(hip.wf_trace_closest_host(self.ctx, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, out, 1 if reference_order else 0), "wf_trace_closest_host")
        return np.frombuffer(out, dtype=np.dtype([("prim", "<i4"), ("t", "<f4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4"),
                                                  ("nodes_visited", "<i4"), ("tris_tested", "<i4"), ("instance", "<i4")])).copy()

    def trace_timed(self, o, d, tmax, time, any_hit=False):
        """wf_trace_closest_host_t / wf_trace_any_host_t: the reference-order walks at the rays' own times (AnimatedPrimitive)"""
        _, hip = libs()
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        time = np.ascontiguousarray(time, dtype=np.float32)
        n = o.shape[0]
        for f in (hip.wf_trace_closest_host_t, hip.wf_trace_any_host_t):
            f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if any_hit:
            occ = np.empty(n, dtype=np.int32)
            _check(hip.wf_trace_any_host_t(self.ctx, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, time.ctypes.data, occ.ctypes.data), "wf_trace_any_host_t")
            return occ
        out = (HitRecord * n)()
        _check(hip.wf_trace_closest_host_t(self.ctx, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, time.ctypes.data, C.addressof(out)), "wf_trace_closest_host_t")
        return np.frombuffer(out, dtype=np.dtype([("prim", "<i4"), ("t", "<f4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4"),
                                                  ("nodes_visited", "<i4"), ("tris_tested", "<i4"), ("instance", "<i4")])).copy()

    def bounds(self):
        """WavefrontAggregate::Bounds() in rendering space: (pMin[3], pMax[3])"""
        _, hip = libs()
        b = (C.c_float * 6)()
        _check(hip.wf_aggregate_bounds(self.ctx, b), "wf_aggregate_bounds")
        return np.array(b[:3], dtype=np.float32), np.array(b[3:], dtype=np.float32)

    def trace_any(self, o, d, tmax, reference_order=True):
        _, hip = libs()
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        n = o.shape[0]
        occ = np.empty(n, dtype=np.int32)
        nodes = np.empty(n, dtype=np.int32)
        tris = np.empty(n, dtype=np.int32)
        _check(hip.wf_trace_any_host(self.ctx, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, occ.ctypes.data,
                                     nodes.ctypes.data if reference_order else None, tris.ctypes.data if reference_order else None),
               "wf_trace_any_host")
        return occ, nodes, tris

    def sampler_probe(self, px, py, sample_index, start_dim, ndims):
        _, hip = libs()
        px = np.ascontiguousarray(px, dtype=np.int32)
        py = np.ascontiguousarray(py, dtype=np.int32)
        si = np.ascontiguousarray(sample_index, dtype=np.int32)
        out = np.empty((px.shape[0], ndims if ndims > 0 else 30 if ndims == -3 else 2), dtype=np.float32)   # record sizes: include/wf_abi.h
        _check(hip.wf_sampler_probe(self.ctx, px.shape[0], px.ctypes.data, py.ctypes.data, si.ctypes.data, start_dim, ndims, out.ctypes.data),
               "wf_sampler_probe")
        return out

    LIBM_FNS = ("sin", "cos", "exp", "log", "atan", "asin", "acos", "cosh", "atanh", "atan2", "sinh", "tan")

    def libm_probe(self, fn, x):
        """Device evaluation of the kernels' elementary function `fn` over float32 array x ((n, 2) (y, x) pairs for atan2)."""
        _, hip = libs()
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        out = np.empty(n, dtype=np.float32)
        _check(hip.wf_libm_probe(self.ctx, self.LIBM_FNS.index(fn), n, x.ctypes.data, out.ctypes.data), "wf_libm_probe")
        return out

    def material_items(self):
        """items the material stage evaluated since the last clear_film(): {material type: count}, plus "medium_sample" """
        _, hip = libs()
        out = (C.c_uint64 * 16)()
        _check(hip.wf_material_items_download(self.ctx, out), "wf_material_items_download")
        d = {i: int(out[i]) for i in range(11)}
        d["medium_sample"] = int(out[11])
        return d

    def kat_probe(self, records):
        """Device evaluation of the known-answer probe (csrc/common/wf_kat.h): (n, 16) uint64 records in, (n, 8) uint64 out."""
        _, hip = libs()
        records = np.ascontiguousarray(records, dtype=np.uint64).reshape(-1, 16)
        out = np.empty((records.shape[0], 8), dtype=np.uint64)
        _check(hip.wf_kat_probe(self.ctx, records.shape[0], records.ctypes.data, out.ctypes.data), "wf_kat_probe")
        return out

    def query(self, key):
        """wf_ctx_query: which kernel variants the uploaded scene runs ("fast_ok", "gen_mode", "defer_general", "anim_fast", "lean_type_2", ...)"""
        _, hip = libs()
        v = C.c_int64(0)
        hip.wf_ctx_query.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        _check(hip.wf_ctx_query(self.ctx, key.encode(), C.byref(v)), "wf_ctx_query")
        return int(v.value)

    def fastbvh_check(self, n_rays=32, seed=1):
        """host-only self-check of the production traversal layout (wf_debug_fastbvh_check): needs no device.  Returns a dict."""
        host, hip = libs()
        out = (C.c_int64 * 8)()
        hip.wf_debug_fastbvh_check.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_int64)]
        rc = hip.wf_debug_fastbvh_check(host.wfh_scene_desc(self.h), n_rays, seed, out)
        if rc != 0:
            hip.wf_last_error.restype = C.c_char_p
            raise WfError("wf_debug_fastbvh_check: " + hip.wf_last_error().decode())
        keys = ["qnodes", "leaf_records", "entries", "nodes_visited", "true_hits", "missed", "tested", "entries_taken"]
        return dict(zip(keys, [int(v) for v in out]))

    def enable_profile(self, on=True):
        _, hip = libs()
        _check(hip.wf_profile_enable(self.ctx, int(on)), "wf_profile_enable")   # 0 off, 1 every launch, 2 the Intersect* launches only

    def profile_report(self):
        _, hip = libs()
        ent = (ProfileEntry * 64)()
        n = C.c_int(0)
        _check(hip.wf_profile_report(self.ctx, ent, 64, C.byref(n)), "wf_profile_report")
        return [{"name": ent[i].name.decode(), "launches": ent[i].launches, "total_ms": ent[i].total_ms,
                 "min_ms": ent[i].min_ms, "max_ms": ent[i].max_ms} for i in range(n.value)]

    def enable_counters(self, on=True):
        _, hip = libs()
        _check(hip.wf_counters_enable(self.ctx, 1 if on else 0), "wf_counters_enable")

    def counters(self):
        _, hip = libs()
        c = TraversalCounters()
        _check(hip.wf_counters_download(self.ctx, C.byref(c)), "wf_counters_download")
        return {n: int(getattr(c, n)) for n, _ in TraversalCounters._fields_}

    def debug_counters(self, reset=True):
        """rare-path counters of the production traversal (wf_debug_counters): spilled stack entries, overflow flag,
        inline near-tie re-traces, cursor path taken"""
        _, hip = libs()
        out = (C.c_uint64 * 4)()
        hip.wf_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        _check(hip.wf_debug_counters(self.ctx, out, 1 if reset else 0), "wf_debug_counters")
        return {"spilled_entries": int(out[0]), "overflow": int(out[1]), "inline_retraces": int(out[2]), "cursor": int(out[3])}

    def film_to_tensor(self, tensor):
        """device-to-device copy of the film accumulators into a torch CUDA float64 tensor (for the RCCL reduce)"""
        _, hip = libs()
        _check(hip.wf_film_copy_to_device(self.ctx, C.c_void_p(tensor.data_ptr())), "wf_film_copy_to_device")

    def film_from_tensor(self, tensor):
        _, hip = libs()
        _check(hip.wf_film_copy_from_device(self.ctx, C.c_void_p(tensor.data_ptr())), "wf_film_copy_from_device")

    def close(self):
        host, _ = libs()
        if self.h:
            host.wfh_scene_free(self.h)
            self.h = None


def write_pfm(path, rgb):
    host, _ = libs()
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    if host.wfh_write_image(os.fsencode(path), rgb.ctypes.data, rgb.shape[1], rgb.shape[0]) != 0:
        raise WfError("could not write %s" % path)


def read_image(path, encoding=None):
    """Image::Read through the host library (.pfm, .png, .exr): (float32 array [h][w][nc], storage format 0 8-bit / 1 half / 2 float)."""
    host, _ = libs()
    w, h, nc, fmt = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    enc = None if encoding is None else encoding.encode()
    if host.wfh_read_image(os.fsencode(path), enc, C.byref(w), C.byref(h), C.byref(nc), C.byref(fmt), None) != 0:
        raise WfError(host.wfh_last_error().decode())
    px = np.empty((h.value, w.value, nc.value), np.float32)
    if host.wfh_read_image(os.fsencode(path), enc, None, None, None, None, px.ctypes.data) != 0:
        raise WfError(host.wfh_last_error().decode())
    return px, fmt.value


def read_nanovdb(path, grid_name):
    """The float grid `grid_name` of a NanoVDB file through the host library's own reader (parity unpinned): None if the file has no
    such grid, else dict(min, dim, inv_mat, vec, background, values[z][y][x])."""
    host, _ = libs()
    mn, dm = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    inv, vec, bg = (C.c_float * 9)(), (C.c_float * 3)(), C.c_float()
    host.wfh_read_nanovdb.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = host.wfh_read_nanovdb(os.fsencode(path), grid_name.encode(), mn, dm, inv, vec, C.byref(bg), None)
    if rc == 1:
        return None
    if rc != 0:
        raise WfError(host.wfh_last_error().decode(errors="replace"))
    vals = np.empty((dm[2], dm[1], dm[0]), np.float32)
    if host.wfh_read_nanovdb(os.fsencode(path), grid_name.encode(), mn, dm, inv, vec, C.byref(bg), vals.ctypes.data) != 0:
        raise WfError(host.wfh_last_error().decode(errors="replace"))
    return {"min": list(mn), "dim": list(dm), "inv_mat": list(inv), "vec": list(vec), "background": bg.value, "values": vals}


def read_pfm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        if magic not in (b"PF", b"Pf"):
            raise WfError("%s: not a PFM file" % path)
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        nc = 3 if magic == b"PF" else 1
        data = np.frombuffer(f.read(), dtype="<f4" if scale < 0 else ">f4").reshape(h, w, nc)
    return data[::-1].astype(np.float32)
