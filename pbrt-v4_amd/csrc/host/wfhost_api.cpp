// wfhost_api.cpp — C entry points of libwfhost.so (include/wf_host.h): scene loading (parser + table
// builder) and the render loop, for callers that are not C++ (tests/, bench.py, __graft_entry__.py via
// ctypes).  The render itself goes through libwfhip.so's C ABI (include/wf_abi.h).
#include "integrator.h"
#include "../../../include/wf_host.h"

#include <dlfcn.h>
#include <cstring>
#include <sys/stat.h>
#include <memory>
#include <string>

using namespace wf;

struct wfh_scene {
    RenderOptions opt;
    ParsedScene parsed;
    SceneTables T;
    std::unique_ptr<WavefrontRenderer> renderer;
};

static bool g_init = false;

static thread_local std::string g_lastError;
// run f(); a thrown SceneError becomes the return value `bad` + wfh_last_error()
template <typename R, typename F>
static R Guard(R bad, F f) {
    try {
        return f();
    } catch (const std::exception &e) {
        g_lastError = e.what();
        fprintf(stderr, "%s\n", e.what());
        return bad;
    }
}

extern "C" {

int wfh_init(const char *data_dir) {
    if (g_init) return 0;
    std::string d = data_dir ? data_dir : "";
    if (d.empty()) return -1;
    SpectralData::Init(d, d + "/cache");
    SetMortonSort(&wf_morton_sort);  // the HLBVH build's Morton sort runs on the GPU when one is visible (WF_HOST_MORTON_SORT=1: never)
    SetSahBuild(&wf_build_bvh_sah);  // large SAH trees are built on the GPU when one is visible (WF_HOST_BVH_BUILD=1: never)
    g_init = true;
    return 0;
}

const char *wfh_last_error(void) { return g_lastError.c_str(); }

static wfh_scene *SceneLoadImpl(const char *path, int spp_override, int seed);
wfh_scene *wfh_scene_load(const char *path, int spp_override, int seed) {
    return Guard<wfh_scene *>(nullptr, [&] { return SceneLoadImpl(path, spp_override, seed); });
}
static wfh_scene *SceneLoadImpl(const char *path, int spp_override, int seed) {
    if (!g_init || !path) return nullptr;
    std::unique_ptr<wfh_scene> s(new wfh_scene());   // (ParseFiles / BuildSceneTables throw SceneError)
    s->opt.pixelSamples = spp_override;
    s->opt.seed = seed;
    s->opt.quiet = true;
    // WF_TABLE_CACHE=<dir>: the built tables are kept on disk, keyed by the scene file (path, size, mtime), spp and seed — the
    // N ranks of a multi-GPU job and repeated runs on one box then build the 10 M-triangle BVHs once (first come builds and
    // writes, atomically; the others load).  The scene's PLY / image files are assumed to change together with the .pbrt.
    std::string cacheFile;
    if (const char *dir = getenv("WF_TABLE_CACHE")) {
        struct stat st;
        if (*dir && stat(path, &st) == 0) {
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](const void *p, size_t n) { for (size_t i = 0; i < n; ++i) h = (h ^ ((const unsigned char *)p)[i]) * 1099511628211ull; };
            mix(path, strlen(path)); mix(&st.st_size, sizeof(st.st_size)); mix(&st.st_mtime, sizeof(st.st_mtime)); mix(&spp_override, 4); mix(&seed, 4);
            if (const char *sp = getenv("WF_BVH_SPLIT")) mix(sp, strlen(sp));
            // the library that builds the tables is part of the key (its file's size and modification time): a rebuilt table
            // builder / BVH builder never loads tables an older build wrote, even when no struct size changed
            Dl_info di;
            struct stat lst;
            if (dladdr((const void *)&wfh_last_error, &di) && di.dli_fname && stat(di.dli_fname, &lst) == 0) {
                mix(&lst.st_size, sizeof(lst.st_size)); mix(&lst.st_mtime, sizeof(lst.st_mtime));
            }
            char name[64];
            snprintf(name, sizeof(name), "/tables_%016llx.wftab", (unsigned long long)h);
            cacheFile = std::string(dir) + name;
            if (s->T.Load(cacheFile)) return s.release();
        }
    }
    ParseFiles({path}, &s->opt, &s->parsed);
    BuildSceneTables(s->parsed, s->opt, &s->T);
    if (!cacheFile.empty() && !s->T.Save(cacheFile)) fprintf(stderr, "Warning: could not write the scene-table cache %s\n", cacheFile.c_str());
    return s.release();
}
wfh_scene *wfh_scene_load_string(const char *text, int spp_override, int seed) {
    if (!g_init || !text) return nullptr;
    return Guard<wfh_scene *>(nullptr, [&] {
        std::unique_ptr<wfh_scene> s(new wfh_scene());
        s->opt.pixelSamples = spp_override;
        s->opt.seed = seed;
        s->opt.quiet = true;
        ParseString(text, &s->opt, &s->parsed);
        BuildSceneTables(s->parsed, s->opt, &s->T);
        return s.release();
    });
}
void wfh_scene_free(wfh_scene *s) { delete s; }

const wf_scene_desc *wfh_scene_desc(wfh_scene *s) { return s ? &s->T.desc : nullptr; }

int wfh_scene_info(wfh_scene *s, wfh_info *out) {
    if (!s || !out) return -1;
    const wf_film &F = s->T.desc.film;
    out->width = F.pixel_max[0] - F.pixel_min[0];
    out->height = F.pixel_max[1] - F.pixel_min[1];
    out->spp = s->T.spp;
    out->max_queue_size = s->T.maxQueueSize;
    out->n_passes = s->T.nPasses;
    out->scanlines_per_pass = s->T.scanlinesPerPass;
    out->n_triangles = s->T.desc.n_triangles;
    out->n_bvh_nodes = s->T.desc.n_bvh_nodes;
    out->n_lights = s->T.desc.n_lights;
    out->max_depth = s->T.desc.max_depth;
    out->save_fp16 = s->T.saveFP16;
    out->y0 = F.pixel_min[1];
    return 0;
}

// the final multi-channel image of a spectral or gbuffer film (SpectralFilm / GBufferFilm::GetImage) from the renderer's accumulators
static void FilmChannels(wfh_scene *s, std::vector<std::string> *nm, std::vector<float> *chans) {
    const wf_film &F = s->T.desc.film;
    const int W = F.pixel_max[0] - F.pixel_min[0], H = F.pixel_max[1] - F.pixel_min[1];
    std::vector<double> film((size_t)W * H * 4);
    s->renderer->DownloadFilm(film.data());
    if (F.type == WF_FILM_SPECTRAL) {
        std::vector<double> spectral((size_t)W * H * 2 * F.n_buckets);
        if (wf_film_spectral_download(s->renderer->Context(), spectral.data()) != 0) throw SceneError(wf_last_error());
        SpectralFilmImage(F, film.data(), spectral.data(), W, H, s->T.saveFP16, nm, chans);
    } else if (F.type == WF_FILM_GBUFFER) {
        std::vector<wf_gbuffer_pixel> gb((size_t)W * H);
        if (wf_film_gbuffer_download(s->renderer->Context(), gb.data()) != 0) throw SceneError(wf_last_error());
        GBufferFilmImage(F, film.data(), gb.data(), W, H, s->T.saveFP16, nm, chans);
    } else throw SceneError("the scene's film is an RGB film (use wfh_download_film / wfh_film_to_rgb)");
}
int wfh_film_channels(wfh_scene *s, int32_t *n_channels, char *names, float *pixels) {
    if (!s || !s->renderer) return -1;
    return Guard<int>(-1, [&] {
        const wf_film &F = s->T.desc.film;
        const int nc = F.type == WF_FILM_SPECTRAL ? 3 + F.n_buckets : F.type == WF_FILM_GBUFFER ? 25 : 0;
        if (nc == 0) throw SceneError("wfh_film_channels: the scene's film is an RGB film");
        if (n_channels) *n_channels = nc;
        if (!names && !pixels) return 0;
        std::vector<std::string> nm;
        std::vector<float> chans;
        FilmChannels(s, &nm, &chans);
        if (names) for (size_t i = 0; i < nm.size(); ++i) snprintf(names + 32 * i, 32, "%s", nm[i].c_str());
        if (pixels) memcpy(pixels, chans.data(), chans.size() * sizeof(float));
        return 0;
    });
}
int wfh_write_film_image(wfh_scene *s, const char *path) {
    if (!s || !s->renderer || !path) return -1;
    return Guard<int>(-1, [&] {
        const wf_film &F = s->T.desc.film;
        const int W = F.pixel_max[0] - F.pixel_min[0], H = F.pixel_max[1] - F.pixel_min[1];
        if (F.type != WF_FILM_RGB) {
            std::vector<std::string> nm;
            std::vector<float> chans;
            FilmChannels(s, &nm, &chans);
            return WriteEXRChannels(path, nm, chans.data(), W, H, s->T.saveFP16) ? 0 : -1;
        }
        std::vector<double> film((size_t)W * H * 4);
        s->renderer->DownloadFilm(film.data());
        std::vector<float> rgb((size_t)W * H * 3);
        FilmToRGB(F, film.data(), W, H, rgb.data(), s->T.saveFP16);
        return WriteFilmImage(s->T, path, rgb, W, H) ? 0 : -1;
    });
}
int wfh_read_nanovdb(const char *path, const char *grid_name, int32_t min[3], int32_t dim[3], float inv_mat[9], float vec[3], float *background, float *values) {
    if (!path || !grid_name) return -1;
    return Guard<int>(-1, [&] {
        VdbGrid g;
        ReadNanoVDBGrid(path, grid_name, &g);
        if (!g.found) return 1;
        for (int a = 0; a < 3; ++a) { if (min) min[a] = g.min[a]; if (dim) dim[a] = g.dim[a]; if (vec) vec[a] = g.vec[a]; }
        if (inv_mat) for (int a = 0; a < 9; ++a) inv_mat[a] = g.invMat[a];
        if (background) *background = g.background;
        if (values && !g.values.empty()) memcpy(values, g.values.data(), g.values.size() * sizeof(float));
        return 0;
    });
}
int wfh_build_bvh_host(int n, const float *bounds, int max_prims_in_node, wf_bvh_node *nodes_out, int32_t *order_out, int32_t *n_nodes_out) {
    if (n <= 0 || !bounds || !nodes_out || !order_out || !n_nodes_out) return -1;
    return Guard<int>(-1, [&] {
        std::vector<std::pair<int, B3>> prims((size_t)n);
        for (int i = 0; i < n; ++i) {
            const float *b = bounds + 6 * (size_t)i;
            prims[i].first = i;
            prims[i].second.pMin = V3{b[0], b[1], b[2]};
            prims[i].second.pMax = V3{b[3], b[4], b[5]};
        }
        std::vector<wf_bvh_node> nodes;
        std::vector<int32_t> order;
        if (BuildBVH(prims, max_prims_in_node, &nodes, &order, 0, true) != 0) return -1;
        memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(wf_bvh_node));
        memcpy(order_out, order.data(), order.size() * sizeof(int32_t));
        *n_nodes_out = (int32_t)nodes.size();
        return 0;
    });
}
int wfh_renderer_create(wfh_scene *s, int device, int samples_per_pass) {
    if (!s) return -1;
    return Guard<int>(-1, [&] { s->renderer = std::make_unique<WavefrontRenderer>(s->T, device, samples_per_pass); return 0; });
}
int wfh_renderer_create_strips(wfh_scene *s, int device, int samples_per_pass, int rank, int count, int height) {
    if (!s) return -1;
    return Guard<int>(-1, [&] { s->renderer = std::make_unique<WavefrontRenderer>(s->T, device, samples_per_pass, rank, count, height); return 0; });
}
int wfh_renderer_set_strips(wfh_scene *s, int rank, int count, int height) {
    if (!s || !s->renderer) return -1;
    return Guard<int>(-1, [&] { s->renderer->SetStrips(rank, count, height); return 0; });
}
int wfh_renderer_samples_per_pass(wfh_scene *s) { return (s && s->renderer) ? s->renderer->SamplesPerPass() : -1; }
wf_ctx *wfh_renderer_ctx(wfh_scene *s) { return (s && s->renderer) ? s->renderer->Context() : nullptr; }

double wfh_render(wfh_scene *s, int sample_begin, int sample_end, int sample_step, int fused) {
    if (!s || !s->renderer) return -1.0;
    return Guard<double>(-1.0, [&] { return s->renderer->Render(sample_begin, sample_end, sample_step, fused != 0); });
}
int wfh_clear_film(wfh_scene *s) {
    if (!s || !s->renderer) return -1;
    return Guard<int>(-1, [&] { s->renderer->ClearFilm(); return 0; });
}
int wfh_download_film(wfh_scene *s, double *dst) {
    if (!s || !s->renderer) return -1;
    return Guard<int>(-1, [&] { s->renderer->DownloadFilm(dst); return 0; });
}
int wfh_stats(wfh_scene *s, wf_render_stats *out) {
    if (!s || !s->renderer) return -1;
    s->renderer->Stats(out);
    return 0;
}
int wfh_film_to_rgb(wfh_scene *s, const double *film, float *rgb) {
    if (!s) return -1;
    const wf_film &F = s->T.desc.film;
    FilmToRGB(F, film, F.pixel_max[0] - F.pixel_min[0], F.pixel_max[1] - F.pixel_min[1], rgb, s->T.saveFP16);
    return 0;
}
int wfh_write_image(const char *path, const float *rgb, int w, int h) { return WriteImage(path, rgb, w, h) ? 0 : -1; }
int wfh_read_image(const char *path, const char *encoding, int32_t *width, int32_t *height, int32_t *n_channels, int32_t *format, float *pixels) {
    return Guard<int>(-1, [&] {
        if (!g_init || !path) throw SceneError("wfh_read_image: library not initialised or no path");
        HostImage img;
        ReadImage(path, ColorEnc::Parse(encoding ? encoding : "sRGB"), &img);
        if (width) *width = img.w;
        if (height) *height = img.h;
        if (n_channels) *n_channels = img.nc;
        if (format) *format = img.format;
        if (pixels) for (size_t i = 0; i < (size_t)img.w * img.h * img.nc; ++i) pixels[i] = img.Get(i);
        return 0;
    });
}

}  // extern "C"
