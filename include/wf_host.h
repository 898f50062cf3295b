/*
 * wf_host.h — C entry points of libwfhost.so: the host half of the wavefront renderer (scene parser,
 * flat-table builder, render loop) for non-C++ callers.  It mirrors, for this path only, what the
 * reference's CLI does between main() and RenderWavefront() (src/pbrt/cmd/pbrt.cpp:280-288,
 * src/pbrt/wavefront/wavefront.cpp:14-70).  The device work goes through include/wf_abi.h.
 */
#ifndef WF_HOST_H
#define WF_HOST_H

#include "wf_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wfh_scene wfh_scene;

typedef struct wfh_info {
    int32_t width, height, spp;
    int32_t max_queue_size, n_passes, scanlines_per_pass;   /* wavefront/integrator.cpp:227-236 */
    int32_t n_triangles, n_bvh_nodes, n_lights, max_depth;
    int32_t save_fp16, y0;
} wfh_info;

/* InitPBRT's table setup (pbrt.cpp:100-110): spectral tables + RGB->spectrum tables (generated on first
   use and cached under <data_dir>/cache) */
int wfh_init(const char *data_dir);
/* the message of the last scene or back-end error on this thread (entry points return NULL / -1; nothing in the library exits the process) */
const char *wfh_last_error(void);
/* ParseFiles + BasicScene::Create* for the supported subset; spp_override <= 0 keeps the file's value
   (--spp); seed as --seed.  Parse errors exit the process with a message, as the reference does. */
wfh_scene *wfh_scene_load(const char *path, int spp_override, int seed);
wfh_scene *wfh_scene_load_string(const char *text, int spp_override, int seed);
void wfh_scene_free(wfh_scene *s);
const wf_scene_desc *wfh_scene_desc(wfh_scene *s);
int wfh_scene_info(wfh_scene *s, wfh_info *out);
/* WavefrontPathIntegrator ctor on HIP device `device`.  samples_per_pass: sample indices carried by one pass
   (see wf_queues_alloc); <= 0 = automatic (~64 M rays in flight, capped at spp; env WF_SAMPLES_PER_PASS) */
int wfh_renderer_create(wfh_scene *s, int device, int samples_per_pass);
int wfh_renderer_samples_per_pass(wfh_scene *s);
wf_ctx *wfh_renderer_ctx(wfh_scene *s);
/* multi-GPU image partition (wf_set_strips): this renderer owns the scanline strips rank, rank + count, ... of `height` lines */
int wfh_renderer_set_strips(wfh_scene *s, int rank, int count, int height);
/* the same constructor for one rank of a multi-GPU job: the partition is known before the queues are allocated, so they are sized
   for the rank's own rows and a pass carries `count` times the sample indices (launches as large as a single GPU's, as far as the
   render has that many sample indices: 1080p on 8 GPUs fills ~64 M rays per pass from 242 sample indices on) */
int wfh_renderer_create_strips(wfh_scene *s, int device, int samples_per_pass, int rank, int count, int height);
/* Render(): sample indices begin, begin+step, ... < end; returns wall seconds (negative on error) */
double wfh_render(wfh_scene *s, int sample_begin, int sample_end, int sample_step, int fused);
int wfh_clear_film(wfh_scene *s);
int wfh_download_film(wfh_scene *s, double *dst /* [H][W][4] */);
int wfh_stats(wfh_scene *s, wf_render_stats *out);
/* RGBFilm::GetImage */
int wfh_film_to_rgb(wfh_scene *s, const double *film, float *rgb /* [H][W][3] */);
int wfh_write_image(const char *path, const float *rgb, int w, int h);
/* Image::Read (util/image.cpp:877-921) for .pfm, .png and .exr: the linear pixel values Image::GetChannel returns, channels
   Y | R G B | R G B A, rows top to bottom.  encoding: "sRGB", "linear" or "gamma <g>" for 8-bit files (NULL = sRGB).
   Call with pixels = NULL to get the size; returns 0, or -1 with wfh_last_error(). format: 0 8-bit, 1 half, 2 float storage. */
int wfh_read_image(const char *path, const char *encoding, int32_t *width, int32_t *height, int32_t *n_channels, int32_t *format, float *pixels);

/* SpectralFilm (film.h:401-530) and GBufferFilm (film.h:319-400): the final multi-channel image of a scene whose film is "spectral" —
   R, G, B and one channel per wavelength bucket ("S0.<centre>nm") — or "gbuffer" — R G B Albedo.{R,G,B} P.{X,Y,Z} dzdx dzdy N.{X,Y,Z}
   Ns.{X,Y,Z} u v Variance.{R,G,B} RelativeVariance.{R,G,B} — from the renderer's accumulators (GetImage).  names = n_channels strings of
   32 bytes each (or NULL); pixels = [height][width][n_channels] floats (or NULL to query n_channels first).
   wfh_write_film_image writes the scene's film — whatever its type — to `path` (.pfm / .exr for RGB films, .exr for the others). */
int wfh_film_channels(wfh_scene *s, int32_t *n_channels, char *names, float *pixels);
int wfh_write_film_image(wfh_scene *s, const char *path);
/* The NanoVDB reader of the "nanovdb" medium (csrc/host/nanovdb_io.cpp; parity unpinned: third-party format), for tools and tests:
   the float grid `grid_name` of `path` expanded over its index bounding box.  min / dim = origin and size of the block, inv_mat (9) and
   vec (3) = the grid's index-from-world map (index = inv_mat * (p - vec)), background; values = dim[0] * dim[1] * dim[2] floats (x
   fastest) or NULL to query the sizes first.  Returns 0, 1 if the file has no grid of that name, -1 on error (wfh_last_error). */
int wfh_read_nanovdb(const char *path, const char *grid_name, int32_t min[3], int32_t dim[3], float inv_mat[9], float vec[3], float *background, float *values);
/* The host SAH builder (csrc/host/bvh_build.cpp: BVHAggregate::buildRecursive + flattenBVH, cpu/aggregates.cpp:198-387, 505-521; pinned to the
   reference by tests/golden/bvh_stats.json) with the signature of the device builder wf_build_bvh_sah (include/wf_abi.h), for tools and the test
   that compares the two node for node: n boxes (min.xyz max.xyz) -> nodes_out (room for 2 n - 1), order_out[n], *n_nodes_out.  0 or -1. */
int wfh_build_bvh_host(int n, const float *bounds, int max_prims_in_node, wf_bvh_node *nodes_out, int32_t *order_out, int32_t *n_nodes_out);

#ifdef __cplusplus
}
#endif
#endif
