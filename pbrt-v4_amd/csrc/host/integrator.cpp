// integrator.cpp — host side of the wavefront path integrator: a restatement of
// WavefrontPathIntegrator::Render (wavefront/integrator.cpp:290-493) that issues one C-ABI call
// (include/wf_abi.h -> libwfhip.so) where the reference launches a kernel or calls the
// WavefrontAggregate.  Everything is enqueued on one in-order HIP stream; the host synchronises once,
// at the end (GPUWait(), integrator.cpp:483-485).  There is no CPU code path here: without libwfhip.so and
// a visible gfx950 device, construction fails.
#include "integrator.h"

#include <chrono>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace wf {

static void Check(int rc, const char *what) {
    if (rc != 0) {
        // CUDA_CHECK -> LOG_FATAL in the reference (gpu/util.h:35-49)
        throw SceneError(std::string("Fatal: ") + what + " failed: " + wf_last_error());
    }
}

WavefrontRenderer::WavefrontRenderer(const SceneTables &tables, int device, int samplesPerPassArg, int stripRank, int stripCount, int stripHeight) : T(tables) {
    // Wavefront sizing.  The reference carries one sample index per pass (<= 2^20 rays, integrator.cpp:227-236).
    // On a 256-CU part a 1 M-ray launch is two rounds of a latency-bound walk; the queues here carry several
    // sample indices per pass (default: up to ~64 M rays in flight, ~35 GB of queues out of 288 GB; measured at 1080p: 16 -> 437, 32 -> 458, 64 -> 472 Msamples/s).
    // A rank of a multi-GPU job owns 1/N of the scanlines: its pass covers its own rows only and carries N times the sample
    // indices instead, so its launches stay as large as a single GPU's (as far as the render has that many sample indices).
    const int W = T.desc.film.pixel_max[0] - T.desc.film.pixel_min[0];
    Check(wf_ctx_create(device, &ctx), "wf_ctx_create");
    Check(wf_scene_upload(ctx, &T.desc), "wf_scene_upload");
    localRows = T.desc.film.pixel_max[1] - T.desc.film.pixel_min[1];
    if (stripCount > 1) Check(wf_set_strips(ctx, stripRank, stripCount, stripHeight, &localRows), "wf_set_strips");
    samplesPerPass = samplesPerPassArg;
    if (samplesPerPass <= 0) {
        const char *env = getenv("WF_SAMPLES_PER_PASS");
        if (env) samplesPerPass = atoi(env);
    }
    // Pass geometry (round 4).  A pass costs a fixed ~6.5 ms of launch latencies and thin late-depth launches on top of its
    // throughput-proportional work, so the frame is cut into as FEW passes as the 64 Mi-item queues allow: with few sample indices
    // (bench --steps 20: 41.5 M rays) ONE pass over all rows instead of the reference's two 540-row bands; with many (256 spp) the
    // band count that minimises bands x ceil(spp / samplesPerPass), ties going to the reference's banding (more sample slots per
    // pixel in a pass: the samples of a pixel are neighbours in every queue).  The film adds a pixel's samples in sample order for
    // any geometry: the image does not depend on it (test_samples_per_pass_invariance).
    const long budget = 64l << 20;
    const int refBands = std::max(1, (localRows + std::max(1, T.scanlinesPerPass) - 1) / std::max(1, T.scanlinesPerPass));
    const int spp = std::max(1, T.spp);
    int bestBands = refBands;
    long bestPasses = -1;
    for (int nb = 1; nb <= refBands; ++nb) {
        const int rows = (localRows + nb - 1) / nb;
        long S = samplesPerPass > 0 ? samplesPerPass : std::max(1l, budget / ((long)W * rows));
        S = std::min<long>(S, spp);
        if ((long)W * rows * S > budget && S > 1) continue;
        const long passes = (long)((localRows + rows - 1) / rows) * ((spp + S - 1) / S);
        if (bestPasses < 0 || passes <= bestPasses) { bestPasses = passes; bestBands = nb; }
    }
    if (getenv("WF_REFERENCE_BANDS")) bestBands = refBands;   // A/B: the round-3 geometry
    rowsPerPass = std::max(1, (localRows + bestBands - 1) / bestBands);
    const int pixelsPerPass = W * rowsPerPass;
    if (samplesPerPass <= 0) samplesPerPass = (int)std::max(1l, budget / pixelsPerPass);
    samplesPerPass = std::min(samplesPerPass, spp);
    Check(wf_queues_alloc(ctx, pixelsPerPass, samplesPerPass), "wf_queues_alloc");
    Check(wf_film_clear(ctx), "wf_film_clear");
    Check(wf_sync(ctx), "wf_sync");
}

WavefrontRenderer::~WavefrontRenderer() {
    if (ctx) wf_ctx_destroy(ctx);
}

// the image partition of multi-GPU rendering: this renderer owns the strips rank, rank + count, ... (wf_set_strips)
void WavefrontRenderer::SetStrips(int rank, int count, int height) {
    Check(wf_set_strips(ctx, rank, count, height, &localRows), "wf_set_strips");
    // (the queues keep the size they were allocated with: a partition set after construction runs in passes of at most that many rows)
}

void WavefrontRenderer::ClearFilm() {
    Check(wf_film_clear(ctx), "wf_film_clear");
}

// Render sample indices sampleBegin, sampleBegin+sampleStep, ... < sampleEnd over every pass of the
// image.  (A single GPU renders [0, spp) with step 1; with N GPUs rank r renders r, r+N, ... — the
// per-pixel sample sets are those of the single-GPU render because the sampler is keyed on
// (pixel, sampleIndex, dimension) only, samplers.h:252-255.)
double WavefrontRenderer::Render(int sampleBegin, int sampleEnd, int sampleStep, bool fused) {
    const wf_film &F = T.desc.film;
    auto t0 = std::chrono::steady_clock::now();
    const int maxDepth = T.desc.max_depth;
    for (int sampleIndex = sampleBegin; sampleIndex < sampleEnd; sampleIndex += sampleStep * samplesPerPass) {
        // this batch of passes carries sampleIndex, sampleIndex + sampleStep, ... (at most samplesPerPass of them)
        const int remaining = (sampleEnd - sampleIndex + sampleStep - 1) / sampleStep;
        Check(wf_set_pass_samples(ctx, sampleStep, std::min(samplesPerPass, remaining)), "wf_set_pass_samples");
        for (int y0 = F.pixel_min[1]; y0 < F.pixel_min[1] + localRows; y0 += rowsPerPass) {
            if (fused) {
                Check(wf_render_pass(ctx, y0, sampleIndex), "wf_render_pass");
                continue;
            }
            // integrator.cpp:357-371
            Check(wf_reset_ray_queue(ctx, 0), "wf_reset_ray_queue");
            Check(wf_gen_camera_rays(ctx, y0, sampleIndex), "wf_gen_camera_rays");
            // integrator.cpp:374-432
            for (int wavefrontDepth = 0; true; ++wavefrontDepth) {
                Check(wf_reset_stage_queues(ctx, wavefrontDepth), "wf_reset_stage_queues");
                Check(wf_gen_ray_samples(ctx, wavefrontDepth, sampleIndex), "wf_gen_ray_samples");
                Check(wf_intersect_closest(ctx, wavefrontDepth), "wf_intersect_closest");
                if (T.desc.have_media) Check(wf_medium_sample(ctx, wavefrontDepth), "wf_medium_sample");  // integrator.cpp:416
                Check(wf_handle_escaped(ctx, wavefrontDepth), "wf_handle_escaped");
                Check(wf_handle_emissive(ctx, wavefrontDepth), "wf_handle_emissive");
                if (wavefrontDepth == maxDepth) break;
                for (int m = 0; m < WF_MAT_NTYPES; ++m)
                    if (T.materialTypePresent[m] && m != WF_MAT_INTERFACE)
                        Check(wf_eval_material(ctx, m, wavefrontDepth), "wf_eval_material");
                // TraceShadowRays, integrator.cpp:575-586
                if (T.desc.have_media) Check(wf_intersect_shadow_tr(ctx, wavefrontDepth), "wf_intersect_shadow_tr");
                else Check(wf_intersect_shadow(ctx, wavefrontDepth), "wf_intersect_shadow");
                // SampleSubsurface, integrator.cpp:431 -> wavefront/subsurface.cpp:18-203 (ends with its own TraceShadowRays)
                if (T.materialTypePresent[WF_MAT_SUBSURFACE]) {
                    Check(wf_subsurface_probe(ctx, wavefrontDepth), "wf_subsurface_probe");
                    Check(wf_intersect_one_random(ctx), "wf_intersect_one_random");
                    Check(wf_subsurface_scatter(ctx, wavefrontDepth), "wf_subsurface_scatter");
                    if (T.desc.have_media) Check(wf_intersect_shadow_tr(ctx, wavefrontDepth), "wf_intersect_shadow_tr");
                    else Check(wf_intersect_shadow(ctx, wavefrontDepth), "wf_intersect_shadow");
                }
            }
            Check(wf_update_film(ctx), "wf_update_film");
        }
    }
    Check(wf_sync(ctx), "wf_sync");
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void WavefrontRenderer::DownloadFilm(double *dst) { Check(wf_film_download(ctx, dst), "wf_film_download"); }
void WavefrontRenderer::UploadFilm(const double *src) { Check(wf_film_upload(ctx, src), "wf_film_upload"); }
void WavefrontRenderer::Stats(wf_render_stats *s) { Check(wf_stats_download(ctx, s), "wf_stats_download"); }

MultiDeviceRenderer::MultiDeviceRenderer(const SceneTables &tables, const std::vector<int> &devices, int samplesPerPass, int stripHeight) {
    const int n = (int)devices.size();
    if (n < 1) throw SceneError("Fatal: MultiDeviceRenderer: no devices");
    renderers.resize(n, nullptr);
    // every context is created on the thread that will drive it later?  Not needed: the C ABI makes the context's device current in the
    // calling thread at every entry (wf_backend.hip: useDevice).  The uploads run concurrently, one thread per device.
    std::vector<std::thread> th;
    std::vector<std::string> err(n);
    for (int k = 0; k < n; ++k)
        th.emplace_back([&, k]() {
            try { renderers[k] = new WavefrontRenderer(tables, devices[k], samplesPerPass, k, n, stripHeight); } catch (const std::exception &e) { err[k] = e.what(); }
        });
    for (auto &t : th) t.join();
    for (int k = 0; k < n; ++k)
        if (!err[k].empty()) {
            for (WavefrontRenderer *r : renderers) delete r;
            throw SceneError(err[k]);
        }
}

MultiDeviceRenderer::~MultiDeviceRenderer() {
    for (WavefrontRenderer *r : renderers) delete r;
}

double MultiDeviceRenderer::Render(int sampleBegin, int sampleEnd, std::vector<double> *perDevice) {
    const int n = (int)renderers.size();
    auto t0 = std::chrono::steady_clock::now();
    std::vector<double> secs(n + 1, 0.0);
    std::vector<std::string> err(n);
    std::vector<std::thread> th;
    for (int k = 0; k < n; ++k)
        th.emplace_back([&, k]() {
            try { secs[k] = renderers[k]->Render(sampleBegin, sampleEnd, 1); } catch (const std::exception &e) { err[k] = e.what(); }
        });
    for (auto &t : th) t.join();
    for (int k = 0; k < n; ++k)
        if (!err[k].empty()) throw SceneError(err[k]);
    auto tg = std::chrono::steady_clock::now();
    for (int k = 1; k < n; ++k) {
        Check(wf_film_gather_strips(renderers[0]->Context(), renderers[k]->Context()), "wf_film_gather_strips");
        Check(wf_stats_add(renderers[0]->Context(), renderers[k]->Context()), "wf_stats_add");
    }
    auto t1 = std::chrono::steady_clock::now();
    secs[n] = std::chrono::duration<double>(t1 - tg).count();
    if (perDevice) *perDevice = secs;
    return std::chrono::duration<double>(t1 - t0).count();
}

}  // namespace wf
