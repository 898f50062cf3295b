// integrator.h — WavefrontRenderer: the host loop of the wavefront path integrator over the C ABI.
#pragma once

#include "scene.h"

namespace wf {

class WavefrontRenderer {
  public:
    // uploads the scene tables to HIP device `device` and allocates the work queues
    // (WavefrontPathIntegrator ctor, wavefront/integrator.cpp:80-287)
    // samplesPerPass <= 0: automatic (env WF_SAMPLES_PER_PASS, else ~64 M rays in flight)
    // stripCount > 1: the renderer is one of stripCount ranks of a multi-GPU job and owns the scanline strips stripRank,
    // stripRank + stripCount, ... from the start: its queues are sized for ITS rows, so that a pass of a rank carries as many
    // rays as a pass of a single GPU does (more sample indices of fewer pixels; bounded by the number of sample indices rendered)
    WavefrontRenderer(const SceneTables &tables, int device, int samplesPerPass = 0, int stripRank = 0, int stripCount = 1, int stripHeight = 16);
    int SamplesPerPass() const { return samplesPerPass; }
    ~WavefrontRenderer();
    WavefrontRenderer(const WavefrontRenderer &) = delete;
    // returns wall seconds (Render(), integrator.cpp:308,483-487).  fused: one wf_render_pass call per
    // pass instead of one C-ABI call per stage (same launches, fewer boundary crossings)
    double Render(int sampleBegin, int sampleEnd, int sampleStep, bool fused = true);
    // multi-GPU image partition: own the scanline strips rank, rank + count, ... of `height` lines (count = 1: whole image)
    void SetStrips(int rank, int count, int height);
    void ClearFilm();
    void DownloadFilm(double *dst /* [H][W][4] */);
    void UploadFilm(const double *src);
    void Stats(wf_render_stats *s);
    wf_ctx *Context() { return ctx; }
    const SceneTables &Tables() const { return T; }

  private:
    const SceneTables &T;
    wf_ctx *ctx = nullptr;
    int samplesPerPass = 1;
    int localRows = 0;  // scanlines this renderer owns (set by the ctor = image height, or by SetStrips)
    int rowsPerPass = 0;   // local rows one pass covers (the queues hold rowsPerPass x width x samplesPerPass items)
};

// Multi-GPU rendering inside ONE process (SURVEY 8(b) "one host thread per device", 8(e) "image tiled across the GPUs"; the reference
// has nothing here: gpu/util.cpp:80-85 is one cudaSetDevice).  The scene tables are replicated, device k of `devices` owns the scanline
// strips k, k + N, ... of 16 lines (wf_set_strips: interleaved, so that sky and foliage are dealt evenly) and renders them from a host
// thread of its own, nothing is exchanged during rendering, and the strips are then copied peer to peer into the first device's film
// (wf_film_gather_strips: 1/N of the film per device).  Every pixel's accumulators are formed on one device in the single-device
// order: the gathered film is BIT-IDENTICAL to a single-device render.  `devices` may name one device several times (several
// contexts on one GPU: how the path is tested on a one-GPU box).
class MultiDeviceRenderer {
  public:
    MultiDeviceRenderer(const SceneTables &tables, const std::vector<int> &devices, int samplesPerPass = 0, int stripHeight = 16);
    ~MultiDeviceRenderer();
    // wall seconds of the slowest device + the gather; perDevice (optional): render seconds of every device, gather seconds last
    double Render(int sampleBegin, int sampleEnd, std::vector<double> *perDevice = nullptr);
    WavefrontRenderer &Primary() { return *renderers[0]; }   // holds the gathered film and the summed statistics after Render()
    int Count() const { return (int)renderers.size(); }

  private:
    std::vector<WavefrontRenderer *> renderers;
};

}  // namespace wf
