// integrator.h — WavefrontRenderer: the host loop of the wavefront path integrator over the C ABI.
#pragma once

#include "scene.h"

namespace wf {

class WavefrontRenderer {
  public:
    // uploads the scene tables to HIP device `device` and allocates the work queues
    // (WavefrontPathIntegrator ctor, wavefront/integrator.cpp:80-287)
    // samplesPerPass <= 0: automatic (env WF_SAMPLES_PER_PASS, else ~64 M rays in flight)
    WavefrontRenderer(const SceneTables &tables, int device, int samplesPerPass = 0);
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
};

}  // namespace wf
