// oracle/ref_build/ref_trace.cpp — TEST INFRASTRUCTURE ONLY (ours; links the shimmed reference build).
// Path tracer's trace: runs the reference's CPU WavefrontPathIntegrator for ONE sample index over the first pass of the image,
// issuing its stages in the order of WavefrontPathIntegrator::Render (wavefront/integrator.cpp:357-432), and prints after every
// stage of every depth what the path state holds — queue sizes, PixelSampleState.L, the rays and shadow rays each material stage
// pushed — as hex floats, one line per item, sorted by pixel index.  `oracle/_build/wf_cpu --trace-path` prints the same lines
// from the restated stages; `diff` of the two shows the first stage at which a fuzz finding diverges (tools/trace_diff.py).
// Meant for images of a few pixels (Film "integer pixelbounds").  Private members are reached with the test-only
// `#define private public`; nothing in the reference is modified.
//   ref_trace scene.pbrt sampleIndex
#include <algorithm>
#include <cstdio>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <thread>
#include <functional>
#include <unordered_map>
#include <set>
#include <list>
#include <array>
#include <optional>
#include <variant>
#include <iostream>
#include <fstream>
#include <cstring>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <future>
#include <shared_mutex>
#include <typeindex>
#include <typeinfo>

#define private public
#define protected public
#include <pbrt/pbrt.h>
#include <pbrt/cameras.h>
#include <pbrt/wavefront/integrator.h>
#undef private
#undef protected
#include <pbrt/options.h>
#include <pbrt/parser.h>
#include <pbrt/scene.h>
#include <pbrt/materials.h>

using namespace pbrt;

static std::string S4(const SampledSpectrum &s) {
    char b[160];
    snprintf(b, sizeof(b), "%a %a %a %a", s[0], s[1], s[2], s[3]);
    return b;
}

static void Lines(const char *tag, int depth, std::vector<std::pair<int, std::string>> &v) {
    std::sort(v.begin(), v.end());
    for (auto &p : v) printf("d%d %s pix %d %s\n", depth, tag, p.first, p.second.c_str());
}

static void DumpL(WavefrontPathIntegrator *in, const char *tag, int depth, int n) {
    for (int i = 0; i < n; ++i) {
        SampledSpectrum L = in->pixelSampleState.L[i];
        printf("d%d L.%s pix %d %s\n", depth, tag, i, S4(L).c_str());
    }
}

static void DumpRays(const char *tag, int depth, RayQueue *q) {
    std::vector<std::pair<int, std::string>> v;
    for (int i = 0; i < q->Size(); ++i) {
        RayWorkItem r = (*q)[i];
        char b[1024];
        snprintf(b, sizeof(b), "o %a %a %a d %a %a %a depth %d beta %s r_u %s r_l %s etaScale %a spec %d anyns %d medium %d", r.ray.o.x, r.ray.o.y, r.ray.o.z, r.ray.d.x,
                 r.ray.d.y, r.ray.d.z, r.depth, S4(r.beta).c_str(), S4(r.r_u).c_str(), S4(r.r_l).c_str(), r.etaScale, (int)r.specularBounce, (int)r.anyNonSpecularBounces,
                 r.ray.medium ? 1 : 0);
        v.emplace_back(r.pixelIndex, b);
    }
    Lines(tag, depth, v);
}

static void DumpShadow(const char *tag, int depth, ShadowRayQueue *q) {
    std::vector<std::pair<int, std::string>> v;
    for (int i = 0; i < q->Size(); ++i) {
        ShadowRayWorkItem r = (*q)[i];
        char b[1024];
        snprintf(b, sizeof(b), "o %a %a %a d %a %a %a tMax %a Ld %s r_u %s r_l %s", r.ray.o.x, r.ray.o.y, r.ray.o.z, r.ray.d.x, r.ray.d.y, r.ray.d.z, r.tMax, S4(r.Ld).c_str(),
                 S4(r.r_u).c_str(), S4(r.r_l).c_str());
        v.emplace_back(r.pixelIndex, b);
    }
    Lines(tag, depth, v);
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_trace scene.pbrt sampleIndex\n"); return 1; }
    const int sampleIndex = atoi(argv[2]);
    PBRTOptions opt;
    opt.wavefront = true;
    opt.quiet = true;
    opt.seed = 0;
    opt.nThreads = 1;
    InitPBRT(opt);
    {
        BasicScene scene;
        BasicSceneBuilder builder(&scene);
        ParseFiles(&builder, {std::string(argv[1])});
        WavefrontPathIntegrator *in = new WavefrontPathIntegrator(pstd::pmr::get_default_resource(), scene);
        Bounds2i pb = in->film.PixelBounds();
        const int y0 = pb.pMin.y;
        const int nPix = std::min(in->maxQueueSize, (pb.pMax.x - pb.pMin.x) * std::min(in->scanlinesPerPass, pb.pMax.y - pb.pMin.y));
        in->rayQueues[0]->Reset();
        in->GenerateCameraRays(y0, Transform(), sampleIndex);
        DumpRays("camera", 0, in->rayQueues[0]);
        for (int depth = 0; true; ++depth) {
            in->NextRayQueue(depth)->Reset();
            if (in->mediumSampleQueue) in->mediumSampleQueue->Reset();
            if (in->mediumScatterQueue) in->mediumScatterQueue->Reset();
            if (in->escapedRayQueue) in->escapedRayQueue->Reset();
            in->hitAreaLightQueue->Reset();
            in->basicEvalMaterialQueue->Reset();
            in->universalEvalMaterialQueue->Reset();
            if (in->bssrdfEvalQueue) in->bssrdfEvalQueue->Reset();
            if (in->subsurfaceScatterQueue) in->subsurfaceScatterQueue->Reset();
            in->GenerateRaySamples(depth, sampleIndex);
            in->aggregate->IntersectClosest(in->maxQueueSize, in->CurrentRayQueue(depth), in->escapedRayQueue, in->hitAreaLightQueue, in->basicEvalMaterialQueue,
                                            in->universalEvalMaterialQueue, in->mediumSampleQueue, in->NextRayQueue(depth));
            printf("d%d after-closest rays %d escaped %d hitlight %d medium %d next_pre %d\n", depth, in->CurrentRayQueue(depth)->Size(),
                   in->escapedRayQueue ? in->escapedRayQueue->Size() : 0, in->hitAreaLightQueue->Size(), in->mediumSampleQueue ? in->mediumSampleQueue->Size() : 0,
                   in->NextRayQueue(depth)->Size());
            if (in->mediumSampleQueue) {
                // the medium-sample items: tMax seeds the delta-tracking RNG (media.cpp:44), the ray's time picks an AnimatedPrimitive's transformation
                std::vector<std::pair<int, std::string>> v;
                for (int i = 0; i < in->mediumSampleQueue->Size(); ++i) {
                    MediumSampleWorkItem w = (*in->mediumSampleQueue)[i];
                    char b[256];
                    snprintf(b, sizeof(b), "tMax %a time %a", w.tMax, w.ray.time);
                    v.emplace_back(w.pixelIndex, b);
                }
                Lines("msample", depth, v);
            }
            in->SampleMediumInteraction(depth);
            if (in->haveMedia) DumpL(in, "medium", depth, nPix);
            in->HandleEscapedRays();
            in->HandleEmissiveIntersection();
            DumpL(in, "emitted", depth, nPix);
            if (depth == in->maxDepth) break;
            in->EvaluateMaterialsAndBSDFs(depth, Transform());
            DumpRays("next", depth, in->NextRayQueue(depth));
            DumpShadow("shadow", depth, in->shadowRayQueue);
            in->TraceShadowRays(depth);
            DumpL(in, "shadowed", depth, nPix);
            if (in->haveSubsurface) {
                printf("d%d bssrdf-items %d\n", depth, in->bssrdfEvalQueue->Size());
                in->SampleSubsurface(depth);
                printf("d%d sss-items %d\n", depth, in->subsurfaceScatterQueue->Size());
                DumpRays("next+sss", depth, in->NextRayQueue(depth));
                DumpL(in, "sss", depth, nPix);
            }
            if (in->NextRayQueue(depth)->Size() == 0) break;
        }
    }
    return 0;
}
