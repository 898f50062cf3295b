// wf_kernels.h — the per-item bodies of the wavefront stages: what the reference writes as lambdas passed
// to ParallelFor / ForAllQueued (wavefront/integrator.h:91-113, workqueue.h:118-137) are named functions
// here, launched by the hand-written HIP kernels in csrc/hip/ (and looped over by the CPU checker under
// oracle/).  One function per K-row of SURVEY.md §2.3.
//
// Queue layout (HBM, SoA, one array per member; everything 16 bytes wide where the data allows so a wave
// reads/writes 1 KiB per instruction):
//   PixelSampleState  per pixel of the pass        workitems.h:107-116 (visibleSurface dropped: RGBFilm)
//   RayQueue x2       144 B/ray                    workitems.h:119-130 (lambda is read through pixelIndex:
//                                                  it always equals PixelSampleState.lambda)
//   HitRecord         16 B per ray slot            {triangle id, b0, b1, b2}: the MaterialEvalWorkItem /
//                                                  HitAreaLightWorkItem payloads (252 B / 192 B in the
//                                                  reference, workitems.h:133-172,265-325) are NOT written
//                                                  out: the consumers rebuild the SurfaceInteraction from
//                                                  the triangle and the barycentrics, bit-identically.
//   index queues      4 B per entry                escaped rays, emitter hits, one per material type
//   ShadowRayQueue    80 B/ray                     workitems.h:160-172
#pragma once

#include "wf_bxdf.h"
#include "wf_camera.h"
#include "wf_lights.h"
#include "wf_media.h"
#include "wf_bssrdf.h"
#include "wf_hair.h"
#include "wf_measured.h"

namespace wf {

struct alignas(16) F4 { float x, y, z, w; };
struct alignas(16) I4 { int32_t x, y, z, w; };
struct alignas(8) I2 { int32_t x, y; };
WF_HD F4 toF4(S4 s) { return F4{s[0], s[1], s[2], s[3]}; }
WF_HD S4 toS4(F4 f) { return S4{{f.x, f.y, f.z, f.w}}; }

enum {
    CNT_RAY0 = 0, CNT_RAY1 = 1, CNT_ESCAPED = 2, CNT_HITLIGHT = 3, CNT_SHADOW = 4, CNT_MAT0 = 5,
    CNT_MEDIUM_SAMPLE = CNT_MAT0 + WF_MAT_NTYPES, CNT_MEDIUM_SCATTER, CNT_MIX, CNT_RETRACE, CNT_BSSRDF, CNT_SSS, CNT_CURSOR, CNT_TR0, CNT_TR1,
    CNT_RETRACE_HEAD, CNT_WAVES_DONE,   // the closest-hit kernel's in-kernel near-tie queue (wf_backend.hip: DrainRetrace)
    CNT_CURSOR_SHADOW,                  // the any-hit launch's work cursor (CNT_CURSOR: the closest-hit launch's)
    CNT_MEDIUM_ROUTE,                   // medium-sample items that reached their surface (KMediumRoute)
    CNT_DEFER, CNT_DEFER_SHADOW,        // rays the triangle walk handed to the general-primitive walk (round 6, wf_backend.hip "TWO-CLASS TRAVERSAL")
    CNT_CURSOR_MEDIUM,                  // the medium-sample launch's work cursor (round 6: k_medium_sample with wave-level refill)
    CNT_COUNT
};
// every counter sits alone in its own 256-byte line: same-line atomics serialise at one L2 channel
// (~88 returning atomics/us on MI355X), and with adjacent ints every queue of a stage shared that budget
constexpr int CNT_STRIDE = 64;
static_assert(CNT_COUNT <= 32, "k_reset takes the counters to zero as a 32-bit mask");
enum { RAYFLAG_SPECULAR_BOUNCE = 1, RAYFLAG_ANY_NONSPECULAR = 2 };

struct RayQueueV {
    F4 *o;     // o.xyz, time
    F4 *d;     // d.xyz, etaScale
    F4 *beta, *r_u, *r_l;
    F4 *ctx0;  // prevIntrCtx.pi lo.xyz, hi.x
    F4 *ctx1;  // prevIntrCtx.pi hi.yz, n.xy
    F4 *ctx2;  // n.z, ns.xyz
    I4 *meta;  // pixelIndex, depth, flags, medium
};
struct ShadowQueueV {
    F4 *o;     // o.xyz, tMax
    F4 *d;     // d.xyz, pixelIndex (int bits)
    F4 *Ld, *r_u, *r_l;
    int32_t *medium;  // ray.medium (allocated when the scene has media)
};
// K12 work items (workitems.h:175-216), array of structures: subsurface scattering is a side path
struct BssrdfItem {
    F4 beta, r_u;
    V3 p; float etaScale;
    V3 wo; int depth;
    N3 n; int pixelIndex;
    N3 ns; int material;
    V3 dpdus; int mediumInside;
    V2 uv; int mediumOutside; int pad;
};
struct SubsurfaceItem {
    V3 p0, p1;
    int depth, material;
    TabulatedBSSRDF bssrdf;
    F4 beta, r_u;
    float reservoirPDF;
    P3i pi;             // ssi (bssrdf.h:32-66)
    N3 n, ns;
    V3 dpdu, dpdv, dpdus, dpdvs;
    int mediumInside, mediumOutside;
    float etaScale;
    int pixelIndex;
};
struct WorkState {
    int maxQueueSize;      // queue capacity = pixelsPerPass * samplesPerPass
    // MI355X-first wavefront sizing: a pass carries `samplesPerPass` sample indices of every pixel of the
    // scanline band (the reference carries one, capped at 2^20 rays: integrator.cpp:227-236).  Item index
    // i = s * pixelsPerPass + p: sample slot s, pixel p of the band.  samplesPerPass = 1 is the reference.
    int pixelsPerPass, samplesPerPass;
    // slotStride > 0: PIXEL-major items, i = p * slotStride + s with slotStride = the sample slots the current pass uses — the samples
    // of one pixel are neighbours in every queue (one wave holds a few pixels' samples: nearly identical camera rays, the same
    // triangles / texels / lights at the first hits).  0: sample-major as above.  Results do not depend on it (every item touches only
    // its own pixelIndex slot; the film adds a pixel's slots in slot order either way).
    int slotStride = 0;
    int drainEpoch = 0;   // tag of the current closest-hit launch's near-tie queue entries (wf_backend.hip: DrainRetrace)
    // image partition for multi-GPU rendering (interleaved strips): this context owns the scanline strips
    // stripRank, stripRank + stripCount, ... of stripHeight lines each — localRows lines in all; a pass covers a band of
    // LOCAL rows.  stripCount = 1: the whole image (local row = image row).
    int stripRank, stripCount, stripHeight, localRows;
    // PixelSampleState
    float *filterWeight;
    I2 *pPixel;
    F4 *lambda, *lambdaPdf, *L, *cameraRayWeight;
    F4 *samples0;  // direct.uc, direct.u.x, direct.u.y, indirect.uc
    F4 *samples1;  // indirect.u.x, indirect.u.y, indirect.rr, -
    F4 *samples2;  // subsurface.uc, subsurface.u.x, subsurface.u.y, - (allocated when the scene has a subsurface material)
    // ZSobol TopDigits() of every pixel of the band for the five dimensions one stage draws (dim0 + {0,1,3,4,6}),
    // [5][pixelsPerPass]; refreshed by KSampleTops before the stage.  Null: not used (CPU checker, > 32-bit indices).
    uint32_t *sampleTops;
    RayQueueV rq[2];
    F4 *hit;       // per ray slot of the current queue: triangle id (int bits), b0, b1, b2
    int32_t *hitInst;  // per ray slot: object instance of the hit (-1 = top level); allocated when the scene has instances
    float *hitT;   // tHit of the closest hit (only with media: MediumSampleWorkItem.tMax, workitems.h:219-250)
    int32_t *escapedQ, *hitLightQ;
    // MediumSampleQueue / MediumScatterQueue (workitems.h:219-262) as index queues over the current ray queue: the
    // items' payload is the ray slot itself (+ hit / hitT), beta, r_u, r_l are updated in place in the ray queue
    int32_t *mediumSampleQ, *mediumScatterQ;
    int32_t *mediumRouteQ = nullptr;   // HIP back end only: ray slots whose medium sampling ended at the surface (routed by KMediumRoute)
    F4 *scatterP;  // per ray slot: scattering point p.xyz, HG g
    int32_t *mixMat;  // per ray slot: the material id a MixMaterial hit resolved to (allocated when sv.haveMix)
    int32_t *mixQ;    // HIP traversal kernel only: hits on a MixMaterial, resolved by the kernel that follows it
    uint32_t *routeCode;  // HIP traversal kernel only (split routing): per ray slot, the hit primitive's routing code | ROUTE_SKIP
    int32_t *deferQ;    // HIP traversal kernels only: rays (indices into the ray / shadow queue) whose triangle walk met a quadric / patch / curve leaf
    int32_t *retraceQ;  // HIP traversal kernel only: rays whose closest hit was a near-tie (wf_traverse.h), re-traced in reference order
    unsigned long long *retraceQ64;  // the same for the kernels that drain the queue themselves: ray index | bound bits << 32, ~0 = not yet written
    int32_t *matQ[WF_MAT_NTYPES];
    struct BssrdfItem *bssrdfQ;       // GetBSSRDFAndProbeRayQueue / SubsurfaceScatterQueue (K12; allocated when sv.haveSubsurface)
    struct SubsurfaceItem *sssQ;
    ShadowQueueV sq;
    // the state of the material stage's items between its two kernels (wf_mat.hip: NeeIO): planes of maxQueueSize x 16 bytes
    F4 *neeRec = nullptr;
    // the time of every pixel sample's path (= its camera ray's: every ray of a path carries it), kept for the shadow rays of scenes with
    // animated primitives (the shadow queue has the pixel index, not the time); null otherwise
    float *pathTime = nullptr;
    // the transmittance wavefront (HIP back end, scenes with media): per shadow ray the state of TraceTransmittance between segments —
    // current origin / direction (+ medium id in trD.w), T_ray, r_u, r_l, the PCG32 state — and two index queues of the live rays
    F4 *trO, *trD, *trT, *trRu, *trRl;
    I4 *trRng;
    int32_t *trQ[2];
    int32_t *counters;              // CNT_* (x CNT_STRIDE ints apart)
    double *film;                   // [pixels][4]: rgbSum[3], weightSum (film.h:302-307)
    double *filmSpectral;           // SpectralFilm: [pixels][2 * n_buckets]: bucketSums, weightSums (film.h:514-524); else null
    wf_gbuffer_pixel *filmGBuffer;  // GBufferFilm: one record per pixel; else null
    // PixelSampleState::visibleSurface (workitems.h:113; film.h:34-62), GBufferFilm only: p.xyz + set flag, n.xyz + uv.x, ns.xyz + uv.y,
    // dpdx.xyz + time, dpdy.xyz, albedo
    F4 *vsP, *vsN, *vsNs, *vsDpdx, *vsDpdy, *vsAlbedo;
    unsigned long long *stats;      // cameraRays, indirect[64], shadow[64]
    unsigned long long *trav;       // wf_traversal_counters (8 x u64) or null
};

// ---------------------------------------------------------------------------------------------
// WorkQueue::AllocateEntry (workqueue.h:92-102).  On the device: one atomic per wave per destination
// queue (ballot + popcount prefix), the slot broadcast from the leader lane.
WF_HD int QueueAlloc(int32_t *counter) {
#if defined(__HIP_DEVICE_COMPILE__)
    int result = 0;
    bool done = false;
    while (!done) {
        // waterfall over the distinct counters addressed by the active lanes
        unsigned long long mine = (unsigned long long)counter;
        unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)mine);
        unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(mine >> 32));
        unsigned long long first = ((unsigned long long)hi << 32) | lo;
        if (mine == first) {
            unsigned long long mask = __ballot(1);
            unsigned int lane = __lane_id();
            unsigned int rank = __popcll(mask & ((1ull << lane) - 1ull));
            int base = 0;
            int leader = __ffsll((long long)mask) - 1;
            if (rank == 0) base = atomicAdd(counter, (int)__popcll(mask));
            base = __shfl(base, leader);
            result = base + (int)rank;
            done = true;
        }
    }
    return result;
#else
    return __atomic_fetch_add(counter, 1, __ATOMIC_RELAXED);
#endif
}

// Block-aggregated AllocateEntry: EVERY thread of the workgroup must call it at the same program point
// (`want` says whether this thread needs a slot).  One atomic per workgroup per call instead of one per wave:
// a single queue counter sustains only ~88 returning atomics/us, which at one atomic per wave is the
// floor of every stage (~190 us per million items).  Returns the slot, or -1 if !want.
WF_HD int BlockAlloc(int32_t *counter, bool want) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ int s_waveCount[16];
    __shared__ int s_base;
    const unsigned long long mask = __ballot(want);
    const unsigned lane = __lane_id();
    const int wave = threadIdx.x >> 6, nWaves = (blockDim.x + 63) >> 6;
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_waveCount[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < nWaves; ++w) total += s_waveCount[w];
        s_base = total > 0 ? atomicAdd(counter, total) : 0;
    }
    __syncthreads();
    int prefix = 0;
    for (int w = 0; w < wave; ++w) prefix += s_waveCount[w];
    const int slot = s_base + prefix + rank;
    __syncthreads();  // s_waveCount / s_base are reused by the next call
    return want ? slot : -1;
#else
    return want ? __atomic_fetch_add(counter, 1, __ATOMIC_RELAXED) : -1;
#endif
}

WF_HD Wavelengths LoadLambda(const WorkState &ws, int pixelIndex) {
    F4 l = ws.lambda[pixelIndex], p = ws.lambdaPdf[pixelIndex];
    Wavelengths w;
    w.lambda[0] = l.x; w.lambda[1] = l.y; w.lambda[2] = l.z; w.lambda[3] = l.w;
    w.pdf[0] = p.x; w.pdf[1] = p.y; w.pdf[2] = p.z; w.pdf[3] = p.w;
    return w;
}
WF_HD void StoreLambda(const WorkState &ws, int pixelIndex, const Wavelengths &w) {
    ws.lambda[pixelIndex] = F4{w.lambda[0], w.lambda[1], w.lambda[2], w.lambda[3]};
    ws.lambdaPdf[pixelIndex] = F4{w.pdf[0], w.pdf[1], w.pdf[2], w.pdf[3]};
}
WF_HD LightCtx LoadCtx(const RayQueueV &q, int i) {
    F4 a = q.ctx0[i], b = q.ctx1[i], c = q.ctx2[i];
    LightCtx ctx;
    ctx.pi.lo = V3{a.x, a.y, a.z};
    ctx.pi.hi = V3{a.w, b.x, b.y};
    ctx.n = N3{b.z, b.w, c.x};
    ctx.ns = N3{c.y, c.z, c.w};
    return ctx;
}
WF_HD void StoreCtx(const RayQueueV &q, int i, const LightCtx &ctx) {
    q.ctx0[i] = F4{ctx.pi.lo.x, ctx.pi.lo.y, ctx.pi.lo.z, ctx.pi.hi.x};
    q.ctx1[i] = F4{ctx.pi.hi.y, ctx.pi.hi.z, ctx.n.x, ctx.n.y};
    q.ctx2[i] = F4{ctx.n.z, ctx.ns.x, ctx.ns.y, ctx.ns.z};
}

// image scanline of local row y0 + r of this context's partition (y0 = pixel_min.y + first local row of the band)
WF_HD int BandScanline(const SceneView &sv, const WorkState &ws, int y0, int r) {
    if (ws.stripCount <= 1) return y0 + r;
    const int l = (y0 - sv.film.pixel_min[1]) + r;
    if (l >= ws.localRows) return sv.film.pixel_max[1];  // past the last owned line
    return sv.film.pixel_min[1] + (l / ws.stripHeight * ws.stripCount + ws.stripRank) * ws.stripHeight + l % ws.stripHeight;
}
// number of in-bounds pixels of the pass starting at (local) scanline y0 (row-major band)
WF_HD int KValidPixels(const SceneView &sv, const WorkState &ws, int y0) {
    const wf_film &F = sv.film;
    int xResolution = F.pixel_max[0] - F.pixel_min[0];
    int rows = (ws.stripCount <= 1 ? F.pixel_max[1] : F.pixel_min[1] + ws.localRows) - y0;
    int maxRows = ws.pixelsPerPass / xResolution;
    if (rows > maxRows) rows = maxRows;
    if (rows < 0) rows = 0;
    return rows * xResolution;
}
// camera rays of the pass: nSamples sample slots x the in-bounds pixels
// (a realistic camera appends its rays one by one: 0 to start with)
WF_HD int KCameraRayCount(const SceneView &sv, const WorkState &ws, int y0, int nSamples) { return sv.camera.type == WF_CAMERA_REALISTIC ? 0 : KValidPixels(sv, ws, y0) * nSamples; }

// ---------------------------------------------------------------------------------------------
// K2: GenerateCameraRays, wavefront/camera.cpp:35-79
// sampler dimension offsets drawn by one stage: camera rays use dims 0, 1(+2), 3, 4(+5) (samplers.h:796-814); ray
// samples use dim0 + {0, 1(+2), 3, 4(+5), 6} (samples.cpp:39-58)
WF_HD int SampleTopOffset(int k) { return k == 0 ? 0 : (k == 1 ? 1 : (k == 2 ? 3 : (k == 3 ? 4 : 6))); }
// TopDigits() of band pixel p for the five dimensions starting at dim0 -> ws.sampleTops
WF_HD void KSampleTops(const SceneView &sv, const WorkState &ws, int item, int y0, int dim0) {
    const wf_film &F = sv.film;
    const int k = item / ws.pixelsPerPass, p = item - k * ws.pixelsPerPass;
    int xResolution = F.pixel_max[0] - F.pixel_min[0];
    int px = F.pixel_min[0] + p % xResolution;
    int py = BandScanline(sv, ws, y0, p / xResolution);
    if (py >= F.pixel_max[1]) return;
    ZSobol sampler(sv);
    sampler.StartPixelSample(px, py, 0, dim0 + SampleTopOffset(k));
    ws.sampleTops[item] = (uint32_t)sampler.TopDigits();
}

template <bool ANIM = true>
WF_HD void KGenerateCameraRay(const SceneView &sv, const WorkState &ws, int pixelIndex, int y0, int sampleBase, int sampleStep, int nSamples,
                              bool useTops = false) {
    // pixelIndex = item index: sample slot s = pixelIndex / pixelsPerPass, band pixel p = pixelIndex % pixelsPerPass
    const wf_film &F = sv.film;
    int xResolution = F.pixel_max[0] - F.pixel_min[0];
    int slot, p;
    if (ws.slotStride > 0) { p = pixelIndex / ws.slotStride; slot = pixelIndex - p * ws.slotStride; }
    else { slot = pixelIndex / ws.pixelsPerPass; p = pixelIndex - slot * ws.pixelsPerPass; }
    if (p >= ws.pixelsPerPass) { ws.pPixel[pixelIndex] = I2{F.pixel_min[0], F.pixel_max[1]}; return; }   // (pixel-major: items past the pass's pixels x slots)
    const int sampleIndex = sampleBase + slot * sampleStep;
    int px = F.pixel_min[0] + p % xResolution;
    int py = BandScanline(sv, ws, y0, p / xResolution);
    if (slot >= nSamples) py = F.pixel_max[1];  // unused sample slot of a short last batch: mark out of bounds
    ws.pPixel[pixelIndex] = I2{px, py};
    if (!(px >= F.pixel_min[0] && px < F.pixel_max[0] && py >= F.pixel_min[1] && py < F.pixel_max[1])) return;
    PixelSampler sampler(sv);
    sampler.StartPixelSample(px, py, sampleIndex, 0);
    const uint32_t *tops = useTops ? ws.sampleTops + p : nullptr;
    const int tstride = ws.pixelsPerPass;
    if (useTops) sampler.SetTop(tops[0]);
    float lu = sampler.Get1D();
    if (sv.options.disable_wavelength_jitter) lu = 0.5f;
    Wavelengths lambda = F.type == WF_FILM_SPECTRAL ? SampleUniformWavelengths(lu, F.lambda_min, F.lambda_max) : SampleVisible(lu);   // Film::SampleWavelengths
    // GetCameraSample, samplers.h:796-814
    if (useTops) sampler.SetTop(tops[tstride]);
    FilterSampleR fs = FilterSample(sv, sampler.GetPixel2D());
    V2 pFilm{px + fs.p.x + 0.5f, py + fs.p.y + 0.5f};
    if (useTops) sampler.SetTop(tops[2 * tstride]);
    float time = sampler.Get1D();
    if (useTops) sampler.SetTop(tops[3 * tstride]);
    V2 pLens = sampler.Get2D();
    float filterWeight = fs.weight;
    if (sv.options.disable_pixel_jitter) {
        pFilm = V2{px + 0.5f, py + 0.5f};
        time = 0.5f;
        pLens = V2{0.5f, 0.5f};
        filterWeight = 1;
    }
    CameraRayR cr = GenerateCameraRay<ANIM>(sv, pFilm, time, pLens);
    ws.L[pixelIndex] = F4{0, 0, 0, 0};
    if (F.type == WF_FILM_GBUFFER) ws.vsP[pixelIndex] = F4{0, 0, 0, 0};   // visibleSurface = VisibleSurface() (wavefront/camera.cpp:70-71)
    StoreLambda(ws, pixelIndex, lambda);
    ws.filterWeight[pixelIndex] = filterWeight;
    if (cr.valid) {
        // RayQueue::PushCameraRay, workitems.h:346-361.  Both projective cameras always produce a ray and
        // the in-bounds pixels of a band are exactly p < rows*width, so the queue slot is analytic (pixel
        // order within each sample slot, no atomic); KCameraRayCount sets the queue size.
        // A realistic camera's rays can be blocked by the lens system: its rays are appended (the counter starts the pass at 0).
        const RayQueueV &q = ws.rq[0];
        int index = sv.camera.type == WF_CAMERA_REALISTIC ? QueueAlloc(&ws.counters[(CNT_RAY0) * CNT_STRIDE])
                    : (ws.slotStride > 0 ? p * ws.slotStride + slot : slot * KValidPixels(sv, ws, y0) + p);
        q.o[index] = F4{cr.o.x, cr.o.y, cr.o.z, cr.time};
        q.d[index] = F4{cr.d.x, cr.d.y, cr.d.z, 1.f};
        q.beta[index] = F4{1, 1, 1, 1};
        q.r_u[index] = F4{1, 1, 1, 1};
        q.r_l[index] = F4{1, 1, 1, 1};
        q.meta[index] = I4{pixelIndex, 0, 0, sv.camera.medium};
        ws.cameraRayWeight[pixelIndex] = F4{cr.weight, cr.weight, cr.weight, cr.weight};
        if (ws.pathTime) ws.pathTime[pixelIndex] = cr.time;
    } else ws.cameraRayWeight[pixelIndex] = F4{0, 0, 0, 0};
}

// K3: GenerateRaySamples, wavefront/samples.cpp:35-65 (no subsurface: dimension = 6 + 7*depth)
// topsDepth >= 0: ws.sampleTops holds the tops of dimension 6 + 7 * topsDepth (rays at another depth — re-pushed
// through interface materials — take the generic path)
WF_HD void KGenerateRaySamples(const SceneView &sv, const WorkState &ws, int cur, int i, int sampleBase, int sampleStep, int topsDepth = -1) {
    I4 m = ws.rq[cur].meta[i];
    int pixelIndex = m.x, depth = m.y;
    int slot, pPass;
    if (ws.slotStride > 0) { pPass = pixelIndex / ws.slotStride; slot = pixelIndex - pPass * ws.slotStride; }
    else { slot = pixelIndex / ws.pixelsPerPass; pPass = pixelIndex - slot * ws.pixelsPerPass; }
    const int sampleIndex = sampleBase + slot * sampleStep;
    int dimension = 6 + 7 * depth;
    if (sv.haveSubsurface) dimension += 3 * depth;  // samples.cpp:40-41
    PixelSampler sampler(sv);
    I2 pp = ws.pPixel[pixelIndex];
    sampler.StartPixelSample(pp.x, pp.y, sampleIndex, dimension);
    const bool useTops = depth == topsDepth;
    const uint32_t *tops = useTops ? ws.sampleTops + pPass : nullptr;
    const int tstride = ws.pixelsPerPass;
    if (useTops) sampler.SetTop(tops[0]);
    float duc = sampler.Get1D();
    if (useTops) sampler.SetTop(tops[tstride]);
    V2 du = sampler.Get2D();
    if (useTops) sampler.SetTop(tops[2 * tstride]);
    float iuc = sampler.Get1D();
    if (useTops) sampler.SetTop(tops[3 * tstride]);
    V2 iu = sampler.Get2D();
    if (useTops) sampler.SetTop(tops[4 * tstride]);
    float rr = sampler.Get1D();
    ws.samples0[pixelIndex] = F4{duc, du.x, du.y, iuc};
    ws.samples1[pixelIndex] = F4{iu.x, iu.y, rr, 0.f};
    if (sv.haveSubsurface) {  // samples.cpp:57-61 (never with sample tops: the host turns them off)
        float suc = sampler.Get1D();
        V2 su = sampler.Get2D();
        ws.samples2[pixelIndex] = F4{suc, su.x, su.y, 0.f};
    }
}

// ---------------------------------------------------------------------------------------------
// K4 tail: EnqueueWorkAfterMiss / EnqueueWorkAfterIntersection (wavefront/intersect.h:16-29,48-156) for
// surfaces without media.  `next` = index of the queue that receives rays re-spawned through "interface"
// surfaces.
// GetMedium(w) of a surface interaction (interaction.h:117-121, 218-229): the primitive's MediumInterface when it
// is a medium transition, else the medium of the ray that found the surface
WF_HD int SurfaceMedium(const wf_mesh &mesh, N3 n, V3 w, int rayMedium) {
    if (mesh.medium_inside != mesh.medium_outside) return Dot(w, n) > 0 ? mesh.medium_outside : mesh.medium_inside;
    return rayMedium;
}
// The MixMaterial resolve loop of EnqueueWorkAfterIntersection (intersect.h:92-97) + MixMaterial::ChooseMaterial
// (materials.h:284-294).  The reference hashes (p, wo, the two Material tagged POINTERS): its choice depends on heap
// addresses and differs from run to run, so this is the one place where parity with it is statistical by construction;
// here the two material ids take the pointers' place.
WF_HD int HitInst(const SceneView &sv, const WorkState &ws, int i) { return sv.nInstances > 0 ? ws.hitInst[i] : -1; }
WF_HD int ResolveMix(const SceneView &sv, int matId, int prim, int inst, float b0, float b1, float b2, V3 wo, V3 ro) {
    if (sv.materials[matId].type != WF_MAT_MIX) return matId;
    SurfIntr si;
    HitInteraction(sv, prim, inst, b0, b1, b2, &si, ro, -wo);
    TexCtx tc;
    tc.p = si.pi.mid(); tc.n = si.n; tc.uv = si.uv;
    while (sv.materials[matId].type == WF_MAT_MIX) {
        const wf_material &m = sv.materials[matId];
        float amt = EvalFloatTexture(sv, m.tex[WF_MT_AMOUNT], tc);
        int pick;
        if (amt <= 0) pick = 0;
        else if (amt >= 1) pick = 1;
        else {
            uint32_t w[8] = {FloatToBits(tc.p.x), FloatToBits(tc.p.y), FloatToBits(tc.p.z), FloatToBits(wo.x), FloatToBits(wo.y), FloatToBits(wo.z),
                             (uint32_t)m.mix[0], (uint32_t)m.mix[1]};
            float u = HashToFloat(HashWords(w, 8));
            pick = (amt < u) ? 0 : 1;
        }
        matId = m.mix[pick];
    }
    return matId;
}
// routing of a surface hit whose record is already in ws.hit[i] (beta, r_u, r_l are read from the ray slot by the
// consumers): interface re-push / area light / material queue
// fromMedium: the caller is the medium stage (SampleMediumInteraction's copy of this routing, media.cpp:176-201).  There the reference
// continues a ray through an interface surface from `Interaction intr(w.pi, w.n)` — an interaction whose time is the member's default, 0 —
// while EnqueueWorkAfterIntersection (intersect.h:99-106) spawns it from the full SurfaceInteraction, which carries the ray's time.  The
// time of a ray only matters to AnimatedPrimitives (and to the shadow rays' path time): fuzz finding s1800074, round 5.
WF_HD void RouteSurfaceHit(const SceneView &sv, const WorkState &ws, int cur, int i, int prim, int inst, float b0, float b1, float b2, bool fromMedium = false) {
    const wf_mesh mesh = sv.meshes[sv.triMesh[prim]];
    if (mesh.material < 0) {
        // "interface" material: the ray continues in the same direction at the same depth (intersect.h:93-101)
        const RayQueueV &q = ws.rq[cur];
        const RayQueueV &nq = ws.rq[cur ^ 1];
        SurfIntr si;
        F4 o = q.o[i], d = q.d[i];
        V3 rd{d.x, d.y, d.z};
        HitInteraction(sv, prim, inst, b0, b1, b2, &si, V3{o.x, o.y, o.z}, rd);
        V3 no = OffsetRayOrigin(si.pi, si.n, rd);
        int slot = QueueAlloc(&ws.counters[(CNT_RAY0 + (cur ^ 1)) * CNT_STRIDE]);
        nq.o[slot] = F4{no.x, no.y, no.z, fromMedium ? 0.f : o.w};
        nq.d[slot] = d;
        nq.beta[slot] = q.beta[i];
        nq.r_u[slot] = q.r_u[i];
        nq.r_l[slot] = q.r_l[i];
        nq.ctx0[slot] = q.ctx0[i];
        nq.ctx1[slot] = q.ctx1[i];
        nq.ctx2[slot] = q.ctx2[i];
        I4 m = q.meta[i];
        m.w = SurfaceMedium(mesh, si.n, rd, m.w);
        nq.meta[slot] = m;
        return;
    }
    if (mesh.first_light >= 0) {
        int slot = QueueAlloc(&ws.counters[(CNT_HITLIGHT) * CNT_STRIDE]);
        ws.hitLightQ[slot] = i;
    }
    int matId = mesh.material;
    if (sv.haveMix && sv.materials[matId].type == WF_MAT_MIX) {
        F4 d = ws.rq[cur].d[i], o = ws.rq[cur].o[i];
        matId = ResolveMix(sv, matId, prim, inst, b0, b1, b2, V3{-d.x, -d.y, -d.z}, V3{o.x, o.y, o.z});
        ws.mixMat[i] = matId;
    }
    int mtype = sv.materials[matId].type;
    int slot = QueueAlloc(&ws.counters[(CNT_MAT0 + mtype) * CNT_STRIDE]);
    ws.matQ[mtype][slot] = i;
}
#if !defined(__HIPCC__)
// CPU checker only (wf_cpu --emulate-stale-medium-depth, sequential execution): the reference's MediumSampleQueue::Push(RayWorkItem, tMax)
// — the push of a ray that MISSED every surface, intersect.h:19-23 — writes every member of the item except `depth`
// (wavefront/workitems.h:468-492), so SampleMediumInteraction reads the depth of whatever item used that queue slot last.  Which slot a
// ray gets depends on the order the threads push in: the reference's image is order-dependent there (not a parity target).
// In sequential order the effect can be reproduced, which is how the diagnosis was checked: one stale depth per medium-sample slot.
inline int32_t *g_msStaleDepth = nullptr;
#endif
WF_HD void KAfterClosestHit(const SceneView &sv, const WorkState &ws, int cur, int i, bool found, int prim, int inst, float tHit, float b0, float b1, float b2) {
    if (sv.nInstances > 0) ws.hitInst[i] = found ? inst : -1;
    if (sv.haveMedia && ws.rq[cur].meta[i].w >= 0) {
        // ray.medium set: the medium is sampled first, up to the surface or to infinity (intersect.h:16-29,53-88)
        ws.hit[i] = F4{BitsToFloat((uint32_t)(found ? prim : -1)), b0, b1, b2};
        ws.hitT[i] = found ? tHit : WF_INFINITY;
        int slot = QueueAlloc(&ws.counters[(CNT_MEDIUM_SAMPLE) * CNT_STRIDE]);
        ws.mediumSampleQ[slot] = i;
#if !defined(__HIPCC__)
        if (g_msStaleDepth) {
            if (found) g_msStaleDepth[slot] = ws.rq[cur].meta[i].y;
            else ws.rq[cur].meta[i].y = g_msStaleDepth[slot];   // the item's consumers read the depth from the ray slot
        }
#endif
        return;
    }
    if (!found) {
        if (sv.nInfiniteLights > 0) {
            int slot = QueueAlloc(&ws.counters[(CNT_ESCAPED) * CNT_STRIDE]);
            ws.escapedQ[slot] = i;
        }
        return;
    }
    ws.hit[i] = F4{BitsToFloat((uint32_t)prim), b0, b1, b2};
    RouteSurfaceHit(sv, ws, cur, i, prim, inst, b0, b1, b2);
}

// the follow-up of the HIP traversal kernel for hits on a MixMaterial (ws.mixQ): resolve, then the material queue
WF_HD void KResolveMix(const SceneView &sv, const WorkState &ws, int cur, int qi) {
    const int i = ws.mixQ[qi];
    F4 h = ws.hit[i], d = ws.rq[cur].d[i], o = ws.rq[cur].o[i];
    int prim = (int)FloatToBits(h.x);
    const int inst = HitInst(sv, ws, i);
    int matId = ResolveMix(sv, sv.meshes[sv.triMesh[prim]].material, prim, inst, h.y, h.z, h.w, V3{-d.x, -d.y, -d.z}, V3{o.x, o.y, o.z});
    ws.mixMat[i] = matId;
    int slot = QueueAlloc(&ws.counters[(CNT_MAT0 + sv.materials[matId].type) * CNT_STRIDE]);
    ws.matQ[sv.materials[matId].type][slot] = i;
}

#if defined(__HIPCC__)
// The routing of a whole workgroup's batch in ONE allocation round (two barriers instead of three per
// destination queue).  `route` = the triangle's build-time routing code (LeafTri.c.w): material type |
// emissive << 4 | interface << 5, so nothing is gathered per hit.  Destinations: 0 escaped, 1 emitter hit,
// 2 re-pushed ray (interface material), 2 + t material type t, MS medium sample (rays travelling in a medium).
// GENERAL: the scene has non-triangle primitives (the plain traversal variant must not even link the quadric code:
// an out-of-line callee's register count becomes the kernel's and costs it a wave of occupancy).
template <bool GENERAL>
__device__ inline void KRouteHitBlock(const SceneView &sv, const WorkState &ws, int cur, int i, bool valid, int prim, uint32_t route,
                                      float tHit, float b0, float b1, float b2, int inst = -1) {
    constexpr int MS = 2 + WF_MAT_NTYPES;
    constexpr int MIXQ = MS + 1;
    constexpr int NID = MIXQ + 1;
    __shared__ int s_cnt[NID][16];
    __shared__ int s_base[NID];
    const bool found = valid && prim >= 0;
    if (valid && sv.nInstances > 0) ws.hitInst[i] = found ? inst : -1;
    unsigned dest = 0;
    const bool inMedium = sv.haveMedia && valid && ws.rq[cur].meta[i].w >= 0;
    if (inMedium) {
        ws.hit[i] = F4{BitsToFloat((uint32_t)(found ? prim : -1)), b0, b1, b2};
        ws.hitT[i] = found ? tHit : WF_INFINITY;
        dest = 1u << MS;
    } else {
        if (valid && !found && sv.nInfiniteLights > 0) dest = 1u;
        if (found) {
            ws.hit[i] = F4{BitsToFloat((uint32_t)prim), b0, b1, b2};
            if (route & 16u) dest |= 2u;
            if ((route & 32u) && sv.haveMedia) dest |= 4u;
            // a MixMaterial hit goes to its own queue; KResolveMix turns it into a material-queue entry (keeps the
            // texture evaluation out of the traversal kernel's register budget)
            if ((route & 15u) == WF_MAT_MIX) dest |= 1u << MIXQ;
            else if (route & 15u) dest |= 4u << (route & 15u);
        }
    }
    const unsigned active = (sv.nInfiniteLights > 0 ? 1u : 0u) | 2u | (sv.haveMedia ? (4u | (1u << MS)) : 0u) | (((unsigned)sv.matTypeMask & ((1u << WF_MAT_NTYPES) - 2u)) << 2) |
                            (sv.haveMix ? 1u << MIXQ : 0u);
    const unsigned lane = __lane_id();
    const int wave = threadIdx.x >> 6, nWaves = (blockDim.x + 63) >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    auto counterOf = [&](int id) {
        int c = id == 0 ? CNT_ESCAPED : (id == 1 ? CNT_HITLIGHT : (id == 2 ? CNT_RAY0 + (cur ^ 1) : (id == MS ? CNT_MEDIUM_SAMPLE : (id == MIXQ ? CNT_MIX : CNT_MAT0 + (id - 2)))));
        return &ws.counters[c * CNT_STRIDE];
    };
    for (unsigned m = active; m; m &= m - 1) {
        const int id = __builtin_ctz(m);
        const unsigned long long mask = __ballot((dest >> id) & 1u);
        if (lane == 0) s_cnt[id][wave] = __popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x < NID && ((active >> threadIdx.x) & 1u)) {
        int total = 0;
        for (int w = 0; w < nWaves; ++w) total += s_cnt[threadIdx.x][w];
        s_base[threadIdx.x] = total > 0 ? atomicAdd(counterOf(threadIdx.x), total) : 0;
    }
    __syncthreads();
    for (unsigned m = active; m; m &= m - 1) {
        const int id = __builtin_ctz(m);
        const bool mine = (dest >> id) & 1u;
        const unsigned long long mask = __ballot(mine);
        if (!mine) continue;
        int slot = s_base[id] + __popcll(mask & below);
        for (int w = 0; w < wave; ++w) slot += s_cnt[id][w];
        if (id == 0) ws.escapedQ[slot] = i;
        else if (id == 1) ws.hitLightQ[slot] = i;
        else if (id == 2) {
            // "interface" material: the ray continues in the same direction at the same depth (intersect.h:93-101)
            const RayQueueV &q = ws.rq[cur];
            const RayQueueV &nq = ws.rq[cur ^ 1];
            SurfIntr si;
            F4 o = q.o[i], d = q.d[i];
            if constexpr (GENERAL) HitInteraction(sv, prim, inst, b0, b1, b2, &si, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z});
            else TriangleInteraction(sv, prim, b0, b1, b2, &si);
            V3 no = OffsetRayOrigin(si.pi, si.n, V3{d.x, d.y, d.z});
            nq.o[slot] = F4{no.x, no.y, no.z, o.w};
            nq.d[slot] = d;
            nq.beta[slot] = q.beta[i];
            nq.r_u[slot] = q.r_u[i];
            nq.r_l[slot] = q.r_l[i];
            nq.ctx0[slot] = q.ctx0[i];
            nq.ctx1[slot] = q.ctx1[i];
            nq.ctx2[slot] = q.ctx2[i];
            I4 m = q.meta[i];
            m.w = SurfaceMedium(sv.meshes[sv.triMesh[prim]], si.n, V3{d.x, d.y, d.z}, m.w);
            nq.meta[slot] = m;
        } else if (id == MS) ws.mediumSampleQ[slot] = i;
        else if (id == MIXQ) ws.mixQ[slot] = i;
        else ws.matQ[id - 2][slot] = i;
    }
    __syncthreads();  // s_cnt / s_base are reused by the next batch
}
#endif

// ---------------------------------------------------------------------------------------------
// K5: SampleMediumInteraction, wavefront/media.cpp:22-257.  The MediumSampleWorkItem is the ray slot i of the
// current queue + its hit record; beta / r_u / r_l are written back into the slot, so the consumers that follow
// (escaped, emitter hit, material evaluation, medium scattering) see the medium-attenuated values.
// Round 6: the delta-tracking loop as an explicit state (MediumTrack) with three pieces — Begin (load the item, seed the RNG, set up the
// majorant iterator), Step (at most one majorant segment fetched and one tentative collision taken; false = the item is finished) and End
// (write-back and routing) — so that the HIP kernel can run it with WAVE-LEVEL REFILL (k_medium_sample, wf_backend.hip: a wave whose items
// are mostly finished retires them and takes new ones instead of idling until the longest walk of its 64 ends; the lever the traversal
// kernels use).  SampleT_maj's two nested loops (media.h:724-800) are the same sequence of operations per item: the
// outer loop's "next segment" and the inner loop's "next exponential step" are the two halves of Step.  The CPU checker and the
// reference-order path run Begin; while (Step); End — the same code.
#ifndef WF_MEDIUM_SPEC
#define WF_MEDIUM_SPEC 0   // measured and LEFT OFF (round 6, cloud scene, 16 spp, same box, profiles/r06_medium_two_collisions_per_step_ab_cloud16.txt): 46.3 against 36.4 ms
#endif
struct MediumTrack {
    int i;               // ray slot (-1: the lane holds no item)
    int pixelIndex, depth, medium;
    V3 o, d;             // origin, NORMALISED direction (SampleT_maj normalises first)
    float tMaxHit;       // the ray's hit distance as recorded (ws.hitT: in units of the unnormalised direction)
    MediumAtLambda ml;
    MajorantIter iter;
    MajorantSeg seg;
    float tMin;
    bool inSeg, scattered, stopped;   // stopped: the callback ended the walk (SampleT_maj then returns 1, not the running T_maj)
    bool pushScatter;                 // a MediumScatterWorkItem is owed (pushed in MediumTrackEnd)
    S4 T_maj, beta, r_u, r_l, L;
    RNG rng;
    float u, uMode;
};
template <bool LEAN = false>
WF_HD void MediumTrackBegin(const SceneView &sv, const WorkState &ws, int cur, int qi, MediumTrack &s) {
    const int i = ws.mediumSampleQ[qi];
    const RayQueueV &q = ws.rq[cur];
    F4 o4 = q.o[i], d4 = q.d[i];
    I4 meta = q.meta[i];
    s.i = i;
    s.pixelIndex = meta.x; s.depth = meta.y; s.medium = meta.w;
    V3 ro{o4.x, o4.y, o4.z}, rd{d4.x, d4.y, d4.z};
    float tMax = ws.hitT[i];
    s.tMaxHit = tMax;
    Wavelengths lambda = LoadLambda(ws, s.pixelIndex);
    s.beta = toS4(q.beta[i]); s.r_u = toS4(q.r_u[i]); s.r_l = toS4(q.r_l[i]);
    s.L = S4c(0.f);
    s.rng = RNG(Hash3f1(ro, tMax), Hash3f(rd));
    s.scattered = false;
    s.stopped = false;
    s.pushScatter = false;
    s.u = s.rng.UniformFloat();
    s.uMode = s.rng.UniformFloat();
    // SampleT_maj's preamble (media.h:728-741)
    const wf_medium &M = sv.media[s.medium];
    tMax *= Length(rd);
    s.o = ro;
    s.d = Normalize(rd);
    s.ml = MediumSpectra(sv, M, lambda);
    s.iter = MediumSampleRay<LEAN>(sv, M, s.ml, s.o, s.d, tMax);
    s.T_maj = S4c(1.f);
    s.inSeg = false;
    s.tMin = 0;
}
// the callback of SampleMediumInteraction (wavefront/media.cpp:60-140) at a tentative collision; false = the walk ends here
WF_HD bool MediumTrackEvent(const SceneView &sv, const WorkState &ws, int cur, MediumTrack &s, V3 p, const MediumProps &mp, S4 sigma_maj, S4 T_maj) {
    // emission, scaled by sigma_a / sigma_maj at every event (media.cpp:72-83)
    if (s.depth < sv.maxDepth && mp.Le) {
        float pr = sigma_maj[0] * T_maj[0];
        S4 r_e = s.r_u * sigma_maj * T_maj / pr;
        if (r_e) s.L = s.L + s.beta * mp.sigma_a * T_maj * mp.Le / (pr * r_e.Average());
    }
    float pAbsorb = mp.sigma_a[0] / sigma_maj[0];
    float pScatter = mp.sigma_s[0] / sigma_maj[0];
    float pNull = fmax(0.f, 1 - pAbsorb - pScatter);
    const float w3[3] = {pAbsorb, pScatter, pNull};
    int mode = SampleDiscreteN(w3, 3, s.uMode);
    if (mode == 0) {
        s.beta = S4c(0.f);
        return false;
    } else if (mode == 1) {
        float pr = T_maj[0] * mp.sigma_s[0];
        s.beta = s.beta * (T_maj * mp.sigma_s / pr);
        s.r_u = s.r_u * (T_maj * mp.sigma_s / pr);
        if (s.beta && s.r_u) {
            // MediumScatterWorkItem push (media.cpp:104-113)
            const RayQueueV &q = ws.rq[cur];
            q.beta[s.i] = toF4(s.beta);
            q.r_u[s.i] = toF4(s.r_u);
            ws.scatterP[s.i] = F4{p.x, p.y, p.z, mp.g};
            // (the queue slot is taken in MediumTrackEnd, where the wave's lanes meet again: here every lane arrives in its own iteration of
            //  the walk and the wave-aggregated QueueAlloc degenerates into one returning atomic per lane or two — the delta-tracking
            //  kernel was bound by the scatter counter's atomic rate, which is why neither occupancy nor the density table's layout moved it)
            s.pushScatter = true;
        }
        s.scattered = true;
        return false;
    } else {
        S4 sigma_n = ClampZero(sigma_maj - mp.sigma_a - mp.sigma_s);
        float pr = T_maj[0] * sigma_n[0];
        s.beta = s.beta * (T_maj * sigma_n / pr);
        if (pr == 0) s.beta = S4c(0.f);
        s.r_u = s.r_u * (T_maj * sigma_n / pr);
        s.r_l = s.r_l * (T_maj * sigma_maj / pr);
        s.uMode = s.rng.UniformFloat();
        return bool(s.beta) && bool(s.r_u);
    }
}
template <bool LEAN = false>
WF_HD bool MediumTrackStep(const SceneView &sv, const WorkState &ws, int cur, MediumTrack &s) {
    if (!s.inSeg) {
        // the outer loop of SampleT_maj: the next majorant segment
        if (!s.iter.Next(&s.seg)) return false;
        if (s.seg.sigma_maj[0] == 0) {
            float dt = s.seg.tMax - s.seg.tMin;
            if (IsInf(dt)) dt = WF_FLT_MAX;
            s.T_maj = s.T_maj * FastExp(-dt * s.seg.sigma_maj);
            return true;
        }
        s.tMin = s.seg.tMin;
        s.inSeg = true;
    }
    // the inner loop: one exponential step inside the segment
    float t = s.tMin + SampleExponential(s.u, s.seg.sigma_maj[0]);
    s.u = s.rng.UniformFloat();
    if (t < s.seg.tMax) {
        s.T_maj = s.T_maj * FastExp(-(t - s.tMin) * s.seg.sigma_maj);
        V3 p = s.o + s.d * t;
        const wf_medium &M = sv.media[s.medium];
#if WF_MEDIUM_SPEC
        // TWO tentative collisions per step for the grid media (round 6; bit-identical, but slower: the walks of the dense cloud are short — one or two
        // events — so most second lookups are wasted, and the step's code doubles).  Where the NEXT exponential step lands depends only on this one's
        // position and on the sample value already drawn (s.u) — not on the density here — so its position is known now, and the density
        // gathers of both points (two dependent-free pairs of 16-byte loads out of a multi-GB table: what this kernel waits for 79 % of its
        // time at two waves per SIMD) are issued together.  If this event ends the walk the second lookup is wasted (once per item);
        // otherwise the second event runs exactly as the next step would: the same operands, the RNG draws in the same order.
        const bool gridMedium = M.type == WF_MEDIUM_GRID || M.type == WF_MEDIUM_RGB_GRID || M.type == WF_MEDIUM_NANOVDB;
        const float t2 = t + SampleExponential(s.u, s.seg.sigma_maj[0]);   // (the next step's `tMin + SampleExponential(u, ...)` with tMin = t)
        const bool spec = gridMedium && t2 < s.seg.tMax;
        MediumProps mp = MediumSamplePoint<LEAN>(sv, M, s.ml, p);
        V3 p2 = p;
        MediumProps mp2 = mp;
        if (spec) { p2 = s.o + s.d * t2; mp2 = MediumSamplePoint<LEAN>(sv, M, s.ml, p2); }
#else
        MediumProps mp = MediumSamplePoint<LEAN>(sv, M, s.ml, p);
#endif
        if (!MediumTrackEvent(sv, ws, cur, s, p, mp, s.seg.sigma_maj, s.T_maj)) {
            s.stopped = true;
            return false;
        }
        s.T_maj = S4c(1.f);
        s.tMin = t;
#if WF_MEDIUM_SPEC
        if (spec) {
            s.u = s.rng.UniformFloat();
            s.T_maj = s.T_maj * FastExp(-(t2 - s.tMin) * s.seg.sigma_maj);
            if (!MediumTrackEvent(sv, ws, cur, s, p2, mp2, s.seg.sigma_maj, s.T_maj)) {
                s.stopped = true;
                return false;
            }
            s.T_maj = S4c(1.f);
            s.tMin = t2;
        }
#endif
    } else {
        float dt = s.seg.tMax - s.tMin;
        if (IsInf(dt)) dt = WF_FLT_MAX;
        s.T_maj = s.T_maj * FastExp(-dt * s.seg.sigma_maj);
        s.inSeg = false;
    }
    return true;
}
WF_HD void MediumTrackEnd(const SceneView &sv, const WorkState &ws, int cur, MediumTrack &s) {
    const int i = s.i;
    const RayQueueV &q = ws.rq[cur];
    const S4 T_maj = s.stopped ? S4c(1.f) : s.T_maj;
    S4 beta = s.beta, r_u = s.r_u, r_l = s.r_l;
    if (!s.scattered && beta) {
        beta = beta * (T_maj / T_maj[0]);
        r_u = r_u * (T_maj / T_maj[0]);
        r_l = r_l * (T_maj / T_maj[0]);
    }
    if (s.L) ws.L[s.pixelIndex] = toF4(toS4(ws.L[s.pixelIndex]) + s.L);
    if (s.pushScatter) {   // MediumScatterWorkItem push (media.cpp:104-113)
        int slot = QueueAlloc(&ws.counters[(CNT_MEDIUM_SCATTER) * CNT_STRIDE]);
        ws.mediumScatterQ[slot] = i;
    }
    if (s.scattered || !beta || !r_u || s.depth == sv.maxDepth) return;
    // the ray reached the surface (or left the scene): route it as EnqueueWorkAfterIntersection would have
    q.beta[i] = toF4(beta);
    q.r_u[i] = toF4(r_u);
    q.r_l[i] = toF4(r_l);
    if (IsInf(s.tMaxHit)) {
        if (sv.nInfiniteLights > 0) {
            int slot = QueueAlloc(&ws.counters[(CNT_ESCAPED) * CNT_STRIDE]);
            ws.escapedQ[slot] = i;
        }
        return;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // the HIP back end routes the rays that reached their surface in a kernel of its own (KMediumRoute over ws.mediumRouteQ): the routing
    // rebuilds the interaction of interface hits and resolves MixMaterials — code whose registers the delta-tracking loop above should
    // not pay for (k_medium_sample: 219 -> 128 VGPRs, round 5)
    {
        int slot = QueueAlloc(&ws.counters[(CNT_MEDIUM_ROUTE) * CNT_STRIDE]);
        ws.mediumRouteQ[slot] = i;
    }
#else
    const F4 h = ws.hit[i];
    RouteSurfaceHit(sv, ws, cur, i, (int)FloatToBits(h.x), HitInst(sv, ws, i), h.y, h.z, h.w, /* fromMedium */ true);
#endif
}
template <bool LEAN = false>
WF_HD void KSampleMediumInteraction(const SceneView &sv, const WorkState &ws, int cur, int qi) {
    MediumTrack s;
    MediumTrackBegin<LEAN>(sv, ws, cur, qi, s);
    while (MediumTrackStep<LEAN>(sv, ws, cur, s)) {}
    MediumTrackEnd(sv, ws, cur, s);
}
// the second half of K5 on the HIP back end: EnqueueWorkAfterIntersection for the medium-sample items that reached their surface
WF_HD void KMediumRoute(const SceneView &sv, const WorkState &ws, int cur, int qi) {
    const int i = ws.mediumRouteQ[qi];
    const F4 h = ws.hit[i];
    RouteSurfaceHit(sv, ws, cur, i, (int)FloatToBits(h.x), HitInst(sv, ws, i), h.y, h.z, h.w, /* fromMedium */ true);
}

// K6: SampleMediumScattering<HGPhaseFunction>, wavefront/media.cpp:259-352
// RARE_LIGHTS = false: the scene has neither portal lights nor emitters that are not triangles nor alpha-masked emitters (LightSampleLi<RARE>)
template <bool RARE_LIGHTS = true>
WF_HD void KSampleMediumScattering(const SceneView &sv, const WorkState &ws, int cur, int qi) {
    const int i = ws.mediumScatterQ[qi];
    const RayQueueV &q = ws.rq[cur];
    const RayQueueV &nq = ws.rq[cur ^ 1];
    F4 o4 = q.o[i], d4 = q.d[i], sp = ws.scatterP[i];
    I4 meta = q.meta[i];
    const int pixelIndex = meta.x, depth = meta.y, medium = meta.w;
    const float time = o4.w, etaScale = d4.w, g = sp.w;
    const V3 p{sp.x, sp.y, sp.z};
    const V3 wo{-d4.x, -d4.y, -d4.z};
    Wavelengths lambda = LoadLambda(ws, pixelIndex);
    const S4 wbeta = toS4(q.beta[i]), wr_u = toS4(q.r_u[i]);
    F4 s0 = ws.samples0[pixelIndex], s1 = ws.samples1[pixelIndex];
    LightCtx ctx{MakeP3i(p), N3{0, 0, 0}, N3{0, 0, 0}};
    // direct lighting
    float lightPMF = 0;
    int lightId = LightSamplerSample(sv, ctx, s0.x, &lightPMF);
    if (lightId >= 0) {
        const wf_light &light = sv.lights[lightId];
        LightLiSample ls = LightSampleLi<RARE_LIGHTS>(sv, light, ctx, V2{s0.y, s0.z}, lambda, true);
        if (ls.valid && ls.L && ls.pdf > 0) {
            V3 wi = ls.wi;
            S4 beta = wbeta * HenyeyGreenstein(Dot(wo, wi), g);
            float lightPDF = ls.pdf * lightPMF;
            float phasePDF = IsDeltaLight(light) ? 0.f : HenyeyGreenstein(Dot(wo, wi), g);
            S4 r_u = wr_u * phasePDF;
            S4 r_l = wr_u * lightPDF;
            S4 Ld = beta * ls.L;
            V3 sd = ls.pLightPi.mid() - p;
            int slot = QueueAlloc(&ws.counters[(CNT_SHADOW) * CNT_STRIDE]);
            ws.sq.o[slot] = F4{p.x, p.y, p.z, 1 - ShadowEpsilon};
            ws.sq.d[slot] = F4{sd.x, sd.y, sd.z, BitsToFloat((uint32_t)pixelIndex)};
            ws.sq.Ld[slot] = toF4(Ld);
            ws.sq.r_u[slot] = toF4(r_u);
            ws.sq.r_l[slot] = toF4(r_l);
            ws.sq.medium[slot] = medium;
        }
    }
    // indirect lighting
    float pdf = 0;
    V3 wi = SampleHenyeyGreenstein(wo, g, V2{s1.x, s1.y}, &pdf);
    if (pdf == 0) return;
    S4 beta = wbeta * pdf / pdf;  // phaseSample->p == phaseSample->pdf
    S4 r_u = wr_u;
    S4 r_l = wr_u / pdf;
    S4 rrBeta = beta * etaScale / r_u.Average();
    if (rrBeta.MaxComponentValue() < 1 && depth >= 1) {
        float qq = fmax(0.f, 1 - rrBeta.MaxComponentValue());
        if (s1.z < qq) return;
        beta = beta / (1 - qq);
    }
    int slot = QueueAlloc(&ws.counters[(CNT_RAY0 + (cur ^ 1)) * CNT_STRIDE]);
    nq.o[slot] = F4{p.x, p.y, p.z, time};
    nq.d[slot] = F4{wi.x, wi.y, wi.z, etaScale};
    nq.beta[slot] = toF4(beta);
    nq.r_u[slot] = toF4(r_u);
    nq.r_l[slot] = toF4(r_l);
    StoreCtx(nq, slot, ctx);
    nq.meta[slot] = I4{pixelIndex, depth + 1, RAYFLAG_ANY_NONSPECULAR, medium};
}

// K11: TraceTransmittance, wavefront/intersect.h:165-274 (the shadow-ray stage when the scene has media).
// The reference's per-ray loop — trace to the next surface, stop at an opaque one, ratio-track the medium along the segment, respawn
// behind an interface — is written as three pieces over an explicit state, so that it can run either as one loop per lane
// (KTraceTransmittance: the CPU checker, the reference-order device kernels) or as a wavefront of its own (round 3: TrBegin, then per
// segment the production closest-hit walk over the live shadow rays + TrSegment, state in HBM between the launches: the walk runs at
// its own occupancy instead of inside a 428-register loop).
struct TrState {
    V3 ro, rd;
    int medium;
    S4 T_ray, r_u, r_l;
    RNG rng;
};
WF_HD void TrBegin(const WorkState &ws, int i, TrState *st) {
    F4 o4 = ws.sq.o[i], d4 = ws.sq.d[i];
    st->ro = V3{o4.x, o4.y, o4.z};
    st->rd = V3{d4.x, d4.y, d4.z};
    st->medium = ws.sq.medium[i];
    st->rng = RNG(Hash3f(st->ro), Hash3f(st->rd));
    st->T_ray = S4c(1.f); st->r_u = S4c(1.f); st->r_l = S4c(1.f);
}
// The shadow queue's d.w: the pixel index in the low 31 bits.  Bit 31 marks the shadow ray of a subsurface exit, which the reference
// spawns at time 0 whatever the path's time ("Float time = 0;  // TODO: pipe through", wavefront/subsurface.cpp:70) — it only matters
// to scenes with animated primitives (ws.pathTime != nullptr), the only ones that set it.
constexpr uint32_t SHADOW_TIME_ZERO = 0x80000000u;
WF_HD int ShadowPixel(float dw) { return (int)(FloatToBits(dw) & ~SHADOW_TIME_ZERO); }
// ShadowRayWorkItem.ray.time: the time of the path the ray was spawned on (ws.pathTime: written with every ray the closest-hit stage takes)
template <bool ANIM>
WF_HD float ShadowTime(const WorkState &ws, float dw) {
    const uint32_t b = FloatToBits(dw);
    return (ANIM && ws.pathTime && !(b & SHADOW_TIME_ZERO)) ? ws.pathTime[b & ~SHADOW_TIME_ZERO] : 0.f;
}
// one turn of the loop body after the closest hit of (ro, rd, tMax) is known; returns whether the ray goes on (a new segment in st)
// ANIM: the scene has animated primitives (the interaction of a hit through one needs the path's time: ws.pathTime)
template <bool ANIM = false, bool MLEAN = false>
WF_HD bool TrSegment(const SceneView &sv, const WorkState &ws, int i, TrState *st, bool hit, int prim, int inst, float b0, float b1, float b2) {
    F4 o4 = ws.sq.o[i], d4 = ws.sq.d[i];
    const float tMax = o4.w;
    const V3 pLight = V3{o4.x, o4.y, o4.z} + V3{d4.x, d4.y, d4.z} * tMax;
    SurfIntr si;
    bool opaque = false;
    if (hit) {
        HitInteraction<!WF_DEV_LEAN, false, ANIM>(sv, prim, inst, b0, b1, b2, &si, st->ro, st->rd, ShadowTime<ANIM>(ws, d4.w));
        opaque = sv.meshes[si.mesh].material >= 0;
    }
    if (opaque) {
        st->T_ray = S4c(0.f);
        return false;
    }
    if (st->medium >= 0) {
        const int pixelIndex = ShadowPixel(d4.w);
        Wavelengths lambda = LoadLambda(ws, pixelIndex);
        float tEnd = !hit ? tMax : (Distance(st->ro, si.pi.mid()) / Length(st->rd));
        float u0 = st->rng.UniformFloat();
        S4 &T_ray = st->T_ray, &r_u = st->r_u, &r_l = st->r_l;
        RNG &rng = st->rng;
        S4 T_maj = SampleT_maj<MLEAN>(sv, st->medium, st->ro, st->rd, tEnd, u0, rng, lambda, [&](V3 p, const MediumProps &mp, S4 sigma_maj, S4 T_maj) {
            S4 sigma_n = ClampZero(sigma_maj - mp.sigma_a - mp.sigma_s);
            // ratio tracking: only null scattering is evaluated
            float pr = T_maj[0] * sigma_maj[0];
            T_ray = T_ray * (T_maj * sigma_n / pr);
            r_l = r_l * (T_maj * sigma_maj / pr);
            r_u = r_u * (T_maj * sigma_n / pr);
            S4 Tr = T_ray / (r_l + r_u).Average();
            if (Tr.MaxComponentValue() < 0.05f) {
                float qq = 0.75f;
                if (rng.UniformFloat() < qq) T_ray = S4c(0.f);
                else T_ray = T_ray / (1 - qq);
            }
            if (!T_ray) return false;
            return true;
        });
        T_ray = T_ray * (T_maj / T_maj[0]);
        r_l = r_l * (T_maj / T_maj[0]);
        r_u = r_u * (T_maj / T_maj[0]);
    }
    if (!hit || !st->T_ray) return false;
    // ray = si->intr.SpawnRayTo(pLight): interaction.h:103-107
    RayOD nr = SpawnRayTo(si.pi, si.n, pLight);
    st->medium = SurfaceMedium(sv.meshes[si.mesh], si.n, nr.d, st->medium);
    st->ro = nr.o;
    st->rd = nr.d;  // (tMax stays sr.tMax, intersect.h:176,185)
    return !(st->rd.x == 0 && st->rd.y == 0 && st->rd.z == 0);
}
WF_HD void TrFinish(const WorkState &ws, int i, const TrState &st) {
    if (st.T_ray) {
        F4 d4 = ws.sq.d[i];
        const int pixelIndex = ShadowPixel(d4.w);
        S4 Ld = toS4(ws.sq.Ld[i]);
        const S4 sr_u = toS4(ws.sq.r_u[i]), sr_l = toS4(ws.sq.r_l[i]);
        Ld = Ld * (st.T_ray / (sr_u * st.r_u + sr_l * st.r_l).Average());
        ws.L[pixelIndex] = toF4(toS4(ws.L[pixelIndex]) + Ld);
    }
}
WF_HD void TrStore(const WorkState &ws, int i, const TrState &st) {
    ws.trO[i] = F4{st.ro.x, st.ro.y, st.ro.z, 0.f};
    ws.trD[i] = F4{st.rd.x, st.rd.y, st.rd.z, BitsToFloat((uint32_t)st.medium)};
    ws.trT[i] = toF4(st.T_ray); ws.trRu[i] = toF4(st.r_u); ws.trRl[i] = toF4(st.r_l);
    ws.trRng[i] = I4{(int32_t)(uint32_t)st.rng.state, (int32_t)(uint32_t)(st.rng.state >> 32), (int32_t)(uint32_t)st.rng.inc, (int32_t)(uint32_t)(st.rng.inc >> 32)};
}
WF_HD void TrLoad(const WorkState &ws, int i, TrState *st) {
    F4 o = ws.trO[i], d = ws.trD[i];
    st->ro = V3{o.x, o.y, o.z}; st->rd = V3{d.x, d.y, d.z};
    st->medium = (int)FloatToBits(d.w);
    st->T_ray = toS4(ws.trT[i]); st->r_u = toS4(ws.trRu[i]); st->r_l = toS4(ws.trRl[i]);
    I4 r = ws.trRng[i];
    st->rng.state = (uint64_t)(uint32_t)r.x | ((uint64_t)(uint32_t)r.y << 32);
    st->rng.inc = (uint64_t)(uint32_t)r.z | ((uint64_t)(uint32_t)r.w << 32);
}
// trace(o, d, tMax, &prim, &inst, &b0, &b1, &b2) -> closest hit?
template <bool ANIM = false, bool MLEAN = false, typename Trace>
WF_HD void KTraceTransmittanceFrom(const SceneView &sv, const WorkState &ws, int i, TrState &st, Trace trace) {
    const float tMax = ws.sq.o[i].w;
    while (!(st.rd.x == 0 && st.rd.y == 0 && st.rd.z == 0)) {
        int prim = -1, inst = -1;
        float b0 = 0, b1 = 0, b2 = 0;
        bool hit = trace(st.ro, st.rd, tMax, &prim, &inst, &b0, &b1, &b2);
        if (!TrSegment<ANIM, MLEAN>(sv, ws, i, &st, hit, prim, inst, b0, b1, b2)) break;
    }
    TrFinish(ws, i, st);
}
template <bool ANIM = false, bool MLEAN = false, typename Trace>
WF_HD void KTraceTransmittance(const SceneView &sv, const WorkState &ws, int i, Trace trace) {
    TrState st;
    TrBegin(ws, i, &st);
    KTraceTransmittanceFrom<ANIM, MLEAN>(sv, ws, i, st, trace);
}

// K7: HandleEscapedRays, wavefront/integrator.cpp:495-537
template <bool RARE_LIGHTS = true>
WF_HD void KHandleEscaped(const SceneView &sv, const WorkState &ws, int cur, int qi) {
    int i = ws.escapedQ[qi];
    const RayQueueV &q = ws.rq[cur];
    I4 m = q.meta[i];
    int pixelIndex = m.x, depth = m.y;
    bool specularBounce = m.z & RAYFLAG_SPECULAR_BOUNCE;
    Wavelengths lambda = LoadLambda(ws, pixelIndex);
    F4 d4 = q.d[i], o4 = q.o[i];
    V3 rayd{d4.x, d4.y, d4.z}, rayo{o4.x, o4.y, o4.z};
    S4 beta = toS4(q.beta[i]), r_u = toS4(q.r_u[i]), r_l0 = toS4(q.r_l[i]);
    S4 L = S4c(0.f);
    for (int k = 0; k < sv.nInfiniteLights; ++k) {
        int lightId = sv.infiniteLights[k];
        const wf_light &light = sv.lights[lightId];
        S4 Le = LightLe<RARE_LIGHTS>(sv, light, rayo, rayd, lambda);
        if (Le) {
            if (depth == 0 || specularBounce) {
                L = L + beta * Le / r_u.Average();
            } else {
                LightCtx ctx = LoadCtx(q, i);
                float lightChoicePDF = LightSamplerPMF(sv, ctx, lightId);
                S4 r_l = r_l0 * lightChoicePDF * LightPDF_Li<false, RARE_LIGHTS>(sv, light, ctx, rayd, true);   // (infinite lights only)
                L = L + beta * Le / (r_u + r_l).Average();
            }
        }
    }
    if (L) {
        L = L + toS4(ws.L[pixelIndex]);
        ws.L[pixelIndex] = toF4(L);
    }
}

// K8: HandleEmissiveIntersection, wavefront/integrator.cpp:539-573
WF_HD void KHandleEmissive(const SceneView &sv, const WorkState &ws, int cur, int qi) {
    int i = ws.hitLightQ[qi];
    const RayQueueV &q = ws.rq[cur];
    I4 m = q.meta[i];
    int pixelIndex = m.x, depth = m.y;
    bool specularBounce = m.z & RAYFLAG_SPECULAR_BOUNCE;
    F4 h = ws.hit[i];
    int prim = (int)FloatToBits(h.x);
    SurfIntr si;
    {
        // emitters are top-level primitives (no area lights inside object instances); an emissive curve's interaction needs the ray
        V3 ro{0, 0, 0}, rd{0, 0, 0};
        if (sv.haveCurves) { F4 o4 = q.o[i], dd = q.d[i]; ro = V3{o4.x, o4.y, o4.z}; rd = V3{dd.x, dd.y, dd.z}; }
        HitInteraction(sv, prim, -1, h.y, h.z, h.w, &si, ro, rd);
    }
    const wf_mesh &mesh = sv.meshes[si.mesh];
    int lightId = mesh.first_light + (prim - mesh.first_tri);
    const wf_light &light = sv.lights[lightId];
    F4 d4 = q.d[i];
    // intr.wo is normalised by the Interaction ctor (interaction.h:40-43); items that went through the medium
    // stage carry -ray.d as it is (media.cpp:206-208)
    V3 wo{-d4.x, -d4.y, -d4.z};
    if (!(sv.haveMedia && m.w >= 0)) wo = IntrWo(sv, prim, -1, wo);
    Wavelengths lambda = LoadLambda(ws, pixelIndex);
    S4 Le = AreaLightL(sv, light, si.pi.mid(), si.n, si.uv, wo, lambda);
    if (!Le) return;
    S4 beta = toS4(q.beta[i]), r_u = toS4(q.r_u[i]);
    S4 L;
    if (depth == 0 || specularBounce) {
        L = beta * Le / r_u.Average();
    } else {
        V3 wi = -wo;
        LightCtx ctx = LoadCtx(q, i);
        float lightChoicePDF = LightSamplerPMF(sv, ctx, lightId);
        float lightPDF = lightChoicePDF * LightPDF_Li(sv, light, ctx, wi, true);
        S4 r_l = toS4(q.r_l[i]) * lightPDF;
        L = beta * Le / (r_u + r_l).Average();
    }
    L = L + toS4(ws.L[pixelIndex]);
    ws.L[pixelIndex] = toF4(L);
}

// ---------------------------------------------------------------------------------------------
// The light sample of next-event estimation (surfscatter.cpp:253-266: lightSampler.Sample, then light.SampleLi) for a whole workgroup.
// Which code a lane runs here depends on the light it draws — the sky (image infinite light: two binary searches), the sun (a handful of
// instructions), an emitter (light-BVH descent, then spherical-triangle sampling with its inverse trigonometry) — and on the spec scene
// the three are interleaved 1 : 1 : 1 over the lanes of a wave, which then executes all three in turn with a third of its lanes each.
// The first decision of the sampler (u against pInfinite, lightsamplers.h:270-283) is known up front, so the workgroup's requests are
// SORTED by it through LDS (a counting sort over eight classes: 21 floats per lane each way), every lane serves the request that lands
// in its slot — whole waves of one class — and the results travel back the same way.  Same arithmetic on the same inputs: bit-identical.
struct LightPick {
    int lightId = -1;
    float pmf = 0;
    LightLiSample ls{};
};
template <bool RARE>
WF_HD LightPick SampleLightDirect(const SceneView &sv, const LightCtx &ctx, float u0, V2 u, const Wavelengths &lambda) {
    LightPick pk;
    pk.ls.valid = false;
    pk.lightId = LightSamplerSample(sv, ctx, u0, &pk.pmf);
    if (pk.lightId >= 0) pk.ls = LightSampleLi<RARE>(sv, sv.lights[pk.lightId], ctx, u, lambda, true);
    return pk;
}
#ifndef WF_NEE_REGROUP
#define WF_NEE_REGROUP 1
#endif
template <bool RARE>
WF_HD LightPick SampleLightForBlock(const SceneView &sv, bool want, const LightCtx &ctx, float u0, V2 u, const Wavelengths &lambda) {
#if defined(__HIP_DEVICE_COMPILE__) && WF_NEE_REGROUP
    constexpr int NF = 21, NT = 256, NK = 8;   // (the material kernels run 256-thread workgroups: wf_mat.hip MBLOCK)
    __shared__ float s_x[NF][NT];
    __shared__ int s_cnt[NK][NT / 64];
    // class of the request: the branch LightSamplerSample takes first
    int key = NK - 1;   // no request
    if (want) {
        if (sv.lightSampler == WF_LS_BVH) {
            const int nInf = sv.nInfiniteLights;
            const float pInfinite = float(nInf) / float(nInf + (sv.nLightBvhNodes == 0 ? 0 : 1));
            if (u0 < pInfinite) {
                int index = (int)(u0 / pInfinite * nInf);
                if (index > nInf - 1) index = nInf - 1;
                key = index < NK - 2 ? index : NK - 3;
            } else key = NK - 2;
        } else key = 0;
    }
    const unsigned lane = __lane_id();
    const int wave = threadIdx.x >> 6;
    int rank = 0;
    for (int k = 0; k < NK; ++k) {
        const unsigned long long m = __ballot(key == k);
        if (lane == 0) s_cnt[k][wave] = __popcll(m);
        if (key == k) rank = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    int dst = rank;
    for (int k = 0; k < NK; ++k)
        for (int w = 0; w < NT / 64; ++w)
            if (k < key || (k == key && w < wave)) dst += s_cnt[k][w];
    // request -> slot dst
    {
        const float f[NF] = {ctx.pi.lo.x, ctx.pi.lo.y, ctx.pi.lo.z, ctx.pi.hi.x, ctx.pi.hi.y, ctx.pi.hi.z, ctx.n.x, ctx.n.y, ctx.n.z, ctx.ns.x, ctx.ns.y, ctx.ns.z,
                             u0, u.x, u.y, lambda.lambda[0], lambda.lambda[1], lambda.lambda[2], lambda.lambda[3], BitsToFloat((uint32_t)key), 0.f};
#pragma unroll
        for (int k = 0; k < NF; ++k) s_x[k][dst] = f[k];
    }
    __syncthreads();
    // serve the request in slot threadIdx.x
    {
        const int t = threadIdx.x;
        LightPick pk;
        pk.ls.valid = false;
        pk.ls.L = S4c(0.f); pk.ls.wi = V3{0, 0, 0}; pk.ls.pdf = 0; pk.ls.pLightPi = P3i{V3{0, 0, 0}, V3{0, 0, 0}}; pk.ls.pLightN = N3{0, 0, 0};
        if ((int)FloatToBits(s_x[19][t]) != NK - 1) {
            LightCtx c;
            c.pi.lo = V3{s_x[0][t], s_x[1][t], s_x[2][t]}; c.pi.hi = V3{s_x[3][t], s_x[4][t], s_x[5][t]};
            c.n = N3{s_x[6][t], s_x[7][t], s_x[8][t]}; c.ns = N3{s_x[9][t], s_x[10][t], s_x[11][t]};
            Wavelengths l;
            for (int k = 0; k < 4; ++k) { l.lambda[k] = s_x[15 + k][t]; l.pdf[k] = 0; }
            pk = SampleLightDirect<RARE>(sv, c, s_x[12][t], V2{s_x[13][t], s_x[14][t]}, l);
        }
        const float r[NF] = {BitsToFloat((uint32_t)pk.lightId), pk.pmf, pk.ls.L[0], pk.ls.L[1], pk.ls.L[2], pk.ls.L[3], pk.ls.wi.x, pk.ls.wi.y, pk.ls.wi.z, pk.ls.pdf,
                             pk.ls.pLightPi.lo.x, pk.ls.pLightPi.lo.y, pk.ls.pLightPi.lo.z, pk.ls.pLightPi.hi.x, pk.ls.pLightPi.hi.y, pk.ls.pLightPi.hi.z,
                             pk.ls.pLightN.x, pk.ls.pLightN.y, pk.ls.pLightN.z, pk.ls.valid ? 1.f : 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NF; ++k) s_x[k][t] = r[k];
    }
    __syncthreads();
    LightPick out;
    out.lightId = (int)FloatToBits(s_x[0][dst]);
    out.pmf = s_x[1][dst];
    out.ls.L = S4{{s_x[2][dst], s_x[3][dst], s_x[4][dst], s_x[5][dst]}};
    out.ls.wi = V3{s_x[6][dst], s_x[7][dst], s_x[8][dst]};
    out.ls.pdf = s_x[9][dst];
    out.ls.pLightPi.lo = V3{s_x[10][dst], s_x[11][dst], s_x[12][dst]}; out.ls.pLightPi.hi = V3{s_x[13][dst], s_x[14][dst], s_x[15][dst]};
    out.ls.pLightN = N3{s_x[16][dst], s_x[17][dst], s_x[18][dst]};
    out.ls.valid = s_x[19][dst] != 0.f;
    __syncthreads();   // the slots are reused by the next call
    if (!want) { out.lightId = -1; out.ls.valid = false; }
    return out;
#else
    if (!want) { LightPick pk; pk.ls.valid = false; return pk; }
    return SampleLightDirect<RARE>(sv, ctx, u0, u, lambda);
#endif
}

// ---------------------------------------------------------------------------------------------
// K9: EvaluateMaterialAndBSDF<M, BasicTextureEvaluator>, wavefront/surfscatter.cpp:57-328 — the BxDF of each material type
template <int MAT> struct MatBxDF;
template <> struct MatBxDF<WF_MAT_DIFFUSE> {
    using T = DiffuseBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetDiffuseBxDF(sv, m, l, tc); }
};
template <> struct MatBxDF<WF_MAT_CONDUCTOR> {
    using T = ConductorBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetConductorBxDF(sv, m, l, tc); }
};
template <> struct MatBxDF<WF_MAT_DIELECTRIC> {
    using T = DielectricBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetDielectricBxDF(sv, m, l, tc); }
};
template <> struct MatBxDF<WF_MAT_THIN_DIELECTRIC> {
    using T = ThinDielectricBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetThinDielectricBxDF(sv, m, l, tc); }
};
template <> struct MatBxDF<WF_MAT_DIFFUSE_TRANSMISSION> {
    using T = DiffuseTransmissionBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetDiffuseTransmissionBxDF(sv, m, l, tc); }
};

template <> struct MatBxDF<WF_MAT_SUBSURFACE> {
    using T = DielectricBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetSubsurfaceBxDF(sv, m, l, tc); }
};
template <> struct MatBxDF<WF_MAT_HAIR> {
    using T = HairBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetHairBxDF(sv, m, l, tc); }
};
template <> struct MatBxDF<WF_MAT_MEASURED> {
    using T = MeasuredBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetMeasuredBxDF(sv, m, l, tc); }
};
template <> struct MatBxDF<WF_MAT_COATED_DIFFUSE> {
    using T = CoatedDiffuseBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetCoatedDiffuseBxDF(sv, m, l, tc); }
};
template <> struct MatBxDF<WF_MAT_COATED_CONDUCTOR> {
    using T = CoatedConductorBxDF;
    WF_HD static T Get(const SceneView &sv, const wf_material &m, Wavelengths &l, const TexCtx &tc) { return GetCoatedConductorBxDF(sv, m, l, tc); }
};

// K9 runs in two halves since round 5 (the reference hands the stage a self-contained 252-byte MaterialEvalWorkItem, workitems.h:265-325;
// the state between the halves below is this build's equivalent, SURVEY 8(d) budgets for it):
//   MatShade   the interaction rebuilt from (primitive, barycentrics), normal / bump mapping, the textures, the BxDF, the BSDF sample
//              and the push of the indirect ray (surfscatter.cpp:57-249) — gather work: triangle -> vertices -> texels
//   MatNee     next-event estimation (surfscatter.cpp:251-327): the light sample of the workgroup (SampleLightForBlock), f / PDF, MIS,
//              the push of the shadow ray — arithmetic on the item's NeeItem and the light tables, nothing of the hit is touched again
// One function after the other is the fused stage (KEvalMaterial: the CPU checker); the HIP back end runs them as two kernels per material
// type with the NeeItem in HBM between them (wf_mat.hip: k_mat_shade / k_mat_nee), so that the light-BVH descents and the light sampling —
// 60 % of the fused kernel's time — run at the occupancy of THEIR register need, not of the interaction + texture + BSDF-sampling code's.
template <int MAT>
struct NeeItem {
    using BxDF = typename MatBxDF<MAT>::T;
    bool want = false;     // next-event estimation takes place: IsNonSpecular(bsdf.Flags()), on a live item
    P3i pi{V3{0, 0, 0}, V3{0, 0, 0}};   // the interaction: position interval, geometric and shading normal, shading dpdu (the BSDF's frame), wo
    N3 n{0, 0, 0}, ns{0, 0, 1};
    V3 dpdus{1, 0, 0}, wo{0, 0, 1};
    // the light sample context's reference point (surfscatter.cpp:255-259): the interaction point offset along +-n for a BSDF that
    // reflects (only / also), as a degenerate interval; ctxIsPoint = false: the interaction's own interval
    V3 ctxP{0, 0, 0};
    bool ctxIsPoint = false;
    float u0 = 0;          // raySamples.direct.uc
    V2 u{0, 0};            // raySamples.direct.u
    int pixelIndex = 0;
    int mediumInside = -1, mediumOutside = -1;   // GetMedium() of a ray leaving the surface against / along n (interaction.h:117-121)
    S4 beta = S4c(0.f), r_u = S4c(0.f);
    float lambda[4] = {0, 0, 0, 0};
    BxDF bxdf{};           // after Regularize()
};

// TEXCTX = false: the scene has neither footprint-dependent textures nor displacement, so the differentials and
// the bump-mapping block (whose only consumers those are) are compiled out — a separate kernel variant, because
// their registers cost the common case ~25 % (35 spilled VGPRs in the diffuse kernel).
// VARIANT 2 = 1 + the rarely used light types (portal infinite lights): their out-of-line samplers cost the material kernels 4-6 % by being
// reachable at all (call-site spills), so scenes without them run variants that cannot reach them.
// `shadowIdle` (the fused device kernel only): an idle lane of the last workgroup shadows the queue's last item with every side effect off, so
// that all lanes reach the workgroup-wide light sampling of MatNee with well-defined operands.
template <int MAT, int VARIANT = 2>
WF_HD void MatShade(const SceneView &sv, const WorkState &ws, int cur, int qi, bool valid, NeeItem<MAT> *out, bool shadowIdle = false) {
    constexpr bool TEXCTX = VARIANT != 0;
    // animated instances (AnimatedPrimitive) and alpha-textured curves are met by the VARIANT 2 kernels only (the back end selects them for
    // such scenes): the interpolation of the transformation and the replay of the alpha recursion are out-of-line callee chains the
    // other variants must not be able to reach (on the host: no such cost)
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool ANIM = VARIANT == 2;
#else
    constexpr bool ANIM = true;
#endif
    // `valid` = this thread has an item.  The queue push goes through BlockAlloc, which every thread of
    // the workgroup must reach: control flow below is flattened into the flag pushRay.
    using BxDF = typename MatBxDF<MAT>::T;
    const RayQueueV &q = ws.rq[cur];
    const RayQueueV &nq = ws.rq[cur ^ 1];
    bool pushRay = false;
    // next-ray payload
    V3 ro{0, 0, 0}, rwi{0, 0, 0};
    S4 rbeta = S4c(0.f), rr_u = S4c(0.f), rr_l = S4c(0.f);
    float retaScale = 0, time = 0;
    int rflags = 0, rmedium = -1, pixelIndex = 0, depth = 0;
    LightCtx rctx{};
    const bool live = valid;
#if defined(__HIP_DEVICE_COMPILE__)
    if (!valid && shadowIdle) qi = ws.counters[(CNT_MAT0 + MAT) * CNT_STRIDE] - 1;
#endif
    if (valid || shadowIdle) {
        int i = ws.matQ[MAT][qi];
        I4 meta = q.meta[i];
        pixelIndex = meta.x;
        depth = meta.y;
        bool anyNonSpecularBounces0 = meta.z & RAYFLAG_ANY_NONSPECULAR;
        F4 h = ws.hit[i];
        int prim = (int)FloatToBits(h.x);
        const int inst = HitInst(sv, ws, i);
        SurfIntr si;
        {
            V3 ro{0, 0, 0}, rd{0, 0, 0};   // only a curve's interaction needs the ray that found it
            if (sv.haveCurves) { F4 o4 = q.o[i], d4 = q.d[i]; ro = V3{o4.x, o4.y, o4.z}; rd = V3{d4.x, d4.y, d4.z}; }
            HitInteraction<!WF_DEV_LEAN, ANIM, ANIM>(sv, prim, inst, h.y, h.z, h.w, &si, ro, rd, q.o[i].w);   // (incl. alpha-textured curves and animated instances)
        }
        const wf_mesh &mesh = sv.meshes[si.mesh];
        int matId = mesh.material;
        if (sv.haveMix && sv.materials[matId].type == WF_MAT_MIX) matId = ws.mixMat[i];
        const wf_material &mat = sv.materials[matId];
        F4 o4 = q.o[i], d4 = q.d[i];
        time = o4.w;
        float etaScale0 = d4.w;
        // intr.wo: the Interaction constructor normalises it (interaction.h:40-43), also for unit-length ray.d;
        // items enqueued by the medium stage carry -ray.d as it is (media.cpp:240)
        V3 wo{-d4.x, -d4.y, -d4.z};
        if (!(sv.haveMedia && meta.w >= 0)) wo = IntrWo<!WF_DEV_LEAN, ANIM>(sv, prim, inst, wo, time);
        // differentials of position and (u, v) at the intersection (surfscatter.cpp:73-104)
        TexCtx tc;
        tc.p = si.pi.mid(); tc.n = si.n; tc.uv = si.uv;
        V3 dpdxVS{0, 0, 0}, dpdyVS{0, 0, 0};   // the dpdx / dpdy of surfscatter.cpp:77-79 (the visible surface keeps them)
        if constexpr (TEXCTX)
        if (!sv.options.disable_texture_filtering) {
            // movingFromCamera is the identity transform
            ApproximateDpDxy<VARIANT == 2>(sv, tc.p, si.n, &tc.dpdx, &tc.dpdy, time);   // (a moving camera: variant 2)
            dpdxVS = tc.dpdx; dpdyVS = tc.dpdy;
            V3 dpdu = si.dpdu, dpdv = si.dpdv;
            float ata00 = Dot(dpdu, dpdu), ata01 = Dot(dpdu, dpdv), ata11 = Dot(dpdv, dpdv);
            float invDet = 1 / DifferenceOfProducts(ata00, ata11, ata01, ata01);
            invDet = IsFinite(invDet) ? invDet : 0.f;
            float atb0x = Dot(dpdu, tc.dpdx), atb1x = Dot(dpdv, tc.dpdx);
            float atb0y = Dot(dpdu, tc.dpdy), atb1y = Dot(dpdv, tc.dpdy);
            float dudx = DifferenceOfProducts(ata11, atb0x, ata01, atb1x) * invDet;
            float dvdx = DifferenceOfProducts(ata00, atb1x, ata01, atb0x) * invDet;
            float dudy = DifferenceOfProducts(ata11, atb0y, ata01, atb1y) * invDet;
            float dvdy = DifferenceOfProducts(ata00, atb1y, ata01, atb0y) * invDet;
            tc.dudx = IsFinite(dudx) ? Clamp(dudx, -1e8f, 1e8f) : 0.f;
            tc.dvdx = IsFinite(dvdx) ? Clamp(dvdx, -1e8f, 1e8f) : 0.f;
            tc.dudy = IsFinite(dudy) ? Clamp(dudy, -1e8f, 1e8f) : 0.f;
            tc.dvdy = IsFinite(dvdy) ? Clamp(dvdy, -1e8f, 1e8f) : 0.f;
            // MaterialEvalWorkItem::GetMaterialEvalContext / GetNormalBumpEvalContext (wavefront/workitems.h:268-302) hand
            // the (u, v) differentials to the textures but leave ctx.dpdx / dpdy at their zero defaults: in the wavefront
            // path the spherical / cylindrical / planar mappings therefore see a zero footprint.  Reproduced as is.
            tc.dpdx = tc.dpdy = V3{0, 0, 0};
        }
        N3 ns = si.ns;
        V3 dpdus = si.dpdus;
        if constexpr (TEXCTX) {
        if (mat.normalmap >= 0) {
            // NormalMap (materials.h:86-106) and the shading frame rebuilt from it (surfscatter.cpp:111-118)
            V3 nm;
            NormalMapTexelP(sv.tableData, sv.texImages + mat.normalmap, tc.uv.x, tc.uv.y, &nm.x, &nm.y, &nm.z);
            nm = Normalize(nm);
            Frame frame = Frame::FromXZ(Normalize(si.dpdus), toV(si.ns));
            nm = frame.FromLocal(nm);
            float ulen = Length(si.dpdus), vlen = Length(si.dpdvs);
            dpdus = Normalize(GramSchmidt(si.dpdus, nm)) * ulen;
            V3 dpdvs = Normalize(Cross(nm, dpdus)) * vlen;
            ns = toN(Normalize(Cross(dpdus, dpdvs)));
            ns = FaceForward(ns, si.n);
        } else if (mat.displacement >= 0) {
            // BumpMap (materials.h:109-138) and the shading frame rebuilt from it (surfscatter.cpp:120-130)
            // MaterialEvalWorkItem::GetNormalBumpEvalContext (wavefront/workitems.h:268-285) leaves ctx.n at its zero default: the
            // three displacement lookups see n = 0 (a directionmix displacement weighs with |n . dir| = 0).  Reproduced as is.
            TexCtx bc = tc;
            bc.n = N3{0, 0, 0};
            TexCtx sh = bc;
            float du = .5f * (abs(tc.dudx) + abs(tc.dudy));
            if (du == 0) du = .0005f;
            sh.p = tc.p + du * si.dpdus;
            sh.uv = V2{tc.uv.x + du, tc.uv.y + 0.f};
            float uDisplace = EvalFloatTexture(sv, mat.displacement, sh);
            float dv = .5f * (abs(tc.dvdx) + abs(tc.dvdy));
            if (dv == 0) dv = .0005f;
            sh.p = tc.p + dv * si.dpdvs;
            sh.uv = V2{tc.uv.x + 0.f, tc.uv.y + dv};
            float vDisplace = EvalFloatTexture(sv, mat.displacement, sh);
            float displace = EvalFloatTexture(sv, mat.displacement, bc);
            dpdus = si.dpdus + (uDisplace - displace) / du * toV(si.ns) + displace * toV(si.dndus);
            V3 dpdvs = si.dpdvs + (vDisplace - displace) / dv * toV(si.ns) + displace * toV(si.dndvs);
            ns = toN(Normalize(Cross(dpdus, dpdvs)));
            ns = FaceForward(ns, si.n);
        }
        }
        Wavelengths lambda = LoadLambda(ws, pixelIndex);
        BxDF bxdf = MatBxDF<MAT>::Get(sv, mat, lambda, tc);
        BSDF<BxDF> bsdf(ns, dpdus, bxdf);
        if (live && lambda.SecondaryTerminated()) StoreLambda(ws, pixelIndex, lambda);
        if (sv.regularize && anyNonSpecularBounces0) bsdf.Regularize();
        if constexpr (VARIANT == 2)
        if (live && depth == 0 && sv.film.type == WF_FILM_GBUFFER) {
            // Initialize VisibleSurface at the first intersection (surfscatter.cpp:147-180): geometry + the BSDF's albedo, estimated
            // with the reference's 16 fixed samples (BxDF::rho, bxdfs.cpp:1131-1144)
            const float ucRho[16] = {0.75741637f, 0.37870818f, 0.7083487f, 0.18935409f, 0.9149363f, 0.35417435f, 0.5990858f, 0.09467703f,
                                     0.8578725f, 0.45746812f, 0.686759f, 0.17708716f, 0.9674518f, 0.2995429f, 0.5083201f, 0.047338516f};
            const float uRho[16][2] = {{0.855985f, 0.570367f}, {0.381823f, 0.851844f}, {0.285328f, 0.764262f}, {0.733380f, 0.114073f},
                                       {0.542663f, 0.344465f}, {0.127274f, 0.414848f}, {0.964700f, 0.947162f}, {0.594089f, 0.643463f},
                                       {0.095109f, 0.170369f}, {0.825444f, 0.263359f}, {0.429467f, 0.454469f}, {0.244460f, 0.816459f},
                                       {0.756135f, 0.731258f}, {0.516165f, 0.152852f}, {0.180888f, 0.214174f}, {0.898579f, 0.503897f}};
            S4 albedo = S4c(0.f);
            const V3 woLocal = bsdf.RenderToLocal(wo);
            if (woLocal.z != 0) {
                for (int k = 0; k < 16; ++k) {
                    BSDFSample rs = bsdf.bxdf.Sample_f(woLocal, ucRho[k], V2{uRho[k][0], uRho[k][1]}, MODE_RADIANCE, REFLTRANS_ALL);
                    if (rs.valid && rs.pdf > 0) albedo = albedo + rs.f * AbsCosTheta(rs.wi) / rs.pdf;
                }
                albedo = albedo / 16.f;
            }
            const N3 nf = FaceForward(si.n, wo), nsf = FaceForward(ns, wo);
            ws.vsP[pixelIndex] = F4{tc.p.x, tc.p.y, tc.p.z, 1.f};
            ws.vsN[pixelIndex] = F4{nf.x, nf.y, nf.z, tc.uv.x};
            ws.vsNs[pixelIndex] = F4{nsf.x, nsf.y, nsf.z, tc.uv.y};
            ws.vsDpdx[pixelIndex] = F4{dpdxVS.x, dpdxVS.y, dpdxVS.z, time};
            ws.vsDpdy[pixelIndex] = F4{dpdyVS.x, dpdyVS.y, dpdyVS.z, 0.f};
            ws.vsAlbedo[pixelIndex] = toF4(albedo);
        }

        S4 wbeta = toS4(q.beta[i]), wr_u = toS4(q.r_u[i]);
        F4 s0 = ws.samples0[pixelIndex], s1 = ws.samples1[pixelIndex];
        // Sample BSDF and enqueue indirect ray
        BSDFSample bs = bsdf.Sample_f(wo, s0.w, V2{s1.x, s1.y});
        if (bs.valid) {
            V3 wi = bs.wi;
            S4 beta = wbeta * bs.f * AbsDot(wi, ns) / bs.pdf;
            S4 r_u = wr_u, r_l;
            if (bs.pdfIsProportional) r_l = r_u / bsdf.PDF(wo, bs.wi);
            else r_l = r_u / bs.pdf;
            float etaScale = etaScale0;
            if (bs.IsTransmission()) etaScale *= Sqr(bs.eta);
            S4 rrBeta = beta * etaScale / r_u.Average();
            if (rrBeta.MaxComponentValue() < 1 && depth >= 1) {
                float qq = fmax(0.f, 1 - rrBeta.MaxComponentValue());
                if (s1.z < qq) beta = S4c(0.f);
                else beta = beta / (1 - qq);
            }
            if (MAT == WF_MAT_SUBSURFACE && beta && bs.IsTransmission()) {
                if (live) {
                // the path enters the medium: K12 takes over (surfscatter.cpp:226-231)
                BssrdfItem b;
                b.beta = toF4(beta); b.r_u = toF4(r_u);
                b.p = tc.p; b.etaScale = etaScale; b.wo = wo; b.depth = depth; b.n = si.n; b.pixelIndex = pixelIndex; b.ns = ns; b.material = matId;
                b.dpdus = dpdus; b.uv = tc.uv;
                const bool transition = mesh.medium_inside != mesh.medium_outside;
                b.mediumInside = transition ? mesh.medium_inside : meta.w;
                b.mediumOutside = transition ? mesh.medium_outside : meta.w;
                ws.bssrdfQ[QueueAlloc(&ws.counters[(CNT_BSSRDF) * CNT_STRIDE])] = b;
                }
            } else
            if (beta) {
                pushRay = true;
                ro = OffsetRayOrigin(si.pi, si.n, wi);
                rwi = wi;
                if (sv.haveMedia) rmedium = SurfaceMedium(mesh, si.n, wi, meta.w);
                bool anyNonSpecularBounces = !bs.IsSpecularS() || anyNonSpecularBounces0;
                rctx = LightCtx{si.pi, si.n, ns};
                rbeta = beta; rr_u = r_u; rr_l = r_l; retaScale = etaScale;
                rflags = (bs.IsSpecularS() ? RAYFLAG_SPECULAR_BOUNCE : 0) | (anyNonSpecularBounces ? RAYFLAG_ANY_NONSPECULAR : 0);
            }
        }
        // what next-event estimation needs of this item (surfscatter.cpp:251-327 reads it from the MaterialEvalWorkItem and the BSDF)
        {
            const int flags = bsdf.Flags();
            out->want = IsNonSpecular(flags);
            if (out->want) {
                if (IsReflective(flags) && !IsTransmissive(flags)) { out->ctxP = OffsetRayOrigin(si.pi, si.n, wo); out->ctxIsPoint = true; }
                else if (IsTransmissive(flags) && IsReflective(flags)) { out->ctxP = OffsetRayOrigin(si.pi, si.n, -wo); out->ctxIsPoint = true; }
            }
        }
        out->pi = si.pi; out->n = si.n; out->ns = ns; out->dpdus = dpdus; out->wo = wo;
        out->u0 = s0.x; out->u = V2{s0.y, s0.z};
        out->pixelIndex = pixelIndex;
        if (sv.haveMedia) {
            const bool transition = mesh.medium_inside != mesh.medium_outside;
            out->mediumInside = transition ? mesh.medium_inside : meta.w;
            out->mediumOutside = transition ? mesh.medium_outside : meta.w;
        }
        out->beta = wbeta; out->r_u = wr_u;
        for (int k = 0; k < 4; ++k) out->lambda[k] = lambda.lambda[k];
        out->bxdf = bsdf.bxdf;
    }

    // the indirect ray leaves NOW (the reference pushes it here too, surfscatter.cpp:232-249): its 35 values are not carried through
    // the light sampling
    pushRay = pushRay && live;
    {
        const int slot = BlockAlloc(&ws.counters[(CNT_RAY0 + (cur ^ 1)) * CNT_STRIDE], pushRay);
        if (pushRay) {
            nq.o[slot] = F4{ro.x, ro.y, ro.z, time};
            nq.d[slot] = F4{rwi.x, rwi.y, rwi.z, retaScale};
            nq.beta[slot] = toF4(rbeta);
            nq.r_u[slot] = toF4(rr_u);
            nq.r_l[slot] = toF4(rr_l);
            StoreCtx(nq, slot, rctx);
            nq.meta[slot] = I4{pixelIndex, depth + 1, rflags, rmedium};
        }
    }
}

// Sample light and enqueue shadow ray (surfscatter.cpp:251-327), in two steps around the workgroup's light sampling: what the light sample
// needs of an item (NeeRequest), and what is done with the sample (MatNeeFinish).  Every thread of the workgroup takes part (`live`: with
// an item whose side effects count); an item without next-event estimation has want = false.
struct NeeRequest {
    bool want = false;
    LightCtx ctx{P3i{V3{0, 0, 0}, V3{0, 0, 0}}, N3{0, 0, 0}, N3{0, 0, 0}};
    float u0 = 0;
    V2 u{0, 0};
    Wavelengths lambda{};
};
template <int MAT>
WF_HD NeeRequest MatNeeRequest(const NeeItem<MAT> &it) {
    NeeRequest r;
    r.want = it.want;
    r.ctx = LightCtx{it.ctxIsPoint ? MakeP3i(it.ctxP) : it.pi, it.n, it.ns};
    r.u0 = it.u0; r.u = it.u;
    for (int k = 0; k < 4; ++k) { r.lambda.lambda[k] = it.lambda[k]; r.lambda.pdf[k] = 0; }   // (the lights read the wavelengths only)
    return r;
}
template <int MAT>
WF_HD void MatNeeFinish(const SceneView &sv, const WorkState &ws, const NeeItem<MAT> &it, const LightPick &pick, bool live) {
    using BxDF = typename MatBxDF<MAT>::T;
    bool pushShadow = false;
    RayOD sr{V3{0, 0, 0}, V3{0, 0, 0}};
    S4 sLd = S4c(0.f), sr_u = S4c(0.f), sr_l = S4c(0.f);
    int smedium = -1;
    if (it.want && pick.lightId >= 0) {
        const float lightPMF = pick.pmf;
        const wf_light &light = sv.lights[pick.lightId];
        const LightLiSample &ls = pick.ls;
        if (ls.valid && ls.L && ls.pdf != 0) {
            const BSDF<BxDF> bsdf(it.ns, it.dpdus, it.bxdf);
            V3 wi = ls.wi;
            S4 f = bsdf.f(it.wo, wi);
            if (f) {
                S4 beta = it.beta * f * AbsDot(wi, it.ns);
                float lightPDF = ls.pdf * lightPMF;
                float bsdfPDF = IsDeltaLight(light) ? 0.f : bsdf.PDF(it.wo, wi);
                sr_u = it.r_u * bsdfPDF;
                sr_l = it.r_u * lightPDF;
                sLd = beta * ls.L;
                sr = SpawnRayTo(it.pi, it.n, ls.pLightPi, ls.pLightN);
                if (sv.haveMedia) smedium = Dot(sr.d, it.n) > 0 ? it.mediumOutside : it.mediumInside;   // SurfaceMedium
                pushShadow = true;
            }
        }
    }
    pushShadow = pushShadow && live;
    const int slot = BlockAlloc(&ws.counters[(CNT_SHADOW) * CNT_STRIDE], pushShadow);
    if (pushShadow) {
        ws.sq.o[slot] = F4{sr.o.x, sr.o.y, sr.o.z, 1 - ShadowEpsilon};
        ws.sq.d[slot] = F4{sr.d.x, sr.d.y, sr.d.z, BitsToFloat((uint32_t)it.pixelIndex)};
        ws.sq.Ld[slot] = toF4(sLd);
        ws.sq.r_u[slot] = toF4(sr_u);
        ws.sq.r_l[slot] = toF4(sr_l);
        if (sv.haveMedia) ws.sq.medium[slot] = smedium;
    }
}
template <int MAT, bool RARE_LIGHTS>
WF_HD void MatNee(const SceneView &sv, const WorkState &ws, const NeeItem<MAT> &it, bool live) {
    const NeeRequest rq = MatNeeRequest(it);
    const LightPick pick = SampleLightForBlock<RARE_LIGHTS>(sv, rq.want, rq.ctx, rq.u0, rq.u, rq.lambda);
    MatNeeFinish(sv, ws, it, pick, live);
}

// K9 fused: EvaluateMaterialAndBSDF<M, BasicTextureEvaluator>, wavefront/surfscatter.cpp:57-328 (the CPU checker; the HIP back end's
// WF_MAT_SPLIT=0 kernels)
template <int MAT, int VARIANT = 2>
WF_HD void KEvalMaterial(const SceneView &sv, const WorkState &ws, int cur, int qi, bool valid) {
    NeeItem<MAT> it;
#if defined(__HIP_DEVICE_COMPILE__)
    MatShade<MAT, VARIANT>(sv, ws, cur, qi, valid, &it, true);
#else
    MatShade<MAT, VARIANT>(sv, ws, cur, qi, valid, &it);
#endif
    MatNee<MAT, VARIANT == 2>(sv, ws, it, valid);
}

// ---------------------------------------------------------------------------------------------
// K12: WavefrontPathIntegrator::SampleSubsurface (wavefront/subsurface.cpp:18-203)
// "Get BSSRDF and enqueue probe ray" (:25-45)
WF_HD void KSubsurfaceProbe(const SceneView &sv, const WorkState &ws, int i) {
    const BssrdfItem w = ws.bssrdfQ[i];
    const wf_material &mat = sv.materials[w.material];
    TexCtx tc{};  // GetBSSRDFAndProbeRayWorkItem::GetMaterialEvalContext (workitems.h:176-186): p, n, ns, dpdus, wo, uv only
    tc.p = w.p; tc.n = w.n; tc.uv = w.uv;
    Wavelengths lambda = LoadLambda(ws, w.pixelIndex);
    TabulatedBSSRDF bssrdf = GetBSSRDF(sv, mat, lambda, tc, w.ns, w.wo);
    F4 s2 = ws.samples2[w.pixelIndex];
    V3 p0, p1;
    if (!bssrdf.SampleSp(s2.x, V2{s2.y, s2.z}, &p0, &p1)) return;
    SubsurfaceItem o{};
    o.p0 = p0; o.p1 = p1; o.depth = w.depth; o.material = w.material; o.bssrdf = bssrdf; o.beta = w.beta; o.r_u = w.r_u;
    o.mediumInside = w.mediumInside; o.mediumOutside = w.mediumOutside; o.etaScale = w.etaScale; o.pixelIndex = w.pixelIndex;
    ws.sssQ[QueueAlloc(&ws.counters[(CNT_SSS) * CNT_STRIDE])] = o;
}
// WavefrontAggregate::IntersectOneRandom (CPUAggregate: wavefront/aggregate.cpp:90-115; OptiX: gpu/optix/optix.cu:474-573)
// returns the reservoir's sample probability (0: the segment meets no surface of `material`) and the kept hit
// ANIM: the scene has animated primitives; the probe rays are spawned from an Interaction of time 0 ("FIXME time", aggregate.cpp:96)
template <bool ANIM = false, typename Stack>
WF_HD float IntersectOneRandom(const SceneView &sv, V3 p0, V3 p1, int material, Stack &st, ClosestHit *kept, SurfIntr *keptSi) {
    RNG rng;
    rng.SetSequence(Hash6f(p0, p1));   // WeightedReservoirSampler(seed) -> RNG(seed)
    float weightSum = 0, reservoirWeight = 0;
    P3i basePi = MakeP3i(p0);          // Interaction base(w.p0, 0.f, Medium()): exact point, zero normal
    N3 baseN{0, 0, 0};
    while (true) {
        RayOD r = SpawnRayTo(basePi, baseN, p1);
        if (r.d.x == 0 && r.d.y == 0 && r.d.z == 0) break;
        ClosestHit ch;
        st.n = 0;
        if (!BVHIntersectClosest<ANIM>(sv, r.o, r.d, 1.f, st, &ch, 0.f)) break;
        SurfIntr si;
        // (CURVE_ALPHA: a probe segment that crosses an alpha-textured curve is respawned from the interaction the reference's recursion ends
        //  with — GeometricPrimitive::Intersect's replay, wf_shapes.h — not from the first candidate's: ADVICE r5)
        HitInteraction<!WF_DEV_LEAN, true, ANIM>(sv, ch.prim, ch.inst, ch.h.b0, ch.h.b1, ch.h.b2, &si, r.o, r.d, 0.f);
        basePi = si.pi; baseN = si.n;
        if (sv.meshes[si.mesh].material == material) {
            // wrs.Add(SubsurfaceInteraction(si->intr), 1.f)  (util/sampling.h:535-546)
            weightSum += 1.f;
            float p = 1.f / weightSum;
            if (rng.UniformFloat() < p) {
                *kept = ch; *keptSi = si;
                reservoirWeight = 1.f;
            }
        }
    }
    return weightSum > 0 ? reservoirWeight / weightSum : 0.f;
}
template <bool ANIM = false, typename Stack>
WF_HD void KIntersectOneRandom(const SceneView &sv, const WorkState &ws, int i, Stack &st) {
    SubsurfaceItem &w = ws.sssQ[i];
    ClosestHit ch;
    SurfIntr si;
    w.reservoirPDF = IntersectOneRandom<ANIM>(sv, w.p0, w.p1, w.material, st, &ch, &si);
    if (w.reservoirPDF != 0) { w.pi = si.pi; w.n = si.n; w.ns = si.ns; w.dpdu = si.dpdu; w.dpdv = si.dpdv; w.dpdus = si.dpdus; w.dpdvs = si.dpdvs; }
}
// "Handle out-scattering after SSS" (:49-199)
WF_HD void KSubsurfaceScatter(const SceneView &sv, const WorkState &ws, int cur, int i) {
    const SubsurfaceItem &w = ws.sssQ[i];
    if (w.reservoirPDF == 0) return;
    const RayQueueV &nq = ws.rq[cur ^ 1];
    // TabulatedBSSRDF::ProbeIntersectionToSample (bssrdf.h:258-264)
    NormalizedFresnelBxDF bxdf{w.bssrdf.eta};
    V3 wo = toV(w.ns);
    BSDF<NormalizedFresnelBxDF> bsdf(w.ns, w.dpdus, bxdf);
    S4 Sp = w.bssrdf.Sp(w.pi.mid()), pdf = w.bssrdf.PDF_Sp(w.pi.mid(), w.n);
    if (!Sp || !pdf) return;
    float pr = w.reservoirPDF * pdf[0];
    S4 betap = toS4(w.beta) * Sp / pr;
    S4 r_u = toS4(w.r_u) * pdf / pdf[0];
    Wavelengths lambda = LoadLambda(ws, w.pixelIndex);
    F4 s0 = ws.samples0[w.pixelIndex], s1 = ws.samples1[w.pixelIndex];
    const float time = 0;  // "TODO: pipe through" (subsurface.cpp:74)
    // Indirect
    {
        BSDFSample bs = bsdf.Sample_f(wo, s0.w, V2{s1.x, s1.y});
        if (bs.valid) {
            V3 wi = bs.wi;
            S4 beta = betap * bs.f * AbsDot(wi, w.ns) / bs.pdf;
            S4 indir_r_u = r_u, r_l;
            if (bs.pdfIsProportional) r_l = r_u / bsdf.PDF(wo, bs.wi);
            else r_l = r_u / bs.pdf;
            float etaScale = w.etaScale;
            if (bs.IsTransmission()) etaScale *= Sqr(bs.eta);
            S4 rrBeta = beta * etaScale / indir_r_u.Average();
            if (rrBeta.MaxComponentValue() < 1 && w.depth > 1) {
                float q = fmax(0.f, 1 - rrBeta.MaxComponentValue());
                if (s1.z < q) beta = S4c(0.f);
                else beta = beta / (1 - q);
            }
            if (beta) {
                V3 ro = OffsetRayOrigin(w.pi, w.n, wi);
                int medium = -1;
                if (sv.haveMedia) medium = Dot(wi, w.n) > 0 ? w.mediumOutside : w.mediumInside;
                int slot = QueueAlloc(&ws.counters[(CNT_RAY0 + (cur ^ 1)) * CNT_STRIDE]);
                nq.o[slot] = F4{ro.x, ro.y, ro.z, time};
                nq.d[slot] = F4{wi.x, wi.y, wi.z, etaScale};
                nq.beta[slot] = toF4(beta);
                nq.r_u[slot] = toF4(indir_r_u);
                nq.r_l[slot] = toF4(r_l);
                StoreCtx(nq, slot, LightCtx{w.pi, w.n, w.ns});
                // anyNonSpecularBounces = true (subsurface.cpp:127)
                nq.meta[slot] = I4{w.pixelIndex, w.depth + 1, (bs.IsSpecularS() ? RAYFLAG_SPECULAR_BOUNCE : 0) | RAYFLAG_ANY_NONSPECULAR, medium};
            }
        }
    }
    // Direct lighting
    if (IsNonSpecular(bsdf.Flags())) {
        LightCtx ctx{w.pi, w.n, w.ns};
        float lightPMF = 0;
        int lightId = LightSamplerSample(sv, ctx, s0.x, &lightPMF);
        if (lightId < 0) return;
        const wf_light &light = sv.lights[lightId];
        LightLiSample ls = LightSampleLi(sv, light, ctx, V2{s0.y, s0.z}, lambda, true);
        if (!ls.valid || !ls.L || ls.pdf == 0) return;
        V3 wi = ls.wi;
        S4 f = bsdf.f(wo, wi);
        if (!f) return;
        S4 beta = betap * f * AbsDot(wi, w.ns);
        float lightPDF = ls.pdf * lightPMF;
        float bsdfPDF = IsDeltaLight(light) ? 0.f : bsdf.PDF(wo, wi);
        S4 r_l = r_u * lightPDF;
        r_u = r_u * bsdfPDF;
        S4 Ld = beta * ls.L;
        RayOD sr = SpawnRayTo(w.pi, w.n, ls.pLightPi, ls.pLightN);
        int slot = QueueAlloc(&ws.counters[(CNT_SHADOW) * CNT_STRIDE]);
        ws.sq.o[slot] = F4{sr.o.x, sr.o.y, sr.o.z, 1 - ShadowEpsilon};
        ws.sq.d[slot] = F4{sr.d.x, sr.d.y, sr.d.z, BitsToFloat((uint32_t)w.pixelIndex | (ws.pathTime ? SHADOW_TIME_ZERO : 0u))};   // time 0: see SHADOW_TIME_ZERO
        ws.sq.Ld[slot] = toF4(Ld);
        ws.sq.r_u[slot] = toF4(r_u);
        ws.sq.r_l[slot] = toF4(r_l);
        if (sv.haveMedia) ws.sq.medium[slot] = Dot(sr.d, w.n) > 0 ? w.mediumOutside : w.mediumInside;
    }
}

// K10 tail: RecordShadowRayResult, wavefront/intersect.h:31-46
WF_HD void KRecordShadowRay(const WorkState &ws, int i, bool occluded) {
    if (occluded) return;
    S4 Ld = toS4(ws.sq.Ld[i]) / (toS4(ws.sq.r_u[i]) + toS4(ws.sq.r_l[i])).Average();
    int pixelIndex = ShadowPixel(ws.sq.d[i].w);
    S4 Lpixel = toS4(ws.L[pixelIndex]);
    ws.L[pixelIndex] = toF4(Lpixel + Ld);
}

// K13: UpdateFilm (wavefront/film.cpp:14-38) -> RGBFilm::AddSample (film.h:239-255) -> PixelSensor::ToSensorRGB (film.h:95-101)
// One thread per band pixel p; it adds the pass's sample slots in sample order (the order the reference's
// successive passes would add them), so the double-precision sums are bit-identical for any samplesPerPass.
// the RGBFilm part of one item: the sample's sensor RGB times its filter weight, and the weight (what AddSample adds to the pixel's
// four double accumulators).  Returns false for an item outside the film's pixel bounds.
WF_HD bool FilmSampleRGBW(const SceneView &sv, const WorkState &ws, int pixelIndex, float out[4], size_t *filmIdx) {
    const wf_film &F = sv.film;
    I2 pp = ws.pPixel[pixelIndex];
    if (!(pp.x >= F.pixel_min[0] && pp.x < F.pixel_max[0] && pp.y >= F.pixel_min[1] && pp.y < F.pixel_max[1])) return false;
    S4 Lw = toS4(ws.L[pixelIndex]) * toS4(ws.cameraRayWeight[pixelIndex]);
    Wavelengths lambda = LoadLambda(ws, pixelIndex);
    float filterWeight = ws.filterWeight[pixelIndex];
    S4 L = SafeDiv(Lw, lambda.PDF());
    float r = F.imaging_ratio * (DenseSample(sv, F.rbar_offset, lambda) * L).Average();
    float g = F.imaging_ratio * (DenseSample(sv, F.gbar_offset, lambda) * L).Average();
    float b = F.imaging_ratio * (DenseSample(sv, F.bbar_offset, lambda) * L).Average();
    float m = fmax(fmax(r, g), b);
    if (m > F.max_component_value) {
        float sc = F.max_component_value / m;
        r *= sc; g *= sc; b *= sc;
    }
    out[0] = filterWeight * r; out[1] = filterWeight * g; out[2] = filterWeight * b; out[3] = filterWeight;
    const int width = F.pixel_max[0] - F.pixel_min[0];
    *filmIdx = (size_t)(pp.y - F.pixel_min[1]) * width + (pp.x - F.pixel_min[0]);
    return true;
}
WF_HD void KUpdateFilm(const SceneView &sv, const WorkState &ws, int p, int nSamples) {
    const wf_film &F = sv.film;
    for (int slot = 0; slot < nSamples; ++slot) {
        const int pixelIndex = ws.slotStride > 0 ? p * ws.slotStride + slot : slot * ws.pixelsPerPass + p;
        I2 pp = ws.pPixel[pixelIndex];
        if (!(pp.x >= F.pixel_min[0] && pp.x < F.pixel_max[0] && pp.y >= F.pixel_min[1] && pp.y < F.pixel_max[1])) return;
        S4 Lw = toS4(ws.L[pixelIndex]) * toS4(ws.cameraRayWeight[pixelIndex]);
        Wavelengths lambda = LoadLambda(ws, pixelIndex);
        float filterWeight = ws.filterWeight[pixelIndex];
        S4 L = SafeDiv(Lw, lambda.PDF());
        float r = F.imaging_ratio * (DenseSample(sv, F.rbar_offset, lambda) * L).Average();
        float g = F.imaging_ratio * (DenseSample(sv, F.gbar_offset, lambda) * L).Average();
        float b = F.imaging_ratio * (DenseSample(sv, F.bbar_offset, lambda) * L).Average();
        float m = fmax(fmax(r, g), b);
        if (m > F.max_component_value) {
            float sc = F.max_component_value / m;
            r *= sc; g *= sc; b *= sc;
        }
        int width = F.pixel_max[0] - F.pixel_min[0];
        size_t idx = (size_t)(pp.y - F.pixel_min[1]) * width + (pp.x - F.pixel_min[0]);
        double *px = ws.film + 4 * idx;
        px[0] += filterWeight * r;
        px[1] += filterWeight * g;
        px[2] += filterWeight * b;
        px[3] += filterWeight;
        if (F.type == WF_FILM_GBUFFER) {
            // GBufferFilm::AddSample (film.cpp:588-641), the part beside the RGB accumulators
            const F4 vp = ws.vsP[pixelIndex];
            if (vp.w != 0) {
                wf_gbuffer_pixel &gb = ws.filmGBuffer[idx];
                gb.gbuffer_weight_sum += filterWeight;
                const float rgbv[3] = {r, g, b};
                for (int c = 0; c < 3; ++c) {   // VarianceEstimator::Add (util/sampling.h:488-494)
                    ++gb.var_n[c];
                    const float delta = rgbv[c] - gb.var_mean[c];
                    gb.var_mean[c] += delta / gb.var_n[c];
                    const float delta2 = rgbv[c] - gb.var_mean[c];
                    gb.var_s[c] += delta * delta2;
                }
                const F4 vn = ws.vsN[pixelIndex], vns = ws.vsNs[pixelIndex], vdx = ws.vsDpdx[pixelIndex], vdy = ws.vsDpdy[pixelIndex];
                const wf_transform &X = F.gbuffer_from_render;
                V3 po, dx, dy;
                N3 no, nso;
                if (F.apply_inverse) {   // Transform::ApplyInverse of points / normals / vectors (util/transform.h:385-414)
                    po = XfInvPointM(X.mInv, V3{vp.x, vp.y, vp.z});
                    no = XfNormal(X.m, N3{vn.x, vn.y, vn.z});
                    nso = XfNormal(X.m, N3{vns.x, vns.y, vns.z});
                    dx = XfVector3(X.mInv, V3{vdx.x, vdx.y, vdx.z});
                    dy = XfVector3(X.mInv, V3{vdy.x, vdy.y, vdy.z});
                } else {
                    po = XfInvPointM(X.m, V3{vp.x, vp.y, vp.z});
                    no = XfNormal(X.mInv, N3{vn.x, vn.y, vn.z});
                    nso = XfNormal(X.mInv, N3{vns.x, vns.y, vns.z});
                    dx = XfVector3(X.m, V3{vdx.x, vdx.y, vdx.z});
                    dy = XfVector3(X.m, V3{vdy.x, vdy.y, vdy.z});
                }
                gb.p_sum[0] += filterWeight * po.x; gb.p_sum[1] += filterWeight * po.y; gb.p_sum[2] += filterWeight * po.z;
                gb.n_sum[0] += filterWeight * no.x; gb.n_sum[1] += filterWeight * no.y; gb.n_sum[2] += filterWeight * no.z;
                gb.ns_sum[0] += filterWeight * nso.x; gb.ns_sum[1] += filterWeight * nso.y; gb.ns_sum[2] += filterWeight * nso.z;
                gb.dzdx_sum += filterWeight * dx.z;
                gb.dzdy_sum += filterWeight * dy.z;
                gb.uv_sum[0] += filterWeight * vn.w; gb.uv_sum[1] += filterWeight * vns.w;
                // albedo * the colour space's illuminant -> RGB (SampledSpectrum::ToRGB, util/spectrum.cpp:205-228)
                const S4 alb = toS4(ws.vsAlbedo[pixelIndex]) * DenseSample(sv, F.illuminant_offset, lambda);
                const S4 pdf = lambda.PDF();
                const float xyz[3] = {SafeDiv(DenseSample(sv, F.rbar_offset, lambda) * alb, pdf).Average() / 106.856895f,
                                      SafeDiv(DenseSample(sv, F.gbar_offset, lambda) * alb, pdf).Average() / 106.856895f,
                                      SafeDiv(DenseSample(sv, F.bbar_offset, lambda) * alb, pdf).Average() / 106.856895f};
                for (int c = 0; c < 3; ++c) {
                    float v = 0;   // Mul<RGB>(RGBFromXYZ, xyz) (util/math.h: generic accumulation)
                    for (int k = 0; k < 3; ++k) v += F.RGBFromXYZ[c][k] * xyz[k];
                    gb.rgb_albedo_sum[c] += filterWeight * v;
                }
            }
        }
        if (F.type == WF_FILM_SPECTRAL) {
            // SpectralFilm::AddSample, the spectral part (film.h:432-457): the radiance itself — not divided by the wavelengths' PDF,
            // which is uniform —, clamped, times weight * CIE_Y_integral, into the buckets of its four wavelengths
            S4 Ls = Lw;
            const float lm = Ls.MaxComponentValue();
            if (lm > F.max_component_value) Ls = Ls * (F.max_component_value / lm);
            Ls = Ls * (filterWeight * 106.856895f);
            double *sp = ws.filmSpectral + (size_t)2 * F.n_buckets * idx;
            for (int i = 0; i < 4; ++i) {
                int b = (int)(F.n_buckets * (lambda.lambda[i] - F.lambda_min) / (F.lambda_max - F.lambda_min));
                b = b < 0 ? 0 : (b > F.n_buckets - 1 ? F.n_buckets - 1 : b);
                sp[b] += Ls[i];
                sp[F.n_buckets + b] += filterWeight;
            }
        }
    }
}

}  // namespace wf
