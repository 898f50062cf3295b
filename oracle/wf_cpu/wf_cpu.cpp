// oracle/wf_cpu/wf_cpu.cpp — TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by, or called from the
// product (libwfhip.so / libwfhost.so / the pbrt_amd CLI); only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may run it, and only as the checker.
//
// CPU restatement ("port") of the reference's wavefront path: the loop of
// WavefrontPathIntegrator::Render (wavefront/integrator.cpp:290-493) over the stage bodies restated in
// pbrt-v4_amd/csrc/common/wf_*.h (each function there cites the reference file:line it follows), with
// CPUAggregate-style traversal (wavefront/aggregate.cpp:34-68: int nodesToVisit[64] stack,
// cpu/aggregates.cpp:529-624).  It exists so that (a) the restated arithmetic can be pinned against the
// real reference (oracle/_ref/pbrt_ref --wavefront, built from /root/reference by oracle/ref_build) on a
// machine without a GPU, and (b) the HIP kernels can be compared with it item by item on the GPU box,
// where /root/reference does not exist.  Pinning status: see DESIGN.md "Oracle".
//
// usage: wf_cpu [--spp N] [--seed N] [--nthreads N] [--outfile out.pfm] [--dump-film film.bin]
//               [--datadir DIR] [--trace rays.bin hits.bin] [--samples begin end step]
//               [--sampler-probe in.bin out.bin startDim ndims] scene.pbrt
#include <atomic>
// Triangle::PDF's own intersection (shapes.h:1133-1141) is one of the tests `pbrt --stats` counts
static std::atomic<unsigned long long> g_pdfTriTests{0};
#define WF_COUNT_TRI_TESTS (++g_pdfTriTests)
#include "../../pbrt-v4_amd/csrc/common/wf_kernels.h"
#include "../../pbrt-v4_amd/csrc/host/scene.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

using namespace wf;

static int gThreads = 1;
static void ParallelFor(int n, const std::function<void(int)> &f) {
    if (n <= 0) return;
    int nt = std::min(gThreads, std::max(1, n / 64));
    if (nt <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
    std::atomic<int> next{0};
    const int chunk = std::max(64, n / (8 * nt));
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&] {
            while (true) {
                int b = next.fetch_add(chunk);
                if (b >= n) break;
                int e = std::min(n, b + chunk);
                for (int i = b; i < e; ++i) f(i);
            }
        });
    for (auto &t : th) t.join();
}

template <typename T> static T *Alloc(size_t n) { return (T *)calloc(n, sizeof(T)); }

static void AllocRayQueue(RayQueueV *q, int n) {
    q->o = Alloc<F4>(n); q->d = Alloc<F4>(n); q->beta = Alloc<F4>(n); q->r_u = Alloc<F4>(n); q->r_l = Alloc<F4>(n);
    q->ctx0 = Alloc<F4>(n); q->ctx1 = Alloc<F4>(n); q->ctx2 = Alloc<F4>(n); q->meta = Alloc<I4>(n);
}

SceneView MakeHostView(const wf_scene_desc &d, const uint32_t *sobol) {
    SceneView sv{};
    sv.P = d.P; sv.N = d.N; sv.UV = d.UV; sv.S = d.S; sv.triIndices = d.tri_indices; sv.triMesh = d.tri_mesh; sv.meshes = d.meshes;
    sv.bvhNodes = d.bvh_nodes; sv.bvhPrims = d.bvh_prims; sv.nTriangles = d.n_triangles; sv.nBvhNodes = d.n_bvh_nodes;
    sv.spectra = d.spectra; sv.spectrumData = d.spectrum_data; sv.textures = d.textures; sv.materials = d.materials;
    sv.lights = d.lights; sv.infiniteLights = d.infinite_lights; sv.lightBvh = d.light_bvh_nodes; sv.lightXforms = d.light_transforms;
    sv.nLights = d.n_lights; sv.nInfiniteLights = d.n_infinite_lights; sv.nLightBvhNodes = d.n_light_bvh_nodes; sv.lightSampler = d.light_sampler;
    for (int i = 0; i < 6; ++i) sv.allLightBounds[i] = d.all_light_bounds[i];
    sv.camera = d.camera; sv.film = d.film; sv.filter = d.filter; sv.filterData = d.filter_data; sv.sampler = d.sampler;
    sv.sobol = sobol;
    sv.powerAlias = d.power_alias;
    sv.texImages = d.tex_images;
    sv.imageLights = d.image_lights; sv.tableData = d.table_data; sv.rgb2specCoeffs = d.rgb2spec_coeffs;
    sv.rgb2specZNodes = d.rgb2spec_znodes;
    sv.noisePerm = d.noise_perm;
    sv.csIlluminantOffset = d.cs_illuminant_offset;
    sv.media = d.media; sv.mediumData = d.medium_data;
    sv.maxDepth = d.max_depth; sv.regularize = d.regularize; sv.haveMedia = d.have_media; sv.options = d.options;
    sv.texNeedsFootprint = 0;
    sv.haveAlpha = 0;
    for (int i = 0; i < d.n_meshes; ++i)
        if (d.meshes[i].alpha_tex >= 0) sv.haveAlpha = 1;
    for (int i = 0; i < d.n_textures; ++i)
        if (d.textures[i].type >= WF_TEX_FLOAT_IMAGE) sv.texNeedsFootprint = 1;
    for (int i = 0; i < d.n_materials; ++i)
        if (d.materials[i].displacement >= 0 || d.materials[i].normalmap >= 0) sv.texNeedsFootprint = 1;
    sv.matTypeMask = 0;
    sv.quadrics = d.quadrics; sv.nQuadrics = d.n_quadrics;
    sv.instances = d.instances; sv.instanceDefs = d.instance_defs; sv.nInstances = d.n_instances;
    sv.animated = d.animated; sv.haveAnimated = d.n_animated > 0;
    sv.sobolMatrices = d.sobol_matrices; sv.vdcSobol = d.vdc_sobol; sv.vdcSobolInv = d.vdc_sobol_inv;
    sv.haltonPrimes = d.halton_primes; sv.haltonPermOffsets = d.halton_perm_offsets; sv.haltonPerms = d.halton_perms;
    sv.haveMix = 0;
    sv.haveSubsurface = 0;
    sv.haveQuadricAlpha = 0;
    for (int i = 0; i < d.n_quadrics; ++i) if (d.meshes[d.quadrics[i].mesh].alpha_tex >= 0) sv.haveQuadricAlpha = 1;
    sv.haveCurves = 0;
    for (int i = 0; i < d.n_quadrics; ++i) if (d.quadrics[i].type == WF_QUADRIC_CURVE) sv.haveCurves = 1;
    for (int i = 0; i < d.n_materials; ++i) {
        if (d.materials[i].type == WF_MAT_MIX) sv.haveMix = 1;
        if (d.materials[i].type == WF_MAT_SUBSURFACE) sv.haveSubsurface = 1;
        else sv.matTypeMask |= 1 << d.materials[i].type;
    }
    return sv;
}


// ---- design study (not a checker): lane utilisation of wave-synchronous traversal schemes ---------------------
// Walks each ray of a queue the way the production kernel does (pair nodes: both children tested at the parent,
// nearest first) and records its event string: 0 = interior step, k > 0 = leaf with k triangle tests.  Then, for
// groups of 64 consecutive queue entries (= one wave), counts the instruction cost of: "while-while" (all lanes
// descend until each sits at a leaf, then leaves together), "unified" (every iteration each lane does one interior
// step or one triangle test) and the ideal (no divergence).
static float SlabEntry(const float bmin[3], const float bmax[3], V3 o, float raytMax, V3 invDir, int negMask, bool *hit) {
    const bool n0 = negMask & 1, n1 = negMask & 2, n2 = negMask & 4;
    float tMin = ((n0 ? bmax[0] : bmin[0]) - o.x) * invDir.x, tMax = ((n0 ? bmin[0] : bmax[0]) - o.x) * invDir.x;
    float tyMin = ((n1 ? bmax[1] : bmin[1]) - o.y) * invDir.y, tyMax = ((n1 ? bmin[1] : bmax[1]) - o.y) * invDir.y;
    float tzMin = ((n2 ? bmax[2] : bmin[2]) - o.z) * invDir.z, tzMax = ((n2 ? bmin[2] : bmax[2]) - o.z) * invDir.z;
    tMin = std::max(tMin, std::max(tyMin, tzMin));
    tMax = std::min(tMax, std::min(tyMax, tzMax)) * (1 + 2 * gamma(3));
    *hit = tMin <= tMax && tMin < raytMax && tMax > 0;
    return tMin;
}
static void RayEvents(const SceneView &sv, V3 o, V3 d, float tMax, std::vector<uint8_t> *ev) {
    ev->clear();
    V3 invDir{1 / d.x, 1 / d.y, 1 / d.z};
    int negMask = int(invDir.x < 0) | (int(invDir.y < 0) << 1) | (int(invDir.z < 0) << 2);
    int stack[128], sp = 0;
    int node = 0;
    if (sv.bvhNodes[0].nprims > 0) return;
    while (true) {
        const wf_bvh_node &n = sv.bvhNodes[node];
        if (n.nprims == 0) {
            ev->push_back(0);
            int c[2] = {node + 1, (int)n.offset};
            bool h[2];
            float t[2];
            for (int k = 0; k < 2; ++k) t[k] = SlabEntry(sv.bvhNodes[c[k]].bmin, sv.bvhNodes[c[k]].bmax, o, tMax, invDir, negMask, &h[k]);
            if (h[0] && h[1]) {
                int nearC = t[1] < t[0] ? 1 : 0;
                stack[sp++] = c[nearC ^ 1];
                node = c[nearC];
                continue;
            } else if (h[0]) { node = c[0]; continue; }
            else if (h[1]) { node = c[1]; continue; }
        } else {
            ev->push_back((uint8_t)n.nprims);
            for (int i = 0; i < n.nprims; ++i) {
                int tri = sv.bvhPrims[n.offset + i];
                V3 p0, p1, p2;
                TriVerts(sv, tri, &p0, &p1, &p2);
                TriHit hh;
                if (IntersectTriangle(o, d, tMax, p0, p1, p2, &hh)) tMax = hh.t;
            }
        }
        if (sp == 0) break;
        node = stack[--sp];
    }
}

// streaming scheme: a wave owns a run of queue entries; finished lanes are refilled when >= R are idle; an iteration
// runs the leaf step when >= LT lanes wait at a leaf (or nobody is interior), else the interior step
static double SimulateStream(const std::vector<std::vector<uint8_t>> &rays, int R, int LT, double *itersOut) {
    const double CI = 60, CT = 130, CR = 160;
    size_t next = 0;
    struct Lane { const std::vector<uint8_t> *ev = nullptr; size_t pos = 0; };
    Lane lane[64];
    double cost = 0, iters = 0;
    while (true) {
        int idle = 0;
        for (auto &L : lane) if (!L.ev || L.pos >= L.ev->size()) ++idle;
        if (idle >= R && next < rays.size()) {
            for (auto &L : lane)
                if ((!L.ev || L.pos >= L.ev->size()) && next < rays.size()) { L.ev = &rays[next++]; L.pos = 0; }
            cost += CR;
            continue;
        }
        int nI = 0, nL = 0, maxk = 0;
        for (auto &L : lane) if (L.ev && L.pos < L.ev->size()) { if ((*L.ev)[L.pos] == 0) ++nI; else { ++nL; maxk = std::max(maxk, (int)(*L.ev)[L.pos]); } }
        if (nI == 0 && nL == 0) { if (next >= rays.size()) break; else continue; }
        iters += 1;
        if (nL >= LT || nI == 0) {
            for (auto &L : lane) if (L.ev && L.pos < L.ev->size() && (*L.ev)[L.pos] != 0) ++L.pos;
            cost += maxk * CT;
        } else {
            for (auto &L : lane) if (L.ev && L.pos < L.ev->size() && (*L.ev)[L.pos] == 0) ++L.pos;
            cost += CI;
        }
    }
    *itersOut = iters;
    return cost;
}
static uint64_t EncodeMorton3(uint32_t x, uint32_t y, uint32_t z) {
    auto spread = [](uint64_t v) { uint64_t r = 0; for (int b = 0; b < 10; ++b) r |= ((v >> b) & 1ull) << (3 * b); return r; };
    return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}
struct WaveSim { double ww = 0, unified = 0, ideal = 0, spec = 0; double wwI = 0, wwT = 0, sumI = 0, sumT = 0; long waves = 0; };
static void SimulateWave(const std::vector<std::vector<uint8_t>> &ev, int lanes, WaveSim *S) {
    const double CI = 60, CT = 130;
    // ideal
    double sI = 0, sT = 0;
    for (int l = 0; l < lanes; ++l) for (uint8_t e : ev[l]) { if (e == 0) sI += 1; else sT += e; }
    S->sumI += sI; S->sumT += sT;
    S->ideal += (sI * CI + sT * CT) / 64.0;
    // while-while
    {
        std::vector<size_t> pos(lanes, 0);
        while (true) {
            int A = 0;
            bool any = false;
            for (int l = 0; l < lanes; ++l) {
                int run = 0;
                while (pos[l] < ev[l].size() && ev[l][pos[l]] == 0) { ++pos[l]; ++run; }
                A = std::max(A, run);
                if (pos[l] < ev[l].size()) any = true;
            }
            S->ww += A * CI; S->wwI += A;
            if (!any) break;
            int B = 0;
            for (int l = 0; l < lanes; ++l) if (pos[l] < ev[l].size()) { B = std::max(B, (int)ev[l][pos[l]]); ++pos[l]; }
            S->ww += B * CT; S->wwT += B;
        }
    }
    // unified: per iteration each lane consumes one unit
    {
        std::vector<size_t> pos(lanes, 0);
        std::vector<int> rem(lanes, 0);
        while (true) {
            bool anyI = false, anyT = false;
            for (int l = 0; l < lanes; ++l) {
                if (rem[l] > 0) { --rem[l]; anyT = true; continue; }
                if (pos[l] >= ev[l].size()) continue;
                uint8_t e = ev[l][pos[l]++];
                if (e == 0) anyI = true;
                else { rem[l] = e - 1; anyT = true; }
            }
            if (!anyI && !anyT) break;
            S->unified += (anyI ? CI : 0) + (anyT ? CT : 0);
        }
    }
    // speculative while-while: a lane parks its first leaf and keeps descending until its second leaf
    {
        std::vector<size_t> pos(lanes, 0);
        std::vector<int> pend(lanes, 0);
        while (true) {
            int A = 0;
            for (int l = 0; l < lanes; ++l) {
                int run = 0;
                while (pos[l] < ev[l].size()) {
                    uint8_t e = ev[l][pos[l]];
                    if (e == 0) { ++pos[l]; ++run; }
                    else if (pend[l] == 0) { pend[l] = e; ++pos[l]; }
                    else break;
                }
                A = std::max(A, run);
            }
            S->spec += A * CI;
            int B = 0;
            bool any = false;
            for (int l = 0; l < lanes; ++l) { B = std::max(B, pend[l]); pend[l] = 0; if (pos[l] < ev[l].size()) any = true; }
            S->spec += B * CT;
            if (!any && B == 0) break;
        }
    }
    S->waves++;
}

static int Main(int argc, char **argv);
int main(int argc, char **argv) {
    try {
        return Main(argc, argv);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
static int Main(int argc, char **argv) {
    RenderOptions opt;
    std::string scenePath, dumpFilm, dataDir, traceRays, traceHits, probeIn, probeOut, dumpStages;
    bool traceTimed = false;
    bool emulateStaleDepth = false;   // g_msStaleDepth (wf_kernels.h): the reference's unwritten MediumSampleWorkItem::depth, sequential order only
    bool tracePath = false;   // print the path state after every stage (the lines oracle/_ref/ref_trace prints; tools/trace_diff.py)
    std::string lightProbeIn, lightProbeOut, reProbeIn, reProbeOut;
    bool simulateWaves = false;
    int sampleBegin = 0, sampleEnd = -1, sampleStep = 1, probeStartDim = 0, probeNDims = 0;
    int stripRank = 0, stripCount = 1, stripHeight = 16;
    gThreads = std::max(1u, std::thread::hardware_concurrency());
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value after %s\n", a.c_str()); exit(1); } return argv[++i]; };
        if (a == "--spp") opt.pixelSamples = atoi(next().c_str());
        else if (a == "--seed") opt.seed = atoi(next().c_str());
        else if (a == "--displacement-edge-scale") opt.displacementEdgeScale = (float)atof(next().c_str());
        else if (a == "--render-coord-sys") { const std::string v = next(); opt.renderingSpace = v == "camera" ? 0 : (v == "world" ? 2 : 1); }
        else if (a == "--nthreads") gThreads = atoi(next().c_str());
        else if (a == "--cropwindow") { if (sscanf(next().c_str(), "%f,%f,%f,%f", &opt.cropWindow[0], &opt.cropWindow[1], &opt.cropWindow[2], &opt.cropWindow[3]) == 4) opt.hasCropWindow = true; }
        else if (a == "--pixelbounds") { if (sscanf(next().c_str(), "%d,%d,%d,%d", &opt.pixelBounds[0], &opt.pixelBounds[1], &opt.pixelBounds[2], &opt.pixelBounds[3]) == 4) opt.hasPixelBounds = true; }
        else if (a == "--disable-pixel-jitter") opt.disablePixelJitter = true;
        else if (a == "--disable-wavelength-jitter") opt.disableWavelengthJitter = true;
        else if (a == "--disable-texture-filtering") opt.disableTextureFiltering = true;
        else if (a == "--quick") opt.quickRender = true;
        else if (a == "--disable-image-textures") opt.disableImageTextures = true;
        else if (a == "--pixel") { int px = 0, py = 0; sscanf(next().c_str(), "%d,%d", &px, &py); opt.pixelBounds[0] = px; opt.pixelBounds[1] = px + 1; opt.pixelBounds[2] = py; opt.pixelBounds[3] = py + 1; opt.hasPixelBounds = true; }
        else if (a == "--debugstart") { int f = 0, c = 1; if (sscanf(next().c_str(), "%d,%d", &f, &c) < 2) c = 1; sampleBegin = f; sampleEnd = f + c; }
        else if (a == "--outfile") opt.imageFile = next();
        else if (a == "--dump-film") dumpFilm = next();
        else if (a == "--datadir") dataDir = next();
        else if (a == "--quiet") opt.quiet = true;
        else if (a == "--trace") { traceRays = next(); traceHits = next(); }
        else if (a == "--trace-timed") { traceRays = next(); traceHits = next(); traceTimed = true; }   // records of 8 floats: o, d, tMax, time (AnimatedPrimitive)
        else if (a == "--dump-stages") dumpStages = next();
        else if (a == "--trace-path") tracePath = true;
        else if (a == "--emulate-stale-medium-depth") emulateStaleDepth = true;
        else if (a == "--simulate-waves") simulateWaves = true;
        else if (a == "--strips") { stripRank = atoi(next().c_str()); stripCount = atoi(next().c_str()); stripHeight = atoi(next().c_str()); }
        else if (a == "--samples") { sampleBegin = atoi(next().c_str()); sampleEnd = atoi(next().c_str()); sampleStep = atoi(next().c_str()); }
        else if (a == "--reintersect-probe") { reProbeIn = next(); reProbeOut = next(); }
        else if (a == "--light-probe") { lightProbeIn = next(); lightProbeOut = next(); }
        else if (a == "--sampler-probe") { probeIn = next(); probeOut = next(); probeStartDim = atoi(next().c_str()); probeNDims = atoi(next().c_str()); }
        else if (a[0] == '-') { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
        else scenePath = a;
    }
    if (scenePath.empty()) { fprintf(stderr, "usage: wf_cpu [options] scene.pbrt\n"); return 1; }
    if (dataDir.empty()) {
        // <repo>/pbrt-v4_amd/data relative to this binary's usual location oracle/_build/wf_cpu
        std::string self = argv[0];
        size_t p = self.rfind('/');
        dataDir = (p == std::string::npos ? std::string(".") : self.substr(0, p)) + "/../../pbrt-v4_amd/data";
    }
    SpectralData::Init(dataDir, dataDir + "/cache");
    ParsedScene parsed;
    ParseFiles({scenePath}, &opt, &parsed);
    SceneTables T;
    BuildSceneTables(parsed, opt, &T);
    static uint32_t sobol[WF_SOBOL_WORDS];
    FillSobol2D(sobol);
    SceneView sv = MakeHostView(T.desc, sobol);
    sv.self = &sv;
    static int32_t fatalWord = 0;
    sv.fatal = &fatalWord;   // RaiseFatal (wf_scene.h): checked at the end of the render

    // sampler probe: in = n x {px, py, sampleIndex} int32 -> out = n x ndims floats (Get1D from startDim)
    if (!probeIn.empty()) {
        FILE *f = fopen(probeIn.c_str(), "rb");
        if (!f) { perror(probeIn.c_str()); return 1; }
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        int n = (int)(sz / (3 * sizeof(int32_t)));
        std::vector<int32_t> in((size_t)n * 3);
        if (fread(in.data(), 4, in.size(), f) != in.size()) return 1;
        fclose(f);
        // ndims = -2: the sample's GetPixel2D() instead (samplers_test.cpp's elementary-interval tests read it)
        // ndims = -3: ten times (Get2D, Get1D), the sequence Sampler.ConsistentValues draws (samplers_test.cpp:46-52): 30 floats;
        // ndims = -4: the ZSobol sample index at the start dimension (ZSobolSampler.ValidIndices, :168-196): low / high 32 bits in two float slots
        const bool pixel2D = probeNDims == -2, pattern = probeNDims == -3, zindex = probeNDims == -4;
        if (pixel2D || zindex) probeNDims = 2;
        if (pattern) probeNDims = 30;
        std::vector<float> out((size_t)n * probeNDims);
        for (int i = 0; i < n; ++i) {
            PixelSampler s(sv);
            s.StartPixelSample(in[3 * i], in[3 * i + 1], in[3 * i + 2], probeStartDim);
            SamplerProbeRecord(s, pixel2D ? -2 : pattern ? -3 : zindex ? -4 : probeNDims, &out[(size_t)i * probeNDims]);
        }
        f = fopen(probeOut.c_str(), "wb");
        fwrite(out.data(), 4, out.size(), f);
        fclose(f);
        return 0;
    }

    // light-sampler probe (the reference's BVHLightSampling / PowerLightSampling unit tests run against the scene's sampler,
    // tests/test_reference_known_answers.py): in = n x {p[3], n[3], uLight, u[2]} floats -> out = n x 7 floats
    // {sampled light id or -1, its probability, LightSampler::PMF of that light, the sampled light's SampleLi is valid,
    //  light 0's SampleLi is valid, light 0's SampleLi carries radiance, LightSampler::PMF of light 0}
    if (!lightProbeIn.empty()) {
        FILE *f = fopen(lightProbeIn.c_str(), "rb");
        if (!f) { perror(lightProbeIn.c_str()); return 1; }
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        int n = (int)(sz / (9 * sizeof(float)));
        std::vector<float> in((size_t)n * 9), out((size_t)n * 7);
        if (fread(in.data(), 4, in.size(), f) != in.size()) return 1;
        fclose(f);
        const Wavelengths lambda = SampleUniformWavelengths(0.5f);
        ParallelFor(n, [&](int i) {
            const float *r = &in[(size_t)i * 9];
            LightCtx ctx;
            ctx.pi = MakeP3i(V3{r[0], r[1], r[2]}, V3{0, 0, 0});
            ctx.n = ctx.ns = N3{r[3], r[4], r[5]};
            float pmf = 0;
            const int id = LightSamplerSample(sv, ctx, r[6], &pmf);
            float *o = &out[(size_t)i * 7];
            o[0] = (float)id;
            o[1] = id >= 0 ? pmf : 0.f;
            o[2] = id >= 0 ? LightSamplerPMF(sv, ctx, id) : 0.f;
            o[3] = id >= 0 && LightSampleLi(sv, sv.lights[id], ctx, V2{r[7], r[8]}, lambda, true).valid ? 1.f : 0.f;
            o[4] = o[5] = o[6] = 0;
            if (sv.nLights > 0) {
                o[6] = LightSamplerPMF(sv, ctx, 0);
                const LightLiSample ls = LightSampleLi(sv, sv.lights[0], ctx, V2{r[7], r[8]}, lambda, true);
                o[4] = ls.valid ? 1.f : 0.f;
                o[5] = ls.valid && (ls.L[0] != 0 || ls.L[1] != 0 || ls.L[2] != 0 || ls.L[3] != 0) ? 1.f : 0.f;
            }
        });
        f = fopen(lightProbeOut.c_str(), "wb");
        fwrite(out.data(), 4, out.size(), f);
        fclose(f);
        return 0;
    }

    // re-intersection probe (the reference's Triangle.Reintersect / TestReintersectConvex, shapes_test.cpp:156-206, 255-312, run on
    // the scene's primitives one at a time): in = n x {prim, convex, o[3], d[3], seed} floats -> out = n x {hit, rays leaving the
    // intersection point in 1000 random directions that meet the same primitive again, the same for 1000 rays toward random points
    // with tMax = 1}.  The interaction is the one the kernels rebuild from a hit record (HitInteraction), the rays leave it through
    // SpawnRay / SpawnRayTo (OffsetRayOrigin over the interaction's error bounds) as the kernels' rays do.
    if (!reProbeIn.empty()) {
        FILE *f = fopen(reProbeIn.c_str(), "rb");
        if (!f) { perror(reProbeIn.c_str()); return 1; }
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        int n = (int)(sz / (9 * sizeof(float)));
        std::vector<float> in((size_t)n * 9), out((size_t)n * 3);
        if (fread(in.data(), 4, in.size(), f) != in.size()) return 1;
        fclose(f);
        auto hitPrim = [&](int prim, V3 o, V3 d, float tMax, float b[3]) -> bool {
            if (prim < sv.nTriangles) {
                V3 p0, p1, p2;
                TriVerts(sv, prim, &p0, &p1, &p2);
                TriHit h;
                if (!IntersectTriangle(o, d, tMax, p0, p1, p2, &h)) return false;
                b[0] = h.b0; b[1] = h.b1; b[2] = h.b2;
                return true;
            }
            QuadricHit qh;
            if (!QuadricIntersect(sv, prim, o, d, tMax, &qh)) return false;
            b[0] = qh.pObj.x; b[1] = qh.pObj.y; b[2] = qh.pObj.z;
            return true;
        };
        ParallelFor(n, [&](int i) {
            const float *r = &in[(size_t)i * 9];
            float *res = &out[(size_t)i * 3];
            res[0] = res[1] = res[2] = 0;
            const int prim = (int)r[0];
            const bool convex = r[1] != 0;
            const V3 o{r[2], r[3], r[4]}, d{r[5], r[6], r[7]};
            float b[3];
            if (prim < 0 || prim >= sv.nTriangles + sv.nQuadrics || !hitPrim(prim, o, d, WF_INFINITY, b)) return;
            res[0] = 1;
            SurfIntr si;
            HitInteraction(sv, prim, -1, b[0], b[1], b[2], &si, o, d);
            uint64_t st = 0x853c49e6748fea9bull ^ ((uint64_t)(uint32_t)(int)r[8] * 0x9E3779B97F4A7C15ull);
            auto uni = [&]() {   // xorshift64*, 24 bits
                st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
                return (float)((st * 0x2545F4914F6CDD1Dull) >> 40) * (1.f / 16777216.f);
            };
            auto pExp = [&]() { return std::pow(10.f, -8.f + 16.f * uni()); };
            if (r[1] == 2) {   // BilinearPatch.Offset (shapes_test.cpp:451-492): the ray continued in its own direction
                RayOD ro = SpawnRay(si.pi, si.n, d);
                float bb[3];
                if (hitPrim(prim, ro.o, ro.d, WF_INFINITY, bb)) res[1] += 1;
                return;
            }
            for (int j = 0; j < 1000; ++j) {
                V3 w = SampleUniformSphere(V2{uni(), uni()});
                if (convex) w = FaceForward(w, toV(si.n));
                RayOD ro = SpawnRay(si.pi, si.n, w);
                float bb[3];
                if (hitPrim(prim, ro.o, ro.d, WF_INFINITY, bb)) res[1] += 1;
                V3 p2{pExp(), pExp(), pExp()};
                if (convex) p2 = si.pi.mid() + FaceForward(p2 - si.pi.mid(), toV(si.n));
                ro = SpawnRayTo(si.pi, si.n, p2);
                if (hitPrim(prim, ro.o, ro.d, 1.f, bb)) res[2] += 1;
            }
        });
        f = fopen(reProbeOut.c_str(), "wb");
        fwrite(out.data(), 4, out.size(), f);
        fclose(f);
        return 0;
    }

    // stand-alone traversal mode: rays.bin = n x {o[3], d[3], tMax} floats -> hits.bin = n x wf_hit_record
    if (!traceRays.empty()) {
        FILE *f = fopen(traceRays.c_str(), "rb");
        if (!f) { perror(traceRays.c_str()); return 1; }
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        const int rec = traceTimed ? 8 : 7;
        int n = (int)(sz / (rec * sizeof(float)));
        std::vector<float> rays((size_t)n * rec);
        if (fread(rays.data(), sizeof(float), rays.size(), f) != rays.size()) return 1;
        fclose(f);
        std::vector<wf_hit_record> hits(n);
        ParallelFor(n, [&](int i) {
            const float *r = &rays[(size_t)i * rec];
            ArrayStack st;
            ClosestHit ch;
            bool found = traceTimed ? BVHIntersectClosest<true>(sv, V3{r[0], r[1], r[2]}, V3{r[3], r[4], r[5]}, r[6], st, &ch, r[7])
                                    : BVHIntersectClosest(sv, V3{r[0], r[1], r[2]}, V3{r[3], r[4], r[5]}, r[6], st, &ch);
            wf_hit_record &h = hits[i];
            h.prim = found ? ch.prim : -1;
            h.t = found ? ch.h.t : 0; h.b0 = found ? ch.h.b0 : 0; h.b1 = found ? ch.h.b1 : 0; h.b2 = found ? ch.h.b2 : 0;
            h.nodes_visited = ch.nodesVisited; h.tris_tested = ch.trisTested; h.instance = found ? ch.inst : -1;
        });
        f = fopen(traceHits.c_str(), "wb");
        fwrite(hits.data(), sizeof(wf_hit_record), hits.size(), f);
        fclose(f);
        return 0;
    }

    const int n = T.maxQueueSize;
    WorkState ws{};
    ws.maxQueueSize = n;
    ws.pixelsPerPass = n;   // the reference's geometry: one sample index per pass
    ws.samplesPerPass = 1;
    // --strips rank count height: the multi-GPU image partition (wf_set_strips): only the owned scanline strips are rendered
    ws.stripRank = stripRank; ws.stripCount = stripCount; ws.stripHeight = stripHeight; ws.localRows = 0;
    for (int y = 0; y < T.desc.film.pixel_max[1] - T.desc.film.pixel_min[1]; ++y) ws.localRows += stripCount <= 1 || (y / stripHeight) % stripCount == stripRank;
    ws.filterWeight = Alloc<float>(n); ws.pPixel = Alloc<I2>(n);
    ws.lambda = Alloc<F4>(n); ws.lambdaPdf = Alloc<F4>(n); ws.L = Alloc<F4>(n); ws.cameraRayWeight = Alloc<F4>(n);
    ws.samples0 = Alloc<F4>(n); ws.samples1 = Alloc<F4>(n);
    AllocRayQueue(&ws.rq[0], n); AllocRayQueue(&ws.rq[1], n);
    ws.hit = Alloc<F4>(n);
    if (sv.nInstances > 0) ws.hitInst = Alloc<int32_t>(n);
    if (sv.haveMix) ws.mixMat = Alloc<int32_t>(n);
    if (sv.haveSubsurface) { ws.samples2 = Alloc<F4>(n); ws.bssrdfQ = Alloc<BssrdfItem>(n); ws.sssQ = Alloc<SubsurfaceItem>(n); }
    ws.escapedQ = Alloc<int32_t>(n); ws.hitLightQ = Alloc<int32_t>(n);
    for (int m = 0; m < WF_MAT_NTYPES; ++m) ws.matQ[m] = Alloc<int32_t>(T.materialTypePresent[m] ? n : 1);
    ws.sq.o = Alloc<F4>(n); ws.sq.d = Alloc<F4>(n); ws.sq.Ld = Alloc<F4>(n); ws.sq.r_u = Alloc<F4>(n); ws.sq.r_l = Alloc<F4>(n);
    if (sv.haveAnimated) ws.pathTime = Alloc<float>(n);
    if (sv.haveMedia) {
        ws.hitT = Alloc<float>(n); ws.mediumSampleQ = Alloc<int32_t>(n); ws.mediumScatterQ = Alloc<int32_t>(n);
        ws.scatterP = Alloc<F4>(n); ws.sq.medium = Alloc<int32_t>(n);
    }
    ws.counters = Alloc<int32_t>(CNT_COUNT * CNT_STRIDE);
    if (emulateStaleDepth && sv.haveMedia) { g_msStaleDepth = Alloc<int32_t>(n); gThreads = 1; }
    const wf_film &F = T.desc.film;
    const int W = F.pixel_max[0] - F.pixel_min[0], H = F.pixel_max[1] - F.pixel_min[1];
    ws.film = Alloc<double>((size_t)W * H * 4);
    ws.filmSpectral = F.type == WF_FILM_SPECTRAL ? Alloc<double>((size_t)W * H * 2 * F.n_buckets) : nullptr;
    ws.filmGBuffer = F.type == WF_FILM_GBUFFER ? Alloc<wf_gbuffer_pixel>((size_t)W * H) : nullptr;
    if (F.type == WF_FILM_GBUFFER) { ws.vsP = Alloc<F4>(n); ws.vsN = Alloc<F4>(n); ws.vsNs = Alloc<F4>(n); ws.vsDpdx = Alloc<F4>(n); ws.vsDpdy = Alloc<F4>(n); ws.vsAlbedo = Alloc<F4>(n); }
    ws.stats = Alloc<unsigned long long>(129);
    unsigned long long nodesVisited = 0, trisTested = 0, sssProbes = 0, sssExits = 0;
    std::atomic<unsigned long long> shadowNodes{0}, shadowTris{0};


    // --trace-path: the lines of oracle/ref_build/ref_trace.cpp
    auto s4 = [](F4 v) { char b[160]; snprintf(b, sizeof(b), "%a %a %a %a", v.x, v.y, v.z, v.w); return std::string(b); };
    auto traceLines = [](const char *tag, int depth, std::vector<std::pair<int, std::string>> &v) {
        std::sort(v.begin(), v.end());
        for (auto &p : v) printf("d%d %s pix %d %s\n", depth, tag, p.first, p.second.c_str());
    };
    auto traceL = [&](const char *tag, int depth) {
        if (!tracePath) return;
        for (int i = 0; i < n; ++i) printf("d%d L.%s pix %d %s\n", depth, tag, i, s4(ws.L[i]).c_str());
    };
    auto tpRays = [&](const char *tag, int depth, int q) {
        if (!tracePath) return;
        std::vector<std::pair<int, std::string>> v;
        for (int i = 0; i < ws.counters[(CNT_RAY0 + q) * CNT_STRIDE]; ++i) {
            const RayQueueV &r = ws.rq[q];
            char b[1024];
            snprintf(b, sizeof(b), "o %a %a %a d %a %a %a depth %d beta %s r_u %s r_l %s etaScale %a spec %d anyns %d medium %d", r.o[i].x, r.o[i].y, r.o[i].z, r.d[i].x, r.d[i].y,
                     r.d[i].z, r.meta[i].y, s4(r.beta[i]).c_str(), s4(r.r_u[i]).c_str(), s4(r.r_l[i]).c_str(), r.d[i].w, (r.meta[i].z & RAYFLAG_SPECULAR_BOUNCE) ? 1 : 0,
                     (r.meta[i].z & RAYFLAG_ANY_NONSPECULAR) ? 1 : 0, r.meta[i].w >= 0 ? 1 : 0);
            v.emplace_back(r.meta[i].x, b);
        }
        traceLines(tag, depth, v);
    };
    auto traceShadow = [&](const char *tag, int depth) {
        if (!tracePath) return;
        std::vector<std::pair<int, std::string>> v;
        for (int i = 0; i < ws.counters[(CNT_SHADOW) * CNT_STRIDE]; ++i) {
            const ShadowQueueV &r = ws.sq;
            char b[1024];
            snprintf(b, sizeof(b), "o %a %a %a d %a %a %a tMax %a Ld %s r_u %s r_l %s", r.o[i].x, r.o[i].y, r.o[i].z, r.d[i].x, r.d[i].y, r.d[i].z, r.o[i].w, s4(r.Ld[i]).c_str(),
                     s4(r.r_u[i]).c_str(), s4(r.r_l[i]).c_str());
            v.emplace_back(ShadowPixel(r.d[i].w), b);   // (the pixel without the SHADOW_TIME_ZERO flag bit: ADVICE r5)
        }
        traceLines(tag, depth, v);
    };
    auto t0 = std::chrono::steady_clock::now();
    const int maxDepth = T.desc.max_depth;
    if (sampleEnd < 0) sampleEnd = T.spp;
    for (int sampleIndex = sampleBegin; sampleIndex < sampleEnd; sampleIndex += sampleStep) {
        for (int y0 = F.pixel_min[1]; y0 < F.pixel_min[1] + ws.localRows; y0 += T.scanlinesPerPass) {
            ws.counters[(CNT_RAY0) * CNT_STRIDE] = KCameraRayCount(sv, ws, y0, 1);
            ParallelFor(n, [&](int i) { KGenerateCameraRay(sv, ws, i, y0, sampleIndex, sampleStep, 1); });
            ws.stats[0] += ws.counters[(CNT_RAY0) * CNT_STRIDE];
            tpRays("camera", 0, 0);
            const bool dumpNow = !dumpStages.empty() && sampleIndex == sampleBegin && y0 == F.pixel_min[1];
            if (dumpNow) {
                {
                    FILE *g = fopen((dumpStages + "/instances.bin").c_str(), "wb");
                    for (int k = 0; k < sv.nInstances; ++k) { fwrite(sv.instances[k].render_from_instance.m, 4, 16, g); fwrite(sv.instances[k].render_from_instance.mInv, 4, 16, g); }
                    fclose(g);
                }
                FILE *f = fopen((dumpStages + "/camera_rays.bin").c_str(), "wb");
                for (int i = 0; i < ws.counters[(CNT_RAY0) * CNT_STRIDE]; ++i) {
                    F4 o = ws.rq[0].o[i], d = ws.rq[0].d[i];
                    float rec[8] = {(float)ws.rq[0].meta[i].x, o.x, o.y, o.z, d.x, d.y, d.z, o.w};
                    fwrite(rec, 4, 8, f);
                }
                fclose(f);
            }
            for (int depth = 0; true; ++depth) {
                const int cur = depth & 1;
                ws.counters[(CNT_RAY0 + (cur ^ 1)) * CNT_STRIDE] = 0;
                ws.counters[(CNT_ESCAPED) * CNT_STRIDE] = ws.counters[(CNT_HITLIGHT) * CNT_STRIDE] = 0;
                for (int m = 0; m < WF_MAT_NTYPES; ++m) ws.counters[(CNT_MAT0 + m) * CNT_STRIDE] = 0;
                ws.counters[(CNT_MEDIUM_SAMPLE) * CNT_STRIDE] = ws.counters[(CNT_MEDIUM_SCATTER) * CNT_STRIDE] = 0;
                ws.counters[(CNT_BSSRDF) * CNT_STRIDE] = ws.counters[(CNT_SSS) * CNT_STRIDE] = 0;
                const int nRays = ws.counters[(CNT_RAY0 + cur) * CNT_STRIDE];
                ws.stats[1 + depth] += nRays;
                ParallelFor(nRays, [&](int i) { KGenerateRaySamples(sv, ws, cur, i, sampleIndex, sampleStep); });
                if (simulateWaves && nRays > 0) {
                    WaveSim S;
                    std::vector<std::vector<uint8_t>> ev(64);
                    for (int base = 0; base < nRays; base += 64 * 16) {  // every 16th wave
                        int lanes = std::min(64, nRays - base);
                        for (int l = 0; l < lanes; ++l) {
                            F4 o = ws.rq[cur].o[base + l], d = ws.rq[cur].d[base + l];
                            RayEvents(sv, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, WF_INFINITY, &ev[l]);
                        }
                        for (int l = lanes; l < 64; ++l) ev[l].clear();
                        SimulateWave(ev, 64, &S);
                    }
                    {
                        // one wave's share of the queue (~1700 consecutive rays) under the streaming scheme
                        std::vector<std::vector<uint8_t>> run;
                        int cnt = std::min(nRays, 1700);
                        run.resize(cnt);
                        for (int l = 0; l < cnt; ++l) {
                            F4 o = ws.rq[cur].o[nRays / 3 + l < nRays ? nRays / 3 + l : l], d = ws.rq[cur].d[nRays / 3 + l < nRays ? nRays / 3 + l : l];
                            RayEvents(sv, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, WF_INFINITY, &run[l]);
                        }
                        for (int R : {4, 8, 16})
                            for (int LT : {8, 16, 24, 32}) {
                                double it;
                                double c = SimulateStream(run, R, LT, &it);
                                fprintf(stderr, "  stream R=%d LT=%d: cost per 64 rays %.0f (iterations %.1f)\n", R, LT, c / cnt * 64, it / cnt * 64);
                            }
                    }
                    {
                        // the same queue sorted by (direction octant, Morton code of the origin cell): what a ray-binning
                        // pass before the launch would give the fixed-batch kernel
                        std::vector<std::pair<uint64_t, int>> keys(nRays);
                        const float *sb = T.desc.scene_bounds;
                        for (int i = 0; i < nRays; ++i) {
                            F4 o = ws.rq[cur].o[i], d = ws.rq[cur].d[i];
                            uint32_t oct = (d.x < 0) | ((d.y < 0) << 1) | ((d.z < 0) << 2);
                            auto cell = [&](float v, int a) { float t = (v - sb[a]) / (sb[3 + a] - sb[a]); int c = (int)(t * 1024); return (uint32_t)std::min(std::max(c, 0), 1023); };
                            uint64_t m = EncodeMorton3(cell(o.x, 0), cell(o.y, 1), cell(o.z, 2));
                            for (int variant = 0; variant < 1; ++variant) keys[i] = {((uint64_t)oct << 40) | m, i};
                        }
                        std::sort(keys.begin(), keys.end());
                        for (int mode = 0; mode < 2; ++mode) {
                            if (mode == 1) {  // origin cell major, octant minor
                                for (auto &k : keys) k.first = ((k.first & ((1ull << 40) - 1)) >> 9 << 3) | (k.first >> 40);
                                std::sort(keys.begin(), keys.end());
                            }
                            WaveSim S2;
                            std::vector<std::vector<uint8_t>> ev2(64);
                            for (int base = 0; base < nRays; base += 64 * 16) {
                                int lanes = std::min(64, nRays - base);
                                for (int l = 0; l < lanes; ++l) {
                                    int i = keys[base + l].second;
                                    F4 o = ws.rq[cur].o[i], d = ws.rq[cur].d[i];
                                    RayEvents(sv, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, WF_INFINITY, &ev2[l]);
                                }
                                for (int l = lanes; l < 64; ++l) ev2[l].clear();
                                SimulateWave(ev2, 64, &S2);
                            }
                            fprintf(stderr, "  sorted (%s): while-while %.0f (I iters %.1f, T iters %.1f) util %.2f\n", mode == 0 ? "octant, origin" : "origin cell/8, octant",
                                    S2.ww / S2.waves, S2.wwI / S2.waves, S2.wwT / S2.waves, S2.ideal / S2.ww);
                        }
                    }
                    {
                        // capped batches with continuation: a lane that has done K interior steps is suspended (state to a
                        // continuation queue); suspended rays are walked later, 64 at a time, again capped
                        for (int K : {16, 24, 32, 48}) {
                            std::vector<std::vector<uint8_t>> cur_, nextq;
                            for (int base = 0; base < nRays; base += 64 * 16)
                                for (int l = 0; l < std::min(64, nRays - base); ++l) {
                                    F4 o = ws.rq[cur].o[base + l], d = ws.rq[cur].d[base + l];
                                    std::vector<uint8_t> e;
                                    RayEvents(sv, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, WF_INFINITY, &e);
                                    cur_.push_back(std::move(e));
                                }
                            const double nTotal = cur_.size();
                            double cost = 0;
                            int rounds = 0;
                            while (!cur_.empty() && rounds < 12) {
                                nextq.clear();
                                for (size_t base = 0; base < cur_.size(); base += 64) {
                                    std::vector<std::vector<uint8_t>> ev3(64);
                                    for (size_t l = 0; l < 64 && base + l < cur_.size(); ++l) {
                                        auto &e = cur_[base + l];
                                        int nI = 0;
                                        size_t cut = e.size();
                                        for (size_t k = 0; k < e.size(); ++k)
                                            if (e[k] == 0 && ++nI > K) { cut = k; break; }
                                        ev3[l].assign(e.begin(), e.begin() + cut);
                                        if (cut < e.size()) nextq.emplace_back(e.begin() + cut, e.end());
                                    }
                                    WaveSim S3;
                                    SimulateWave(ev3, 64, &S3);
                                    cost += S3.ww + (rounds > 0 ? 200 : 0);   // + restore cost
                                }
                                cost += nextq.size() * 200.0 / 64;   // suspend cost
                                cur_.swap(nextq);
                                ++rounds;
                            }
                            fprintf(stderr, "  capped K=%d: cost per 64 rays %.0f (%d rounds)\n", K, cost / nTotal * 64, rounds);
                        }
                    }
                    fprintf(stderr, "sim depth %d: waves %ld  per wave: ideal %.0f  while-while %.0f (I iters %.1f, T iters %.1f)  unified %.0f  speculative %.0f  | per ray: interior %.1f tris %.1f | util ww %.2f\n",
                            depth, S.waves, S.ideal / S.waves, S.ww / S.waves, S.wwI / S.waves, S.wwT / S.waves, S.unified / S.waves, S.spec / S.waves,
                            S.sumI / S.waves / 64, S.sumT / S.waves / 64, S.ideal / S.ww);
                }
                std::atomic<unsigned long long> nv{0}, nt{0};
                ParallelFor(nRays, [&](int i) {
                    F4 o = ws.rq[cur].o[i], d = ws.rq[cur].d[i];
                    if (ws.pathTime) ws.pathTime[ws.rq[cur].meta[i].x] = o.w;   // the path's time, for the shadow rays this depth spawns
                    ArrayStack st;
                    ClosestHit ch;
                    bool found = BVHIntersectClosest<true>(sv, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, WF_INFINITY, st, &ch, o.w);   // (o.w: the ray's time, for AnimatedPrimitive)
                    nv += ch.nodesVisited; nt += ch.trisTested;
                    if (getenv("WF_DEBUG_PIXEL") && depth == 0 && ws.rq[cur].meta[i].x == atoi(getenv("WF_DEBUG_PIXEL")) && found) {
                        fprintf(stderr, "dbg hit prim %d inst %d t %a b %a %a %a\n  o %a %a %a d %a %a %a\n", ch.prim, ch.inst, ch.h.t, ch.h.b0, ch.h.b1, ch.h.b2, o.x, o.y, o.z, d.x, d.y, d.z);
                        if (ch.inst >= 0) {
                            float tI = WF_INFINITY; V3 oI, dI;
                            InstanceRay(sv.instances[ch.inst], V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, &tI, &oI, &dI);
                            V3 p0, p1, p2; TriVerts(sv, ch.prim, &p0, &p1, &p2);
                            fprintf(stderr, "  oI %a %a %a dI %a %a %a\n  p0 %a %a %a p1 %a %a %a p2 %a %a %a\n", oI.x, oI.y, oI.z, dI.x, dI.y, dI.z, p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z);
                            const float *m = &sv.instances[ch.inst].render_from_instance.m[0][0];
                            fprintf(stderr, "  m"); for (int k = 0; k < 32; ++k) fprintf(stderr, " %a", m[k]); fprintf(stderr, "\n");
                        }
                    }
                    KAfterClosestHit(sv, ws, cur, i, found, ch.prim, ch.inst, ch.h.t, ch.h.b0, ch.h.b1, ch.h.b2);
                });
                nodesVisited += nv; trisTested += nt;
                if (dumpNow && depth == 0) {
                    // same record as oracle/ref_build/ref_stages.cpp::dumpMat
                    FILE *f = fopen((dumpStages + "/mat_items.bin").c_str(), "wb");
                    for (int m = 1; m < WF_MAT_NTYPES; ++m)
                        for (int k = 0; k < ws.counters[(CNT_MAT0 + m) * CNT_STRIDE]; ++k) {
                            int i = ws.matQ[m][k];
                            F4 h = ws.hit[i], d = ws.rq[cur].d[i], o = ws.rq[cur].o[i];
                            SurfIntr si;
                            HitInteraction(sv, (int)FloatToBits(h.x), HitInst(sv, ws, i), h.y, h.z, h.w, &si, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z});
                            V3 wo_ = IntrWo(sv, (int)FloatToBits(h.x), HitInst(sv, ws, i), V3{-d.x, -d.y, -d.z});
                            float rec[28] = {(float)m, (float)ws.rq[cur].meta[i].x, si.pi.lo.x, si.pi.lo.y, si.pi.lo.z, si.pi.hi.x, si.pi.hi.y, si.pi.hi.z,
                                             si.n.x, si.n.y, si.n.z, si.ns.x, si.ns.y, si.ns.z, si.dpdus.x, si.dpdus.y, si.dpdus.z, wo_.x, wo_.y, wo_.z,
                                             si.uv.x, si.uv.y, si.dpdu.x, si.dpdu.y, si.dpdu.z, si.dpdv.x, si.dpdv.y, si.dpdv.z};
                            fwrite(rec, 4, 28, f);
                        }
                    fclose(f);
                    {
                        FILE *g = fopen((dumpStages + "/camera_diffs.bin").c_str(), "wb");
                        const wf_camera &C = sv.camera;
                        fwrite(C.minPosDifferentialX, 4, 3, g); fwrite(C.minPosDifferentialY, 4, 3, g);
                        fwrite(C.minDirDifferentialX, 4, 3, g); fwrite(C.minDirDifferentialY, 4, 3, g);
                        fwrite(C.renderFromCamera.m, 4, 16, g); fwrite(C.renderFromCamera.mInv, 4, 16, g);
                        fclose(g);
                    }
                    f = fopen((dumpStages + "/mat_diffs.bin").c_str(), "wb");
                    for (int m = 1; m <= 2; ++m)
                        for (int k = 0; k < ws.counters[(CNT_MAT0 + m) * CNT_STRIDE]; ++k) {
                            int i = ws.matQ[m][k];
                            F4 h = ws.hit[i], d = ws.rq[cur].d[i], o = ws.rq[cur].o[i];
                            SurfIntr si;
                            HitInteraction(sv, (int)FloatToBits(h.x), HitInst(sv, ws, i), h.y, h.z, h.w, &si, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z});
                            V3 dpdx, dpdy;
                            ApproximateDpDxy(sv, si.pi.mid(), si.n, &dpdx, &dpdy);
                            float rec[7] = {(float)ws.rq[cur].meta[i].x, dpdx.x, dpdx.y, dpdx.z, dpdy.x, dpdy.y, dpdy.z};
                            fwrite(rec, 4, 7, f);
                            if (getenv("WF_DEBUG_PIXEL") && ws.rq[cur].meta[i].x == atoi(getenv("WF_DEBUG_PIXEL"))) {
                                V3 pc = XfInvPointM(sv.camera.renderFromCamera.mInv, si.pi.mid());
                                const float (*mm)[4] = sv.camera.renderFromCamera.m;
                                N3 n = si.n;
                                N3 nc{mm[0][0] * n.x + mm[1][0] * n.y + mm[2][0] * n.z, mm[0][1] * n.x + mm[1][1] * n.y + mm[2][1] * n.z, mm[0][2] * n.x + mm[1][2] * n.y + mm[2][2] * n.z};
                                fprintf(stderr, "dbg pc %.9g %.9g %.9g nc %.9g %.9g %.9g p %.9g %.9g %.9g\n", pc.x, pc.y, pc.z, nc.x, nc.y, nc.z, si.pi.mid().x, si.pi.mid().y, si.pi.mid().z);
                            }
                        }
                    fclose(f);
                    f = fopen((dumpStages + "/samples.bin").c_str(), "wb");
                    for (int i = 0; i < nRays; ++i) {
                        int pi = ws.rq[cur].meta[i].x;
                        F4 a = ws.samples0[pi], b = ws.samples1[pi];
                        float rec[8] = {(float)pi, a.x, a.y, a.z, a.w, b.x, b.y, b.z};
                        fwrite(rec, 4, 8, f);
                    }
                    fclose(f);
                }
                if (tracePath)
                    printf("d%d after-closest rays %d escaped %d hitlight %d medium %d next_pre %d\n", depth, nRays, ws.counters[(CNT_ESCAPED) * CNT_STRIDE],
                           ws.counters[(CNT_HITLIGHT) * CNT_STRIDE], ws.counters[(CNT_MEDIUM_SAMPLE) * CNT_STRIDE], ws.counters[(CNT_RAY0 + (cur ^ 1)) * CNT_STRIDE]);
                if (tracePath && sv.haveMedia)
                    for (int k = 0; k < ws.counters[(CNT_MEDIUM_SAMPLE) * CNT_STRIDE]; ++k) {
                        const int i = ws.mediumSampleQ[k];
                        printf("d%d msample pix %d tMax %a time %a\n", depth, ws.rq[cur].meta[i].x, ws.hitT[i], ws.rq[cur].o[i].w);
                    }
                if (sv.haveMedia) {
                    // SampleMediumInteraction, integrator.cpp:416 (K5, then K6 unless this is the last depth)
                    ParallelFor(ws.counters[(CNT_MEDIUM_SAMPLE) * CNT_STRIDE], [&](int i) { KSampleMediumInteraction(sv, ws, cur, i); });
                    if (depth != maxDepth)
                        ParallelFor(ws.counters[(CNT_MEDIUM_SCATTER) * CNT_STRIDE], [&](int i) { KSampleMediumScattering(sv, ws, cur, i); });
                }
                if (sv.haveMedia) traceL("medium", depth);
                ParallelFor(ws.counters[(CNT_ESCAPED) * CNT_STRIDE], [&](int i) { KHandleEscaped(sv, ws, cur, i); });
                ParallelFor(ws.counters[(CNT_HITLIGHT) * CNT_STRIDE], [&](int i) { KHandleEmissive(sv, ws, cur, i); });
                traceL("emitted", depth);
                if (depth == maxDepth) break;
                // in the order of the reference's Material::Types (base/material.h:36-39; ForEachType, integrator.cpp EvaluateMaterialsAndBSDFs): the image
                // does not depend on it, but the ORDER of the next ray queue does, and with it which medium-sample slot a ray gets at the next
                // depth — what --emulate-stale-medium-depth reproduces (the reference's unwritten MediumSampleWorkItem::depth)
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_COATED_DIFFUSE) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_COATED_DIFFUSE>(sv, ws, cur, i, true); });
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_COATED_CONDUCTOR) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_COATED_CONDUCTOR>(sv, ws, cur, i, true); });
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_CONDUCTOR) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_CONDUCTOR>(sv, ws, cur, i, true); });
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_DIELECTRIC) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_DIELECTRIC>(sv, ws, cur, i, true); });
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_DIFFUSE) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_DIFFUSE>(sv, ws, cur, i, true); });
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_DIFFUSE_TRANSMISSION) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_DIFFUSE_TRANSMISSION>(sv, ws, cur, i, true); });
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_HAIR) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_HAIR>(sv, ws, cur, i, true); });
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_MEASURED) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_MEASURED>(sv, ws, cur, i, true); });
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_SUBSURFACE) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_SUBSURFACE>(sv, ws, cur, i, true); });
                ParallelFor(ws.counters[(CNT_MAT0 + WF_MAT_THIN_DIELECTRIC) * CNT_STRIDE], [&](int i) { KEvalMaterial<WF_MAT_THIN_DIELECTRIC>(sv, ws, cur, i, true); });
                auto traceShadowRays = [&]() {  // TraceShadowRays, integrator.cpp:575-586
                    const int nShadow = ws.counters[(CNT_SHADOW) * CNT_STRIDE];
                    if (sv.haveMedia)
                        ParallelFor(nShadow, [&](int i) {
                            const float time = ShadowTime<true>(ws, ws.sq.d[i].w);   // the shadow ray's (and its respawned segments') time
                            KTraceTransmittance<true>(sv, ws, i, [&](V3 o, V3 d, float tMax, int *prim, int *inst, float *b0, float *b1, float *b2) {
                                ArrayStack st;
                                ClosestHit ch;
                                bool found = BVHIntersectClosest<true>(sv, o, d, tMax, st, &ch, time);
                                if (found) { *prim = ch.prim; *inst = ch.inst; *b0 = ch.h.b0; *b1 = ch.h.b1; *b2 = ch.h.b2; }
                                return found;
                            });
                        });
                    else
                    ParallelFor(nShadow, [&](int i) {
                        F4 o = ws.sq.o[i], d = ws.sq.d[i];
                        ArrayStack st;
                        int v = 0, t = 0;
                        const float time = ShadowTime<true>(ws, d.w);   // ShadowRayWorkItem.ray.time
                        bool occluded = BVHIntersectAny<true>(sv, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, o.w, st, &v, &t, time);
                        shadowNodes += (unsigned long long)v; shadowTris += (unsigned long long)t;
                        KRecordShadowRay(ws, i, occluded);
                    });
                    ws.stats[65 + depth] += nShadow;
                    ws.counters[(CNT_SHADOW) * CNT_STRIDE] = 0;
                };
                tpRays("next", depth, cur ^ 1);
                traceShadow("shadow", depth);
                traceShadowRays();
                traceL("shadowed", depth);
                if (sv.haveSubsurface) {
                    if (tracePath) printf("d%d bssrdf-items %d\n", depth, ws.counters[(CNT_BSSRDF) * CNT_STRIDE]);
                    // SampleSubsurface, integrator.cpp:431 -> wavefront/subsurface.cpp:18-203
                    ParallelFor(ws.counters[(CNT_BSSRDF) * CNT_STRIDE], [&](int i) { KSubsurfaceProbe(sv, ws, i); });
                    ParallelFor(ws.counters[(CNT_SSS) * CNT_STRIDE], [&](int i) { ArrayStack st; KIntersectOneRandom<true>(sv, ws, i, st); });
                    ParallelFor(ws.counters[(CNT_SSS) * CNT_STRIDE], [&](int i) { KSubsurfaceScatter(sv, ws, cur, i); });
                    sssProbes += ws.counters[(CNT_SSS) * CNT_STRIDE];
                    for (int i = 0; i < ws.counters[(CNT_SSS) * CNT_STRIDE]; ++i) sssExits += ws.sssQ[i].reservoirPDF != 0;
                    traceShadowRays();
                    if (tracePath) printf("d%d sss-items %d\n", depth, ws.counters[(CNT_SSS) * CNT_STRIDE]);
                    tpRays("next+sss", depth, cur ^ 1);
                    traceL("sss", depth);
                }
                if (tracePath && ws.counters[(CNT_RAY0 + (cur ^ 1)) * CNT_STRIDE] == 0) break;
            }
            ParallelFor(n, [&](int i) { KUpdateFilm(sv, ws, i, 1); });
        }
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (fatalWord) {
        // the reference's LOG_FATAL inside a kernel body (shapes.cpp:736-760: a sample drawn from an emissive curve) aborts the process
        fprintf(stderr, "Fatal: %s\n", FatalMessage(fatalWord));
        return 1;
    }

    unsigned long long rays = ws.stats[0];
    for (int d = 0; d < 64; ++d) rays += ws.stats[1 + d] * (d > 0) + ws.stats[65 + d];
    // stats[1+0] are the camera rays again (indirect[0] counts the depth-0 queue); do not double count
    if (!opt.quiet)
        fprintf(stderr, "wf_cpu: %dx%d %d spp, %d threads: %.3f s, %.3f Msamples/s, %.3f Mray/s (nodes %llu tris %llu)\n", W, H, T.spp,
                gThreads, secs, (double)W * H * T.spp / secs / 1e6, rays / secs / 1e6, nodesVisited, trisTested);
    printf("{\"seconds\": %.6f, \"width\": %d, \"height\": %d, \"spp\": %d, \"threads\": %d, \"rays\": %llu, \"camera_rays\": %llu, \"indirect_rays\": [", secs, W, H,
           T.spp, gThreads, rays, ws.stats[0]);
    for (int d = 0; d < 64; ++d) printf("%s%llu", d ? ", " : "", ws.stats[1 + d]);
    {   // the statistics pbrt --stats prints for its BVHAggregate (cpu/aggregates.cpp:27-31,577; shapes.cpp:318,339)
        unsigned long long interior = 0, leaf = 0, leafPrims = 0;
        for (int k = 0; k < T.desc.n_bvh_nodes; ++k) {
            if (T.desc.bvh_nodes[k].nprims > 0) { ++leaf; leafPrims += T.desc.bvh_nodes[k].nprims; } else ++interior;
        }
        printf("], \"subsurface_probes\": %llu, \"subsurface_exits\": %llu", sssProbes, sssExits);
        printf(", \"bvh_interior_nodes\": %llu, \"bvh_leaf_nodes\": %llu, \"bvh_leaf_prims\": %llu, \"bvh_nodes_visited\": %llu, \"tri_tests\": %llu, \"closest_nodes_visited\": %llu, \"closest_tri_tests\": %llu",
               interior, leaf, leafPrims, nodesVisited + shadowNodes.load(), trisTested + shadowTris.load() + g_pdfTriTests.load(), nodesVisited, trisTested);
        printf(", \"shadow_rays\": [");
    }
    if (false) printf("], \"shadow_rays\": [");
    for (int d = 0; d < 64; ++d) printf("%s%llu", d ? ", " : "", ws.stats[65 + d]);
    printf("]}\n");

    if (!dumpFilm.empty()) {
        FILE *f = fopen(dumpFilm.c_str(), "wb");
        fwrite(ws.film, sizeof(double), (size_t)W * H * 4, f);
        fclose(f);
    }
    // RGBFilm::GetPixelRGB (film.h:258-275) + GetImage (film.cpp:533-565), float output
    if (F.type != WF_FILM_RGB) {   // SpectralFilm / GBufferFilm::GetImage + WriteImage: the multi-channel .exr
        std::vector<std::string> names;
        std::vector<float> chans;
        if (F.type == WF_FILM_SPECTRAL) SpectralFilmImage(F, ws.film, ws.filmSpectral, W, H, T.saveFP16, &names, &chans);
        else GBufferFilmImage(F, ws.film, ws.filmGBuffer, W, H, T.saveFP16, &names, &chans);
        if (!T.imageFile.empty()) WriteEXRChannels(T.imageFile, names, chans.data(), W, H, T.saveFP16);
        return 0;
    }
    std::vector<float> rgb((size_t)W * H * 3);
    FilmToRGB(F, ws.film, W, H, rgb.data(), T.saveFP16);
    if (!T.imageFile.empty()) WriteFilmImage(T, T.imageFile, rgb, W, H);
    return 0;
}
