// wf_backend.hip — libwfhip.so: the HIP/CDNA4 (gfx950) back end behind include/wf_abi.h.
//
// What the reference does with one generic kernel template instantiated per lambda (gpu/util.h:77-114)
// plus OptiX programs (gpu/optix/optix.cu), this file does with named, hand-written kernels:
//   k_gen_camera_rays / k_gen_ray_samples / k_handle_escaped / k_handle_emissive / k_eval_material<M> /
//   k_update_film            grid-stride "for all queued" kernels (sizes are read on the device; the host
//                            never reads a queue size, as in wavefront/workqueue.h:118-137)
//   k_intersect_closest      BVH closest hit, replaces OptiX __raygen__findClosest + closest-hit programs
//   k_intersect_shadow       BVH any hit + RecordShadowRayResult, replaces __raygen__shadow
//   k_reset                  the one-thread "Reset queues" / stats kernels of integrator.cpp:357-397
// Traversal keeps the node stack in LDS (one column per lane, STACK_LDS entries; deeper levels spill to a
// per-lane HBM column) — no private-memory (scratch) arrays anywhere in the hot loop.
// Queue pushes are wave-aggregated (wf_kernels.h: QueueAlloc): one atomic per wave per destination queue.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the only fused operations are the explicit
// fma calls of the restated arithmetic, as in the reference's CPU build).
#include <hip/hip_runtime.h>
#include <array>

#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

// This unit is compiled with -ftrivial-auto-var-init=zero (Makefile, BACKENDFLAGS; DESIGN 4.1 "Locals that start as undef"): the walk
// loops below and in wf_traverse.h spill a third less when no local starts as undef.  The shared stage code keeps the default: its
// out-of-line shape / texture functions grow by 15 VGPRs when initialised, which takes the general-primitive walk kernels (GEN = 2)
// from 165 to 180 registers and from 3 waves to 2 (-9 / -12 % on the spec scene plus one sphere).
#pragma clang attribute push(__attribute__((uninitialized)), apply_to = variable(is_local))
#include "../common/wf_kernels.h"
#include "../common/wf_kat.h"
#pragma clang attribute pop
#include "wf_traverse.h"

using namespace wf;

// ---------------------------------------------------------------------------------------------
// errors
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code ? code : -1;
}
#define HIPCHK(call)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return fail((int)e_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
constexpr int BLOCK = 256;
constexpr int STACK_LDS = 24;   // LDS stack entries per lane: 24 x 4 B x 256 lanes = 24 KiB per workgroup
constexpr int STACK_MAX = 64;   // nodesToVisit[64], cpu/aggregates.cpp:538
constexpr int MAX_GRID = 256 * 8;  // 256 CUs x up to 8 resident 256-thread workgroups

struct SpillArea { int *base; int rows; int *dbg; F4 *save; };   // the context's stackSpill rows + the debug words (kernel argument of the production traversal)

struct wf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    SceneView svHost{};          // device pointers inside; passed to every kernel by value (kernarg)
    const SceneView *svDev = nullptr;  // the same struct in device memory, for the out-of-line callbacks of the traversal kernels
    WorkState ws{};
    int maxQueueSize = 0;
    int *stackSpill = nullptr;   // [rows][MAX_GRID*BLOCK], rows from the trees' depths (wf_scene_upload)
    int spillRows = 0;
    int *dbgWords = nullptr;     // [4] spilled stack entries, stack-overflow flag, inline near-tie re-traces, cursor path taken (wf_debug_counters)
    F4 *walkSave = nullptr;      // [4][MAX_GRID*BLOCK]: every lane's render-space walk constants, reloaded when it leaves an instance (LdsStackT::loadWorld)
    SpillArea spillArea() const { return SpillArea{stackSpill, spillRows, dbgWords, walkSave}; }
    FastBVH fast{};              // production traversal layout (wf_traverse.h); built at upload
    bool fastOk = false;         // false: leaf sizes > 16 -> only the reference-order kernels are used
    hipStream_t stream2 = nullptr;   // the near-tie re-trace runs here, beside the routing pass and the next stage's sample generation
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    bool retracePending = false, deferJoin = false;
    int overlapRetrace = 1;      // WF_OVERLAP_RETRACE=0: everything on one stream
    // the transmittance wavefront: -1 = for two-level scenes only (default), 1 = always, 0 = never (WF_TR_WAVEFRONT).  Measured on the
    // cloud-like spec scene (14 triangles, 512^3 grid; gpurun_out/r3j_bench_cloud_tr*.json): per-lane loop 27.9 ms per 16 spp, wavefront
    // 31.4 ms (begin 3.3 + trace 3.7 + segment 24.1 + rest 0.3) — the time is the ratio tracking through the grid, not the walk, and the
    // per-lane loop keeps its state in registers; with object instances the per-lane alternative is the reference-order walk (1 wave / SIMD)
    int trWavefront = -1;
    bool cursorDirty[3] = {false, false, false};   // the closest-hit / any-hit / medium-sample work cursor has been used since a k_reset last zeroed it
    int mediumGrid = 512;        // resident workgroups of the persistent k_medium_sample
    bool mediumLean = false;     // every medium is homogeneous or a non-emissive uniform grid: k_medium_sample<true>
    int matStreams = 0;          // WF_MAT_STREAMS=0: the material kernels of one depth one after the other on the render stream
    hipStream_t matStream[WF_MAT_NTYPES] = {};
    hipEvent_t evMatFork = nullptr, evMatJoin[WF_MAT_NTYPES] = {};
    bool portalLights = false;   // the scene has a portal infinite light (k_handle_escaped<RARE>)
    bool leanShade = false;      // the scene qualifies for the lean shade kernels (SceneLean: set at upload)
    // ... per MATERIAL TYPE since round 6: a type none of whose materials sits on a quadric / patch / curve keeps its lean shade kernel
    // when such shapes appear elsewhere in the scene (their hits are items of other types' queues)
    bool leanType[WF_MAT_NTYPES] = {};
    bool matSplit = true;        // the material stage as two kernels per type (WF_MAT_SPLIT=0 with a MATFUSED build: the one-kernel stage)
    bool rareLights = false;     // the scene has a light type only the VARIANT 2 material kernels sample (portal infinite lights)
    int genMode = 0;             // general-primitive strength of the traversal kernels: 0 triangles only, 1 simple alpha, 2 anything but curves and alpha on quadrics, 3 anything (see GeneralPrims)
    int persistentGrid = 1024;   // resident workgroups for the persistent traversal kernels (closest-hit variant of the scene)
    int persistentGridShadow = 1024;
    // TWO-CLASS TRAVERSAL (round 6): the triangle kernels (genTri = 0 | 1, in their handing-over variants) walk every ray, the general
    // kernels (genMode >= 2) only the rays handed to deferQ; persistentGrid / persistentGridShadow are then the triangle kernels' grids
    bool deferGeneral = false;
    int genTri = 0;
    bool animFast = false;       // the scene's AnimatedPrimitives are walked by the production kernels' ANIM variants (round 6; genMode <= 1 only)
    int persistentGridGen = 1024, persistentGridShadowGen = 1024;
    static bool splitRouteWanted() { return true; }
    // ray-coherence pass (SortRayQueue): bit 0 sorts the ray queue before the closest-hit launch of depth >= 1, bit 1 the shadow queue
    int raySort = 0;
    int cursorChunk = 2;         // 64-ray batches a closest-hit wave takes per cursor fetch (WF_CURSOR_CHUNK sets both)
    int cursorChunkShadow = 2;   // ... an any-hit wave (round 6, with the descent scheduling, 10 M-triangle scene: closest-hit 27.7 ms at 1, 26.7 at 3, 27.0 at 4, 27.4 at 8, 29.9 at 16;
                                 // any-hit 12.15 at 1, 12.16 at 3, 12.4 at 4, 13.0 at 8: profiles/r06_cursor_chunk_ab_sm16.txt)
                                 // (-3 %), but on a 30 k-triangle scene one fetch per 64 rays is 83 atomics/us on one counter: the kernel's bound
    int splitRoute = 2;          // WF_SPLIT_ROUTE: 0 = the closest-hit walk routes its hits per workgroup (KRouteHitBlock inside the walk);
                                 // 1 = walk without the workgroup barrier + k_route_hits; 2 (default) = 1 + waves draw their rays from a shared cursor
    int sortMin = 4096;
    int sortOriginBits = 6, sortDirBits = 4;  // per axis of the origin grid / per axis of the octahedral direction map
    float sceneMin[3] = {0, 0, 0}, sceneMax[3] = {0, 0, 0};  // bounds of the top-level BVH
    float sortBase[3] = {0, 0, 0}, sortScale[3] = {0, 0, 0};
    uint32_t *sortKeys[2] = {nullptr, nullptr}, *sortVals[2] = {nullptr, nullptr};
    void *sortTemp = nullptr;
    size_t sortTempBytes = 0;
    RayQueueV rqTmp{};
    ShadowQueueV sqTmp{};
    int32_t *probeCursor = nullptr;
    bool matPresent[WF_MAT_NTYPES] = {};
    int W = 0, H = 0;
    int maxDepth = 5;
    float sceneBounds[6] = {};
    bool sceneLoaded = false, queuesAllocated = false;
    // profiling (gpu/util.cpp:136-209)
    int profile = 0;             // 0 off, 1 every launch, 2 traversal kernels only
    struct Ev { std::string name; hipEvent_t a, b; };
    std::vector<Ev> events;
    std::vector<hipEvent_t> eventPool;
    int passY0 = INT_MIN;        // first scanline of the band of the last wf_gen_camera_rays
    int passStep = 1, passSamples = 1;  // wf_set_pass_samples: sample-index stride and sample slots used by the current pass
    bool pixelMajor = true;      // items of a pass ordered pixel by pixel (WorkState::slotStride); WF_PIXEL_MAJOR=0: sample by sample
    bool countTraversal = false;
    bool traceLaunch = false;    // WF_TRACE_LAUNCH=1: print every launch and synchronise after it (debugging)
};

template <typename T>
static int devAlloc(wf_ctx *c, T **p, size_t n) {
    void *d = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(hipMalloc(&d, bytes));
    HIPCHK(hipMemsetAsync(d, 0, bytes, c->stream));
    c->allocs.push_back(d);
    *p = (T *)d;
    return 0;
}
template <typename T>
static int devUpload(wf_ctx *c, const T **p, const T *src, size_t n) {
    T *d = nullptr;
    if (int e = devAlloc(c, &d, n)) return e;
    if (n && src) HIPCHK(hipMemcpyAsync(d, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *p = d;
    return 0;
}

template <typename T>
static int devUpload(wf_ctx *c, wf::GPtr<const T> *p, const T *src, size_t n) {   // (a SceneView table pointer, wf_scene.h)
    const T *d = nullptr;
    if (int e = devUpload(c, &d, src, n)) return e;
    *p = d;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// kernels
// Round 4: a kernel may read the scene view through the pointer to its device-resident copy (sv.self) instead of from its by-value
// argument — what took 100 spilled SGPRs out of the material kernels (wf_mat.hip, WF_MAT_SV_PTR).  Measured per kernel on the spec scene
// (gpurun_out/r04ad, 16 spp, same box): camera rays 4.03 -> 3.32 ms, route hits 1.81 -> 1.71; closest-hit 36.8 -> 39.0 (worse: it hardly
// touches the view, and the indirection costs), escaped 3.02 -> 3.16, the others unchanged — so only the first two use it.
template <bool VIA_SELF>
__device__ inline const SceneView &SvOf(const SceneView &a) { return VIA_SELF ? *a.self : a; }
__global__ void __launch_bounds__(BLOCK) k_reset(WorkState ws, unsigned mask, int statSlot, int statCounter) {
    // mask bit i: zero counters[(i) * CNT_STRIDE].  statSlot >= 0: stats[statSlot] += counters[(statCounter) * CNT_STRIDE] first.
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (statSlot >= 0) ws.stats[statSlot] += (unsigned long long)ws.counters[(statCounter) * CNT_STRIDE];
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < CNT_COUNT && ((mask >> threadIdx.x) & 1u)) {
        // the items a material queue held go to stats[129 + type] before the counter is zeroed (wf_material_items_download: the item
        // counts of the material stage's roofline) — the same for the medium-sample queue at stats[129 + WF_MAT_NTYPES]
        const int t = (int)threadIdx.x;
        if (t >= CNT_MAT0 && t <= CNT_MEDIUM_SAMPLE) ws.stats[129 + (t - CNT_MAT0)] += (unsigned long long)ws.counters[t * CNT_STRIDE];
        ws.counters[t * CNT_STRIDE] = 0;
    }
}

__global__ void __launch_bounds__(BLOCK) k_sample_tops(const SceneView svArg, WorkState ws, int y0, int dim0) {
    const SceneView &sv = SvOf<false>(svArg);
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < 5 * ws.pixelsPerPass; i += gridDim.x * BLOCK) KSampleTops(sv, ws, i, y0, dim0);
}
// ANIM: the instance for a moving camera (AnimatedTransform::Interpolate per ray); scenes with a static camera launch the other one
template <bool ANIM>
__global__ void __launch_bounds__(BLOCK) k_gen_camera_rays(const SceneView svArg, WorkState ws, int y0, int sampleBase, int sampleStep, int nSamples) {
    const SceneView &sv = SvOf<true>(svArg);
    if (sv.camera.type != WF_CAMERA_REALISTIC && blockIdx.x == 0 && threadIdx.x == 0) ws.counters[(CNT_RAY0) * CNT_STRIDE] = KCameraRayCount(sv, ws, y0, nSamples);
    const bool useTops = ws.sampleTops != nullptr;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < ws.maxQueueSize; i += gridDim.x * BLOCK)
        KGenerateCameraRay<ANIM>(sv, ws, i, y0, sampleBase, sampleStep, nSamples, useTops);
}

__global__ void __launch_bounds__(BLOCK) k_gen_ray_samples(const SceneView svArg, WorkState ws, int cur, int sampleBase, int sampleStep, int topsDepth) {
    const SceneView &sv = SvOf<false>(svArg);
    const int n = ws.counters[(CNT_RAY0 + cur) * CNT_STRIDE];
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) KGenerateRaySamples(sv, ws, cur, i, sampleBase, sampleStep, topsDepth);
}

// LDS short stack with HBM spill: one column per lane ([entry][lane] so a wave's same-depth accesses hit
// 64 consecutive banks).  The array lives at file scope so every access is a ds_read/ds_write (LDS address
// space), never a flat access.
__shared__ int g_sstack[STACK_LDS * BLOCK];
struct LdsStack {
    int *spill;   // &stackSpill[global thread], stride = total threads
    int spillStride;
    int n;
    __device__ void push(int v) {
        if (n < STACK_LDS) g_sstack[n * BLOCK + threadIdx.x] = v;
        else spill[(size_t)(n - STACK_LDS) * spillStride] = v;
        ++n;
    }
    __device__ int pop() {
        --n;
        int v = g_sstack[(n < STACK_LDS ? n : 0) * BLOCK + threadIdx.x];  // always a ds_read
        if (__builtin_expect(n >= STACK_LDS, 0)) v = spill[(size_t)(n - STACK_LDS) * spillStride];
        return v;
    }
    __device__ bool empty() const { return n == 0; }
};

__device__ inline unsigned long long waveSum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// ANIM: the scene has animated shapes / instances (AnimatedPrimitive, round 5): the walk interpolates their transformation at the ray's time.
// Such scenes are walked by these reference-order kernels only (no production tree: ctx->fastOk is false for them).
template <bool COUNT, bool ANIM = false>
__global__ void __launch_bounds__(BLOCK) k_intersect_closest(const SceneView sv, WorkState ws, int cur, int *stackSpill) {
    const int n = ws.counters[(CNT_RAY0 + cur) * CNT_STRIDE];
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    unsigned long long nv = 0, nt = 0, nh = 0, nr = 0;
    for (int i = gtid; i < n; i += stride) {
        F4 o = ws.rq[cur].o[i], d = ws.rq[cur].d[i];
        // the path's time, for the shadow rays this depth spawns (a subsurface exit continues at time 0: subsurface.cpp:70)
        if (ANIM) ws.pathTime[ws.rq[cur].meta[i].x] = o.w;
        ClosestHit ch;
        st.n = 0;
        bool found = BVHIntersectClosest<ANIM>(sv, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, WF_INFINITY, st, &ch, o.w);
        KAfterClosestHit(sv, ws, cur, i, found, ch.prim, ch.inst, ch.h.t, ch.h.b0, ch.h.b1, ch.h.b2);
        if (COUNT) { nv += ch.nodesVisited; nt += ch.trisTested; nh += found; nr += 1; }
    }
    if (COUNT) {
        nv = waveSum(nv); nt = waveSum(nt); nh = waveSum(nh); nr = waveSum(nr);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&ws.trav[0], nr); atomicAdd(&ws.trav[1], nv); atomicAdd(&ws.trav[2], nt); atomicAdd(&ws.trav[3], nh);
        }
    }
}

template <bool COUNT, bool ANIM = false>
__global__ void __launch_bounds__(BLOCK) k_intersect_shadow(const SceneView sv, WorkState ws, int *stackSpill) {
    const int n = ws.counters[(CNT_SHADOW) * CNT_STRIDE];
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    unsigned long long nv = 0, nt = 0, nu = 0, nr = 0;
    for (int i = gtid; i < n; i += stride) {
        F4 o = ws.sq.o[i], d = ws.sq.d[i];
        int v = 0, t = 0;
        st.n = 0;
        const float time = ShadowTime<ANIM>(ws, d.w);
        bool occluded = BVHIntersectAny<ANIM>(sv, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, o.w, st, &v, &t, time);
        KRecordShadowRay(ws, i, occluded);
        if (COUNT) { nv += v; nt += t; nu += !occluded; nr += 1; }
    }
    if (COUNT) {
        nv = waveSum(nv); nt = waveSum(nt); nu = waveSum(nu); nr = waveSum(nr);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&ws.trav[4], nr); atomicAdd(&ws.trav[5], nv); atomicAdd(&ws.trav[6], nt); atomicAdd(&ws.trav[7], nu);
        }
    }
}

// ---- production traversal: persistent waves over QNode/LeafTri with the tree top in LDS (wf_traverse.h) ----
__shared__ int g_tstack[TSTACK * TBLOCK];
__shared__ U4 g_top[QNODE_U4 * (TOP_NODES > 0 ? TOP_NODES : 1)];   // (WF_TOP_NODES=0: no LDS copy of the tree top — every node fetch is a plain global load)
// the render-space ray of every lane (o.xyz, d.xyz), for the instance transitions and the alpha test of the two-level / general
// variants: re-reading it from the queue (round 2) put an L2 round trip in front of every transition — eight per ray on the spec scene
__shared__ float g_ray[6 * TBLOCK];
__device__ inline void StoreWorldRay(V3 o, V3 d) {
    g_ray[0 * TBLOCK + threadIdx.x] = o.x; g_ray[1 * TBLOCK + threadIdx.x] = o.y; g_ray[2 * TBLOCK + threadIdx.x] = o.z;
    g_ray[3 * TBLOCK + threadIdx.x] = d.x; g_ray[4 * TBLOCK + threadIdx.x] = d.y; g_ray[5 * TBLOCK + threadIdx.x] = d.z;
}
// WF_SAVE_WORLD == 2: nine of the lane's render-space walk constants (a, bk of WalkSetSlab and the shear's S: what costs a reciprocal or an
// IEEE division to rebuild), kept in LDS from the ray's start for every exit from an instance — 9 KiB per workgroup: with them the
// two-level kernels use 39 KiB, four workgroups still fit a CU's 160 KiB
__shared__ float g_save[9 * TBLOCK];
// the ray's time, for the kernels of scenes with animated primitives (written by their fetch functors, read at instance entries)
__shared__ float g_time[TBLOCK];
__device__ inline V3 WorldRayO() { return V3{g_ray[0 * TBLOCK + threadIdx.x], g_ray[1 * TBLOCK + threadIdx.x], g_ray[2 * TBLOCK + threadIdx.x]}; }
__device__ inline V3 WorldRayD() { return V3{g_ray[3 * TBLOCK + threadIdx.x], g_ray[4 * TBLOCK + threadIdx.x], g_ray[5 * TBLOCK + threadIdx.x]}; }
// the HBM spill path of the stack is out of line so that the compiler cannot merge it with the LDS path
// into a pointer select (which turns every pop into a flat load)
__device__ __attribute__((noinline)) int SpillRead(const int *p) { return *p; }
__device__ __attribute__((noinline)) void SpillWrite(int *p, int v) { *p = v; }
// The node stack of the production walk: a ring of TSTACK entries per lane in LDS holding the NEWEST entries; when it is full the
// OLDEST entry moves to the lane's HBM column (round 3).  Until round 2 the entries above TSTACK went to HBM instead: a walk deep in
// a two-level tree (20-30 entries) then paid an HBM store and an HBM load on nearly every push / pop — rocprofv3 WRITE_SIZE showed
// 230 B per closest-hit ray and 394 B per shadow ray for kernels whose results are 28 B and 16 B.  With the bottom of the stack
// spilled, an entry crosses the LDS / HBM boundary at most once in each direction, and the entries that do are the ones pushed near
// the root, which are popped last.
// dbg (null = off): [0] += entries spilled, [1] |= 1 when a push would have run past the lane's spill rows (the entry is dropped and
// wf_sync reports the overflow: ADVICE r2, "the traversal spill stack can overflow on unbalanced trees").
struct LdsStackT {
    int *spill;   // &stackSpill[global thread], stride = total threads
    int spillStride;
    int n;        // entries on the stack
    int lo;       // entries [0, lo) are in the HBM column, [lo, n) in the LDS ring (n - lo <= TSTACK)
    int rows;     // capacity of the HBM column
    int *dbg;
    // The render-space constants of the lane's ray (slab constants on the top-level grid, the triangle test's shear: 16 dwords), written
    // once when the ray starts and RELOADED whenever the lane leaves an instance (round 6).  Until round 5 ExitInstance recomputed them —
    // three IEEE divisions, three v_rcp, ~95 VALU instructions of a walk that is bound by VALU issue at a quarter of its lanes — eight
    // times per ray on the spec scene.  Four 16-byte planes per lane, lane-major: a wave reads 1 KiB per instruction, mostly from L2.
    F4 *save = nullptr;
    __device__ void saveWorld(const RayWalk &w) const {
        const size_t S = (size_t)spillStride;
        save[0] = F4{w.a.x, w.a.y, w.a.z, w.bn.x};
        save[S] = F4{w.bn.y, w.bn.z, w.af.x, w.af.y};
        save[2 * S] = F4{w.af.z, w.bf.x, w.bf.y, w.bf.z};
        save[3 * S] = F4{BitsToFloat((uint32_t)w.sh.kz), w.sh.Sx, w.sh.Sy, w.sh.Sz};
    }
    __device__ void saveWorldLds(const float a[3], const float bk[3], const RayShear &sh) const {
        const int t = threadIdx.x;
        g_save[0 * TBLOCK + t] = a[0]; g_save[1 * TBLOCK + t] = a[1]; g_save[2 * TBLOCK + t] = a[2];
        g_save[3 * TBLOCK + t] = bk[0]; g_save[4 * TBLOCK + t] = bk[1]; g_save[5 * TBLOCK + t] = bk[2];
        g_save[6 * TBLOCK + t] = sh.Sx; g_save[7 * TBLOCK + t] = sh.Sy; g_save[8 * TBLOCK + t] = sh.Sz;
    }
    __device__ void loadWorld(RayWalk &w, V3 oW, V3 dW) const {
#if WF_SAVE_WORLD == 2
        const int t = threadIdx.x;
        const float a[3] = {g_save[0 * TBLOCK + t], g_save[1 * TBLOCK + t], g_save[2 * TBLOCK + t]};
        const float bk[3] = {g_save[3 * TBLOCK + t], g_save[4 * TBLOCK + t], g_save[5 * TBLOCK + t]};
        w.o = oW;
        w.sh.kz = MaxComponentIndex(Abs(dW));   // (MakeRayShear's permutation; its three divisions are the saved S)
        w.sh.Sx = g_save[6 * TBLOCK + t]; w.sh.Sy = g_save[7 * TBLOCK + t]; w.sh.Sz = g_save[8 * TBLOCK + t];
        WalkSlabFromAB(w, a, bk);
        return;
#endif
        const size_t S = (size_t)spillStride;
        const F4 p0 = save[0], p1 = save[S], p2 = save[2 * S], p3 = save[3 * S];
        w.o = oW;
        w.a = V3{p0.x, p0.y, p0.z}; w.bn = V3{p0.w, p1.x, p1.y};
        w.af = V3{p1.z, p1.w, p2.x}; w.bf = V3{p2.y, p2.z, p2.w};
        w.sh.kz = (int)FloatToBits(p3.x); w.sh.Sx = p3.y; w.sh.Sy = p3.z; w.sh.Sz = p3.w;
        // (the v_perm selectors follow the direction's sign, which is the sign of `a`: WalkSetSlab)
        w.selx = (FloatToBits(p0.x) >> 31) ? 0x01000302u : 0x03020100u;
        w.sely = (FloatToBits(p0.y) >> 31) ? 0x01000302u : 0x03020100u;
        w.selz = (FloatToBits(p0.z) >> 31) ? 0x01000302u : 0x03020100u;
    }
    static constexpr int MASK = TSTACK - 1;
    static_assert((TSTACK & (TSTACK - 1)) == 0, "WF_TSTACK must be a power of two (ring indexing)");
    __device__ void reset() { n = 0; lo = 0; }
    __device__ void push(int v) {
        if (__builtin_expect(n - lo == TSTACK, 0)) {
            if (__builtin_expect(lo >= rows, 0)) { if (dbg) atomicOr(dbg + 1, 1); return; }
            SpillWrite(&spill[(size_t)lo * spillStride], g_tstack[(lo & MASK) * TBLOCK + threadIdx.x]);
            if (dbg) atomicAdd(dbg, 1);
            ++lo;
        }
        g_tstack[(n & MASK) * TBLOCK + threadIdx.x] = v;
        ++n;
    }
    // room for k more entries at once (k <= 3: the interior step's pushes), so that the pushes themselves need no test (WF_PUSH_RESERVE)
    __device__ void reserve(int k) {
        while (__builtin_expect(n - lo + k > TSTACK, 0)) {
            if (__builtin_expect(lo >= rows, 0)) { if (dbg) atomicOr(dbg + 1, 1); return; }
            SpillWrite(&spill[(size_t)lo * spillStride], g_tstack[(lo & MASK) * TBLOCK + threadIdx.x]);
            if (dbg) atomicAdd(dbg, 1);
            ++lo;
        }
    }
    __device__ void pushIf(int v, bool wanted) {
        g_tstack[(n & MASK) * TBLOCK + threadIdx.x] = v;
        n += wanted ? 1 : 0;
    }
    __device__ void pushReserved(int v) {
        g_tstack[(n & MASK) * TBLOCK + threadIdx.x] = v;
        ++n;
    }
    __device__ int pop() {
        --n;
        int v = g_tstack[(n & MASK) * TBLOCK + threadIdx.x];  // always a ds_read
        if (__builtin_expect(n < lo, 0)) { lo = n; v = SpillRead(&spill[(size_t)n * spillStride]); }
        return v;
    }
    __device__ bool empty() const { return n == 0; }
};

__device__ inline void LoadTreeTop(const FastBVH &bvh) {
    const U4 *src = reinterpret_cast<const U4 *>(bvh.nodes);
    if constexpr (TOP_NODES == 0) return;
    const int n = QNODE_U4 * (bvh.nNodes < TOP_NODES ? bvh.nNodes : TOP_NODES);
    for (int i = threadIdx.x; i < n; i += TBLOCK) g_top[i] = src[i];
    __syncthreads();
}
// a node's 16-byte words: tree top from LDS, the rest from global memory.  (Measured: splitting the two fetch paths
// into separate loops so that each is a pure ds_read / global_load costs more in extra wave serialisation than the
// merged flat load does: 0.57 ms vs 0.45 ms per launch.)
#ifndef WF_FETCH_ALL
#define WF_FETCH_ALL 0
#endif
__device__ inline void FetchNode(const FastBVH &bvh, int node, U4 *n) {
    if constexpr (TOP_NODES == 0) {
        const U4 *p = reinterpret_cast<const U4 *>(bvh.nodes + node);
        for (int k = 0; k < QNODE_U4; ++k) n[k] = p[k];
#if WF_FETCH_ALL
        // the child references (the node's last 16 bytes) are only read when a child is hit, and the compiler sinks their load behind the
        // slab tests: a second dependent round trip per step.  Pinned here, the four loads of a node issue together.
        asm volatile("" : "+v"(n[QNODE_U4 - 1].x), "+v"(n[QNODE_U4 - 1].y), "+v"(n[QNODE_U4 - 1].z), "+v"(n[QNODE_U4 - 1].w));
#endif
        return;
    }
    const U4 *p = node < TOP_NODES ? g_top + QNODE_U4 * node : reinterpret_cast<const U4 *>(bvh.nodes + node);
    for (int k = 0; k < QNODE_U4; ++k) n[k] = p[k];
}

// One batch of TBLOCK rays per workgroup iteration: ray index = thread index within the batch (no cursor
// atomic).  Waves walk independently ("while-while": all lanes descend interior nodes until every one of
// them sits at a leaf or is done, then the leaves are processed together); `finish` runs once per batch for
// the whole workgroup, so its queue pushes are block-aggregated (BlockAlloc).
// what the general-primitive traversal variants call back into (wf_traverse.h LeafStep): the alpha test and the spheres
// The general-primitive work of the traversal kernels comes in two strengths (template parameter GEN of the kernels):
//   GEN = 1  "simple alpha": every alpha texture of the scene is a constant, a uv-mapped image map (not EWA-filtered) or a bilerp
//            (no texture graph: EvalFloatTextureSimple) and there are no quadrics.  The test is a real call into a small out-of-line function
//            that reads the device-resident SceneView: the walk keeps its registers, the call is paid only when an
//            alpha-tested triangle is actually hit.  (Foliage cut-outs are image maps: this is the san-miguel case.)
//   GEN = 2  anything else (texture graphs as alpha, spheres / disks / cylinders): evaluated inline as before — an
//            out-of-line callee for the full texture graph needs 194 VGPRs, which would become the kernel's own count.
// The ray is NOT kept in registers for either: the origin of the space being walked is w.o, the direction is re-fetched
// from the queue (and taken into the instance's space) when such a primitive is hit.
__device__ __attribute__((noinline)) bool AlphaTestSimpleP(const SceneView *svp, int tri, float b0, float b1, float b2, float ox, float oy, float oz,
                                                            float dx, float dy, float dz) {
    // AlphaTestPasses (common/wf_shapes.h) for the alpha textures genMode 1 admits: a constant, or a uv-mapped bilerp / image map
    // looked up without a footprint.  The same arithmetic as EvalFloatTextureSimple on those inputs (UVMapping::Map, textures.h:76-98;
    // FloatImageTexture::Evaluate, FloatBilerpTexture::Evaluate), without the interaction point the other mappings need and without
    // the calls into the general filter: the test runs for every candidate hit of a cut-out triangle.
    const SceneView &sv = *svp;
    const wf_mesh &mesh = sv.meshes[sv.triMesh[tri]];
    if (mesh.alpha_tex < 0) return true;
    const wf_texture &t = sv.textures[mesh.alpha_tex];
    float a;
    if (t.type == WF_TEX_FLOAT_CONSTANT) a = t.f0;
    else {
        const auto v = sv.triIndices + 3 * (size_t)tri;
        V2 uv0{0, 0}, uv1{1, 0}, uv2{1, 1};
        if (mesh.flags & WF_MESH_HAS_UV) {
            if ((const ShadeTri *)sv.shadeTris != nullptr) {   // the triangle's de-indexed record (wf_scene.h): one 24-byte gather instead of an index triple + three
                const float *q = sv.shadeTris[tri].uv;
                uv0 = V2{q[0], q[1]}; uv1 = V2{q[2], q[3]}; uv2 = V2{q[4], q[5]};
            } else { uv0 = LoadUV(sv, v[0]); uv1 = LoadUV(sv, v[1]); uv2 = LoadUV(sv, v[2]); }
        }
        const V2 uv{b0 * uv0.x + b1 * uv1.x + b2 * uv2.x, b0 * uv0.y + b1 * uv1.y + b2 * uv2.y};
        const V2 st{t.map[0] * uv.x + t.map[2], t.map[1] * uv.y + t.map[3]};
        if (t.type == WF_TEX_FLOAT_IMAGE) {
            const float val = t.f0 * MIPFilterFloatZeroP(sv.tableData, sv.texImages + t.i0, st.x, 1 - st.y);
            a = t.f1 != 0 ? fmax(0.f, 1 - val) : val;
        } else {
            const float v00 = t.f0, v01 = t.f1, v10 = t.map[10], v11 = t.map[11];
            a = (1 - st.x) * (1 - st.y) * v00 + st.x * (1 - st.y) * v10 + (1 - st.x) * st.y * v01 + st.x * st.y * v11;
        }
    }
    if (!(a < 1)) return true;
    float u = (a <= 0) ? 1.f : HashToFloat(Hash6f(V3{ox, oy, oz}, V3{dx, dy, dz}));
    return !(u > a);
}
// ---- TWO-CLASS TRAVERSAL (round 6) ----------------------------------------------------------------------------------------------
// Until round 5 ONE quadric, patch or curve anywhere in a scene moved every ray onto the general-primitive walk kernels (GEN >= 2: the
// shape intersectors inline, 165-260 VGPRs, a separate near-tie re-trace launch): the 10 M-triangle scene plus one sphere rendered 1.56x
// slower.  Now the triangle kernels (GEN 0 / 1) walk such a scene first in their DEFER variants (template value 4 + GEN): a walk that
// meets a leaf entry of the other class (LeafTri c.z == 3) stops and hands its ray to `deferQ`; a second, small launch of the general
// kernel walks only those rays, from the start.  A ray that is not handed over never had such a primitive's leaf box within its
// pruning bound — a superset of the primitives that could have been its hit or a near-tie partner of its hit (the wide band of a pair
// that involves such a shape only matters to candidates NEARER than the triangle hit: wf_traverse.h) — so both classes of rays end
// with the result the general kernel alone computes.  The reference keeps such shapes in acceleration structures of their own too
// (gpu/optix/aggregate.cpp:916-1025: one GAS per shape class under the root IAS).
// (kernel template value = strength 0 .. 3 | 4: hands general primitives over | 8: ANIMATED instances, see EnterInstance<ANIM>)
constexpr int GenBase(int g) { return g & 3; }
constexpr bool GenDefer(int g) { return (g & 4) != 0; }
constexpr bool GenAnim(int g) { return (g & 8) != 0; }
template <typename Fetch, int GEN, bool DEFER = false, bool ANIM = false>
struct GeneralPrims {
    static constexpr bool pairBands = GEN >= 2;   // quadrics / patches / curves in the scene: the near-tie band depends on the pair (wf_traverse.h)
    static constexpr bool deferGeneral = DEFER;
    const SceneView &sv;
    const FastBVH &bvh;
    const RayWalk &w;
    const Fetch &fetch;
    int idx;
    __device__ V3 dir() const {
        V3 d = WorldRayD();
        if (w.curInst >= 0) {
            wf_instance moving;
            d = XfVector3(InstanceAt<ANIM>(*bvh.sv, bvh.instances[w.curInst], g_time[threadIdx.x], &moving).render_from_instance.mInv, d);  // = InstanceRay's direction
        }
        return d;
    }
    __device__ bool accept(int prim, float b0, float b1, float b2) const {
        const V3 d = dir();
        if constexpr (GEN == 1) return AlphaTestSimpleP(bvh.sv, prim, b0, b1, b2, w.o.x, w.o.y, w.o.z, d.x, d.y, d.z);
        else return AlphaTestPasses(sv, prim, b0, b1, b2, w.o, d);
    }
    __device__ bool sphere(int prim, float tMax, QuadricHit *qh) const {
        if constexpr (GEN == 1) return false;
        else return QuadricIntersect<GEN == 3>(sv, prim, w.o, dir(), tMax, qh);
    }
    __device__ void exact(RayWalk &wm) const { WalkMakeExact(bvh, wm, WorldRayO(), WorldRayD()); }
};
// the two-level walk of a scene without alpha cut-outs or quadrics: only the lazy instance transition's hook (wf_traverse.h)
template <bool DEFER = false>
struct InstOnlyPrims {
    static constexpr bool pairBands = false;
    static constexpr bool deferGeneral = DEFER;
    const FastBVH &bvh;
    __device__ bool accept(int, float, float, float) const { return true; }
    __device__ bool sphere(int, float, QuadricHit *) const { return false; }
    __device__ void exact(RayWalk &wm) const { WalkMakeExact(bvh, wm, WorldRayO(), WorldRayD()); }
};
struct DeferOnlyPrims {   // the one-level triangle walk of a scene that also holds primitives of the other class
    static constexpr bool pairBands = false;
    static constexpr bool deferGeneral = true;
    __device__ bool accept(int, float, float, float) const { return true; }
    __device__ bool sphere(int, float, QuadricHit *) const { return false; }
    __device__ void exact(RayWalk &) const {}
};
// The part of a "while-while" iteration that follows the interior descent: every lane sits at a leaf run, at an instance transition
// (an instance entry popped from the stack, or the NODE_EXIT marker) or is done.  Leaves are processed first; the transitions —
// ~400 instructions each (the reference's interval-arithmetic ray transform, two WalkSetRay with three IEEE divisions each), eight
// per ray on the spec scene, about half of the walk's VALU work — are PARKED until WF_TRANS_BATCH lanes of the wave wait for one, or
// until no lane has anything else to do: a parked lane loses nothing (its wave-mates' steps would have been issued with its lane
// masked anyway), and the expensive code then runs with several times the lanes.  (0: run every transition in the iteration it is
// popped in, as in round 2.)
#ifndef WF_WALK_STATS
#define WF_WALK_STATS 0   // (diagnostic builds: WalkStats below)
#endif
#ifndef WF_TRANS_Q
#define WF_TRANS_Q 2   // spec scene, 16 spp, same box (profiles/r06_transition_parking_ab_sm16.txt): closest / any-hit 28.2 / 12.4 ms at 0, 27.6 / 12.1 at 2, 28.0 / 12.2 at 4
#endif
#ifndef WF_TRANS_BATCH
#define WF_TRANS_BATCH 6   // spec scene, 16 spp, same box (gpurun_out/r3f_ab_sm16.txt): closest / any-hit 60.1 / 22.0 ms at 0, 56.5 / 20.2 at 6, 58.4 / 20.5 at 12, 62.4 / 21.1 at 24
#endif
__device__ inline bool AtTransition(int node) { return node < 0 && node != NODE_NONE && (node == NODE_EXIT || (int)((~(unsigned)node) >> 4) >= INST_FIRST); }
template <bool ANY, int GENX, bool INST, typename Fetch>
__device__ inline int LeafPhase(const SceneView &sv, const FastBVH &bvh, RayWalk &w, LdsStackT &st, const Fetch &fetch, int idx) {
    constexpr int GEN = GenBase(GENX);
    constexpr bool DF = GenDefer(GENX), ANIM = GenAnim(GENX);
    if constexpr (INST) {
        bool tr = AtTransition(w.node);
        // (round 4, WF_LAZY_INST) a lane that sits at a leaf while its ray state is not exact for the space it walks (RayWalk::lazy)
        // owes WalkMakeExact — the reference's interval-arithmetic ray transform, the shear, the slab constants again — before its
        // leaf can be processed: PARKED like a transition, so that this code too runs with several lanes at once (inside the leaf
        // loop, lane by lane, the lazy transition LOST: closest 49.4 vs 45.0 ms, any-hit 26.1 vs 23.0)
        const bool owes = WF_LAZY_INST && w.node != NODE_NONE && !tr && WF_LAZY_GET(w) != 0;
        if (w.node < 0 && w.node != NODE_NONE && !tr && !owes) {   // (w.node >= 0: a lane the descent left at an interior node, WF_SCHED_Q)
            if constexpr (GEN > 0) LeafStep<ANY, true, true>(bvh, w, st, GeneralPrims<Fetch, GEN, DF, ANIM>{sv, bvh, w, fetch, idx});
            else LeafStep<ANY, false, true>(bvh, w, st, InstOnlyPrims<DF>{bvh});
        }
        if constexpr (WF_TRANS_BATCH > 0) {
            tr = AtTransition(w.node);
            const int nT = __popcll(__ballot(tr || owes));
            if (nT == 0) return 0;
#if WF_TRANS_Q > 0
            {
                // (round 6) ... and, like the descent (WalkDescend), until the lanes that wait for a transition are at least WF_TRANS_Q / 4 of
                // the lanes that have nodes and leaves to visit: the transition is the walk's most expensive code (~400 instructions)
                const int nO = __popcll(__ballot(w.node != NODE_NONE && !tr && !owes));
                if (nO > 0 && (nT < WF_TRANS_BATCH || 4 * nT < WF_TRANS_Q * nO)) return 0;
            }
#else
            if (nT < WF_TRANS_BATCH && __any(w.node != NODE_NONE && !tr && !owes)) return 0;   // parked: the others still have nodes and leaves to visit
#endif
        } else tr = AtTransition(w.node);
        if (owes) WalkMakeExact(bvh, w, WorldRayO(), WorldRayD());   // (its leaf is processed in the next iteration)
        else if (tr) {
            const V3 o = WorldRayO(), d = WorldRayD();
#if WF_FUSE_EXIT_ENTER
            if (w.node == NODE_EXIT) ExitInstance(bvh, w, st, o, d);
            if (IsInstanceEntry(w.node)) EnterInstance<ANIM>(bvh, w, st, o, d, (int)((~(unsigned)w.node) >> 4) - INST_FIRST, ANIM ? g_time[threadIdx.x] : 0.f);   // (also the entry an exit has just popped)
#else
            if (w.node == NODE_EXIT) ExitInstance(bvh, w, st, o, d);
            else EnterInstance<ANIM>(bvh, w, st, o, d, (int)((~(unsigned)w.node) >> 4) - INST_FIRST, ANIM ? g_time[threadIdx.x] : 0.f);
#endif
        }
        return WF_WALK_STATS ? __popcll(__ballot(tr)) : 0;
    } else if (w.node < 0 && w.node != NODE_NONE) {
        if constexpr (GEN > 0) LeafStep<ANY, true>(bvh, w, st, GeneralPrims<Fetch, GEN, DF>{sv, bvh, w, fetch, idx});
        else if constexpr (DF) LeafStep<ANY, false, false>(bvh, w, st, DeferOnlyPrims{});
        else LeafStep<ANY>(bvh, w, st);
    }
    return 0;
}

// ---- near-tie resolution inside the production walk (round 3) -------------------------------------------------------------------
// A ray the production walk marked as a near-tie (wf_traverse.h) is walked once more in the REFERENCE's order — BVHAggregate::Intersect
// over the reference-layout 32-byte nodes (cpu/aggregates.cpp:529-579), the exact Bounds3::IntersectP, the reference's accept rule —
// starting from tMax = t* + twice the band instead of infinity (see k_closest_retrace).  Until round 2 this was a separate launch whose
// duration was the latency of its longest single walk on a 388-VGPR kernel (1.5-3 ms per depth on the 10 M-triangle scene: a tenth
// of the closest-hit stage); now the lane does it itself at the end of its batch, in this out-of-line function — triangles, simple
// alpha cut-outs and object instances only (the GEN <= 1 kernels; scenes with quadrics / curves / texture-graph alpha keep the second
// launch), so that it fits the walk's own register budget.  Cost: the few waves that hold a marked lane run one short walk at 1/64
// lane utilisation (a marked ray's bound is tight: it descends to its hit and little else).
struct RefHit { int prim, inst; float t, b0, b1, b2; uint32_t route; };
template <int GEN, bool TOP, bool ANIM = false>
__device__ inline __attribute__((always_inline)) bool RefOrderTree(const SceneView *svp, int root, V3 o, V3 d, float *tMaxIO, LdsStackT &st, RefHit *out, float time = 0) {
    const SceneView &sv = *svp;
    float tMax = *tMaxIO;
    bool hitAny = false;
    const int base = st.n;
    const V3 invDir{1 / d.x, 1 / d.y, 1 / d.z};
    const int negMask = int(invDir.x < 0) | (int(invDir.y < 0) << 1) | (int(invDir.z < 0) << 2);
    const RayShear sh = MakeRayShear(d);
    int cur = root;
    while (true) {
        const wf_bvh_node *node = &sv.bvhNodes[cur];
        if (BoxIntersectP(node->bmin, node->bmax, o, tMax, invDir, negMask)) {
            if (node->nprims > 0) {
                for (int i = 0; i < node->nprims; ++i) {
                    const int tri = sv.bvhPrims[node->offset + i];
                    if constexpr (TOP) {
                        if (tri >= sv.nTriangles + sv.nQuadrics) {  // TransformedPrimitive::Intersect (cpu/primitive.cpp:112-125)
                            const int inst = tri - sv.nTriangles - sv.nQuadrics;
                            wf_instance moving;
                            const wf_instance &in = InstanceAt<ANIM>(sv, sv.instances[inst], time, &moving);   // (AnimatedPrimitive::Intersect, cpu/primitive.cpp:140-153)
                            float tI = tMax;
                            V3 oI, dI;
                            InstanceRay(in, o, d, &tI, &oI, &dI);
                            if (RefOrderTree<GEN, false>(svp, sv.instanceDefs[in.def].bvh_root, oI, dI, &tI, st, out)) {
                                out->inst = inst;
                                tMax = tI;
                                hitAny = true;
                            }
                            continue;
                        }
                    }
                    // (a quadric / patch / curve — only in the scenes of the two-class traversal — cannot decide a near tie between the
                    //  triangles of a ray that was not handed to the general walk: it lies beyond their band)
                    if (tri >= sv.nTriangles) continue;
                    V3 p0, p1, p2;
                    TriVerts(sv, tri, &p0, &p1, &p2);
                    TriHit h;
                    if (IntersectTriangleSheared(o, sh, tMax, p0, p1, p2, &h, true)) {
                        if (GEN == 0 || AlphaTestSimpleP(svp, tri, h.b0, h.b1, h.b2, o.x, o.y, o.z, d.x, d.y, d.z)) {
                            out->prim = tri;
                            if (TOP) out->inst = -1;
                            out->t = h.t; out->b0 = h.b0; out->b1 = h.b1; out->b2 = h.b2;
                            tMax = h.t;
                            hitAny = true;
                        }
                    }
                }
                if (st.n == base) break;
                cur = st.pop();
            } else if ((negMask >> node->axis) & 1) {
                st.push(cur + 1);
                cur = node->offset;
            } else {
                st.push(node->offset);
                cur = cur + 1;
            }
        } else {
            if (st.n == base) break;
            cur = st.pop();
        }
    }
    *tMaxIO = tMax;
    return hitAny;
}
template <int GEN, bool ANIM = false>
__device__ inline __attribute__((always_inline)) RefHit RetraceRefOrder(const SceneView *svp, float ox, float oy, float oz, float dx, float dy, float dz, float tBound,
                                                           int *spill, int spillStride, int rows, int *dbg, float time = 0) {
    LdsStackT st{spill, spillStride, 0, 0, rows, dbg};
    RefHit out{-1, -1, 0, 0, 0, 0, 0};
    float tMax = tBound;
    RefOrderTree<GEN, true, ANIM>(svp, 0, V3{ox, oy, oz}, V3{dx, dy, dz}, &tMax, st, &out, time);
    if (out.prim >= 0) {
        // the routing code BuildFastBVH stores per LeafTri (EnqueueWorkAfterIntersection, intersect.h:48-156)
        const wf_mesh &mesh = svp->meshes[svp->triMesh[out.prim]];
        out.route = mesh.material >= 0 ? (uint32_t)svp->materials[mesh.material].type | (mesh.first_light >= 0 ? 16u : 0u) : 32u;
    }
    if (dbg) atomicAdd(dbg + 2, 1);
    return out;
}
// which kernel variants resolve their near-ties themselves
constexpr bool RetraceInline(int gen) { return gen <= 1; }

// -DWF_WALK_STATS (diagnostic builds only): per-phase counts of the closest-hit (slot 0) and any-hit (slot 1) production walks — how often a
// wave ran the interior step / the leaf step / an instance transition / a refill, and how many lanes were active each time — summed
// over the launch into 64-bit words behind the debug words (wf_sync prints them under WF_DEBUG_DRAIN).  What the round-6 scheduling
// decisions were checked against: profiles/r06_walk_phase_stats.txt.
struct WalkStats {
    unsigned long long c[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // iterations / lanes of: interior, leaf, transition, refill
    __device__ void add(int phase, bool active) {
#if WF_WALK_STATS
        const int n = __popcll(__ballot(active));
        if (n) { c[2 * phase] += 1; c[2 * phase + 1] += (unsigned long long)n; }
#endif
    }
    __device__ void flush(int *dbg, int slot) {
#if WF_WALK_STATS
        if (dbg && (threadIdx.x & 63) == 0)
            for (int k = 0; k < 8; ++k) atomicAdd(reinterpret_cast<unsigned long long *>(dbg + 8) + slot * 8 + k, c[k]);
#endif
    }
};
// Does the wave take another interior step?  WF_SCHED_Q = 0: while ANY lane sits at an interior node (the "while-while" loop of rounds
// 1-5: the descent ends when its LAST lane has reached a leaf, the other lanes wait masked off).  WF_SCHED_Q = q > 0 (round 6): only while
// the lanes at interior nodes are at least q/4 of the lanes that wait at a leaf — otherwise the leaves are processed
// first and the few lanes still descending go on in the next round, together with the lanes whose leaves sent them back into the tree.
// Which lanes step when changes no result: every lane's own sequence of visits is the same.
#ifndef WF_SCHED_Q
#define WF_SCHED_Q 3   // spec scene, 16 spp, same box (profiles/r06_descent_scheduling_ab_sm16.txt): closest / any-hit 35.9 / 16.4 ms at 0, 31.5 / 13.4 at 2, 31.6 / 13.4 at 4, 31.9 / 13.8 at 8; second box (r06_tree_top_global_loads_and_knobs_ab_sm16.txt): 31.6 / 13.9 at 1, 31.0 / 13.4 at 3, 31.3 / 13.5 at 4
#endif
#ifndef WF_SCHED_Q_ANY
#define WF_SCHED_Q_ANY WF_SCHED_Q   // the any-hit walks' own threshold (measured: see DESIGN 4.1)
#endif
template <bool INST, bool ANY = false>
__device__ inline bool WalkDescend(const RayWalk &w) {
    constexpr int Q = ANY ? WF_SCHED_Q_ANY : WF_SCHED_Q;
    if constexpr (Q == 0) return __any(w.node >= 0);
    const int nI = __popcll(__ballot(w.node >= 0));
    if (nI == 0) return false;
    // (lanes at an instance transition do not count: LeafPhase may leave them parked, and a round must always make progress)
    const int nL = __popcll(__ballot(w.node < 0 && w.node != NODE_NONE && !(INST && AtTransition(w.node))));
    return 4 * nI >= Q * nL;
}
template <bool ANY, int GEN, bool INST = false, typename Fetch, typename Finish>
__device__ inline void BatchTrace(const SceneView &sv, const FastBVH &bvh, int n, LdsStackT &st, Fetch fetch, Finish finish, int *cursor = nullptr, int chunk = 4) {
    LoadTreeTop(bvh);
    // cursor != nullptr (only when `finish` has no workgroup barrier): every wave takes its next `chunk` x 64 rays from a shared
    // cursor, so a wave with short walks serves more of the queue and the launch ends without a tail of waves that drew long ones
    // (one returning atomic per chunk x 64 rays: a counter sustains ~88 of them per microsecond)
    // (a queue shorter than two rounds of the resident grid is dealt statically: every wave starts at once, no atomics)
    if (n < 2 * (int)gridDim.x * TBLOCK) cursor = nullptr;
    if (cursor && st.dbg && threadIdx.x == 0 && blockIdx.x == 0) atomicOr(st.dbg + 3, 1);
    int chunkBase = 0, chunkSub = chunk;
    for (int base = blockIdx.x * TBLOCK; true; base += gridDim.x * TBLOCK) {
        int idx;
        if (cursor) {
            if (chunkSub == chunk) {
                int b = 0;
                if ((threadIdx.x & 63) == 0) b = atomicAdd(cursor, 64 * chunk);
                chunkBase = __shfl(b, 0);
                chunkSub = 0;
            }
            if (chunkBase + chunkSub * 64 >= n) break;
            idx = chunkBase + chunkSub * 64 + (threadIdx.x & 63);
            ++chunkSub;
        } else {
            if (base >= n) break;
            idx = base + threadIdx.x;
        }
        const bool valid = idx < n;
        RayWalk w;
        w.node = NODE_NONE;
        w.prim = -1;
        w.route = 0;
        w.tMax = 0;
        w.b0 = w.b1 = w.b2 = 0;
        w.inst = w.curInst = -1;
        if (valid) {
            V3 o, d;
            float tMax;
            fetch(idx, &o, &d, &tMax);
            float sa[3], sb[3];
            WalkInit(bvh, w, o, d, tMax, sa, sb);
            if constexpr (INST || GEN > 0) StoreWorldRay(o, d);
            if constexpr (INST && WF_SAVE_WORLD == 1) st.saveWorld(w);
            if constexpr (INST && WF_SAVE_WORLD == 2) st.saveWorldLds(sa, sb, w.sh);
            st.reset();
        }
        // (Measured and dropped: parking a lane's first leaf and descending on speculatively — 11 % slower; continuous
        // per-lane refill ("streaming": idle lanes take the next rays of the wave's run, leaf / interior step chosen per
        // iteration) — 28 % fewer VALU instructions at 53 % instead of 31 % active lanes, but 5x the vector-L1 line
        // accesses (7.2e8 vs 1.45e8 per 8.5 M-ray launch): lanes at unrelated depths of the tree no longer share
        // cache lines, and the walk becomes L1-bound: 33 ms vs 27 ms per 16 spp.  DESIGN.md §4.)
        while (__any(w.node != NODE_NONE)) {
            while (WalkDescend<INST, ANY>(w)) {
                if (w.node >= 0) {
                    U4 nd[QNODE_U4];
                    FetchNode(bvh, w.node, nd);
                    InteriorStep<!ANY>(bvh, w, st, nd);
                }
            }
            LeafPhase<ANY, GEN, INST>(sv, bvh, w, st, fetch, idx);
        }
        if constexpr (!ANY && RetraceInline(GenBase(GEN))) {
            if (valid && WalkAmbiguous(w)) {
                V3 o, d;
                float t0;
                fetch(idx, &o, &d, &t0);
                const float tB = __builtin_fminf(2 * WalkBound(bvh, WalkT(w)) - WalkT(w), t0);
                const RefHit rh = RetraceRefOrder<GenBase(GEN), GenAnim(GEN)>(bvh.sv, o.x, o.y, o.z, d.x, d.y, d.z, tB, st.spill, st.spillStride, st.rows, st.dbg, GenAnim(GEN) ? g_time[threadIdx.x] : 0.f);
                w.prim = rh.prim; w.inst = rh.inst; w.route = rh.route;
                w.tMax = rh.t; w.b0 = rh.b0; w.b1 = rh.b1; w.b2 = rh.b2;
            }
        }
        finish(idx, valid, w);
    }
}

// ---- the same walk with wave-level replacement of finished rays (round 3) ---------------------------------------------------------
// BatchTrace keeps a wave on its 64 rays until the LAST of them is done: on the 10 M-triangle scene the SQ counters showed 24 % of
// the lanes active over a wave's lifetime — walk lengths within a wave differ by an order of magnitude.  Here a wave whose active
// lanes drop to WF_REFILL_AT or fewer retires its finished lanes (`finish` is a per-lane store in every caller that uses this
// variant: no workgroup barrier) and deals them the next rays of the wave's private run of the queue (runs of chunk x 64 rays, taken
// from the shared cursor with one returning atomic each, or dealt statically when the queue is short).  Between refills the loop is
// the same "while-while" as before, so the lanes of one refill — neighbours in the queue — descend the top of the tree together
// (LDS-cached nodes), which is what the continuous per-lane refill measured in round 2 lost (5x the L1 line accesses).
#ifndef WF_REFILL_AT
#define WF_REFILL_AT 40
#endif
#ifndef WF_GUIDED_CHUNK
#define WF_GUIDED_CHUNK 6   // 0: fixed runs of `chunk` x 64 rays; > 0: the any-hit walks take guided runs of at most this many x 64
#endif
#ifndef WF_GUIDED_DIV
#define WF_GUIDED_DIV 4
#endif
template <bool ANY, int GEN, bool INST = false, bool DEFER = false, typename Fetch, typename Finish>
__device__ inline void BatchTraceRefill(const SceneView &sv, const FastBVH &bvh, int n, LdsStackT &st, Fetch fetch, Finish finish, int *cursor = nullptr, int chunk = 4,
                                        int workBlocks = 0) {
    LoadTreeTop(bvh);
    if (workBlocks <= 0) workBlocks = (int)gridDim.x;   // (the first workBlocks workgroups of the grid walk rays; the rest, if any, have another job)
    if (n < 2 * workBlocks * TBLOCK) cursor = nullptr;
    if (cursor && st.dbg && threadIdx.x == 0 && blockIdx.x == 0) atomicOr(st.dbg + 3, 1);
    const int lane = threadIdx.x & 63;
    const int waveId = (blockIdx.x * TBLOCK + threadIdx.x) >> 6, nWaves = (workBlocks * TBLOCK) >> 6;
    const int runRays = cursor ? 64 * chunk : 64;
    int next = 0, end = 0, staticJ = 0;   // the wave's private run [next, end) of ray indices (uniform)
    int lastB = 0;                        // the cursor's value at the wave's last fetch (WF_GUIDED_CHUNK)
    bool exhausted = false;
    int idx = -1;
#if WF_WALK_ZERO
    RayWalk w{};
#else
    RayWalk w;
#endif
    w.node = NODE_NONE;
    w.prim = -1;
    w.route = 0;
    w.tMax = 0;
    w.b0 = w.b1 = w.b2 = 0;
    w.inst = w.curInst = -1;
    WalkStats ws_;
    auto retire = [&]() {
        // DEFER: `finish` queues a near-tie ray and the kernel resolves it after its walks (DrainRetrace) — the reference-order walk inlined HERE
        // costs the closest-hit kernel 37 % (76.9 vs 56.0 ms per 16 spp on the spec scene), without it the refill gains 28 % (40.3 ms)
        if constexpr (!ANY && RetraceInline(GenBase(GEN)) && !DEFER) {
            if (WalkAmbiguous(w)) {
                V3 o, d;
                float t0;
                fetch(idx, &o, &d, &t0);
                const float tB = __builtin_fminf(2 * WalkBound(bvh, WalkT(w)) - WalkT(w), t0);
                const RefHit rh = RetraceRefOrder<GenBase(GEN), GenAnim(GEN)>(bvh.sv, o.x, o.y, o.z, d.x, d.y, d.z, tB, st.spill, st.spillStride, st.rows, st.dbg, GenAnim(GEN) ? g_time[threadIdx.x] : 0.f);
                w.prim = rh.prim; w.inst = rh.inst; w.route = rh.route;
                w.tMax = rh.t; w.b0 = rh.b0; w.b1 = rh.b1; w.b2 = rh.b2;
            }
        }
        finish(idx, true, w);
        idx = -1;
    };
    while (true) {
        const unsigned long long act = __ballot(w.node != NODE_NONE);
        const int nAct = __popcll(act);
        if (nAct <= WF_REFILL_AT && !exhausted) {
            const bool idle = w.node == NODE_NONE;
            if (idle && idx >= 0) retire();
            const int need = 64 - nAct;
            const int rank = __popcll(~act & ((1ull << lane) - 1ull));
            int served = 0;
            while (served < need) {
                if (next >= end) {
                    int b;
                    int rr = runRays;
                    if (cursor) {
#if WF_GUIDED_CHUNK
                        // guided self-scheduling (any-hit walks): long runs (neighbouring rays, refill after refill) while the queue is long,
                        // short ones towards its end, where a wave that still owns a long run would be the launch's tail.  Spec scene, 16 spp,
                        // same box, three runs (profiles/r06_guided_chunk_ab_sm16.txt): any-hit 12.2 -> 11.95 ms; the closest-hit walk
                        // LOSES with it (26.9 -> 28.4) and keeps its fixed runs of 3 x 64
                        // (triangle kernels only: with it k_shadow_fast<2, false> spilled an SGPR-spill carrier — tools/check_spill_carriers.py)
                        if constexpr (ANY && GenBase(GEN) <= 1) { int c = (n - lastB) / (nWaves * 64 * WF_GUIDED_DIV); c = c < 1 ? 1 : (c > WF_GUIDED_CHUNK ? WF_GUIDED_CHUNK : c); rr = 64 * c; }
#endif
                        b = 0;
                        if (lane == 0) b = atomicAdd(cursor, rr);
                        b = __builtin_amdgcn_readfirstlane(b);
                        lastB = b < n ? b : n;
                    } else {
                        b = (staticJ * nWaves + waveId) * 64;
                        ++staticJ;
                    }
                    next = b;
                    end = b + rr < n ? b + rr : n;
                    if (next >= n) { exhausted = true; break; }
                }
                const int take = need - served < end - next ? need - served : end - next;
                if (idle && rank >= served && rank < served + take) idx = next + (rank - served);
                next += take;
                served += take;
            }
            ws_.add(3, idle && idx >= 0);
            if (idle && idx >= 0) {
                V3 o, d;
                float tMax;
                fetch(idx, &o, &d, &tMax);
                float sa[3], sb[3];
                WalkInit(bvh, w, o, d, tMax, sa, sb);
                if constexpr (INST || GEN > 0) StoreWorldRay(o, d);
                if constexpr (INST && WF_SAVE_WORLD == 1) st.saveWorld(w);
                if constexpr (INST && WF_SAVE_WORLD == 2) st.saveWorldLds(sa, sb, w.sh);
                st.reset();
            }
            if (exhausted && !__any(w.node != NODE_NONE)) break;
        } else if (nAct == 0) break;
        while (WalkDescend<INST, ANY>(w)) {
            ws_.add(0, w.node >= 0);
            if (w.node >= 0) {
                U4 nd[QNODE_U4];
                FetchNode(bvh, w.node, nd);
                InteriorStep<!ANY>(bvh, w, st, nd);
            }
        }
        if constexpr (WF_WALK_STATS != 0) ws_.add(1, w.node < 0 && w.node != NODE_NONE && !(INST && AtTransition(w.node)));
        const int ranT = LeafPhase<ANY, GEN, INST>(sv, bvh, w, st, fetch, idx);
        if constexpr (WF_WALK_STATS != 0) { if (ranT) { ws_.c[4] += 1; ws_.c[5] += (unsigned long long)ranT; } }
    }
    ws_.flush(st.dbg, ANY ? 1 : 0);
    if (idx >= 0) retire();   // the rays still held when the queue ran out
}
// Measured on the spec scene (16 spp, same box; gpurun_out/r3b_ab_sm16.txt): the any-hit walk gains (30.1 -> 25.2 ms at a threshold of
// 40 lanes, 26.2 at 24, 27.5 at 12), the closest-hit walk loses (58.3 -> 65.7 / 74.9 / 72.2 ms: it carries the hit record and the
// near-tie code through the refill, 400 B of scratch instead of 304) — so only the any-hit kernels use it by default.
#ifndef WF_REFILL_SHADOW
#define WF_REFILL_SHADOW 1
#endif
#ifndef WF_REFILL_INLINE
#define WF_REFILL_INLINE 0    // diagnostic: the closest-hit walk refills its lanes but re-walks near ties inline (no queue, no service workgroups)
#endif
#ifndef WF_REFILL_GEN2
#define WF_REFILL_GEN2 1   // spec scene + one sphere, 16 spp, same box: closest-hit 73.4 -> 64.6 ms (profiles/r04_one_sphere_refill_ab_sm16.txt)
#endif
#ifndef WF_REFILL_CLOSEST
#define WF_REFILL_CLOSEST 1   // with the near-tie walk out of the loop (DrainRetrace): 56.0 -> see DESIGN 4.1
#endif
// PERLANE: `finish` has no workgroup barrier (every caller but the workgroup-routed closest-hit variant, SPLIT = false)
// DEFER (closest-hit only): the caller queues near-tie rays in `finish` and drains the queue itself after this returns
template <bool ANY, int GEN, bool INST, bool PERLANE, bool DEFER = false, typename Fetch, typename Finish>
__device__ inline void TraceQueue(const SceneView &sv, const FastBVH &bvh, int n, LdsStackT &st, Fetch fetch, Finish finish, int *cursor = nullptr, int chunk = 4,
                                  int workBlocks = 0) {
    // (the closest-hit walk of scenes with general primitives, GEN >= 2: its near-ties go to the separate re-trace launch anyway, so the
    //  refill loop carries no reference-order walk — WF_REFILL_GEN2, round 4)
    if constexpr (PERLANE && (ANY ? WF_REFILL_SHADOW != 0 : (DEFER || WF_REFILL_INLINE != 0 || (WF_REFILL_GEN2 != 0 && !RetraceInline(GenBase(GEN)))))) BatchTraceRefill<ANY, GEN, INST, DEFER>(sv, bvh, n, st, fetch, finish, cursor, chunk, workBlocks);
    else BatchTrace<ANY, GEN, INST>(sv, bvh, n, st, fetch, finish, cursor, chunk);
}
// The near-tie rays of a closest-hit launch, resolved inside the launch (round 3, second step).  A walk that ends on a near-tie publishes
// (ray index | starting bound) with ONE 64-bit atomic into retraceQ64 (slots are ~0 until written).  The reference-order walks of the
// queued rays are long single-lane walks (the old re-trace launch: 1.4-2 ms for its longest one), so they must run BESIDE the production
// walks, not after them: the last WF_SERVICE_BLOCKS workgroups of the grid walk no rays — they poll the queue (s_sleep between polls),
// take up to 64 entries per wave at a time, one walk per lane, until every worker wave has signed off and the queue is empty.  A worker
// wave that runs out of rays also takes what is queued at that moment before it signs off.  The wave that signs off last rewinds the
// counters for the next launch.  Nothing waits for a wave that is not resident: workers never wait, and the service workgroups occupy a
// fixed handful of slots.  (Grids too small to spare service workgroups: the worker that signs off last takes the rest.)
#ifndef WF_SERVICE_BLOCKS
#define WF_SERVICE_BLOCKS 16   // spec scene, 16 spp, same box: closest-hit 54.6 ms at 4, 48.0 at 8, 42.5 at 16, 44.4 at 32, 48.1 at 64 (gpurun_out/r3t_, r3u_ab_sm16.txt)
#endif
__device__ inline int ServiceBlocks() { return (int)gridDim.x >= 8 * WF_SERVICE_BLOCKS ? WF_SERVICE_BLOCKS : 0; }
template <int GEN, bool INST, bool ANIM = false>
__device__ inline void DrainRetrace(const SceneView &sv, const WorkState &ws, const FastBVH &bvh, int cur, LdsStackT &st, bool service) {
    int *cnt = ws.counters + CNT_RETRACE * CNT_STRIDE, *head = ws.counters + CNT_RETRACE_HEAD * CNT_STRIDE, *done = ws.counters + CNT_WAVES_DONE * CNT_STRIDE;
    const int lane = threadIdx.x & 63;
    const int totalWaves = (int)gridDim.x * (TBLOCK / 64), workerWaves = ((int)gridDim.x - ServiceBlocks()) * (TBLOCK / 64);
    const RayQueueV &q = ws.rq[cur];
    auto takeSome = [&]() -> bool {
        int base = 0, take = 0;
        if (lane == 0) {
            while (true) {
                const int h = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), c = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (h >= c) break;
                const int t = c - h < 64 ? c - h : 64;
                if (atomicCAS(head, h, h + t) == h) { base = h; take = t; break; }
            }
        }
        base = __builtin_amdgcn_readfirstlane(base);
        take = __builtin_amdgcn_readfirstlane(take);
        if (take == 0) return false;
        if (lane < take) {
            // a slot is published when its tag is THIS launch's epoch (no sentinel to restore, nothing a stale slot could be mistaken for);
            // the bound lies in a word of its own, written before the tag (release) and read after it (acquire)
            unsigned long long e;
            while ((uint32_t)((e = __hip_atomic_load(&ws.retraceQ64[base + lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != (uint32_t)ws.drainEpoch) {}   // reserved, store in flight
            const int i = (int)(uint32_t)(e & 0xffffffffull);
            const float tB = BitsToFloat((uint32_t)__hip_atomic_load(&ws.retraceQ[base + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const F4 o = q.o[i], d = q.d[i];
            // A queued ray HAS a hit inside its bound (the production walk accepted one at t < tB with the same triangle and alpha tests), so a
            // re-walk that returns none is a fault, not an answer.  Round 3 measured exactly that on the 10 M-triangle scene: 2-6 of 4443
            // re-walks per frame came back empty in about half of the runs (gpurun_out/det3_sm16.txt; never with the re-walk inlined in the
            // worker, never once this check was compiled in) — a pixel sample lost per occurrence.  The walk is repeated; what it costs is
            // counted (dbg + 5), and a ray that stays without a hit raises wf_sync's error (dbg + 6) instead of a silently wrong pixel.
            RefHit rh;
            int tries = 0;
            do { rh = RetraceRefOrder<GEN, ANIM>(bvh.sv, o.x, o.y, o.z, d.x, d.y, d.z, tB, st.spill, st.spillStride, st.rows, st.dbg, o.w); } while (rh.prim < 0 && ++tries < 4);
            if (st.dbg && tries) { atomicAdd(st.dbg + 5, tries); if (rh.prim < 0) atomicOr(st.dbg + 6, 1); }
            // ... and the entry itself is validated instead of trusted: the ray index lies inside this launch's queue, the
            // re-walk's hit lies inside the entry's bound, and the slot still carries this launch's tag after the walk.  A violation is
            // wf_sync's error like an unresolved ray: wrong inputs that happen to yield SOME hit must not pass silently.
            if (st.dbg) {
                const bool badIndex = i < 0 || i >= ws.counters[(CNT_RAY0 + cur) * CNT_STRIDE];
                const bool badHit = rh.prim >= 0 && !(rh.t <= tB);
                const bool badTag = (uint32_t)(__hip_atomic_load(&ws.retraceQ64[base + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != (uint32_t)ws.drainEpoch;
                if (badIndex || badHit || badTag) atomicOr(st.dbg + 6, 2);
            }
            ws.routeCode[i] = rh.route;
            ws.hit[i] = F4{BitsToFloat((uint32_t)rh.prim), rh.b0, rh.b1, rh.b2};
            if (INST) ws.hitInst[i] = rh.prim >= 0 ? rh.inst : -1;
            if (sv.haveMedia) ws.hitT[i] = rh.prim >= 0 ? rh.t : WF_INFINITY;
        }
        return true;
    };
    auto signOff = [&]() -> int {
        int prev = 0;
        if (lane == 0) { __threadfence(); prev = atomicAdd(done, 1); }
        return __builtin_amdgcn_readfirstlane(prev);
    };
    bool last;
    if (!service) {
        while (takeSome()) {}
        last = signOff() == totalWaves - 1;   // (only without service workgroups can a worker be the last)
        if (last) while (takeSome()) {}
    } else {
        while (true) {
            if (takeSome()) continue;
            int dn = 0;
            if (lane == 0) dn = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dn = __builtin_amdgcn_readfirstlane(dn);
            if (dn >= workerWaves) {      // no more pushes: one last look
                if (takeSome()) continue;
                break;
            }
            __builtin_amdgcn_s_sleep(64);
        }
        last = signOff() == totalWaves - 1;
    }
    if (last && lane == 0) {
        __hip_atomic_store(head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a second launch without the stage reset must not find stale entries)
        __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// SPLIT = false: a workgroup routes its 256 hits together at the end of every batch (KRouteHitBlock: one atomic per destination queue
// per workgroup) — its four waves wait for the slowest walk of the 256.  SPLIT = true: the walk only records the hit (ws.hit, hitInst,
// hitT, routeCode) and k_route_hits pushes the queue entries afterwards in one streaming pass; waves never meet, so a wave whose 64
// walks are done moves on to its next 64 rays while the others still walk.
constexpr uint32_t ROUTE_SKIP = 0x80000000u;   // near-tie: the re-trace routes this ray
// The occupancy target of a walk kernel.  The general-primitive variants (GEN >= 2) cannot reach the triangle kernels' 4-5 waves, but the
// target still matters: the allocator spills towards it before the callees' frames are added (asked for 3 / 2 waves, which they could
// reach, GEN = 2 / 3 ended at 2 / 1; asked for the triangle kernels' 4-5 they end at 3 / 2).  WF_TWAVES_GEN2: see the Makefile's note on
// zero-initialised locals, which cost GEN = 2 fifteen registers at the old target.
#ifndef WF_TWAVES_GEN2
#define WF_TWAVES_GEN2 0   // 0: the triangle kernels' target
#endif
// (GEN = 3, the curve kernels: at the one-level kernels' 5-wave target they were the only kernels of the library that spilled an SGPR-spill
//  carrier register — tools/check_spill_carriers.py; at the two-level kernels' 4 they do not)
// (round 5: with alpha textures on curves the GEN = 3 kernels reach the curve intersector through the alpha recursion as well — a chain of
//  out-of-line callees the compiler gives 200+ VGPRs; it reported "final occupancy is 2" for every target above, and at a target of 4 the
//  kernels spilled carriers again: their target is what they get)
constexpr int TWavesFor(int gen, int triangleWaves) { return gen == 2 && WF_TWAVES_GEN2 > 0 ? WF_TWAVES_GEN2 : gen == 3 ? 2 : triangleWaves; }
// GENX: 0 - 3 as above; 4 / 5 = GEN 0 / 1 handing rays that meet a quadric / patch / curve to `deferQ` (TWO-CLASS TRAVERSAL).
// list != nullptr: walk the rays list[0 .. counters[CNT_DEFER]) of the queue instead of all of it (the second launch of that scheme).
template <int GENX, bool INST = false, bool SPLIT = false>
__global__ void __launch_bounds__(TBLOCK, TWavesFor(GenBase(GENX), INST ? WF_TWAVES_INST : WF_TWAVES_CLOSEST)) k_closest_fast(const SceneView svArg, WorkState ws, FastBVH bvh, int cur, SpillArea sp, int *cursor = nullptr, int chunk = 4, const int *list = nullptr) {
    constexpr int GEN = GenBase(GENX);
    const SceneView &sv = SvOf<false>(svArg);
    if constexpr (!SPLIT) list = nullptr;   // (only the launches of the routing-split path take a list)
    const int n = list ? ws.counters[(CNT_DEFER) * CNT_STRIDE] : ws.counters[(CNT_RAY0 + cur) * CNT_STRIDE];
    const int gtid = blockIdx.x * TBLOCK + threadIdx.x, stride = gridDim.x * TBLOCK;
    LdsStackT st{sp.base + gtid, stride, 0, 0, sp.rows, sp.dbg, sp.save + gtid};
    const RayQueueV q = ws.rq[cur];
    // near-ties resolved by this launch itself: the kernels that would otherwise inline the reference-order walk (RetraceInline), walking with refill
    constexpr bool DRAIN = SPLIT && RetraceInline(GEN) && WF_REFILL_CLOSEST != 0 && WF_REFILL_INLINE == 0;
    const int workBlocks = DRAIN ? (int)gridDim.x - ServiceBlocks() : (int)gridDim.x;
    if (DRAIN && (int)blockIdx.x >= workBlocks) { DrainRetrace<GEN, INST, GenAnim(GENX)>(sv, ws, bvh, cur, st, true); return; }   // a service workgroup
    TraceQueue<false, GENX, INST, SPLIT, DRAIN>(
        sv, bvh, n, st,
        [&](int i0, V3 *o, V3 *d, float *tMax) {
            const int i = list ? list[i0] : i0;
            F4 o4 = q.o[i], d4 = q.d[i];
            *o = V3{o4.x, o4.y, o4.z}; *d = V3{d4.x, d4.y, d4.z}; *tMax = WF_INFINITY;
            if constexpr (GenAnim(GENX)) {
                g_time[threadIdx.x] = o4.w;
                ws.pathTime[q.meta[i].x] = o4.w;   // the path's time, for the shadow rays this depth spawns (k_intersect_closest<.., ANIM>)
            }
        },
        [&](int i0, bool valid, const RayWalk &w) {
            const int i = (list && valid) ? list[i0] : i0;
            if constexpr (GenDefer(GENX) && SPLIT)
                if (valid && (w.route & WALK_DEFER)) {   // the general-primitive launch walks this ray (and writes its record)
                    ws.deferQ[QueueAlloc(&ws.counters[(CNT_DEFER) * CNT_STRIDE])] = i;
                    ws.routeCode[i] = ROUTE_SKIP;
                    return;
                }
            // near-tie seen (wf_traverse.h): the reference-order walk decides (k_closest_retrace)
            const bool amb = valid && WalkAmbiguous(w);
            if (amb) {
                const int slot = atomicAdd(&ws.counters[(CNT_RETRACE) * CNT_STRIDE], 1);
                const float bound = 2 * WalkBound(bvh, WalkT(w)) - WalkT(w);   // the re-trace's starting bound, see k_closest_retrace
                if constexpr (DRAIN) {
                    if (sp.dbg) atomicAdd(sp.dbg + 4, 1);
                    __hip_atomic_store(&ws.retraceQ[slot], (int)FloatToBits(bound), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ws.retraceQ64[slot], (unsigned long long)(uint32_t)i | ((unsigned long long)(uint32_t)ws.drainEpoch << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    return;   // DrainRetrace writes the ray's record
                } else {
                    ws.retraceQ[slot] = i;
                    ws.hit[i] = F4{0, bound, 0, 0};
                }
            }
            if constexpr (SPLIT) {
                if (!valid) return;
                ws.routeCode[i] = amb ? ROUTE_SKIP : (uint32_t)w.route;
                if (amb) return;
                ws.hit[i] = F4{BitsToFloat((uint32_t)w.prim), w.b0, w.b1, w.b2};
                if (INST) ws.hitInst[i] = w.prim >= 0 ? w.inst : -1;
                if (sv.haveMedia) ws.hitT[i] = w.prim >= 0 ? WalkT(w) : WF_INFINITY;
            } else
            KRouteHitBlock<(GEN > 1) || INST>(sv, ws, cur, i, valid && !amb, w.prim, w.route, WalkT(w), w.b0, w.b1, w.b2, INST ? w.inst : -1);
        }, SPLIT ? cursor : nullptr, chunk, workBlocks);
    if constexpr (DRAIN) DrainRetrace<GEN, INST, GenAnim(GENX)>(sv, ws, bvh, cur, st, false);
}
// the routing pass of the SPLIT traversal: EnqueueWorkAfterIntersection / Miss for every ray of the queue (block-aggregated pushes)
// (1024 threads per workgroup: one returning atomic per destination queue per 1024 rays — a queue counter sustains ~88 of them per
// microsecond, which at 256 rays per workgroup was the whole cost of this pass)
constexpr int RBLOCK = 1024;
template <bool GENERAL>
__global__ void __launch_bounds__(RBLOCK) k_route_hits(const SceneView svArg, WorkState ws, int cur) {
    const SceneView &sv = SvOf<true>(svArg);
    const int n = ws.counters[(CNT_RAY0 + cur) * CNT_STRIDE];
    for (int base = blockIdx.x * RBLOCK; base < n; base += gridDim.x * RBLOCK) {
        const int i = base + threadIdx.x;
        bool valid = i < n;
        uint32_t route = 0;
        F4 h{0, 0, 0, 0};
        int inst = -1;
        float tHit = 0;
        if (valid) {
            route = ws.routeCode[i];
            if (route & ROUTE_SKIP) valid = false;
            else {
                h = ws.hit[i];
                if (sv.nInstances > 0) inst = ws.hitInst[i];
                if (sv.haveMedia) tHit = ws.hitT[i];
            }
        }
        KRouteHitBlock<GENERAL>(sv, ws, cur, i, valid, (int)FloatToBits(h.x), route, tHit, h.y, h.z, h.w, inst);
    }
}
// ---- ray-coherence pass -----------------------------------------------------------------------------------------
// Rays past the first bounce arrive in the order the material kernels pushed them: neighbouring lanes start anywhere in the scene and
// point anywhere, a wavefront's 64 walks share neither nodes nor length.  Before such a queue is traced it is SORTED by a key made of
// the cell of the ray's origin (Morton order of a 2^b grid over the scene bounds) and its direction (Morton order of the octahedral
// map), and the queue is permuted physically — one gather per ray, after which the traversal and every later stage of the depth
// read coalesced again.  Results do not depend on the order of a queue (each item owns its pixel-sample slot), only the time does.
extern "C" int wf_sort_pairs_u32(hipStream_t stream, void *temp, size_t *tempBytes, const uint32_t *keysIn, uint32_t *keysOut, const uint32_t *valsIn, uint32_t *valsOut,
                                 unsigned n, unsigned endBit);
__device__ inline uint32_t Part1By2(uint32_t x) {  // 10 bits -> every third bit
    x &= 0x3ff;
    x = (x ^ (x << 16)) & 0xff0000ff;
    x = (x ^ (x << 8)) & 0x0300f00f;
    x = (x ^ (x << 4)) & 0x030c30c3;
    x = (x ^ (x << 2)) & 0x09249249;
    return x;
}
__device__ inline uint32_t Part1By1(uint32_t x) {  // 16 bits -> every second bit
    x &= 0xffff;
    x = (x ^ (x << 8)) & 0x00ff00ff;
    x = (x ^ (x << 4)) & 0x0f0f0f0f;
    x = (x ^ (x << 2)) & 0x33333333;
    x = (x ^ (x << 1)) & 0x55555555;
    return x;
}
struct SortGrid { float base[3], scale[3]; int obits, dbits; };
__device__ inline uint32_t RaySortKey(const SortGrid &g, F4 o, F4 d) {
    const int oMax = (1 << g.obits) - 1;
    int cx = min(max((int)((o.x - g.base[0]) * g.scale[0]), 0), oMax);
    int cy = min(max((int)((o.y - g.base[1]) * g.scale[1]), 0), oMax);
    int cz = min(max((int)((o.z - g.base[2]) * g.scale[2]), 0), oMax);
    uint32_t key = Part1By2(cx) | (Part1By2(cy) << 1) | (Part1By2(cz) << 2);
    if (g.dbits > 0) {
        // octahedral map of the direction to [0, 1)^2
        float inv = 1.f / (fabsf(d.x) + fabsf(d.y) + fabsf(d.z));
        float u = d.x * inv, v = d.y * inv;
        if (d.z < 0) {
            float tu = (1 - fabsf(v)) * (u >= 0 ? 1.f : -1.f), tv = (1 - fabsf(u)) * (v >= 0 ? 1.f : -1.f);
            u = tu; v = tv;
        }
        const int dMax = (1 << g.dbits) - 1;
        int du = min(max((int)((u * .5f + .5f) * (dMax + 1)), 0), dMax), dv = min(max((int)((v * .5f + .5f) * (dMax + 1)), 0), dMax);
        key = (key << (2 * g.dbits)) | Part1By1(du) | (Part1By1(dv) << 1);
    }
    return key;
}
__global__ void __launch_bounds__(BLOCK) k_ray_sort_keys(const F4 *o, const F4 *d, int n, SortGrid g, uint32_t *keys, uint32_t *vals) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        keys[i] = RaySortKey(g, o[i], d[i]);
        vals[i] = (uint32_t)i;
    }
}
__global__ void __launch_bounds__(BLOCK) k_permute_rays(RayQueueV src, RayQueueV dst, const uint32_t *perm, int n) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t j = perm[i];
        F4 a = src.o[j], b = src.d[j], c = src.beta[j], e = src.r_u[j], f = src.r_l[j], g = src.ctx0[j], h = src.ctx1[j], k = src.ctx2[j];
        I4 m = src.meta[j];
        dst.o[i] = a; dst.d[i] = b; dst.beta[i] = c; dst.r_u[i] = e; dst.r_l[i] = f; dst.ctx0[i] = g; dst.ctx1[i] = h; dst.ctx2[i] = k; dst.meta[i] = m;
    }
}
__global__ void __launch_bounds__(BLOCK) k_permute_shadow(ShadowQueueV src, ShadowQueueV dst, const uint32_t *perm, int n) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const uint32_t j = perm[i];
        F4 a = src.o[j], b = src.d[j], c = src.Ld[j], e = src.r_u[j], f = src.r_l[j];
        dst.o[i] = a; dst.d[i] = b; dst.Ld[i] = c; dst.r_u[i] = e; dst.r_l[i] = f;
        if (src.medium) dst.medium[i] = src.medium[j];
    }
}

// the rays k_closest_fast marked as near-ties, in the reference's own traversal order (rare: coplanar overlapping geometry)
__global__ void __launch_bounds__(BLOCK) k_closest_retrace(const SceneView sv, WorkState ws, int cur, int *stackSpill) {
    const int n = ws.counters[(CNT_RETRACE) * CNT_STRIDE];
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    for (int qi = gtid; qi < n; qi += stride) {
        const int i = ws.retraceQ[qi];
        F4 o = ws.rq[cur].o[i], d = ws.rq[cur].d[i];
        ClosestHit ch;
        st.n = 0;
        // The walk starts from tMax = t* + twice the near-tie band instead of infinity (t* = the production walk's hit): every
        // candidate that can end up as the reference's hit lies inside the band around t*, a candidate beyond the starting
        // bound could only have been a temporary hit that the band's members replace — the result is the reference's, and
        // the walk visits a fraction of the nodes (the latency of this launch is one ray's dependent chain).
        bool found = BVHIntersectClosest(sv, V3{o.x, o.y, o.z}, V3{d.x, d.y, d.z}, ws.hit[i].y, st, &ch);
        KAfterClosestHit(sv, ws, cur, i, found, ch.prim, ch.inst, ch.h.t, ch.h.b0, ch.h.b1, ch.h.b2);
    }
}
// K12 (wavefront/subsurface.cpp): probe segment, one random intersection of it with the same material, out-scattering
__global__ void __launch_bounds__(BLOCK) k_subsurface_probe(const SceneView sv, WorkState ws) {
    const int n = ws.counters[(CNT_BSSRDF) * CNT_STRIDE];
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) KSubsurfaceProbe(sv, ws, i);
}
template <bool ANIM>
__global__ void __launch_bounds__(BLOCK) k_intersect_one_random(const SceneView sv, WorkState ws, int *stackSpill) {
    const int n = ws.counters[(CNT_SSS) * CNT_STRIDE];
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    for (int i = gtid; i < n; i += stride) KIntersectOneRandom<ANIM>(sv, ws, i, st);
}
// IntersectOneRandom on caller-supplied probe segments (the boundary adapter's path): segs = p0.xyz p1.xyz per item
__global__ void __launch_bounds__(BLOCK) k_trace_one_random(const SceneView sv, int n, const float *segs, const int32_t *material, wf_hit_record *out, float *pdf,
                                                            int *stackSpill) {
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    for (int i = gtid; i < n; i += stride) {
        const float *s = segs + (size_t)6 * i;
        ClosestHit ch;
        SurfIntr si;
        float p = IntersectOneRandom(sv, V3{s[0], s[1], s[2]}, V3{s[3], s[4], s[5]}, material[i], st, &ch, &si);
        wf_hit_record h{};
        h.prim = p != 0 ? ch.prim : -1;
        if (p != 0) { h.t = ch.h.t; h.b0 = ch.h.b0; h.b1 = ch.h.b1; h.b2 = ch.h.b2; h.instance = ch.inst; } else h.instance = -1;
        out[i] = h;
        pdf[i] = p;
    }
}
__global__ void __launch_bounds__(BLOCK) k_subsurface_scatter(const SceneView sv, WorkState ws, int cur) {
    const int n = ws.counters[(CNT_SSS) * CNT_STRIDE];
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) KSubsurfaceScatter(sv, ws, cur, i);
}
template <int GEN, bool INST = false>
__global__ void __launch_bounds__(TBLOCK, TWavesFor(GenBase(GEN), INST ? WF_TWAVES_INST_SHADOW : WF_TWAVES)) k_shadow_fast(const SceneView svArg, WorkState ws, FastBVH bvh, SpillArea sp, int *cursor = nullptr, int chunk = 4, const int *list = nullptr) {
    const SceneView &sv = SvOf<false>(svArg);
    const int n = list ? ws.counters[(CNT_DEFER_SHADOW) * CNT_STRIDE] : ws.counters[(CNT_SHADOW) * CNT_STRIDE];
    const int gtid = blockIdx.x * TBLOCK + threadIdx.x, stride = gridDim.x * TBLOCK;
    LdsStackT st{sp.base + gtid, stride, 0, 0, sp.rows, sp.dbg, sp.save + gtid};
    TraceQueue<true, GEN, INST, true>(
        sv, bvh, n, st,
        [&](int i0, V3 *o, V3 *d, float *tMax) {
            const int i = list ? list[i0] : i0;
            F4 o4 = ws.sq.o[i], d4 = ws.sq.d[i];
            *o = V3{o4.x, o4.y, o4.z}; *d = V3{d4.x, d4.y, d4.z}; *tMax = o4.w;
            if constexpr (GenAnim(GEN)) g_time[threadIdx.x] = ShadowTime<true>(ws, d4.w);
        },
        [&](int i0, bool valid, const RayWalk &w) {
            if (!valid) return;
            const int i = list ? list[i0] : i0;
            if constexpr (GenDefer(GEN))
                if (w.route & WALK_DEFER) {   // (the walk stops at its first occluder: a ray handed over has none among the triangles before the leaf it met)
                    ws.deferQ[QueueAlloc(&ws.counters[(CNT_DEFER_SHADOW) * CNT_STRIDE])] = i;
                    return;
                }
            KRecordShadowRay(ws, i, w.prim >= 0);
        }, cursor, chunk);
}

// GENERAL: the scene has alpha-tested triangles or quadrics (the variant the render uses then)
template <int GEN, bool INST>
__global__ void __launch_bounds__(TBLOCK) k_trace_closest_fast(const SceneView sv, FastBVH bvh, int n, const float *rays, wf_hit_record *out, SpillArea sp) {
    const int gtid = blockIdx.x * TBLOCK + threadIdx.x, stride = gridDim.x * TBLOCK;
    LdsStackT st{sp.base + gtid, stride, 0, 0, sp.rows, sp.dbg, sp.save + gtid};
    TraceQueue<false, GEN, INST, true>(
        sv, bvh, n, st,
        [&](int i, V3 *o, V3 *d, float *tMax) {
            const float *r = rays + (size_t)7 * i;
            *o = V3{r[0], r[1], r[2]}; *d = V3{r[3], r[4], r[5]}; *tMax = r[6];
        },
        [&](int i, bool valid, const RayWalk &w) {
            if (!valid) return;
            wf_hit_record h;
            bool found = w.prim >= 0;
            h.prim = w.prim;
            h.t = found ? WalkT(w) : 0; h.b0 = w.b0; h.b1 = w.b1; h.b2 = w.b2;
            h.nodes_visited = WalkAmbiguous(w) ? -1 : 0;  // -1: near-tie, re-traced in reference order by the pass that follows
            h.tris_tested = 0; h.instance = (INST && found) ? w.inst : -1;
            out[i] = h;
        });
}
template <int GEN, bool INST>
__global__ void __launch_bounds__(TBLOCK) k_trace_any_fast(const SceneView sv, FastBVH bvh, int n, const float *rays, int32_t *occluded, SpillArea sp) {
    const int gtid = blockIdx.x * TBLOCK + threadIdx.x, stride = gridDim.x * TBLOCK;
    LdsStackT st{sp.base + gtid, stride, 0, 0, sp.rows, sp.dbg, sp.save + gtid};
    TraceQueue<true, GEN, INST, true>(
        sv, bvh, n, st,
        [&](int i, V3 *o, V3 *d, float *tMax) {
            const float *r = rays + (size_t)7 * i;
            *o = V3{r[0], r[1], r[2]}; *d = V3{r[3], r[4], r[5]}; *tMax = r[6];
        },
        [&](int i, bool valid, const RayWalk &w) { if (valid) occluded[i] = w.prim >= 0; });
}

__global__ void __launch_bounds__(BLOCK) k_resolve_mix(const SceneView sv, WorkState ws, int cur) {
    const int n = ws.counters[(CNT_MIX) * CNT_STRIDE];
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) KResolveMix(sv, ws, cur, i);
}
// K5 / K6 / K11: participating media (wf_media.h, wf_kernels.h)
// (round 5) the delta-tracking loop alone: the routing of the items that reach their surface — interaction rebuild of interface hits,
// MixMaterial resolve — is the kernel after it (219 VGPRs / 604 B of scratch -> 207 / 0 for the loop; WF_MEDIUM_WAVES sets its occupancy target)
#ifndef WF_MEDIUM_WAVES
#define WF_MEDIUM_WAVES 2
#endif
// (round 6) WAVE-LEVEL REFILL of the delta-tracking loop.  Until round 5 a wave kept its 64 items until the longest
// of their walks ended: on the cloud scene 24 % of the VALU lanes were active over a wave's lifetime (walks of one to a hundred and more
// steps: majorant cells crossed, null collisions taken), at two waves per SIMD that wait for their density gathers 79 % of the time.
// Now the kernel is persistent (one resident grid, work dealt in runs of 64 items from a shared cursor): a wave whose active lanes drop
// to WF_MEDIUM_REFILL_AT or fewer takes new items for its idle lanes (MediumTrackBegin) and goes on stepping; a lane whose walk ends
// retires its item at once (MediumTrackEnd: per-lane stores, wave-aggregated queue pushes — nothing waits for the workgroup).  The same
// operations per item in the same order as the one-loop form (the CPU checker's KSampleMediumInteraction): bit-identical images.
#ifndef WF_MEDIUM_REFILL
#define WF_MEDIUM_REFILL 0
#endif
#ifndef WF_MEDIUM_REFILL_AT
#define WF_MEDIUM_REFILL_AT 40
#endif
#ifndef WF_MEDIUM_STEPS
#define WF_MEDIUM_STEPS 4   // steps between two looks at the wave's occupancy
#endif
// LEAN: every medium of the scene is homogeneous or a non-emissive uniform grid (ctx->mediumLean, set at upload): the kernel is built without the
// procedural cloud, NanoVDB, RGB-grid and blackbody code and compiled for WF_MEDIUM_WAVES_LEAN waves
#ifndef WF_MEDIUM_WAVES_LEAN
#define WF_MEDIUM_WAVES_LEAN 3
#endif
template <bool LEAN>
__global__ void __launch_bounds__(BLOCK, LEAN ? WF_MEDIUM_WAVES_LEAN : WF_MEDIUM_WAVES) k_medium_sample(const SceneView sv, WorkState ws, int cur, int *cursor) {
    const int n = ws.counters[(CNT_MEDIUM_SAMPLE) * CNT_STRIDE];
#if WF_MEDIUM_REFILL
    const int lane = threadIdx.x & 63;
    const int waveId = (blockIdx.x * BLOCK + threadIdx.x) >> 6, nWaves = ((int)gridDim.x * BLOCK) >> 6;
    if (n < 2 * nWaves * 64) cursor = nullptr;   // (a short queue is dealt statically: every wave starts at once, no atomics)
    MediumTrack s;
    s.i = -1;
    int next = 0, end = 0, staticJ = 0;   // the wave's private run [next, end) of queue indices (uniform)
    bool exhausted = false;
    while (true) {
        const unsigned long long act = __ballot(s.i >= 0);
        const int nAct = __popcll(act);
        if (nAct <= WF_MEDIUM_REFILL_AT && !exhausted) {
            const bool idle = s.i < 0;
            const int need = 64 - nAct;
            const int rank = __popcll(~act & ((1ull << lane) - 1ull));
            int served = 0, qi = -1;
            while (served < need) {
                if (next >= end) {
                    int b;
                    if (cursor) {
                        b = 0;
                        if (lane == 0) b = atomicAdd(cursor, 64);
                        b = __builtin_amdgcn_readfirstlane(b);
                    } else {
                        b = (staticJ * nWaves + waveId) * 64;
                        ++staticJ;
                    }
                    next = b;
                    end = b + 64 < n ? b + 64 : n;
                    if (next >= n) { exhausted = true; break; }
                }
                const int take = need - served < end - next ? need - served : end - next;
                if (idle && rank >= served && rank < served + take) qi = next + (rank - served);
                next += take;
                served += take;
            }
            if (qi >= 0) MediumTrackBegin<LEAN>(sv, ws, cur, qi, s);
            if (!__any(s.i >= 0)) break;   // (nothing was dealt and nothing is left)
        } else if (nAct == 0) break;
        for (int k = 0; k < WF_MEDIUM_STEPS; ++k) {
            if (s.i >= 0 && !MediumTrackStep<LEAN>(sv, ws, cur, s)) {
                MediumTrackEnd(sv, ws, cur, s);
                s.i = -1;
            }
        }
    }
#else
    // (the nested-loop form of rounds 1-5 measured 43.2 ms against this form's 36.5 on the cloud scene, same box: profiles/r06_medium_nested_vs_state_machine_ab_cloud16.txt)
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) KSampleMediumInteraction<LEAN>(sv, ws, cur, i);
#endif
}
__global__ void __launch_bounds__(BLOCK) k_medium_route(const SceneView sv, WorkState ws, int cur) {
    const int n = ws.counters[(CNT_MEDIUM_ROUTE) * CNT_STRIDE];
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) KMediumRoute(sv, ws, cur, i);
}
template <bool RARE>
__global__ void __launch_bounds__(BLOCK) k_medium_scatter(const SceneView sv, WorkState ws, int cur) {
    const int n = ws.counters[(CNT_MEDIUM_SCATTER) * CNT_STRIDE];
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) KSampleMediumScattering<RARE>(sv, ws, cur, i);
}
// reference-order variant (no production BVH, or WF_NO_FAST)
template <bool ANIM>
__global__ void __launch_bounds__(BLOCK) k_shadow_tr(const SceneView sv, WorkState ws, int *stackSpill) {
    const int n = ws.counters[(CNT_SHADOW) * CNT_STRIDE];
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    for (int i = gtid; i < n; i += stride) {
        const float time = ShadowTime<ANIM>(ws, ws.sq.d[i].w);
        KTraceTransmittance<ANIM>(sv, ws, i, [&](V3 o, V3 d, float tMax, int *prim, int *inst, float *b0, float *b1, float *b2) {
            ClosestHit ch;
            st.n = 0;
            bool found = BVHIntersectClosest<ANIM>(sv, o, d, tMax, st, &ch, time);
            if (found) { *prim = ch.prim; *inst = ch.inst; *b0 = ch.h.b0; *b1 = ch.h.b1; *b2 = ch.h.b2; }
            return found;
        });
    }
}
// production layout, one independent walk per lane (a transmittance ray alternates between tracing and
// ratio tracking, so there is no batch to share)
struct TrPrims {  // the same callbacks for the transmittance walk, whose ray is a local of its loop
    static constexpr bool pairBands = true;   // (a per-lane loop over whole scenes, general shapes included)
    static constexpr bool deferGeneral = false;
    const SceneView &sv;
    V3 o, d;
    __device__ bool accept(int prim, float b0, float b1, float b2) const { return AlphaTestPasses(sv, prim, b0, b1, b2, o, d); }
    __device__ bool sphere(int prim, float tMax, QuadricHit *qh) const { return QuadricIntersect(sv, prim, o, d, tMax, qh); }
};
// MLEAN: the lean medium code (ctx->mediumLean, as k_medium_sample<true>)
template <bool ALPHA, bool MLEAN = false>
__global__ void __launch_bounds__(TBLOCK) k_shadow_tr_fast(const SceneView sv, WorkState ws, FastBVH bvh, SpillArea sp) {
    const int n = ws.counters[(CNT_SHADOW) * CNT_STRIDE];
    const int gtid = blockIdx.x * TBLOCK + threadIdx.x, stride = gridDim.x * TBLOCK;
    LdsStackT st{sp.base + gtid, stride, 0, 0, sp.rows, sp.dbg, sp.save + gtid};
    LoadTreeTop(bvh);
    for (int i = gtid; i < n; i += stride)
        KTraceTransmittance<false, MLEAN>(sv, ws, i, [&](V3 o, V3 d, float tMax, int *prim, int *inst, float *b0, float *b1, float *b2) {
            RayWalk w;
            WalkInit(bvh, w, o, d, tMax);
            st.reset();
            while (w.node != NODE_NONE) {
                if (w.node >= 0) {
                    U4 nd[QNODE_U4];
                    FetchNode(bvh, w.node, nd);
                    InteriorStep(bvh, w, st, nd);
                } else if constexpr (ALPHA) LeafStep<false, true>(bvh, w, st, TrPrims{sv, o, d});
                else LeafStep<false>(bvh, w, st);
            }
            if (WalkAmbiguous(w)) {  // near-tie (wf_traverse.h): the reference-order walk decides
                ClosestHit ch;
                st.reset();
                bool found = BVHIntersectClosest(sv, o, d, tMax, st, &ch);
                if (found) { *prim = ch.prim; *inst = ch.inst; *b0 = ch.h.b0; *b1 = ch.h.b1; *b2 = ch.h.b2; }
                return found;
            }
            if (w.prim >= 0) { *prim = w.prim; *b0 = w.b0; *b1 = w.b1; *b2 = w.b2; }
            return w.prim >= 0;
        });
}

// ---- the transmittance wavefront (round 3): TraceTransmittance as launches instead of one loop per lane ------------------------------
// k_tr_begin sets up the state of every shadow ray (TrBegin) and the first index queue; per segment k_tr_trace walks the live rays
// with the production closest-hit traversal (two-level scenes included; near ties resolved inside the walk: the scenes of the
// GEN <= 1 variants) and records the hits in ws.hit / ws.hitInst — free at this point of the depth — and k_tr_segment runs the rest
// of the loop body (interaction, ratio tracking along the segment, respawn behind an interface), finishing rays or queuing them for
// the next segment.  After WF_TR_SEGMENTS rounds k_tr_rest finishes what is still alive with the per-lane loop (reference-order walk).
#ifndef WF_TR_SEGMENTS
#define WF_TR_SEGMENTS 4
#endif
__global__ void __launch_bounds__(BLOCK) k_tr_begin(const SceneView sv, WorkState ws) {
    const int n = ws.counters[(CNT_SHADOW) * CNT_STRIDE];
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        TrState st;
        TrBegin(ws, i, &st);
        if (st.rd.x == 0 && st.rd.y == 0 && st.rd.z == 0) { TrFinish(ws, i, st); continue; }
        TrStore(ws, i, st);
        ws.trQ[0][QueueAlloc(&ws.counters[(CNT_TR0) * CNT_STRIDE])] = i;
    }
}
template <int GEN, bool INST>
__global__ void __launch_bounds__(TBLOCK, INST ? WF_TWAVES_INST : WF_TWAVES_CLOSEST) k_tr_trace(const SceneView sv, WorkState ws, FastBVH bvh, int cur, SpillArea sp) {
    const int n = ws.counters[(CNT_TR0 + cur) * CNT_STRIDE];
    const int gtid = blockIdx.x * TBLOCK + threadIdx.x, stride = gridDim.x * TBLOCK;
    LdsStackT st{sp.base + gtid, stride, 0, 0, sp.rows, sp.dbg, sp.save + gtid};
    const int32_t *q = ws.trQ[cur];
    TraceQueue<false, GEN, INST, true>(
        sv, bvh, n, st,
        [&](int j, V3 *o, V3 *d, float *tMax) {
            const int i = q[j];
            F4 o4 = ws.trO[i], d4 = ws.trD[i];
            *o = V3{o4.x, o4.y, o4.z}; *d = V3{d4.x, d4.y, d4.z}; *tMax = ws.sq.o[i].w;
        },
        [&](int j, bool valid, const RayWalk &w) {
            if (!valid) return;
            const int i = q[j];
            ws.hit[i] = F4{BitsToFloat((uint32_t)w.prim), w.b0, w.b1, w.b2};
            if (INST) ws.hitInst[i] = w.prim >= 0 ? w.inst : -1;
        });
}
template <bool MLEAN>
__global__ void __launch_bounds__(BLOCK) k_tr_segment(const SceneView sv, WorkState ws, int cur) {
    const int n = ws.counters[(CNT_TR0 + cur) * CNT_STRIDE];
    for (int j = blockIdx.x * BLOCK + threadIdx.x; j < n; j += gridDim.x * BLOCK) {
        const int i = ws.trQ[cur][j];
        TrState st;
        TrLoad(ws, i, &st);
        const F4 h = ws.hit[i];
        const int prim = (int)FloatToBits(h.x);
        if (TrSegment<false, MLEAN>(sv, ws, i, &st, prim >= 0, prim, HitInst(sv, ws, i), h.y, h.z, h.w)) {
            TrStore(ws, i, st);
            ws.trQ[cur ^ 1][QueueAlloc(&ws.counters[(CNT_TR0 + (cur ^ 1)) * CNT_STRIDE])] = i;
        } else TrFinish(ws, i, st);
    }
}
__global__ void __launch_bounds__(BLOCK) k_tr_rest(const SceneView sv, WorkState ws, int cur, int *stackSpill) {
    const int n = ws.counters[(CNT_TR0 + cur) * CNT_STRIDE];
    if (n == 0) return;
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    for (int j = gtid; j < n; j += stride) {
        const int i = ws.trQ[cur][j];
        TrState ts;
        TrLoad(ws, i, &ts);
        KTraceTransmittanceFrom(sv, ws, i, ts, [&](V3 o, V3 d, float tMax, int *prim, int *inst, float *b0, float *b1, float *b2) {
            ClosestHit ch;
            st.n = 0;
            bool found = BVHIntersectClosest(sv, o, d, tMax, st, &ch);
            if (found) { *prim = ch.prim; *inst = ch.inst; *b0 = ch.h.b0; *b1 = ch.h.b1; *b2 = ch.h.b2; }
            return found;
        });
    }
}

// RARE = false: no portal infinite light (k_handle_escaped: 328 VGPRs + 72 AGPRs — one wave per SIMD — with the portal callees and the area
// lights' PDFs reachable; round 5)
template <bool RARE>
__global__ void __launch_bounds__(BLOCK) k_handle_escaped(const SceneView svArg, WorkState ws, int cur) {
    const SceneView &sv = SvOf<false>(svArg);
    const int n = ws.counters[(CNT_ESCAPED) * CNT_STRIDE];
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) KHandleEscaped<RARE>(sv, ws, cur, i);
}
__global__ void __launch_bounds__(BLOCK) k_handle_emissive(const SceneView svArg, WorkState ws, int cur) {
    const SceneView &sv = SvOf<false>(svArg);
    const int n = ws.counters[(CNT_HITLIGHT) * CNT_STRIDE];
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) KHandleEmissive(sv, ws, cur, i);
}
// the material kernels live in wf_mat.hip (one translation unit per material type)
extern "C" {
#define WF_DECL_MAT(n) void wf_launch_eval_material_##n##_0(hipStream_t, int, const SceneView *, const WorkState *, int); \
                       void wf_launch_eval_material_##n##_1(hipStream_t, int, const SceneView *, const WorkState *, int); \
                       void wf_launch_eval_material_##n##_2(hipStream_t, int, const SceneView *, const WorkState *, int);
#if defined(WF_HAVE_FUSED_MAT)   // (make MATFUSED=1: the one-kernel material stage of rounds 1-4 beside the split one, for same-box A/B runs)
WF_DECL_MAT(1) WF_DECL_MAT(2) WF_DECL_MAT(3) WF_DECL_MAT(4) WF_DECL_MAT(5) WF_DECL_MAT(6) WF_DECL_MAT(7) WF_DECL_MAT(8) WF_DECL_MAT(9) WF_DECL_MAT(10)
#endif
// the two halves of the material stage (wf_mat.hip): shade variant 0 | 1 (lean) | 2 | 3, next-event estimation variant 0 | 1
#define WF_DECL_SPLIT(n) void wf_launch_mat_shade_##n##_0(hipStream_t, int, const SceneView *, const WorkState *, int); \
                         void wf_launch_mat_shade_##n##_1(hipStream_t, int, const SceneView *, const WorkState *, int); \
                         void wf_launch_mat_shade_##n##_2(hipStream_t, int, const SceneView *, const WorkState *, int); \
                         void wf_launch_mat_shade_##n##_3(hipStream_t, int, const SceneView *, const WorkState *, int); \
                         void wf_launch_mat_nee_##n##_0(hipStream_t, int, const SceneView *, const WorkState *); \
                         void wf_launch_mat_nee_##n##_1(hipStream_t, int, const SceneView *, const WorkState *);
WF_DECL_SPLIT(1) WF_DECL_SPLIT(2) WF_DECL_SPLIT(3) WF_DECL_SPLIT(4) WF_DECL_SPLIT(5) WF_DECL_SPLIT(6) WF_DECL_SPLIT(7) WF_DECL_SPLIT(8) WF_DECL_SPLIT(9) WF_DECL_SPLIT(10)
}
// bytes of one NeeItem's BxDF (wf_kernels.h) -> 16-byte planes of the record between the two kernels (wf_mat.hip: NeeIO)
template <int MAT> constexpr int NeePlanesOf() { return 10 + (int)((sizeof(typename MatBxDF<MAT>::T) + 15) / 16); }
static int NeePlanes(int m) {
    switch (m) {
    case 1: return NeePlanesOf<1>(); case 2: return NeePlanesOf<2>(); case 3: return NeePlanesOf<3>(); case 4: return NeePlanesOf<4>(); case 5: return NeePlanesOf<5>();
    case 6: return NeePlanesOf<6>(); case 7: return NeePlanesOf<7>(); case 8: return NeePlanesOf<8>(); case 9: return NeePlanesOf<9>(); case 10: return NeePlanesOf<10>();
    }
    return 0;
}
__global__ void __launch_bounds__(BLOCK) k_update_film(const SceneView sv, WorkState ws, int nSamples) {
    for (int p = blockIdx.x * BLOCK + threadIdx.x; p < ws.pixelsPerPass; p += gridDim.x * BLOCK) KUpdateFilm(sv, ws, p, nSamples);
}
// The RGB film under the pixel-major item order (WorkState::slotStride = nSamples): a pixel's samples are neighbours, so the workgroup
// reads FILM_PIX pixels x nSamples items with coalesced loads, one item per thread and round, leaves each item's four addends in LDS, and
// one thread per pixel then adds its samples in slot order into the double accumulators — the order KUpdateFilm adds them in.
constexpr int FILM_PIX = 32, FILM_MAX_SLOTS = 64;
__global__ void __launch_bounds__(BLOCK) k_update_film_pm(const SceneView sv, WorkState ws, int nSamples) {
    __shared__ float add[FILM_PIX * FILM_MAX_SLOTS][4];
    __shared__ unsigned long long filmIdx[FILM_PIX];
    const int nItems = FILM_PIX * nSamples;
    for (int p0 = blockIdx.x * FILM_PIX; p0 < ws.pixelsPerPass; p0 += gridDim.x * FILM_PIX) {
        if (threadIdx.x < FILM_PIX) filmIdx[threadIdx.x] = ~0ull;
        __syncthreads();
        for (int j = threadIdx.x; j < nItems; j += BLOCK) {
            const int pl = j / nSamples;
            if (p0 + pl >= ws.pixelsPerPass) break;
            size_t idx;
            float v[4];
            if (FilmSampleRGBW(sv, ws, p0 * nSamples + j, v, &idx)) {
                add[j][0] = v[0]; add[j][1] = v[1]; add[j][2] = v[2]; add[j][3] = v[3];
                if (j - pl * nSamples == 0) filmIdx[pl] = idx;
            }
        }
        __syncthreads();
        if (threadIdx.x < FILM_PIX && filmIdx[threadIdx.x] != ~0ull) {
            double *px = ws.film + 4 * filmIdx[threadIdx.x];
            double a0 = px[0], a1 = px[1], a2 = px[2], a3 = px[3];
            for (int s = 0; s < nSamples; ++s) {
                const float *v = add[threadIdx.x * nSamples + s];
                a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
            }
            px[0] = a0; px[1] = a1; px[2] = a2; px[3] = a3;
        }
        __syncthreads();
    }
}

// stand-alone traversal for parity tests / counters: rays as packed {o[3], d[3], tMax}
// onlyMarked: the re-trace pass after k_trace_closest_fast — only the records it marked as near-ties (nodes_visited == -1)
#ifndef WF_DBG_RETRACED
#define WF_DBG_RETRACED 0
#endif
__global__ void __launch_bounds__(BLOCK) k_trace_closest(const SceneView sv, int n, const float *rays, wf_hit_record *out, int *stackSpill, int onlyMarked) {
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    for (int i = gtid; i < n; i += stride) {
        if (onlyMarked && out[i].nodes_visited != -1) continue;
        const float *r = rays + (size_t)7 * i;
        ClosestHit ch;
        st.n = 0;
        bool found = BVHIntersectClosest(sv, V3{r[0], r[1], r[2]}, V3{r[3], r[4], r[5]}, r[6], st, &ch);
        wf_hit_record h;
        h.prim = found ? ch.prim : -1;
        h.t = found ? ch.h.t : 0; h.b0 = found ? ch.h.b0 : 0; h.b1 = found ? ch.h.b1 : 0; h.b2 = found ? ch.h.b2 : 0;
        h.nodes_visited = onlyMarked ? WF_DBG_RETRACED : ch.nodesVisited; h.tris_tested = onlyMarked ? 0 : ch.trisTested; h.instance = found ? ch.inst : -1;
        out[i] = h;
    }
}
// ... and at the rays' own TIMES (AnimatedPrimitive, cpu/primitive.cpp:132-158: the reference's WavefrontAggregate reads ray.time): the
// reference-order walks with the per-ray interpolation of the animated transformations (wf_trace_*_host_t)
__global__ void __launch_bounds__(BLOCK) k_trace_closest_timed(const SceneView sv, int n, const float *rays, const float *times, wf_hit_record *out, int *stackSpill) {
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    for (int i = gtid; i < n; i += stride) {
        const float *r = rays + (size_t)7 * i;
        ClosestHit ch;
        st.n = 0;
        bool found = BVHIntersectClosest<true>(sv, V3{r[0], r[1], r[2]}, V3{r[3], r[4], r[5]}, r[6], st, &ch, times[i]);
        wf_hit_record h;
        h.prim = found ? ch.prim : -1;
        h.t = found ? ch.h.t : 0; h.b0 = found ? ch.h.b0 : 0; h.b1 = found ? ch.h.b1 : 0; h.b2 = found ? ch.h.b2 : 0;
        h.nodes_visited = ch.nodesVisited; h.tris_tested = ch.trisTested; h.instance = found ? ch.inst : -1;
        out[i] = h;
    }
}
__global__ void __launch_bounds__(BLOCK) k_trace_any_timed(const SceneView sv, int n, const float *rays, const float *times, int32_t *occluded, int *stackSpill) {
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    for (int i = gtid; i < n; i += stride) {
        const float *r = rays + (size_t)7 * i;
        int v = 0, t = 0;
        st.n = 0;
        occluded[i] = BVHIntersectAny<true>(sv, V3{r[0], r[1], r[2]}, V3{r[3], r[4], r[5]}, r[6], st, &v, &t, times[i]);
    }
}
__global__ void __launch_bounds__(BLOCK) k_trace_any(const SceneView sv, int n, const float *rays, int32_t *occluded, int32_t *nodes, int32_t *tris, int *stackSpill) {
    const int gtid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
    LdsStack st{stackSpill + gtid, stride, 0};
    for (int i = gtid; i < n; i += stride) {
        const float *r = rays + (size_t)7 * i;
        int v = 0, t = 0;
        st.n = 0;
        bool occ = BVHIntersectAny(sv, V3{r[0], r[1], r[2]}, V3{r[3], r[4], r[5]}, r[6], st, &v, &t);
        occluded[i] = occ;
        if (nodes) nodes[i] = v;
        if (tris) tris[i] = t;
    }
}
__global__ void __launch_bounds__(BLOCK) k_sampler_probe(const SceneView sv, int n, const int32_t *px, const int32_t *py, const int32_t *si, int startDim, int ndims, float *out) {
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        PixelSampler s(sv);
        s.StartPixelSample(px[i], py[i], si[i], startDim);
        SamplerProbeRecord(s, ndims, out + (size_t)i * (ndims > 0 ? ndims : ndims == -3 ? 30 : 2));
    }
}

// ---------------------------------------------------------------------------------------------
static int gridFor(int n) {
    int g = (n + BLOCK - 1) / BLOCK;
    if (g < 1) g = 1;
    return g > MAX_GRID ? MAX_GRID : g;
}

struct Prof {
    wf_ctx *c;
    hipEvent_t a = nullptr, b = nullptr;
    const char *name;
    bool on;
    hipStream_t s;
    Prof(wf_ctx *c, const char *name, hipStream_t onStream = nullptr) : c(c), name(name), s(onStream ? onStream : c->stream) {
        if (c->traceLaunch) { fprintf(stderr, "[wf] launch %s\n", name); fflush(stderr); }
        on = c->profile == 1 || (c->profile == 2 && (strncmp(name, "Intersect", 9) == 0 || strcmp(name, "Route hits") == 0 || strstr(name, "Material") != nullptr ||
                                                      strncmp(name, "Sample medium", 13) == 0));   // (2: the stages bench.py prices against a roofline)
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->eventPool.empty()) { e = c->eventPool.back(); c->eventPool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get(); b = get();
        (void)hipEventRecord(a, s);
    }
    ~Prof() {
        if (c->traceLaunch) {
            hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) { fprintf(stderr, "[wf] %s: %s\n", name, hipGetErrorString(e)); fflush(stderr); }
        }
        if (!on) return;
        (void)hipEventRecord(b, s);
        c->events.push_back({name, a, b});
    }
};

#define LAUNCH(name, kernel, grid, ...)                                                    \
    do {                                                                                   \
        Prof prof_(ctx, name);                                                             \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), 0, ctx->stream, __VA_ARGS__);  \
    } while (0)

// launch KERNEL<GEN, INST> (or <INST, GEN> for the host-ray probes: ORDER = 1) for the context's scene
#define LAUNCHT_VARIANT(name, KERNEL, ORDER, ...)                                                              \
    do {                                                                                                       \
        const int gen_ = ctx->genMode;                                                                         \
        const bool inst_ = ctx->svHost.nInstances > 0;                                                         \
        if (ORDER == 0) {                                                                                      \
            if (inst_) { if (gen_ == 0) LAUNCHT(name, (KERNEL<0, true>), __VA_ARGS__); else if (gen_ == 1) LAUNCHT(name, (KERNEL<1, true>), __VA_ARGS__); else if (gen_ == 2) LAUNCHT(name, (KERNEL<2, true>), __VA_ARGS__); else LAUNCHT(name, (KERNEL<3, true>), __VA_ARGS__); } \
            else { if (gen_ == 0) LAUNCHT(name, (KERNEL<0, false>), __VA_ARGS__); else if (gen_ == 1) LAUNCHT(name, (KERNEL<1, false>), __VA_ARGS__); else if (gen_ == 2) LAUNCHT(name, (KERNEL<2, false>), __VA_ARGS__); else LAUNCHT(name, (KERNEL<3, false>), __VA_ARGS__); } \
        }                                                                                                      \
    } while (0)
// ... and with the variant given (GENV = 0 .. 3, or 4 / 5: the handing-over triangle kernels of the two-class traversal)
#define LAUNCHT_VARIANT_GEN(name, KERNEL, GENV, ...)                                                           \
    do {                                                                                                       \
        const int gen_ = (GENV);                                                                               \
        const bool inst_ = ctx->svHost.nInstances > 0;                                                         \
        if (inst_) { if (gen_ == 0) LAUNCHT(name, (KERNEL<0, true>), __VA_ARGS__); else if (gen_ == 1) LAUNCHT(name, (KERNEL<1, true>), __VA_ARGS__); else if (gen_ == 2) LAUNCHT(name, (KERNEL<2, true>), __VA_ARGS__); else if (gen_ == 3) LAUNCHT(name, (KERNEL<3, true>), __VA_ARGS__); else if (gen_ == 4) LAUNCHT(name, (KERNEL<4, true>), __VA_ARGS__); else if (gen_ == 5) LAUNCHT(name, (KERNEL<5, true>), __VA_ARGS__); else if (gen_ == 8) LAUNCHT(name, (KERNEL<8, true>), __VA_ARGS__); else LAUNCHT(name, (KERNEL<9, true>), __VA_ARGS__); } \
        else { if (gen_ == 0) LAUNCHT(name, (KERNEL<0, false>), __VA_ARGS__); else if (gen_ == 1) LAUNCHT(name, (KERNEL<1, false>), __VA_ARGS__); else if (gen_ == 2) LAUNCHT(name, (KERNEL<2, false>), __VA_ARGS__); else if (gen_ == 3) LAUNCHT(name, (KERNEL<3, false>), __VA_ARGS__); else if (gen_ == 4) LAUNCHT(name, (KERNEL<4, false>), __VA_ARGS__); else LAUNCHT(name, (KERNEL<5, false>), __VA_ARGS__); } \
    } while (0)
// the closest-hit walk with the routing split off (ctx->splitRoute)
#define LAUNCHT_CLOSEST_SPLIT(name, ...) LAUNCHT_CLOSEST_SPLIT_GEN(name, ctx->genMode, __VA_ARGS__)
#define LAUNCHT_CLOSEST_SPLIT_GEN(name, GENV, ...)                                                             \
    do {                                                                                                       \
        const int gen_ = (GENV);                                                                               \
        const bool inst_ = ctx->svHost.nInstances > 0;                                                         \
        if (inst_) { if (gen_ == 0) LAUNCHT(name, (k_closest_fast<0, true, true>), __VA_ARGS__); else if (gen_ == 1) LAUNCHT(name, (k_closest_fast<1, true, true>), __VA_ARGS__); else if (gen_ == 2) LAUNCHT(name, (k_closest_fast<2, true, true>), __VA_ARGS__); else if (gen_ == 3) LAUNCHT(name, (k_closest_fast<3, true, true>), __VA_ARGS__); else if (gen_ == 4) LAUNCHT(name, (k_closest_fast<4, true, true>), __VA_ARGS__); else if (gen_ == 5) LAUNCHT(name, (k_closest_fast<5, true, true>), __VA_ARGS__); else if (gen_ == 8) LAUNCHT(name, (k_closest_fast<8, true, true>), __VA_ARGS__); else LAUNCHT(name, (k_closest_fast<9, true, true>), __VA_ARGS__); } \
        else { if (gen_ == 0) LAUNCHT(name, (k_closest_fast<0, false, true>), __VA_ARGS__); else if (gen_ == 1) LAUNCHT(name, (k_closest_fast<1, false, true>), __VA_ARGS__); else if (gen_ == 2) LAUNCHT(name, (k_closest_fast<2, false, true>), __VA_ARGS__); else if (gen_ == 3) LAUNCHT(name, (k_closest_fast<3, false, true>), __VA_ARGS__); else if (gen_ == 4) LAUNCHT(name, (k_closest_fast<4, false, true>), __VA_ARGS__); else LAUNCHT(name, (k_closest_fast<5, false, true>), __VA_ARGS__); } \
    } while (0)
#define LAUNCHT(name, kernel, grid, ...)                                                   \
    do {                                                                                   \
        Prof prof_(ctx, name);                                                             \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(TBLOCK), 0, ctx->stream, __VA_ARGS__); \
    } while (0)

// wf_render_stats carries 64 per-depth slots (indirect_rays[64], shadow_rays[64]); deeper bounces (volumetric scenes
// set maxdepth 100 and more) accumulate in the last slot instead of running past the array
static int statDepth(int depth) { return depth < 63 ? depth : 63; }

// SceneView::gridCorners: cell (cx, cy, cz) of the (nx + 1)(ny + 1)(nz + 1) table <- the eight values a trilinear lookup with
// floor(p * res - .5) = (cx - 1, cy - 1, cz - 1) reads (wf_media.h: GridLookupPacked), zeros outside the grid
__global__ void k_pack_grid_corners(const float *v, int nx, int ny, int nz, float *out) {
    const size_t cells = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x) {
        const int cx = (int)(c % (size_t)(nx + 1)), cy = (int)((c / (size_t)(nx + 1)) % (size_t)(ny + 1)), cz = (int)(c / ((size_t)(nx + 1) * (ny + 1)));
        const int ix = cx - 1, iy = cy - 1, iz = cz - 1;
        float *o = out + 8 * wf::GridCornerIndex(nx, ny, cx, cy, cz);   // (bricked: wf_media.h)
        o[0] = wf::GridLookupI(v, nx, ny, nz, ix, iy, iz);         o[1] = wf::GridLookupI(v, nx, ny, nz, ix + 1, iy, iz);
        o[2] = wf::GridLookupI(v, nx, ny, nz, ix, iy + 1, iz);     o[3] = wf::GridLookupI(v, nx, ny, nz, ix + 1, iy + 1, iz);
        o[4] = wf::GridLookupI(v, nx, ny, nz, ix, iy, iz + 1);     o[5] = wf::GridLookupI(v, nx, ny, nz, ix + 1, iy, iz + 1);
        o[6] = wf::GridLookupI(v, nx, ny, nz, ix, iy + 1, iz + 1); o[7] = wf::GridLookupI(v, nx, ny, nz, ix + 1, iy + 1, iz + 1);
    }
}
// HIP's current device is a property of the calling host thread: a context may be driven from any thread (pbrt_amd --gpus N runs one
// host thread per device, SURVEY 8(b) "multi-GPU = one host thread per device"), so every stage entry makes the context's device current
// — unconditionally: other entry points of this library (wf_ctx_create, wf_build_bvh_sah, ...) and the application itself (torch in the
// same thread) change the current device too, so a cached "current" would go stale; hipSetDevice on the current device is a TLS write
static void useDevice(const wf_ctx *ctx) { (void)hipSetDevice(ctx->device); }
static int checkReady(wf_ctx *ctx) {
    if (!ctx) return fail(-1, "null context");
    if (!ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    if (!ctx->queuesAllocated) return fail(-1, "queues not allocated (wf_queues_alloc)");
    useDevice(ctx);
    return 0;
}

// Reference LinearBVHNode arrays (depth-first: left child = i + 1, right child = offset) -> QNode (breadth-first
// numbering per tree, quantised child boxes on the tree's own grid) + LeafTri (vertices in leaf order).  The top-level
// tree comes first; every instance definition's tree follows with its own grid (FastDef).  See wf_traverse.h.
struct FastDepths { int top = 0, def = 0, maxLeafInstances = 0; };   // levels of the four-wide trees (top level / deepest definition), most instances in one leaf
// ---- the top-level tree of a two-level scene, rebuilt over RE-BRAIDED instances (round 6; SubEntry in wf_traverse.h) ----
// One primitive of the builder: a top-level triangle / quadric (its LeafTri record of the scene's leaf order) or an instance entry.
struct TopPrim {
    float b[6];
    int kind;   // 0: the LeafTri record `idx` of the reference's leaf order; 1: instance entry `idx` (FastBVH::subs)
    int idx;
};
// Binned SAH (16 bins on the axis of the largest centroid extent, the reference's cost model: cpu/aggregates.cpp:270-370 — this tree is
// the production walk's own, nothing is pinned to it), written as reference-layout nodes (depth first: left child = i + 1, right child =
// offset) so that the four-wide collapse and the quantisation below take it like a reference tree.  A leaf holds at most four LeafTri
// records; an instance entry is always a leaf of its own, referenced by the node itself (offset = INST_FIRST + entry: the walk pops
// the entry without a LeafTri fetch).  `order` receives the primitives in leaf order.
struct TopTreeBuilder {
    std::vector<TopPrim> &P;
    std::vector<wf_bvh_node> &out;
    int leafBase;   // LeafTri index of P[0] in the new leaf order
    static double Area(const float b[6]) {
        const double dx = (double)b[3] - b[0], dy = (double)b[4] - b[1], dz = (double)b[5] - b[2];
        return dx * dy + dy * dz + dz * dx;
    }
    int Build(int lo, int hi) {
        const int me = (int)out.size();
        out.emplace_back();
        float bb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}, cb[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        bool anyEntry = false;
        for (int i = lo; i < hi; ++i) {
            const TopPrim &p = P[i];
            anyEntry = anyEntry || p.kind == 1;
            for (int a = 0; a < 3; ++a) {
                bb[a] = std::min(bb[a], p.b[a]); bb[3 + a] = std::max(bb[3 + a], p.b[3 + a]);
                const float c = 0.5f * p.b[a] + 0.5f * p.b[3 + a];
                cb[a] = std::min(cb[a], c); cb[3 + a] = std::max(cb[3 + a], c);
            }
        }
        const int count = hi - lo;
        auto makeLeaf = [&]() {
            wf_bvh_node &nd = out[me];
            for (int a = 0; a < 3; ++a) { nd.bmin[a] = bb[a]; nd.bmax[a] = bb[3 + a]; }
            nd.axis = 0; nd.pad = 0;
            nd.nprims = (uint16_t)count;
            nd.offset = (count == 1 && P[lo].kind == 1) ? INST_FIRST + P[lo].idx : leafBase + lo;
            return me;
        };
        if (count == 1) return makeLeaf();
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (cb[3 + a] - cb[a] > cb[3 + axis] - cb[axis]) axis = a;
        int mid = (lo + hi) / 2;
        const float cmin = cb[axis], cext = cb[3 + axis] - cb[axis];
        bool split = false;
        if (cext > 0) {
            constexpr int NB = 16;
            int cnt[NB] = {};
            float bbox[NB][6];
            for (int k = 0; k < NB; ++k) for (int a = 0; a < 3; ++a) { bbox[k][a] = INFINITY; bbox[k][3 + a] = -INFINITY; }
            auto binOf = [&](const TopPrim &p) {
                const float c = 0.5f * p.b[axis] + 0.5f * p.b[3 + axis];
                int k = (int)(NB * ((c - cmin) / cext));
                return k < 0 ? 0 : (k >= NB ? NB - 1 : k);
            };
            for (int i = lo; i < hi; ++i) {
                const int k = binOf(P[i]);
                ++cnt[k];
                for (int a = 0; a < 3; ++a) { bbox[k][a] = std::min(bbox[k][a], P[i].b[a]); bbox[k][3 + a] = std::max(bbox[k][3 + a], P[i].b[3 + a]); }
            }
            double costR[NB] = {};
            {
                float r[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int c = 0;
                for (int k = NB - 1; k >= 1; --k) {
                    if (cnt[k]) for (int a = 0; a < 3; ++a) { r[a] = std::min(r[a], bbox[k][a]); r[3 + a] = std::max(r[3 + a], bbox[k][3 + a]); }
                    c += cnt[k];
                    costR[k] = c ? c * Area(r) : 0;
                }
            }
            float l[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int c = 0, best = -1;
            double bestCost = 0;
            for (int k = 0; k + 1 < NB; ++k) {
                if (cnt[k]) for (int a = 0; a < 3; ++a) { l[a] = std::min(l[a], bbox[k][a]); l[3 + a] = std::max(l[3 + a], bbox[k][3 + a]); }
                c += cnt[k];
                if (c == 0 || c == count) continue;
                const double cost = c * Area(l) + costR[k + 1];
                if (best < 0 || cost < bestCost) { best = k; bestCost = cost; }
            }
            if (best >= 0) {
                const double area = Area(bb);
                const double minCost = 0.5 + (area > 0 ? bestCost / area : 0);
                if (count <= 4 && !anyEntry && !(minCost < count)) return makeLeaf();
                TopPrim *m = std::partition(P.data() + lo, P.data() + hi, [&](const TopPrim &p) { return binOf(p) <= best; });
                mid = (int)(m - P.data());
                split = mid > lo && mid < hi;
            }
        }
        if (!split) {
            // coincident centroids: up to four records share a leaf, more (or any instance entry) are dealt half and half
            if (count <= 4 && !anyEntry) return makeLeaf();
            mid = (lo + hi) / 2;
        }
        Build(lo, mid);
        const int right = Build(mid, hi);
        wf_bvh_node &nd = out[me];
        for (int a = 0; a < 3; ++a) { nd.bmin[a] = bb[a]; nd.bmax[a] = bb[3 + a]; }
        nd.axis = (uint8_t)axis; nd.pad = 0;
        nd.nprims = 0;
        nd.offset = right;
        return me;
    }
};

static bool BuildFastBVH(const wf_scene_desc *d, std::vector<QNode> *nodes, std::vector<LeafTri> *tris, std::vector<FastDef> *defs, std::vector<SubEntry> *subs,
                         FastBVH *out, FastDepths *depths) {
    const wf_bvh_node *L = d->bvh_nodes;   // (rebound to the extended array once the re-braided top-level tree has been appended)
    int n = d->n_bvh_nodes;
    const int nRef = n;                    // nodes of the reference's trees
    const int nGeom = d->n_triangles + d->n_quadrics;
    const int nPrims = nGeom + d->n_instances;  // entries of bvh_prims: every triangle / quadric once, every instance once
    if (n == 0 || (size_t)nPrims >= (size_t)INST_FIRST || d->n_instances >= INST_FIRST) return false;
    for (int i = 0; i < n; ++i)
        if (L[i].nprims > 16) return false;
    tris->resize((size_t)nPrims);
    for (int k = 0; k < nPrims; ++k) {
        int t = d->bvh_prims[k];
        if (t >= nGeom) {
            // an object instance: c.z == 4, entered by the INST kernel variants
            LeafTri lt;
            lt.a = F4{0, 0, 0, 0};
            lt.b = F4{0, 0, 0, 0};
            lt.c = F4{0, BitsToFloat((uint32_t)(t - nGeom)), 4.f, BitsToFloat(0u)};
            (*tris)[k] = lt;
            continue;
        }
        if (t >= d->n_triangles) {
            // a sphere: c.z == 3, tested by the general-primitive kernel variants from wf_quadric (object space)
            const wf_mesh &mesh = d->meshes[d->tri_mesh[t]];
            uint32_t route = mesh.material >= 0 ? (uint32_t)d->materials[mesh.material].type | (mesh.first_light >= 0 ? 16u : 0u) : 32u;
            LeafTri lt;
            lt.a = F4{0, 0, 0, 0};
            lt.b = F4{0, 0, 0, 0};
            lt.c = F4{0, BitsToFloat((uint32_t)t), 3.f, BitsToFloat(route)};
            (*tris)[k] = lt;
            continue;
        }
        const int32_t *v = d->tri_indices + 3 * (size_t)t;
        const float *p0 = d->P + 3 * (size_t)v[0], *p1 = d->P + 3 * (size_t)v[1], *p2 = d->P + 3 * (size_t)v[2];
        LeafTri lt;
        lt.a = F4{p0[0], p0[1], p0[2], p1[0]};
        lt.b = F4{p1[1], p1[2], p2[0], p2[1]};
        // IntersectTriangle's first test (shapes.cpp:172-173), hoisted to build time
        V3 q0{p0[0], p0[1], p0[2]}, q1{p1[0], p1[1], p1[2]}, q2{p2[0], p2[1], p2[2]};
        bool degenerate = LengthSquared(Cross(q2 - q0, q1 - q0)) == 0;
        // routing code of EnqueueWorkAfterIntersection (intersect.h:48-156), so that the traversal kernel needs no
        // per-hit mesh / material gathers: material type | emissive << 4 | interface << 5
        const wf_mesh &mesh = d->meshes[d->tri_mesh[t]];
        uint32_t route = 0;
        if (mesh.material >= 0) route = (uint32_t)d->materials[mesh.material].type | (mesh.first_light >= 0 ? 16u : 0u);
        else route = 32u;
        // c.z: 0 = test, 1 = degenerate (never hit), 2 = test, then the mesh's alpha texture decides (ALPHA kernel variants)
        lt.c = F4{p2[2], BitsToFloat((uint32_t)t), degenerate ? 1.f : mesh.alpha_tex >= 0 ? 2.f : 0.f, BitsToFloat(route)};
        (*tris)[k] = lt;
    }
    nodes->clear();
    bool gridOk = true;
    std::vector<int> bfsIndex(n, -1);
    // Subtree collapse: the production tree need not mirror the reference's leaves — a reference subtree holding at most
    // `collapse` primitives (contiguous in bvh_prims: leaf order is depth-first) becomes ONE leaf here.  The walk then tests
    // a superset of the triangles the reference tests, which changes no result (the exact triangle test decides, near-ties
    // are re-traced in reference order) and removes the bottom levels of dependent node fetches: instance definitions are
    // built with one primitive per leaf (maxPrimsInNode = 1, scene.cpp:1539).
    int collapse = 1;
    if (const char *e = getenv("WF_LEAF_COLLAPSE")) collapse = std::min(16, std::max(1, atoi(e)));
    std::vector<int> subFirst(n), subCount(n);
    auto fillSubRanges = [&](int from, int to) {   // nodes [from, to): a complete set of depth-first trees
        for (int i = to - 1; i >= from; --i) {
            if (L[i].nprims > 0) { subFirst[i] = L[i].offset; subCount[i] = L[i].nprims; }
            else { subFirst[i] = subFirst[i + 1]; subCount[i] = subCount[i + 1] + subCount[L[i].offset]; }
        }
    };
    fillSubRanges(0, n);
    auto leafLike = [&](int i) { return L[i].nprims > 0 || subCount[i] <= collapse; };
    // TIGHT INSTANCE BOXES (round 5).  The reference bounds an instance by the box of the eight transformed corners of its definition's
    // box (TransformedPrimitive::Bounds) — for a rotated definition up to three times the surface area of the box of the transformed
    // GEOMETRY — and the spec scene's rays entered eight instances each, three of four visits without a single primitive test.  The
    // production walk only has to visit a superset of what can be hit, so its top-level tree carries, for an instance of an all-triangle
    // definition, the box of the definition's transformed vertices (widened by 2^-18 of the magnitudes involved: the transform's rounding
    // and the instance-space ray's), clipped to the reference's box, and interior boxes re-united bottom-up.  Topology, leaves and the
    // reference-layout nodes (counting kernels, near-tie re-walk, CPU checker) are untouched: same hits, fewer entries.  WF_TIGHT_INSTANCES=0: off.
    // WF_BRAID = most entries one instance is opened into (round 6, SubEntry in wf_traverse.h; 0: one entry per instance in the reference's
    // own top-level tree, as in round 5)
    int braidMax = 2;   // (measured on the spec scene, profiles/r06_rebraid_ab_sm16.txt: 2 is the optimum — DESIGN 4.1)
    if (const char *e = getenv("WF_BRAID")) braidMax = std::min(256, std::max(0, atoi(e)));
    const bool braid = braidMax > 0 && d->n_instances > 0 && d->n_top_bvh_nodes > 0 && L[0].nprims == 0;
    std::vector<wf_bvh_node> tightTop;
    if (!braid && d->n_instances > 0 && d->n_top_bvh_nodes > 0 && !(getenv("WF_TIGHT_INSTANCES") && atoi(getenv("WF_TIGHT_INSTANCES")) == 0)) {
        const int nTop = d->n_top_bvh_nodes;
        std::vector<std::vector<int32_t>> defVerts((size_t)d->n_instance_defs);
        std::vector<char> defGeneral((size_t)d->n_instance_defs, 0);
        for (int k = 0; k < d->n_instance_defs; ++k) {
            const wf_instance_def &def = d->instance_defs[k];
            std::vector<int32_t> &v = defVerts[k];
            for (int j = def.first_prim; j < def.first_prim + def.n_prims; ++j) {
                const int t = d->bvh_prims[j];
                if (t >= d->n_triangles) { defGeneral[k] = 1; break; }
                const int32_t *ix = d->tri_indices + 3 * (size_t)t;
                v.push_back(ix[0]); v.push_back(ix[1]); v.push_back(ix[2]);
            }
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            if (v.empty()) defGeneral[k] = 1;
        }
        std::vector<std::array<float, 6>> ib((size_t)d->n_instances);
        std::vector<char> ibOk((size_t)d->n_instances, 0);
        // (the pad also carries 2^-19 of the SCENE's magnitude, like the re-braided entries' boxes below: the instance-space ray the
        //  reference decides hits with carries the rounding of the render-space origin and of the distance travelled — ADVICE r5)
        double sceneMagT = 0;
        for (int a = 0; a < 3; ++a) sceneMagT = std::max(sceneMagT, std::max(std::fabs((double)L[0].bmin[a]), std::fabs((double)L[0].bmax[a])) + ((double)L[0].bmax[a] - L[0].bmin[a]));
        for (int i = 0; i < d->n_instances; ++i) {
            const wf_instance &in = d->instances[i];
            if (defGeneral[in.def] || in.anim_plus1 != 0) continue;   // (an AnimatedPrimitive keeps the reference's motion bounds)
            const float(*m)[4] = in.render_from_instance.m;
            if (m[3][0] != 0 || m[3][1] != 0 || m[3][2] != 0 || m[3][3] != 1) continue;
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, mag = 0;
            for (int32_t vi : defVerts[in.def]) {
                const float *pp = d->P + 3 * (size_t)vi;
                for (int a = 0; a < 3; ++a) {
                    const double c = (double)m[a][0] * pp[0] + (double)m[a][1] * pp[1] + (double)m[a][2] * pp[2] + (double)m[a][3];
                    lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c);
                    mag = std::max(mag, std::fabs((double)m[a][0] * pp[0]) + std::fabs((double)m[a][1] * pp[1]) + std::fabs((double)m[a][2] * pp[2]) + std::fabs((double)m[a][3]));
                }
            }
            bool finite = true;
            for (int a = 0; a < 3; ++a) {
                const double pad = 0x1p-18 * (mag + (hi[a] - lo[a])) + 0x1p-19 * sceneMagT + 1e-30;
                ib[i][a] = (float)(lo[a] - pad); ib[i][3 + a] = (float)(hi[a] + pad);
                if (!((double)ib[i][a] <= lo[a] - 0.5 * pad)) ib[i][a] = NextFloatDown(ib[i][a]);
                if (!((double)ib[i][3 + a] >= hi[a] + 0.5 * pad)) ib[i][3 + a] = NextFloatUp(ib[i][3 + a]);
                finite = finite && std::isfinite(ib[i][a]) && std::isfinite(ib[i][3 + a]);
            }
            ibOk[i] = finite;
        }
        tightTop.assign(L, L + nTop);
        for (int i = nTop - 1; i >= 0; --i) {
            wf_bvh_node &t = tightTop[i];
            float b[6];
            bool have = false, keep = false;
            if (L[i].nprims > 0) {
                for (int j = L[i].offset; j < L[i].offset + L[i].nprims && !keep; ++j) {
                    const int pr = d->bvh_prims[j];
                    if (pr < nGeom || !ibOk[pr - nGeom]) { keep = true; break; }   // a triangle / quadric or an instance left alone: the reference's box stands
                    const std::array<float, 6> &q = ib[pr - nGeom];
                    if (!have) { for (int a = 0; a < 6; ++a) b[a] = q[a]; have = true; }
                    else for (int a = 0; a < 3; ++a) { b[a] = std::min(b[a], q[a]); b[3 + a] = std::max(b[3 + a], q[3 + a]); }
                }
            } else {
                const wf_bvh_node &c0 = tightTop[i + 1], &c1 = tightTop[L[i].offset];
                for (int a = 0; a < 3; ++a) { b[a] = std::min(c0.bmin[a], c1.bmin[a]); b[3 + a] = std::max(c0.bmax[a], c1.bmax[a]); }
                have = true;
            }
            if (keep || !have) continue;
            for (int a = 0; a < 3; ++a) {   // clipped to the reference's box (the geometry lies in both)
                t.bmin[a] = std::max(L[i].bmin[a], b[a]);
                t.bmax[a] = std::min(L[i].bmax[a], b[3 + a]);
                if (!(t.bmin[a] <= t.bmax[a])) { t.bmin[a] = L[i].bmin[a]; t.bmax[a] = L[i].bmax[a]; }
            }
        }
    }
    // the boxes the production tree is packed from: the tightened ones for the top-level tree, the reference's everywhere else
    auto BoxOf = [&](int i) -> const wf_bvh_node & { return (!tightTop.empty() && i < (int)tightTop.size()) ? tightTop[i] : L[i]; };
    // one tree: linear nodes [root, ...) reachable from root; grid written to base / cell; returns the root's QNode index
    int lastTreeDepth = 0;   // levels of the four-wide tree buildTree made last (a single leaf-like root: 1)
    std::vector<std::array<int, 4>> qKids;   // per QNode: the reference-layout nodes its four children stand for (-1: empty slot)
    auto buildTree = [&](int root, float baseOut[3], float cellOut[3]) -> int {
        lastTreeDepth = 1;
        // Quantisation grid over the root bounds.  A plane is the REAL number base + q * cell (the device never forms
        // it as a float: WalkSetRay folds base and cell into per-ray fma constants).  Every stored plane lies at least
        // `margin` outside the float box it bounds; margin = 2^-20 of the largest coordinate magnitude, which covers the
        // rounding of (base - o) and of the box's own float planes in the reference's slab test.
        double margin[3];
        for (int a = 0; a < 3; ++a) {
            double lo = L[root].bmin[a], hi = L[root].bmax[a];
            double ext = hi - lo;
            margin[a] = 0x1p-20 * (std::max(std::fabs(lo), std::fabs(hi)) + ext) + 1e-37;
            float base = (float)(lo - 2 * margin[a]);
            while ((double)base > lo - 2 * margin[a]) base = NextFloatDown(base);
            float cell = (float)((hi + 2 * margin[a] - (double)base) / 65535.0);
            if (!(cell > 0)) cell = 1e-30f;
            cell = NextFloatUp(NextFloatUp(cell));
            baseOut[a] = base;
            cellOut[a] = cell;
        }
        auto plane = [&](int q, int a) { return (double)baseOut[a] + (double)q * (double)cellOut[a]; };
        auto qlo = [&](float v, int a) {
            double target = (double)v - margin[a];
            int q = (int)std::floor((target - baseOut[a]) / cellOut[a]);
            q = std::min(std::max(q, 0), 65535);
            while (q > 0 && plane(q, a) > target) --q;
            if (plane(q, a) > target) gridOk = false;
            return (uint32_t)q;
        };
        auto qhi = [&](float v, int a) {
            double target = (double)v + margin[a];
            int q = (int)std::ceil((target - baseOut[a]) / cellOut[a]);
            q = std::min(std::max(q, 0), 65535);
            while (q < 65535 && plane(q, a) < target) ++q;
            if (plane(q, a) < target) gridOk = false;
            return (uint32_t)q;
        };
        auto leafRef = [&](int i) { return (int)~(((unsigned)subFirst[i] << 4) | (unsigned)(subCount[i] - 1)); };
        auto packBox = [&](const wf_bvh_node &b, uint32_t q[6], int slot) {
            for (int a = 0; a < 3; ++a) q[slot * 3 + a] = qlo(b.bmin[a], a) | (qhi(b.bmax[a], a) << 16);
        };
        const int qBase = (int)nodes->size();
#if WF_BVH4
        auto emptyBox = [&](uint32_t q[12], int slot) { for (int a = 0; a < 3; ++a) q[slot * 3 + a] = 0x0000ffffu; };
        auto area = [&](int i) {
            const wf_bvh_node &bx = BoxOf(i);
            double dx = (double)bx.bmax[0] - bx.bmin[0], dy = (double)bx.bmax[1] - bx.bmin[1], dz = (double)bx.bmax[2] - bx.bmin[2];
            return dx * dy + dy * dz + dz * dx;
        };
        auto packBox4 = [&](const wf_bvh_node &b, uint32_t q[12], int slot) {
            for (int a = 0; a < 3; ++a) q[slot * 3 + a] = qlo(b.bmin[a], a) | (qhi(b.bmax[a], a) << 16);
        };
        if (leafLike(root)) {
            QNode qn{};
            packBox4(BoxOf(root), qn.q, 0);
            qn.child[0] = leafRef(root);
            for (int c = 1; c < 4; ++c) { emptyBox(qn.q, c); qn.child[c] = NODE_NONE; }
            nodes->push_back(qn);
            qKids.push_back({root, -1, -1, -1});
            return qBase;
        }
        // four-way collapse of the binary tree: a node's children are its two binary children, the largest interior ones
        // replaced by their own children until there are four (or only leaves are left); breadth-first numbering
        std::vector<int> order;
        order.push_back(root);
        bfsIndex[root] = qBase;
        std::vector<std::array<int, 4>> kidsOf;
        std::vector<int> level{1};
        for (size_t h = 0; h < order.size(); ++h) {
            int i = order[h];
            lastTreeDepth = std::max(lastTreeDepth, level[h] + 1);   // + 1: the leaves hanging off this node
            std::array<int, 4> kids = {i + 1, (int)L[i].offset, -1, -1};
            int nk = 2;
            while (nk < 4) {
                int best = -1;
                for (int k = 0; k < nk; ++k)
                    if (!leafLike(kids[k]) && (best < 0 || area(kids[k]) > area(kids[best]))) best = k;
                if (best < 0) break;
                int c = kids[best];
                kids[best] = c + 1;
                kids[nk++] = L[c].offset;
            }
            for (int k = 0; k < nk; ++k)
                if (!leafLike(kids[k])) { bfsIndex[kids[k]] = qBase + (int)order.size(); order.push_back(kids[k]); level.push_back(level[h] + 1); }
            kidsOf.push_back(kids);
        }
        nodes->resize((size_t)qBase + order.size());
        qKids.insert(qKids.end(), kidsOf.begin(), kidsOf.end());
        for (size_t h = 0; h < order.size(); ++h) {
            QNode qn{};
            for (int c = 0; c < 4; ++c) {
                int k = kidsOf[h][c];
                if (k < 0) { emptyBox(qn.q, c); qn.child[c] = NODE_NONE; continue; }
                packBox4(BoxOf(k), qn.q, c);
                qn.child[c] = !leafLike(k) ? bfsIndex[k] : leafRef(k);
            }
            (*nodes)[(size_t)qBase + h] = qn;
        }
#else
        if (leafLike(root)) {
            // the whole tree is one leaf: a root node whose two children are both that leaf (testing it twice
            // changes neither the closest hit nor occlusion)
            QNode qn{};
            packBox(L[root], qn.q, 0);
            packBox(L[root], qn.q, 1);
            qn.left = leafRef(root);
            qn.right = leafRef(root);
            nodes->push_back(qn);
            return qBase;
        }
        lastTreeDepth = 64;   // (two-wide legacy layout: the reference's own bound, nodesToVisit[64])
        // breadth-first numbering of the interior nodes
        std::vector<int> order;
        order.push_back(root);
        bfsIndex[root] = qBase;
        for (size_t h = 0; h < order.size(); ++h) {
            int i = order[h];
            for (int c : {i + 1, (int)L[i].offset})
                if (!leafLike(c)) { bfsIndex[c] = qBase + (int)order.size(); order.push_back(c); }
        }
        nodes->resize((size_t)qBase + order.size());
        for (size_t h = 0; h < order.size(); ++h) {
            int i = order[h];
            int l = i + 1, r = L[i].offset;
            QNode qn{};
            packBox(L[l], qn.q, 0);
            packBox(L[r], qn.q, 1);
            qn.left = !leafLike(l) ? bfsIndex[l] : leafRef(l);
            qn.right = !leafLike(r) ? bfsIndex[r] : leafRef(r);
            (*nodes)[(size_t)qBase + h] = qn;
        }
#endif
        return qBase;
    };
    // the definitions' trees first (the entries of the re-braided top-level tree name their nodes), then the top-level tree; afterwards
    // the top-level tree is moved to the FRONT of the array: the walk starts at node 0 and caches nodes [0, TOP_NODES) in LDS
    for (int i = 0; i < nRef; ++i)
        if (leafLike(i)) {
            int ni = 0;
            for (int k = 0; k < subCount[i]; ++k) ni += d->bvh_prims[subFirst[i] + k] >= nGeom;
            depths->maxLeafInstances = std::max(depths->maxLeafInstances, ni);
        }
    defs->clear();
    for (int k = 0; k < d->n_instance_defs; ++k) {
        FastDef fd{};
        fd.root = d->instance_defs[k].bvh_root >= 0 ? buildTree(d->instance_defs[k].bvh_root, fd.base, fd.cell) : -1;
        if (d->instance_defs[k].bvh_root >= 0) depths->def = std::max(depths->def, lastTreeDepth);
        defs->push_back(fd);
    }
    const int nDefQ = (int)nodes->size();
    subs->clear();
    std::vector<wf_bvh_node> Lx;
    int topRoot = 0;
    if (braid && WF_BVH4) {
        // ---- partial re-braiding: the entries of every instance
        const int nTop = d->n_top_bvh_nodes;
        std::vector<char> defGeneral((size_t)d->n_instance_defs, 0);
        for (int k = 0; k < d->n_instance_defs; ++k) {
            const wf_instance_def &def = d->instance_defs[k];
            if (def.bvh_root < 0 || def.n_prims <= 0) { defGeneral[k] = 1; continue; }
            for (int j = def.first_prim; j < def.first_prim + def.n_prims; ++j)
                if (d->bvh_prims[j] >= d->n_triangles) { defGeneral[k] = 1; break; }
        }
        double sceneMag = 0;
        for (int a = 0; a < 3; ++a) sceneMag = std::max(sceneMag, std::max(std::fabs((double)L[0].bmin[a]), std::fabs((double)L[0].bmax[a])) + ((double)L[0].bmax[a] - L[0].bmin[a]));
        struct Entry { int bnode; float b[6]; };
        std::vector<std::vector<Entry>> entriesOf((size_t)d->n_instances);
        // the box of a definition's subtree under an instance's transformation: its triangles' transformed vertices (double arithmetic),
        // widened by 2^-18 of the magnitudes involved plus 2^-19 of the SCENE's (the instance-space ray the reference decides hits with
        // carries the rounding of the render-space ray's origin and of the distance travelled — ADVICE r5: not of the instance's own
        // coordinates only)
        auto subtreeBox = [&](const wf_instance &in, int bnode, float b[6]) {
            const float(*m)[4] = in.render_from_instance.m;
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, mag = 0;
            for (int j = subFirst[bnode]; j < subFirst[bnode] + subCount[bnode]; ++j) {
                const LeafTri &lt = (*tris)[j];
                const float v[3][3] = {{lt.a.x, lt.a.y, lt.a.z}, {lt.a.w, lt.b.x, lt.b.y}, {lt.b.z, lt.b.w, lt.c.x}};
                for (int q = 0; q < 3; ++q)
                    for (int a = 0; a < 3; ++a) {
                        const double t0 = (double)m[a][0] * v[q][0], t1 = (double)m[a][1] * v[q][1], t2 = (double)m[a][2] * v[q][2];
                        const double c = t0 + t1 + t2 + (double)m[a][3];
                        lo[a] = std::min(lo[a], c); hi[a] = std::max(hi[a], c);
                        mag = std::max(mag, std::fabs(t0) + std::fabs(t1) + std::fabs(t2) + std::fabs((double)m[a][3]));
                    }
            }
            bool finite = true;
            for (int a = 0; a < 3; ++a) {
                const double pad = 0x1p-18 * (mag + (hi[a] - lo[a])) + 0x1p-19 * sceneMag + 1e-30;
                b[a] = (float)(lo[a] - pad); b[3 + a] = (float)(hi[a] + pad);
                if (!((double)b[a] <= lo[a] - 0.5 * pad)) b[a] = NextFloatDown(b[a]);
                if (!((double)b[3 + a] >= hi[a] + 0.5 * pad)) b[3 + a] = NextFloatUp(b[3 + a]);
                finite = finite && std::isfinite(b[a]) && std::isfinite(b[3 + a]);
            }
            return finite;
        };
        auto refOf = [&](int bnode) { return leafLike(bnode) ? (int)~(((unsigned)subFirst[bnode] << 4) | (unsigned)(subCount[bnode] - 1)) : bfsIndex[bnode]; };
        double minFrac = 1.0 / 64;   // an entry smaller than this fraction of the instance's own box is not opened further
        if (const char *e = getenv("WF_BRAID_MIN_FRAC")) minFrac = atof(e);
        auto openInstance = [&](int i) {
            const wf_instance &in = d->instances[i];
            std::vector<Entry> &es = entriesOf[i];
            if (in.def < 0 || in.def >= d->n_instance_defs || defGeneral[in.def]) return;
            if (in.anim_plus1 != 0) return;   // an AnimatedPrimitive: one entry under the reference's motion bounds (its leaf's box)
            const float(*m)[4] = in.render_from_instance.m;
            if (m[3][0] != 0 || m[3][1] != 0 || m[3][2] != 0 || m[3][3] != 1) return;
            Entry root;
            root.bnode = d->instance_defs[in.def].bvh_root;
            if (!subtreeBox(in, root.bnode, root.b)) return;
            const double rootArea = TopTreeBuilder::Area(root.b);
            es.push_back(root);
            while ((int)es.size() < braidMax) {
                int best = -1;
                double bestArea = minFrac * rootArea;
                for (int k = 0; k < (int)es.size(); ++k) {
                    if (leafLike(es[k].bnode)) continue;
                    const double ar = TopTreeBuilder::Area(es[k].b);
                    if (ar > bestArea) { best = k; bestArea = ar; }
                }
                if (best < 0) break;
                const std::array<int, 4> &kids = qKids[(size_t)bfsIndex[es[best].bnode]];
                int nk = 0;
                for (int c = 0; c < 4; ++c) nk += kids[c] >= 0;
                if ((int)es.size() - 1 + nk > braidMax) break;
                Entry ch[4];
                bool ok = true;
                int w = 0;
                for (int c = 0; c < 4 && ok; ++c)
                    if (kids[c] >= 0) { ch[w].bnode = kids[c]; ok = subtreeBox(in, kids[c], ch[w].b); ++w; }
                if (!ok) break;
                es[best] = ch[0];
                for (int c = 1; c < w; ++c) es.push_back(ch[c]);
            }
        };
        {
            unsigned nThreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
            if ((unsigned)d->n_instances < 4 * nThreads) nThreads = 1;
            std::atomic<int> next{0};
            auto worker = [&]() { for (int i; (i = next.fetch_add(1)) < d->n_instances;) openInstance(i); };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(worker);
            worker();
            for (std::thread &t : pool) t.join();
        }
        // ---- the primitives of the new top-level tree, in the reference's leaf order (deterministic)
        std::vector<TopPrim> P;
        P.reserve((size_t)subCount[0] + (size_t)d->n_instances * 4);
        for (int i = 0; i < nTop; ++i) {
            if (L[i].nprims == 0) continue;
            for (int j = L[i].offset; j < L[i].offset + L[i].nprims; ++j) {
                const int t = d->bvh_prims[j];
                TopPrim p;
                if (t < d->n_triangles) {
                    const LeafTri &lt = (*tris)[j];
                    const float v[3][3] = {{lt.a.x, lt.a.y, lt.a.z}, {lt.a.w, lt.b.x, lt.b.y}, {lt.b.z, lt.b.w, lt.c.x}};
                    for (int a = 0; a < 3; ++a) { p.b[a] = std::min(v[0][a], std::min(v[1][a], v[2][a])); p.b[3 + a] = std::max(v[0][a], std::max(v[1][a], v[2][a])); }
                    p.kind = 0; p.idx = j;
                    P.push_back(p);
                } else if (t < nGeom) {   // a quadric / patch / curve: the reference's leaf box bounds it
                    for (int a = 0; a < 3; ++a) { p.b[a] = L[i].bmin[a]; p.b[3 + a] = L[i].bmax[a]; }
                    p.kind = 0; p.idx = j;
                    P.push_back(p);
                } else {
                    const int ii = t - nGeom;
                    const wf_instance &in = d->instances[ii];
                    if (entriesOf[ii].empty()) {   // left alone (general primitives inside, a projective matrix): one entry at the definition's root, the reference's leaf box
                        for (int a = 0; a < 3; ++a) { p.b[a] = L[i].bmin[a]; p.b[3 + a] = L[i].bmax[a]; }
                        p.kind = 1; p.idx = (int)subs->size();
                        const int root = (in.def >= 0 && in.def < d->n_instance_defs) ? d->instance_defs[in.def].bvh_root : -1;
                        subs->push_back(SubEntry{ii, root >= 0 ? refOf(root) : NODE_NONE});
                        P.push_back(p);
                    } else
                        for (const Entry &e : entriesOf[ii]) {
                            for (int a = 0; a < 6; ++a) p.b[a] = e.b[a];
                            p.kind = 1; p.idx = (int)subs->size();
                            subs->push_back(SubEntry{ii, refOf(e.bnode)});
                            P.push_back(p);
                        }
                }
            }
        }
        if (P.empty() || subs->size() >= (size_t)INST_FIRST) return false;
        // ---- the tree, as reference-layout nodes behind the reference's own
        const int leafBase = (int)tris->size();
        Lx.assign(L, L + n);
        TopTreeBuilder tb{P, Lx, leafBase};
        // (node indices of the builder are positions in Lx: it appends)
        topRoot = tb.Build(0, (int)P.size());
        tris->resize((size_t)leafBase + P.size());
        for (size_t k = 0; k < P.size(); ++k) {
            if (P[k].kind == 0) (*tris)[(size_t)leafBase + k] = (*tris)[(size_t)P[k].idx];
            else (*tris)[(size_t)leafBase + k].c = F4{0, BitsToFloat((uint32_t)P[k].idx), 4.f, BitsToFloat(0u)};   // (never read: an entry is a leaf of its own)
        }
        if (tris->size() >= (size_t)INST_FIRST) return false;
        L = Lx.data();
        n = (int)Lx.size();
        subFirst.resize(n); subCount.resize(n); bfsIndex.resize(n, -1);
        fillSubRanges(nRef, n);
        if (const char *e = getenv("WF_BRAID_VERBOSE")) if (atoi(e)) fprintf(stderr, "[wf] re-braided top-level tree: %d instances -> %zu entries, %zu primitives, %d nodes\n", d->n_instances, subs->size(), P.size(), n - nRef);
    } else {
        // one entry per instance, at its definition's root
        for (int i = 0; i < d->n_instances; ++i) {
            const int def = d->instances[i].def;
            const int root = (def >= 0 && def < d->n_instance_defs) ? d->instance_defs[def].bvh_root : -1;
            subs->push_back(SubEntry{i, root < 0 ? NODE_NONE : (leafLike(root) ? (int)~(((unsigned)subFirst[root] << 4) | (unsigned)(subCount[root] - 1)) : bfsIndex[root])});
        }
    }
    buildTree(topRoot, out->base, out->cell);
    depths->top = lastTreeDepth;
    {
        // the top-level tree to the front
        const int nAll = (int)nodes->size(), nTopQ = nAll - nDefQ;
        auto remap = [&](int r) { return r < 0 ? r : (r >= nDefQ ? r - nDefQ : r + nTopQ); };
        std::vector<QNode> moved((size_t)nAll);
        for (int i = 0; i < nAll; ++i) {
            QNode qn = (*nodes)[(size_t)i];
#if WF_BVH4
            for (int c = 0; c < 4; ++c) qn.child[c] = remap(qn.child[c]);
#else
            qn.left = remap(qn.left); qn.right = remap(qn.right);
#endif
            moved[(size_t)remap(i)] = qn;
        }
        nodes->swap(moved);
        for (FastDef &fd : *defs) fd.root = fd.root < 0 ? 0 : remap(fd.root);
        for (SubEntry &se : *subs) se.node = remap(se.node);
    }
    {
        double ext = 0;
        for (int a = 0; a < 3; ++a) ext = std::max(ext, std::max(std::fabs((double)L[0].bmin[a]), std::fabs((double)L[0].bmax[a])) + ((double)L[0].bmax[a] - L[0].bmin[a]));
        const double band = d->n_quadrics > 0 ? 0x1p-10 : 0x1p-20;   // (FastBVH::tieRel: quadric hits are accepted by interval bounds)
        out->absBand = (float)(band * ext);
        out->tieRel = (float)(1 + band);
        out->absBandTri = (float)(0x1p-20 * ext);   // the band of a triangle / triangle pair (wf_traverse.h, WalkAccept<PAIRS>)
        out->tieRelTri = (float)(1 + 0x1p-20);
        out->firstGeneral = d->n_quadrics > 0 ? d->n_triangles : INT_MAX;
    }
    if (!gridOk) return false;
    out->nNodes = (int)nodes->size();
    return true;
}

// ---- host-only self-check of the production layout (wf_debug_fastbvh_check; CPU suite) ----------------------------------------------
// Walks the QNode / LeafTri / SubEntry arrays on the HOST with random rays, in double arithmetic and without pruning by distance, and
// checks that every triangle a ray really hits (brute force over the top-level triangles and every (instance, triangle) pair, in the
// instance's space) is among the triangles the walk tests — the one property the production tree owes (wf_traverse.h: "a superset of
// visited nodes, the exact triangle test decides").  No device is needed: BuildFastBVH is host code.
namespace {
struct CheckWalk {
    const std::vector<QNode> &nodes;
    const std::vector<LeafTri> &tris;
    const std::vector<FastDef> &defs;
    const std::vector<SubEntry> &subs;
    const wf_scene_desc *d;
    const FastBVH &top;
    std::vector<std::pair<int, int>> tested;   // (triangle id, instance or -1)
    long long nodesVisited = 0, entries = 0;
    static bool Slab(const float base[3], const float cell[3], const uint32_t q[3], const double o[3], const double dir[3]) {
        double t0 = 0, t1 = 1e300;
        for (int a = 0; a < 3; ++a) {
            const double lo = (double)base[a] + (double)(q[a] & 0xffffu) * (double)cell[a], hi = (double)base[a] + (double)(q[a] >> 16) * (double)cell[a];
            if (lo > hi) return false;   // an empty slot
            if (dir[a] == 0) { if (o[a] < lo || o[a] > hi) return false; continue; }
            double tn = (lo - o[a]) / dir[a], tf = (hi - o[a]) / dir[a];
            if (tn > tf) std::swap(tn, tf);
            t0 = std::max(t0, tn); t1 = std::min(t1, tf);
        }
        return t0 <= t1;
    }
    void Leaf(int ref, int inst, const double o[3], const double dir[3]) {
        const unsigned r = ~(unsigned)ref;
        const int first = (int)(r >> 4), count = (int)(r & 15u) + 1;
        if (first >= INST_FIRST) { Enter(first - INST_FIRST, o, dir); return; }
        for (int i = 0; i < count; ++i) {
            const LeafTri &lt = tris[(size_t)first + i];
            if (lt.c.z == 4.f) { if (inst < 0) Enter((int)FloatToBits(lt.c.y), o, dir); continue; }
            tested.push_back({(int)FloatToBits(lt.c.y), inst});
        }
    }
    void Tree(int ref, const float base[3], const float cell[3], int inst, const double o[3], const double dir[3]) {
        std::vector<int> st{ref};
        while (!st.empty()) {
            const int r = st.back();
            st.pop_back();
            if (r == NODE_NONE) continue;
            if (r < 0) { Leaf(r, inst, o, dir); continue; }
            ++nodesVisited;
            const QNode &qn = nodes[(size_t)r];
#if WF_BVH4
            for (int c = 0; c < 4; ++c)
                if (qn.child[c] != NODE_NONE && Slab(base, cell, qn.q + 3 * c, o, dir)) st.push_back(qn.child[c]);
#else
            if (Slab(base, cell, qn.q, o, dir)) st.push_back(qn.left);
            if (Slab(base, cell, qn.q + 3, o, dir)) st.push_back(qn.right);
#endif
        }
    }
    void Enter(int entry, const double oW[3], const double dW[3]) {
        ++entries;
        const SubEntry se = subs[(size_t)entry];
        if (se.node == NODE_NONE) return;
        const wf_instance &in = d->instances[se.inst];
        const float(*mi)[4] = in.render_from_instance.mInv;
        double o[3], dir[3];
        for (int a = 0; a < 3; ++a) {
            o[a] = (double)mi[a][0] * oW[0] + (double)mi[a][1] * oW[1] + (double)mi[a][2] * oW[2] + (double)mi[a][3];
            dir[a] = (double)mi[a][0] * dW[0] + (double)mi[a][1] * dW[1] + (double)mi[a][2] * dW[2];
        }
        const FastDef &fd = defs[(size_t)in.def];
        Tree(se.node, fd.base, fd.cell, se.inst, o, dir);
    }
};
// does the ray hit the triangle well inside (barycentrics > eps, in front of the origin)?  Moeller-Trumbore in double arithmetic
bool HitsClearly(const double o[3], const double dir[3], const float *p0, const float *p1, const float *p2) {
    double e1[3], e2[3], pv[3], tv[3], qv[3];
    for (int a = 0; a < 3; ++a) { e1[a] = (double)p1[a] - p0[a]; e2[a] = (double)p2[a] - p0[a]; tv[a] = o[a] - p0[a]; }
    pv[0] = dir[1] * e2[2] - dir[2] * e2[1]; pv[1] = dir[2] * e2[0] - dir[0] * e2[2]; pv[2] = dir[0] * e2[1] - dir[1] * e2[0];
    const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    const double scale = std::sqrt((e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) * (e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]) * (dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]));
    if (!(std::fabs(det) > 1e-9 * scale)) return false;
    const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
    qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
    const double v = (dir[0] * qv[0] + dir[1] * qv[1] + dir[2] * qv[2]) / det;
    const double t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) / det;
    return u > 1e-4 && v > 1e-4 && u + v < 1 - 1e-4 && t > 1e-6;
}
}  // namespace

extern "C" {

const char *wf_last_error(void) { return g_err; }
int wf_abi_version(void) { return WF_ABI_VERSION; }

int wf_debug_fastbvh_check(const wf_scene_desc *d, int n_rays, uint64_t seed, int64_t out[8]) {
    if (!d || !out || n_rays < 0) return fail(-1, "wf_debug_fastbvh_check: bad arguments");
    std::vector<QNode> qn;
    std::vector<LeafTri> lt;
    std::vector<FastDef> fdefs;
    std::vector<SubEntry> fsubs;
    FastBVH fast{};
    FastDepths fdep;
    for (int k = 0; k < 8; ++k) out[k] = 0;
    if (!BuildFastBVH(d, &qn, &lt, &fdefs, &fsubs, &fast, &fdep)) return fail(-1, "wf_debug_fastbvh_check: the scene has no production layout");
    out[0] = (int64_t)qn.size(); out[1] = (int64_t)lt.size(); out[2] = (int64_t)fsubs.size(); out[3] = fdep.top;
    // structure: every child reference names a node / a leaf run / an entry inside the arrays
    for (size_t i = 0; i < qn.size(); ++i)
#if WF_BVH4
        for (int c = 0; c < 4; ++c) {
            const int r = qn[i].child[c];
#else
        for (int c = 0; c < 2; ++c) {
            const int r = c ? qn[i].right : qn[i].left;
#endif
            if (r == NODE_NONE) continue;
            if (r >= 0) { if ((size_t)r >= qn.size()) return fail(-1, "wf_debug_fastbvh_check: node %zu child %d out of range", i, c); continue; }
            const unsigned u = ~(unsigned)r;
            const size_t first = u >> 4, count = (u & 15u) + 1;
            if (first >= (size_t)INST_FIRST ? first - INST_FIRST >= fsubs.size() : first + count > lt.size()) return fail(-1, "wf_debug_fastbvh_check: node %zu child %d: leaf run out of range", i, c);
        }
    for (const SubEntry &se : fsubs)
        if (se.inst < 0 || se.inst >= d->n_instances || (se.node >= 0 && (size_t)se.node >= qn.size())) return fail(-1, "wf_debug_fastbvh_check: entry out of range");
    // coverage with random rays through the scene's box
    uint64_t rng = seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) * (1.0 / 9007199254740992.0); };
    const wf_bvh_node &root = d->bvh_nodes[0];
    CheckWalk cw{qn, lt, fdefs, fsubs, d, fast, {}};
    const int nGeom = d->n_triangles + d->n_quadrics;
    for (int r = 0; r < n_rays; ++r) {
        double o[3], dir[3], tgt[3];
        for (int a = 0; a < 3; ++a) {
            const double lo = root.bmin[a], hi = root.bmax[a], ext = hi - lo;
            o[a] = lo - 0.1 * ext + 1.2 * ext * rnd();
            tgt[a] = lo + ext * rnd();
            dir[a] = tgt[a] - o[a];
        }
        cw.tested.clear();
        cw.Tree(0, fast.base, fast.cell, -1, o, dir);
        std::sort(cw.tested.begin(), cw.tested.end());
        out[6] += (int64_t)cw.tested.size();
        auto check = [&](int tri, int inst, const double oo[3], const double dd[3]) {
            const int32_t *v = d->tri_indices + 3 * (size_t)tri;
            if (!HitsClearly(oo, dd, d->P + 3 * (size_t)v[0], d->P + 3 * (size_t)v[1], d->P + 3 * (size_t)v[2])) return;
            ++out[4];
            if (!std::binary_search(cw.tested.begin(), cw.tested.end(), std::make_pair(tri, inst))) ++out[5];
        };
        // the top-level primitives and, per instance, its definition's
        int nTopPrims = 0;
        for (int i = 0; i < (d->n_top_bvh_nodes > 0 ? d->n_top_bvh_nodes : d->n_bvh_nodes); ++i) nTopPrims += d->bvh_nodes[i].nprims;
        for (int j = 0; j < nTopPrims; ++j) {
            const int t = d->bvh_prims[j];
            if (t < d->n_triangles) { check(t, -1, o, dir); continue; }
            if (t < nGeom) continue;
            const int ii = t - nGeom;
            const wf_instance &in = d->instances[ii];
            if (in.def < 0 || in.def >= d->n_instance_defs) continue;
            const float(*mi)[4] = in.render_from_instance.mInv;
            double oI[3], dI[3];
            for (int a = 0; a < 3; ++a) {
                oI[a] = (double)mi[a][0] * o[0] + (double)mi[a][1] * o[1] + (double)mi[a][2] * o[2] + (double)mi[a][3];
                dI[a] = (double)mi[a][0] * dir[0] + (double)mi[a][1] * dir[1] + (double)mi[a][2] * dir[2];
            }
            const wf_instance_def &def = d->instance_defs[in.def];
            for (int k = def.first_prim; k < def.first_prim + def.n_prims; ++k)
                if (d->bvh_prims[k] < d->n_triangles) check(d->bvh_prims[k], ii, oI, dI);
        }
    }
    out[7] = cw.entries;
    out[3] = cw.nodesVisited;
    return 0;
}

// The stream's scratch (private segment) grows whenever a kernel needs more per lane than any kernel before it; on about a third of the
// pool's boxes every such growth costs the launch that triggers it 20-30 ms (round 4: the first frame of a process took 200 ms instead of
// 112: "Handle escaped rays" 1264 B, "Handle emitters" 1424 B, conductor 1468 B, coated diffuse 1500 B per lane — four steps in launch
// order).  One launch of a kernel whose private segment is at least the largest of the library's (k_eval_material<7, 2>: 3372 B) pays for
// the growth once, when the context is created.  WF_SCRATCH_PRIME=0 skips it.
constexpr int SCRATCH_PRIME_WORDS = 896;   // 3584 B per lane
__global__ void k_scratch_prime(int *out, int n) {
    volatile int buf[SCRATCH_PRIME_WORDS];
    for (int i = 0; i < n; ++i) buf[(i * 131) % SCRATCH_PRIME_WORDS] = i;
    out[threadIdx.x] = buf[(n * 7) % SCRATCH_PRIME_WORDS];
}
int wf_ctx_create(int device, wf_ctx **out) {
    if (!out) return fail(-1, "null out");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(-1, "no HIP device visible: libwfhip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(-1, "device %d out of range (0..%d)", device, ndev - 1);
    HIPCHK(hipSetDevice(device));
    wf_ctx *c = new wf_ctx();
    c->device = device;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming));
    if (const char *e = getenv("WF_OVERLAP_RETRACE")) c->overlapRetrace = atoi(e);
    if (const char *e = getenv("WF_MAT_STREAMS")) c->matStreams = atoi(e);
    HIPCHK(hipEventCreateWithFlags(&c->evMatFork, hipEventDisableTiming));
    c->traceLaunch = getenv("WF_TRACE_LAUNCH") != nullptr;
    if (const char *e = getenv("WF_SCRATCH_PRIME"); !e || atoi(e) != 0) {
        int *tmp = nullptr;
        HIPCHK(hipMalloc(&tmp, 64 * sizeof(int)));
        hipLaunchKernelGGL(k_scratch_prime, dim3(1), dim3(64), 0, c->stream, tmp, 8);
        hipLaunchKernelGGL(k_scratch_prime, dim3(1), dim3(64), 0, c->stream2, tmp, 8);
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipStreamSynchronize(c->stream2));
        HIPCHK(hipFree(tmp));
    }
    *out = c;
    return 0;
}

int wf_ctx_destroy(wf_ctx *ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (void *p : ctx->allocs) (void)hipFree(p);
    for (auto &e : ctx->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto e : ctx->eventPool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->evFork) (void)hipEventDestroy(ctx->evFork);
    if (ctx->evJoin) (void)hipEventDestroy(ctx->evJoin);
    if (ctx->evMatFork) (void)hipEventDestroy(ctx->evMatFork);
    for (int m = 0; m < WF_MAT_NTYPES; ++m) {
        if (ctx->matStream[m]) (void)hipStreamDestroy(ctx->matStream[m]);
        if (ctx->evMatJoin[m]) (void)hipEventDestroy(ctx->evMatJoin[m]);
    }
    delete ctx;
    return 0;
}

int wf_sync(wf_ctx *ctx) {
    if (!ctx) return fail(-1, "null context");
    useDevice(ctx);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    if (ctx->dbgWords) {
        int overflow = 0;
        HIPCHK(hipMemcpy(&overflow, ctx->dbgWords + 1, sizeof(int), hipMemcpyDeviceToHost));
        if (overflow) return fail(-1, "traversal stack overflow: a walk needed more than %d + %d node-stack entries (results are incomplete)", TSTACK, ctx->spillRows);
        int unresolved = 0;
        HIPCHK(hipMemcpy(&unresolved, ctx->dbgWords + 6, sizeof(int), hipMemcpyDeviceToHost));
        if (unresolved & 1) return fail(-1, "a near-tie re-walk found no hit where the production walk had one (results are incomplete)");
        if (unresolved & 2) return fail(-1, "a near-tie queue entry failed its validation (ray index, bound or launch tag): results are incomplete");
        int fatal = 0;
        HIPCHK(hipMemcpy(&fatal, ctx->dbgWords + 7, sizeof(int), hipMemcpyDeviceToHost));
        // the reference's LOG_FATAL / CHECK inside a kernel body (wf_scene.h: WF_FATAL_*), e.g. a sample drawn from an emissive curve
        if (fatal) return fail(-1, "%s", FatalMessage(fatal));
        if (getenv("WF_DEBUG_DRAIN")) {
            int h[8];
            HIPCHK(hipMemcpy(h, ctx->dbgWords, sizeof(h), hipMemcpyDeviceToHost));
            fprintf(stderr, "[drain] spilled %d, re-walks %d, pushes %d, repeated re-walks %d, unresolved %d\n", h[0], h[2], h[4], h[5], h[6]);
#if WF_WALK_STATS
            unsigned long long c[16];
            HIPCHK(hipMemcpy(c, ctx->dbgWords + 8, sizeof(c), hipMemcpyDeviceToHost));
            const char *ph[4] = {"interior", "leaf", "transition", "refill"};
            for (int k = 0; k < 2; ++k)
                for (int q = 0; q < 4; ++q)
                    fprintf(stderr, "[walk stats] %s %-26s wave iterations %llu, lanes %llu (%.1f per iteration)\n", k ? "any-hit" : "closest", ph[q], c[k * 8 + 2 * q], c[k * 8 + 2 * q + 1],
                            c[k * 8 + 2 * q] ? (double)c[k * 8 + 2 * q + 1] / (double)c[k * 8 + 2 * q] : 0.0);
#endif
        }
    }
    return 0;
}
int wf_debug_counters(wf_ctx *ctx, uint64_t out[4], int reset) {
    if (!ctx || !ctx->sceneLoaded || !out) return fail(-1, "wf_debug_counters: no scene uploaded");
    int h[4] = {0, 0, 0, 0};
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(h, ctx->dbgWords, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out[i] = (uint64_t)(unsigned)h[i];
    if (reset) HIPCHK(hipMemset(ctx->dbgWords, 0, sizeof(h)));
    return 0;
}
void *wf_stream(wf_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int wf_scene_upload(wf_ctx *ctx, const wf_scene_desc *d) {
    if (!ctx || !d) return fail(-1, "null argument");
    if (d->abi_version != WF_ABI_VERSION) return fail(-1, "ABI version mismatch: desc %d, library %d", d->abi_version, WF_ABI_VERSION);
    if (ctx->sceneLoaded) return fail(-1, "a scene is already uploaded to this context");
    HIPCHK(hipSetDevice(ctx->device));
    SceneView &sv = ctx->svHost;
    int e = 0;
    if ((e = devUpload(ctx, &sv.P, d->P, (size_t)3 * d->n_vertices))) return e;
    if ((e = devUpload(ctx, &sv.N, d->N, (size_t)3 * d->n_vertices))) return e;
    if ((e = devUpload(ctx, &sv.UV, d->UV, (size_t)2 * d->n_vertices))) return e;
    if (d->n_tangents > 0 && (e = devUpload(ctx, &sv.S, d->S, (size_t)3 * d->n_tangents))) return e;
    if ((e = devUpload(ctx, &sv.triIndices, d->tri_indices, (size_t)3 * d->n_triangles))) return e;
    if ((e = devUpload(ctx, &sv.triMesh, d->tri_mesh, (size_t)d->n_triangles + d->n_quadrics))) return e;
    // the de-indexed per-triangle vertex records of the material stage (wf_scene.h ShadeTri); WF_SHADE_TRIS=0: the indexed tables only
    sv.shadeTris = nullptr;
    if (d->n_triangles > 0 && !(getenv("WF_SHADE_TRIS") && atoi(getenv("WF_SHADE_TRIS")) == 0)) {
        std::vector<ShadeTri> st((size_t)d->n_triangles);
        for (size_t t = 0; t < (size_t)d->n_triangles; ++t) {
            const int32_t *ix = d->tri_indices + 3 * t;
            const int flags = d->meshes[d->tri_mesh[t]].flags;
            ShadeTri r{};
            for (int k = 0; k < 3; ++k) {
                for (int a = 0; a < 3; ++a) r.p[3 * k + a] = d->P[3 * (size_t)ix[k] + a];
                if ((flags & WF_MESH_HAS_N) && d->N) for (int a = 0; a < 3; ++a) r.n[3 * k + a] = d->N[3 * (size_t)ix[k] + a];
                if ((flags & WF_MESH_HAS_UV) && d->UV) for (int a = 0; a < 2; ++a) r.uv[2 * k + a] = d->UV[2 * (size_t)ix[k] + a];
            }
            st[t] = r;
        }
        const ShadeTri *dst = nullptr;
        if ((e = devUpload(ctx, &dst, st.data(), st.size()))) return e;
        HIPCHK(hipStreamSynchronize(ctx->stream));   // (the staging vector dies here)
        sv.shadeTris = dst;
    }
    if ((e = devUpload(ctx, &sv.quadrics, d->quadrics, (size_t)d->n_quadrics))) return e;
    if ((e = devUpload(ctx, &sv.sobolMatrices, d->sobol_matrices, d->sobol_matrices ? (size_t)1024 * 52 : (size_t)0)) ||
        (e = devUpload(ctx, &sv.vdcSobol, d->vdc_sobol, d->vdc_sobol ? (size_t)25 * 52 : (size_t)0)) ||
        (e = devUpload(ctx, &sv.vdcSobolInv, d->vdc_sobol_inv, d->vdc_sobol_inv ? (size_t)26 * 52 : (size_t)0)))
        return e;
    if (d->sampler.type == WF_SAMPLER_SOBOL && (!d->sobol_matrices || !d->vdc_sobol || !d->vdc_sobol_inv)) return fail(-1, "the Sobol sampler needs the sobol_matrices / vdc_sobol tables");
    if ((e = devUpload(ctx, &sv.haltonPrimes, d->halton_primes, d->halton_primes ? (size_t)1000 : (size_t)0))) return e;
    if ((e = devUpload(ctx, &sv.haltonPermOffsets, d->halton_perm_offsets, d->halton_perm_offsets ? (size_t)1000 : (size_t)0))) return e;
    if ((e = devUpload(ctx, &sv.haltonPerms, d->halton_perms, (size_t)d->n_halton_perms))) return e;
    sv.nQuadrics = d->n_quadrics;
    if ((e = devUpload(ctx, &sv.meshes, d->meshes, (size_t)d->n_meshes))) return e;
    if ((e = devUpload(ctx, &sv.bvhNodes, d->bvh_nodes, (size_t)d->n_bvh_nodes))) return e;
    if ((e = devUpload(ctx, &sv.bvhPrims, d->bvh_prims, (size_t)d->n_triangles + d->n_quadrics + d->n_instances))) return e;
    if ((e = devUpload(ctx, &sv.instances, d->instances, (size_t)d->n_instances))) return e;
    if ((e = devUpload(ctx, &sv.instanceDefs, d->instance_defs, (size_t)d->n_instance_defs))) return e;
    sv.nInstances = d->n_instances;
    if ((e = devUpload(ctx, &sv.animated, d->animated, (size_t)d->n_animated))) return e;
    sv.haveAnimated = d->n_animated > 0;
    sv.nTriangles = d->n_triangles;
    sv.nBvhNodes = d->n_bvh_nodes;
    if ((e = devUpload(ctx, &sv.spectra, d->spectra, (size_t)d->n_spectra))) return e;
    if ((e = devUpload(ctx, &sv.spectrumData, d->spectrum_data, (size_t)d->n_spectrum_floats))) return e;
    if ((e = devUpload(ctx, &sv.textures, d->textures, (size_t)d->n_textures))) return e;
    if ((e = devUpload(ctx, &sv.materials, d->materials, (size_t)d->n_materials))) return e;
    if ((e = devUpload(ctx, &sv.lights, d->lights, (size_t)d->n_lights))) return e;
    if ((e = devUpload(ctx, &sv.infiniteLights, d->infinite_lights, (size_t)d->n_infinite_lights))) return e;
    if ((e = devUpload(ctx, &sv.lightBvh, d->light_bvh_nodes, (size_t)d->n_light_bvh_nodes))) return e;
    {
        // the light BVH's nodes with their per-node constants expanded (wf_lights.h ExpandLightNode: the reference's own expressions, evaluated
        // once here instead of at every visit of every descent)
        std::vector<wf::LightNodeX> xs((size_t)d->n_light_bvh_nodes);
        for (int i = 0; i < d->n_light_bvh_nodes; ++i) xs[i] = wf::ExpandLightNode(d->all_light_bounds, d->light_bvh_nodes[i]);
        if ((e = devUpload(ctx, &sv.lightBvhX, xs.data(), xs.size()))) return e;
    }
    if ((e = devUpload(ctx, &sv.lightXforms, d->light_transforms, (size_t)d->n_light_transforms))) return e;
    if ((e = devUpload(ctx, &sv.powerAlias, d->power_alias, d->light_sampler == WF_LS_POWER ? (size_t)3 * d->n_lights : (size_t)0))) return e;
    if ((e = devUpload(ctx, &sv.imageLights, d->image_lights, (size_t)d->n_image_lights))) return e;
    if ((e = devUpload(ctx, &sv.texImages, d->tex_images, (size_t)d->n_tex_images))) return e;
    if ((e = devUpload(ctx, &sv.tableData, d->table_data, (size_t)d->n_table_floats))) return e;
    if ((e = devUpload(ctx, &sv.rgb2specCoeffs, d->rgb2spec_coeffs, d->rgb2spec_coeffs ? (size_t)3 * 64 * 64 * 64 * 3 : (size_t)0))) return e;
    if ((e = devUpload(ctx, &sv.rgb2specZNodes, d->rgb2spec_znodes, (size_t)64))) return e;
    if ((e = devUpload(ctx, &sv.noisePerm, d->noise_perm, d->noise_perm ? (size_t)512 : (size_t)0))) return e;
    sv.csIlluminantOffset = d->cs_illuminant_offset;
    if ((e = devUpload(ctx, &sv.media, d->media, (size_t)d->n_media))) return e;
    if ((e = devUpload(ctx, &sv.mediumData, d->medium_data, (size_t)d->n_medium_floats))) return e;
    {
        // the corner-packed copies of the GridMedium density grids (SceneView::gridCorners), built on the device from the dense grids
        // just uploaded.  WF_GRID_CORNERS=0: none (A/B); tables beyond WF_GRID_CORNERS_GB (default 32) gigabytes in all are left out
        sv.gridCorners = nullptr;
        sv.gridCornerBase = nullptr;
        const char *knob = getenv("WF_GRID_CORNERS");
        if (d->n_media > 0 && !(knob && atoi(knob) == 0)) {
            const double budget = (getenv("WF_GRID_CORNERS_GB") ? atof(getenv("WF_GRID_CORNERS_GB")) : 32.0) * 1024.0 * 1024.0 * 1024.0;
            std::vector<long long> base(d->n_media, -1);
            size_t total = 0;
            for (int m = 0; m < d->n_media; ++m) {
                const wf_medium &M = d->media[m];
                if (M.type != WF_MEDIUM_GRID || M.density_offset < 0 || M.nx < 1 || M.ny < 1 || M.nz < 1) continue;
                const size_t cells = wf::GridCornerCells(M.nx, M.ny, M.nz);   // (whole 8 x 8 x 8 bricks)
                if ((double)(total + 8 * cells) * sizeof(float) > budget) continue;
                base[m] = (long long)total;
                total += 8 * cells;
            }
            if (total > 0) {
                float *corners = nullptr;
                const long long *dbase = nullptr;
                if ((e = devAlloc(ctx, &corners, total))) return e;
                if ((e = devUpload(ctx, &dbase, base.data(), (size_t)d->n_media))) return e;
                for (int m = 0; m < d->n_media; ++m) {
                    if (base[m] < 0) continue;
                    const wf_medium &M = d->media[m];
                    const size_t cells = (size_t)(M.nx + 1) * (M.ny + 1) * (M.nz + 1);
                    const int grid = (int)std::min<size_t>((cells + BLOCK - 1) / BLOCK, (size_t)MAX_GRID * 16);
                    LAUNCH("Pack grid corners", k_pack_grid_corners, grid, sv.mediumData + M.density_offset, M.nx, M.ny, M.nz, corners + base[m]);
                }
                sv.gridCorners = corners;
                sv.gridCornerBase = dbase;
            }
        }
    }
    sv.nLights = d->n_lights;
    sv.nInfiniteLights = d->n_infinite_lights;
    sv.nLightBvhNodes = d->n_light_bvh_nodes;
    sv.lightSampler = d->light_sampler;
    for (int i = 0; i < 6; ++i) sv.allLightBounds[i] = d->all_light_bounds[i];
    sv.camera = d->camera;
    sv.film = d->film;
    sv.filter = d->filter;
    if ((e = devUpload(ctx, &sv.filterData, d->filter_data, (size_t)d->n_filter_floats))) return e;
    sv.sampler = d->sampler;
    if (d->sampler.type < WF_SAMPLER_ZSOBOL || d->sampler.type > WF_SAMPLER_SOBOL) return fail(-1, "unknown sampler type %d", d->sampler.type);
    if (d->sampler.type == WF_SAMPLER_HALTON && (!d->halton_primes || (d->sampler.randomize == WF_RAND_PERMUTE_DIGITS && (!d->halton_perm_offsets || !d->halton_perms))))
        return fail(-1, "Halton sampler without its prime / digit-permutation tables");
    static uint32_t sobol[WF_SOBOL_WORDS];
    FillSobol2D(sobol);
    if ((e = devUpload(ctx, &sv.sobol, sobol, (size_t)WF_SOBOL_WORDS))) return e;
    sv.texNeedsFootprint = 0;
    sv.haveAlpha = 0;
    for (int i = 0; i < d->n_meshes; ++i)
        if (d->meshes[i].alpha_tex >= 0) sv.haveAlpha = 1;
    for (int i = 0; i < d->n_textures; ++i)
        if (d->textures[i].type >= WF_TEX_FLOAT_IMAGE) sv.texNeedsFootprint = 1;
    for (int i = 0; i < d->n_materials; ++i)
        if (d->materials[i].displacement >= 0 || d->materials[i].normalmap >= 0) sv.texNeedsFootprint = 1;
    sv.maxDepth = d->max_depth;
    sv.regularize = d->regularize;
    sv.haveMedia = d->have_media;
    // the lean delta-tracking kernel (k_medium_sample<true>): no procedural cloud, NanoVDB, RGB grid or emissive grid in the scene (WF_MEDIUM_LEAN=0: off)
    ctx->mediumLean = d->n_media > 0 && !(getenv("WF_MEDIUM_LEAN") && atoi(getenv("WF_MEDIUM_LEAN")) == 0);
    for (int m = 0; m < d->n_media; ++m)
        if (!(d->media[m].type == WF_MEDIUM_HOMOGENEOUS || (d->media[m].type == WF_MEDIUM_GRID && !d->media[m].is_emissive))) ctx->mediumLean = false;
    sv.options = d->options;
    for (int m = 0; m < WF_MAT_NTYPES; ++m) ctx->matPresent[m] = false;
    sv.matTypeMask = 0;
    sv.haveMix = 0;
    sv.haveSubsurface = 0;
    ctx->rareLights = ctx->portalLights = false;
    // the lean shade kernels (wf_scene.h "LEAN DEVICE VARIANTS"): no quadrics / patches / curves, every texture a constant, an image map or a
    // bilerp (WF_LEAN_SHADE=0 turns them off)
    ctx->leanShade = d->n_quadrics == 0 && d->n_animated == 0 && !(getenv("WF_LEAN_SHADE") && atoi(getenv("WF_LEAN_SHADE")) == 0);
    bool simpleTextures = true;
    for (int i = 0; i < d->n_textures && simpleTextures; ++i)
        if (!wf::IsSimpleFloatTexture(d->textures[i].type) && !wf::IsSimpleSpectrumTexture(d->textures[i].type)) simpleTextures = false;
    ctx->leanShade = ctx->leanShade && simpleTextures;
    {
        // the material types met on shapes that are not triangles (through MixMaterials, whose hits join the queue of the chosen material's type)
        bool onGeneral[WF_MAT_NTYPES] = {};
        std::vector<int> todo;
        for (int i = 0; i < d->n_quadrics; ++i) {
            const int m = d->meshes[d->quadrics[i].mesh].material;
            if (m >= 0 && m < d->n_materials) todo.push_back(m);
        }
        std::vector<char> seen((size_t)std::max(d->n_materials, 1), 0);
        while (!todo.empty()) {
            const int m = todo.back();
            todo.pop_back();
            if (m < 0 || m >= d->n_materials || seen[m]) continue;
            seen[m] = 1;
            const int t = d->materials[m].type;
            if (t == WF_MAT_MIX) { todo.push_back(d->materials[m].mix[0]); todo.push_back(d->materials[m].mix[1]); }
            else if (t >= 0 && t < WF_MAT_NTYPES) onGeneral[t] = true;
        }
        const bool wanted = !(getenv("WF_LEAN_SHADE") && atoi(getenv("WF_LEAN_SHADE")) == 0) && !(getenv("WF_LEAN_PER_TYPE") && atoi(getenv("WF_LEAN_PER_TYPE")) == 0);
        for (int t = 0; t < WF_MAT_NTYPES; ++t) ctx->leanType[t] = ctx->leanShade || (wanted && simpleTextures && d->n_animated == 0 && !onGeneral[t]);
    }
    // ... and, since round 5, emitters that are not triangles (sphere / disk / cylinder / patch / curve lights: an out-of-line sampler of
    // 214 VGPRs) and emitters with an alpha texture (the texture-graph evaluator): LightSampleLi<RARE>, AreaLightL<ALPHA> (wf_lights.h)
    for (int i = 0; i < d->n_lights; ++i) {
        const wf_light &l = d->lights[i];
        if (l.type == WF_LIGHT_PORTAL_INFINITE) ctx->rareLights = ctx->portalLights = true;
        if (l.type == WF_LIGHT_DIFFUSE_AREA && (l.tri >= d->n_triangles || l.alpha_tex_plus1 != 0)) ctx->rareLights = true;
    }
    sv.haveQuadricAlpha = 0;
    for (int i = 0; i < d->n_quadrics; ++i) if (d->meshes[d->quadrics[i].mesh].alpha_tex >= 0) sv.haveQuadricAlpha = 1;
    sv.haveCurves = 0;
    for (int i = 0; i < d->n_quadrics; ++i) if (d->quadrics[i].type == WF_QUADRIC_CURVE) sv.haveCurves = 1;
    for (int i = 0; i < d->n_materials; ++i) {
        int t = d->materials[i].type;
        if (t == WF_MAT_MIX) {
            const int32_t *mx = d->materials[i].mix;
            if (mx[0] < 0 || mx[0] >= d->n_materials || mx[1] < 0 || mx[1] >= d->n_materials || mx[0] >= i || mx[1] >= i)
                return fail(-1, "mix material %d must name two earlier materials", i);
            sv.haveMix = 1;
            continue;
        }
        if (t < 0 || t >= WF_MAT_NTYPES) return fail(-1, "material %d has unknown type %d", i, t);
        ctx->matPresent[t] = true;
        sv.matTypeMask |= 1 << t;
        if (t == WF_MAT_SUBSURFACE) {
            sv.haveSubsurface = 1;
            if (d->materials[i].sss_table < 0 || (size_t)d->materials[i].sss_table + BSSRDF_TABLE_FLOATS > (size_t)d->n_table_floats)
                return fail(-1, "subsurface material %d: BSSRDF table outside table_data", i);
        }
        if (t == WF_MAT_MEASURED) {
            // every array the five interpolants point to must lie inside table_data (the kernels index them unchecked)
            const int64_t nT = d->n_table_floats, h = d->materials[i].measured_table;
            if (h < 0 || h + WF_MEASURED_HEADER_WORDS > nT) return fail(-1, "measured material %d: header outside table_data", i);
            auto word = [&](int64_t k) { int32_t v; memcpy(&v, &d->table_data[h + k], 4); return (int64_t)v; };
            static const int nParams[5] = {0, 0, 2, 2, 3};
            static const bool hasCdf[5] = {false, false, true, true, false};
            for (int k = 0; k < 5; ++k) {
                const int64_t b = 16 + 16 * k, sx = word(b), sy = word(b + 1);
                if (sx < 2 || sy < 2 || sx > (1 << 20) || sy > (1 << 20)) return fail(-1, "measured material %d: interpolant %d has size %lld x %lld", i, k, (long long)sx, (long long)sy);
                int64_t slices = 1;
                for (int p = nParams[k] - 1; p >= 0; --p) {
                    const int64_t ps = word(b + 2 + p), st = word(b + 5 + p), po = word(b + 8 + p);
                    if (ps < 1 || ps > (1 << 20) || po < 0 || po + ps > nT || st != (ps > 1 ? slices : 0)) return fail(-1, "measured material %d: interpolant %d, parameter %d invalid", i, k, p);
                    slices *= ps;
                    if (slices > nT) return fail(-1, "measured material %d: interpolant %d larger than table_data", i, k);
                }
                const int64_t dataOff = word(b + 11), margOff = word(b + 12), condOff = word(b + 13);
                if (dataOff < 0 || dataOff + slices * sx * sy > nT) return fail(-1, "measured material %d: interpolant %d data outside table_data", i, k);
                if (hasCdf[k] && (margOff < 0 || margOff + slices * sy > nT || condOff < 0 || condOff + slices * sx * sy > nT))
                    return fail(-1, "measured material %d: interpolant %d cdf outside table_data", i, k);
            }
        }
    }
    ctx->W = d->film.pixel_max[0] - d->film.pixel_min[0];
    ctx->H = d->film.pixel_max[1] - d->film.pixel_min[1];
    ctx->maxDepth = d->max_depth;
    for (int i = 0; i < 6; ++i) ctx->sceneBounds[i] = d->scene_bounds[i];
    {
        std::vector<QNode> qn;
        std::vector<LeafTri> lt;
        {
            ctx->genMode = 0;
            if (d->n_quadrics > 0) ctx->genMode = (sv.haveCurves || sv.haveQuadricAlpha) ? 3 : 2;
            int alphaGen = 0;   // what the TRIANGLES of the scene ask of the walk: 0 nothing, 1 simple alpha cut-outs, 2 texture-graph alpha
            for (int i = 0; i < d->n_meshes && alphaGen < 2; ++i)
                if (d->meshes[i].alpha_tex >= 0) {
                    const int tt = d->textures[d->meshes[i].alpha_tex].type;
                    // the inline test looks an image map up without a footprint (MIPFilterFloatZeroP): uv-mapped, not EWA-filtered
                    const wf_texture &at = d->textures[d->meshes[i].alpha_tex];
                    const bool lean = tt == WF_TEX_FLOAT_CONSTANT || (at.mapping == WF_TEXMAP_UV && (tt != WF_TEX_FLOAT_IMAGE || d->tex_images[at.i0].filter != WF_MIP_EWA));
                    alphaGen = std::max(alphaGen, ((tt == WF_TEX_FLOAT_CONSTANT || tt == WF_TEX_FLOAT_IMAGE || tt == WF_TEX_FLOAT_BILERP) && lean) ? 1 : 2);
                }
            ctx->genMode = std::max(ctx->genMode, alphaGen);
            if (getenv("WF_GEN_MODE")) ctx->genMode = std::max(ctx->genMode, atoi(getenv("WF_GEN_MODE")));  // timing experiments: force the general variant
            // TWO-CLASS TRAVERSAL: the scene's quadrics / patches / curves are few beside its triangles, and the triangles themselves need no
            // more than the simple alpha test — the triangle kernels walk first, the general kernels only the rays handed over
            // (WF_DEFER_GENERAL=1 | 0 forces / forbids it for any scene with such shapes)
            ctx->genTri = std::min(alphaGen, 1);
            ctx->deferGeneral = false;
            if (d->n_quadrics > 0 && alphaGen <= 1 && ctx->genMode >= 2 && wf_ctx::splitRouteWanted()) {
                const bool few = (int64_t)d->n_quadrics * 16 <= (int64_t)d->n_triangles;
                ctx->deferGeneral = getenv("WF_DEFER_GENERAL") ? atoi(getenv("WF_DEFER_GENERAL")) != 0 : few;
            }
        }
        std::vector<FastDef> fdefs;
        std::vector<SubEntry> fsubs;
        FastDepths fdep;
        ctx->fastOk = BuildFastBVH(d, &qn, &lt, &fdefs, &fsubs, &ctx->fast, &fdep);
        {
            // traversal stacks: LDS entries per lane + rows of `stackSpill` behind them, sized from the trees' ACTUAL depths: the
            // reference-order walk pushes one sibling per level of the reference's binary trees (top level, then an instance
            // definition's on top); the four-wide production walk up to three per level of ITS collapsed trees (BuildFastBVH records
            // their depths: the greedy largest-area collapse does not halve the depth of an unbalanced tree), one entry per instance
            // of a leaf, and the two instance markers.  A push past the rows is dropped and flagged (LdsStackT, wf_sync).
            auto treeDepth = [&](int root) {
                int best = 0;
                if (root < 0 || root >= d->n_bvh_nodes) return best;
                std::vector<std::pair<int, int>> st{{root, 1}};
                while (!st.empty()) {
                    auto [i, dep] = st.back();
                    st.pop_back();
                    best = std::max(best, dep);
                    if (d->bvh_nodes[i].nprims == 0) { st.push_back({i + 1, dep + 1}); st.push_back({d->bvh_nodes[i].offset, dep + 1}); }
                }
                return best;
            };
            int depthTop = d->n_bvh_nodes > 0 ? treeDepth(0) : 0, depthDef = 0;
            for (int k = 0; k < d->n_instance_defs; ++k) depthDef = std::max(depthDef, treeDepth(d->instance_defs[k].bvh_root));
            const int needRef = depthTop + depthDef + 4;
            const int needFast = 3 * fdep.top + fdep.maxLeafInstances + 3 * fdep.def + 6;
            const int rows = std::max(std::max(needRef - std::min(STACK_LDS, TSTACK), needFast - TSTACK), STACK_MAX - std::min(STACK_LDS, TSTACK));
            if (rows > 2048) return fail(-1, "BVH too deep for the traversal stacks (depth %d + %d)", depthTop, depthDef);
            if ((e = devAlloc(ctx, &ctx->stackSpill, (size_t)rows * MAX_GRID * BLOCK))) return e;
            if ((e = devAlloc(ctx, &ctx->walkSave, (size_t)4 * MAX_GRID * BLOCK))) return e;
            ctx->spillRows = rows;
            if ((e = devAlloc(ctx, &ctx->dbgWords, (size_t)8 + 32))) return e;   // [8..39]: -DWF_WALK_STATS builds: sixteen 64-bit phase counters (WalkStats)   // [4..7]: near-tie queue diagnostics (pushes, re-walks without a hit, -, -)
            HIPCHK(hipMemset(ctx->dbgWords, 0, (8 + 32) * sizeof(int)));
        }
        if (d->n_bvh_nodes > 0)
            for (int a = 0; a < 3; ++a) { ctx->sceneMin[a] = d->bvh_nodes[0].bmin[a]; ctx->sceneMax[a] = d->bvh_nodes[0].bmax[a]; }
        if (ctx->fastOk) {
            if ((e = devUpload(ctx, &ctx->fast.nodes, qn.data(), qn.size()))) return e;
            if ((e = devUpload(ctx, &ctx->fast.tris, lt.data(), lt.size()))) return e;
            if ((e = devUpload(ctx, &ctx->fast.defs, fdefs.data(), fdefs.size()))) return e;
            if ((e = devUpload(ctx, &ctx->fast.subs, fsubs.data(), fsubs.size()))) return e;
            // rays of a scene whose trees do not fit the caches walk long enough for one cursor fetch per 64 rays (measured: -3 % on
            // the 10 M-triangle scene); a cache-resident scene traces so fast that the cursor's atomics would bound it (see cursorChunk)
            if (!getenv("WF_CURSOR_CHUNK")) {
                const bool big = qn.size() * sizeof(QNode) + lt.size() * sizeof(LeafTri) > ((size_t)256 << 20);
                ctx->cursorChunk = big ? 3 : 2;
                ctx->cursorChunkShadow = big ? 1 : 2;
            }
            ctx->fast.instances = ctx->svHost.instances;
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
        // resident workgroups of the traversal kernel variants this scene launches (closest-hit and shadow differ in registers)
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
        const int maxG = MAX_GRID * BLOCK / TBLOCK;  // stackSpill is sized for MAX_GRID * BLOCK threads
        auto residentGrid = [&](const void *kernel, int *out) {
            int perCU = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, TBLOCK, 0));
            int g = std::max(1, perCU) * prop.multiProcessorCount;
            if (const char *m = getenv("WF_PGRID_MULT")) g = (int)(g * atof(m));
            *out = g > maxG ? maxG : (g < 1 ? 1 : g);
            return 0;
        };
        {
            const int gen = ctx->genMode;
            const bool inst = ctx->svHost.nInstances > 0, split = ctx->splitRouteWanted();
            const void *kc, *ks;
#define WF_PICK(K, ...) (inst ? (gen == 0 ? (const void *)K<0, true __VA_ARGS__> : gen == 1 ? (const void *)K<1, true __VA_ARGS__> : gen == 2 ? (const void *)K<2, true __VA_ARGS__> : (const void *)K<3, true __VA_ARGS__>) \
                              : (gen == 0 ? (const void *)K<0, false __VA_ARGS__> : gen == 1 ? (const void *)K<1, false __VA_ARGS__> : gen == 2 ? (const void *)K<2, false __VA_ARGS__> : (const void *)K<3, false __VA_ARGS__>))
            (void)split;
            kc = WF_PICK(k_closest_fast, , true);
            ks = WF_PICK(k_shadow_fast);
#undef WF_PICK
            if ((e = residentGrid(kc, &ctx->persistentGrid)) || (e = residentGrid(ks, &ctx->persistentGridShadow))) return e;
            if (ctx->deferGeneral) {
                ctx->persistentGridGen = ctx->persistentGrid; ctx->persistentGridShadowGen = ctx->persistentGridShadow;
                const bool t1 = ctx->genTri == 1;
                kc = inst ? (t1 ? (const void *)k_closest_fast<5, true, true> : (const void *)k_closest_fast<4, true, true>) : (t1 ? (const void *)k_closest_fast<5, false, true> : (const void *)k_closest_fast<4, false, true>);
                ks = inst ? (t1 ? (const void *)k_shadow_fast<5, true> : (const void *)k_shadow_fast<4, true>) : (t1 ? (const void *)k_shadow_fast<5, false> : (const void *)k_shadow_fast<4, false>);
                if ((e = residentGrid(kc, &ctx->persistentGrid)) || (e = residentGrid(ks, &ctx->persistentGridShadow))) return e;
            }
        }
        if (getenv("WF_NO_FAST")) ctx->fastOk = false;
        // AnimatedPrimitive: the production walks' ANIM variants (triangles + simple alpha cut-outs, two-level: an animated shape entity is an
        // instance) interpolate the transformation per ray since round 6; scenes that also hold quadrics / curves / texture-graph alpha keep the
        // reference-order walks (WF_ANIM_FAST=0: every animated scene does)
        ctx->animFast = d->n_animated > 0 && ctx->fastOk && ctx->genMode <= 1 && ctx->svHost.nInstances > 0 && !(getenv("WF_ANIM_FAST") && atoi(getenv("WF_ANIM_FAST")) == 0);
        if (d->n_animated > 0 && !ctx->animFast) ctx->fastOk = false;
        if (ctx->animFast) {
            const bool g1 = ctx->genMode == 1;
            if ((e = residentGrid(g1 ? (const void *)k_closest_fast<9, true, true> : (const void *)k_closest_fast<8, true, true>, &ctx->persistentGrid)) ||
                (e = residentGrid(g1 ? (const void *)k_shadow_fast<9, true> : (const void *)k_shadow_fast<8, true>, &ctx->persistentGridShadow))) return e;
        }
        if ((e = devAlloc(ctx, &ctx->probeCursor, (size_t)1))) return e;
        {
            int perCU = 0;
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void *)k_medium_sample<false>, BLOCK, 0));
            ctx->mediumGrid = std::min(MAX_GRID, std::max(1, perCU) * prop.multiProcessorCount);
        }
    }
    if ((e = devAlloc(ctx, &ctx->ws.film, (size_t)ctx->W * ctx->H * 4))) return e;
    ctx->ws.filmSpectral = nullptr;
    ctx->ws.filmGBuffer = nullptr;
    if (d->film.type == WF_FILM_GBUFFER && (e = devAlloc(ctx, &ctx->ws.filmGBuffer, (size_t)ctx->W * ctx->H))) return e;
    if (d->film.type == WF_FILM_SPECTRAL) {
        if (d->film.n_buckets < 1 || d->film.n_buckets > 4096 || !(d->film.lambda_max > d->film.lambda_min)) return fail(-1, "spectral film: bad bucket count / wavelength range");
        if ((e = devAlloc(ctx, &ctx->ws.filmSpectral, (size_t)ctx->W * ctx->H * 2 * d->film.n_buckets))) return e;
    }
    if ((e = devAlloc(ctx, &ctx->ws.stats, (size_t)(129 + 16)))) return e;   // [129 ..]: items per material queue / the medium-sample queue
    if ((e = devAlloc(ctx, &ctx->ws.trav, (size_t)8))) return e;
    if ((e = devAlloc(ctx, &ctx->ws.counters, (size_t)CNT_COUNT * CNT_STRIDE))) return e;
    {
        SceneView *dev = nullptr;
        if ((e = devAlloc(ctx, &dev, (size_t)1))) return e;
        ctx->svHost.self = dev;  // the device copy points at itself: the address the out-of-line device functions are given
        ctx->svHost.fatal = ctx->dbgWords ? ctx->dbgWords + 7 : nullptr;   // RaiseFatal (wf_scene.h): reported by wf_sync
        HIPCHK(hipMemcpyAsync(dev, &ctx->svHost, sizeof(SceneView), hipMemcpyHostToDevice, ctx->stream));
        ctx->svDev = dev;
    }
    ctx->fast.sv = ctx->svDev;
    HIPCHK(hipStreamSynchronize(ctx->stream));  // sv / sobol live on the host stack
    ctx->sceneLoaded = true;
    return 0;
}

int wf_ctx_query(wf_ctx *ctx, const char *key, int64_t *value) {
    if (!ctx || !key || !value) return fail(-1, "wf_ctx_query: null argument");
    if (!ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    const std::string k = key;
    if (k == "fast_ok") *value = ctx->fastOk;
    else if (k == "gen_mode") *value = ctx->genMode;
    else if (k == "gen_tri") *value = ctx->genTri;
    else if (k == "defer_general") *value = ctx->deferGeneral;
    else if (k == "anim_fast") *value = ctx->animFast;
    else if (k == "lean_shade") *value = ctx->leanShade;
    else if (k == "rare_lights") *value = ctx->rareLights;
    else if (k == "medium_lean") *value = ctx->mediumLean;   // k_medium_sample<true> / k_tr_segment<true>: every medium is homogeneous or a non-emissive uniform grid
    else if (k.rfind("lean_type_", 0) == 0 && atoi(key + 10) >= 0 && atoi(key + 10) < WF_MAT_NTYPES) *value = ctx->leanType[atoi(key + 10)];
    else if (k == "instances") *value = ctx->svHost.nInstances;
    else return fail(-1, "wf_ctx_query: unknown key '%s'", key);
    return 0;
}

int wf_aggregate_bounds(wf_ctx *ctx, float out_bounds[6]) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    for (int i = 0; i < 6; ++i) out_bounds[i] = ctx->sceneBounds[i];
    return 0;
}

static int allocRayQueue(wf_ctx *c, RayQueueV *q, size_t n) {
    int e;
    if ((e = devAlloc(c, &q->o, n)) || (e = devAlloc(c, &q->d, n)) || (e = devAlloc(c, &q->beta, n)) || (e = devAlloc(c, &q->r_u, n)) ||
        (e = devAlloc(c, &q->r_l, n)) || (e = devAlloc(c, &q->ctx0, n)) || (e = devAlloc(c, &q->ctx1, n)) || (e = devAlloc(c, &q->ctx2, n)) ||
        (e = devAlloc(c, &q->meta, n)))
        return e;
    return 0;
}

int wf_queues_alloc(wf_ctx *ctx, int pixels_per_pass, int samples_per_pass) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    if (ctx->queuesAllocated) return fail(-1, "queues already allocated");
    if (pixels_per_pass <= 0 || samples_per_pass <= 0) return fail(-1, "pixels_per_pass and samples_per_pass must be positive");
    if ((long long)pixels_per_pass * samples_per_pass > (1ll << 30)) return fail(-1, "queue capacity %lld too large", (long long)pixels_per_pass * samples_per_pass);
    if (pixels_per_pass % ctx->W != 0) return fail(-1, "pixels_per_pass must be a whole number of scanlines (width %d)", ctx->W);
    HIPCHK(hipSetDevice(ctx->device));
    const int max_queue_size = pixels_per_pass * samples_per_pass;
    const size_t n = (size_t)max_queue_size;
    WorkState &ws = ctx->ws;
    ws.maxQueueSize = max_queue_size;
    ws.pixelsPerPass = pixels_per_pass;
    ws.samplesPerPass = samples_per_pass;
    if (ws.stripCount < 1) { ws.stripRank = 0; ws.stripCount = 1; ws.stripHeight = 1; ws.localRows = ctx->H; }
    ctx->passSamples = 1;
    if (const char *e = getenv("WF_PIXEL_MAJOR")) ctx->pixelMajor = atoi(e) != 0;
    ctx->ws.slotStride = ctx->pixelMajor ? 1 : 0;
    ctx->passStep = 1;
    int e;
    if ((e = devAlloc(ctx, &ws.filterWeight, n)) || (e = devAlloc(ctx, &ws.pPixel, n)) || (e = devAlloc(ctx, &ws.lambda, n)) ||
        (e = devAlloc(ctx, &ws.lambdaPdf, n)) || (e = devAlloc(ctx, &ws.L, n)) || (e = devAlloc(ctx, &ws.cameraRayWeight, n)) ||
        (e = devAlloc(ctx, &ws.samples0, n)) || (e = devAlloc(ctx, &ws.samples1, n)))
        return e;
    if (ctx->svHost.film.type == WF_FILM_GBUFFER &&
        ((e = devAlloc(ctx, &ws.vsP, n)) || (e = devAlloc(ctx, &ws.vsN, n)) || (e = devAlloc(ctx, &ws.vsNs, n)) || (e = devAlloc(ctx, &ws.vsDpdx, n)) ||
         (e = devAlloc(ctx, &ws.vsDpdy, n)) || (e = devAlloc(ctx, &ws.vsAlbedo, n))))
        return e;
    // pixel-only sample-index digits (wf_camera.h TopDigits): usable while the permuted index fits 32 bits
    ws.sampleTops = nullptr;
    if (ctx->svHost.sampler.type == WF_SAMPLER_ZSOBOL && 2 * ctx->svHost.sampler.nBase4Digits <= 32 && !getenv("WF_NO_SAMPLE_TOPS") &&
        !ctx->svHost.haveSubsurface)  // (K12 scenes draw 10 dimensions per depth, not 7)
        if ((e = devAlloc(ctx, &ws.sampleTops, (size_t)5 * pixels_per_pass))) return e;
    if ((e = allocRayQueue(ctx, &ws.rq[0], n)) || (e = allocRayQueue(ctx, &ws.rq[1], n))) return e;
    if (ctx->svHost.haveMedia) {
        if (getenv("WF_TR_WAVEFRONT")) ctx->trWavefront = atoi(getenv("WF_TR_WAVEFRONT"));
        if ((e = devAlloc(ctx, &ws.trO, n)) || (e = devAlloc(ctx, &ws.trD, n)) || (e = devAlloc(ctx, &ws.trT, n)) || (e = devAlloc(ctx, &ws.trRu, n)) ||
            (e = devAlloc(ctx, &ws.trRl, n)) || (e = devAlloc(ctx, &ws.trRng, n)) || (e = devAlloc(ctx, &ws.trQ[0], n)) || (e = devAlloc(ctx, &ws.trQ[1], n)))
            return e;
        if ((e = devAlloc(ctx, &ws.hitT, n)) || (e = devAlloc(ctx, &ws.mediumSampleQ, n)) || (e = devAlloc(ctx, &ws.mediumScatterQ, n)) || (e = devAlloc(ctx, &ws.mediumRouteQ, n)) ||
            (e = devAlloc(ctx, &ws.scatterP, n)) || (e = devAlloc(ctx, &ws.sq.medium, n)))
            return e;
    }
    if (ctx->svHost.haveMix && ((e = devAlloc(ctx, &ws.mixMat, n)) || (e = devAlloc(ctx, &ws.mixQ, n)))) return e;
    if (ctx->svHost.haveSubsurface && ((e = devAlloc(ctx, &ws.samples2, n)) || (e = devAlloc(ctx, &ws.bssrdfQ, n)) || (e = devAlloc(ctx, &ws.sssQ, n)))) return e;
    if ((e = devAlloc(ctx, &ws.hit, n)) || (e = devAlloc(ctx, &ws.escapedQ, n)) || (e = devAlloc(ctx, &ws.hitLightQ, n)) || (e = devAlloc(ctx, &ws.retraceQ, n)) || (e = devAlloc(ctx, &ws.retraceQ64, n))) return e;
    if ((e = devAlloc(ctx, &ws.deferQ, n))) return e;
    HIPCHK(hipMemset(ws.retraceQ64, 0xff, (size_t)n * sizeof(unsigned long long)));   // every slot "not yet written"
    if (ctx->svHost.nInstances > 0 && (e = devAlloc(ctx, &ws.hitInst, n))) return e;
    for (int m = 0; m < WF_MAT_NTYPES; ++m)
        if ((e = devAlloc(ctx, &ws.matQ[m], ctx->matPresent[m] ? n : (size_t)1))) return e;  // workqueue.h:152-155
    if ((e = devAlloc(ctx, &ws.sq.o, n)) || (e = devAlloc(ctx, &ws.sq.d, n)) || (e = devAlloc(ctx, &ws.sq.Ld, n)) ||
        (e = devAlloc(ctx, &ws.sq.r_u, n)) || (e = devAlloc(ctx, &ws.sq.r_l, n)))
        return e;
#if defined(WF_HAVE_FUSED_MAT)
#if defined(WF_MAT_SPLIT_DEFAULT)
    ctx->matSplit = WF_MAT_SPLIT_DEFAULT != 0;
#endif
    if (const char *sp = getenv("WF_MAT_SPLIT")) ctx->matSplit = atoi(sp) != 0;
#endif
    if (ctx->svHost.haveAnimated && (e = devAlloc(ctx, &ws.pathTime, n))) return e;   // the paths' times, for the shadow rays (AnimatedPrimitive)
    if (ctx->matSplit) {
        // the NeeItems between the material stage's two kernels: as many 16-byte planes as the widest record of the material types present
        int planes = 0;
        for (int m = 1; m < WF_MAT_NTYPES; ++m) if (ctx->matPresent[m]) planes = std::max(planes, NeePlanes(m));
        if (planes > 0 && (e = devAlloc(ctx, &ws.neeRec, n * (size_t)planes))) return e;
    }
    // (round 6: WF_SPLIT_ROUTE=0 — the walk routing its hits per workgroup, the round-2 path — is gone: its kernel variants were the ones that
    //  kept tripping the spill-carrier lint whenever anything near them changed; 1 = no work cursor, 2 = the default)
    ctx->splitRoute = getenv("WF_SPLIT_ROUTE") ? std::max(1, atoi(getenv("WF_SPLIT_ROUTE"))) : 2;
    if (getenv("WF_CURSOR_CHUNK")) ctx->cursorChunk = ctx->cursorChunkShadow = std::max(1, atoi(getenv("WF_CURSOR_CHUNK")));  // (default: chosen at scene upload)
    if (ctx->splitRoute && (e = devAlloc(ctx, &ws.routeCode, n))) return e;
    ctx->raySort = getenv("WF_RAY_SORT") ? atoi(getenv("WF_RAY_SORT")) : 0;
    if (!ctx->fastOk) ctx->raySort = 0;
    if (ctx->raySort) {
        if (getenv("WF_SORT_MIN")) ctx->sortMin = atoi(getenv("WF_SORT_MIN"));
        if (getenv("WF_SORT_OBITS")) ctx->sortOriginBits = std::min(std::max(atoi(getenv("WF_SORT_OBITS")), 1), 10);
        if (getenv("WF_SORT_DBITS")) ctx->sortDirBits = std::min(std::max(atoi(getenv("WF_SORT_DBITS")), 0), (32 - 3 * ctx->sortOriginBits) / 2);
        for (int k = 0; k < 2; ++k)
            if ((e = devAlloc(ctx, &ctx->sortKeys[k], n)) || (e = devAlloc(ctx, &ctx->sortVals[k], n))) return e;
        size_t bytes = 0;
        if (wf_sort_pairs_u32(ctx->stream, nullptr, &bytes, ctx->sortKeys[0], ctx->sortKeys[1], ctx->sortVals[0], ctx->sortVals[1], (unsigned)n, 32u) != 0)
            return fail(-1, "ray sort: scratch size query failed");
        unsigned char *tmp = nullptr;
        if ((e = devAlloc(ctx, &tmp, bytes))) return e;
        ctx->sortTemp = tmp; ctx->sortTempBytes = bytes;
        if ((ctx->raySort & 1) && (e = allocRayQueue(ctx, &ctx->rqTmp, n))) return e;
        if (ctx->raySort & 2) {
            if ((e = devAlloc(ctx, &ctx->sqTmp.o, n)) || (e = devAlloc(ctx, &ctx->sqTmp.d, n)) || (e = devAlloc(ctx, &ctx->sqTmp.Ld, n)) ||
                (e = devAlloc(ctx, &ctx->sqTmp.r_u, n)) || (e = devAlloc(ctx, &ctx->sqTmp.r_l, n)))
                return e;
            if (ctx->svHost.haveMedia && (e = devAlloc(ctx, &ctx->sqTmp.medium, n))) return e;
        }
        const float cells = (float)(1 << ctx->sortOriginBits);
        for (int a = 0; a < 3; ++a) {
            float ext = ctx->sceneMax[a] - ctx->sceneMin[a];
            ctx->sortBase[a] = ctx->sceneMin[a];
            ctx->sortScale[a] = ext > 0 ? cells / ext : 0.f;
        }
    }
    ctx->maxQueueSize = max_queue_size;
    ctx->queuesAllocated = true;
    return 0;
}

// sample indices carried by the following passes: sample_index, sample_index + sample_step, ... (n_samples of
// them, n_samples <= samples_per_pass of wf_queues_alloc)
int wf_set_pass_samples(wf_ctx *ctx, int sample_step, int n_samples) {
    if (int e = checkReady(ctx)) return e;
    if (n_samples < 1 || n_samples > ctx->ws.samplesPerPass) return fail(-1, "n_samples %d outside 1..%d", n_samples, ctx->ws.samplesPerPass);
    if (sample_step < 1) return fail(-1, "sample_step must be >= 1");
    ctx->passStep = sample_step;
    ctx->passSamples = n_samples;
    ctx->ws.slotStride = ctx->pixelMajor ? n_samples : 0;
    return 0;
}

int wf_set_strips(wf_ctx *ctx, int rank, int count, int height, int *local_rows) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    if (count < 1 || rank < 0 || rank >= count || height < 1) return fail(-1, "wf_set_strips: rank %d of %d, height %d", rank, count, height);
    int rows = 0;
    for (int y = 0; y < ctx->H; ++y) rows += (y / height) % count == rank;
    ctx->ws.stripRank = rank; ctx->ws.stripCount = count; ctx->ws.stripHeight = height; ctx->ws.localRows = count > 1 ? rows : ctx->H;
    if (local_rows) *local_rows = ctx->ws.localRows;
    return 0;
}

int wf_film_clear(wf_ctx *ctx) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    HIPCHK(hipMemsetAsync(ctx->ws.film, 0, (size_t)ctx->W * ctx->H * 4 * sizeof(double), ctx->stream));
    if (ctx->ws.filmGBuffer) HIPCHK(hipMemsetAsync(ctx->ws.filmGBuffer, 0, (size_t)ctx->W * ctx->H * sizeof(wf_gbuffer_pixel), ctx->stream));
    if (ctx->ws.filmSpectral) HIPCHK(hipMemsetAsync(ctx->ws.filmSpectral, 0, (size_t)ctx->W * ctx->H * 2 * ctx->svHost.film.n_buckets * sizeof(double), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->ws.stats, 0, (129 + 16) * sizeof(unsigned long long), ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->ws.trav, 0, 8 * sizeof(unsigned long long), ctx->stream));
    return 0;
}

int wf_reset_ray_queue(wf_ctx *ctx, int which) {
    if (int e = checkReady(ctx)) return e;
    LAUNCH("Reset ray queue", k_reset, 1, ctx->ws, 1u << (CNT_RAY0 + (which & 1)), -1, 0);
    return 0;
}
int wf_reset_stage_queues(wf_ctx *ctx, int depth) {
    if (int e = checkReady(ctx)) return e;
    const int cur = depth & 1;
    unsigned mask = (1u << (CNT_RAY0 + (cur ^ 1))) | (1u << CNT_ESCAPED) | (1u << CNT_HITLIGHT);
    for (int m = 0; m < WF_MAT_NTYPES; ++m) mask |= 1u << (CNT_MAT0 + m);
    mask |= (1u << CNT_MEDIUM_SAMPLE) | (1u << CNT_MEDIUM_SCATTER) | (1u << CNT_MIX) | (1u << CNT_RETRACE) | (1u << CNT_BSSRDF) | (1u << CNT_SSS);
    mask |= (1u << CNT_RETRACE_HEAD) | (1u << CNT_WAVES_DONE) | (1u << CNT_CURSOR) | (1u << CNT_MEDIUM_ROUTE) | (1u << CNT_DEFER) | (1u << CNT_CURSOR_MEDIUM);
    // stats->indirectRays[depth] += queue size (integrator.cpp:411-414)
    LAUNCH("Reset queues before tracing rays", k_reset, 1, ctx->ws, mask, 1 + statDepth(depth), CNT_RAY0 + cur);
    ctx->cursorDirty[0] = false;
    ctx->cursorDirty[2] = false;
    return 0;
}
int wf_gen_camera_rays(wf_ctx *ctx, int y0, int sample_index) {
    if (int e = checkReady(ctx)) return e;
    ctx->passY0 = y0;
    if (ctx->ws.sampleTops) LAUNCH("Sampler index prefixes", k_sample_tops, gridFor(5 * ctx->ws.pixelsPerPass), ctx->svHost, ctx->ws, y0, 0);
    if (ctx->svHost.camera.type == WF_CAMERA_REALISTIC)  // rays blocked by the lenses leave no queue entry: the kernel appends
        LAUNCH("Reset ray queue", k_reset, 1, ctx->ws, 1u << CNT_RAY0, -1, 0);
    if (ctx->svHost.camera.anim.actually_animated)
        LAUNCH("Generate camera rays", k_gen_camera_rays<true>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, y0, sample_index, ctx->passStep, ctx->passSamples);
    else
        LAUNCH("Generate camera rays", k_gen_camera_rays<false>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, y0, sample_index, ctx->passStep, ctx->passSamples);
    LAUNCH("Update camera ray stats", k_reset, 1, ctx->ws, 0u, 0, CNT_RAY0);
    return 0;
}
int wf_gen_ray_samples(wf_ctx *ctx, int depth, int sample_index) {
    if (int e = checkReady(ctx)) return e;
    // the pixel-only digits of the sample-index permutation, once per pixel and dimension instead of once per ray
    // (needs the band of the pass: the y0 of the last wf_gen_camera_rays)
    const bool tops = ctx->ws.sampleTops != nullptr && ctx->passY0 != INT_MIN;
    if (tops) LAUNCH("Sampler index prefixes", k_sample_tops, gridFor(5 * ctx->ws.pixelsPerPass), ctx->svHost, ctx->ws, ctx->passY0, 6 + 7 * depth);
    LAUNCH("Generate ray samples - ZSobolSampler", k_gen_ray_samples, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1, sample_index, ctx->passStep,
           tops ? depth : -1);
    return 0;
}
// ray-coherence pass: sort + physically permute the ray queue `cur` (shadow = false) or the shadow queue (see k_ray_sort_keys)
static int SortQueue(wf_ctx *ctx, bool shadow, int cur) {
    WorkState &ws = ctx->ws;
    int n = 0;
    HIPCHK(hipMemcpyAsync(&n, &ws.counters[(shadow ? CNT_SHADOW : CNT_RAY0 + cur) * CNT_STRIDE], sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (n < ctx->sortMin) return 0;  // (a launch this small is latency, not throughput)
    SortGrid g;
    for (int a = 0; a < 3; ++a) { g.base[a] = ctx->sortBase[a]; g.scale[a] = ctx->sortScale[a]; }
    g.obits = ctx->sortOriginBits; g.dbits = ctx->sortDirBits;
    const F4 *o = shadow ? ws.sq.o : ws.rq[cur].o, *d = shadow ? ws.sq.d : ws.rq[cur].d;
    {
        Prof prof_(ctx, shadow ? "Sort shadow rays" : "Sort rays");
        hipLaunchKernelGGL(k_ray_sort_keys, dim3(gridFor(n)), dim3(BLOCK), 0, ctx->stream, o, d, n, g, ctx->sortKeys[0], ctx->sortVals[0]);
        size_t bytes = ctx->sortTempBytes;
        if (wf_sort_pairs_u32(ctx->stream, ctx->sortTemp, &bytes, ctx->sortKeys[0], ctx->sortKeys[1], ctx->sortVals[0], ctx->sortVals[1], (unsigned)n,
                              (unsigned)(3 * g.obits + 2 * g.dbits)) != 0)
            return fail(-1, "ray sort failed");
        if (shadow) {
            hipLaunchKernelGGL(k_permute_shadow, dim3(gridFor(n)), dim3(BLOCK), 0, ctx->stream, ws.sq, ctx->sqTmp, ctx->sortVals[1], n);
            std::swap(ws.sq, ctx->sqTmp);
        } else {
            hipLaunchKernelGGL(k_permute_rays, dim3(gridFor(n)), dim3(BLOCK), 0, ctx->stream, ws.rq[cur], ctx->rqTmp, ctx->sortVals[1], n);
            std::swap(ws.rq[cur], ctx->rqTmp);
        }
    }
    return 0;
}
static int JoinRetrace(wf_ctx *ctx) {
    if (!ctx->retracePending) return 0;
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->evJoin, 0));
    ctx->retracePending = false;
    return 0;
}
int wf_intersect_closest(wf_ctx *ctx, int depth) {
    if (int e = checkReady(ctx)) return e;
    // counting on: the reference-order walk (its visit counts define the algorithmic bytes);
    // otherwise the production traversal (wf_traverse.h)
    if (ctx->countTraversal)
        { if (ctx->svHost.haveAnimated) LAUNCH("Intersect closest", (k_intersect_closest<true, true>), gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1, ctx->stackSpill);
          else LAUNCH("Intersect closest", k_intersect_closest<true>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1, ctx->stackSpill); }
    else if (ctx->fastOk) {
        if ((ctx->raySort & 1) && depth >= 1)
            if (int e = SortQueue(ctx, false, depth & 1)) return e;
        if (ctx->splitRoute) {
            int *cursor = nullptr;
            if (ctx->splitRoute > 1) {
                // the work cursor starts at 0: zeroed by the stage's "Reset queues" launch (wf_reset_stage_queues) — a memset node of its own
                // cost a launch's latency per traversal launch (139 fillBuffer launches, 6 ms, in three renders of the spec scene, round 4);
                // only a caller that launches twice without that reset in between pays for one here
                cursor = ctx->ws.counters + CNT_CURSOR * CNT_STRIDE;
                if (ctx->cursorDirty[0]) HIPCHK(hipMemsetAsync(cursor, 0, sizeof(int), ctx->stream));
                ctx->cursorDirty[0] = true;
            }
            ctx->ws.drainEpoch = (ctx->ws.drainEpoch + 1) & 0x7fffffff;   // tag of this launch's near-tie queue entries (DrainRetrace)
            if (ctx->ws.drainEpoch == 0) ctx->ws.drainEpoch = 1;
            if (ctx->deferGeneral) {
                // TWO-CLASS TRAVERSAL: every ray through the triangle kernel; the rays it hands over (deferQ) through the general kernel
                // (the triangle kernels know one near-tie band: the triangles' own, 2^-20 — the scene-wide band of FastBVH is the wide one of
                //  the pairs that involve a quadric, which only the general kernel's rays can meet)
                FastBVH triFast = ctx->fast;
                triFast.absBand = triFast.absBandTri; triFast.tieRel = triFast.tieRelTri;
                LAUNCHT_CLOSEST_SPLIT_GEN("Intersect closest", 4 + ctx->genTri, ctx->persistentGrid, ctx->svHost, ctx->ws, triFast, depth & 1, ctx->spillArea(), cursor, ctx->cursorChunk, (const int *)nullptr);
                LAUNCHT_CLOSEST_SPLIT_GEN("Intersect closest: rays that met a general primitive", ctx->genMode, ctx->persistentGridGen, ctx->svHost, ctx->ws, ctx->fast, depth & 1, ctx->spillArea(), (int *)nullptr, ctx->cursorChunk, (const int *)ctx->ws.deferQ);
            } else
            LAUNCHT_CLOSEST_SPLIT_GEN("Intersect closest", ctx->animFast ? 8 + ctx->genMode : ctx->genMode, ctx->persistentGrid, ctx->svHost, ctx->ws, ctx->fast, depth & 1, ctx->spillArea(), cursor, ctx->cursorChunk, (const int *)nullptr);
            // The near-tie re-trace of the scenes whose walk does not resolve its ties itself (genMode >= 2: quadrics, curves, texture-graph
            // alpha; RetraceInline) is a handful of long single walks: it runs on a second stream beside the routing pass (and, in the
            // fused pass, the next sample-generation launch) — they touch disjoint rays and share only the queue counters, through
            // atomics — and the main stream waits for it before anything consumes the queues.
            const bool overlap = !RetraceInline(ctx->genMode) && ctx->overlapRetrace && ctx->profile != 1 && !ctx->traceLaunch;   // (the full per-stage profile times every launch on the main stream)
            if (overlap) {
                HIPCHK(hipEventRecord(ctx->evFork, ctx->stream));
                HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->evFork, 0));
                hipLaunchKernelGGL(k_closest_retrace, dim3(128), dim3(BLOCK), 0, ctx->stream2, ctx->svHost, ctx->ws, depth & 1, ctx->stackSpill);   // (nothing that runs beside it walks a tree: the spill rows are its own)
                HIPCHK(hipEventRecord(ctx->evJoin, ctx->stream2));
                ctx->retracePending = true;
            }
            {
                Prof prof_(ctx, "Route hits");
                const int g = std::min(MAX_GRID, std::max(1, (ctx->maxQueueSize + RBLOCK - 1) / RBLOCK));
                if (ctx->genMode > 1 || ctx->svHost.nInstances > 0) hipLaunchKernelGGL(k_route_hits<true>, dim3(g), dim3(RBLOCK), 0, ctx->stream, ctx->svHost, ctx->ws, depth & 1);
                else hipLaunchKernelGGL(k_route_hits<false>, dim3(g), dim3(RBLOCK), 0, ctx->stream, ctx->svHost, ctx->ws, depth & 1);
            }
        }
        if (ctx->retracePending) {
            if (ctx->deferJoin) return 0;   // the fused pass joins after its sample-generation launch (JoinRetrace)
            if (int e = JoinRetrace(ctx)) return e;
        } else if (!RetraceInline(ctx->genMode))
        LAUNCH("Intersect closest: near-tie re-trace", k_closest_retrace, 128, ctx->svHost, ctx->ws, depth & 1, ctx->stackSpill);
        if (ctx->svHost.haveMix) LAUNCH("Resolve MixMaterial hits", k_resolve_mix, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1);
    } else
        { if (ctx->svHost.haveAnimated) LAUNCH("Intersect closest", (k_intersect_closest<false, true>), gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1, ctx->stackSpill);
          else LAUNCH("Intersect closest", k_intersect_closest<false>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1, ctx->stackSpill); }
    return 0;
}
// SampleMediumInteraction (wavefront/media.cpp:22-257): K5, then K6 for the Henyey-Greenstein phase function
int wf_medium_sample(wf_ctx *ctx, int depth) {
    if (int e = checkReady(ctx)) return e;
    if (!ctx->svHost.haveMedia) return 0;
    {
        // persistent launch with wave-level refill: one resident grid, the items dealt from a work cursor (zeroed by the stage's reset launch)
        int *cursor = ctx->ws.counters + CNT_CURSOR_MEDIUM * CNT_STRIDE;
        if (ctx->cursorDirty[2]) HIPCHK(hipMemsetAsync(cursor, 0, sizeof(int), ctx->stream));
        ctx->cursorDirty[2] = true;
        const int grid = WF_MEDIUM_REFILL ? std::min(ctx->mediumGrid, gridFor(ctx->maxQueueSize)) : gridFor(ctx->maxQueueSize);
        if (ctx->mediumLean) LAUNCH("Sample medium interaction", k_medium_sample<true>, grid, ctx->svHost, ctx->ws, depth & 1, cursor);
        else LAUNCH("Sample medium interaction", k_medium_sample<false>, grid, ctx->svHost, ctx->ws, depth & 1, cursor);
    }
    LAUNCH("Sample medium interaction: route surface hits", k_medium_route, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1);
    if (depth == ctx->maxDepth) return 0;
    if (ctx->rareLights) LAUNCH("Sample direct/indirect - Henyey-Greenstein", k_medium_scatter<true>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1);
    else LAUNCH("Sample direct/indirect - Henyey-Greenstein", k_medium_scatter<false>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1);
    return 0;
}
// TraceShadowRays with media: IntersectShadowTr (wavefront/aggregate.cpp:70-88, intersect.h:165-274)
int wf_intersect_shadow_tr(wf_ctx *ctx, int depth) {
    if (int e = checkReady(ctx)) return e;
    if (!ctx->svHost.haveMedia) return fail(-1, "wf_intersect_shadow_tr: the scene has no media (use wf_intersect_shadow)");
    // (a scene with AnimatedPrimitives keeps the reference-order transmittance walk, which interpolates their transformations at the shadow
    //  ray's time: the transmittance wavefront's walk kernel has no ANIM variant — fuzz scene s6300008, round 6)
    if (ctx->fastOk && !ctx->animFast && !ctx->countTraversal && RetraceInline(ctx->genMode) && (ctx->trWavefront == 1 || (ctx->trWavefront < 0 && (ctx->svHost.nInstances > 0 || ctx->mediumLean)))) {
        // (round 6: also for one-level scenes whose media are all lean — k_tr_segment<true> runs at 3 waves (162 VGPRs) where the per-lane kernel is one
        //  wave of 366 + 110 registers per SIMD: cloud scene 17.3 against 18.5 ms, profiles/r06_transmittance_lean_wavefront_ab_cloud16.txt)
        // the transmittance wavefront (see k_tr_begin)
        LAUNCH("Reset transmittance queues", k_reset, 1, ctx->ws, (1u << CNT_TR0) | (1u << CNT_TR1), -1, 0);
        LAUNCH("Intersect shadow (Tr): begin", k_tr_begin, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws);
        for (int seg = 0; seg < WF_TR_SEGMENTS; ++seg) {
            const int cur = seg & 1;
            if (ctx->svHost.nInstances > 0) {
                if (ctx->genMode == 0) LAUNCHT("Intersect shadow (Tr): trace", (k_tr_trace<0, true>), ctx->persistentGrid, ctx->svHost, ctx->ws, ctx->fast, cur, ctx->spillArea());
                else LAUNCHT("Intersect shadow (Tr): trace", (k_tr_trace<1, true>), ctx->persistentGrid, ctx->svHost, ctx->ws, ctx->fast, cur, ctx->spillArea());
            } else {
                if (ctx->genMode == 0) LAUNCHT("Intersect shadow (Tr): trace", (k_tr_trace<0, false>), ctx->persistentGrid, ctx->svHost, ctx->ws, ctx->fast, cur, ctx->spillArea());
                else LAUNCHT("Intersect shadow (Tr): trace", (k_tr_trace<1, false>), ctx->persistentGrid, ctx->svHost, ctx->ws, ctx->fast, cur, ctx->spillArea());
            }
            if (ctx->mediumLean) LAUNCH("Intersect shadow (Tr): segment", k_tr_segment<true>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, cur);
            else LAUNCH("Intersect shadow (Tr): segment", k_tr_segment<false>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, cur);
            LAUNCH("Reset transmittance queues", k_reset, 1, ctx->ws, 1u << (CNT_TR0 + cur), -1, 0);
        }
        LAUNCH("Intersect shadow (Tr): rest", k_tr_rest, 128, ctx->svHost, ctx->ws, WF_TR_SEGMENTS & 1, ctx->stackSpill);
    } else if (ctx->fastOk && !ctx->animFast && !ctx->countTraversal && ctx->svHost.nInstances == 0)  // (the per-lane production walk has no two-level variant)
        if (ctx->svHost.haveAlpha || ctx->svHost.nQuadrics > 0) LAUNCHT("Intersect shadow (Tr)", k_shadow_tr_fast<true>, ctx->persistentGrid, ctx->svHost, ctx->ws, ctx->fast, ctx->spillArea());
        else if (ctx->mediumLean) LAUNCHT("Intersect shadow (Tr)", (k_shadow_tr_fast<false, true>), ctx->persistentGrid, ctx->svHost, ctx->ws, ctx->fast, ctx->spillArea());
        else LAUNCHT("Intersect shadow (Tr)", k_shadow_tr_fast<false>, ctx->persistentGrid, ctx->svHost, ctx->ws, ctx->fast, ctx->spillArea());
    else
        { if (ctx->svHost.haveAnimated) LAUNCH("Intersect shadow (Tr)", k_shadow_tr<true>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, ctx->stackSpill);
          else LAUNCH("Intersect shadow (Tr)", k_shadow_tr<false>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, ctx->stackSpill); }
    LAUNCH("Reset shadowRayQueue", k_reset, 1, ctx->ws, (1u << CNT_SHADOW), 65 + statDepth(depth), CNT_SHADOW);
    return 0;
}
int wf_handle_escaped(wf_ctx *ctx, int depth) {
    if (int e = checkReady(ctx)) return e;
    if (ctx->svHost.nInfiniteLights == 0) return 0;  // escapedRayQueue == nullptr (integrator.cpp:496-497)
    if (ctx->portalLights) LAUNCH("Handle escaped rays", k_handle_escaped<true>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1);
    else LAUNCH("Handle escaped rays", k_handle_escaped<false>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1);
    return 0;
}
int wf_handle_emissive(wf_ctx *ctx, int depth) {
    if (int e = checkReady(ctx)) return e;
    LAUNCH("Handle emitters hit by indirect rays", k_handle_emissive, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1);
    return 0;
}
static int EvalMaterialOn(wf_ctx *ctx, int material_type, int depth, hipStream_t stream, bool timed);
int wf_eval_material(wf_ctx *ctx, int material_type, int depth) {
    if (int e = checkReady(ctx)) return e;
    return EvalMaterialOn(ctx, material_type, depth, ctx->stream, true);
}
// The material stage of one depth for every type present.  The kernels of different types share nothing but queue counters (atomic pushes
// into the next ray queue and the shadow queue; every item touches only its own pixel sample's state): with WF_MAT_STREAMS=1
// each type's pair of kernels runs on a stream of its own, forked from and joined to the render stream.  MEASURED AND LEFT OFF (round 5,
// spec scene, 16 spp, same box, profiles/r05_material_split_ab_sm16.txt): 0.111-0.112 s per render with the streams against 0.108 s
// without — the kernels run at 2-4 waves per SIMD by their register need, a second kernel finds no free slots beside the first, and the
// interleaved queues only cost cache locality; VERDICT r4 item 8 asked for the experiment.
static int EvalMaterials(wf_ctx *ctx, int depth) {
    int present = 0;
    for (int m = 1; m < WF_MAT_NTYPES; ++m) present += ctx->matPresent[m] ? 1 : 0;
    const bool parallel = ctx->matStreams && present > 1 && ctx->profile != 1 && !ctx->traceLaunch && ctx->matSplit;
    if (!parallel) {
        for (int m = 1; m < WF_MAT_NTYPES; ++m)
            if (ctx->matPresent[m])
                if (int e = EvalMaterialOn(ctx, m, depth, ctx->stream, true)) return e;
        return 0;
    }
    Prof prof_(ctx, "Material stage: all types, shade + next-event estimation (parallel streams)");   // (profile 2: what bench.py prices; on the render stream, fork to join)
    HIPCHK(hipEventRecord(ctx->evMatFork, ctx->stream));
    for (int m = 1; m < WF_MAT_NTYPES; ++m) {
        if (!ctx->matPresent[m]) continue;
        if (!ctx->matStream[m]) {
            HIPCHK(hipStreamCreateWithFlags(&ctx->matStream[m], hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ctx->evMatJoin[m], hipEventDisableTiming));
        }
        HIPCHK(hipStreamWaitEvent(ctx->matStream[m], ctx->evMatFork, 0));
        if (int e = EvalMaterialOn(ctx, m, depth, ctx->matStream[m], false)) return e;
        HIPCHK(hipEventRecord(ctx->evMatJoin[m], ctx->matStream[m]));
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->evMatJoin[m], 0));
    }
    return 0;
}
static int EvalMaterialOn(wf_ctx *ctx, int material_type, int depth, hipStream_t stream, bool timed) {
    const int g = gridFor(ctx->maxQueueSize), cur = depth & 1;
    static const char *names[WF_MAT_NTYPES] = {"", "DiffuseMaterial + BxDF eval (Basic tex)", "ConductorMaterial + BxDF eval (Basic tex)",
                                               "DielectricMaterial + BxDF eval (Basic tex)", "ThinDielectricMaterial + BxDF eval (Basic tex)",
                                               "DiffuseTransmissionMaterial + BxDF eval (Basic tex)", "CoatedDiffuseMaterial + BxDF eval (Basic tex)",
                                               "CoatedConductorMaterial + BxDF eval (Basic tex)", "SubsurfaceMaterial + BxDF eval (Basic tex)", "HairMaterial + BxDF eval (Basic tex)",
                                               "MeasuredMaterial + BxDF eval (Basic tex)"};
    if (material_type == WF_MAT_INTERFACE) return 0;
    if (material_type < 0 || material_type >= WF_MAT_NTYPES) return fail(-1, "material type %d has no HIP kernel", material_type);
    const bool tex = ctx->svHost.texNeedsFootprint != 0;
    const bool vs = ctx->svHost.film.type == WF_FILM_GBUFFER || ctx->svHost.camera.anim.actually_animated || ctx->svHost.haveAnimated ||
                    (ctx->svHost.haveCurves && ctx->svHost.haveQuadricAlpha);   // the variant that fills the visible surface / differentiates a moving camera / meets animated instances or alpha-textured curves
#if defined(WF_HAVE_FUSED_MAT)
    if (!ctx->matSplit) {
        Prof prof_(ctx, timed ? names[material_type] : "(untimed)", stream);
        const bool rare = ctx->rareLights || vs;   // a portal infinite light, or a GBufferFilm: the variant that can reach the portal samplers / fills the visible surface
        switch (material_type) {
        case 1: (rare ? wf_launch_eval_material_1_2 : tex ? wf_launch_eval_material_1_1 : wf_launch_eval_material_1_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 2: (rare ? wf_launch_eval_material_2_2 : tex ? wf_launch_eval_material_2_1 : wf_launch_eval_material_2_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 3: (rare ? wf_launch_eval_material_3_2 : tex ? wf_launch_eval_material_3_1 : wf_launch_eval_material_3_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 4: (rare ? wf_launch_eval_material_4_2 : tex ? wf_launch_eval_material_4_1 : wf_launch_eval_material_4_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 5: (rare ? wf_launch_eval_material_5_2 : tex ? wf_launch_eval_material_5_1 : wf_launch_eval_material_5_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 6: (rare ? wf_launch_eval_material_6_2 : tex ? wf_launch_eval_material_6_1 : wf_launch_eval_material_6_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 7: (rare ? wf_launch_eval_material_7_2 : tex ? wf_launch_eval_material_7_1 : wf_launch_eval_material_7_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 8: (rare ? wf_launch_eval_material_8_2 : tex ? wf_launch_eval_material_8_1 : wf_launch_eval_material_8_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 9: (rare ? wf_launch_eval_material_9_2 : tex ? wf_launch_eval_material_9_1 : wf_launch_eval_material_9_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 10: (rare ? wf_launch_eval_material_10_2 : tex ? wf_launch_eval_material_10_1 : wf_launch_eval_material_10_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        }
        return 0;
    }
#endif
    // the two halves (wf_kernels.h MatShade / MatNee; the items' NeeItems stay in ws.neeRec between them)
    {
        Prof prof_(ctx, timed ? names[material_type] : "(untimed)", stream);
        // shade variant: 0 / 1 the LEAN kernels (triangle-only scenes whose textures are all constants, image maps or bilerps: SceneLean),
        // without / with the texture footprint and bump block; 2 general; 3 general + visible surface / moving camera
        const int v = vs ? 3 : (!ctx->leanType[material_type] ? 2 : (tex ? 1 : 0));
        switch (material_type) {
        case 1: (v == 3 ? wf_launch_mat_shade_1_3 : v == 2 ? wf_launch_mat_shade_1_2 : v == 1 ? wf_launch_mat_shade_1_1 : wf_launch_mat_shade_1_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 2: (v == 3 ? wf_launch_mat_shade_2_3 : v == 2 ? wf_launch_mat_shade_2_2 : v == 1 ? wf_launch_mat_shade_2_1 : wf_launch_mat_shade_2_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 3: (v == 3 ? wf_launch_mat_shade_3_3 : v == 2 ? wf_launch_mat_shade_3_2 : v == 1 ? wf_launch_mat_shade_3_1 : wf_launch_mat_shade_3_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 4: (v == 3 ? wf_launch_mat_shade_4_3 : v == 2 ? wf_launch_mat_shade_4_2 : v == 1 ? wf_launch_mat_shade_4_1 : wf_launch_mat_shade_4_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 5: (v == 3 ? wf_launch_mat_shade_5_3 : v == 2 ? wf_launch_mat_shade_5_2 : v == 1 ? wf_launch_mat_shade_5_1 : wf_launch_mat_shade_5_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 6: (v == 3 ? wf_launch_mat_shade_6_3 : v == 2 ? wf_launch_mat_shade_6_2 : v == 1 ? wf_launch_mat_shade_6_1 : wf_launch_mat_shade_6_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 7: (v == 3 ? wf_launch_mat_shade_7_3 : v == 2 ? wf_launch_mat_shade_7_2 : v == 1 ? wf_launch_mat_shade_7_1 : wf_launch_mat_shade_7_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 8: (v == 3 ? wf_launch_mat_shade_8_3 : v == 2 ? wf_launch_mat_shade_8_2 : v == 1 ? wf_launch_mat_shade_8_1 : wf_launch_mat_shade_8_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 9: (v == 3 ? wf_launch_mat_shade_9_3 : v == 2 ? wf_launch_mat_shade_9_2 : v == 1 ? wf_launch_mat_shade_9_1 : wf_launch_mat_shade_9_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        case 10: (v == 3 ? wf_launch_mat_shade_10_3 : v == 2 ? wf_launch_mat_shade_10_2 : v == 1 ? wf_launch_mat_shade_10_1 : wf_launch_mat_shade_10_0)(stream, g, &ctx->svHost, &ctx->ws, cur); break;
        }
    }
    {
        static const char *neeNames[WF_MAT_NTYPES] = {"", "DiffuseMaterial: next-event estimation", "ConductorMaterial: next-event estimation", "DielectricMaterial: next-event estimation",
                                                      "ThinDielectricMaterial: next-event estimation", "DiffuseTransmissionMaterial: next-event estimation",
                                                      "CoatedDiffuseMaterial: next-event estimation", "CoatedConductorMaterial: next-event estimation",
                                                      "SubsurfaceMaterial: next-event estimation", "HairMaterial: next-event estimation", "MeasuredMaterial: next-event estimation"};
        Prof prof_(ctx, timed ? neeNames[material_type] : "(untimed)", stream);
        switch (material_type) {
        case 1: (ctx->rareLights ? wf_launch_mat_nee_1_1 : wf_launch_mat_nee_1_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        case 2: (ctx->rareLights ? wf_launch_mat_nee_2_1 : wf_launch_mat_nee_2_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        case 3: (ctx->rareLights ? wf_launch_mat_nee_3_1 : wf_launch_mat_nee_3_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        case 4: (ctx->rareLights ? wf_launch_mat_nee_4_1 : wf_launch_mat_nee_4_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        case 5: (ctx->rareLights ? wf_launch_mat_nee_5_1 : wf_launch_mat_nee_5_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        case 6: (ctx->rareLights ? wf_launch_mat_nee_6_1 : wf_launch_mat_nee_6_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        case 7: (ctx->rareLights ? wf_launch_mat_nee_7_1 : wf_launch_mat_nee_7_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        case 8: (ctx->rareLights ? wf_launch_mat_nee_8_1 : wf_launch_mat_nee_8_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        case 9: (ctx->rareLights ? wf_launch_mat_nee_9_1 : wf_launch_mat_nee_9_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        case 10: (ctx->rareLights ? wf_launch_mat_nee_10_1 : wf_launch_mat_nee_10_0)(stream, g, &ctx->svHost, &ctx->ws); break;
        }
    }
    return 0;
}
int wf_intersect_shadow(wf_ctx *ctx, int depth) {
    if (int e = checkReady(ctx)) return e;
    if (ctx->countTraversal)
        { if (ctx->svHost.haveAnimated) LAUNCH("Intersect shadow", (k_intersect_shadow<true, true>), gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, ctx->stackSpill);
          else LAUNCH("Intersect shadow", k_intersect_shadow<true>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, ctx->stackSpill); }
    else if (ctx->fastOk) {
        if ((ctx->raySort & 2) && (depth >= 1 || (ctx->raySort & 4)))
            if (int e = SortQueue(ctx, true, 0)) return e;
        int *cursor = nullptr;
        if (ctx->splitRoute > 1) {
            cursor = ctx->ws.counters + CNT_CURSOR_SHADOW * CNT_STRIDE;   // (zeroed by the "Reset shadowRayQueue" launch that follows every any-hit launch)
            if (ctx->cursorDirty[1]) HIPCHK(hipMemsetAsync(cursor, 0, sizeof(int), ctx->stream));
            ctx->cursorDirty[1] = true;
        }
        if (ctx->deferGeneral && ctx->splitRoute) {
            LAUNCHT_VARIANT_GEN("Intersect shadow", k_shadow_fast, 4 + ctx->genTri, ctx->persistentGridShadow, ctx->svHost, ctx->ws, ctx->fast, ctx->spillArea(), cursor, ctx->cursorChunkShadow, (const int *)nullptr);
            LAUNCHT_VARIANT_GEN("Intersect shadow: rays that met a general primitive", k_shadow_fast, ctx->genMode, ctx->persistentGridShadowGen, ctx->svHost, ctx->ws, ctx->fast, ctx->spillArea(), (int *)nullptr, ctx->cursorChunkShadow, (const int *)ctx->ws.deferQ);
        } else
        LAUNCHT_VARIANT_GEN("Intersect shadow", k_shadow_fast, ctx->animFast ? 8 + ctx->genMode : ctx->genMode, ctx->persistentGridShadow, ctx->svHost, ctx->ws, ctx->fast, ctx->spillArea(), cursor, ctx->cursorChunkShadow, (const int *)nullptr);
    } else
        { if (ctx->svHost.haveAnimated) LAUNCH("Intersect shadow", (k_intersect_shadow<false, true>), gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, ctx->stackSpill);
          else LAUNCH("Intersect shadow", k_intersect_shadow<false>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, ctx->stackSpill); }
    // "Reset shadowRayQueue": stats->shadowRays[depth] += size; Reset (integrator.cpp:581-585)
    LAUNCH("Reset shadowRayQueue", k_reset, 1, ctx->ws, (1u << CNT_SHADOW) | (1u << CNT_CURSOR_SHADOW) | (1u << CNT_DEFER_SHADOW), 65 + statDepth(depth), CNT_SHADOW);
    ctx->cursorDirty[1] = false;
    return 0;
}
// K12: SampleSubsurface (wavefront/subsurface.cpp:18-203) in its three launches
int wf_subsurface_probe(wf_ctx *ctx, int depth) {
    if (int e = checkReady(ctx)) return e;
    if (!ctx->svHost.haveSubsurface) return 0;
    (void)depth;
    LAUNCH("Get BSSRDF and enqueue probe ray", k_subsurface_probe, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws);
    return 0;
}
int wf_intersect_one_random(wf_ctx *ctx) {
    if (int e = checkReady(ctx)) return e;
    if (!ctx->svHost.haveSubsurface) return 0;
    // reference-order walk (the chain of probe hits is a dependent sequence per item; the subsurface path is a side path)
    if (ctx->svHost.haveAnimated) LAUNCH("Intersect one random (subsurface probe)", k_intersect_one_random<true>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, ctx->stackSpill);
    else LAUNCH("Intersect one random (subsurface probe)", k_intersect_one_random<false>, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, ctx->stackSpill);
    return 0;
}
int wf_subsurface_scatter(wf_ctx *ctx, int depth) {
    if (int e = checkReady(ctx)) return e;
    if (!ctx->svHost.haveSubsurface) return 0;
    LAUNCH("Handle out-scattering after SSS", k_subsurface_scatter, gridFor(ctx->maxQueueSize), ctx->svHost, ctx->ws, depth & 1);
    return 0;
}
int wf_update_film(wf_ctx *ctx) {
    if (int e = checkReady(ctx)) return e;
    if (ctx->ws.slotStride > 0 && ctx->ws.slotStride == ctx->passSamples && ctx->passSamples <= FILM_MAX_SLOTS && ctx->svHost.film.type == WF_FILM_RGB)
        LAUNCH("Update film", k_update_film_pm, std::min(4096, (ctx->ws.pixelsPerPass + FILM_PIX - 1) / FILM_PIX), ctx->svHost, ctx->ws, ctx->passSamples);
    else
        LAUNCH("Update film", k_update_film, gridFor(ctx->ws.pixelsPerPass), ctx->svHost, ctx->ws, ctx->passSamples);
    return 0;
}

// integrator.cpp:357-434 for one (y0, sampleIndex): everything is enqueued, nothing synchronises
int wf_render_pass(wf_ctx *ctx, int y0, int sample_index) {
    if (int e = checkReady(ctx)) return e;
    int e;
    if ((e = wf_reset_ray_queue(ctx, 0))) return e;
    if ((e = wf_gen_camera_rays(ctx, y0, sample_index))) return e;
    for (int depth = 0; true; ++depth) {
        if ((e = wf_reset_stage_queues(ctx, depth))) return e;
        // (GenerateRaySamples does not depend on the intersections: it is issued after the closest-hit launch so that it runs beside
        // the re-trace; the per-stage entry points keep the reference's order)
        ctx->deferJoin = ctx->overlapRetrace && ctx->fastOk && ctx->splitRoute && !ctx->countTraversal && !ctx->svHost.haveMix;
        if ((e = wf_intersect_closest(ctx, depth))) { ctx->deferJoin = false; return e; }
        ctx->deferJoin = false;
        if ((e = wf_gen_ray_samples(ctx, depth, sample_index))) return e;
        if ((e = JoinRetrace(ctx))) return e;
        if ((e = wf_medium_sample(ctx, depth))) return e;
        if ((e = wf_handle_escaped(ctx, depth))) return e;
        if ((e = wf_handle_emissive(ctx, depth))) return e;
        if (depth == ctx->maxDepth) break;
        if ((e = EvalMaterials(ctx, depth))) return e;
        if ((e = ctx->svHost.haveMedia ? wf_intersect_shadow_tr(ctx, depth) : wf_intersect_shadow(ctx, depth))) return e;
        if (ctx->svHost.haveSubsurface) {  // SampleSubsurface (integrator.cpp:431) ends with its own TraceShadowRays
            if ((e = wf_subsurface_probe(ctx, depth)) || (e = wf_intersect_one_random(ctx)) || (e = wf_subsurface_scatter(ctx, depth))) return e;
            if ((e = ctx->svHost.haveMedia ? wf_intersect_shadow_tr(ctx, depth) : wf_intersect_shadow(ctx, depth))) return e;
        }
    }
    return wf_update_film(ctx);
}

int wf_film_download(wf_ctx *ctx, double *dst) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    HIPCHK(hipMemcpyAsync(dst, ctx->ws.film, (size_t)ctx->W * ctx->H * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int wf_film_spectral_download(wf_ctx *ctx, double *dst) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    if (!ctx->ws.filmSpectral) return fail(-1, "wf_film_spectral_download: the scene's film is not a spectral film");
    HIPCHK(hipMemcpyAsync(dst, ctx->ws.filmSpectral, (size_t)ctx->W * ctx->H * 2 * ctx->svHost.film.n_buckets * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int wf_film_gbuffer_download(wf_ctx *ctx, wf_gbuffer_pixel *dst) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    if (!ctx->ws.filmGBuffer) return fail(-1, "wf_film_gbuffer_download: the scene's film is not a gbuffer film");
    HIPCHK(hipMemcpyAsync(dst, ctx->ws.filmGBuffer, (size_t)ctx->W * ctx->H * sizeof(wf_gbuffer_pixel), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int wf_film_upload(wf_ctx *ctx, const double *src) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    HIPCHK(hipMemcpyAsync(ctx->ws.film, src, (size_t)ctx->W * ctx->H * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int wf_film_device_ptr(wf_ctx *ctx, void **dptr, uint64_t *nbytes) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    *dptr = ctx->ws.film;
    *nbytes = (uint64_t)ctx->W * ctx->H * 4 * sizeof(double);
    return 0;
}
// device-to-device copies of the film accumulators, for the multi-GPU film reduce: the caller owns a
// device buffer of wf_film_device_ptr's size (e.g. a torch tensor handed to RCCL)
int wf_film_copy_to_device(wf_ctx *ctx, void *dst_device) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    HIPCHK(hipMemcpyAsync(dst_device, ctx->ws.film, (size_t)ctx->W * ctx->H * 4 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int wf_film_copy_from_device(wf_ctx *ctx, const void *src_device) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    HIPCHK(hipMemcpyAsync(ctx->ws.film, src_device, (size_t)ctx->W * ctx->H * 4 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
// The gather of a strip-partitioned render (wf_set_strips) without a host round trip or a collective: the scanline strips `src` owns
// are copied from its film into `dst`'s film — peer to peer over xGMI when the contexts sit on different devices, device to device on
// one.  A strip is `height` whole scanlines = one contiguous range of the [H][W][4] double film; a rank of N moves 1/N of the film
// (the reduce(SUM) of full films it replaces moved N x the film, mostly zeros: VERDICT r3).  Both contexts must hold the same scene and
// be idle (wf_sync'ed); the copies run on dst's stream and are synchronised before returning.
int wf_film_gather_strips(wf_ctx *dst, wf_ctx *src) {
    if (!dst || !src || !dst->sceneLoaded || !src->sceneLoaded) return fail(-1, "wf_film_gather_strips: no scene uploaded");
    if (dst->W != src->W || dst->H != src->H) return fail(-1, "wf_film_gather_strips: the films differ in size");
    if (dst == src) return 0;
    useDevice(dst);
    const int count = src->ws.stripCount > 1 ? src->ws.stripCount : 1, rank = count > 1 ? src->ws.stripRank : 0, height = count > 1 ? src->ws.stripHeight : src->H;
    const size_t rowBytes = (size_t)src->W * 4 * sizeof(double);
    if (dst->device != src->device) {
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, dst->device, src->device);
        if (can) { hipError_t e = hipDeviceEnablePeerAccess(src->device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(); else (void)hipGetLastError(); }
    }
    for (int y0 = 0; y0 < src->H; y0 += height) {
        if ((y0 / height) % count != rank) continue;
        const int rows = y0 + height <= src->H ? height : src->H - y0;
        char *d = reinterpret_cast<char *>(dst->ws.film) + (size_t)y0 * rowBytes;
        const char *sp = reinterpret_cast<const char *>(src->ws.film) + (size_t)y0 * rowBytes;
        if (dst->device == src->device) HIPCHK(hipMemcpyAsync(d, sp, rows * rowBytes, hipMemcpyDeviceToDevice, dst->stream));
        else HIPCHK(hipMemcpyPeerAsync(d, dst->device, sp, src->device, rows * rowBytes, dst->stream));
    }
    HIPCHK(hipStreamSynchronize(dst->stream));
    return 0;
}
// the ray counters of another context added to this one's (the statistics of a multi-device render, summed on the gathering context)
int wf_stats_add(wf_ctx *dst, wf_ctx *src) {
    if (!dst || !src || !dst->sceneLoaded || !src->sceneLoaded) return fail(-1, "wf_stats_add: no scene uploaded");
    // cameraRays + indirect[64] + shadow[64], then the per-material-type and medium-sample item counters ([129, 129 + 16))
    constexpr int NSTAT = 129 + 16;
    unsigned long long a[NSTAT], b[NSTAT];
    useDevice(src);
    HIPCHK(hipMemcpy(b, src->ws.stats, sizeof(b), hipMemcpyDeviceToHost));
    useDevice(dst);
    HIPCHK(hipMemcpy(a, dst->ws.stats, sizeof(a), hipMemcpyDeviceToHost));
    for (int i = 0; i < NSTAT; ++i) a[i] += b[i];
    HIPCHK(hipMemcpy(dst->ws.stats, a, sizeof(a), hipMemcpyHostToDevice));
    useDevice(src);
    HIPCHK(hipMemset(src->ws.stats, 0, sizeof(b)));   // (moved, not copied: a second call adds only what src counted since)
    useDevice(dst);
    return 0;
}
// items the material stage evaluated since the last wf_film_clear, per material type (out[0 .. WF_MAT_NTYPES)), and the items of the
// medium-sample stage (out[WF_MAT_NTYPES]): the counts the queues held when they were reset, plus what they hold now
int wf_material_items_download(wf_ctx *ctx, uint64_t out[16]) {
    if (int e = checkReady(ctx)) return e;
    unsigned long long h[16];
    std::vector<int32_t> cnt((size_t)CNT_COUNT * CNT_STRIDE);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(h, ctx->ws.stats + 129, sizeof(h), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cnt.data(), ctx->ws.counters, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) out[i] = i <= WF_MAT_NTYPES ? h[i] + (uint64_t)cnt[(size_t)(CNT_MAT0 + i) * CNT_STRIDE] : 0;
    return 0;
}
int wf_stats_download(wf_ctx *ctx, wf_render_stats *out) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    unsigned long long h[129];
    HIPCHK(hipMemcpyAsync(h, ctx->ws.stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    out->camera_rays = h[0];
    for (int i = 0; i < 64; ++i) { out->indirect_rays[i] = h[1 + i]; out->shadow_rays[i] = h[65 + i]; }
    return 0;
}

int wf_profile_enable(wf_ctx *ctx, int enabled) {
    if (!ctx) return fail(-1, "null context");
    ctx->profile = enabled;
    return 0;
}
// total milliseconds and launch count of the named kernel since the last report (drains nothing)
int wf_kernel_time_ms(wf_ctx *ctx, const char *name, double *total_ms, int *launches) {
    if (!ctx || !name) return fail(-1, "null argument");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    double sum = 0;
    int n = 0;
    for (auto &e : ctx->events)
        if (e.name == name) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
            sum += ms;
            ++n;
        }
    if (total_ms) *total_ms = sum;
    if (launches) *launches = n;
    return 0;
}
int wf_profile_report(wf_ctx *ctx, wf_kernel_profile_entry *entries, int max_entries, int *n_out) {
    if (!ctx) return fail(-1, "null context");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::map<std::string, wf_kernel_profile_entry> agg;
    std::vector<std::string> order;
    for (auto &e : ctx->events) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        auto it = agg.find(e.name);
        if (it == agg.end()) {
            wf_kernel_profile_entry pe{};
            snprintf(pe.name, sizeof(pe.name), "%s", e.name.c_str());
            pe.launches = 1; pe.total_ms = pe.min_ms = pe.max_ms = ms;
            agg[e.name] = pe;
            order.push_back(e.name);
        } else {
            it->second.launches++;
            it->second.total_ms += ms;
            if (ms < it->second.min_ms) it->second.min_ms = ms;
            if (ms > it->second.max_ms) it->second.max_ms = ms;
        }
        ctx->eventPool.push_back(e.a);
        ctx->eventPool.push_back(e.b);
    }
    ctx->events.clear();
    int n = 0;
    for (auto &name : order) {
        if (n >= max_entries) break;
        entries[n++] = agg[name];
    }
    if (n_out) *n_out = n;
    return 0;
}

int wf_counters_enable(wf_ctx *ctx, int enabled) {
    if (!ctx) return fail(-1, "null context");
    ctx->countTraversal = enabled != 0;
    return 0;
}
int wf_counters_download(wf_ctx *ctx, wf_traversal_counters *out) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    unsigned long long h[8];
    HIPCHK(hipMemcpyAsync(h, ctx->ws.trav, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    out->closest_rays = h[0]; out->closest_nodes = h[1]; out->closest_tris = h[2]; out->closest_hits = h[3];
    out->shadow_rays = h[4]; out->shadow_nodes = h[5]; out->shadow_tris = h[6]; out->shadow_unoccluded = h[7];
    return 0;
}

int wf_trace_closest_device(wf_ctx *ctx, int n, const float *rays7, wf_hit_record *out) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    if (ctx->svHost.haveAnimated) return fail(-1, "%s: the scene has animated primitives — a ray needs its time (wf_trace_closest_host_t / wf_trace_any_host_t)", __func__);
    useDevice(ctx);
    if (n <= 0) return 0;
    if (!ctx->fastOk) { LAUNCH("trace closest (device rays)", k_trace_closest, gridFor(n), ctx->svHost, n, rays7, out, ctx->stackSpill, 0); return 0; }
    LAUNCHT_VARIANT("trace closest fast (device rays)", k_trace_closest_fast, 0, ctx->persistentGrid, ctx->svHost, ctx->fast, n, rays7, out, ctx->spillArea());
    // (the variants that do not resolve their near ties inside the walk mark them: re-traced in reference order)
    if (!RetraceInline(ctx->genMode)) LAUNCH("trace closest (near-tie re-trace)", k_trace_closest, gridFor(n), ctx->svHost, n, rays7, out, ctx->stackSpill, 1);
    return 0;
}
int wf_trace_any_device(wf_ctx *ctx, int n, const float *rays7, int32_t *occluded) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    if (ctx->svHost.haveAnimated) return fail(-1, "%s: the scene has animated primitives — a ray needs its time (wf_trace_closest_host_t / wf_trace_any_host_t)", __func__);
    useDevice(ctx);
    if (n <= 0) return 0;
    if (!ctx->fastOk) { LAUNCH("trace any (device rays)", k_trace_any, gridFor(n), ctx->svHost, n, rays7, occluded, (int32_t *)nullptr, (int32_t *)nullptr, ctx->stackSpill); return 0; }
    LAUNCHT_VARIANT("trace any fast (device rays)", k_trace_any_fast, 0, ctx->persistentGrid, ctx->svHost, ctx->fast, n, rays7, occluded, ctx->spillArea());
    return 0;
}
int wf_device_alloc(wf_ctx *ctx, uint64_t nbytes, void **dptr) {
    if (!ctx || !dptr) return fail(-1, "null argument");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMalloc(dptr, nbytes ? nbytes : 1));
    return 0;
}
int wf_device_free(wf_ctx *ctx, void *dptr) {
    if (!ctx) return fail(-1, "null context");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(dptr));
    return 0;
}
int wf_device_upload(wf_ctx *ctx, void *dst_device, const void *src_host, uint64_t nbytes) {
    if (!ctx) return fail(-1, "null context");
    HIPCHK(hipMemcpyAsync(dst_device, src_host, nbytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int wf_device_download(wf_ctx *ctx, void *dst_host, const void *src_device, uint64_t nbytes) {
    if (!ctx) return fail(-1, "null context");
    HIPCHK(hipMemcpyAsync(dst_host, src_device, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int wf_trace_closest_host(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax, wf_hit_record *out, int count_visits) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    if (ctx->svHost.haveAnimated) return fail(-1, "%s: the scene has animated primitives — a ray needs its time (wf_trace_closest_host_t / wf_trace_any_host_t)", __func__);
    useDevice(ctx);
    if (n <= 0) return 0;
    std::vector<float> rays((size_t)n * 7);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) { rays[(size_t)i * 7 + k] = o[3 * i + k]; rays[(size_t)i * 7 + 3 + k] = d[3 * i + k]; }
        rays[(size_t)i * 7 + 6] = tmax[i];
    }
    float *dr = nullptr;
    wf_hit_record *dh = nullptr;
    HIPCHK(hipMalloc((void **)&dr, rays.size() * sizeof(float)));
    HIPCHK(hipMalloc((void **)&dh, (size_t)n * sizeof(wf_hit_record)));
    HIPCHK(hipMemcpyAsync(dr, rays.data(), rays.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (count_visits || !ctx->fastOk) {
        LAUNCH("trace closest (host rays)", k_trace_closest, gridFor(n), ctx->svHost, n, dr, dh, ctx->stackSpill, 0);
    } else if (int e = wf_trace_closest_device(ctx, n, dr, dh)) return e;
    HIPCHK(hipMemcpyAsync(out, dh, (size_t)n * sizeof(wf_hit_record), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(dr));
    HIPCHK(hipFree(dh));
    return 0;
}
// IntersectShadowTr on caller-supplied shadow rays: a scratch WorkState over temporary device arrays (one "pixel" per ray) run through
// the same transmittance kernels as the render
int wf_trace_shadow_tr_host(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax, const int32_t *medium, const float *lambda,
                            const float *Ld, const float *r_u, const float *r_l, float *out_L) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    if (!ctx->svHost.haveMedia) return fail(-1, "wf_trace_shadow_tr_host: the scene has no media");
    if (ctx->svHost.haveAnimated) return fail(-1, "wf_trace_shadow_tr_host: the scene has animated primitives and this entry point carries no ray times (not supported)");
    if (n <= 0) return 0;
    std::vector<F4> ho(n), hd(n), hl(n), hp(n, F4{1, 1, 1, 1});
    for (int i = 0; i < n; ++i) {
        ho[i] = F4{o[3 * i], o[3 * i + 1], o[3 * i + 2], tmax[i]};
        hd[i] = F4{d[3 * i], d[3 * i + 1], d[3 * i + 2], BitsToFloat((uint32_t)i)};   // pixelIndex = i
        hl[i] = F4{lambda[4 * i], lambda[4 * i + 1], lambda[4 * i + 2], lambda[4 * i + 3]};
    }
    WorkState ws = ctx->ws;   // counters / stats of the context, every per-item array replaced below
    std::vector<void *> tmp;
    auto up = [&](auto **dst, const void *src, size_t bytes) -> int {
        void *p = nullptr;
        HIPCHK(hipMalloc(&p, bytes));
        tmp.push_back(p);
        if (src) HIPCHK(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        else HIPCHK(hipMemsetAsync(p, 0, bytes, ctx->stream));
        *dst = (std::remove_reference_t<decltype(**dst)> *)p;
        return 0;
    };
    int e;
    int32_t *cnt = nullptr;
    if ((e = up(&ws.sq.o, ho.data(), n * sizeof(F4))) || (e = up(&ws.sq.d, hd.data(), n * sizeof(F4))) || (e = up(&ws.sq.Ld, Ld, n * sizeof(F4))) ||
        (e = up(&ws.sq.r_u, r_u, n * sizeof(F4))) || (e = up(&ws.sq.r_l, r_l, n * sizeof(F4))) || (e = up(&ws.sq.medium, medium, n * sizeof(int32_t))) ||
        (e = up(&ws.lambda, hl.data(), n * sizeof(F4))) || (e = up(&ws.lambdaPdf, hp.data(), n * sizeof(F4))) || (e = up(&ws.L, nullptr, n * sizeof(F4))) ||
        (e = up(&cnt, nullptr, (size_t)CNT_COUNT * CNT_STRIDE * sizeof(int32_t))))
        return e;
    HIPCHK(hipMemcpyAsync(cnt + CNT_SHADOW * CNT_STRIDE, &n, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    ws.counters = cnt;
    if (ctx->fastOk && ctx->svHost.nInstances == 0) {
        if (ctx->svHost.haveAlpha || ctx->svHost.nQuadrics > 0) LAUNCHT("shadow Tr (host rays)", k_shadow_tr_fast<true>, ctx->persistentGrid, ctx->svHost, ws, ctx->fast, ctx->spillArea());
        else LAUNCHT("shadow Tr (host rays)", k_shadow_tr_fast<false>, ctx->persistentGrid, ctx->svHost, ws, ctx->fast, ctx->spillArea());
    } else LAUNCH("shadow Tr (host rays)", k_shadow_tr<false>, gridFor(n), ctx->svHost, ws, ctx->stackSpill);
    HIPCHK(hipMemcpyAsync(out_L, ws.L, n * sizeof(F4), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (void *p : tmp) HIPCHK(hipFree(p));
    return 0;
}
int wf_trace_one_random_host(wf_ctx *ctx, int n, const float *p0, const float *p1, const int32_t *material, wf_hit_record *out, float *reservoir_pdf) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    // (the reference walks its subsurface probe segments at time 0, wavefront/subsurface.cpp:70: an animated scene would need that walk's
    //  ANIM variant here — refused instead of answered for the start-time geometry)
    if (ctx->svHost.haveAnimated) return fail(-1, "wf_trace_one_random_host: the scene has animated primitives (not supported by this entry point)");
    if (n <= 0) return 0;
    std::vector<float> segs((size_t)n * 6);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) { segs[(size_t)i * 6 + k] = p0[3 * i + k]; segs[(size_t)i * 6 + 3 + k] = p1[3 * i + k]; }
    float *ds = nullptr, *dp = nullptr;
    int32_t *dm = nullptr;
    wf_hit_record *dh = nullptr;
    HIPCHK(hipMalloc((void **)&ds, segs.size() * sizeof(float)));
    HIPCHK(hipMalloc((void **)&dm, (size_t)n * sizeof(int32_t)));
    HIPCHK(hipMalloc((void **)&dh, (size_t)n * sizeof(wf_hit_record)));
    HIPCHK(hipMalloc((void **)&dp, (size_t)n * sizeof(float)));
    HIPCHK(hipMemcpyAsync(ds, segs.data(), segs.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(dm, material, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    LAUNCH("intersect one random (host segments)", k_trace_one_random, gridFor(n), ctx->svHost, n, ds, dm, dh, dp, ctx->stackSpill);
    HIPCHK(hipMemcpyAsync(out, dh, (size_t)n * sizeof(wf_hit_record), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(reservoir_pdf, dp, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(ds)); HIPCHK(hipFree(dm)); HIPCHK(hipFree(dh)); HIPCHK(hipFree(dp));
    return 0;
}
int wf_trace_any_host(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax, int32_t *occluded, int32_t *nodes_visited, int32_t *tris_tested) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    if (ctx->svHost.haveAnimated) return fail(-1, "%s: the scene has animated primitives — a ray needs its time (wf_trace_closest_host_t / wf_trace_any_host_t)", __func__);
    useDevice(ctx);
    if (n <= 0) return 0;
    std::vector<float> rays((size_t)n * 7);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) { rays[(size_t)i * 7 + k] = o[3 * i + k]; rays[(size_t)i * 7 + 3 + k] = d[3 * i + k]; }
        rays[(size_t)i * 7 + 6] = tmax[i];
    }
    float *dr = nullptr;
    int32_t *dres = nullptr;
    HIPCHK(hipMalloc((void **)&dr, rays.size() * sizeof(float)));
    HIPCHK(hipMalloc((void **)&dres, (size_t)3 * n * sizeof(int32_t)));
    HIPCHK(hipMemcpyAsync(dr, rays.data(), rays.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (nodes_visited || tris_tested || !ctx->fastOk) {
        LAUNCH("trace any (host rays)", k_trace_any, gridFor(n), ctx->svHost, n, dr, dres, dres + n, dres + 2 * (size_t)n, ctx->stackSpill);
    } else if (int e = wf_trace_any_device(ctx, n, dr, dres)) return e;
    HIPCHK(hipMemcpyAsync(occluded, dres, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (nodes_visited) HIPCHK(hipMemcpyAsync(nodes_visited, dres + n, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (tris_tested) HIPCHK(hipMemcpyAsync(tris_tested, dres + 2 * (size_t)n, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(dr));
    HIPCHK(hipFree(dres));
    return 0;
}
// IntersectClosest / IntersectShadow on caller-supplied rays WITH their times (the untimed entry points would walk an animated
// scene at its start time): the reference-order walks, interpolating every AnimatedPrimitive's transformation at the ray's time
static int TraceTimed(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax, const float *time, wf_hit_record *hits, int32_t *occluded) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    if (!o || !d || !tmax || !time) return fail(-1, "wf_trace_*_host_t: null ray arrays");
    useDevice(ctx);
    if (n <= 0) return 0;
    std::vector<float> rays((size_t)n * 8);
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) { rays[(size_t)i * 7 + k] = o[3 * i + k]; rays[(size_t)i * 7 + 3 + k] = d[3 * i + k]; }
        rays[(size_t)i * 7 + 6] = tmax[i];
        rays[(size_t)n * 7 + i] = time[i];
    }
    float *dr = nullptr;
    void *dres = nullptr;
    const size_t resBytes = hits ? (size_t)n * sizeof(wf_hit_record) : (size_t)n * sizeof(int32_t);
    HIPCHK(hipMalloc((void **)&dr, rays.size() * sizeof(float)));
    HIPCHK(hipMalloc(&dres, resBytes));
    HIPCHK(hipMemcpyAsync(dr, rays.data(), rays.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if (hits) LAUNCH("trace closest (host rays with times)", k_trace_closest_timed, gridFor(n), ctx->svHost, n, dr, dr + (size_t)n * 7, (wf_hit_record *)dres, ctx->stackSpill);
    else LAUNCH("trace any (host rays with times)", k_trace_any_timed, gridFor(n), ctx->svHost, n, dr, dr + (size_t)n * 7, (int32_t *)dres, ctx->stackSpill);
    HIPCHK(hipMemcpyAsync(hits ? (void *)hits : (void *)occluded, dres, resBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(dr));
    HIPCHK(hipFree(dres));
    return 0;
}
int wf_trace_closest_host_t(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax, const float *time, wf_hit_record *out) {
    if (!out) return fail(-1, "wf_trace_closest_host_t: null out");
    return TraceTimed(ctx, n, o, d, tmax, time, out, nullptr);
}
int wf_trace_any_host_t(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax, const float *time, int32_t *occluded) {
    if (!occluded) return fail(-1, "wf_trace_any_host_t: null occluded");
    return TraceTimed(ctx, n, o, d, tmax, time, nullptr, occluded);
}
int wf_sampler_probe(wf_ctx *ctx, int n, const int32_t *px, const int32_t *py, const int32_t *sample_index, int start_dim, int ndims, float *out) {
    if (!ctx || !ctx->sceneLoaded) return fail(-1, "no scene uploaded");
    useDevice(ctx);
    if (n <= 0) return 0;
    if (ndims == 0 || ndims < -4 || ndims == -1) return fail(-1, "wf_sampler_probe: ndims %d", ndims);
    const int mode = ndims;
    ndims = mode > 0 ? mode : mode == -3 ? 30 : 2;   // floats per record (SamplerProbeRecord, wf_camera.h)
    int32_t *din = nullptr;
    float *dout = nullptr;
    HIPCHK(hipMalloc((void **)&din, (size_t)3 * n * sizeof(int32_t)));
    HIPCHK(hipMalloc((void **)&dout, (size_t)n * ndims * sizeof(float)));
    HIPCHK(hipMemcpyAsync(din, px, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(din + n, py, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(din + 2 * (size_t)n, sample_index, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    LAUNCH("sampler probe", k_sampler_probe, gridFor(n), ctx->svHost, n, din, din + n, din + 2 * (size_t)n, start_dim, mode, dout);
    HIPCHK(hipMemcpyAsync(out, dout, (size_t)n * ndims * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(din));
    HIPCHK(hipFree(dout));
    return 0;
}

__global__ void k_libm_probe(int fn, int n, const float *in, float *out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float r;
    if (fn == 9) r = wf::atan2(in[2 * i], in[2 * i + 1]);
    else {
        float x = in[i];
        switch (fn) {
        case 0: r = wf::sin(x); break;
        case 1: r = wf::cos(x); break;
        case 2: r = wf::exp(x); break;
        case 3: r = wf::log(x); break;
        case 4: r = wf::atan(x); break;
        case 5: r = wf::asin(x); break;
        case 6: r = wf::acos(x); break;
        case 7: r = wf::cosh(x); break;
        case 10: r = wf::sinh(x); break;
        case 11: r = wf::tan(x); break;
        default: r = wf::atanh(x); break;
        }
    }
    out[i] = r;
    }
}
int wf_libm_probe(wf_ctx *ctx, int fn, int n, const float *in, float *out) {
    if (!ctx) return fail(-1, "null context");
    if (fn < 0 || fn > 11) return fail(-1, "wf_libm_probe: unknown function %d", fn);
    if (n <= 0) return 0;
    size_t nin = (size_t)n * (fn == 9 ? 2 : 1);
    float *din = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc((void **)&din, nin * sizeof(float)));
    HIPCHK(hipMalloc((void **)&dout, (size_t)n * sizeof(float)));
    HIPCHK(hipMemcpyAsync(din, in, nin * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    LAUNCH("libm probe", k_libm_probe, gridFor(n), fn, n, din, dout);
    HIPCHK(hipMemcpyAsync(out, dout, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(din));
    HIPCHK(hipFree(dout));
    return 0;
}

__global__ void k_kat_probe(int n, const uint64_t *in, uint64_t *out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) wf::KatRun(in + 16 * (size_t)i, out + 8 * (size_t)i);
}
int wf_kat_probe(wf_ctx *ctx, int n, const uint64_t *in, uint64_t *out) {
    if (!ctx) return fail(-1, "null context");
    if (n <= 0) return 0;
    uint64_t *din = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc((void **)&din, (size_t)n * 16 * sizeof(uint64_t)));
    HIPCHK(hipMalloc((void **)&dout, (size_t)n * 8 * sizeof(uint64_t)));
    HIPCHK(hipMemcpyAsync(din, in, (size_t)n * 16 * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    LAUNCH("known-answer probe", k_kat_probe, gridFor(n), n, din, dout);
    HIPCHK(hipMemcpyAsync(out, dout, (size_t)n * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipFree(din));
    HIPCHK(hipFree(dout));
    return 0;
}

int wf_queue_size(wf_ctx *ctx, const char *queue, int *size) {
    if (int e = checkReady(ctx)) return e;
    static const std::map<std::string, int> idx = {{"ray0", CNT_RAY0}, {"ray1", CNT_RAY1}, {"escaped", CNT_ESCAPED}, {"hitlight", CNT_HITLIGHT},
                                                   {"shadow", CNT_SHADOW}, {"retrace", CNT_RETRACE}, {"mat_diffuse", CNT_MAT0 + WF_MAT_DIFFUSE},
                                                   {"mat_conductor", CNT_MAT0 + WF_MAT_CONDUCTOR}, {"mat_dielectric", CNT_MAT0 + WF_MAT_DIELECTRIC},
                                                   {"mat_thindielectric", CNT_MAT0 + WF_MAT_THIN_DIELECTRIC},
                                                   {"mat_diffusetransmission", CNT_MAT0 + WF_MAT_DIFFUSE_TRANSMISSION},
                                                   {"mat_coateddiffuse", CNT_MAT0 + WF_MAT_COATED_DIFFUSE},
                                                   {"mat_coatedconductor", CNT_MAT0 + WF_MAT_COATED_CONDUCTOR}};
    auto it = idx.find(queue ? queue : "");
    if (it == idx.end()) return fail(-1, "unknown queue \"%s\"", queue ? queue : "(null)");
    HIPCHK(hipMemcpyAsync(size, ctx->ws.counters + it->second * CNT_STRIDE, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int wf_queue_download(wf_ctx *ctx, const char *queue, const char *member, void *dst, uint64_t nbytes) {
    if (int e = checkReady(ctx)) return e;
    const WorkState &ws = ctx->ws;
    std::string q = queue ? queue : "", m = member ? member : "";
    const void *src = nullptr;
    auto rayMember = [&](const RayQueueV &r) -> const void * {
        if (m == "o") return r.o; if (m == "d") return r.d; if (m == "beta") return r.beta; if (m == "r_u") return r.r_u;
        if (m == "r_l") return r.r_l; if (m == "ctx0") return r.ctx0; if (m == "ctx1") return r.ctx1; if (m == "ctx2") return r.ctx2;
        if (m == "meta") return r.meta;
        return nullptr;
    };
    if (q == "ray0") src = rayMember(ws.rq[0]);
    else if (q == "ray1") src = rayMember(ws.rq[1]);
    else if (q == "pixel") {
        if (m == "L") src = ws.L; else if (m == "lambda") src = ws.lambda; else if (m == "lambda_pdf") src = ws.lambdaPdf;
        else if (m == "pPixel") src = ws.pPixel; else if (m == "filterWeight") src = ws.filterWeight;
        else if (m == "cameraRayWeight") src = ws.cameraRayWeight; else if (m == "samples0") src = ws.samples0; else if (m == "samples1") src = ws.samples1;
    } else if (q == "hit") src = ws.hit;
    else if (q == "shadow") {
        if (m == "o") src = ws.sq.o; else if (m == "d") src = ws.sq.d; else if (m == "Ld") src = ws.sq.Ld;
        else if (m == "r_u") src = ws.sq.r_u; else if (m == "r_l") src = ws.sq.r_l;
    }
    if (!src) return fail(-1, "unknown queue member \"%s\".\"%s\"", q.c_str(), m.c_str());
    HIPCHK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
