// wf_mat.hip — one translation unit per material type, stage half and variant (compiled with
// -DWF_MAT_INSTANCE=<wf_material_type> -DWF_MAT_PART=<0|1|2> -DWF_MAT_TEXCTX=<variant>):
// the K9 kernels "<Material> + BxDF eval" (EvaluateMaterialAndBSDF<M, BasicTextureEvaluator>, wavefront/surfscatter.cpp:57-328) and
// their launchers.  Split from wf_backend.hip so that the material kernels compile in parallel (the layered ones take minutes).
//   WF_MAT_PART 1  k_mat_shade<M, V>   interaction + textures + BxDF + BSDF sample + indirect-ray push; leaves the item's NeeItem
//                                      (wf_kernels.h) in HBM                                  V = 0 | 1 | 2 (texture context / GBuffer)
//   WF_MAT_PART 2  k_mat_nee<M, R>     next-event estimation from the NeeItem                 R = 0 | 1 (rare light types reachable)
//   WF_MAT_PART 0  k_eval_material<M, V>  both halves in one kernel (round 1-4; WF_MAT_SPLIT=0 selects it at run time for A/B runs;
//                                      built only with `make MATFUSED=1`)
#include <hip/hip_runtime.h>

#if !defined(WF_MAT_INSTANCE) || !defined(WF_MAT_TEXCTX) || !defined(WF_MAT_PART)
#error "compile with -DWF_MAT_INSTANCE=<wf_material_type> -DWF_MAT_PART=<0|1|2> -DWF_MAT_TEXCTX=<variant>"
#endif
// the LEAN shade variants (0, 1): device code without the quadric / patch / curve interaction callees and without the texture-graph
// evaluator (wf_scene.h "LEAN DEVICE VARIANTS") — must be defined before the common headers are read
#if WF_MAT_PART == 1 && WF_MAT_TEXCTX <= 1
#define WF_LEAN 1
#endif

#include "../common/wf_kernels.h"

using namespace wf;

constexpr int MBLOCK = 256;

// minimum waves per SIMD (the register budget: 2 -> 256 VGPRs, 3 -> 168, 4 -> 128).  The fused kernel, measured per material on the spec
// scene, 2 vs 3 waves: diffuse 24.1 -> 22.7 ms per 16 spp, conductor 6.27 -> 6.40, coated diffuse 17.4 -> 20.2 (its stochastic walks spill).
// Rounds 3-4 ran the diffuse kernel at 3 waves for those 6 %; round 4 took it back: at 168 VGPRs the unit spills ~170 VGPRs beside ~300 SGPRs
// that the compiler spills THROUGH VGPR lanes, and that build was the common factor of three wrong-code incidents that no source change
// explains; tools/check_spill_carriers.py (CPU suite) fails any build in which a kernel spills such a carrier register.
#ifndef WF_MAT_WAVES
#define WF_MAT_WAVES 2
#endif
#ifndef WF_SHADE_WAVES
#define WF_SHADE_WAVES 2
#endif
// the lean shade kernels (profiles/r05_lean_variants_ab_sm16.txt; spec scene, 16 spp, same box; general kernel at 2 waves -> lean at 3 / 4 / 5):
//   diffuse 10.87 -> 9.26 / 9.03 / 13.96 ms   conductor 3.19 -> 3.04 / 3.31 / 4.73   coated diffuse 7.43 -> 6.55 / 8.14 / 10.48
// per-type overrides for A/B builds (tools/build_variant.sh ... "-DWF_SHADE_LEAN_W1=3 -DWF_NEE_W6=2")
#if WF_MAT_INSTANCE == 1 && defined(WF_SHADE_LEAN_W1)
#define WF_SHADE_WAVES_LEAN WF_SHADE_LEAN_W1
#elif WF_MAT_INSTANCE == 2 && defined(WF_SHADE_LEAN_W2)
#define WF_SHADE_WAVES_LEAN WF_SHADE_LEAN_W2
#elif WF_MAT_INSTANCE == 6 && defined(WF_SHADE_LEAN_W6)
#define WF_SHADE_WAVES_LEAN WF_SHADE_LEAN_W6
#endif
#if WF_MAT_INSTANCE == 6 && defined(WF_NEE_W6)
#define WF_NEE_WAVES WF_NEE_W6
#elif WF_MAT_INSTANCE == 2 && defined(WF_NEE_W2)
#define WF_NEE_WAVES WF_NEE_W2
#endif
// (round 6: the diffuse kernel too runs at 3 waves — 167 VGPRs, NOTHING spilled, 24 scratch stores in the whole kernel — instead of 4
//  (128 VGPRs, 44 spilled): 7.39 against 7.34 ms per 16 spp, profiles/r06_material_occupancy_ab2_sm16.txt: the same time without the
//  spills' HBM traffic)
#ifndef WF_SHADE_WAVES_LEAN
#define WF_SHADE_WAVES_LEAN 3
#endif
// Round 5, the two halves on the spec scene (16 spp, same box; profiles/r05_material_split_ab_sm16.txt, r05_material_occupancy_ab_sm16.txt):
//   next-event estimation, diffuse / conductor / coated diffuse, ms:  2 waves 12.75 / 2.89 / 9.89   3 waves 9.95 / 2.33 / 9.08
//                                                                      4 waves  8.47 / 2.09 / 8.80   5 waves 8.99 / 2.13 / 11.09   6 waves 9.34 / 2.50 / 12.38
//   shade:                                                             2 waves 10.88 / 3.17 / 7.44   3 waves 11.27 / 4.47 / 8.49   4 waves 12.24 / 4.81 / 9.00
// The light sampling (descent of the light BVH, shape sampling) is a chain of dependent gathers and pays for occupancy even with 100-600
// spilled VGPRs; the shade half is dominated by its own loads and stores, and spills only add to them.
#ifndef WF_NEE_WAVES
#if (WF_MAT_INSTANCE == 10 || WF_MAT_INSTANCE == 7 || WF_MAT_INSTANCE == 9) && WF_MAT_PART == 2 && WF_MAT_TEXCTX == 1
#define WF_NEE_WAVES 2   // (k_mat_nee<measured, rare lights> — and, since round 6's inlined item I/O, <coated conductor, rare lights>, since its global table loads <type 9, rare lights> — at 4 waves spill a carrier register: tools/check_spill_carriers.py)
#else
#define WF_NEE_WAVES 4
#endif
#endif
// The scene view is read through the pointer to its device-resident copy instead of from the kernel arguments (round 4) — every field that
// is live across the kernel is an SGPR pair either way, but a by-value argument invites the compiler to keep all of them (300-400 spilled
// SGPRs per material kernel; spec scene, 16 spp, same box: diffuse 21.4 -> 20.6, conductor 5.25 -> 4.98, coated diffuse 16.0 -> 15.7 ms,
// profiles/r04_material_sv_pointer_ab_sm16.txt)

// ---- the NeeItem in HBM: planes of 16 bytes per item (a wave reads / writes 1 KiB per instruction), plane p of item k at
// neeRec[p * maxQueueSize + k]; k = the item's index in its type's queue + the item counts of the types before it (the queues of one depth
// hold at most maxQueueSize items together: one per ray)
template <int MAT>
struct NeeIO {
    using BxDF = typename MatBxDF<MAT>::T;
    static constexpr int NBX = (sizeof(BxDF) + 15) / 16;
    struct Packed { F4 v[NBX]; };
    __device__ __attribute__((always_inline)) static int Base(const WorkState &ws) {
        int base = 0;
#pragma unroll
        for (int t = 1; t < MAT; ++t) base += ws.counters[(CNT_MAT0 + t) * CNT_STRIDE];
        return base;
    }
    // planes: 0 pi.lo, pi.hi.x | 1 pi.hi.yz, n.xy | 2 n.z, ns | 3 dpdus, wo.x | 4 wo.yz, u0, u.x | 5 u.y, pixelIndex (-1: no next-event
    // estimation), mediumInside, mediumOutside | 6 beta | 7 r_u | 8 lambda | 9 ctxP, ctxIsPoint | 10.. the BxDF
    static constexpr int NFIX = 10;
    // (always_inline: out of line — the conductor and the layered types' units until round 6 — the whole item crossed scratch: written by
    //  the shade code, read back here, 200-270 B each way per item, and the same again on the next-event side)
    __device__ __attribute__((always_inline)) static void Store(const WorkState &ws, int k, const NeeItem<MAT> &it) {
        const size_t S = (size_t)ws.maxQueueSize;
        F4 *r = ws.neeRec + k;
        r[5 * S] = F4{it.u.y, BitsToFloat((uint32_t)(it.want ? it.pixelIndex : -1)), BitsToFloat((uint32_t)it.mediumInside), BitsToFloat((uint32_t)it.mediumOutside)};
        if (!it.want) return;
        r[0] = F4{it.pi.lo.x, it.pi.lo.y, it.pi.lo.z, it.pi.hi.x};
        r[1 * S] = F4{it.pi.hi.y, it.pi.hi.z, it.n.x, it.n.y};
        r[2 * S] = F4{it.n.z, it.ns.x, it.ns.y, it.ns.z};
        r[3 * S] = F4{it.dpdus.x, it.dpdus.y, it.dpdus.z, it.wo.x};
        r[4 * S] = F4{it.wo.y, it.wo.z, it.u0, it.u.x};
        r[6 * S] = toF4(it.beta);
        r[7 * S] = toF4(it.r_u);
        r[8 * S] = F4{it.lambda[0], it.lambda[1], it.lambda[2], it.lambda[3]};
        r[9 * S] = F4{it.ctxP.x, it.ctxP.y, it.ctxP.z, it.ctxIsPoint ? 1.f : 0.f};
        Packed pk{};
        __builtin_memcpy(&pk, &it.bxdf, sizeof(BxDF));
#pragma unroll
        for (int b = 0; b < NBX; ++b) r[(NFIX + b) * S] = pk.v[b];
    }
    // what the light sample needs (MatNeeRequest of the stored item) ...
    __device__ __attribute__((always_inline)) static void LoadRequest(const WorkState &ws, int k, NeeRequest *rq) {
        const size_t S = (size_t)ws.maxQueueSize;
        const F4 *r = ws.neeRec + k;
        const F4 p5 = r[5 * S];
        rq->want = (int)FloatToBits(p5.y) >= 0;
        if (!rq->want) return;
        const F4 p1 = r[1 * S], p2 = r[2 * S], p4 = r[4 * S], p8 = r[8 * S], p9 = r[9 * S];
        if (p9.w != 0.f) rq->ctx.pi = MakeP3i(V3{p9.x, p9.y, p9.z});
        else { const F4 p0 = r[0]; rq->ctx.pi.lo = V3{p0.x, p0.y, p0.z}; rq->ctx.pi.hi = V3{p0.w, p1.x, p1.y}; }
        rq->ctx.n = N3{p1.z, p1.w, p2.x}; rq->ctx.ns = N3{p2.y, p2.z, p2.w};
        rq->u0 = p4.z; rq->u = V2{p4.w, p5.x};
        rq->lambda.lambda[0] = p8.x; rq->lambda.lambda[1] = p8.y; rq->lambda.lambda[2] = p8.z; rq->lambda.lambda[3] = p8.w;
        rq->lambda.pdf[0] = rq->lambda.pdf[1] = rq->lambda.pdf[2] = rq->lambda.pdf[3] = 0;
    }
    // ... and what MatNeeFinish reads of it
    __device__ __attribute__((always_inline)) static void Load(const WorkState &ws, int k, NeeItem<MAT> *it) {
        const size_t S = (size_t)ws.maxQueueSize;
        const F4 *r = ws.neeRec + k;
        const F4 p5 = r[5 * S];
        const int pix = (int)FloatToBits(p5.y);
        it->want = pix >= 0;
        if (!it->want) return;
        const F4 p0 = r[0], p1 = r[1 * S], p2 = r[2 * S], p3 = r[3 * S], p4 = r[4 * S], p6 = r[6 * S], p7 = r[7 * S];
        it->pi.lo = V3{p0.x, p0.y, p0.z}; it->pi.hi = V3{p0.w, p1.x, p1.y};
        it->n = N3{p1.z, p1.w, p2.x}; it->ns = N3{p2.y, p2.z, p2.w};
        it->dpdus = V3{p3.x, p3.y, p3.z}; it->wo = V3{p3.w, p4.x, p4.y};
        it->pixelIndex = pix; it->mediumInside = (int)FloatToBits(p5.z); it->mediumOutside = (int)FloatToBits(p5.w);
        it->beta = toS4(p6); it->r_u = toS4(p7);
        Packed pk;
#pragma unroll
        for (int b = 0; b < NBX; ++b) pk.v[b] = r[(NFIX + b) * S];
        __builtin_memcpy(&it->bxdf, &pk, sizeof(BxDF));
    }
};

#define WF_CAT4_(a, b, c, d) a##b##c##d
#define WF_CAT4(a, b, c, d) WF_CAT4_(a, b, c, d)

#if WF_MAT_PART == 0
template <int MAT, int TEXCTX>
__global__ void __launch_bounds__(MBLOCK, WF_MAT_WAVES) k_eval_material(const SceneView *__restrict__ svp, WorkState ws, int cur) {
    const SceneView &sv = *svp;
    const int n = ws.counters[(CNT_MAT0 + MAT) * CNT_STRIDE];
    // block-uniform trip count: BlockAlloc inside the body synchronises the workgroup
    for (int base = blockIdx.x * MBLOCK; base < n; base += gridDim.x * MBLOCK) {
        const int i = base + threadIdx.x;
        KEvalMaterial<MAT, TEXCTX>(sv, ws, cur, i, i < n);
    }
}
// WF_MAT_TEXCTX = 1: some texture depends on the footprint, or some material has a displacement texture / normal map;
// 2: the same plus the rarely used light types (KEvalMaterial)
extern "C" void WF_CAT4(wf_launch_eval_material_, WF_MAT_INSTANCE, _, WF_MAT_TEXCTX)(hipStream_t stream, int grid, const SceneView *sv, const WorkState *ws, int cur) {
    hipLaunchKernelGGL((k_eval_material<WF_MAT_INSTANCE, WF_MAT_TEXCTX>), dim3(grid), dim3(MBLOCK), 0, stream, sv->self, *ws, cur);
}
#elif WF_MAT_PART == 1
// VARIANT (WF_MAT_TEXCTX): 0 lean, no texture needs the footprint and nothing is displaced; 1 lean, with the footprint / bump block;
// 2 general (quadrics, patches, curves, texture graphs); 3 = 2 + the GBufferFilm's visible surface and the moving camera's differentials
template <int MAT, int VARIANT>
__global__ void __launch_bounds__(MBLOCK, VARIANT <= 1 ? WF_SHADE_WAVES_LEAN : WF_SHADE_WAVES) k_mat_shade(const SceneView *__restrict__ svp, WorkState ws, int cur) {
    const SceneView &sv = *svp;
    const int n = ws.counters[(CNT_MAT0 + MAT) * CNT_STRIDE];
    const int rec0 = NeeIO<MAT>::Base(ws);
    for (int base = blockIdx.x * MBLOCK; base < n; base += gridDim.x * MBLOCK) {
        const int i = base + threadIdx.x;
        NeeItem<MAT> it;
        MatShade<MAT, (VARIANT == 0 ? 0 : (VARIANT == 3 ? 2 : 1))>(sv, ws, cur, i, i < n, &it);
        if (i < n) NeeIO<MAT>::Store(ws, rec0 + i, it);
    }
}
extern "C" void WF_CAT4(wf_launch_mat_shade_, WF_MAT_INSTANCE, _, WF_MAT_TEXCTX)(hipStream_t stream, int grid, const SceneView *sv, const WorkState *ws, int cur) {
    hipLaunchKernelGGL((k_mat_shade<WF_MAT_INSTANCE, WF_MAT_TEXCTX>), dim3(grid), dim3(MBLOCK), 0, stream, sv->self, *ws, cur);
}
#else
template <int MAT, bool RARE>
__global__ void __launch_bounds__(MBLOCK, WF_NEE_WAVES) k_mat_nee(const SceneView *__restrict__ svp, WorkState ws) {
    const SceneView &sv = *svp;
    const int n = ws.counters[(CNT_MAT0 + MAT) * CNT_STRIDE];
    const int rec0 = NeeIO<MAT>::Base(ws);
    for (int base = blockIdx.x * MBLOCK; base < n; base += gridDim.x * MBLOCK) {
        const int i = base + threadIdx.x;
        // the request, the workgroup's light samples, THEN the rest of the item: nothing of it is live while this lane serves another
        // lane's request (light-BVH descent, shape sampling: the register-hungry part)
        NeeRequest rq;
        if (i < n) NeeIO<MAT>::LoadRequest(ws, rec0 + i, &rq);
        const LightPick pick = SampleLightForBlock<RARE>(sv, rq.want, rq.ctx, rq.u0, rq.u, rq.lambda);
        NeeItem<MAT> it;
        if (i < n && pick.lightId >= 0) NeeIO<MAT>::Load(ws, rec0 + i, &it);
        MatNeeFinish(sv, ws, it, pick, i < n);
    }
}
// WF_MAT_TEXCTX = 0: the common light types; 1: + the rarely used ones (the portal infinite light's out-of-line samplers)
extern "C" void WF_CAT4(wf_launch_mat_nee_, WF_MAT_INSTANCE, _, WF_MAT_TEXCTX)(hipStream_t stream, int grid, const SceneView *sv, const WorkState *ws) {
    hipLaunchKernelGGL((k_mat_nee<WF_MAT_INSTANCE, (WF_MAT_TEXCTX != 0)>), dim3(grid), dim3(MBLOCK), 0, stream, sv->self, *ws);
}
#endif
