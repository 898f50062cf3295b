// wf_traverse.h — the production BVH traversal for CDNA4 (device only; included by wf_backend.hip).
//
// The reference-order walk in common/wf_shapes.h (BVHIntersectClosest/Any) visits one 32-byte
// LinearBVHNode per step and chases three levels of indirection per triangle (bvh_prims -> tri_indices
// -> P); it stays as the *counting* variant, because SURVEY.md §8(d) defines the roofline's algorithmic
// bytes on exactly those visit counts.  The kernels that render use the layout and loop below instead.
//
// What bounds traversal on MI355X (profiles/r01_*pmc*): not HBM.  With launches of millions of rays the walk is
// bounded by VALU issue (56-75 % busy at ~30 % active lanes) and by the latency of the dependent node fetches, so
// the design minimises instructions and fetches per visited node:
//
//  * QNode (64 B, four dwordx4; WF_BVH4): an interior node carries FOUR children's bounds (the reference's binary tree with every
//    second level collapsed), quantised to 16 bits per plane on the tree's grid, with build-time outward margins.  One fetch
//    feeds four slab tests and halves the dependent fetches per ray (the two-child 32-byte layout of round 1 remains as WF_BVH4=0).
//  * the slab test runs in grid coordinates: WalkInit folds the grid (base, cell), the ray (o, 1/d), the
//    reference's (1 + 2 gamma(3)) factor and an evaluation-error slack into per-ray constants, so a plane costs one
//    v_cvt (SDWA half-word select) and half a v_pk_fma; near/far planes are swapped per ray with v_perm; min/max
//    are v_max3/v_min3.  No branches inside the step.  The test is a superset of Bounds3::IntersectP on the exact
//    box (it passes whenever the reference's test passes); WHICH triangle is hit is decided only by the exact,
//    float triangle test, so results are the reference's.
//  * the top TOP_NODES nodes of the tree (breadth-first numbering, ~9 levels) are copied into LDS by each
//    workgroup at kernel start: most of a ray's interior visits are served by ds_read_b128.
//  * LeafTri (48 B, three dwordx4): the three vertices of each triangle in BVH leaf order — the
//    "3 indices + 3 Point3f" of the §8(d) formula as one contiguous record, no index chase — plus the triangle's
//    routing code (material type / emissive / interface), so the end-of-batch routing gathers nothing.
//  * children are visited nearest-entry first; the node stack lives in LDS, one column per lane.
//  * "while-while": all lanes of a wave descend interior nodes until every lane sits at a leaf, leaves are
//    processed together, a workgroup finishes a batch of TBLOCK rays together (block-aggregated queue pushes).
//
// Near-ties in t (coplanar overlapping geometry, e.g. a glass box standing on the floor seen from inside) are the one
// case where the visiting order decides the result: IntersectTriangle (shapes.cpp:234-237) accepts a triangle while
// tScaled <= fl(tMax * det), so among candidates whose t agree to ~3 * 2^-24 the reference keeps the LAST one its own
// depth-first order accepts.  This walk visits nearest-entry first, so it cannot know; instead it tests and prunes
// against tMax * (1 + 2^-20) + 2^-20 of the scene extent (one fma), and a candidate that lands inside that band of the
// current best marks the ray (sign bit of RayWalk::tMax).  Marked rays are re-traced by the reference-order walk
// (k_closest_retrace / BVHIntersectClosest).  The band has to cover more than the acceptance test's own rounding: the
// computed t of two coplanar triangles differ by their evaluation errors (absolute: a few ulps of the vertex
// coordinates, not of t), and the reference prunes subtrees with the EXACT tMax against box entries that carry errors
// of the same size — a flat leaf box coplanar with the current hit is skipped or not by a hair.  Size of the band: inside a
// triangle t = sum b_i z_i with b_i in [0, 1], so its rounding error is a few ulps of max |z_i|, the sheared depth of the
// farthest vertex, which is at most sqrt(3) x (scene extent + |o|) for any triangle; 2^-20 = 16 ulps of t plus 16 ulps of the
// scene extent covers that with a margin (measured on the coplanar floor / glass-box golden: ~1e-5 at extent 16).  A wider band
// only re-traces more rays (2^-16 re-traced 1 ray in 40 on the 100-unit san-miguel-like scene: every blob resting on the ground).
// (Measured and dropped: deciding marks by the triangles' own deltaT bounds of shapes.cpp:252-266 in a slow branch —
// sound by pbrt's error analysis, but deltaT grows as distance^2 / triangle size, which forces a 2^-8 pruning band:
// +24 % closest-hit time on the bench scene.)
//
// Object instances (two-level BVH): an instance definition's tree lives in the same QNode / LeafTri arrays with its own
// quantisation grid (FastDef); a top-level leaf entry marked c.z == 4 is an instance.  The walk pushes it as a leaf
// reference with first >= INST_FIRST, and when that reference is popped it switches the lane to the instance's space
// (InstanceRay = the reference's Transform::ApplyInverse(Ray, &tMax), per-ray constants recomputed for the definition's
// grid), with the world tMax and a NODE_EXIT marker underneath on the stack; popping the marker switches back.
#pragma once

namespace wf {

#ifndef WF_BVH4
#define WF_BVH4 1   // four children per node (64-byte nodes, half the dependent fetches per ray: measured -10 % closest-hit and
                    // shadow time on both the 30 k-triangle and the 10 M-triangle instanced scene); 0 = the two-child layout
#endif
#if WF_BVH4
struct alignas(16) QNode {
    // q[3c + a]: child c, axis a: min plane (low half) | max plane << 16 on the 16-bit grid of the tree (plane = base + q * cell,
    // with build-time outward margins); an empty slot has min > max on every axis (never hit)
    uint32_t q[12];
    int32_t child[4];  // >= 0: interior QNode index; < 0: leaf ~((first << 4) | (count - 1))
};
constexpr int QNODE_U4 = 4;
#else
struct alignas(16) QNode {
    // q[0..2]: left child x, y, z;  q[3..5]: right child x, y, z.  Each dword = min plane (low half) | max plane << 16
    // on the 16-bit grid of FastBVH (plane = base + q * cell, with build-time outward margins)
    uint32_t q[6];
    int32_t left, right;  // >= 0: interior QNode index; < 0: leaf ~((first << 4) | (count - 1))
};
constexpr int QNODE_U4 = 2;
#endif
struct alignas(16) LeafTri {
    F4 a;  // p0.xyz, p1.x
    F4 b;  // p1.yz, p2.xy
    F4 c;  // p2.z, triangle id (int bits), 1.0 if the triangle is degenerate (zero-length normal) else 0, routing code (int bits)
};
struct alignas(16) U4 { uint32_t x, y, z, w; };

constexpr int NODE_NONE = (int)0x80000000;
#ifndef WF_TOP_NODES
// QNodes of the tree's top copied into LDS by every workgroup.  Rounds 1-5: 128 (WF_BVH4; 256: any-hit -9 % in round 3) — the walk then fetched a node
// through a pointer that is either LDS or global memory, which the compiler can only issue as FLAT loads.  Round 6: 0 — no copy, every
// node fetch is a plain global load (the top of the tree lives in the vector L1 / L2 anyway): closest-hit 31.3 -> 28.4 ms, any-hit
// 13.5 -> 12.5 ms per 16 spp on the spec scene, same box (profiles/r06_tree_top_global_loads_and_knobs_ab_sm16.txt), and 8 KiB of LDS less per workgroup.
#define WF_TOP_NODES 0
#endif
#ifndef WF_TBLOCK
#define WF_TBLOCK 256
#endif
#ifndef WF_TSTACK
#define WF_TSTACK 16   // a power of two: the LDS part of the stack is a ring (LdsStackT)
#endif
#ifndef WF_TWAVES
#define WF_TWAVES 5   // __launch_bounds__ second argument (minimum waves per SIMD) of the traversal kernels
#endif
#ifndef WF_TWAVES_CLOSEST
#define WF_TWAVES_CLOSEST WF_TWAVES   // the one-level closest-hit walk (it carries the hit record and the near-tie code)
#endif
#ifndef WF_TWAVES_INST
#define WF_TWAVES_INST 4   // the same for the two-level (object instance) variants, which carry the render-space ray as well
#endif
#ifndef WF_TWAVES_INST_SHADOW
#define WF_TWAVES_INST_SHADOW 4   // round 3: 5 waves (96 VGPRs, 28 spilled) beat 4 by 3 %; round 4, after the zero-initialised locals: 4 waves (122 VGPRs, NOTHING spilled) 17.4 vs 18.5 ms per 16 spp (gpurun_out/r04aa)
#endif
constexpr int TOP_NODES = WF_TOP_NODES;  // QNodes cached in LDS per workgroup
constexpr int TBLOCK = WF_TBLOCK;        // threads per workgroup of the traversal kernels
constexpr int TSTACK = WF_TSTACK;        // LDS stack entries per lane (x 4 B x TBLOCK)

struct FastBVH {
    const QNode *nodes;
    const LeafTri *tris;
    int nNodes;
    float base[3], cell[3];  // grid: plane(q) = base + q * cell (real arithmetic; the builder keeps a margin, see BuildFastBVH)
    float absBand;           // 2^-20 x the scene extent: absolute part of the near-tie band
    // relative part of the band (1 + 2^-20).  Scenes with quadrics / patches / curves use 2^-10 for both parts (round 4): those shapes
    // accept a hit when the UPPER BOUND of its interval-arithmetic t is <= tMax (shapes.h:147-233: Sphere::BasicIntersect and
    // friends), so two candidates closer than the interval's width — 10^-5 t and more, far outside the triangles' 2^-20 — are
    // decided by the visiting order: of two coincident cylinders the reference keeps the FIRST, this walk kept whichever its own
    // order met first and never marked the ray (fuzz finding s200010 on the GPU: 7 % of the pixels)
    float tieRel;
    // ... but only a pair of candidates that INVOLVES such a shape needs the wide band (round 4, second step): two triangles are ordered
    // by their exact t on both sides, 2^-20 decides.  With the wide band applied to every pair the 10 M-triangle scene plus ONE sphere
    // marked so many rays that the re-trace launches took 162 ms of a 358 ms frame.  tieRelTri / absBandTri: the triangles' own band;
    // firstGeneral: the first primitive id that is not a triangle (INT_MAX without such shapes; then both bands are the same).
    float tieRelTri, absBandTri;
    int firstGeneral;
    const struct FastDef *defs;       // per instance definition (scenes with object instances)
    const struct SubEntry *subs;      // the instance entries of the top-level tree (round 6: partial re-braiding, see SubEntry)
    const wf_instance *instances;
    const SceneView *sv;              // device-resident copy of the scene view, for the out-of-line general-primitive callbacks
};
struct FastDef {
    int root;                // QNode index of the definition's root
    float base[3], cell[3];  // its quantisation grid
    int pad;
};
// An ENTRY of the top-level tree into an object instance (round 6).  Until round 5 an instance was one leaf entry of the top-level tree,
// bounded by one box, and its walk started at the definition's root: on the spec scene a ray entered eight instances and six of the
// visits ended without a primitive test — the box of a cluster of objects is mostly empty, and every visit pays the reference's
// interval-arithmetic ray transform twice (in and out).  PARTIAL RE-BRAIDING (Benthin, Woop, Wald, Afra: "Improved two-level BVHs using
// partial re-braiding", HPG 2017): the top-level tree is built over the instances OPENED a few levels into their definitions' trees —
// an entry is (instance, node of the definition's production tree), bounded by the box of that subtree's transformed vertices — so
// the top-level tree separates the objects of a cluster, and a ray changes spaces only for subtrees whose own box it meets.  The walk
// inside is the same; only its starting node differs.  Which triangles can be reached is a superset of the reference's hits as
// before (the exact triangle test decides, near ties are re-walked in the reference's order).
struct SubEntry {
    int inst;   // wf_instance index
    int node;   // where the walk starts in the definition's production tree: QNode index (>= 0) or a leaf reference (< 0)
};
constexpr int INST_FIRST = 1 << 26;          // leaf references with first >= INST_FIRST: instance entry (first - INST_FIRST) of FastBVH::subs
constexpr int NODE_EXIT = (int)0x80000001;   // stack marker: leave the instance (the world tMax is the entry below it)
constexpr int INST_STALE = 1 << 30;          // RayWalk::inst flag while its instance is being re-visited (EnterInstance)

typedef float f2 __attribute__((ext_vector_type(2)));


#ifndef WF_LAZY_INST
#define WF_LAZY_INST 0   // (measured and dropped, see below)
#endif
struct RayWalk {
    V3 o;
    RayShear sh;   // per-ray part of the triangle test (MakeRayShear)
    float tMax;    // |tMax| = nearest hit so far; sign bit set = a near-tie was seen (see above): re-trace in reference order
    // slab test in grid coordinates: entry t of an axis = fma(qNear, a, bn), exit t = fma(qFar, af, bf)
    V3 a, bn, af, bf;
    uint32_t selx, sely, selz;  // v_perm selectors: put the near plane in the low half, the far plane in the high half
    // (round 4: deriving them from the sign of `a` at every node instead frees three registers on paper; the two-level closest-hit kernel
    //  still spills the same 192 VGPRs at compile time — not pursued)
    int node;  // current ref; NODE_NONE = finished
    int prim;
    uint32_t route;  // routing code of the hit triangle (LeafTri.c.w)
    float b0, b1, b2;
    int inst;        // instance the hit primitive was reached through (-1: top level); only the INST kernel variants use it
    int curInst;     // instance whose definition is being walked (-1: top level)
    // (round 4, WF_LAZY_INST) what of the per-ray state is still owed for the space being walked: 0 nothing; 1 inside an instance whose
    // ray so far is the reference's transform WITHOUT its interval-arithmetic origin shift — good for the conservative box tests, not
    // for a primitive test: o, the shear and tMax are made exact (WalkMakeExact) when a leaf holds one; 2 back at the top level with
    // the shear of the render-space ray not recomputed yet
#if WF_LAZY_INST
    int lazy;
#endif
};
#if WF_LAZY_INST
#define WF_LAZY_GET(w) ((w).lazy)
#define WF_LAZY_SET(w, v) ((w).lazy = (v))
#else
#define WF_LAZY_GET(w) 0          // (the field costs a register and a test per leaf primitive: compiled out with the experiment)
#define WF_LAZY_SET(w, v) ((void)0)
#endif

// Per-ray constants of the box test.  Bounds3::IntersectP (util/vecmath.h:1574-1608) computes, per axis,
// tNear = (pNear - o) * invDir and tFar = (pFar - o) * invDir * (1 + 2 gamma(3)).  With pNear = base + q * cell
// that is q * (cell * invDir) + (base - o) * invDir: one fma per plane.  The evaluation error of that form is
// bounded by a few ulps of (65535 |a| + |b|); SLACK times that bound is folded into the constants (subtracted on
// the near side, added on the far side) so that the test passes whenever the reference's test on the exact
// box would: a superset of visited nodes, while the hit itself is decided by the exact triangle test.
// the ray-dependent part: origin, shear, slab constants on the grid (base, cell)
#ifndef WF_SLAB_RCP
#define WF_SLAB_RCP 1
#endif
// the slab constants alone (the box tests' part of the per-ray state)
// the slab constants from the per-axis products a = cell / d and bk = (base - o) / d (the part of WalkSetSlab that needs no reciprocal)
__device__ inline void WalkSlabFromAB(RayWalk &w, const float a[3], const float bk[3]) {
    constexpr float SLACK = 0x1p-20f;            // 16 ulp
    constexpr float G = 1 + 2 * gamma(3);        // the reference's tMax factor
    float bn[3], af[3], bf[3];
    uint32_t sel[3];
    for (int k = 0; k < 3; ++k) {
        const float delta = SLACK * fma(65535.f, fabsf(a[k]), fabsf(bk[k]));
        bn[k] = bk[k] - delta;
        af[k] = a[k] * G; bf[k] = fma(bk[k], G, delta * 1.001f);
        sel[k] = (FloatToBits(a[k]) >> 31) ? 0x01000302u : 0x03020100u;  // negative direction (incl. -0: the sign of 1/d, kept by the clamp): swap halves
    }
    w.a = V3{a[0], a[1], a[2]}; w.bn = V3{bn[0], bn[1], bn[2]};
    w.af = V3{af[0], af[1], af[2]}; w.bf = V3{bf[0], bf[1], bf[2]};
    w.selx = sel[0]; w.sely = sel[1]; w.selz = sel[2];
}
__device__ inline void WalkSetSlab(const float base[3], const float cell[3], RayWalk &w, V3 o, V3 d, float *aOut = nullptr, float *bkOut = nullptr) {
    constexpr float INV_MAX = 1e28f;             // |1/d| clamp: keeps every product finite (no 0 * inf NaNs)
    const float dd[3] = {d.x, d.y, d.z}, oo[3] = {o.x, o.y, o.z};
    float a[3], bk[3];
    for (int k = 0; k < 3; ++k) {
        // v_rcp_f32 (1 ulp) instead of the IEEE division (ten instructions): the constants feed the conservative slab test only, and
        // the 16-ulp SLACK covers one more ulp in a and b (the exact triangle test keeps its IEEE divisions, MakeRayShear)
        float inv = WF_SLAB_RCP ? __builtin_amdgcn_rcpf(dd[k]) : 1 / dd[k];
        if (!(fabsf(inv) <= INV_MAX)) inv = copysignf(INV_MAX, dd[k]);
        a[k] = cell[k] * inv; bk[k] = (base[k] - oo[k]) * inv;
        if (aOut) { aOut[k] = a[k]; bkOut[k] = bk[k]; }
    }
    WalkSlabFromAB(w, a, bk);
}
__device__ inline void WalkSetRay(const float base[3], const float cell[3], RayWalk &w, V3 o, V3 d, float *aOut = nullptr, float *bkOut = nullptr) {
    w.o = o;
    w.sh = MakeRayShear(d);
    WalkSetSlab(base, cell, w, o, d, aOut, bkOut);
}
__device__ inline void WalkInit(const FastBVH &bvh, RayWalk &w, V3 o, V3 d, float tMax, float *aOut = nullptr, float *bkOut = nullptr) {
    WalkSetRay(bvh.base, bvh.cell, w, o, d, aOut, bkOut);
    w.tMax = tMax;
    w.node = 0;
    w.prim = -1;
    w.route = 0;
    w.b0 = w.b1 = w.b2 = 0;
    w.inst = -1;
    w.curInst = -1;
    WF_LAZY_SET(w, 0);
}
// Switch the lane into / out of an object instance (see the header comment).  oW, dW: the ray in render space.
// Returns false when the instance is skipped.  (Round 3) The top-level leaf that holds an instance only says that the ray meets the
// instance's RENDER-space box; clusters of rotated instances overlap heavily, so most entries ended at the definition's root node —
// after the reference's full ray transform (interval arithmetic), two WalkSetRay (six divisions), two stack markers and a node fetch.
// A cheap, conservative test comes first: the ray is taken into the instance's space with plain fmas and tested against the
// definition's root box (its quantisation grid, which already lies outside the float box) widened by 2^-13 of the magnitudes
// involved — three orders of magnitude more than the differences between this transform and Transform::ApplyInverse(Ray)
// (a few ulps of the coordinates plus the interval-width shift of the origin, util/transform.h:416-429).  The reference's own
// root test (Bounds3::IntersectP on the exact box with the exactly transformed ray) cannot pass where this one fails, so skipping
// changes no result.  Non-affine matrices (bottom row not 0 0 0 1) are never skipped.
#ifndef WF_LAZY_INST
#define WF_LAZY_INST 0   // measured on the spec scene, 16 spp, same box (gpurun_out/r04e, r04f): closest / any-hit 45.0 / 23.0 ms without, 49.4 / 26.1 with (exactified inside the leaf loop), 52.9 / 27.0 with the exactification parked like a transition: off
#endif
#ifndef WF_INST_PRETEST
#define WF_INST_PRETEST 0   // measured on the spec scene (gpurun_out/r3e_ab_sm16.txt): closest 56.8 vs 56.5 ms, any-hit 22.9 vs 21.7 ms per 16 spp with / without — the entries it saves are too few to pay for the test; off
#endif
__device__ inline float WalkBound(const FastBVH &bvh, float t);
__device__ inline bool InstancePretestMiss(const wf_instance &in, const FastDef &fd, V3 o, V3 d, float tBound) {
    const float(*mi)[4] = in.render_from_instance.mInv;
    if (mi[3][0] != 0 || mi[3][1] != 0 || mi[3][2] != 0 || mi[3][3] != 1) return false;
    const float oI[3] = {fma(mi[0][0], o.x, fma(mi[0][1], o.y, fma(mi[0][2], o.z, mi[0][3]))),
                         fma(mi[1][0], o.x, fma(mi[1][1], o.y, fma(mi[1][2], o.z, mi[1][3]))),
                         fma(mi[2][0], o.x, fma(mi[2][1], o.y, fma(mi[2][2], o.z, mi[2][3])))};
    const float dI[3] = {fma(mi[0][0], d.x, fma(mi[0][1], d.y, mi[0][2] * d.z)), fma(mi[1][0], d.x, fma(mi[1][1], d.y, mi[1][2] * d.z)),
                         fma(mi[2][0], d.x, fma(mi[2][1], d.y, mi[2][2] * d.z))};
    float t0 = 0, t1 = tBound * (1 + 0x1p-13f);
    for (int k = 0; k < 3; ++k) {
        const float lo = fd.base[k], hi = fma(65535.f, fd.cell[k], fd.base[k]);
        const float pad = 0x1p-13f * (__builtin_fabsf(lo) + __builtin_fabsf(hi) + __builtin_fabsf(oI[k]) + (hi - lo));
        const float a = (lo - pad) - oI[k], b = (hi + pad) - oI[k];   // the origin relative to the widened slab
        if (__builtin_fabsf(dI[k]) < 1e-30f) {          // parallel to the slab: inside or outside for good
            if (a > 0 || b < 0) return true;
            continue;
        }
        const float inv = __builtin_amdgcn_rcpf(dI[k]);
        float tn = a * inv, tf = b * inv;
        if (tn > tf) { const float x = tn; tn = tf; tf = x; }
        // rcp and the products are good to a few ulps: widen the interval relatively and absolutely before intersecting
        tn -= 0x1p-13f * __builtin_fabsf(tn);
        tf += 0x1p-13f * __builtin_fabsf(tf);
        t0 = __builtin_fmaxf(t0, tn);
        t1 = __builtin_fminf(t1, tf);
    }
    return !(t0 <= t1);   // (NaNs never skip)
}
// ANIM (round 6): the instance may be an AnimatedPrimitive (cpu/primitive.cpp:132-158) — its transformation is interpolated at the ray's
// `time` (wf_animated.h: the reference's AnimatedTransform::Interpolate restated), as the reference-order walks do (InstanceAt<true>, wf_shapes.h).
// An animated instance is one entry of the top-level tree, bounded by the reference's motion bounds; the walk inside is the static one.
// Only the kernels of scenes that have such primitives are built with ANIM: the interpolation is an out-of-line callee whose registers
// every kernel that can reach it is allocated.
template <bool ANIM = false, typename Stack>
__device__ inline bool EnterInstance(const FastBVH &bvh, RayWalk &w, Stack &st, V3 oW, V3 dW, int entry, float time = 0) {
    const SubEntry se = bvh.subs[entry];
    const int inst = se.inst;
    if (se.node == NODE_NONE) {   // an instance of an empty definition
        WalkSetRay(bvh.base, bvh.cell, w, oW, dW);   // (a fused exit may have left the previous instance's constants: ExitInstance)
        w.node = st.empty() ? NODE_NONE : st.pop();
        return false;
    }
    wf_instance moving;
    const wf_instance &in = InstanceAt<ANIM>(*bvh.sv, bvh.instances[inst], time, &moving);
    const FastDef fd = bvh.defs[in.def];
#if WF_INST_PRETEST
    if (InstancePretestMiss(in, fd, oW, dW, WalkBound(bvh, __builtin_fabsf(w.tMax)))) {
        WalkSetRay(bvh.base, bvh.cell, w, oW, dW);
        w.node = st.empty() ? NODE_NONE : st.pop();
        return false;
    }
#endif
#if WF_LAZY_INST
    {
        // Round 4: the LAZY transition.  Eight instances are entered per ray on the spec scene, and three of four visits end without a
        // single primitive test (3.2 triangle tests per ray in all) — yet every one paid the reference's interval-arithmetic ray
        // transform, the triangle test's shear (two IEEE divisions) and the same again on the way out.  Transform::ApplyInverse(Ray,
        // &tMax) (util/transform.h:416-429) = the plain transform of origin and direction, then the origin moved ALONG the ray by
        // dt (the interval width over |d|) and tMax reduced by dt.  A ray whose origin slides along its own line meets every box at
        // parameters shifted by exactly dt: with the UNSHIFTED origin and the unshifted (render-space) tMax the box tests prune
        // nothing the reference's would keep (entry <= exit and entry <= tMax are the same comparisons, exit >= 0 is weaker by
        // dt).  So the visit starts with the two plain transforms — the reference's own expressions, InstanceRay's first half —
        // and the slab constants; the exact ray, its shear and the shifted tMax follow only if a leaf of the definition holds a
        // primitive to test (WalkMakeExact), which also rebuilds the slab constants from the exact origin.
        const float(*mi)[4] = in.render_from_instance.mInv;
        if (mi[3][0] == 0 && mi[3][1] == 0 && mi[3][2] == 0 && mi[3][3] == 1) {   // (a projective instance matrix divides by w: exact path)
            const V3 oI{(mi[0][0] * oW.x + mi[0][1] * oW.y) + (mi[0][2] * oW.z + mi[0][3]), (mi[1][0] * oW.x + mi[1][1] * oW.y) + (mi[1][2] * oW.z + mi[1][3]),
                        (mi[2][0] * oW.x + mi[2][1] * oW.y) + (mi[2][2] * oW.z + mi[2][3])};
            const V3 dI = XfVector3(mi, dW);
            st.push((int)FloatToBits(w.tMax));
            st.push(NODE_EXIT);
            WalkSetSlab(fd.base, fd.cell, w, oI, dI);
            w.curInst = inst;
            w.node = se.node;
            WF_LAZY_SET(w, 1);
            return true;
        }
    }
#endif
    float tI = __builtin_fabsf(w.tMax);
    V3 oI, dI;
    InstanceRay(in, oW, dW, &tI, &oI, &dI);
    st.push((int)FloatToBits(w.tMax));
    st.push(NODE_EXIT);
    WalkSetRay(fd.base, fd.cell, w, oI, dI);
    w.tMax = (FloatToBits(w.tMax) >> 31) ? -tI : tI;
    // (a re-braided instance is entered once per subtree the ray meets: the best hit so far may lie in THIS instance from an earlier
    //  visit — marked stale, so that ExitInstance can tell a hit found during this visit, whose t becomes the render-space tMax as it
    //  is, from none, after which the saved tMax is restored)
    if (w.inst == inst) w.inst = inst | INST_STALE;
    w.curInst = inst;
    w.node = se.node;
    WF_LAZY_SET(w, 0);
    return true;
}
// the per-ray state a primitive test needs, made exact for the space being walked (RayWalk::lazy); oW, dW: the render-space ray
__device__ inline void WalkMakeExact(const FastBVH &bvh, RayWalk &w, V3 oW, V3 dW) {
    if (WF_LAZY_GET(w) == 1) {
        const wf_instance &in = bvh.instances[w.curInst];
        const FastDef fd = bvh.defs[in.def];
        // no primitive of this visit has been tested yet: |w.tMax| is still the render-space bound the visit started with
        float tI = __builtin_fabsf(w.tMax);
        V3 oI, dI;
        InstanceRay(in, oW, dW, &tI, &oI, &dI);
        WalkSetRay(fd.base, fd.cell, w, oI, dI);
        w.tMax = (FloatToBits(w.tMax) >> 31) ? -tI : tI;
    } else {
        w.o = oW;
        w.sh = MakeRayShear(dW);
    }
    WF_LAZY_SET(w, 0);
}
// an instance ENTRY on the stack / in a child slot (not the exit marker)
__device__ inline bool IsInstanceEntry(int node) { return node < 0 && node != NODE_NONE && node != NODE_EXIT && (int)((~(unsigned)node) >> 4) >= INST_FIRST; }
#ifndef WF_SAVE_WORLD
#define WF_SAVE_WORLD 0   // 0 = recompute (shipped); 2 = nine of the constants kept in LDS (flat: 35.8 against 36.0 ms, profiles/r06_walk_constants_lds_ab_sm16.txt); 1 = through HBM, MEASURED AND LEFT OFF (round 6, spec scene, 16 spp, same box, profiles/r06_walk_constants_reload_ab_sm16.txt): ExitInstance
                          // reloading the lane's render-space walk constants (16 dwords saved per ray: LdsStackT::loadWorld) instead of recomputing
                          // them (three IEEE divisions, three v_rcp: ~95 VALU instructions) — closest-hit 39.3 ms against 35.9, any-hit 16.8 against
                          // 16.3: four dependent 16-byte loads in front of every walk that leaves an instance stall the whole wave longer than the
                          // arithmetic occupies its issue slots
#endif
#ifndef WF_FUSE_EXIT_ENTER
#define WF_FUSE_EXIT_ENTER 1   // round 6: a lane that leaves an instance and pops another instance's entry enters it in the same step, and
                               // the render-space shear / slab constants in between are not rebuilt (three IEEE divisions, three v_rcp)
#endif
template <typename Stack>
__device__ inline void ExitInstance(const FastBVH &bvh, RayWalk &w, Stack &st, V3 oW, V3 dW) {
    const float saved = BitsToFloat((uint32_t)st.pop());
    // a hit found inside: its t (the instance ray's parameter) becomes the world tMax as it is, like the reference's
    // si->tHit (cpu/primitive.cpp:112-125); otherwise the world tMax is restored.  Near-tie marks are kept either way.
    const float tW = (w.inst == w.curInst) ? __builtin_fabsf(w.tMax) : __builtin_fabsf(saved);
    const bool mark = ((FloatToBits(w.tMax) | FloatToBits(saved)) >> 31) != 0;
    const int next = st.empty() ? NODE_NONE : st.pop();
#if WF_FUSE_EXIT_ENTER
    // the next entry on the stack is another subtree of the SAME instance (re-braided instances): the lane stays in the instance's space
    // and walks on there — one visit for the reference too, whose tMax runs through the whole definition
    if (IsInstanceEntry(next)) {
        const SubEntry se = bvh.subs[(int)((~(unsigned)next) >> 4) - INST_FIRST];
        if (se.inst == w.curInst && se.node != NODE_NONE) {
            st.push((int)FloatToBits(saved));
            st.push(NODE_EXIT);
            w.node = se.node;
            return;
        }
    }
#endif
    if (w.inst == (w.curInst | INST_STALE)) w.inst = w.curInst;   // the hit of an earlier visit of this instance stands
#if WF_LAZY_INST
    WalkSetSlab(bvh.base, bvh.cell, w, oW, dW);   // the shear of the render-space ray when a top-level leaf asks for it (WalkMakeExact)
    WF_LAZY_SET(w, 2);
#else
#if WF_SAVE_WORLD
    if (!(WF_FUSE_EXIT_ENTER && IsInstanceEntry(next))) st.loadWorld(w, oW, dW);   // the render-space constants, saved when the ray started (LdsStackT)
#else
    if (!(WF_FUSE_EXIT_ENTER && IsInstanceEntry(next))) WalkSetRay(bvh.base, bvh.cell, w, oW, dW);
#endif
#endif
    w.tMax = mark ? -tW : tW;
    w.curInst = -1;
    w.node = next;
}

__device__ inline bool WalkAmbiguous(const RayWalk &w) { return (FloatToBits(w.tMax) >> 31) != 0; }
__device__ inline float WalkT(const RayWalk &w) { return __builtin_fabsf(w.tMax); }
constexpr float TIE_BAND = 1 + 0x1p-20f;
#ifndef WF_TIE
#define WF_TIE 1   // 0: timing experiments only — no near-tie handling (round-1 behaviour: the visiting order decides ties)
#endif
#if WF_TIE
__device__ inline float WalkBound(const FastBVH &bvh, float t) { return fma(t, bvh.tieRel, bvh.absBand); }
#else
__device__ inline float WalkBound(const FastBVH &, float t) { return t; }
#endif
// a candidate hit at t: clearly nearer -> new best (an older mark is dropped: those candidates lie beyond);
// within the band of the best -> keep the nearer one, mark the ray
// PAIRS (the kernels of scenes with quadrics / patches / curves): the band of a comparison is the wide one when the candidate or the best
// hit so far is such a shape — or when the ray is already marked: a mark may stand for a general shape seen earlier whose t lies within
// the wide band of the candidate, and it must not be dropped by a triangle that is nearer by the narrow band only
template <bool PAIRS = false>
__device__ inline float WalkTestBound(const FastBVH &bvh, const RayWalk &w, bool candGeneral) {
    const float cur = __builtin_fabsf(w.tMax);
    if constexpr (PAIRS)
        if (!(candGeneral || w.prim >= bvh.firstGeneral || WalkAmbiguous(w))) return fma(cur, bvh.tieRelTri, bvh.absBandTri);
    return WalkBound(bvh, cur);
}
template <bool PAIRS = false>
__device__ inline bool WalkAccept(const FastBVH &bvh, RayWalk &w, float t, bool candGeneral = false) {
    const float cur = __builtin_fabsf(w.tMax);
    float bt = WalkBound(bvh, t);
    if constexpr (PAIRS)
        if (!(candGeneral || w.prim >= bvh.firstGeneral || WalkAmbiguous(w))) bt = fma(t, bvh.tieRelTri, bvh.absBandTri);
    if (!WF_TIE || bt < cur) { w.tMax = t; return true; }
    const bool nearer = t < cur;
    w.tMax = -(nearer ? t : cur);
    return nearer;
}

__device__ inline float CvtLo(uint32_t v) { return (float)(v & 0xffffu); }
__device__ inline float CvtHi(uint32_t v) { return (float)(v >> 16); }

// Interior visit: branch-free test of both children (packed left/right), nearest entry first.
// a, b = the node's two 16-byte halves (from LDS or global).  Precondition: w.node >= 0.
// RELAX: prune against the near-tie band (closest hit); any-hit walks prune against the exact tMax (their result
// does not depend on the visiting order)
#if WF_BVH4
// one child's slab test: entry t (lowest admissible), exit t
__device__ inline void ChildSlab(const RayWalk &w, uint32_t qx, uint32_t qy, uint32_t qz, float *tN, float *tF) {
    const uint32_t x = __builtin_amdgcn_perm(qx, qx, w.selx), y = __builtin_amdgcn_perm(qy, qy, w.sely), z = __builtin_amdgcn_perm(qz, qz, w.selz);
    const f2 X = __builtin_elementwise_fma(f2{CvtLo(x), CvtHi(x)}, f2{w.a.x, w.af.x}, f2{w.bn.x, w.bf.x});
    const f2 Y = __builtin_elementwise_fma(f2{CvtLo(y), CvtHi(y)}, f2{w.a.y, w.af.y}, f2{w.bn.y, w.bf.y});
    const f2 Z = __builtin_elementwise_fma(f2{CvtLo(z), CvtHi(z)}, f2{w.a.z, w.af.z}, f2{w.bn.z, w.bf.z});
    *tN = __builtin_fmaxf(__builtin_fmaxf(X.x, Y.x), Z.x);
    *tF = __builtin_fminf(__builtin_fminf(X.y, Y.y), Z.y);
}
__device__ inline void CSwap(float &ka, int &ra, float &kb, int &rb) {
    const bool s = kb < ka;
    const float k = s ? kb : ka, K = s ? ka : kb;
    const int r = s ? rb : ra, R = s ? ra : rb;
    ka = k; kb = K; ra = r; rb = R;
}
#ifndef WF_ANY_NOSORT
#define WF_ANY_NOSORT 0
#endif
#ifndef WF_PUSH_RESERVE
#define WF_PUSH_RESERVE 0
#endif
template <bool RELAX = true, typename Stack>
__device__ inline void InteriorStep(const FastBVH &bvh, RayWalk &w, Stack &st, const U4 *n) {
    const float tPrune = RELAX ? WalkBound(bvh, __builtin_fabsf(w.tMax)) : w.tMax;
    const float lim = __builtin_fminf(tPrune, 3.0e38f);
    float k0, k1, k2, k3, e;
    ChildSlab(w, n[0].x, n[0].y, n[0].z, &k0, &e);
    const bool h0 = __builtin_fmaxf(k0, 0.f) <= __builtin_fminf(e, tPrune);
    ChildSlab(w, n[0].w, n[1].x, n[1].y, &k1, &e);
    const bool h1 = __builtin_fmaxf(k1, 0.f) <= __builtin_fminf(e, tPrune);
    ChildSlab(w, n[1].z, n[1].w, n[2].x, &k2, &e);
    const bool h2 = __builtin_fmaxf(k2, 0.f) <= __builtin_fminf(e, tPrune);
    ChildSlab(w, n[2].y, n[2].z, n[2].w, &k3, &e);
    const bool h3 = __builtin_fmaxf(k3, 0.f) <= __builtin_fminf(e, tPrune);
    (void)lim;
#if WF_ANY_NOSORT
    // any-hit walks (RELAX = false): the result does not depend on the visiting order, and the nearest-first network below is 25 of the
    // step's ~130 VALU instructions — the hit children are visited in slot order instead (the builder's order: the binary tree's split
    // first, then the children opened by area)
    if constexpr (!RELAX) {
        const int c0 = (int)n[3].x, c1 = (int)n[3].y, c2 = (int)n[3].z, c3 = (int)n[3].w;
        if (!(h0 | h1 | h2 | h3)) { w.node = st.empty() ? NODE_NONE : st.pop(); return; }
        if (h3 & (h0 | h1 | h2)) st.push(c3);
        if (h2 & (h0 | h1)) st.push(c2);
        if (h1 & h0) st.push(c1);
        w.node = h0 ? c0 : h1 ? c1 : h2 ? c2 : c3;
        return;
    }
#endif
    k0 = h0 ? k0 : WF_INFINITY; k1 = h1 ? k1 : WF_INFINITY; k2 = h2 ? k2 : WF_INFINITY; k3 = h3 ? k3 : WF_INFINITY;
    int r0 = (int)n[3].x, r1 = (int)n[3].y, r2 = (int)n[3].z, r3 = (int)n[3].w;
    // nearest entry first: 5-comparator sorting network; missed children (key = inf) sink to the end
    CSwap(k0, r0, k1, r1); CSwap(k2, r2, k3, r3); CSwap(k0, r0, k2, r2); CSwap(k1, r1, k3, r3); CSwap(k1, r1, k2, r2);
    const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    if (nh == 0) { w.node = st.empty() ? NODE_NONE : st.pop(); return; }
#if WF_PUSH_RESERVE == 2
    // branch-free pushes: room for three entries is made first (rarely needed), then every candidate is WRITTEN to the ring and the
    // count advances only for the ones that are wanted — no exec-masked region per push
    st.reserve(3);
    if (st.n - st.lo + 3 <= TSTACK) {
        st.pushIf(r3, nh > 3);
        st.pushIf(r2, nh > 2);
        st.pushIf(r1, nh > 1);
    }
#elif WF_PUSH_RESERVE
    // one ring-full test for the step's pushes instead of one per push (a full ring with a spill row overflow drops entries in both forms: wf_sync reports it)
    // (measured, round 6: 28.5 against 27.4 ms closest-hit — left off)
    if (nh > 1) {
        st.reserve(nh - 1);
        if (st.n - st.lo + (nh - 1) <= TSTACK) {
            if (nh > 3) st.pushReserved(r3);
            if (nh > 2) st.pushReserved(r2);
            st.pushReserved(r1);
        }
    }
#else
    if (nh > 3) st.push(r3);
    if (nh > 2) st.push(r2);
    if (nh > 1) st.push(r1);
#endif
    w.node = r0;
}
#else
template <bool RELAX = true, typename Stack>
__device__ inline void InteriorStep(const FastBVH &bvh, RayWalk &w, Stack &st, const U4 *n) {
    const U4 a = n[0], b = n[1];
    // near plane -> low half, far plane -> high half of each dword
    const uint32_t lx = __builtin_amdgcn_perm(a.x, a.x, w.selx), ly = __builtin_amdgcn_perm(a.y, a.y, w.sely), lz = __builtin_amdgcn_perm(a.z, a.z, w.selz);
    const uint32_t rx = __builtin_amdgcn_perm(a.w, a.w, w.selx), ry = __builtin_amdgcn_perm(b.x, b.x, w.sely), rz = __builtin_amdgcn_perm(b.y, b.y, w.selz);
    const int left = (int)b.z, right = (int)b.w;
    // one packed fma per child and axis: {entry, exit} = {qNear, qFar} * {a, af} + {bn, bf}
    const f2 cx{w.a.x, w.af.x}, cy{w.a.y, w.af.y}, cz{w.a.z, w.af.z};
    const f2 dx{w.bn.x, w.bf.x}, dy{w.bn.y, w.bf.y}, dz{w.bn.z, w.bf.z};
    const f2 Lx = __builtin_elementwise_fma(f2{CvtLo(lx), CvtHi(lx)}, cx, dx);
    const f2 Ly = __builtin_elementwise_fma(f2{CvtLo(ly), CvtHi(ly)}, cy, dy);
    const f2 Lz = __builtin_elementwise_fma(f2{CvtLo(lz), CvtHi(lz)}, cz, dz);
    const f2 Rx = __builtin_elementwise_fma(f2{CvtLo(rx), CvtHi(rx)}, cx, dx);
    const f2 Ry = __builtin_elementwise_fma(f2{CvtLo(ry), CvtHi(ry)}, cy, dy);
    const f2 Rz = __builtin_elementwise_fma(f2{CvtLo(rz), CvtHi(rz)}, cz, dz);
    // tMin < raytMax && tMax > 0 && tMin <= tMax, relaxed to max(tMin, 0) <= min(tMax, raytMax)
    const float tL = __builtin_fmaxf(__builtin_fmaxf(Lx.x, Ly.x), Lz.x), tR = __builtin_fmaxf(__builtin_fmaxf(Rx.x, Ry.x), Rz.x);
    const float eL = __builtin_fminf(__builtin_fminf(Lx.y, Ly.y), Lz.y), eR = __builtin_fminf(__builtin_fminf(Rx.y, Ry.y), Rz.y);
    const float tPrune = RELAX ? WalkBound(bvh, __builtin_fabsf(w.tMax)) : w.tMax;
    const bool hitL = __builtin_fmaxf(tL, 0.f) <= __builtin_fminf(eL, tPrune);
    const bool hitR = __builtin_fmaxf(tR, 0.f) <= __builtin_fminf(eR, tPrune);
    const bool rightFirst = tR < tL;
    if (hitL & hitR) {
        st.push(rightFirst ? left : right);
        w.node = rightFirst ? right : left;
    } else if (hitL | hitR) w.node = hitL ? left : right;
    else w.node = st.empty() ? NODE_NONE : st.pop();
}
#endif
// Leaf: <= 16 triangle tests.  ANY: stop at the first hit.  Precondition: w.node < 0 && w.node != NODE_NONE.
// The general-primitive variants (ALPHA): leaf entries marked c.z == 2 are triangles whose mesh carries an alpha
// texture (ex.accept(prim, b0, b1, b2) decides), entries marked c.z == 3 are spheres (ex.sphere(prim, tMax, &hit);
// the hit's pObj travels in b0..b2).  Scenes without either use the plain variants, which never see the marks.
// RayWalk::route bit: the walk met a leaf entry only the general-primitive kernels can test (a quadric / patch / curve: c.z == 3) and
// stopped — the ray is walked again by those kernels (round 6, wf_backend.hip "TWO-CLASS TRAVERSAL").  Extra::deferGeneral selects it.
constexpr uint32_t WALK_DEFER = 0x40000000u;
struct NoExtra {
    static constexpr bool pairBands = false;
    static constexpr bool deferGeneral = false;
    __device__ bool accept(int, float, float, float) const { return true; }
    __device__ bool sphere(int, float, QuadricHit *) const { return false; }
    __device__ void exact(RayWalk &) const {}
};
template <bool ANY, bool ALPHA = false, bool INST = false, typename Stack, typename Extra = NoExtra>
__device__ inline void LeafStep(const FastBVH &bvh, RayWalk &w, Stack &st, const Extra &ex = Extra()) {
    unsigned ref = ~(unsigned)w.node;
    int first = (int)(ref >> 4), count = (int)(ref & 15u) + 1;
    bool done = false;
    for (int i = 0; i < count; ++i) {
        const LeafTri *lt = bvh.tris + first + i;
        const F4 ta = lt->a, tb = lt->b, tc = lt->c;
        TriHit h;
        if constexpr (INST)
            if (tc.z == 4.f) {  // an object instance: visited through the stack (the main loop enters it when it is popped)
                st.push((int)~(((unsigned)INST_FIRST + FloatToBits(tc.y)) << 4));
                continue;
            }
        if constexpr (INST)
            if (WF_LAZY_GET(w)) ex.exact(w);   // the first primitive test since the walk changed spaces (WF_LAZY_INST)
        if constexpr (Extra::deferGeneral)
            if (tc.z == 3.f) { w.route |= WALK_DEFER; done = true; break; }
        // closest hit: test against the relaxed bound so that near-ties are seen (WalkAccept sorts them out)
        constexpr bool PAIRS = Extra::pairBands;
        if constexpr (ALPHA)
            if (tc.z == 3.f) {
                const float tTest = ANY ? w.tMax : WalkBound(bvh, __builtin_fabsf(w.tMax));
                QuadricHit qh;
                // (closest hit: against the RELAXED bound, like the triangles below — a quadric whose t lies inside the near-tie band of
                //  the current best must reach WalkAccept to mark the ray; tested against the exact bound, the second of two coincident
                //  quadrics was silently dropped and the visiting order decided: fuzz finding s200010 on the GPU, round 4)
                if (ex.sphere((int)FloatToBits(tc.y), tTest, &qh)) {
                    if (ANY) { w.prim = (int)FloatToBits(tc.y); w.tMax = qh.tHit; done = true; break; }
                    if (WalkAccept<PAIRS>(bvh, w, qh.tHit, true)) {
                        w.prim = (int)FloatToBits(tc.y);
                        w.inst = w.curInst;
                        w.route = FloatToBits(tc.w);
                        w.b0 = qh.pObj.x; w.b1 = qh.pObj.y; w.b2 = qh.pObj.z;
                    }
                }
                continue;
            }
        const float tTest = ANY ? w.tMax : WalkTestBound<PAIRS>(bvh, w, false);
        if ((ALPHA ? tc.z != 1.f : tc.z == 0.f) &&
            IntersectTriangleSheared(w.o, w.sh, tTest, V3{ta.x, ta.y, ta.z}, V3{ta.w, tb.x, tb.y}, V3{tb.z, tb.w, tc.x}, &h, false)) {
            if constexpr (ALPHA)
                if (tc.z == 2.f && !ex.accept((int)FloatToBits(tc.y), h.b0, h.b1, h.b2)) continue;
            if (ANY) { w.prim = (int)FloatToBits(tc.y); w.tMax = h.t; done = true; break; }
            if (WalkAccept<PAIRS>(bvh, w, h.t, false)) {
                w.prim = (int)FloatToBits(tc.y);
                w.inst = w.curInst;
                w.route = FloatToBits(tc.w);
                w.b0 = h.b0; w.b1 = h.b1; w.b2 = h.b2;
            }
        }
    }
    w.node = (done || st.empty()) ? NODE_NONE : st.pop();
}

}  // namespace wf
