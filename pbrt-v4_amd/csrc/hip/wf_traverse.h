// wf_traverse.h — the production BVH traversal for CDNA4 (device only; included by wf_backend.hip).
//
// The reference-order walk in common/wf_shapes.h (BVHIntersectClosest/Any) visits one 32-byte
// LinearBVHNode per step and chases three levels of indirection per triangle (bvh_prims -> tri_indices
// -> P); it stays as the *counting* variant, because SURVEY.md §8(d) defines the roofline's algorithmic
// bytes on exactly those visit counts.  The kernels that render use the layout and loop below instead.
//
// What bounds traversal on MI355X (profiles/r01_*pmc*): not HBM and not VALU (10 % busy) but the per-CU
// vector L1 (TCP): every lane of a wave gathers its own node, the TCP retires about one tag lookup per
// clock, and it is ~87 % busy.  So the design minimises L1 lookups per ray:
//
//  * QNode (32 B, two dwordx4): an interior node carries BOTH children's bounds, quantised to 16 bits per
//    plane on a global grid over the scene bounds, rounded outwards.  One fetch feeds two slab tests;
//    half the bytes (and lookups) of two float boxes.  Outward rounding keeps results exact: the slab
//    test (same formula as Bounds3::IntersectP) on a superset box passes whenever the reference's test on
//    the exact box passes, and WHICH triangle is hit is decided only by the (exact, float) triangle test.
//  * the top TOP_NODES nodes of the tree (breadth-first numbering, ~10 levels) are copied into LDS by each
//    workgroup at kernel start: most of a ray's interior visits are served by ds_read_b128 (LDS has
//    128 B/clk/CU to spare) and never touch the TCP.
//  * LeafTri (48 B, three dwordx4): the three vertices of each triangle in BVH leaf order — the
//    "3 indices + 3 Point3f" of the §8(d) formula as one contiguous record, no index chase.
//  * children are visited nearest-entry first; the node stack lives in LDS, one column per lane.
//  * persistent waves ("while-while"): a wave pulls 64 consecutive rays with one atomic, all lanes descend
//    interior nodes until every lane sits at a leaf, leaves are processed together, results are written
//    together when the whole wave is done (whole-wave refill measured faster than per-lane refill: mixing
//    rays destroys the little coherence consecutive queue entries have).
//
// Exact ties in t between two triangles are the one case where visiting order can pick the other triangle
// (same t): tests/test_gpu_parity.py allows a different triangle id only at bit-equal t.
#pragma once

namespace wf {

struct alignas(16) QNode {
    uint32_t q[6];  // 12 x u16: L.min.xyz, L.max.xyz, R.min.xyz, R.max.xyz (two per dword, low half first)
    int32_t left, right;  // >= 0: interior QNode index; < 0: leaf ~((first << 4) | (count - 1)); NODE_NONE: absent
};
struct alignas(16) LeafTri {
    F4 a;  // p0.xyz, p1.x
    F4 b;  // p1.yz, p2.xy
    F4 c;  // p2.z, triangle id (int bits), 1.0 if the triangle is degenerate (zero-length normal) else 0, -
};
struct alignas(16) U4 { uint32_t x, y, z, w; };

constexpr int NODE_NONE = (int)0x80000000;
#ifndef WF_TOP_NODES
#define WF_TOP_NODES 512
#endif
#ifndef WF_TBLOCK
#define WF_TBLOCK 256
#endif
#ifndef WF_TSTACK
#define WF_TSTACK 12
#endif
#ifndef WF_TWAVES
#define WF_TWAVES 6   // __launch_bounds__ second argument (minimum waves per SIMD) of the traversal kernels
#endif
constexpr int TOP_NODES = WF_TOP_NODES;  // QNodes cached in LDS per workgroup (32 B each)
constexpr int TBLOCK = WF_TBLOCK;        // threads per workgroup of the traversal kernels
constexpr int TSTACK = WF_TSTACK;        // LDS stack entries per lane (x 4 B x TBLOCK)

struct FastBVH {
    const QNode *nodes;
    const LeafTri *tris;
    int nNodes;
    float base[3], cell[3];  // dequantisation: v = fma(q, cell, base)
};

struct RayWalk {
    V3 o, invDir;
    RayShear sh;   // per-ray part of the triangle test (MakeRayShear)
    float tMax;
    int negMask;
    int node;  // current ref; NODE_NONE = finished
    int prim;
    float b0, b1, b2;
};

__device__ inline void WalkInit(RayWalk &w, V3 o, V3 d, float tMax) {
    w.o = o; w.tMax = tMax;
    w.sh = MakeRayShear(d);
    w.invDir = V3{1 / d.x, 1 / d.y, 1 / d.z};
    w.negMask = int(w.invDir.x < 0) | (int(w.invDir.y < 0) << 1) | (int(w.invDir.z < 0) << 2);
    w.node = 0;
    w.prim = -1;
    w.b0 = w.b1 = w.b2 = 0;
}

// Bounds3::IntersectP (util/vecmath.h:1574-1608) that also reports the entry distance for child ordering
__device__ inline bool SlabTestT(const float bmin[3], const float bmax[3], V3 o, float raytMax, V3 invDir, int negMask, float *tEntry) {
    const bool n0 = negMask & 1, n1 = negMask & 2, n2 = negMask & 4;
    float tMin = ((n0 ? bmax[0] : bmin[0]) - o.x) * invDir.x;
    float tMax = ((n0 ? bmin[0] : bmax[0]) - o.x) * invDir.x;
    float tyMin = ((n1 ? bmax[1] : bmin[1]) - o.y) * invDir.y;
    float tyMax = ((n1 ? bmin[1] : bmax[1]) - o.y) * invDir.y;
    tMax *= 1 + 2 * gamma(3);
    tyMax *= 1 + 2 * gamma(3);
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = ((n2 ? bmax[2] : bmin[2]) - o.z) * invDir.z;
    float tzMax = ((n2 ? bmin[2] : bmax[2]) - o.z) * invDir.z;
    tzMax *= 1 + 2 * gamma(3);
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    *tEntry = tMin;
    return (tMin < raytMax) && (tMax > 0);
}

// Interior visit.  a, b = the node's two 16-byte halves (from LDS or global).  Precondition: w.node >= 0.
template <typename Stack>
__device__ inline void InteriorStep(const FastBVH &bvh, RayWalk &w, Stack &st, U4 a, U4 b) {
    auto dq = [&](uint32_t word, int hi, int axis) { return fma((float)(hi ? (word >> 16) : (word & 0xffffu)), bvh.cell[axis], bvh.base[axis]); };
    const float lmin[3] = {dq(a.x, 0, 0), dq(a.x, 1, 1), dq(a.y, 0, 2)};
    const float lmax[3] = {dq(a.y, 1, 0), dq(a.z, 0, 1), dq(a.z, 1, 2)};
    const float rmin[3] = {dq(a.w, 0, 0), dq(a.w, 1, 1), dq(b.x, 0, 2)};
    const float rmax[3] = {dq(b.x, 1, 0), dq(b.y, 0, 1), dq(b.y, 1, 2)};
    const int left = (int)b.z, right = (int)b.w;
    float tL = 0, tR = 0;
    bool hitL = SlabTestT(lmin, lmax, w.o, w.tMax, w.invDir, w.negMask, &tL);
    bool hitR = right != NODE_NONE && SlabTestT(rmin, rmax, w.o, w.tMax, w.invDir, w.negMask, &tR);
    if (hitL && hitR) {
        bool rightFirst = tR < tL;
        st.push(rightFirst ? left : right);
        w.node = rightFirst ? right : left;
    } else if (hitL) w.node = left;
    else if (hitR) w.node = right;
    else w.node = st.empty() ? NODE_NONE : st.pop();
}
// Leaf: <= 16 triangle tests.  Precondition: w.node < 0 && w.node != NODE_NONE.  ANY: stop at the first hit.
template <bool ANY, typename Stack>
__device__ inline void LeafStep(const FastBVH &bvh, RayWalk &w, Stack &st) {
    unsigned ref = ~(unsigned)w.node;
    int first = (int)(ref >> 4), count = (int)(ref & 15u) + 1;
    bool done = false;
    for (int i = 0; i < count; ++i) {
        const LeafTri *lt = bvh.tris + first + i;
        const F4 ta = lt->a, tb = lt->b, tc = lt->c;
        TriHit h;
        if (tc.z == 0.f &&
            IntersectTriangleSheared(w.o, w.sh, w.tMax, V3{ta.x, ta.y, ta.z}, V3{ta.w, tb.x, tb.y}, V3{tb.z, tb.w, tc.x}, &h, false)) {
            w.prim = (int)FloatToBits(tc.y);
            w.b0 = h.b0; w.b1 = h.b1; w.b2 = h.b2;
            w.tMax = h.t;
            if (ANY) { done = true; break; }
        }
    }
    w.node = (done || st.empty()) ? NODE_NONE : st.pop();
}

}  // namespace wf
