// bvh_build.cpp — host SAH BVH builder producing the reference's LinearBVHNode layout.
// Restates BVHAggregate's SAH path: constructor cpu/aggregates.cpp:140-196, buildRecursive :198-387
// (12 buckets, leaf when cost says so and n <= maxPrimsInNode), flattenBVH :505-521.  The build is
// sequential (the reference forks above 128k primitives, which only permutes leaf storage order).
#include "scene.h"

#include <algorithm>

namespace wf {
namespace {

struct BVHPrim {
    int index;
    B3 bounds;
    V3 Centroid() const { return .5f * bounds.pMin + .5f * bounds.pMax; }
};
struct BuildNode {
    B3 bounds;
    BuildNode *children[2] = {nullptr, nullptr};
    int splitAxis = 0, firstPrimOffset = 0, nPrimitives = 0;
};
struct Builder {
    int maxPrimsInNode;
    std::vector<BuildNode> pool;
    std::vector<int32_t> *ordered;
    int totalNodes = 0;
    size_t poolUsed = 0;
    BuildNode *NewNode() { return &pool[poolUsed++]; }

    void InitLeaf(BuildNode *node, BVHPrim *prims, int n, const B3 &bounds) {
        int first = (int)ordered->size();
        for (int i = 0; i < n; ++i) ordered->push_back(prims[i].index);
        node->firstPrimOffset = first;
        node->nPrimitives = n;
        node->bounds = bounds;
    }

    BuildNode *Build(BVHPrim *prims, int n) {
        BuildNode *node = NewNode();
        ++totalNodes;
        B3 bounds;
        for (int i = 0; i < n; ++i) bounds = Union(bounds, prims[i].bounds);
        if (bounds.SurfaceArea() == 0 || n == 1) { InitLeaf(node, prims, n, bounds); return node; }
        B3 centroidBounds;
        for (int i = 0; i < n; ++i) centroidBounds = Union(centroidBounds, prims[i].Centroid());
        int dim = centroidBounds.MaxDimension();
        if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) { InitLeaf(node, prims, n, bounds); return node; }
        int mid = n / 2;
        if (n <= 2) {
            std::nth_element(prims, prims + mid, prims + n,
                             [dim](const BVHPrim &a, const BVHPrim &b) { return a.Centroid()[dim] < b.Centroid()[dim]; });
        } else {
            constexpr int nBuckets = 12;
            struct Bucket { int count = 0; B3 bounds; } buckets[nBuckets];
            for (int i = 0; i < n; ++i) {
                int b = nBuckets * centroidBounds.Offset(prims[i].Centroid())[dim];
                if (b == nBuckets) b = nBuckets - 1;
                buckets[b].count++;
                buckets[b].bounds = Union(buckets[b].bounds, prims[i].bounds);
            }
            constexpr int nSplits = nBuckets - 1;
            float costs[nSplits] = {};
            int countBelow = 0;
            B3 boundBelow;
            for (int i = 0; i < nSplits; ++i) {
                boundBelow = Union(boundBelow, buckets[i].bounds);
                countBelow += buckets[i].count;
                costs[i] += countBelow * boundBelow.SurfaceArea();
            }
            int countAbove = 0;
            B3 boundAbove;
            for (int i = nSplits; i >= 1; --i) {
                boundAbove = Union(boundAbove, buckets[i].bounds);
                countAbove += buckets[i].count;
                costs[i - 1] += countAbove * boundAbove.SurfaceArea();
            }
            int minCostSplitBucket = -1;
            float minCost = WF_INFINITY;
            for (int i = 0; i < nSplits; ++i)
                if (costs[i] < minCost) { minCost = costs[i]; minCostSplitBucket = i; }
            float leafCost = n;
            minCost = 1.f / 2.f + minCost / bounds.SurfaceArea();
            if (n > maxPrimsInNode || minCost < leafCost) {
                BVHPrim *midIter = std::partition(prims, prims + n, [=](const BVHPrim &bp) {
                    int b = nBuckets * centroidBounds.Offset(bp.Centroid())[dim];
                    if (b == nBuckets) b = nBuckets - 1;
                    return b <= minCostSplitBucket;
                });
                mid = int(midIter - prims);
            } else { InitLeaf(node, prims, n, bounds); return node; }
        }
        BuildNode *c0 = Build(prims, mid);
        BuildNode *c1 = Build(prims + mid, n - mid);
        node->children[0] = c0;
        node->children[1] = c1;
        node->bounds = Union(c0->bounds, c1->bounds);
        node->splitAxis = dim;
        node->nPrimitives = 0;
        return node;
    }

    int Flatten(const BuildNode *node, std::vector<wf_bvh_node> *out, int *offset) {
        wf_bvh_node *ln = &(*out)[*offset];
        for (int c = 0; c < 3; ++c) { ln->bmin[c] = node->bounds.pMin[c]; ln->bmax[c] = node->bounds.pMax[c]; }
        int nodeOffset = (*offset)++;
        if (node->nPrimitives > 0) {
            ln->offset = node->firstPrimOffset;
            ln->nprims = (uint16_t)node->nPrimitives;
            ln->axis = 0;
        } else {
            ln->axis = (uint8_t)node->splitAxis;
            ln->nprims = 0;
            Flatten(node->children[0], out, offset);
            int second = Flatten(node->children[1], out, offset);
            (*out)[nodeOffset].offset = second;
        }
        return nodeOffset;
    }
};

}  // namespace

B3 TriangleBounds(const std::vector<float> &P, const std::vector<int32_t> &triIndices, int i) {
    auto vtx = [&](int k) { int v = triIndices[3 * (size_t)i + k]; return V3{P[3 * (size_t)v], P[3 * (size_t)v + 1], P[3 * (size_t)v + 2]}; };
    // Triangle::Bounds (shapes.cpp:283-290): Union(Bounds3f(p0, p1), p2)
    V3 p0 = vtx(0), p1 = vtx(1), p2 = vtx(2);
    B3 b;
    b.pMin = {fmin(p0.x, p1.x), fmin(p0.y, p1.y), fmin(p0.z, p1.z)};
    b.pMax = {fmax(p0.x, p1.x), fmax(p0.y, p1.y), fmax(p0.z, p1.z)};
    return Union(b, p2);
}

int BuildBVH(const std::vector<std::pair<int, B3>> &primsIn, int maxPrimsInNode, std::vector<wf_bvh_node> *nodes, std::vector<int32_t> *orderedPrims) {
    const int nAll = (int)primsIn.size();
    if (nAll == 0) return -1;
    std::vector<BVHPrim> prims(nAll);
    for (int i = 0; i < nAll; ++i) { prims[i].index = primsIn[i].first; prims[i].bounds = primsIn[i].second; }
    const int nodeBase = (int)nodes->size(), primBase = (int)orderedPrims->size();
    std::vector<int32_t> ordered;
    ordered.reserve(nAll);
    Builder bld;
    bld.maxPrimsInNode = std::min(255, maxPrimsInNode);
    bld.pool.resize(2 * (size_t)nAll);
    bld.ordered = &ordered;
    BuildNode *root = bld.Build(prims.data(), nAll);
    std::vector<wf_bvh_node> local(bld.totalNodes, wf_bvh_node{});
    int offset = 0;
    bld.Flatten(root, &local, &offset);
    for (wf_bvh_node &n : local) n.offset += n.nprims > 0 ? primBase : nodeBase;  // leaf: primitivesOffset, interior: secondChildOffset
    nodes->insert(nodes->end(), local.begin(), local.end());
    orderedPrims->insert(orderedPrims->end(), ordered.begin(), ordered.end());
    return nodeBase;
}

}  // namespace wf
