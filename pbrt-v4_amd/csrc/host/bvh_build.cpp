// bvh_build.cpp — host BVH builders producing the reference's LinearBVHNode layout.
// Restates BVHAggregate: constructor cpu/aggregates.cpp:140-196; the SAH path buildRecursive :198-387 (12 buckets, leaf when
// cost says so and n <= maxPrimsInNode); the HLBVH path buildHLBVH :389-447 + emitLBVH :449-503 + buildUpperSAH :626-722
// (Morton codes of the centroids, stable radix sort — on the GPU through wf_morton_sort when a device is there —, one LBVH
// treelet per 12-bit Morton prefix, SAH over the treelet roots); flattenBVH :505-521.  The builds are sequential (the
// reference's parallel sections only permute the storage order of leaves, which no result depends on).
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
MortonSortFn g_mortonSort = nullptr;

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

    // ---- HLBVH (splitmethod "hlbvh") ----
    struct MortonPrim { int primitiveIndex; uint32_t mortonCode; };
    static uint32_t LeftShift3(uint32_t x) {  // util/math.h:99-112
        if (x == (1 << 10)) --x;
        x = (x | (x << 16)) & 0b00000011000000000000000011111111;
        x = (x | (x << 8)) & 0b00000011000000001111000000001111;
        x = (x | (x << 4)) & 0b00000011000011000011000011000011;
        x = (x | (x << 2)) & 0b00001001001001001001001001001001;
        return x;
    }
    static uint32_t EncodeMorton3(float x, float y, float z) { return (LeftShift3((uint32_t)z) << 2) | (LeftShift3((uint32_t)y) << 1) | LeftShift3((uint32_t)x); }
    const BVHPrim *all = nullptr;  // bvhPrimitives, indexed by position in the input
    BuildNode *EmitLBVH(const MortonPrim *mp, int n, int bitIndex) {
        if (bitIndex == -1 || n < maxPrimsInNode) {
            ++totalNodes;
            BuildNode *node = NewNode();
            B3 bounds;
            int first = (int)ordered->size();
            for (int i = 0; i < n; ++i) {
                ordered->push_back(all[mp[i].primitiveIndex].index);
                bounds = Union(bounds, all[mp[i].primitiveIndex].bounds);
            }
            node->firstPrimOffset = first; node->nPrimitives = n; node->bounds = bounds;
            return node;
        }
        uint32_t mask = 1u << bitIndex;
        if ((mp[0].mortonCode & mask) == (mp[n - 1].mortonCode & mask)) return EmitLBVH(mp, n, bitIndex - 1);
        int splitOffset = FindInterval(n, [&](int index) { return (mp[0].mortonCode & mask) == (mp[index].mortonCode & mask); });
        ++splitOffset;
        ++totalNodes;
        BuildNode *node = NewNode();
        BuildNode *c0 = EmitLBVH(mp, splitOffset, bitIndex - 1);
        BuildNode *c1 = EmitLBVH(mp + splitOffset, n - splitOffset, bitIndex - 1);
        node->children[0] = c0; node->children[1] = c1;
        node->bounds = Union(c0->bounds, c1->bounds);
        node->splitAxis = bitIndex % 3;
        node->nPrimitives = 0;
        return node;
    }
    BuildNode *BuildUpperSAH(std::vector<BuildNode *> &roots, int start, int end) {
        int nNodes = end - start;
        if (nNodes == 1) return roots[start];
        ++totalNodes;
        BuildNode *node = NewNode();
        B3 bounds;
        for (int i = start; i < end; ++i) bounds = Union(bounds, roots[i]->bounds);
        B3 centroidBounds;
        for (int i = start; i < end; ++i) centroidBounds = Union(centroidBounds, (roots[i]->bounds.pMin + roots[i]->bounds.pMax) * 0.5f);
        int dim = centroidBounds.MaxDimension();
        constexpr int nBuckets = 12;
        struct Bucket { int count = 0; B3 bounds; } buckets[nBuckets];
        const float cmin = centroidBounds.pMin[dim], cmax = centroidBounds.pMax[dim];
        auto bucketOf = [=](const BuildNode *nd) {
            float centroid = (nd->bounds.pMin[dim] + nd->bounds.pMax[dim]) * 0.5f;
            int b = nBuckets * ((centroid - cmin) / (cmax - cmin));
            if (b == nBuckets) b = nBuckets - 1;
            return b;
        };
        for (int i = start; i < end; ++i) {
            int b = bucketOf(roots[i]);
            buckets[b].count++;
            buckets[b].bounds = Union(buckets[b].bounds, roots[i]->bounds);
        }
        float cost[nBuckets - 1];
        for (int i = 0; i < nBuckets - 1; ++i) {
            B3 b0, b1;
            int count0 = 0, count1 = 0;
            for (int j = 0; j <= i; ++j) { b0 = Union(b0, buckets[j].bounds); count0 += buckets[j].count; }
            for (int j = i + 1; j < nBuckets; ++j) { b1 = Union(b1, buckets[j].bounds); count1 += buckets[j].count; }
            cost[i] = .125f + (count0 * b0.SurfaceArea() + count1 * b1.SurfaceArea()) / bounds.SurfaceArea();
        }
        float minCost = cost[0];
        int minCostSplitBucket = 0;
        for (int i = 1; i < nBuckets - 1; ++i)
            if (cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; }
        BuildNode **pmid = std::partition(&roots[start], &roots[end - 1] + 1, [=](const BuildNode *nd) { return bucketOf(nd) <= minCostSplitBucket; });
        int mid = int(pmid - &roots[0]);
        BuildNode *c0 = BuildUpperSAH(roots, start, mid);
        BuildNode *c1 = BuildUpperSAH(roots, mid, end);
        node->children[0] = c0; node->children[1] = c1;
        node->bounds = Union(c0->bounds, c1->bounds);
        node->splitAxis = dim;
        node->nPrimitives = 0;
        return node;
    }
    BuildNode *BuildHLBVH(const BVHPrim *prims, int n) {
        all = prims;
        B3 bounds;
        for (int i = 0; i < n; ++i) bounds = Union(bounds, prims[i].Centroid());
        std::vector<MortonPrim> mortonPrims(n);
        bool sorted = false;
        if (g_mortonSort && n >= 4096 && !getenv("WF_HOST_MORTON_SORT")) {
            // device path: Morton codes + stable 30-bit radix sort on the GPU (wf_morton_sort, libwfhip.so)
            std::vector<float> c(3 * (size_t)n);
            for (int i = 0; i < n; ++i) { V3 p = prims[i].Centroid(); c[3 * (size_t)i] = p.x; c[3 * (size_t)i + 1] = p.y; c[3 * (size_t)i + 2] = p.z; }
            std::vector<uint32_t> codes(n), order(n);
            const float b6[6] = {bounds.pMin.x, bounds.pMin.y, bounds.pMin.z, bounds.pMax.x, bounds.pMax.y, bounds.pMax.z};
            if (g_mortonSort(n, c.data(), b6, codes.data(), order.data()) == 0) {
                for (int i = 0; i < n; ++i) mortonPrims[i] = MortonPrim{(int)order[i], codes[i]};
                sorted = true;
            }
        }
        if (!sorted) {
            constexpr int mortonScale = 1 << 10;
            for (int i = 0; i < n; ++i) {
                mortonPrims[i].primitiveIndex = i;
                V3 offset = bounds.Offset(prims[i].Centroid()) * (float)mortonScale;
                mortonPrims[i].mortonCode = EncodeMorton3(offset.x, offset.y, offset.z);
            }
            // RadixSort (cpu/aggregates.cpp:92-127): LSD, 6 bits per pass over 30 bits, stable
            std::stable_sort(mortonPrims.begin(), mortonPrims.end(), [](const MortonPrim &a, const MortonPrim &b) { return a.mortonCode < b.mortonCode; });
        }
        std::vector<BuildNode *> treelets;
        for (size_t start = 0, end = 1; end <= (size_t)n; ++end) {
            const uint32_t mask = 0b00111111111111000000000000000000;
            if (end == (size_t)n || ((mortonPrims[start].mortonCode & mask) != (mortonPrims[end].mortonCode & mask))) {
                treelets.push_back(EmitLBVH(&mortonPrims[start], (int)(end - start), 29 - 12));
                start = end;
            }
        }
        return BuildUpperSAH(treelets, 0, (int)treelets.size());
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

void SetMortonSort(MortonSortFn fn) { g_mortonSort = fn; }

int BuildBVH(const std::vector<std::pair<int, B3>> &primsIn, int maxPrimsInNode, std::vector<wf_bvh_node> *nodes, std::vector<int32_t> *orderedPrims, int splitMethod) {
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
    BuildNode *root = splitMethod == 1 ? bld.BuildHLBVH(prims.data(), nAll) : bld.Build(prims.data(), nAll);
    std::vector<wf_bvh_node> local(bld.totalNodes, wf_bvh_node{});
    int offset = 0;
    bld.Flatten(root, &local, &offset);
    for (wf_bvh_node &n : local) n.offset += n.nprims > 0 ? primBase : nodeBase;  // leaf: primitivesOffset, interior: secondChildOffset
    nodes->insert(nodes->end(), local.begin(), local.end());
    orderedPrims->insert(orderedPrims->end(), ordered.begin(), ordered.end());
    return nodeBase;
}

}  // namespace wf
