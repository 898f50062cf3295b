// bvh_build.cpp — host BVH builders producing the reference's LinearBVHNode layout.
// Restates BVHAggregate: constructor cpu/aggregates.cpp:140-196; the SAH path buildRecursive :198-387 (12 buckets, leaf when
// cost says so and n <= maxPrimsInNode); the HLBVH path buildHLBVH :389-447 + emitLBVH :449-503 + buildUpperSAH :626-722
// (Morton codes of the centroids, stable radix sort — on the GPU through wf_morton_sort when a device is there —, one LBVH
// treelet per 12-bit Morton prefix, SAH over the treelet roots); flattenBVH :505-521.
// The SAH build is task-parallel like the reference's (cpu/aggregates.cpp:366-381 builds the two children of a large span
// concurrently) and, unlike it, deterministic: a leaf's primitives are written at the span's own offset — in sequential depth-first
// order leaf k's first primitive IS position k of the partitioned span array — instead of at an atomic counter, so the node and
// primitive arrays are the sequential build's for any thread count (round 3: 2.6 s -> see DESIGN.md 4.3 on the 10 M-triangle scene).
#include "scene.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>

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
    int nodeCount = 1;   // nodes of the subtree (flattening lays subtrees out at known offsets: in parallel)
};
MortonSortFn g_mortonSort = nullptr;
SahBuildFn g_sahBuild = nullptr;

struct Builder {
    int maxPrimsInNode;
    int splitMethod = 0;   // 0 SAH, 2 Middle, 3 EqualCounts (1 = HLBVH: BuildHLBVH)
    // raw node storage (not value-initialised: 450 MB for the 4 M-primitive top level), in chunks that are added on demand: a tree has
    // 2 n - 1 nodes, but every helper thread abandons the tail of its last run of 1024 when it exits, and the number of helper threads
    // grows with n (a fixed pool of 2 n + 1 M nodes could be overrun by a large or skewed build — silently)
    std::vector<BuildNode *> chunks;
    size_t chunkCap = 0, chunkUsed = 0, firstChunk = 0;
    std::mutex chunkMutex;
    ~Builder() { for (BuildNode *c : chunks) free(c); }
    BuildNode *NewRun() {   // 1024 nodes for one thread
        std::lock_guard<std::mutex> lock(chunkMutex);
        if (chunkUsed + 1024 > chunkCap) {
            chunkCap = std::max<size_t>(firstChunk, 1u << 20);
            BuildNode *c = (BuildNode *)malloc(chunkCap * sizeof(BuildNode));
            if (!c) throw SceneError("Error: out of memory for the BVH build's nodes");
            chunks.push_back(c);
            chunkUsed = 0;
        }
        BuildNode *r = chunks.back() + chunkUsed;
        chunkUsed += 1024;
        return r;
    }
    std::vector<int32_t> *ordered;
    std::atomic<int> totalNodes{0};

    static inline std::atomic<uint64_t> nextId{1};
    const uint64_t id = nextId++;
    // nodes are handed out in runs of 1024 per thread (one shared counter touched by 256 builder threads for each of 8 M nodes is a
    // contended cache line: measured 27 s for the 4 M-primitive top level on the 256-core box, against 1.3 s sequentially)
    BuildNode *NewNode() {
        static thread_local uint64_t owner = 0;   // (a builder's id, not its address: successive builders share stack addresses)
        static thread_local BuildNode *next = nullptr, *end = nullptr;
        if (owner != id || next == end) {
            owner = id;
            next = NewRun();
            end = next + 1024;
        }
        return new (next++) BuildNode();
    }
    // task parallelism of the SAH build: a span larger than kParallelSpan hands its first child to a helper thread while helpers are left
    static constexpr int kParallelSpan = 16 * 1024;
    static constexpr int kChunkedSpan = 256 * 1024;
    std::atomic<int> helpersLeft{0};

    // (SAH path) prims = the span of the partitioned array that starts at position `first` of the whole array
    void InitLeaf(BuildNode *node, BVHPrim *prims, int n, const B3 &bounds, int first) {
        for (int i = 0; i < n; ++i) (*ordered)[first + i] = prims[i].index;
        node->firstPrimOffset = first;
        node->nPrimitives = n;
        node->bounds = bounds;
    }

    BuildNode *Build(BVHPrim *prims, int n, int first) {
        BuildNode *node = NewNode();
        // (spans of millions of primitives — the first levels of a large tree, where there are not yet enough subtrees to keep the
        // helpers busy — run their reductions in chunks on helper threads: unions and counts are exact, so the result is the loop's)
        const int nChunks = n >= kChunkedSpan ? std::min(16, std::max(1, helpersLeft.load())) : 1;
        auto chunked = [&](auto &&body) {   // body(chunk, i0, i1)
            if (nChunks <= 1) { body(0, 0, n); return; }
            std::vector<std::thread> th;
            for (int c = 1; c < nChunks; ++c) th.emplace_back([&, c] { body(c, (int)((long long)n * c / nChunks), (int)((long long)n * (c + 1) / nChunks)); });
            body(0, 0, (int)((long long)n / nChunks));
            for (auto &t : th) t.join();
        };
        B3 bounds, centroidBounds;
        {
            B3 pb[16], pc[16];
            chunked([&](int c, int i0, int i1) {
                B3 b, cb;
                for (int i = i0; i < i1; ++i) { b = Union(b, prims[i].bounds); cb = Union(cb, prims[i].Centroid()); }
                pb[c] = b; pc[c] = cb;
            });
            for (int c = 0; c < nChunks; ++c) { bounds = Union(bounds, pb[c]); centroidBounds = Union(centroidBounds, pc[c]); }
        }
        if (bounds.SurfaceArea() == 0 || n == 1) { InitLeaf(node, prims, n, bounds, first); return node; }
        int dim = centroidBounds.MaxDimension();
        if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) { InitLeaf(node, prims, n, bounds, first); return node; }
        int mid = n / 2;
        if (splitMethod == 2 || splitMethod == 3) {
            // SplitMethod::Middle / EqualCounts (cpu/aggregates.cpp:239-263): a leaf only for one primitive or degenerate bounds (above)
            bool parted = false;
            if (splitMethod == 2) {
                const float pmid = (centroidBounds.pMin[dim] + centroidBounds.pMax[dim]) / 2;
                BVHPrim *midIter = std::partition(prims, prims + n, [dim, pmid](const BVHPrim &pi) { return pi.Centroid()[dim] < pmid; });
                mid = int(midIter - prims);
                parted = mid != 0 && mid != n;   // "for lots of prims with large overlapping bounding boxes, this may fail to partition": EqualCounts then
            }
            if (!parted) {
                mid = n / 2;
                std::nth_element(prims, prims + mid, prims + n,
                                 [dim](const BVHPrim &a, const BVHPrim &b) { return a.Centroid()[dim] < b.Centroid()[dim]; });
            }
        } else if (n <= 2) {
            std::nth_element(prims, prims + mid, prims + n,
                             [dim](const BVHPrim &a, const BVHPrim &b) { return a.Centroid()[dim] < b.Centroid()[dim]; });
        } else {
            constexpr int nBuckets = 12;
            struct Bucket { int count = 0; B3 bounds; } buckets[nBuckets];
            if (nChunks <= 1) {
                for (int i = 0; i < n; ++i) {
                    int b = nBuckets * centroidBounds.Offset(prims[i].Centroid())[dim];
                    if (b == nBuckets) b = nBuckets - 1;
                    buckets[b].count++;
                    buckets[b].bounds = Union(buckets[b].bounds, prims[i].bounds);
                }
            } else {
                std::vector<std::array<Bucket, nBuckets>> part(nChunks);
                chunked([&](int c, int i0, int i1) {
                    std::array<Bucket, nBuckets> &bk = part[c];
                    for (int i = i0; i < i1; ++i) {
                        int b = nBuckets * centroidBounds.Offset(prims[i].Centroid())[dim];
                        if (b == nBuckets) b = nBuckets - 1;
                        bk[b].count++;
                        bk[b].bounds = Union(bk[b].bounds, prims[i].bounds);
                    }
                });
                for (int c = 0; c < nChunks; ++c)
                    for (int b = 0; b < nBuckets; ++b) { buckets[b].count += part[c][b].count; buckets[b].bounds = Union(buckets[b].bounds, part[c][b].bounds); }
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
            if (minCostSplitBucket < 0) {
                // every cost overflowed or is NaN (scene_build refuses such extents before they get here; the reference would recurse
                // on the unsplit span for ever): split the span in the middle
                mid = n / 2;
                std::nth_element(prims, prims + mid, prims + n,
                                 [dim](const BVHPrim &a, const BVHPrim &b) { return a.Centroid()[dim] < b.Centroid()[dim]; });
            } else if (n > maxPrimsInNode || minCost < leafCost) {
                BVHPrim *midIter = std::partition(prims, prims + n, [=](const BVHPrim &bp) {
                    int b = nBuckets * centroidBounds.Offset(bp.Centroid())[dim];
                    if (b == nBuckets) b = nBuckets - 1;
                    return b <= minCostSplitBucket;
                });
                mid = int(midIter - prims);
            } else { InitLeaf(node, prims, n, bounds, first); return node; }
        }
        BuildNode *c0 = nullptr, *c1 = nullptr;
        if (n > kParallelSpan && helpersLeft.fetch_sub(1) > 0) {
            std::thread helper([&] { c0 = Build(prims, mid, first); });
            c1 = Build(prims + mid, n - mid, first + mid);
            helper.join();
            helpersLeft.fetch_add(1);
        } else {
            if (n > kParallelSpan) helpersLeft.fetch_add(1);   // (the failed reservation)
            c0 = Build(prims, mid, first);
            c1 = Build(prims + mid, n - mid, first + mid);
        }
        node->children[0] = c0;
        node->children[1] = c1;
        node->bounds = Union(c0->bounds, c1->bounds);
        node->splitAxis = dim;
        node->nPrimitives = 0;
        node->nodeCount = 1 + c0->nodeCount + c1->nodeCount;
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
        node->nodeCount = 1 + c0->nodeCount + c1->nodeCount;
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
        node->nodeCount = 1 + c0->nodeCount + c1->nodeCount;
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

    // flattenBVH (cpu/aggregates.cpp:505-521): depth-first, first child right behind its parent; a subtree of k nodes occupies
    // [at, at + k), so the two children of a large subtree are laid out concurrently
    void Flatten(const BuildNode *node, wf_bvh_node *out, int at) {
        wf_bvh_node *ln = &out[at];
        *ln = wf_bvh_node{};
        for (int c = 0; c < 3; ++c) { ln->bmin[c] = node->bounds.pMin[c]; ln->bmax[c] = node->bounds.pMax[c]; }
        if (node->nPrimitives > 0) {
            ln->offset = node->firstPrimOffset;
            ln->nprims = (uint16_t)node->nPrimitives;
            ln->axis = 0;
            return;
        }
        ln->axis = (uint8_t)node->splitAxis;
        ln->nprims = 0;
        const int second = at + 1 + node->children[0]->nodeCount;
        ln->offset = second;
        if (node->nodeCount > 2 * kParallelSpan && helpersLeft.fetch_sub(1) > 0) {
            std::thread helper([&] { Flatten(node->children[0], out, at + 1); });
            Flatten(node->children[1], out, second);
            helper.join();
            helpersLeft.fetch_add(1);
        } else {
            if (node->nodeCount > 2 * kParallelSpan) helpersLeft.fetch_add(1);
            Flatten(node->children[0], out, at + 1);
            Flatten(node->children[1], out, second);
        }
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
void SetSahBuild(SahBuildFn fn) { g_sahBuild = fn; }

int BuildBVH(const std::vector<std::pair<int, B3>> &primsIn, int maxPrimsInNode, std::vector<wf_bvh_node> *nodes, std::vector<int32_t> *orderedPrims, int splitMethod, bool forceHost) {
    const int nAll = (int)primsIn.size();
    if (nAll == 0) return -1;
    std::vector<BVHPrim> prims(nAll);
    for (int i = 0; i < nAll; ++i) { prims[i].index = primsIn[i].first; prims[i].bounds = primsIn[i].second; }
    const int nodeBase = (int)nodes->size(), primBase = (int)orderedPrims->size();
    const bool timing = getenv("WF_LOAD_TIMING") != nullptr && nAll > 1000000;
    if (splitMethod == 0 && g_sahBuild && !forceHost && !getenv("WF_HOST_BVH_BUILD")) {
        // device path (csrc/hip/wf_bvh_build.hip): the same nodes and primitive order as the recursion below
        int minPrims = 200000;
        if (const char *e = getenv("WF_DEVICE_BVH_MIN")) minPrims = atoi(e);
        if (nAll >= minPrims) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<float> b6(6 * (size_t)nAll);
            for (int i = 0; i < nAll; ++i) {
                const B3 &b = primsIn[i].second;
                float *o = &b6[6 * (size_t)i];
                o[0] = b.pMin.x; o[1] = b.pMin.y; o[2] = b.pMin.z; o[3] = b.pMax.x; o[4] = b.pMax.y; o[5] = b.pMax.z;
            }
            std::unique_ptr<wf_bvh_node[]> local(new wf_bvh_node[2 * (size_t)nAll]);   // (not value-initialised: 256 MB for the 4 M-primitive top level)
            std::unique_ptr<int32_t[]> order(new int32_t[nAll]);
            int32_t nNodes = 0;
            const int rc = g_sahBuild(nAll, b6.data(), std::min(255, maxPrimsInNode), local.get(), order.get(), &nNodes);
            if (rc != 0 && rc != -1 && getenv("WF_LOAD_TIMING")) fprintf(stderr, "[load]   BuildBVH(%d prims): device build gave up (%d), host build instead\n", nAll, rc);
            if (rc == 0 && nNodes > 0) {
                if (primBase != 0 || nodeBase != 0)
                    for (int i = 0; i < nNodes; ++i) local[i].offset += local[i].nprims > 0 ? primBase : nodeBase;
                nodes->insert(nodes->end(), local.get(), local.get() + nNodes);
                const size_t at = orderedPrims->size();
                orderedPrims->resize(at + nAll);
                for (int i = 0; i < nAll; ++i) (*orderedPrims)[at + i] = primsIn[order[i]].first;
                if (timing) fprintf(stderr, "[load]   BuildBVH(%d prims): device build %.3f s\n", nAll, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                return nodeBase;
            }
        }
    }
    std::vector<int32_t> ordered;
    if (splitMethod == 1) ordered.reserve(nAll);   // (the HLBVH path appends leaf by leaf)
    else ordered.assign(nAll, -1);
    Builder bld;
    bld.maxPrimsInNode = std::min(255, maxPrimsInNode);
    bld.splitMethod = splitMethod;
    bld.firstChunk = 2 * (size_t)nAll + 64 * 1024;   // the first chunk holds a whole tree plus some abandoned run tails; more chunks follow on demand
    bld.ordered = &ordered;
    int threads = (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("WF_BUILD_THREADS")) threads = atoi(e);
    bld.helpersLeft = std::max(0, std::min(threads, 256) - 1);
    auto t0 = std::chrono::steady_clock::now();
    BuildNode *root = splitMethod == 1 ? bld.BuildHLBVH(prims.data(), nAll) : bld.Build(prims.data(), nAll, 0);
    auto t1 = std::chrono::steady_clock::now();
    if (splitMethod == 1 && root->nodeCount != bld.totalNodes) return -1;
    bld.totalNodes = root->nodeCount;
    std::vector<wf_bvh_node> local((size_t)bld.totalNodes);
    bld.Flatten(root, local.data(), 0);
    if (timing) fprintf(stderr, "[load]   BuildBVH(%d prims): build %.3f s, flatten %.3f s\n", nAll, std::chrono::duration<double>(t1 - t0).count(),
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    for (wf_bvh_node &n : local) n.offset += n.nprims > 0 ? primBase : nodeBase;  // leaf: primitivesOffset, interior: secondChildOffset
    nodes->insert(nodes->end(), local.begin(), local.end());
    orderedPrims->insert(orderedPrims->end(), ordered.begin(), ordered.end());
    return nodeBase;
}

}  // namespace wf
