// wf_bvh_build.hip — the SAH BVH build on the device: BVHAggregate::buildRecursive (cpu/aggregates.cpp:198-387) + flattenBVH
// (:505-521), reproduced NODE FOR NODE — the tree, its depth-first layout and the order of the primitives inside every leaf are the ones the
// reference's sequential recursion (and csrc/host/bvh_build.cpp) produce, so either builder can feed the traversal kernels and the parity contract
// (ties are resolved in this tree's order) holds.
//
// The recursion is turned into a level-synchronous build over the array of primitive positions:
//   * a node is a span [start, start + n) of positions; `idx[pos]` = which input primitive sits there, `nodeOf[pos]` = the open node it belongs to;
//   * per level, for nodes of more than SMALL primitives: bounds and centroid bounds by a segmented wave reduction + one atomic per run (float
//     min / max are exact, so the order of the reduction does not matter), the 12 bucket counts / bounds the same way, the SAH decision by one
//     thread per node with the reference's float expressions in the reference's order, and std::partition's PERMUTATION — libstdc++'s
//     bidirectional __partition swaps the k-th misplaced element from the left with the k-th misplaced element from the right and touches
//     nothing else — by one prefix sum over two flags per position;
//   * nodes of at most SMALL primitives are finished by one thread each per level (bounds, buckets, decision, the two-pointer partition itself);
//   * two-primitive nodes: std::nth_element on two elements is an insertion sort (swap iff the second centroid is smaller);
//   * when no node is open any more: subtree sizes bottom-up and depth-first positions top-down over the level ranges (a subtree of k nodes
//     occupies [at, at + k), first child at + 1), then the LinearBVHNode array is written.
// A leaf's primitives keep their span's positions (firstPrimOffset = start), which is what the sequential build's orderedPrims counter yields.
// Signed zeros: the reference's Union keeps the zero it meets first; the atomics canonicalise to +0.  No consumer distinguishes them.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wf_abi.h"
#include "../common/wf_math.h"

using namespace wf;

namespace {
constexpr int SMALL = 32;       // nodes of at most this many primitives are handled by one thread
constexpr int NB = 12;          // SAH buckets (cpu/aggregates.cpp:269)
constexpr int BLK = 256;
enum { ST_OPEN = 0, ST_LEAF = 1, ST_SPLIT = 2, ST_BUCKETS = 3 };

struct Build {
    int n, maxPrims;
    const float *pb;       // [n][6] input bounds: min.xyz max.xyz
    int *idx, *nodeOf;
    uint8_t *bkt;
    unsigned long long *flags, *scan;   // low 32 bits: misplaced on the left, high 32: misplaced on the right
    int *badL, *badR;
    // nodes
    float *nb;             // [cap][12]: bounds min, bounds max, centroid min, centroid max
    int *nStart, *nCount, *nChild, *nAxis, *nState, *nSplit, *nMid, *nSize, *nPos;
    // buckets of the open large nodes of one level, indexed by the slot k_decide hands out (kept in nSize until the flattening)
    int *bCount;           // [levelCap][12]
    float *bBounds;        // [levelCap][12][6]
    int *counters;         // [0] nodes allocated, [1] nodes waiting for buckets / a partition this level, [2] error flag
};

__device__ inline void AtomicMinF(float *a, float v) {
    v += 0.f;
    if (v >= 0) atomicMin((int *)a, __float_as_int(v));
    else atomicMax((unsigned *)a, __float_as_uint(v));
}
__device__ inline void AtomicMaxF(float *a, float v) {
    v += 0.f;
    if (v >= 0) atomicMax((int *)a, __float_as_int(v));
    else atomicMin((unsigned *)a, __float_as_uint(v));
}
__device__ inline B3 PrimBounds(const Build &b, int prim) {
    const float *p = b.pb + 6 * (size_t)prim;
    B3 r;
    r.pMin = V3{p[0], p[1], p[2]};
    r.pMax = V3{p[3], p[4], p[5]};
    return r;
}
__device__ inline V3 Centroid(const B3 &b) { return .5f * b.pMin + .5f * b.pMax; }
__device__ inline int BucketOf(const B3 &centroidBounds, int dim, V3 c) {
    int k = NB * centroidBounds.Offset(c)[dim];
    if (k == NB) k = NB - 1;
    return k;
}
// the split decision of buildRecursive's SAH branch (cpu/aggregates.cpp:268-330): returns the bucket to split at, or -1 for a leaf
__device__ inline int ChooseSplit(const int *count, const B3 *bounds, const B3 &nodeBounds, int n, int maxPrims) {
    constexpr int nSplits = NB - 1;
    float costs[nSplits] = {};
    int countBelow = 0;
    B3 boundBelow;
    for (int i = 0; i < nSplits; ++i) {
        boundBelow = Union(boundBelow, bounds[i]);
        countBelow += count[i];
        costs[i] += countBelow * boundBelow.SurfaceArea();
    }
    int countAbove = 0;
    B3 boundAbove;
    for (int i = nSplits; i >= 1; --i) {
        boundAbove = Union(boundAbove, bounds[i]);
        countAbove += count[i];
        costs[i - 1] += countAbove * boundAbove.SurfaceArea();
    }
    int minCostSplitBucket = -1;
    float minCost = WF_INFINITY;
    for (int i = 0; i < nSplits; ++i)
        if (costs[i] < minCost) { minCost = costs[i]; minCostSplitBucket = i; }
    float leafCost = n;
    minCost = 1.f / 2.f + minCost / nodeBounds.SurfaceArea();
    if (n > maxPrims || minCost < leafCost) return minCostSplitBucket < 0 ? -2 : minCostSplitBucket;
    return -1;
}
__device__ inline void LoadNodeBounds(const Build &b, int node, B3 *bounds, B3 *cb) {
    const float *p = b.nb + 12 * (size_t)node;
    bounds->pMin = V3{p[0], p[1], p[2]}; bounds->pMax = V3{p[3], p[4], p[5]};
    cb->pMin = V3{p[6], p[7], p[8]}; cb->pMax = V3{p[9], p[10], p[11]};
}
__device__ inline void StoreNodeBounds(const Build &b, int node, const B3 &bounds, const B3 &cb) {
    float *p = b.nb + 12 * (size_t)node;
    p[0] = bounds.pMin.x; p[1] = bounds.pMin.y; p[2] = bounds.pMin.z; p[3] = bounds.pMax.x; p[4] = bounds.pMax.y; p[5] = bounds.pMax.z;
    p[6] = cb.pMin.x; p[7] = cb.pMin.y; p[8] = cb.pMin.z; p[9] = cb.pMax.x; p[10] = cb.pMax.y; p[11] = cb.pMax.z;
}
__device__ inline void MakeLeaf(const Build &b, int node) {
    b.nState[node] = ST_LEAF;
    b.nChild[node] = -1;
    const int s = b.nStart[node], n = b.nCount[node];
    for (int i = 0; i < n; ++i) b.nodeOf[s + i] = -1;
}
// children of `node`, which splits its span after `mid` primitives
__device__ inline void MakeChildren(const Build &b, int node, int dim, int mid) {
    const int c = atomicAdd(&b.counters[0], 2);
    const int s = b.nStart[node], n = b.nCount[node];
    b.nState[node] = ST_SPLIT;
    b.nAxis[node] = dim;
    b.nChild[node] = c;
    b.nMid[node] = mid;
    b.nStart[c] = s; b.nCount[c] = mid; b.nState[c] = ST_OPEN;
    b.nStart[c + 1] = s + mid; b.nCount[c + 1] = n - mid; b.nState[c + 1] = ST_OPEN;
}

__global__ void k_init(Build b) {
    for (int i = blockIdx.x * BLK + threadIdx.x; i < b.n; i += gridDim.x * BLK) { b.idx[i] = i; b.nodeOf[i] = 0; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { b.nStart[0] = 0; b.nCount[0] = b.n; b.nState[0] = ST_OPEN; b.counters[0] = 1; b.counters[1] = 0; b.counters[2] = 0; }
}
// large open nodes of the level: empty bounds
__global__ void k_level_clear(Build b, int l0, int l1) {
    for (int node = l0 + blockIdx.x * BLK + threadIdx.x; node < l1; node += gridDim.x * BLK) {
        if (b.nCount[node] <= SMALL) continue;
        B3 e;
        StoreNodeBounds(b, node, e, e);
    }
}
// a run of equal `node` values over adjacent lanes is reduced into its first lane
template <bool MIN>
__device__ inline float SegReduce(float v, int node, int lane) {
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_down(v, off);
        const int on = __shfl_down(node, off);
        if (lane + off < 64 && on == node) v = MIN ? fmin(v, o) : fmax(v, o);
    }
    return v;
}
__device__ inline int SegSum(int v, int node, int lane) {
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_down(v, off);
        const int on = __shfl_down(node, off);
        if (lane + off < 64 && on == node) v += o;
    }
    return v;
}
// bounds and centroid bounds of the large open nodes (cpu/aggregates.cpp:212-214, 235-238)
__global__ void k_bounds(Build b) {
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * BLK; base < b.n; base += gridDim.x * BLK) {
        const int pos = base + threadIdx.x;
        int node = pos < b.n ? b.nodeOf[pos] : -1;
        if (node >= 0 && b.nCount[node] <= SMALL) node = -1;
        B3 pbx, cb;
        if (node >= 0) {
            pbx = PrimBounds(b, b.idx[pos]);
            cb = Union(cb, Centroid(pbx));
        }
        float v[12] = {pbx.pMin.x, pbx.pMin.y, pbx.pMin.z, pbx.pMax.x, pbx.pMax.y, pbx.pMax.z, cb.pMin.x, cb.pMin.y, cb.pMin.z, cb.pMax.x, cb.pMax.y, cb.pMax.z};
        if (__ballot(node >= 0) == 0) continue;
        const int prev = __shfl_up(node, 1);
        const bool head = node >= 0 && (lane == 0 || prev != node);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const bool isMin = k < 3 || (k >= 6 && k < 9);
            v[k] = isMin ? SegReduce<true>(v[k], node, lane) : SegReduce<false>(v[k], node, lane);
        }
        if (head) {
            float *p = b.nb + 12 * (size_t)node;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const bool isMin = k < 3 || (k >= 6 && k < 9);
                if (isMin) AtomicMinF(p + k, v[k]); else AtomicMaxF(p + k, v[k]);
            }
        }
    }
}
// one thread per open node of the level: small nodes are finished here; large ones are classified (leaf / two-way split / buckets needed)
__global__ void k_decide(Build b, int l0, int l1) {
    for (int node = l0 + blockIdx.x * BLK + threadIdx.x; node < l1; node += gridDim.x * BLK) {
        const int s = b.nStart[node], n = b.nCount[node];
        B3 bounds, cb;
        if (n <= SMALL) {
            for (int i = 0; i < n; ++i) {
                const B3 p = PrimBounds(b, b.idx[s + i]);
                bounds = Union(bounds, p);
                cb = Union(cb, Centroid(p));
            }
            StoreNodeBounds(b, node, bounds, cb);
        } else LoadNodeBounds(b, node, &bounds, &cb);
        if (bounds.SurfaceArea() == 0 || n == 1) { MakeLeaf(b, node); continue; }
        const int dim = cb.MaxDimension();
        if (cb.pMax[dim] == cb.pMin[dim]) { MakeLeaf(b, node); continue; }
        if (n <= 2) {
            // std::nth_element(begin, begin + 1, end) on two elements: insertion sort
            const int i0 = b.idx[s], i1 = b.idx[s + 1];
            if (Centroid(PrimBounds(b, i1))[dim] < Centroid(PrimBounds(b, i0))[dim]) { b.idx[s] = i1; b.idx[s + 1] = i0; }
            MakeChildren(b, node, dim, 1);
            const int c = b.nChild[node];
            b.nodeOf[s] = c; b.nodeOf[s + 1] = c + 1;
            continue;
        }
        if (n > SMALL) {
            b.nState[node] = ST_BUCKETS;
            b.nAxis[node] = dim;
            const int slot = atomicAdd(&b.counters[1], 1);   // (at most n / (SMALL + 1) large nodes are open at once: levelCap)
            b.nSize[node] = slot;                            // (nSize is free until the flattening)
            int *bc = b.bCount + NB * (size_t)slot;
            float *bb = b.bBounds + 6 * NB * (size_t)slot;
            for (int k = 0; k < NB; ++k) {
                bc[k] = 0;
                for (int c = 0; c < 3; ++c) { bb[6 * k + c] = 3.402823466e+38f; bb[6 * k + 3 + c] = -3.402823466e+38f; }
            }
            continue;
        }
        // small node: buckets, decision and std::partition by this thread
        int count[NB] = {};
        B3 bbounds[NB];
        uint8_t kb[SMALL];
        for (int i = 0; i < n; ++i) {
            const B3 p = PrimBounds(b, b.idx[s + i]);
            const int k = BucketOf(cb, dim, Centroid(p));
            kb[i] = (uint8_t)k;
            count[k]++;
            bbounds[k] = Union(bbounds[k], p);
        }
        const int split = ChooseSplit(count, bbounds, bounds, n, b.maxPrims);
        if (split == -1) { MakeLeaf(b, node); continue; }
        // libstdc++ __partition (bidirectional): bits/stl_algo.h
        int first = 0, last = n;
        while (true) {
            while (true) {
                if (first == last) goto done;
                else if ((int)kb[first] <= split) ++first;
                else break;
            }
            --last;
            while (true) {
                if (first == last) goto done;
                else if (!((int)kb[last] <= split)) --last;
                else break;
            }
            { const int t = b.idx[s + first]; b.idx[s + first] = b.idx[s + last]; b.idx[s + last] = t; const uint8_t u = kb[first]; kb[first] = kb[last]; kb[last] = u; }
            ++first;
        }
    done:
        if (first == 0 || first == n) { atomicOr(&b.counters[2], 1); MakeLeaf(b, node); continue; }   // (the reference would recurse forever)
        MakeChildren(b, node, dim, first);
        const int c = b.nChild[node];
        for (int i = 0; i < n; ++i) b.nodeOf[s + i] = i < first ? c : c + 1;
    }
}
// bucket counts and bounds of the large nodes that need them (cpu/aggregates.cpp:272-283)
__global__ void k_buckets(Build b) {
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * BLK; base < b.n; base += gridDim.x * BLK) {
        const int pos = base + threadIdx.x;
        int node = pos < b.n ? b.nodeOf[pos] : -1;
        if (node >= 0 && b.nState[node] != ST_BUCKETS) node = -1;
        if (__ballot(node >= 0) == 0) continue;
        B3 p;
        int k = -1;
        if (node >= 0) {
            B3 bounds, cb;
            LoadNodeBounds(b, node, &bounds, &cb);
            p = PrimBounds(b, b.idx[pos]);
            k = BucketOf(cb, b.nAxis[node], Centroid(p));
            b.bkt[pos] = (uint8_t)k;
        }
        const int prev = __shfl_up(node, 1);
        const bool head = node >= 0 && (lane == 0 || prev != node);
        for (int q = 0; q < NB; ++q) {
            const bool mine = k == q;
            if (__ballot(mine) == 0) continue;
            B3 e;
            const B3 &src = mine ? p : e;
            const int cnt = SegSum(mine ? 1 : 0, node, lane);
            const float v0 = SegReduce<true>(src.pMin.x, node, lane), v1 = SegReduce<true>(src.pMin.y, node, lane), v2 = SegReduce<true>(src.pMin.z, node, lane);
            const float v3 = SegReduce<false>(src.pMax.x, node, lane), v4 = SegReduce<false>(src.pMax.y, node, lane), v5 = SegReduce<false>(src.pMax.z, node, lane);
            if (head && cnt > 0) {
                atomicAdd(b.bCount + NB * (size_t)b.nSize[node] + q, cnt);
                float *bb = b.bBounds + 6 * NB * (size_t)b.nSize[node] + 6 * q;
                AtomicMinF(bb + 0, v0); AtomicMinF(bb + 1, v1); AtomicMinF(bb + 2, v2);
                AtomicMaxF(bb + 3, v3); AtomicMaxF(bb + 4, v4); AtomicMaxF(bb + 5, v5);
            }
        }
    }
}
// the SAH decision of the large nodes
__global__ void k_sah(Build b, int l0, int l1) {
    for (int node = l0 + blockIdx.x * BLK + threadIdx.x; node < l1; node += gridDim.x * BLK) {
        if (b.nState[node] != ST_BUCKETS) continue;
        const int n = b.nCount[node];
        B3 bounds, cb;
        LoadNodeBounds(b, node, &bounds, &cb);
        int count[NB];
        B3 bbounds[NB];
        const int *bc = b.bCount + NB * (size_t)b.nSize[node];
        const float *bb = b.bBounds + 6 * NB * (size_t)b.nSize[node];
        for (int k = 0; k < NB; ++k) {
            count[k] = bc[k];
            bbounds[k].pMin = V3{bb[6 * k], bb[6 * k + 1], bb[6 * k + 2]};
            bbounds[k].pMax = V3{bb[6 * k + 3], bb[6 * k + 4], bb[6 * k + 5]};
        }
        const int split = ChooseSplit(count, bbounds, bounds, n, b.maxPrims);
        if (split == -1) { b.nSplit[node] = -1; continue; }   // leaf: nodeOf is cleared by k_assign
        int mid = 0;
        for (int k = 0; k <= split && k < NB; ++k) mid += count[k];
        if (split < 0 || mid == 0 || mid == n) { atomicOr(&b.counters[2], 1); b.nSplit[node] = -1; continue; }
        b.nSplit[node] = split;
        MakeChildren(b, node, b.nAxis[node], mid);
        b.nState[node] = ST_BUCKETS;   // (still to be partitioned: k_flags / k_assign look at it; k_assign sets ST_SPLIT)
    }
}
// std::partition's misplaced elements: on the left of the split point with a bucket above it, on the right with a bucket at or below it
__global__ void k_flags(Build b) {
    for (int pos = blockIdx.x * BLK + threadIdx.x; pos < b.n; pos += gridDim.x * BLK) {
        unsigned long long f = 0;
        const int node = b.nodeOf[pos];
        if (node >= 0 && b.nState[node] == ST_BUCKETS && b.nSplit[node] >= 0) {
            const bool pred = (int)b.bkt[pos] <= b.nSplit[node];
            const bool left = pos < b.nStart[node] + b.nMid[node];
            if (left && !pred) f = 1ull;
            else if (!left && pred) f = 1ull << 32;
        }
        b.flags[pos] = f;
    }
}
// k-th misplaced element from the left <-> k-th misplaced element from the right of the same node
__global__ void k_pairs(Build b) {
    for (int pos = blockIdx.x * BLK + threadIdx.x; pos < b.n; pos += gridDim.x * BLK) {
        const unsigned long long f = b.flags[pos];
        if (f == 0) continue;
        const int node = b.nodeOf[pos];
        const int s = b.nStart[node], e = s + b.nCount[node];
        const unsigned long long here = b.scan[pos];
        if (f == 1ull) b.badL[(unsigned)(here & 0xffffffffull)] = pos;
        else {
            // rank from the right end of the node = (misplaced-right elements of the node) - 1 - (those before pos)
            const unsigned rBefore = (unsigned)(here >> 32) - (unsigned)(b.scan[s] >> 32);
            const unsigned long long endScan = b.scan[e - 1] + b.flags[e - 1];
            const unsigned rTotal = (unsigned)(endScan >> 32) - (unsigned)(b.scan[s] >> 32);
            const unsigned slot = (unsigned)(b.scan[s] & 0xffffffffull) + (rTotal - 1 - rBefore);
            b.badR[slot] = pos;
        }
    }
}
__global__ void k_swap(Build b, int nPairs) {
    for (int j = blockIdx.x * BLK + threadIdx.x; j < nPairs; j += gridDim.x * BLK) {
        const int l = b.badL[j], r = b.badR[j];
        const int t = b.idx[l];
        b.idx[l] = b.idx[r];
        b.idx[r] = t;
    }
}
// positions of the large nodes decided this level: child ids, or -1 for a leaf
__global__ void k_assign(Build b) {
    for (int pos = blockIdx.x * BLK + threadIdx.x; pos < b.n; pos += gridDim.x * BLK) {
        const int node = b.nodeOf[pos];
        if (node < 0 || b.nState[node] != ST_BUCKETS) continue;
        if (b.nSplit[node] < 0) b.nodeOf[pos] = -1;
        else b.nodeOf[pos] = b.nChild[node] + (pos < b.nStart[node] + b.nMid[node] ? 0 : 1);
    }
}
__global__ void k_close(Build b, int l0, int l1) {
    for (int node = l0 + blockIdx.x * BLK + threadIdx.x; node < l1; node += gridDim.x * BLK) {
        if (b.nState[node] != ST_BUCKETS) continue;
        if (b.nSplit[node] < 0) { b.nState[node] = ST_LEAF; b.nChild[node] = -1; }
        else b.nState[node] = ST_SPLIT;
    }
}
// flattenBVH: subtree sizes (bottom-up), then depth-first positions (top-down) and the LinearBVHNode records
__global__ void k_sizes(Build b, int l0, int l1) {
    for (int node = l0 + blockIdx.x * BLK + threadIdx.x; node < l1; node += gridDim.x * BLK) {
        const int c = b.nChild[node];
        b.nSize[node] = c < 0 ? 1 : 1 + b.nSize[c] + b.nSize[c + 1];
    }
}
__global__ void k_emit(Build b, int l0, int l1, wf_bvh_node *out) {
    for (int node = l0 + blockIdx.x * BLK + threadIdx.x; node < l1; node += gridDim.x * BLK) {
        const int at = node == 0 ? 0 : b.nPos[node];
        const int c = b.nChild[node];
        wf_bvh_node ln{};
        const float *p = b.nb + 12 * (size_t)node;
        for (int k = 0; k < 3; ++k) { ln.bmin[k] = p[k]; ln.bmax[k] = p[3 + k]; }
        if (c < 0) {
            ln.offset = b.nStart[node];
            ln.nprims = (uint16_t)b.nCount[node];
            ln.axis = 0;
        } else {
            const int second = at + 1 + b.nSize[c];
            b.nPos[c] = at + 1;
            b.nPos[c + 1] = second;
            ln.offset = second;
            ln.nprims = 0;
            ln.axis = (uint8_t)b.nAxis[node];
        }
        out[at] = ln;
    }
}
inline int GridFor(long long n) { long long g = (n + BLK - 1) / BLK; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }
}  // namespace

#define CK(x) do { if ((x) != hipSuccess) { rc = -2; goto out; } } while (0)

extern "C" int wf_build_bvh_sah(int n, const float *bounds, int max_prims_in_node, wf_bvh_node *nodes_out, int32_t *order_out, int32_t *n_nodes_out) {
    int devCount = 0;
    if (hipGetDeviceCount(&devCount) != hipSuccess || devCount == 0) { (void)hipGetLastError(); return -1; }
    // The tables are built before the renderer's device is known, on whatever device is current for the calling thread — device 0 in a
    // fresh thread, for every rank of a multi-GPU job on one box.  WF_BUILD_DEVICE, else LOCAL_RANK (torch.distributed.run),
    // names the device of this process's build; an allocation failure below returns a negative code and the host builder takes over.
    {
        const char *e = getenv("WF_BUILD_DEVICE");
        if (!e) e = getenv("LOCAL_RANK");
        if (e && atoi(e) >= 0 && atoi(e) < devCount) (void)hipSetDevice(atoi(e));
    }
    if (n <= 0 || !bounds || !nodes_out || !order_out || !n_nodes_out) return -3;
    const size_t cap = 2 * (size_t)n + 2;
    const size_t levelCap = (size_t)n / (SMALL + 1) + 2;   // large open nodes of one level are disjoint spans of more than SMALL primitives
    Build b{};
    b.n = n;
    b.maxPrims = max_prims_in_node < 255 ? max_prims_in_node : 255;
    int rc = 0;
    // one device allocation, carved up (25 hipMalloc calls cost more than the build of a 100 k-primitive tree)
    std::vector<void *> allocs;
    char *arena = nullptr;
    size_t arenaUsed = 0, arenaBytes = 0;
    auto alloc = [&](size_t bytes) -> void * {
        const size_t at = arenaUsed;
        arenaUsed += (bytes + 255) & ~(size_t)255;
        return arena ? (void *)(arena + at) : nullptr;
    };
    void *scanTemp = nullptr;
    size_t scanBytes = 0;
    wf_bvh_node *dOut = nullptr;
    std::vector<std::pair<int, int>> levels;
    int h[3] = {0, 0, 0};
    const auto tStart = std::chrono::steady_clock::now();
    double tUpload = 0, tLevels = 0, tFlatten = 0;
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    {
        auto plus = rocprim::plus<unsigned long long>();
        if (rocprim::exclusive_scan(nullptr, scanBytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull, (size_t)n, plus, (hipStream_t)0) != hipSuccess) { rc = -2; goto out; }
        float *pb = nullptr;
        for (int pass = 0; pass < 2; ++pass) {   // pass 0 sizes the arena, pass 1 hands out the pointers
            arenaUsed = 0;
            pb = (float *)alloc(24 * (size_t)n);
            b.idx = (int *)alloc(4 * (size_t)n); b.nodeOf = (int *)alloc(4 * (size_t)n); b.bkt = (uint8_t *)alloc((size_t)n);
            b.flags = (unsigned long long *)alloc(8 * (size_t)n); b.scan = (unsigned long long *)alloc(8 * (size_t)n);
            b.badL = (int *)alloc(4 * ((size_t)n / 2 + 1)); b.badR = (int *)alloc(4 * ((size_t)n / 2 + 1));
            b.nb = (float *)alloc(48 * cap);
            int **ints[] = {&b.nStart, &b.nCount, &b.nChild, &b.nAxis, &b.nState, &b.nSplit, &b.nMid, &b.nSize, &b.nPos};
            for (int **p : ints) *p = (int *)alloc(4 * cap);
            b.bCount = (int *)alloc(4 * NB * levelCap);
            b.bBounds = (float *)alloc(4 * 6 * NB * levelCap);
            b.counters = (int *)alloc(16);
            dOut = (wf_bvh_node *)alloc(sizeof(wf_bvh_node) * cap);
            scanTemp = alloc(scanBytes ? scanBytes : 1);
            if (pass == 0) {
                arenaBytes = arenaUsed;
                if (hipMalloc((void **)&arena, arenaBytes) != hipSuccess) { arena = nullptr; rc = -2; goto out; }
                allocs.push_back(arena);
            }
        }
        b.pb = pb;
        CK(hipMemcpy(pb, bounds, 24 * (size_t)n, hipMemcpyHostToDevice));
        tUpload = since(tStart);
        const auto tL = std::chrono::steady_clock::now();
        const int gp = GridFor(n);
        hipLaunchKernelGGL(k_init, dim3(gp), dim3(BLK), 0, 0, b);
        int l0 = 0, l1 = 1;
        while (l1 > l0) {
            if (levels.size() > 8192) { rc = -4; goto out; }   // a degenerate input (one primitive peeled off per level): left to the host builder
            levels.emplace_back(l0, l1);
            const int gn = GridFor(l1 - l0);
            hipLaunchKernelGGL(k_level_clear, dim3(gn), dim3(BLK), 0, 0, b, l0, l1);
            hipLaunchKernelGGL(k_bounds, dim3(gp), dim3(BLK), 0, 0, b);
            hipLaunchKernelGGL(k_decide, dim3(gn), dim3(BLK), 0, 0, b, l0, l1);
            CK(hipMemcpy(h, b.counters, 12, hipMemcpyDeviceToHost));
            if (h[1] > 0) {
                if ((size_t)h[1] > levelCap) { rc = -5; goto out; }
                hipLaunchKernelGGL(k_buckets, dim3(gp), dim3(BLK), 0, 0, b);
                hipLaunchKernelGGL(k_sah, dim3(gn), dim3(BLK), 0, 0, b, l0, l1);
                hipLaunchKernelGGL(k_flags, dim3(gp), dim3(BLK), 0, 0, b);
                CK(rocprim::exclusive_scan(scanTemp, scanBytes, b.flags, b.scan, 0ull, (size_t)n, plus, (hipStream_t)0));
                unsigned long long lastScan = 0, lastFlag = 0;
                CK(hipMemcpy(&lastScan, b.scan + (n - 1), 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&lastFlag, b.flags + (n - 1), 8, hipMemcpyDeviceToHost));
                const unsigned long long tot = lastScan + lastFlag;
                const int nPairs = (int)(tot & 0xffffffffull);
                if (nPairs != (int)(tot >> 32)) { rc = -6; goto out; }
                if (nPairs > 0) {
                    hipLaunchKernelGGL(k_pairs, dim3(gp), dim3(BLK), 0, 0, b);
                    hipLaunchKernelGGL(k_swap, dim3(GridFor(nPairs)), dim3(BLK), 0, 0, b, nPairs);
                }
                hipLaunchKernelGGL(k_assign, dim3(gp), dim3(BLK), 0, 0, b);
                hipLaunchKernelGGL(k_close, dim3(gn), dim3(BLK), 0, 0, b, l0, l1);
                const int zero = 0;
                CK(hipMemcpy(b.counters + 1, &zero, 4, hipMemcpyHostToDevice));
                CK(hipMemcpy(h, b.counters, 12, hipMemcpyDeviceToHost));
            }
            if (h[2]) { rc = -7; goto out; }
            if ((size_t)h[0] > cap) { rc = -8; goto out; }
            l0 = l1;
            l1 = h[0];
        }
        const int total = h[0];
        tLevels = since(tL);
        const auto tF = std::chrono::steady_clock::now();
        for (int l = (int)levels.size() - 1; l >= 0; --l)
            hipLaunchKernelGGL(k_sizes, dim3(GridFor(levels[l].second - levels[l].first)), dim3(BLK), 0, 0, b, levels[l].first, levels[l].second);
        for (size_t l = 0; l < levels.size(); ++l)
            hipLaunchKernelGGL(k_emit, dim3(GridFor(levels[l].second - levels[l].first)), dim3(BLK), 0, 0, b, levels[l].first, levels[l].second, dOut);
        CK(hipDeviceSynchronize());
        tFlatten = since(tF);
        const auto tD = std::chrono::steady_clock::now();
        CK(hipMemcpy(nodes_out, dOut, sizeof(wf_bvh_node) * (size_t)total, hipMemcpyDeviceToHost));
        CK(hipMemcpy(order_out, b.idx, 4 * (size_t)n, hipMemcpyDeviceToHost));
        *n_nodes_out = total;
        if (getenv("WF_LOAD_TIMING") && n > 1000000)
            fprintf(stderr, "[load]   wf_build_bvh_sah(%d prims): %zu levels, %d nodes; alloc + upload %.3f s, levels %.3f s, flatten %.3f s, download %.3f s\n", n, levels.size(), total,
                    tUpload, tLevels, tFlatten, since(tD));
    }
out:
    for (void *p : allocs) if (p) (void)hipFree(p);
    if (rc != 0) (void)hipGetLastError();
    return rc;
}
