// lightbvh_build.cpp — host build of the light BVH used for next-event estimation.
// Restates BVHLightSampler::BVHLightSampler / buildBVH (lightsamplers.cpp:105-232), EvaluateCost
// (lightsamplers.h:378-392), LightBounds Union (lights.h:137-153), DirectionCone Union
// (util/vecmath.cpp:56-83) and CompactLightBounds (lightsamplers.h:101-230).  Nodes are emitted in the
// same depth-first order with the same 32-byte content; each light's bit trail is stored in wf_light.
#include "scene.h"

#include <algorithm>
#include <cmath>

namespace wf {
namespace {

struct Cone { V3 w{0, 0, 0}; float cosTheta = WF_INFINITY; bool IsEmpty() const { return cosTheta == WF_INFINITY; } };
Cone MakeCone(V3 w, float c) { return Cone{Normalize(w), c}; }

Cone UnionCone(const Cone &a, const Cone &b) {
    if (a.IsEmpty()) return b;
    if (b.IsEmpty()) return a;
    float theta_a = SafeACos(a.cosTheta), theta_b = SafeACos(b.cosTheta);
    float theta_d = AngleBetween(a.w, b.w);
    if (std::min(theta_d + theta_b, Pi) <= theta_a) return a;
    if (std::min(theta_d + theta_a, Pi) <= theta_b) return b;
    float theta_o = (theta_a + theta_d + theta_b) / 2;
    if (theta_o >= Pi) return MakeCone(V3{0, 0, 1}, -1);
    float theta_r = theta_o - theta_a;
    V3 wr = Cross(a.w, b.w);
    if (LengthSquared(wr) == 0) return MakeCone(V3{0, 0, 1}, -1);
    V3 w = Rotate(Degrees(theta_r), wr).Vector(a.w);
    return MakeCone(w, std::cos(theta_o));
}

LightBoundsH MakeLB(const B3 &b, V3 w, float phi, float cosTheta_o, float cosTheta_e, bool twoSided) {
    LightBoundsH lb;
    lb.bounds = b; lb.w = Normalize(w); lb.phi = phi;
    lb.cosTheta_o = cosTheta_o; lb.cosTheta_e = cosTheta_e; lb.twoSided = twoSided;
    return lb;
}
LightBoundsH UnionLB(const LightBoundsH &a, const LightBoundsH &b) {
    if (a.phi == 0) return b;
    if (b.phi == 0) return a;
    Cone cone = UnionCone(MakeCone(a.w, a.cosTheta_o), MakeCone(b.w, b.cosTheta_o));
    float cosTheta_o = cone.cosTheta;
    float cosTheta_e = std::min(a.cosTheta_e, b.cosTheta_e);
    return MakeLB(Union(a.bounds, b.bounds), cone.w, a.phi + b.phi, cosTheta_o, cosTheta_e, a.twoSided | b.twoSided);
}

float EvaluateCost(const LightBoundsH &b, const B3 &bounds, int dim) {
    float theta_o = std::acos(b.cosTheta_o), theta_e = std::acos(b.cosTheta_e);
    float theta_w = std::min(theta_o + theta_e, Pi);
    float sinTheta_o = SafeSqrt(1 - Sqr(b.cosTheta_o));
    float M_omega = 2 * Pi * (1 - b.cosTheta_o) +
                    Pi / 2 * (2 * theta_w * sinTheta_o - std::cos(theta_o - 2 * theta_w) - 2 * theta_o * sinTheta_o + b.cosTheta_o);
    float Kr = MaxComponentValue(bounds.Diagonal()) / bounds.Diagonal()[dim];
    return b.phi * M_omega * Kr * b.bounds.SurfaceArea();
}

unsigned QuantizeCos(float c) { return (unsigned)std::floor(32767.f * ((c + 1) / 2)); }
float QuantizeBounds(float c, float mn, float mx) {
    if (mn == mx) return 0;
    return 65535.f * Clamp((c - mn) / (mx - mn), 0.f, 1.f);
}
wf_light_bvh_node Compact(const LightBoundsH &lb, const B3 &allb) {
    wf_light_bvh_node n{};
    OctahedralFromVector(Normalize(lb.w), &n.w_oct[0], &n.w_oct[1]);
    n.phi = lb.phi;
    n.cos_bits = (QuantizeCos(lb.cosTheta_o) & 0x7fff) | ((QuantizeCos(lb.cosTheta_e) & 0x7fff) << 15) | ((lb.twoSided ? 1u : 0u) << 30);
    for (int c = 0; c < 3; ++c) {
        n.qb[0][c] = (uint16_t)std::floor(QuantizeBounds(lb.bounds.pMin[c], allb.pMin[c], allb.pMax[c]));
        n.qb[1][c] = (uint16_t)std::ceil(QuantizeBounds(lb.bounds.pMax[c], allb.pMin[c], allb.pMax[c]));
    }
    return n;
}

struct Builder {
    std::vector<std::pair<int, LightBoundsH>> bvhLights;
    B3 allLightBounds;
    std::vector<wf_light_bvh_node> *nodes;
    std::vector<wf_light> *lights;

    std::pair<int, LightBoundsH> Build(int start, int end, uint32_t bitTrail, int depth) {
        if (end - start == 1) {
            int nodeIndex = (int)nodes->size();
            wf_light_bvh_node cb = Compact(bvhLights[start].second, allLightBounds);
            int lightIndex = bvhLights[start].first;
            cb.child_or_light = (uint32_t)lightIndex | (1u << 31);
            nodes->push_back(cb);
            (*lights)[lightIndex].bit_trail = (int32_t)bitTrail;
            return {nodeIndex, bvhLights[start].second};
        }
        B3 bounds, centroidBounds;
        for (int i = start; i < end; ++i) {
            const LightBoundsH &lb = bvhLights[i].second;
            bounds = Union(bounds, lb.bounds);
            centroidBounds = Union(centroidBounds, lb.Centroid());
        }
        float minCost = WF_INFINITY;
        int minCostSplitBucket = -1, minCostSplitDim = -1;
        constexpr int nBuckets = 12;
        for (int dim = 0; dim < 3; ++dim) {
            if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) continue;
            LightBoundsH bucketLightBounds[nBuckets];
            for (int i = start; i < end; ++i) {
                V3 pc = bvhLights[i].second.Centroid();
                int b = nBuckets * centroidBounds.Offset(pc)[dim];
                if (b == nBuckets) b = nBuckets - 1;
                bucketLightBounds[b] = UnionLB(bucketLightBounds[b], bvhLights[i].second);
            }
            float cost[nBuckets - 1];
            for (int i = 0; i < nBuckets - 1; ++i) {
                LightBoundsH b0, b1;
                for (int j = 0; j <= i; ++j) b0 = UnionLB(b0, bucketLightBounds[j]);
                for (int j = i + 1; j < nBuckets; ++j) b1 = UnionLB(b1, bucketLightBounds[j]);
                cost[i] = EvaluateCost(b0, bounds, dim) + EvaluateCost(b1, bounds, dim);
            }
            for (int i = 1; i < nBuckets - 1; ++i)
                if (cost[i] > 0 && cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; minCostSplitDim = dim; }
        }
        int mid;
        if (minCostSplitDim == -1) mid = (start + end) / 2;
        else {
            auto *pmid = std::partition(&bvhLights[start], &bvhLights[end - 1] + 1, [=](const std::pair<int, LightBoundsH> &l) {
                int b = nBuckets * centroidBounds.Offset(l.second.Centroid())[minCostSplitDim];
                if (b == nBuckets) b = nBuckets - 1;
                return b <= minCostSplitBucket;
            });
            mid = int(pmid - &bvhLights[0]);
            if (mid == start || mid == end) mid = (start + end) / 2;
        }
        int nodeIndex = (int)nodes->size();
        nodes->push_back(wf_light_bvh_node{});
        auto child0 = Build(start, mid, bitTrail, depth + 1);
        auto child1 = Build(mid, end, bitTrail | (1u << depth), depth + 1);
        LightBoundsH lb = UnionLB(child0.second, child1.second);
        wf_light_bvh_node cb = Compact(lb, allLightBounds);
        cb.child_or_light = (uint32_t)child1.first;
        (*nodes)[nodeIndex] = cb;
        return {nodeIndex, lb};
    }
};

}  // namespace

void BuildLightBVH(const std::vector<std::pair<int, LightBoundsH>> &bvhLightsIn, const B3 &allLightBounds,
                   std::vector<wf_light_bvh_node> *nodes, std::vector<wf_light> *lights) {
    nodes->clear();
    if (bvhLightsIn.empty()) return;
    Builder b;
    b.bvhLights = bvhLightsIn;
    b.allLightBounds = allLightBounds;
    b.nodes = nodes;
    b.lights = lights;
    b.Build(0, (int)b.bvhLights.size(), 0, 0);
}

}  // namespace wf
