// oracle/ref_build/hip_aggregate_adapter.cpp — TEST INFRASTRUCTURE (ours), and the compiled form of INTEGRATION.md §2:
// the binding a pbrt-v4 maintainer would add to put the MI355X traversal under the reference's own
// WavefrontPathIntegrator.  `class HipAggregate` implements the reference's `WavefrontAggregate` interface
// (wavefront/integrator.h:32-54) on top of the C ABI of libwfhip.so / libwfhost.so (include/wf_abi.h, include/wf_host.h):
//
//   IntersectClosest  rays of the RayQueue -> wf_trace_closest_host (production two-level walk + near-tie re-trace)
//                     -> per hit the reference's own Triangle::InteractionFromIntersection + SetIntersectionProperties
//                     -> the reference's own EnqueueWorkAfterIntersection / EnqueueWorkAfterMiss (wavefront/intersect.h)
//   IntersectShadow   shadow rays -> wf_trace_any_host -> the reference's RecordShadowRayResult
//   Bounds            wf_aggregate_bounds
//   IntersectOneRandom  probe segments -> wf_trace_one_random_host (reservoir over the hits of the item's material) -> SubsurfaceInteraction
//   IntersectShadowTr  shadow rays + the items' wavelengths / Ld / r_u / r_l -> wf_trace_shadow_tr_host (TraceTransmittance on the GPU:
//                     the walk AND the ratio tracking through the scene's media) -> the contribution is added to the pixel sample
//
// Everything else — camera rays, samplers, materials, lights, film — stays the reference's CPU code, so the image must be
// the one `pbrt --wavefront` writes, bit for bit (tests/test_gpu_parity.py::test_reference_integrator_over_hip_aggregate).
// Scope (round 3): triangle meshes with or without alpha textures (the GPU walk makes the reference's stochastic alpha test itself),
// object instances (TransformedPrimitive: the hit's instance id selects the reference's primitive, whose transform takes the
// interaction to render space exactly as TransformedPrimitive::Intersect does), media, and (round 3, second step) spheres / disks /
// cylinders / bilinear patches / curves: the hit's primitive id selects the reference's primitive (matched by content), whose own
// Intersect rebuilds the interaction the reference would have built for that primitive (alpha recursion included).  Links the shimmed reference build (libpbrt_ref.a); nothing of the reference is modified —
// private members are reached with the test-only `#define private public`.
//   pbrt_hipagg [--spp N] [--outfile out.pfm] scene.pbrt
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <set>
#include <shared_mutex>
#include <sstream>
#include <string>
#include <thread>
#include <typeindex>
#include <typeinfo>
#include <unordered_map>
#include <variant>
#include <vector>

#define private public
#define protected public
#include <pbrt/pbrt.h>
#include <pbrt/cameras.h>
#include <pbrt/cpu/aggregates.h>
#include <pbrt/cpu/primitive.h>
#include <pbrt/shapes.h>
#include <pbrt/wavefront/aggregate.h>
#include <pbrt/wavefront/integrator.h>
#undef private
#undef protected
#include <pbrt/film.h>
#include <pbrt/options.h>
#include <pbrt/parser.h>
#include <pbrt/scene.h>
#include <pbrt/util/image.h>
#include <pbrt/util/parallel.h>
#include <pbrt/wavefront/intersect.h>

#include "../../include/wf_host.h"

using namespace pbrt;

class HipAggregate : public WavefrontAggregate {
  public:
    HipAggregate(CPUAggregate *cpu, wf_ctx *ctx, const wf_scene_desc *desc) : cpu(cpu), ctx(ctx) {
        // primitive id (global triangle id of the flat tables) -> the reference's primitive.  The reference registers its meshes
        // in whatever order its parallel shape creation finishes, so meshes are matched by content: triangle count + the bytes
        // of their render-space vertex positions (both sides apply the same transform arithmetic: bit-identical).
        const int nTriangles = desc->n_triangles;
        prims.assign(nTriangles, Primitive());
        auto key = [](int ntris, const float *p, int nverts) {
            uint64_t h = 1469598103934665603ull ^ (uint64_t)ntris;
            for (size_t i = 0; i < (size_t)nverts * 3; ++i) {
                float f = p[i] + 0.0f;  // -0 and +0 are the same position
                uint32_t u;
                std::memcpy(&u, &f, 4);
                h = (h ^ u) * 1099511628211ull;
            }
            return h;
        };
        // (the reference creates the shapes of emitters twice — once for the lights, scene.cpp:1290-1340, once for the
        // primitives — so a reference mesh may repeat; identical meshes on our side are handed out in turn)
        std::map<uint64_t, std::pair<std::vector<int>, size_t>> ours;  // key -> first global triangle ids, next to hand out
        for (int m = 0; m < desc->n_meshes; ++m)
            if (desc->meshes[m].ntris > 0) ours[key(desc->meshes[m].ntris, desc->P + 3 * (size_t)desc->meshes[m].first_vertex, desc->meshes[m].nverts)].first.push_back(desc->meshes[m].first_tri);
        std::vector<int> firstTri(Triangle::allMeshes->size(), -1);
        for (size_t m = 0; m < Triangle::allMeshes->size(); ++m) {
            const TriangleMesh *tm = (*Triangle::allMeshes)[m];
            std::vector<float> P(3 * (size_t)tm->nVertices);
            for (int v = 0; v < tm->nVertices; ++v) { P[3 * v] = tm->p[v].x; P[3 * v + 1] = tm->p[v].y; P[3 * v + 2] = tm->p[v].z; }
            auto it = ours.find(key(tm->nTriangles, P.data(), tm->nVertices));
            if (it == ours.end()) ErrorExit("pbrt_hipagg: a reference mesh (%d triangles) has no counterpart in the flat tables", tm->nTriangles);
            firstTri[m] = it->second.first[it->second.second++ % it->second.first.size()];
        }
        // quadric records (spheres, disks, cylinders, bilinear patches, curve segments) by content, like the meshes
        nTri = nTriangles;
        quadPrims.assign(desc->n_quadrics, Primitive());
        std::map<uint64_t, std::pair<std::vector<int>, size_t>> ourQuadrics;
        for (int q = 0; q < desc->n_quadrics; ++q) ourQuadrics[QuadricKeyOurs(desc->quadrics[q])].first.push_back(q);
        std::vector<const TransformedPrimitive *> refInstances;
        std::set<const void *> visitedDefs;
        std::function<void(Primitive)> visit = [&](Primitive p) {
            if (!p) return;
            if (p.Is<BVHAggregate>()) {
                for (Primitive q : p.Cast<BVHAggregate>()->primitives) visit(q);
                return;
            }
            if (p.Is<TransformedPrimitive>()) {
                const TransformedPrimitive *tp = p.Cast<TransformedPrimitive>();
                refInstances.push_back(tp);
                if (visitedDefs.insert(tp->primitive.ptr()).second) visit(tp->primitive);   // the definition's own primitives, once
                return;
            }
            Shape shape;
            if (p.Is<SimplePrimitive>()) shape = p.Cast<SimplePrimitive>()->shape;
            else if (p.Is<GeometricPrimitive>()) shape = p.Cast<GeometricPrimitive>()->shape;   // (alpha textures: tested by the GPU walk)
            else ErrorExit("pbrt_hipagg: animated primitives are outside this adapter's scope");
            if (!shape.Is<Triangle>()) {
                // spheres / disks / cylinders / bilinear patches / curves: the quadric record with the same content
                const uint64_t k = QuadricKeyRef(shape);
                auto it = ourQuadrics.find(k);
                if (it == ourQuadrics.end()) ErrorExit("pbrt_hipagg: a reference %s has no counterpart among the flat tables' quadric records", shape.ToString());
                const int q = it->second.first[it->second.second++ % it->second.first.size()];
                quadPrims[q] = p;
                const wf_mesh &mesh = desc->meshes[desc->quadrics[q].mesh];
                Material mat = p.Is<SimplePrimitive>() ? p.Cast<SimplePrimitive>()->material : p.Cast<GeometricPrimitive>()->material;
                materialIds[mat.ptr()] = mesh.material;
                if (p.Is<GeometricPrimitive>()) {
                    const GeometricPrimitive *g = p.Cast<GeometricPrimitive>();
                    if (g->mediumInterface.inside) mediumIds[g->mediumInterface.inside.ptr()] = mesh.medium_inside;
                    if (g->mediumInterface.outside) mediumIds[g->mediumInterface.outside.ptr()] = mesh.medium_outside;
                }
                return;
            }
            const Triangle *t = shape.Cast<Triangle>();
            int id = firstTri[t->meshIndex] + t->triIndex;
            CHECK(id >= 0 && id < nTriangles);
            prims[id] = p;
            Material mat = p.Is<SimplePrimitive>() ? p.Cast<SimplePrimitive>()->material : p.Cast<GeometricPrimitive>()->material;
            materialIds[mat.ptr()] = desc->meshes[desc->tri_mesh[id]].material;
            if (p.Is<GeometricPrimitive>()) {
                // the reference's Medium objects <-> the medium ids of the flat tables, through the surfaces that carry both
                const GeometricPrimitive *g = p.Cast<GeometricPrimitive>();
                const wf_mesh &mesh = desc->meshes[desc->tri_mesh[id]];
                if (g->mediumInterface.inside) mediumIds[g->mediumInterface.inside.ptr()] = mesh.medium_inside;
                if (g->mediumInterface.outside) mediumIds[g->mediumInterface.outside.ptr()] = mesh.medium_outside;
            }
        };
        visit(cpu->aggregate);
        // identical meshes (same geometry, possibly different materials) may have been handed out in a different order than
        // the primitives were created in: every triangle must still have found exactly one primitive
        for (const Primitive &p : prims)
            if (!p) ErrorExit("pbrt_hipagg: ambiguous mesh matching (identical meshes with different roles)");
        for (const Primitive &p : quadPrims)
            if (!p) ErrorExit("pbrt_hipagg: a quadric record of the flat tables has no counterpart among the reference's primitives");
        // object instances: ours (definition id, render-from-instance matrix) <-> the reference's TransformedPrimitive (the definition
        // is identified through one of its triangles; identical placements of one definition are interchangeable)
        if (desc->n_instances > 0) {
            std::vector<int> triDef(nTriangles + desc->n_quadrics, -1);   // primitive id (triangle or quadric) -> its definition
            for (int k = 0; k < desc->n_instance_defs; ++k)
                for (int j = 0; j < desc->instance_defs[k].n_prims; ++j) {
                    const int t = desc->bvh_prims[desc->instance_defs[k].first_prim + j];
                    if (t >= 0 && t < nTriangles + desc->n_quadrics) triDef[t] = k;
                }
            std::map<const void *, int> quadOfPrim;   // reference primitive -> quadric id (filled by visit above)
            for (int q = 0; q < desc->n_quadrics; ++q) quadOfPrim[quadPrims[q].ptr()] = q;
            std::function<int(Primitive)> anyTriangle = [&](Primitive p) -> int {
                if (p.Is<BVHAggregate>()) { for (Primitive q : p.Cast<BVHAggregate>()->primitives) { int r = anyTriangle(q); if (r >= 0) return r; } return -1; }
                Shape shape = p.Is<SimplePrimitive>() ? p.Cast<SimplePrimitive>()->shape : p.Is<GeometricPrimitive>() ? p.Cast<GeometricPrimitive>()->shape : Shape();
                if (!shape) return -1;
                if (!shape.Is<Triangle>()) { auto it = quadOfPrim.find(p.ptr()); return it == quadOfPrim.end() ? -1 : nTriangles + it->second; }
                return firstTri[shape.Cast<Triangle>()->meshIndex] + shape.Cast<Triangle>()->triIndex;
            };
            instPrims.assign(desc->n_instances, nullptr);
            std::vector<bool> used(refInstances.size(), false);
            for (int k = 0; k < desc->n_instances; ++k) {
                for (size_t r = 0; r < refInstances.size() && !instPrims[k]; ++r) {
                    if (used[r]) continue;
                    const int t = anyTriangle(refInstances[r]->primitive);
                    if (t < 0 || triDef[t] != desc->instances[k].def) continue;
                    const SquareMatrix<4> &m = refInstances[r]->renderFromPrimitive->GetMatrix();
                    bool same = true;
                    for (int a = 0; a < 4; ++a)
                        for (int b = 0; b < 4; ++b) same &= std::memcmp(&m[a][b], &desc->instances[k].render_from_instance.m[a][b], 4) == 0 || m[a][b] == desc->instances[k].render_from_instance.m[a][b];
                    if (same) { instPrims[k] = refInstances[r]; used[r] = true; }
                }
                if (!instPrims[k]) ErrorExit("pbrt_hipagg: object instance %d of the flat tables has no counterpart among the reference's primitives", k);
            }
        }
    }

    Bounds3f Bounds() const override {
        float b[6];
        if (wf_aggregate_bounds(ctx, b) != 0) ErrorExit("wf_aggregate_bounds: %s", wf_last_error());
        return Bounds3f(Point3f(b[0], b[1], b[2]), Point3f(b[3], b[4], b[5]));
    }

    void IntersectClosest(int maxRays, const RayQueue *rayQueue, EscapedRayQueue *escapedRayQueue, HitAreaLightQueue *hitAreaLightQueue,
                          MaterialEvalQueue *basicEvalMaterialQueue, MaterialEvalQueue *universalEvalMaterialQueue,
                          MediumSampleQueue *mediumSampleQueue, RayQueue *nextRayQueue) const override {
        const int n = rayQueue->Size();
        if (n == 0) return;
        std::vector<float> o(3 * (size_t)n), d(3 * (size_t)n), tmax(n, Infinity);
        for (int i = 0; i < n; ++i) {
            const RayWorkItem r = (*rayQueue)[i];
            o[3 * i] = r.ray.o.x; o[3 * i + 1] = r.ray.o.y; o[3 * i + 2] = r.ray.o.z;
            d[3 * i] = r.ray.d.x; d[3 * i + 1] = r.ray.d.y; d[3 * i + 2] = r.ray.d.z;
        }
        std::vector<wf_hit_record> hits(n);
        // count_visits = 0: the production traversal (wf_traverse.h); near-ties are re-traced in the reference's order
        if (wf_trace_closest_host(ctx, n, o.data(), d.data(), tmax.data(), hits.data(), 0) != 0) ErrorExit("wf_trace_closest_host: %s", wf_last_error());
        ParallelFor(0, n, [&](int64_t index) {
            const RayWorkItem r = (*rayQueue)[index];
            const wf_hit_record &h = hits[index];
            if (h.prim < 0) {
                EnqueueWorkAfterMiss(r, mediumSampleQueue, escapedRayQueue);
                return;
            }
            if (h.prim >= nTri) {
                // a sphere / disk / cylinder / patch / curve segment: the reference's own primitive rebuilds its interaction from the
                // (instance-space) ray — Shape::Intersect + the alpha recursion + SetIntersectionProperties of
                // GeometricPrimitive / SimplePrimitive::Intersect (cpu/primitive.cpp:50-110) — and must find the GPU's distance
                const TransformedPrimitive *tpq = h.instance >= 0 ? instPrims[h.instance] : nullptr;
                Ray rq = r.ray;
                if (tpq) { Float tMax = Infinity; rq = tpq->renderFromPrimitive->ApplyInverse(r.ray, &tMax); }
                pstd::optional<ShapeIntersection> si = quadPrims[h.prim - nTri].Intersect(rq, Infinity);
                if (!si || si->tHit != h.t) {
                    if (tMismatch.fetch_add(1) < 8) fprintf(stderr, "pbrt_hipagg: primitive %d: the GPU's hit distance %a is not the reference primitive's %a\n", h.prim, h.t, si ? si->tHit : -1.f);
                    if (!si) { EnqueueWorkAfterMiss(r, mediumSampleQueue, escapedRayQueue); return; }
                }
                SurfaceInteraction intr = si->intr;
                if (tpq) intr = (*tpq->renderFromPrimitive)(intr);
                EnqueueWorkAfterIntersection(r, r.ray.medium, si->tHit, intr, mediumSampleQueue, nextRayQueue, hitAreaLightQueue, basicEvalMaterialQueue,
                                             universalEvalMaterialQueue);
                return;
            }
            // what Triangle::Intersect + GeometricPrimitive / SimplePrimitive::Intersect build from the hit
            Primitive p = prims[h.prim];
            const Triangle *tri = (p.Is<SimplePrimitive>() ? p.Cast<SimplePrimitive>()->shape : p.Cast<GeometricPrimitive>()->shape).Cast<Triangle>();
            TriangleIntersection ti{h.b0, h.b1, h.b2, h.t};
            // inside an object instance the interaction is built in the instance's space from the transformed ray and taken to render
            // space by the instance's transform, as TransformedPrimitive::Intersect does (cpu/primitive.cpp:112-125)
            const TransformedPrimitive *tp = h.instance >= 0 ? instPrims[h.instance] : nullptr;
            Ray ray = r.ray;
            if (tp) { Float tMax = Infinity; ray = tp->renderFromPrimitive->ApplyInverse(r.ray, &tMax); }
            SurfaceInteraction intr = Triangle::InteractionFromIntersection(tri->GetMesh(), tri->triIndex, ti, ray.time, -ray.d);
            if (p.Is<SimplePrimitive>()) intr.SetIntersectionProperties(p.Cast<SimplePrimitive>()->material, nullptr, nullptr, ray.medium);
            else {
                const GeometricPrimitive *g = p.Cast<GeometricPrimitive>();
                intr.SetIntersectionProperties(g->material, g->areaLight, &g->mediumInterface, ray.medium);
            }
            if (tp) intr = (*tp->renderFromPrimitive)(intr);
            EnqueueWorkAfterIntersection(r, r.ray.medium, h.t, intr, mediumSampleQueue, nextRayQueue, hitAreaLightQueue, basicEvalMaterialQueue,
                                         universalEvalMaterialQueue);
        });
    }

    void IntersectShadow(int maxRays, ShadowRayQueue *shadowRayQueue, SOA<PixelSampleState> *pixelSampleState) const override {
        const int n = shadowRayQueue->Size();
        if (n == 0) return;
        std::vector<float> o(3 * (size_t)n), d(3 * (size_t)n), tmax(n);
        for (int i = 0; i < n; ++i) {
            const ShadowRayWorkItem w = (*shadowRayQueue)[i];
            o[3 * i] = w.ray.o.x; o[3 * i + 1] = w.ray.o.y; o[3 * i + 2] = w.ray.o.z;
            d[3 * i] = w.ray.d.x; d[3 * i + 1] = w.ray.d.y; d[3 * i + 2] = w.ray.d.z;
            tmax[i] = w.tMax;
        }
        std::vector<int32_t> occluded(n);
        if (wf_trace_any_host(ctx, n, o.data(), d.data(), tmax.data(), occluded.data(), nullptr, nullptr) != 0) ErrorExit("wf_trace_any_host: %s", wf_last_error());
        ParallelFor(0, n, [&](int64_t index) { RecordShadowRayResult((*shadowRayQueue)[index], pixelSampleState, occluded[index] != 0); });
    }

    // TraceTransmittance (wavefront/intersect.h:165-274) on the GPU: walk and ratio tracking over the flat tables' media
    void IntersectShadowTr(int maxRays, ShadowRayQueue *q, SOA<PixelSampleState> *ps) const override {
        const int n = q->Size();
        if (n == 0) return;
        std::vector<float> o(3 * (size_t)n), d(3 * (size_t)n), tmax(n), lambda(4 * (size_t)n), Ld(4 * (size_t)n), ru(4 * (size_t)n), rl(4 * (size_t)n), out(4 * (size_t)n);
        std::vector<int32_t> medium(n);
        for (int i = 0; i < n; ++i) {
            const ShadowRayWorkItem w = (*q)[i];
            o[3 * i] = w.ray.o.x; o[3 * i + 1] = w.ray.o.y; o[3 * i + 2] = w.ray.o.z;
            d[3 * i] = w.ray.d.x; d[3 * i + 1] = w.ray.d.y; d[3 * i + 2] = w.ray.d.z;
            tmax[i] = w.tMax;
            medium[i] = -1;
            if (w.ray.medium) {
                auto it = mediumIds.find(w.ray.medium.ptr());
                if (it == mediumIds.end()) ErrorExit("pbrt_hipagg: a shadow ray's medium has no counterpart in the flat tables");
                medium[i] = it->second;
            }
            for (int c = 0; c < 4; ++c) { lambda[4 * i + c] = w.lambda[c]; Ld[4 * i + c] = w.Ld[c]; ru[4 * i + c] = w.r_u[c]; rl[4 * i + c] = w.r_l[c]; }
        }
        if (wf_trace_shadow_tr_host(ctx, n, o.data(), d.data(), tmax.data(), medium.data(), lambda.data(), Ld.data(), ru.data(), rl.data(), out.data()) != 0)
            ErrorExit("wf_trace_shadow_tr_host: %s", wf_last_error());
        ParallelFor(0, n, [&](int64_t i) {
            const ShadowRayWorkItem w = (*q)[i];
            SampledSpectrum add;
            for (int c = 0; c < 4; ++c) add[c] = out[4 * i + c];
            if (!add) return;   // blocked / ended by the roulette: nothing is added (intersect.h:267-272)
            SampledSpectrum Lpixel = ps->L[w.pixelIndex];
            ps->L[w.pixelIndex] = Lpixel + add;
        });
    }
    void SetCameraMedium(Medium m, int id) { if (m) mediumIds[m.ptr()] = id; }
    // wavefront/aggregate.cpp:90-115: the probe segment's hits with the item's own material, one kept by weighted reservoir sampling
    void IntersectOneRandom(int maxRays, SubsurfaceScatterQueue *q) const override {
        const int n = q->Size();
        if (n == 0) return;
        std::vector<float> p0(3 * (size_t)n), p1(3 * (size_t)n), pdf(n);
        std::vector<int32_t> material(n);
        for (int i = 0; i < n; ++i) {
            const SubsurfaceScatterWorkItem &w = (*q)[i];
            p0[3 * i] = w.p0.x; p0[3 * i + 1] = w.p0.y; p0[3 * i + 2] = w.p0.z;
            p1[3 * i] = w.p1.x; p1[3 * i + 1] = w.p1.y; p1[3 * i + 2] = w.p1.z;
            auto it = materialIds.find(w.material.ptr());
            if (it == materialIds.end()) ErrorExit("pbrt_hipagg: a subsurface work item's material has no counterpart in the flat tables");
            material[i] = it->second;
        }
        std::vector<wf_hit_record> hits(n);
        if (wf_trace_one_random_host(ctx, n, p0.data(), p1.data(), material.data(), hits.data(), pdf.data()) != 0)
            ErrorExit("wf_trace_one_random_host: %s", wf_last_error());
        ParallelFor(0, n, [&](int64_t index) {
            const wf_hit_record &h = hits[index];
            q->reservoirPDF[index] = h.prim < 0 ? 0.f : pdf[index];
            if (h.prim < 0) return;
            if (h.prim >= nTri) ErrorExit("pbrt_hipagg: subsurface probes that hit quadrics / patches / curves are outside this adapter's scope");
            Primitive p = prims[h.prim];
            const Triangle *tri = (p.Is<SimplePrimitive>() ? p.Cast<SimplePrimitive>()->shape : p.Cast<GeometricPrimitive>()->shape).Cast<Triangle>();
            TriangleIntersection ti{h.b0, h.b1, h.b2, h.t};
            const SubsurfaceScatterWorkItem &w = (*q)[index];
            // (SubsurfaceInteraction keeps pi, n, dpdu, dpdv and the shading frame: none of them depends on wo)
            SurfaceInteraction intr = Triangle::InteractionFromIntersection(tri->GetMesh(), tri->triIndex, ti, 0.f, Normalize(w.p0 - w.p1));
            q->ssi[index] = SubsurfaceInteraction(intr);
        });
    }

    int64_t DistanceMismatches() const { return tMismatch.load(); }

  private:
    // content keys of the non-triangle shapes: the same fields on both sides (FNV-1a over their bits; -0 == +0)
    static void Mix(uint64_t &h, float f) { f += 0.0f; uint32_t u; std::memcpy(&u, &f, 4); h = (h ^ u) * 1099511628211ull; }
    static void MixMatrix(uint64_t &h, const SquareMatrix<4> &m) { for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) Mix(h, m[a][b]); }
    static void MixMatrix(uint64_t &h, const float m[4][4]) { for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) Mix(h, m[a][b]); }
    static uint64_t QuadricKeyOurs(const wf_quadric &q) {
        uint64_t h = 1469598103934665603ull ^ (uint64_t)q.type;
        switch (q.type) {
        case WF_QUADRIC_SPHERE: Mix(h, q.radius); Mix(h, q.z_min); Mix(h, q.z_max); Mix(h, q.phi_max); MixMatrix(h, q.render_from_object.m); break;
        case WF_QUADRIC_DISK: Mix(h, q.z_min); Mix(h, q.radius); Mix(h, q.inner_radius); Mix(h, q.phi_max); MixMatrix(h, q.render_from_object.m); break;
        case WF_QUADRIC_CYLINDER: Mix(h, q.radius); Mix(h, q.z_min); Mix(h, q.z_max); Mix(h, q.phi_max); MixMatrix(h, q.render_from_object.m); break;
        case WF_QUADRIC_BILINEAR: { const float *p = &q.render_from_object.m[0][0]; for (int i = 0; i < 12; ++i) Mix(h, p[i]); break; }
        case WF_QUADRIC_CURVE:
            for (int i = 0; i < 12; ++i) Mix(h, q.ext[i]);
            Mix(h, q.radius); Mix(h, q.theta_z_min); Mix(h, q.z_min); Mix(h, q.z_max); Mix(h, q.inner_radius); MixMatrix(h, q.render_from_object.m);
            break;
        }
        return h;
    }
    static uint64_t QuadricKeyRef(Shape shape) {
        uint64_t h = 1469598103934665603ull;
        if (shape.Is<Sphere>()) {
            const Sphere *s = shape.Cast<Sphere>();
            h ^= (uint64_t)WF_QUADRIC_SPHERE;
            Mix(h, s->radius); Mix(h, s->zMin); Mix(h, s->zMax); Mix(h, s->phiMax); MixMatrix(h, s->renderFromObject->GetMatrix());
        } else if (shape.Is<Disk>()) {
            const Disk *s = shape.Cast<Disk>();
            h ^= (uint64_t)WF_QUADRIC_DISK;
            Mix(h, s->height); Mix(h, s->radius); Mix(h, s->innerRadius); Mix(h, s->phiMax); MixMatrix(h, s->renderFromObject->GetMatrix());
        } else if (shape.Is<Cylinder>()) {
            const Cylinder *s = shape.Cast<Cylinder>();
            h ^= (uint64_t)WF_QUADRIC_CYLINDER;
            Mix(h, s->radius); Mix(h, s->zMin); Mix(h, s->zMax); Mix(h, s->phiMax); MixMatrix(h, s->renderFromObject->GetMatrix());
        } else if (shape.Is<BilinearPatch>()) {
            const BilinearPatch *s = shape.Cast<BilinearPatch>();
            const BilinearPatchMesh *mesh = s->GetMesh();
            const int *v = &mesh->vertexIndices[4 * s->blpIndex];
            h ^= (uint64_t)WF_QUADRIC_BILINEAR;
            for (int k = 0; k < 4; ++k) { Mix(h, mesh->p[v[k]].x); Mix(h, mesh->p[v[k]].y); Mix(h, mesh->p[v[k]].z); }
        } else if (shape.Is<Curve>()) {
            const Curve *s = shape.Cast<Curve>();
            const CurveCommon *c = s->common;
            h ^= (uint64_t)WF_QUADRIC_CURVE;
            for (int k = 0; k < 4; ++k) { Mix(h, c->cpObj[k].x); Mix(h, c->cpObj[k].y); Mix(h, c->cpObj[k].z); }
            Mix(h, c->width[0]); Mix(h, c->width[1]); Mix(h, s->uMin); Mix(h, s->uMax); Mix(h, (float)(int)c->type); MixMatrix(h, c->renderFromObject->GetMatrix());
        } else ErrorExit("pbrt_hipagg: shape type outside this adapter's scope: %s", shape.ToString());
        return h;
    }
    CPUAggregate *cpu;
    wf_ctx *ctx;
    int nTri = 0;
    std::vector<Primitive> quadPrims;              // quadric id of the flat tables -> the reference's primitive
    mutable std::atomic<int64_t> tMismatch{0};
    std::vector<Primitive> prims;
    std::map<const void *, int32_t> materialIds;   // the reference's Material (tagged pointer payload) -> material id of the flat tables
    std::map<const void *, int32_t> mediumIds;     // the reference's Medium -> medium id of the flat tables
    std::vector<const TransformedPrimitive *> instPrims;   // object instance id of the flat tables -> the reference's primitive
};

int main(int argc, char **argv) {
    std::string scenePath, outfile, dataDir;
    int spp = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--spp" && i + 1 < argc) spp = atoi(argv[++i]);
        else if (a == "--outfile" && i + 1 < argc) outfile = argv[++i];
        else if (a == "--datadir" && i + 1 < argc) dataDir = argv[++i];
        else scenePath = a;
    }
    if (scenePath.empty()) { fprintf(stderr, "usage: pbrt_hipagg [--spp N] [--outfile out.pfm] [--datadir DIR] scene.pbrt\n"); return 1; }
    if (dataDir.empty()) {
        std::string self = argv[0];
        size_t p = self.rfind('/');
        dataDir = (p == std::string::npos ? std::string(".") : self.substr(0, p)) + "/../../pbrt-v4_amd/data";
    }
    // the HIP side: the same scene file through libwfhost's loader, uploaded to device 0
    if (wfh_init(dataDir.c_str()) != 0) { fprintf(stderr, "wfh_init failed\n"); return 1; }
    wfh_scene *hs = wfh_scene_load(scenePath.c_str(), spp, 0);
    const bool dryRun = getenv("WF_ADAPTER_DRYRUN") != nullptr;  // mesh matching only (no GPU): debugging aid
    if (!hs || (!dryRun && wfh_renderer_create(hs, 0, 1) != 0)) { fprintf(stderr, "libwfhost: could not load / upload the scene: %s\n", wf_last_error()); return 1; }
    wfh_info info;
    wfh_scene_info(hs, &info);

    PBRTOptions opt;
    opt.wavefront = true;
    opt.quiet = true;
    opt.seed = 0;
    if (spp > 0) opt.pixelSamples = spp;
    if (!outfile.empty()) opt.imageFile = outfile;
    InitPBRT(opt);
    {
        BasicScene scene;
        BasicSceneBuilder builder(&scene);
        ParseFiles(&builder, {scenePath});
        WavefrontPathIntegrator *in = new WavefrontPathIntegrator(pstd::pmr::get_default_resource(), scene);
        CPUAggregate *cpu = dynamic_cast<CPUAggregate *>(in->aggregate);
        CHECK(cpu);
        if (dryRun) {
            const wf_scene_desc *dd = wfh_scene_desc(hs);
            for (int m = 0; m < dd->n_meshes; ++m) { if (dd->meshes[m].nverts <= 0) continue; const float *P = dd->P + 3 * (size_t)dd->meshes[m].first_vertex; fprintf(stderr, "ours mesh %d ntris %d nverts %d p0 %a %a %a\n", m, dd->meshes[m].ntris, dd->meshes[m].nverts, P[0], P[1], P[2]); }
            for (size_t m = 0; m < Triangle::allMeshes->size(); ++m) { const TriangleMesh *tm = (*Triangle::allMeshes)[m]; fprintf(stderr, "ref mesh %zu ntris %d nverts %d p0 %a %a %a\n", m, tm->nTriangles, tm->nVertices, tm->p[0].x, tm->p[0].y, tm->p[0].z); }
        }
        HipAggregate *agg = new HipAggregate(cpu, dryRun ? nullptr : wfh_renderer_ctx(hs), wfh_scene_desc(hs));
        {
            Medium camMedium = nullptr;   // CameraBase::medium (every camera type derives from it)
            if (in->camera.Is<PerspectiveCamera>()) camMedium = in->camera.Cast<PerspectiveCamera>()->medium;
            else if (in->camera.Is<OrthographicCamera>()) camMedium = in->camera.Cast<OrthographicCamera>()->medium;
            else if (in->camera.Is<SphericalCamera>()) camMedium = in->camera.Cast<SphericalCamera>()->medium;
            else if (in->camera.Is<RealisticCamera>()) camMedium = in->camera.Cast<RealisticCamera>()->medium;
            agg->SetCameraMedium(camMedium, wfh_scene_desc(hs)->camera.medium);
        }
        in->aggregate = agg;
        if (dryRun) { printf("dry run: meshes matched\n"); return 0; }
        Float seconds = in->Render();
        ImageMetadata metadata;
        in->camera.InitMetadata(&metadata);
        metadata.renderTimeSeconds = seconds;
        metadata.samplesPerPixel = in->sampler.SamplesPerPixel();
        in->film.WriteImage(metadata);
        printf("{\"adapter\": \"HipAggregate\", \"seconds\": %.3f, \"triangles\": %d, \"distance_mismatches\": %lld}\n", (double)seconds, info.n_triangles,
               (long long)agg->DistanceMismatches());
    }
    CleanupPBRT();
    return 0;
}
