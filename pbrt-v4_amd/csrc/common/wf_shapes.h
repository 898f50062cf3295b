// wf_shapes.h — ray/triangle and ray/box arithmetic, the BVH walk, surface-interaction reconstruction
// and triangle sampling.  Operation-for-operation restatements of
//   shapes.cpp:168-269        IntersectTriangle (watertight test, fp64 edge fallback, delta-t bound)
//   shapes.h:884-1010         Triangle::InteractionFromIntersection
//   shapes.h:1013-1180        Triangle::Sample / PDF (area + spherical-triangle sampling)
//   util/vecmath.h:1574-1608  Bounds3::IntersectP (slab test with the 1+2*gamma(3) widening)
//   cpu/aggregates.cpp:529-624 BVHAggregate::Intersect / IntersectP (near-child-first by dirIsNeg[axis])
// so that hits, barycentrics and error bounds are bit-identical to the reference's CPU path.
#pragma once

#include "wf_scene.h"

namespace wf {

struct TriHit { float b0, b1, b2, t; };

// The per-ray part of IntersectTriangle (shapes.cpp:181-198): the permutation (a rotation chosen by the
// largest |d| component) and the shear constants depend on the ray only, so the traversal kernels compute
// them once per ray instead of once per triangle test.  Same operations, same values.
struct RayShear {
    int kz;
    float Sx, Sy, Sz;
};
WF_HD V3 ShearRot(int kz, V3 v) { return kz == 0 ? V3{v.y, v.z, v.x} : (kz == 1 ? V3{v.z, v.x, v.y} : v); }
WF_HD RayShear MakeRayShear(V3 rd) {
    // Permute(v, {kx, ky, kz}) with kx = (kz+1)%3, ky = (kx+1)%3 is one of three rotations; written as
    // selects so that nothing is a runtime-indexed array on the device
    RayShear sh;
    sh.kz = MaxComponentIndex(Abs(rd));
    V3 d = ShearRot(sh.kz, rd);
    sh.Sx = -d.x / d.z;
    sh.Sy = -d.y / d.z;
    sh.Sz = 1 / d.z;
    return sh;
}
// The per-triangle part.  checkDegenerate = false when the caller knows the triangle is not degenerate
// (the production BVH layout drops the test by flagging degenerate triangles at build time).
WF_HD bool IntersectTriangleSheared(V3 ro, const RayShear &sh, float tMax, V3 p0, V3 p1, V3 p2, TriHit *hit, bool checkDegenerate = true) {
    if (checkDegenerate && LengthSquared(Cross(p2 - p0, p1 - p0)) == 0) return false;
    V3 p0t = p0 - ro, p1t = p1 - ro, p2t = p2 - ro;
    p0t = ShearRot(sh.kz, p0t);
    p1t = ShearRot(sh.kz, p1t);
    p2t = ShearRot(sh.kz, p2t);
    const float Sx = sh.Sx, Sy = sh.Sy, Sz = sh.Sz;
    p0t.x += Sx * p0t.z;
    p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z;
    p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z;
    p2t.y += Sy * p2t.z;
    float e0 = DifferenceOfProducts(p1t.x, p2t.y, p1t.y, p2t.x);
    float e1 = DifferenceOfProducts(p2t.x, p0t.y, p2t.y, p0t.x);
    float e2 = DifferenceOfProducts(p0t.x, p1t.y, p0t.y, p1t.x);
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y;
        double p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y;
        double p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y;
        double p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= Sz;
    p1t.z *= Sz;
    p2t.z *= Sz;
    float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMax * det)) return false;
    float invDet = 1 / det;
    float b0 = e0 * invDet, b1 = e1 * invDet, b2 = e2 * invDet;
    float t = tScaled * invDet;
    float maxZt = MaxComponentValue(Abs(V3{p0t.z, p1t.z, p2t.z}));
    float deltaZ = gamma(3) * maxZt;
    float maxXt = MaxComponentValue(Abs(V3{p0t.x, p1t.x, p2t.x}));
    float maxYt = MaxComponentValue(Abs(V3{p0t.y, p1t.y, p2t.y}));
    float deltaX = gamma(5) * (maxXt + maxZt);
    float deltaY = gamma(5) * (maxYt + maxZt);
    float deltaE = 2 * (gamma(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = MaxComponentValue(Abs(V3{e0, e1, e2}));
    float deltaT = 3 * (gamma(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * abs(invDet);
    if (t <= deltaT) return false;
    hit->b0 = b0; hit->b1 = b1; hit->b2 = b2; hit->t = t;
    return true;
}
// IntersectTriangle, shapes.cpp:168-269
WF_HD bool IntersectTriangle(V3 ro, V3 rd, float tMax, V3 p0, V3 p1, V3 p2, TriHit *hit) {
    // (the degenerate-triangle test comes first in the reference; it does not depend on the ray)
    if (LengthSquared(Cross(p2 - p0, p1 - p0)) == 0) return false;
    RayShear sh = MakeRayShear(rd);
    return IntersectTriangleSheared(ro, sh, tMax, p0, p1, p2, hit, false);
}

// Bounds3::IntersectP with precomputed invDir/dirIsNeg.  dirIsNeg travels as a 3-bit mask (bit a = ray
// direction negative along axis a) so that nothing is a runtime-indexed private array on the device.
WF_HD bool BoxIntersectP(const float bmin[3], const float bmax[3], V3 o, float raytMax, V3 invDir, int negMask) {
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
    return (tMin < raytMax) && (tMax > 0);
}

WF_HD void TriVerts(const SceneView &sv, int tri, V3 *p0, V3 *p1, V3 *p2) {
    const auto v = sv.triIndices + 3 * (size_t)tri;
    *p0 = LoadP(sv, v[0]); *p1 = LoadP(sv, v[1]); *p2 = LoadP(sv, v[2]);
}

// ---------------------------------------------------------------------------------------------
// Sphere (shapes.h:107-383).  The quadric test runs in object space on intervals, as the reference's.
// Transform::operator()(Point3fi) for an exact point and Transform::operator()(Vector3fi) for an exact vector
// (util/transform.h:133-176, 272-306); affine transforms only (w' == 1, checked at load).
// Interval / Float (util/math.h:1027-1035) and the homogeneous divide the reference applies whenever w' != 1: the
// numerically inverted matrix of a composite transform can carry 0.99999994 in its last element
WF_HD Ivl DivF(Ivl i, float f) {
    if (f == 0) return Ivl(-WF_INFINITY, WF_INFINITY);
    if (f > 0) return Ivl(NextFloatDown(i.lo / f), NextFloatUp(i.hi / f));
    return Ivl(NextFloatDown(i.hi / f), NextFloatUp(i.lo / f));
}
WF_HD float XfW(const float m[4][4], V3 p) { return (m[3][0] * p.x + m[3][1] * p.y) + (m[3][2] * p.z + m[3][3]); }
WF_HD Ivl3 XfPointExactI(const float m[4][4], V3 p) {
    float x = p.x, y = p.y, z = p.z;
    float xp = (m[0][0] * x + m[0][1] * y) + (m[0][2] * z + m[0][3]);
    float yp = (m[1][0] * x + m[1][1] * y) + (m[1][2] * z + m[1][3]);
    float zp = (m[2][0] * x + m[2][1] * y) + (m[2][2] * z + m[2][3]);
    float ex = gamma(3) * (abs(m[0][0] * x) + abs(m[0][1] * y) + abs(m[0][2] * z) + abs(m[0][3]));
    float ey = gamma(3) * (abs(m[1][0] * x) + abs(m[1][1] * y) + abs(m[1][2] * z) + abs(m[1][3]));
    float ez = gamma(3) * (abs(m[2][0] * x) + abs(m[2][1] * y) + abs(m[2][2] * z) + abs(m[2][3]));
    Ivl3 r{Ivl::FromValueAndError(xp, ex), Ivl::FromValueAndError(yp, ey), Ivl::FromValueAndError(zp, ez)};
    const float wp = XfW(m, p);
    if (wp != 1) { r.x = DivF(r.x, wp); r.y = DivF(r.y, wp); r.z = DivF(r.z, wp); }
    return r;
}
WF_HD Ivl3 XfVectorExactI(const float m[4][4], V3 v) {
    float x = v.x, y = v.y, z = v.z;
    float ex = gamma(3) * (abs(m[0][0] * x) + abs(m[0][1] * y) + abs(m[0][2] * z));
    float ey = gamma(3) * (abs(m[1][0] * x) + abs(m[1][1] * y) + abs(m[1][2] * z));
    float ez = gamma(3) * (abs(m[2][0] * x) + abs(m[2][1] * y) + abs(m[2][2] * z));
    float xp = m[0][0] * x + m[0][1] * y + m[0][2] * z;
    float yp = m[1][0] * x + m[1][1] * y + m[1][2] * z;
    float zp = m[2][0] * x + m[2][1] * y + m[2][2] * z;
    return Ivl3{Ivl::FromValueAndError(xp, ex), Ivl::FromValueAndError(yp, ey), Ivl::FromValueAndError(zp, ez)};
}
// Transform::operator()(Point3fi) for a point with error bounds (util/transform.h:152-170)
WF_HD P3i XfPointI(const float m[4][4], V3 pIn, V3 eIn) {
    // Point3fi(p, e) first: the transform then sees the interval's midpoint and half width, not p and e themselves
    const P3i in = MakeP3i(pIn, eIn);
    if (in.exact()) {
        Ivl3 r = XfPointExactI(m, pIn);
        return P3i{V3{r.x.lo, r.y.lo, r.z.lo}, V3{r.x.hi, r.y.hi, r.z.hi}};
    }
    const V3 p = in.mid(), e = in.err();
    float x = p.x, y = p.y, z = p.z;
    float xp = (m[0][0] * x + m[0][1] * y) + (m[0][2] * z + m[0][3]);
    float yp = (m[1][0] * x + m[1][1] * y) + (m[1][2] * z + m[1][3]);
    float zp = (m[2][0] * x + m[2][1] * y) + (m[2][2] * z + m[2][3]);
    V3 pe;
    pe.x = (gamma(3) + 1) * (abs(m[0][0]) * e.x + abs(m[0][1]) * e.y + abs(m[0][2]) * e.z) +
           gamma(3) * (abs(m[0][0] * x) + abs(m[0][1] * y) + abs(m[0][2] * z) + abs(m[0][3]));
    pe.y = (gamma(3) + 1) * (abs(m[1][0]) * e.x + abs(m[1][1]) * e.y + abs(m[1][2]) * e.z) +
           gamma(3) * (abs(m[1][0] * x) + abs(m[1][1] * y) + abs(m[1][2] * z) + abs(m[1][3]));
    pe.z = (gamma(3) + 1) * (abs(m[2][0]) * e.x + abs(m[2][1]) * e.y + abs(m[2][2]) * e.z) +
           gamma(3) * (abs(m[2][0] * x) + abs(m[2][1] * y) + abs(m[2][2] * z) + abs(m[2][3]));
    P3i r = MakeP3i(V3{xp, yp, zp}, pe);
    const float wp = XfW(m, p);
    if (wp != 1) {
        Ivl rx = DivF(Ivl(r.lo.x, r.hi.x), wp), ry = DivF(Ivl(r.lo.y, r.hi.y), wp), rz = DivF(Ivl(r.lo.z, r.hi.z), wp);
        r = P3i{V3{rx.lo, ry.lo, rz.lo}, V3{rx.hi, ry.hi, rz.hi}};
    }
    return r;
}
WF_HD V3 XfPoint3(const float m[4][4], V3 p) {  // Transform::operator()(Point3f), affine
    return V3{m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3], m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3],
              m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3]};
}
WF_HD V3 XfVector3(const float m[4][4], V3 v) {
    return V3{m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
              m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
}
WF_HD N3 XfNormal3(const float mInv[4][4], N3 n) {
    return N3{mInv[0][0] * n.x + mInv[1][0] * n.y + mInv[2][0] * n.z, mInv[0][1] * n.x + mInv[1][1] * n.y + mInv[2][1] * n.z,
              mInv[0][2] * n.x + mInv[1][2] * n.y + mInv[2][2] * n.z};
}

struct QuadricHit { float tHit; V3 pObj; float phi; };
// Sphere::BasicIntersect, shapes.h:147-233
WF_HD bool SphereBasicIntersect(const wf_quadric &s, V3 ro, V3 rd, float tMax, QuadricHit *out) {
    const float radius = s.radius;
    Ivl3 oi = XfPointExactI(s.render_from_object.mInv, ro);
    Ivl3 di = XfVectorExactI(s.render_from_object.mInv, rd);
    Ivl a = Sqr(di.x) + Sqr(di.y) + Sqr(di.z);
    Ivl b = 2.f * (di.x * oi.x + di.y * oi.y + di.z * oi.z);
    Ivl c = Sqr(oi.x) + Sqr(oi.y) + Sqr(oi.z) - Sqr(Ivl(radius));
    Ivl k = b / (2.f * a);
    Ivl vx = oi.x - k * di.x, vy = oi.y - k * di.y, vz = oi.z - k * di.z;
    Ivl length = Sqrt(Sqr(vx) + Sqr(vy) + Sqr(vz));
    Ivl discrim = 4.f * a * (Ivl(radius) + length) * (Ivl(radius) - length);
    if (discrim.lo < 0) return false;
    Ivl rootDiscrim = Sqrt(discrim);
    Ivl q;
    if (b.mid() < 0) q = -.5f * (b - rootDiscrim);
    else q = -.5f * (b + rootDiscrim);
    Ivl t0 = q / a;
    Ivl t1 = c / q;
    if (t0.lo > t1.lo) { Ivl t = t0; t0 = t1; t1 = t; }
    if (t0.hi > tMax || t1.lo <= 0) return false;
    Ivl tShapeHit = t0;
    if (tShapeHit.lo <= 0) {
        tShapeHit = t1;
        if (tShapeHit.hi > tMax) return false;
    }
    const V3 om{oi.x.mid(), oi.y.mid(), oi.z.mid()}, dm{di.x.mid(), di.y.mid(), di.z.mid()};
    V3 pHit;
    float phi;
    auto hitPoint = [&]() {
        pHit = om + tShapeHit.mid() * dm;
        pHit = pHit * (radius / Length(pHit));
        if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
        phi = atan2(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * Pi;
    };
    hitPoint();
    auto clipped = [&]() { return (s.z_min > -radius && pHit.z < s.z_min) || (s.z_max < radius && pHit.z > s.z_max) || phi > s.phi_max; };
    if (clipped()) {
        if (tShapeHit == t1) return false;
        if (t1.hi > tMax) return false;
        tShapeHit = t1;
        hitPoint();
        if (clipped()) return false;
    }
    out->tHit = tShapeHit.mid();
    out->pObj = pHit;
    out->phi = phi;
    return true;
}

// Disk::BasicIntersect, shapes.h:427-452
WF_HD bool DiskBasicIntersect(const wf_quadric &s, V3 ro, V3 rd, float tMax, QuadricHit *out) {
    Ivl3 oi = XfPointExactI(s.render_from_object.mInv, ro);
    Ivl3 di = XfVectorExactI(s.render_from_object.mInv, rd);
    if (di.z.mid() == 0) return false;
    float tShapeHit = (s.z_min - oi.z.mid()) / di.z.mid();
    if (tShapeHit <= 0 || tShapeHit >= tMax) return false;
    V3 pHit = V3{oi.x.mid(), oi.y.mid(), oi.z.mid()} + tShapeHit * V3{di.x.mid(), di.y.mid(), di.z.mid()};
    float dist2 = Sqr(pHit.x) + Sqr(pHit.y);
    if (dist2 > Sqr(s.radius) || dist2 < Sqr(s.inner_radius)) return false;
    float phi = atan2(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * Pi;
    if (phi > s.phi_max) return false;
    out->tHit = tShapeHit;
    out->pObj = pHit;
    out->phi = phi;
    return true;
}
// Cylinder::BasicIntersect, shapes.h:575-654
WF_HD bool CylinderBasicIntersect(const wf_quadric &s, V3 ro, V3 rd, float tMax, QuadricHit *out) {
    const float radius = s.radius;
    Ivl3 oi = XfPointExactI(s.render_from_object.mInv, ro);
    Ivl3 di = XfVectorExactI(s.render_from_object.mInv, rd);
    Ivl a = Sqr(di.x) + Sqr(di.y);
    Ivl b = 2.f * (di.x * oi.x + di.y * oi.y);
    Ivl c = Sqr(oi.x) + Sqr(oi.y) - Sqr(Ivl(radius));
    Ivl f = b / (2.f * a);
    Ivl vx = oi.x - f * di.x, vy = oi.y - f * di.y;
    Ivl length = Sqrt(Sqr(vx) + Sqr(vy));
    Ivl discrim = 4.f * a * (Ivl(radius) + length) * (Ivl(radius) - length);
    if (discrim.lo < 0) return false;
    Ivl rootDiscrim = Sqrt(discrim);
    Ivl q;
    if (b.mid() < 0) q = -.5f * (b - rootDiscrim);
    else q = -.5f * (b + rootDiscrim);
    Ivl t0 = q / a;
    Ivl t1 = c / q;
    if (t0.lo > t1.lo) { Ivl t = t0; t0 = t1; t1 = t; }
    if (t0.hi > tMax || t1.lo <= 0) return false;
    Ivl tShapeHit = t0;
    if (tShapeHit.lo <= 0) {
        tShapeHit = t1;
        if (tShapeHit.hi > tMax) return false;
    }
    const V3 om{oi.x.mid(), oi.y.mid(), oi.z.mid()}, dm{di.x.mid(), di.y.mid(), di.z.mid()};
    V3 pHit;
    float phi;
    auto hitPoint = [&]() {
        pHit = om + tShapeHit.mid() * dm;
        float hitRad = sqrt(Sqr(pHit.x) + Sqr(pHit.y));
        pHit.x *= radius / hitRad;
        pHit.y *= radius / hitRad;
        phi = atan2(pHit.y, pHit.x);
        if (phi < 0) phi += 2 * Pi;
    };
    hitPoint();
    auto clipped = [&]() { return pHit.z < s.z_min || pHit.z > s.z_max || phi > s.phi_max; };
    if (clipped()) {
        if (tShapeHit == t1) return false;
        tShapeHit = t1;
        if (t1.hi > tMax) return false;
        hitPoint();
        if (clipped()) return false;
    }
    out->tHit = tShapeHit.mid();
    out->pObj = pHit;
    out->phi = phi;
    return true;
}
// ---------------------------------------------------------------------------------------------
// PiecewiseConstant1D::Sample (util/sampling.h:657-675) and PiecewiseConstant2D::Sample / PDF (:760-780) over [0,1]^2 (wf_pc2d tables)
WF_HD float PC1DSample(const float *func, const float *cdf, int n, float funcInt, float mn, float mx, float u,
                       float *pdf, int *offset) {
    int o = FindInterval(n + 1, [&](int index) { return cdf[index] <= u; });
    *offset = o;
    float du = u - cdf[o];
    if (cdf[o + 1] - cdf[o] > 0) du /= cdf[o + 1] - cdf[o];
    *pdf = (funcInt > 0) ? func[o] / funcInt : 0;
    return Lerp((o + du) / n, mn, mx);
}
WF_HD V2 PC2DSample(const float *D, const wf_pc2d &t, V2 u, float *pdf) {
    float pdf1, pdf0;
    int iv, iu;
    float d1 = PC1DSample(D + t.marg_func_offset, D + t.marg_cdf_offset, t.ny, t.marg_int, 0.f, 1.f, u.y, &pdf1, &iv);
    float d0 = PC1DSample(D + t.cond_func_offset + (size_t)iv * t.nx, D + t.cond_cdf_offset + (size_t)iv * (t.nx + 1), t.nx,
                          D[t.cond_int_offset + iv], 0.f, 1.f, u.x, &pdf0, &iu);
    *pdf = pdf0 * pdf1;
    return V2{d0, d1};
}
WF_HD float PC2DPDF(const float *D, const wf_pc2d &t, V2 p) {
    // domain.Offset(p) with domain [0,1]^2: (p - 0) / (1 - 0)
    V2 o{(p.x - 0.f) / (1.f - 0.f), (p.y - 0.f) / (1.f - 0.f)};
    int iu = Clamp((int)(o.x * t.nx), 0, t.nx - 1);
    int iv = Clamp((int)(o.y * t.ny), 0, t.ny - 1);
    return D[t.cond_func_offset + (size_t)iv * t.nx + iu] / t.marg_int;
}
// BilinearPatch (shapes.h:1279-1510) as a wf_quadric of type WF_QUADRIC_BILINEAR (payload layout: include/wf_abi.h)
struct BlpData { V3 p00, p10, p01, p11; N3 n00, n10, n01, n11; V2 uv00, uv10, uv01, uv11; bool hasN, hasUV; };
WF_HD BlpData LoadBlp(const wf_quadric &s) {
    const float *a = &s.render_from_object.m[0][0], *b = &s.render_from_object.mInv[0][0];
    BlpData d;
    d.p00 = V3{a[0], a[1], a[2]}; d.p10 = V3{a[3], a[4], a[5]}; d.p01 = V3{a[6], a[7], a[8]}; d.p11 = V3{a[9], a[10], a[11]};
    d.uv00 = V2{a[12], a[13]}; d.uv10 = V2{a[14], a[15]};
    d.n00 = N3{b[0], b[1], b[2]}; d.n10 = N3{b[3], b[4], b[5]}; d.n01 = N3{b[6], b[7], b[8]}; d.n11 = N3{b[9], b[10], b[11]};
    d.uv01 = V2{b[12], b[13]}; d.uv11 = V2{b[14], b[15]};
    const int flags = (int)s.pad[0];
    d.hasN = flags & 1; d.hasUV = flags & 2;
    return d;
}
// util/math.h:613-636
WF_HD bool QuadraticF(float a, float b, float c, float *t0, float *t1) {
    if (a == 0) {
        if (b == 0) return false;
        *t0 = *t1 = -c / b;
        return true;
    }
    float discrim = DifferenceOfProducts(b, b, 4 * a, c);
    if (discrim < 0) return false;
    float rootDiscrim = sqrt(discrim);
    float q = -0.5f * (b + copysignf(rootDiscrim, b));
    *t0 = q / a;
    *t1 = c / q;
    if (*t0 > *t1) { float t = *t0; *t0 = *t1; *t1 = t; }
    return true;
}
// Determinant(SquareMatrix<3>) (util/math.h:1419-1425), rows given
WF_HD float Det3(float m00, float m01, float m02, float m10, float m11, float m12, float m20, float m21, float m22) {
    float minor12 = DifferenceOfProducts(m11, m22, m12, m21);
    float minor02 = DifferenceOfProducts(m10, m22, m12, m20);
    float minor01 = DifferenceOfProducts(m10, m21, m11, m20);
    return fmaf(m02, minor01, DifferenceOfProducts(m00, minor12, m01, minor02));
}
WF_HD V3 LerpV(float t, V3 a, V3 b) { return (1 - t) * a + t * b; }
WF_HD float MaxAbsComp(V3 v) { return fmax(fmax(abs(v.x), abs(v.y)), abs(v.z)); }
// IntersectBilinearPatch (shapes.h:1279-1347): (u, v, t) of the hit
WF_HD bool IntersectBilinearPatch(V3 ro, V3 rd, float tMax, V3 p00, V3 p10, V3 p01, V3 p11, float *uOut, float *vOut, float *tOut) {
    float a = Dot(Cross(p10 - p00, p01 - p11), rd);
    float c = Dot(Cross(p00 - ro, rd), p01 - p00);
    float b = Dot(Cross(p10 - ro, rd), p11 - p10) - (a + c);
    float u1, u2;
    if (!QuadraticF(a, b, c, &u1, &u2)) return false;
    float eps = gamma(30) * (MaxAbsComp(ro) + MaxAbsComp(rd) + MaxAbsComp(p00) + MaxAbsComp(p10) + MaxAbsComp(p01) + MaxAbsComp(p11));
    float t = tMax, u = 0, v = 0;
    if (0 <= u1 && u1 <= 1) {
        V3 uo = LerpV(u1, p00, p10);
        V3 ud = LerpV(u1, p01, p11) - uo;
        V3 deltao = uo - ro;
        V3 perp = Cross(rd, ud);
        float p2 = LengthSquared(perp);
        float v1 = Det3(deltao.x, rd.x, perp.x, deltao.y, rd.y, perp.y, deltao.z, rd.z, perp.z);
        float t1 = Det3(deltao.x, ud.x, perp.x, deltao.y, ud.y, perp.y, deltao.z, ud.z, perp.z);
        if (t1 > p2 * eps && 0 <= v1 && v1 <= p2) {
            u = u1;
            v = v1 / p2;
            t = t1 / p2;
        }
    }
    if (0 <= u2 && u2 <= 1 && u2 != u1) {
        V3 uo = LerpV(u2, p00, p10);
        V3 ud = LerpV(u2, p01, p11) - uo;
        V3 deltao = uo - ro;
        V3 perp = Cross(rd, ud);
        float p2 = LengthSquared(perp);
        float v2 = Det3(deltao.x, rd.x, perp.x, deltao.y, rd.y, perp.y, deltao.z, rd.z, perp.z);
        float t2 = Det3(deltao.x, ud.x, perp.x, deltao.y, ud.y, perp.y, deltao.z, ud.z, perp.z);
        t2 /= p2;
        if (0 <= v2 && v2 <= p2 && t > t2 && t2 > eps) {
            t = t2;
            u = u2;
            v = v2 / p2;
        }
    }
    if (t >= tMax) return false;
    *uOut = u; *vOut = v; *tOut = t;
    return true;
}
WF_HD bool BilinearBasicIntersect(const wf_quadric &s, V3 ro, V3 rd, float tMax, QuadricHit *out) {
    const float *a = &s.render_from_object.m[0][0];
    float u, v, t;
    if (!IntersectBilinearPatch(ro, rd, tMax, V3{a[0], a[1], a[2]}, V3{a[3], a[4], a[5]}, V3{a[6], a[7], a[8]}, V3{a[9], a[10], a[11]}, &u, &v, &t)) return false;
    out->tHit = t;
    out->pObj = V3{u, v, 0.f};   // the hit record's three floats: (u, v) of the patch
    out->phi = 0;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Curve (shapes.h:1200-1270, shapes.cpp:519-733): one u-range [uMin, uMax] of a cubic Bezier curve with a width, tested by
// recursive subdivision in a ray-aligned frame.  The record (wf_quadric, type WF_QUADRIC_CURVE) carries the CurveCommon:
// radius = width[0], theta_z_min = width[1], z_min = uMin, z_max = uMax, theta_z_max = normalAngle, phi_max = invSinNormalAngle,
// inner_radius = type (0 flat, 1 cylinder, 2 ribbon), ext[0..11] = cpObj, ext[12..17] = n[0], n[1]; render_from_object as usual.
struct CFloatD { float v, err; };
WF_HD CFloatD TwoProdD(float a, float b) { float ab = a * b; return {ab, fma(a, b, -ab)}; }
WF_HD CFloatD TwoSumD(float a, float b) { float s = a + b, delta = s - a; return {s, (a - (s - delta)) + (b - delta)}; }
WF_HD CFloatD IPD(float a, float b) { return TwoProdD(a, b); }
template <typename... T>
WF_HD CFloatD IPD(float a, float b, T... terms) {
    CFloatD ab = TwoProdD(a, b);
    CFloatD tp = IPD(terms...);
    CFloatD sum = TwoSumD(ab.v, tp.v);
    return {sum.v, ab.err + (tp.err + sum.err)};
}
template <typename... T>
WF_HD float InnerProductD(T... terms) { CFloatD ip = IPD(terms...); return ip.v + ip.err; }   // util/math.h:594-610
struct M44 { float m[4][4]; };
// Inverse(SquareMatrix<4>) (util/math.h:1560-1632)
WF_HD bool Inverse44(const M44 &a, M44 *out) {
    const auto &m = a.m;
    float s0 = DifferenceOfProducts(m[0][0], m[1][1], m[1][0], m[0][1]);
    float s1 = DifferenceOfProducts(m[0][0], m[1][2], m[1][0], m[0][2]);
    float s2 = DifferenceOfProducts(m[0][0], m[1][3], m[1][0], m[0][3]);
    float s3 = DifferenceOfProducts(m[0][1], m[1][2], m[1][1], m[0][2]);
    float s4 = DifferenceOfProducts(m[0][1], m[1][3], m[1][1], m[0][3]);
    float s5 = DifferenceOfProducts(m[0][2], m[1][3], m[1][2], m[0][3]);
    float c0 = DifferenceOfProducts(m[2][0], m[3][1], m[3][0], m[2][1]);
    float c1 = DifferenceOfProducts(m[2][0], m[3][2], m[3][0], m[2][2]);
    float c2 = DifferenceOfProducts(m[2][0], m[3][3], m[3][0], m[2][3]);
    float c3 = DifferenceOfProducts(m[2][1], m[3][2], m[3][1], m[2][2]);
    float c4 = DifferenceOfProducts(m[2][1], m[3][3], m[3][1], m[2][3]);
    float c5 = DifferenceOfProducts(m[2][2], m[3][3], m[3][2], m[2][3]);
    float determinant = InnerProductD(s0, c5, -s1, c4, s2, c3, s3, c2, s5, c0, -s4, c1);
    if (determinant == 0) return false;
    float s = 1 / determinant;
    auto &r = out->m;
    r[0][0] = s * InnerProductD(m[1][1], c5, m[1][3], c3, -m[1][2], c4);
    r[0][1] = s * InnerProductD(-m[0][1], c5, m[0][2], c4, -m[0][3], c3);
    r[0][2] = s * InnerProductD(m[3][1], s5, m[3][3], s3, -m[3][2], s4);
    r[0][3] = s * InnerProductD(-m[2][1], s5, m[2][2], s4, -m[2][3], s3);
    r[1][0] = s * InnerProductD(-m[1][0], c5, m[1][2], c2, -m[1][3], c1);
    r[1][1] = s * InnerProductD(m[0][0], c5, m[0][3], c1, -m[0][2], c2);
    r[1][2] = s * InnerProductD(-m[3][0], s5, m[3][2], s2, -m[3][3], s1);
    r[1][3] = s * InnerProductD(m[2][0], s5, m[2][3], s1, -m[2][2], s2);
    r[2][0] = s * InnerProductD(m[1][0], c4, m[1][3], c0, -m[1][1], c2);
    r[2][1] = s * InnerProductD(-m[0][0], c4, m[0][1], c2, -m[0][3], c0);
    r[2][2] = s * InnerProductD(m[3][0], s4, m[3][3], s0, -m[3][1], s2);
    r[2][3] = s * InnerProductD(-m[2][0], s4, m[2][1], s2, -m[2][3], s0);
    r[3][0] = s * InnerProductD(-m[1][0], c3, m[1][1], c1, -m[1][2], c0);
    r[3][1] = s * InnerProductD(m[0][0], c3, m[0][2], c0, -m[0][1], c1);
    r[3][2] = s * InnerProductD(-m[3][0], s3, m[3][1], s1, -m[3][2], s0);
    r[3][3] = s * InnerProductD(m[2][0], s3, m[2][2], s0, -m[2][1], s1);
    return true;
}
#include "wf_animated.h"   // AnimatedTransform::Interpolate (needs M44 / Inverse44 above)
// LookAt (util/transform.cpp:81-113): returns cameraFromWorld (the numeric inverse) and worldFromCamera
WF_HD bool LookAtD(V3 pos, V3 look, V3 up, M44 *cameraFromWorld, M44 *worldFromCamera) {
    M44 w{};
    w.m[0][3] = pos.x; w.m[1][3] = pos.y; w.m[2][3] = pos.z; w.m[3][3] = 1;
    V3 dir = Normalize(look - pos);
    V3 right = Normalize(Cross(Normalize(up), dir));
    V3 newUp = Cross(dir, right);
    w.m[0][0] = right.x; w.m[1][0] = right.y; w.m[2][0] = right.z; w.m[3][0] = 0;
    w.m[0][1] = newUp.x; w.m[1][1] = newUp.y; w.m[2][1] = newUp.z; w.m[3][1] = 0;
    w.m[0][2] = dir.x; w.m[1][2] = dir.y; w.m[2][2] = dir.z; w.m[3][2] = 0;
    *worldFromCamera = w;
    return Inverse44(w, cameraFromWorld);
}
// Transform::operator()(Point3f) with the homogeneous divide (util/transform.h:303-313)
WF_HD V3 XfPointW(const float m[4][4], V3 p) {
    float xp = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
    float yp = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
    float zp = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
    float wp = m[3][0] * p.x + m[3][1] * p.y + m[3][2] * p.z + m[3][3];
    if (wp == 1) return V3{xp, yp, zp};
    return V3{xp, yp, zp} / wp;
}
// Transform::operator()(const Ray &, Float *tMax = nullptr) (util/transform.h:372-385) with the matrix m: the origin's
// rounding-error bound is walked off along d; Curve::IntersectRay passes no tMax
WF_HD void XfRayOffset(const float m[4][4], V3 o, V3 d, V3 *oOut, V3 *dOut) {
    P3i oi = XfPointI(m, o, V3{0, 0, 0});
    V3 dd = XfVector3(m, d);
    Ivl ox(oi.lo.x, oi.hi.x), oy(oi.lo.y, oi.hi.y), oz(oi.lo.z, oi.hi.z);
    const float lengthSquared = LengthSquared(dd);
    if (lengthSquared > 0) {
        const V3 oError{(ox.hi - ox.lo) / 2, (oy.hi - oy.lo) / 2, (oz.hi - oz.lo) / 2};
        const float dt = Dot(Abs(dd), oError) / lengthSquared;
        const V3 off = dd * dt;
        ox = ox + Ivl(off.x); oy = oy + Ivl(off.y); oz = oz + Ivl(off.z);
    }
    *oOut = V3{ox.mid(), oy.mid(), oz.mid()};
    *dOut = dd;
}
// util/splines.h:17-62
WF_HD V3 BlossomCubicBezier(const V3 *p, float u0, float u1, float u2) {
    V3 a[3] = {LerpV(u0, p[0], p[1]), LerpV(u0, p[1], p[2]), LerpV(u0, p[2], p[3])};
    V3 b[2] = {LerpV(u1, a[0], a[1]), LerpV(u1, a[1], a[2])};
    return LerpV(u2, b[0], b[1]);
}
WF_HD V3 EvaluateCubicBezierD(const V3 *cp, float u, V3 *deriv) {
    V3 cp1[3] = {LerpV(u, cp[0], cp[1]), LerpV(u, cp[1], cp[2]), LerpV(u, cp[2], cp[3])};
    V3 cp2[2] = {LerpV(u, cp1[0], cp1[1]), LerpV(u, cp1[1], cp1[2])};
    if (deriv) {
        if (LengthSquared(cp2[1] - cp2[0]) > 0) *deriv = 3 * (cp2[1] - cp2[0]);
        else *deriv = cp[3] - cp[0];
    }
    return LerpV(u, cp2[0], cp2[1]);
}
WF_HD void CubicBezierControlPoints(const V3 *cp, float uMin, float uMax, V3 *out) {
    out[0] = BlossomCubicBezier(cp, uMin, uMin, uMin);
    out[1] = BlossomCubicBezier(cp, uMin, uMin, uMax);
    out[2] = BlossomCubicBezier(cp, uMin, uMax, uMax);
    out[3] = BlossomCubicBezier(cp, uMax, uMax, uMax);
}
// util/math.h:372-384.  The reference recurses once for v < 1 (-Log2Int(1 / v)); written out here: a recursive device function makes
// every kernel that can REACH it a kernel with a dynamically sized stack (.uses_dynamic_stack, tools/stack_audit.py) — all the general
// shade / traversal variants, through the curve code.  (A negative v recurses for ever in the reference; here it is an answer.)
WF_HD int Log2IntF(float v) {
    const bool inverted = v < 1;
    if (inverted) v = 1 / v;
    const uint32_t midsignif = 0x3504f3u;
    const uint32_t bits = FloatToBits(v);
    const int r = ((int)(bits >> 23) - 127) + (((bits & ((1u << 23) - 1)) >= midsignif) ? 1 : 0);
    return inverted ? -r : r;
}
struct CurveData { V3 cp[4]; float width0, width1, uMin, uMax, normalAngle, invSinNormalAngle; int type; N3 n0, n1; };
WF_HD CurveData LoadCurve(const wf_quadric &s) {
    CurveData c;
    for (int i = 0; i < 4; ++i) c.cp[i] = V3{s.ext[3 * i], s.ext[3 * i + 1], s.ext[3 * i + 2]};
    c.width0 = s.radius; c.width1 = s.theta_z_min; c.uMin = s.z_min; c.uMax = s.z_max;
    c.normalAngle = s.theta_z_max; c.invSinNormalAngle = s.phi_max; c.type = (int)s.inner_radius;
    c.n0 = N3{s.ext[12], s.ext[13], s.ext[14]}; c.n1 = N3{s.ext[15], s.ext[16], s.ext[17]};
    return c;
}
// the ray-aligned frame of Curve::IntersectRay (shapes.cpp:560-571): object-space segment control points, LookAt(o, o + d, dx)
WF_HD void CurveRayFrame(const CurveData &c, V3 o, V3 d, V3 *cpObj, M44 *rayFromObject, M44 *objectFromRay) {
    CubicBezierControlPoints(c.cp, c.uMin, c.uMax, cpObj);
    V3 dx = Cross(d, cpObj[3] - cpObj[0]);
    if (LengthSquared(dx) == 0) { V3 dy; CoordinateSystem(d, &dx, &dy); }
    LookAtD(o, o + d, dx, rayFromObject, objectFromRay);
}
// the ribbon normal at u (shapes.cpp:665-675)
WF_HD N3 CurveRibbonNormal(const CurveData &c, float u) {
    if (c.normalAngle == 0) return c.n0;
    float sin0 = sin((1 - u) * c.normalAngle) * c.invSinNormalAngle;
    float sin1 = sin(u * c.normalAngle) * c.invSinNormalAngle;
    return sin0 * c.n0 + sin1 * c.n1;
}
WF_HD bool CurveBoundsOverlapRay(const V3 *cps, float maxWidth, float zMax) {
    // Union(Bounds3f(cps[0], cps[1]), Bounds3f(cps[2], cps[3])) expanded by maxWidth / 2 against [0, 0, 0] - [0, 0, zMax]
    const float e = 0.5f * maxWidth;
    float lox = fmin(fmin(cps[0].x, cps[1].x), fmin(cps[2].x, cps[3].x)) - e, hix = fmax(fmax(cps[0].x, cps[1].x), fmax(cps[2].x, cps[3].x)) + e;
    float loy = fmin(fmin(cps[0].y, cps[1].y), fmin(cps[2].y, cps[3].y)) - e, hiy = fmax(fmax(cps[0].y, cps[1].y), fmax(cps[2].y, cps[3].y)) + e;
    float loz = fmin(fmin(cps[0].z, cps[1].z), fmin(cps[2].z, cps[3].z)) - e, hiz = fmax(fmax(cps[0].z, cps[1].z), fmax(cps[2].z, cps[3].z)) + e;
    // Overlaps(rayBounds, curveBounds) (util/vecmath.h): pMax >= pMin && pMin <= pMax per axis; rayBounds = Bounds3f((0,0,0), (0,0,zMax))
    const float rzlo = fmin(0.f, zMax), rzhi = fmax(0.f, zMax);
    bool x = (0.f >= lox) && (0.f <= hix);
    bool y = (0.f >= loy) && (0.f <= hiy);
    bool z = (rzhi >= loz) && (rzlo <= hiz);
    return x && y && z;
}
// Curve::IntersectRay + RecursiveIntersect (shapes.cpp:552-733), the recursion unrolled onto an explicit stack (both halves of a
// split are bounds-tested before the first is descended, as the loop over seg does).  wantHit = 0 is IntersectP: the first leaf hit
// returns.  Out of line, pointer arguments only: the traversal kernels keep their register budget.
WF_NI bool CurveIntersectP(const wf_quadric *sp, float rox, float roy, float roz, float rdx, float rdy, float rdz, float tMax, int wantHit,
                           float *uOut, float *vOut, float *tOut) {
    const CurveData c = LoadCurve(*sp);
    V3 o, d;
    XfRayOffset(sp->render_from_object.mInv, V3{rox, roy, roz}, V3{rdx, rdy, rdz}, &o, &d);
    V3 cpObj[4];
    M44 rayFromObject, objectFromRay;
    CurveRayFrame(c, o, d, cpObj, &rayFromObject, &objectFromRay);
    struct Seg { V3 cp[4]; float u0, u1; int depth; };
    Seg stack[12];
    int sp_ = 0;
    Seg root;
    for (int i = 0; i < 4; ++i) root.cp[i] = XfPointW(rayFromObject.m, cpObj[i]);
    const float rayLength = Length(d);
    const float zMax = rayLength * tMax;
    {
        float maxWidth = fmax(Lerp(c.uMin, c.width0, c.width1), Lerp(c.uMax, c.width0, c.width1));
        if (!CurveBoundsOverlapRay(root.cp, maxWidth, zMax)) return false;
    }
    float L0 = 0;
    for (int i = 0; i < 2; ++i)
        L0 = fmax(L0, fmax(fmax(abs(root.cp[i].x - 2 * root.cp[i + 1].x + root.cp[i + 2].x), abs(root.cp[i].y - 2 * root.cp[i + 1].y + root.cp[i + 2].y)),
                           abs(root.cp[i].z - 2 * root.cp[i + 1].z + root.cp[i + 2].z)));
    int maxDepth = 0;
    if (L0 > 0) {
        float eps = fmax(c.width0, c.width1) * .05f;
        int r0 = Log2IntF(1.41421356237f * 6.f * L0 / (8.f * eps)) / 2;
        maxDepth = Clamp(r0, 0, 10);
    }
    root.u0 = c.uMin; root.u1 = c.uMax; root.depth = maxDepth;
    stack[sp_++] = root;
    bool have = false;
    float bestT = 0, bestU = 0, bestV = 0;
    while (sp_ > 0) {
        const Seg g = stack[--sp_];
        if (g.depth > 0) {
            // SubdivideCubicBezier (util/splines.h:45-54)
            V3 sp7[7] = {g.cp[0], (g.cp[0] + g.cp[1]) / 2, (g.cp[0] + 2 * g.cp[1] + g.cp[2]) / 4, (g.cp[0] + 3 * g.cp[1] + 3 * g.cp[2] + g.cp[3]) / 8,
                         (g.cp[1] + 2 * g.cp[2] + g.cp[3]) / 4, (g.cp[2] + g.cp[3]) / 2, g.cp[3]};
            float u[3] = {g.u0, (g.u0 + g.u1) / 2, g.u1};
            // second half first onto the stack: the first half is descended first
            for (int seg = 1; seg >= 0; --seg) {
                float maxWidth = fmax(Lerp(u[seg], c.width0, c.width1), Lerp(u[seg + 1], c.width0, c.width1));
                if (!CurveBoundsOverlapRay(&sp7[3 * seg], maxWidth, zMax)) continue;
                Seg n;
                for (int i = 0; i < 4; ++i) n.cp[i] = sp7[3 * seg + i];
                n.u0 = u[seg]; n.u1 = u[seg + 1]; n.depth = g.depth - 1;
                stack[sp_++] = n;
            }
            continue;
        }
        const V3 *cp = g.cp;
        float edge = (cp[1].y - cp[0].y) * -cp[0].y + cp[0].x * (cp[0].x - cp[1].x);
        if (edge < 0) continue;
        edge = (cp[2].y - cp[3].y) * -cp[3].y + cp[3].x * (cp[3].x - cp[2].x);
        if (edge < 0) continue;
        V2 segmentDir{cp[3].x - cp[0].x, cp[3].y - cp[0].y};
        float denom = Sqr(segmentDir.x) + Sqr(segmentDir.y);
        if (denom == 0) continue;
        float w = SumOfProducts(-cp[0].x, segmentDir.x, -cp[0].y, segmentDir.y) / denom;   // Dot(Vector2f, Vector2f)
        float u = Clamp(Lerp(w, g.u0, g.u1), g.u0, g.u1);
        float hitWidth = Lerp(u, c.width0, c.width1);
        if (c.type == 2) {
            N3 nHit = CurveRibbonNormal(c, u);
            hitWidth *= AbsDot(nHit, d) / rayLength;
        }
        V3 dpcdw;
        V3 pc = EvaluateCubicBezierD(cp, Clamp(w, 0.f, 1.f), &dpcdw);
        float ptCurveDist2 = Sqr(pc.x) + Sqr(pc.y);
        if (ptCurveDist2 > Sqr(hitWidth) * 0.25f) continue;
        if (pc.z < 0 || pc.z > zMax) continue;
        if (!wantHit) return true;
        float tHit = pc.z / rayLength;
        if (have && tHit > bestT) continue;
        float ptCurveDist = sqrt(ptCurveDist2);
        float edgeFunc = dpcdw.x * -pc.y + pc.x * dpcdw.y;
        float v = (edgeFunc > 0) ? 0.5f + ptCurveDist / hitWidth : 0.5f - ptCurveDist / hitWidth;
        have = true; bestT = tHit; bestU = u; bestV = v;
    }
    if (have) { *uOut = bestU; *vOut = bestV; *tOut = bestT; }
    return have;
}
WF_HD bool CurveBasicIntersect(const wf_quadric &s, V3 ro, V3 rd, float tMax, QuadricHit *out) {
    float u, v, t;
    if (!CurveIntersectP(&s, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, tMax, 1, &u, &v, &t)) return false;
    out->tHit = t;
    out->pObj = V3{u, v, t};   // the hit record's three floats: (u, v) on the curve and the hit's parametric distance
    out->phi = 0;
    return true;
}
// CURVES = false: a caller that knows the scene has no curves leaves the curve code (and its register budget: a kernel is
// allocated the maximum over its out-of-line callees) out
template <bool CURVES = true>
WF_HD bool QuadricBasicIntersect(const wf_quadric &s, V3 ro, V3 rd, float tMax, QuadricHit *out) {
    if constexpr (CURVES) if (s.type == WF_QUADRIC_CURVE) return CurveBasicIntersect(s, ro, rd, tMax, out);
    if (s.type == WF_QUADRIC_BILINEAR) return BilinearBasicIntersect(s, ro, rd, tMax, out);
    if (s.type == WF_QUADRIC_DISK) return DiskBasicIntersect(s, ro, rd, tMax, out);
    if (s.type == WF_QUADRIC_CYLINDER) return CylinderBasicIntersect(s, ro, rd, tMax, out);
    return SphereBasicIntersect(s, ro, rd, tMax, out);
}
// GeometricPrimitive::Intersect for a quadric / patch with an alpha texture (cpu/primitive.cpp:50-78), defined below
WF_NI bool QuadricAlphaIntersectP(const SceneView *svp, int prim, float ox, float oy, float oz, float dx, float dy, float dz, float tMax,
                                  float *tHit, float *px, float *py, float *pz);
// the primitive `prim` (>= nTriangles) against the ray: the shape test, then the alpha test of its GeometricPrimitive when it has one
// (RARE = false: a traversal kernel for scenes without curves and without alpha on quadrics — the two out-of-line callees stay out of its budget)
// ... and for a curve (round 5): the same recursion; the accepted hit's (u, v), the SUM of the parametric distances, and — for the
// interaction, which a curve builds in a frame aligned with the ray it was intersected with — the origin of the respawned ray that
// found it and the distance along that ray
WF_NI bool CurveAlphaIntersectP(const SceneView *svp, int prim, float ox, float oy, float oz, float dx, float dy, float dz, float tMax,
                                float *tTotal, float *u, float *v, float *lastOrigin, float *tLocal);
template <bool RARE = true>
WF_HD bool QuadricIntersect(const SceneView &sv, int prim, V3 o, V3 d, float tMax, QuadricHit *qh) {
    constexpr bool CURVES = RARE;
    const wf_quadric &s = sv.quadrics[prim - sv.nTriangles];
    if (RARE && sv.haveQuadricAlpha && sv.meshes[s.mesh].alpha_tex >= 0) {
        float t, x, y, z;
        // (a curve: (x, y, z) = (u, v, distance), the record of a curve hit — CurveHitInteractionP replays the recursion; the dispatch is
        //  inside the callee, so that this site stays ONE out-of-line call)
        if (!QuadricAlphaIntersectP(sv.self, prim, o.x, o.y, o.z, d.x, d.y, d.z, tMax, &t, &x, &y, &z)) return false;
        qh->tHit = t; qh->pObj = V3{x, y, z}; qh->phi = 0;
        return true;
    }
    return QuadricBasicIntersect<CURVES>(s, o, d, tMax, qh);
}
WF_HD float QuadricArea(const wf_quadric &s) {
    if (s.type == WF_QUADRIC_DISK) return s.phi_max * 0.5f * (Sqr(s.radius) - Sqr(s.inner_radius));  // shapes.h:407
    if (s.type == WF_QUADRIC_CYLINDER) return (s.z_max - s.z_min) * s.radius * s.phi_max;             // shapes.h:559
    return s.phi_max * s.radius * (s.z_max - s.z_min);                                                // shapes.h:292
}

// GeometricPrimitive::Intersect's stochastic alpha test (cpu/primitive.cpp:57-72; IntersectP goes through Intersect,
// :79-84).  A triangle cannot be hit again by the ray respawned behind it, so a failed test simply drops the hit.
// The texture sees TextureEvalContext(SurfaceInteraction) with all differentials zero (interaction.h: set only by
// ComputeDifferentials, which this path never calls).
// The interaction's geometric normal as Triangle::InteractionFromIntersection leaves it (shapes.h:936-938: Normalize(Cross(dp02, dp12)),
// flipped by reverseOrientation ^ transformSwapsHandedness; shapes.h:1005 through SetShadingGeometry: face-forwarded to the interpolated
// normal when the mesh has normals) — what TextureEvalContext(SurfaceInteraction) hands an alpha texture as ctx.n (textures.h:36-46).
// Only a directionmix node reads it, so only texture GRAPHS pay for it.
WF_HD N3 TriangleAlphaCtxNormal(const SceneView &sv, const wf_mesh &mesh, const int32_t *v, float b0, float b1, float b2) {
    const V3 p0 = LoadP(sv, v[0]), p1 = LoadP(sv, v[1]), p2 = LoadP(sv, v[2]);
    N3 n = toN(Normalize(Cross(p0 - p2, p1 - p2)));
    if (mesh.flags & WF_MESH_FLIP_NORMAL) n = -n;
    if (mesh.flags & WF_MESH_HAS_N) {
        N3 ns = b0 * LoadN(sv, v[0]) + b1 * LoadN(sv, v[1]) + b2 * LoadN(sv, v[2]);
        ns = LengthSquared(ns) > 0 ? Normalize(ns) : n;
        n = FaceForward(n, ns);
    }
    return n;
}
WF_HD bool AlphaTestPasses(const SceneView &sv, int tri, float b0, float b1, float b2, V3 o, V3 d) {
    const wf_mesh &mesh = sv.meshes[sv.triMesh[tri]];
    if (mesh.alpha_tex < 0) return true;
    const auto v = sv.triIndices + 3 * (size_t)tri;
    V2 uv0{0, 0}, uv1{1, 0}, uv2{1, 1};
    if (mesh.flags & WF_MESH_HAS_UV) { uv0 = LoadUV(sv, v[0]); uv1 = LoadUV(sv, v[1]); uv2 = LoadUV(sv, v[2]); }
    TexCtx tc;
    tc.uv = V2{b0 * uv0.x + b1 * uv1.x + b2 * uv2.x, b0 * uv0.y + b1 * uv1.y + b2 * uv2.y};
    tc.p = b0 * LoadP(sv, v[0]) + b1 * LoadP(sv, v[1]) + b2 * LoadP(sv, v[2]);  // intr.p() for the non-uv mappings (shapes.h:900)
    if (!IsSimpleFloatTexture(sv.textures[mesh.alpha_tex].type)) tc.n = TriangleAlphaCtxNormal(sv, mesh, v, b0, b1, b2);
    float a = EvalFloatTexture(sv, mesh.alpha_tex, tc);
    if (!(a < 1)) return true;
    float u = (a <= 0) ? 1.f : HashToFloat(Hash6f(o, d));
    return !(u > a);
}

struct ClosestHit { int prim; TriHit h; int nodesVisited, trisTested; int inst; };

// TransformedPrimitive::Intersect / IntersectP (cpu/primitive.cpp:112-130): the ray in the instance's space,
// Transform::ApplyInverse(Ray, &tMax) (util/transform.h:416-429) over ApplyInverse(Point3fi) for an exact point
// (util/transform.cpp:263-306: no translation term in the exact case) and ApplyInverse(Vector3f).
WF_HD void InstanceRay(const wf_instance &in, V3 o, V3 d, float *tMax, V3 *oOut, V3 *dOut) {
    const float(*mi)[4] = in.render_from_instance.mInv;
    const float x = o.x, y = o.y, z = o.z;
    const float xp = (mi[0][0] * x + mi[0][1] * y) + (mi[0][2] * z + mi[0][3]);
    const float yp = (mi[1][0] * x + mi[1][1] * y) + (mi[1][2] * z + mi[1][3]);
    const float zp = (mi[2][0] * x + mi[2][1] * y) + (mi[2][2] * z + mi[2][3]);
    const float ex = gamma(3) * (abs(mi[0][0] * x) + abs(mi[0][1] * y) + abs(mi[0][2] * z));
    const float ey = gamma(3) * (abs(mi[1][0] * x) + abs(mi[1][1] * y) + abs(mi[1][2] * z));
    const float ez = gamma(3) * (abs(mi[2][0] * x) + abs(mi[2][1] * y) + abs(mi[2][2] * z));
    Ivl ox = Ivl::FromValueAndError(xp, ex), oy = Ivl::FromValueAndError(yp, ey), oz = Ivl::FromValueAndError(zp, ez);
    const float wp = XfW(mi, o);
    if (wp != 1) { ox = DivF(ox, wp); oy = DivF(oy, wp); oz = DivF(oz, wp); }
    const V3 dd = XfVector3(mi, d);
    const float lengthSquared = LengthSquared(dd);
    if (lengthSquared > 0) {
        const V3 oError{(ox.hi - ox.lo) / 2, (oy.hi - oy.lo) / 2, (oz.hi - oz.lo) / 2};
        const float dt = Dot(Abs(dd), oError) / lengthSquared;
        const V3 off = dd * dt;
        ox = ox + Ivl(off.x); oy = oy + Ivl(off.y); oz = oz + Ivl(off.z);
        *tMax -= dt;
    }
    *oOut = V3{ox.mid(), oy.mid(), oz.mid()};
    *dOut = dd;
}

// AnimatedPrimitive::Intersect (cpu/primitive.cpp:140-153): the instance record with renderFromPrimitive.Interpolate(ray.time) in place of
// the static transformation — what InstanceRay / InstanceInteraction / InstanceWoP are then applied to.  A static instance is returned as it is.
// ANIM = false (the default): the caller cannot meet an animated instance — the interpolation (a quaternion slerp, a 4 x 4 inverse: an
// out-of-line callee) is then not reachable from it and does not set its register allocation (wf_scene.h "LEAN DEVICE VARIANTS").  The
// scene builder admits animated shapes / instances only where the consumers that ask for ANIM can meet them: ordinary materials, no media.
template <bool ANIM>
WF_HD const wf_instance &InstanceAt(const SceneView &sv, const wf_instance &in, float time, wf_instance *tmp) {
    if constexpr (!ANIM) return in;
    if (!sv.haveAnimated || in.anim_plus1 == 0) return in;
    tmp->def = in.def;
    tmp->anim_plus1 = in.anim_plus1;
    AnimatedInterpolateP(sv.animated + (in.anim_plus1 - 1), time, &tmp->render_from_instance);
    return *tmp;
}

// BVHAggregate::Intersect / IntersectP of an instance definition's own BVH (triangles only), on the shared stack above
// its current top.  tMax is updated in place; returns whether a hit was recorded.
template <typename Stack>
WF_HD bool BVHIntersectClosestDef(const SceneView &sv, int root, V3 o, V3 d, float *tMaxIO, Stack &stack, ClosestHit *out) {
    float tMax = *tMaxIO;
    bool hitAny = false;
    const int base = stack.n;
    V3 invDir{1 / d.x, 1 / d.y, 1 / d.z};
    int negMask = int(invDir.x < 0) | (int(invDir.y < 0) << 1) | (int(invDir.z < 0) << 2);
    int currentNodeIndex = root;
    while (true) {
        ++out->nodesVisited;
        const wf_bvh_node *node = &sv.bvhNodes[currentNodeIndex];
        if (BoxIntersectP(node->bmin, node->bmax, o, tMax, invDir, negMask)) {
            if (node->nprims > 0) {
                for (int i = 0; i < node->nprims; ++i) {
                    int tri = sv.bvhPrims[node->offset + i];
                    ++out->trisTested;
                    if (tri >= sv.nTriangles) {
                        // a quadric or bilinear patch of the definition, in the definition's space
                        QuadricHit qh;
                        if (QuadricIntersect(sv, tri, o, d, tMax, &qh)) {
                            out->prim = tri;
                            out->h.t = qh.tHit; out->h.b0 = qh.pObj.x; out->h.b1 = qh.pObj.y; out->h.b2 = qh.pObj.z;
                            tMax = qh.tHit;
                            hitAny = true;
                        }
                        continue;
                    }
                    V3 p0, p1, p2;
                    TriVerts(sv, tri, &p0, &p1, &p2);
                    TriHit h;
                    if (IntersectTriangle(o, d, tMax, p0, p1, p2, &h)) {
                        if (!sv.haveAlpha || AlphaTestPasses(sv, tri, h.b0, h.b1, h.b2, o, d)) {
                            out->prim = tri;
                            out->h = h;
                            tMax = h.t;
                            hitAny = true;
                        } else ++out->trisTested;  // the reference re-intersects the primitive with the respawned ray (cpu/primitive.cpp:62-70): one more test, never a hit
                    }
                }
                if (stack.n == base) break;
                currentNodeIndex = stack.pop();
            } else {
                if ((negMask >> node->axis) & 1) {
                    stack.push(currentNodeIndex + 1);
                    currentNodeIndex = node->offset;
                } else {
                    stack.push(node->offset);
                    currentNodeIndex = currentNodeIndex + 1;
                }
            }
        } else {
            if (stack.n == base) break;
            currentNodeIndex = stack.pop();
        }
    }
    *tMaxIO = tMax;
    return hitAny;
}
template <typename Stack>
WF_HD bool BVHIntersectAnyDef(const SceneView &sv, int root, V3 o, V3 d, float tMax, Stack &stack, int *nv, int *nt) {
    const int base = stack.n;
    V3 invDir{1.f / d.x, 1.f / d.y, 1.f / d.z};
    int negMask = int(invDir.x < 0) | (int(invDir.y < 0) << 1) | (int(invDir.z < 0) << 2);
    int currentNodeIndex = root;
    bool found = false;
    while (true) {
        ++*nv;
        const wf_bvh_node *node = &sv.bvhNodes[currentNodeIndex];
        if (BoxIntersectP(node->bmin, node->bmax, o, tMax, invDir, negMask)) {
            if (node->nprims > 0) {
                for (int i = 0; i < node->nprims && !found; ++i) {
                    int tri = sv.bvhPrims[node->offset + i];
                    ++*nt;
                    if (tri >= sv.nTriangles) {
                        QuadricHit qh;
                        if (QuadricIntersect(sv, tri, o, d, tMax, &qh)) found = true;
                        continue;
                    }
                    V3 p0, p1, p2;
                    TriVerts(sv, tri, &p0, &p1, &p2);
                    TriHit h;
                    if (IntersectTriangle(o, d, tMax, p0, p1, p2, &h)) {
                        if (!sv.haveAlpha || AlphaTestPasses(sv, tri, h.b0, h.b1, h.b2, o, d)) found = true;
                        else ++*nt;
                    }
                }
                if (found || stack.n == base) break;
                currentNodeIndex = stack.pop();
            } else {
                if ((negMask >> node->axis) & 1) {
                    stack.push(currentNodeIndex + 1);
                    currentNodeIndex = node->offset;
                } else {
                    stack.push(node->offset);
                    currentNodeIndex = currentNodeIndex + 1;
                }
            }
        } else {
            if (stack.n == base) break;
            currentNodeIndex = stack.pop();
        }
    }
    stack.n = base;  // an early out leaves the definition's entries behind
    return found;
}

// Reference-order BVH walk.  Stack is any type with push(int)/pop()/empty(); the HIP kernels pass an
// LDS-backed short stack (csrc/hip/wf_traverse.hip), the CPU checker a plain array.

template <bool ANIM = false, typename Stack>
WF_HD bool BVHIntersectClosest(const SceneView &sv, V3 o, V3 d, float tMax, Stack &stack, ClosestHit *out, float time = 0) {
    out->prim = -1;
    out->inst = -1;
    out->nodesVisited = 0;
    out->trisTested = 0;
    V3 invDir{1 / d.x, 1 / d.y, 1 / d.z};
    int negMask = int(invDir.x < 0) | (int(invDir.y < 0) << 1) | (int(invDir.z < 0) << 2);
    int currentNodeIndex = 0;
    while (true) {
        ++out->nodesVisited;
        const wf_bvh_node *node = &sv.bvhNodes[currentNodeIndex];
        if (BoxIntersectP(node->bmin, node->bmax, o, tMax, invDir, negMask)) {
            if (node->nprims > 0) {
                for (int i = 0; i < node->nprims; ++i) {
                    int tri = sv.bvhPrims[node->offset + i];
                    if (tri >= sv.nTriangles + sv.nQuadrics) {
                        // an object instance: TransformedPrimitive::Intersect (cpu/primitive.cpp:112-125).  tHit stays the
                        // instance ray's parameter (the reference does not add the origin shift dt back either).
                        const int inst = tri - sv.nTriangles - sv.nQuadrics;
                        wf_instance inTmp;
                        const wf_instance &in = InstanceAt<ANIM>(sv, sv.instances[inst], time, &inTmp);
                        float tI = tMax;
                        V3 oI, dI;
                        InstanceRay(in, o, d, &tI, &oI, &dI);
                        if (BVHIntersectClosestDef(sv, sv.instanceDefs[in.def].bvh_root, oI, dI, &tI, stack, out)) {
                            out->inst = inst;
                            tMax = tI;
                        }
                        continue;
                    }
                    ++out->trisTested;
                    if (tri >= sv.nTriangles) {
                        // a sphere: the hit record carries pObj in place of the barycentrics
                        QuadricHit qh;
                        if (QuadricIntersect(sv, tri, o, d, tMax, &qh)) {
                            out->prim = tri;
                            out->inst = -1;
                            out->h.t = qh.tHit; out->h.b0 = qh.pObj.x; out->h.b1 = qh.pObj.y; out->h.b2 = qh.pObj.z;
                            tMax = qh.tHit;
                        }
                        continue;
                    }
                    V3 p0, p1, p2;
                    TriVerts(sv, tri, &p0, &p1, &p2);
                    TriHit h;
                    if (IntersectTriangle(o, d, tMax, p0, p1, p2, &h)) {
                        if (!sv.haveAlpha || AlphaTestPasses(sv, tri, h.b0, h.b1, h.b2, o, d)) {
                            out->prim = tri;
                            out->inst = -1;
                            out->h = h;
                            tMax = h.t;
                        } else ++out->trisTested;  // see BVHIntersectClosestDef
                    }
                }
                if (stack.empty()) break;
                currentNodeIndex = stack.pop();
            } else {
                if ((negMask >> node->axis) & 1) {
                    stack.push(currentNodeIndex + 1);
                    currentNodeIndex = node->offset;
                } else {
                    stack.push(node->offset);
                    currentNodeIndex = currentNodeIndex + 1;
                }
            }
        } else {
            if (stack.empty()) break;
            currentNodeIndex = stack.pop();
        }
    }
    return out->prim >= 0;
}

template <bool ANIM = false, typename Stack>
WF_HD bool BVHIntersectAny(const SceneView &sv, V3 o, V3 d, float tMax, Stack &stack, int *nodesVisited, int *trisTested, float time = 0) {
    V3 invDir{1.f / d.x, 1.f / d.y, 1.f / d.z};
    int negMask = int(invDir.x < 0) | (int(invDir.y < 0) << 1) | (int(invDir.z < 0) << 2);
    int currentNodeIndex = 0;
    int nv = 0, nt = 0;
    bool found = false;
    while (true) {
        ++nv;
        const wf_bvh_node *node = &sv.bvhNodes[currentNodeIndex];
        if (BoxIntersectP(node->bmin, node->bmax, o, tMax, invDir, negMask)) {
            if (node->nprims > 0) {
                for (int i = 0; i < node->nprims && !found; ++i) {
                    int tri = sv.bvhPrims[node->offset + i];
                    if (tri >= sv.nTriangles + sv.nQuadrics) {
                        wf_instance inTmp;
                        const wf_instance &in = InstanceAt<ANIM>(sv, sv.instances[tri - sv.nTriangles - sv.nQuadrics], time, &inTmp);
                        float tI = tMax;
                        V3 oI, dI;
                        InstanceRay(in, o, d, &tI, &oI, &dI);
                        if (BVHIntersectAnyDef(sv, sv.instanceDefs[in.def].bvh_root, oI, dI, tI, stack, &nv, &nt)) found = true;
                        continue;
                    }
                    ++nt;
                    if (tri >= sv.nTriangles) {
                        QuadricHit qh;
                        if (QuadricIntersect(sv, tri, o, d, tMax, &qh)) found = true;
                        continue;
                    }
                    V3 p0, p1, p2;
                    TriVerts(sv, tri, &p0, &p1, &p2);
                    TriHit h;
                    if (IntersectTriangle(o, d, tMax, p0, p1, p2, &h)) {
                        if (!sv.haveAlpha || AlphaTestPasses(sv, tri, h.b0, h.b1, h.b2, o, d)) found = true;
                        else ++nt;
                    }
                }
                if (found || stack.empty()) break;
                currentNodeIndex = stack.pop();
            } else {
                if ((negMask >> node->axis) & 1) {
                    stack.push(currentNodeIndex + 1);
                    currentNodeIndex = node->offset;
                } else {
                    stack.push(node->offset);
                    currentNodeIndex = currentNodeIndex + 1;
                }
            }
        } else {
            if (stack.empty()) break;
            currentNodeIndex = stack.pop();
        }
    }
    if (nodesVisited) *nodesVisited = nv;
    if (trisTested) *trisTested = nt;
    return found;
}

struct ArrayStack {  // int nodesToVisit[64], cpu/aggregates.cpp:538
    int s[128];  // the top-level walk's entries + an instance definition's on top of them
    int n = 0;
    WF_HD void push(int v) { s[n++] = v; }
    WF_HD int pop() { return s[--n]; }
    WF_HD bool empty() const { return n == 0; }
};

// ---------------------------------------------------------------------------------------------
// SurfaceInteraction as the material kernels need it (interaction.h:131-260)
struct SurfIntr {
    P3i pi;
    N3 n;
    V2 uv;
    V3 dpdu, dpdv;
    N3 dndu, dndv;
    N3 ns;            // shading.n
    V3 dpdus, dpdvs;  // shading.dpdu/dpdv
    N3 dndus, dndvs;
    int mesh;
};

// vector-valued DifferenceOfProducts(float, Tuple, float, Tuple) (util/math.h:569-575 over Tuple3 FMA)
WF_HD V3 DifferenceOfProductsV(float a, V3 b, float c, V3 d) {
    return V3{DifferenceOfProducts(a, b.x, c, d.x), DifferenceOfProducts(a, b.y, c, d.y), DifferenceOfProducts(a, b.z, c, d.z)};
}
WF_HD N3 DifferenceOfProductsN(float a, N3 b, float c, N3 d) {
    return N3{DifferenceOfProducts(a, b.x, c, d.x), DifferenceOfProducts(a, b.y, c, d.y), DifferenceOfProducts(a, b.z, c, d.z)};
}

// Triangle::InteractionFromIntersection, shapes.h:884-1010.  `full` = also the shading derivatives.
WF_HD void TriangleInteraction(const SceneView &sv, int tri, float b0, float b1, float b2, SurfIntr *si) {
    const auto v = sv.triIndices + 3 * (size_t)tri;
    const int meshId = sv.triMesh[tri];
    const wf_mesh mesh = sv.meshes[meshId];
    si->mesh = meshId;
    V3 p0, p1, p2;
    V2 uv0{0, 0}, uv1{1, 0}, uv2{1, 1};
    N3 n0{0, 0, 0}, n1{0, 0, 0}, n2{0, 0, 0};
    const bool deindexed = (const ShadeTri *)sv.shadeTris != nullptr;   // (the same values either way: the record is a copy of the tables' entries)
    if (deindexed) {
        const ShadeTri st = sv.shadeTris[tri];
        p0 = V3{st.p[0], st.p[1], st.p[2]}; p1 = V3{st.p[3], st.p[4], st.p[5]}; p2 = V3{st.p[6], st.p[7], st.p[8]};
        if (mesh.flags & WF_MESH_HAS_UV) { uv0 = V2{st.uv[0], st.uv[1]}; uv1 = V2{st.uv[2], st.uv[3]}; uv2 = V2{st.uv[4], st.uv[5]}; }
        if (mesh.flags & WF_MESH_HAS_N) { n0 = N3{st.n[0], st.n[1], st.n[2]}; n1 = N3{st.n[3], st.n[4], st.n[5]}; n2 = N3{st.n[6], st.n[7], st.n[8]}; }
    } else {
        p0 = LoadP(sv, v[0]); p1 = LoadP(sv, v[1]); p2 = LoadP(sv, v[2]);
        if (mesh.flags & WF_MESH_HAS_UV) { uv0 = LoadUV(sv, v[0]); uv1 = LoadUV(sv, v[1]); uv2 = LoadUV(sv, v[2]); }
    }
    V2 duv02{uv0.x - uv2.x, uv0.y - uv2.y}, duv12{uv1.x - uv2.x, uv1.y - uv2.y};
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = DifferenceOfProducts(duv02.x, duv12.y, duv02.y, duv12.x);
    V3 dpdu{0, 0, 0}, dpdv{0, 0, 0};
    bool degenerateUV = abs(determinant) < 1e-9f;
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        dpdu = DifferenceOfProductsV(duv12.y, dp02, duv02.y, dp12) * invdet;
        dpdv = DifferenceOfProductsV(duv02.x, dp12, duv12.x, dp02) * invdet;
    }
    if (degenerateUV || LengthSquared(Cross(dpdu, dpdv)) == 0) {
        V3 ng = Cross(p2 - p0, p1 - p0);
        if (LengthSquared(ng) == 0) {
            // Cross(Vector3<double>, Vector3<double>) — the double DifferenceOfProducts, util/math.h:569
            V3 a = p2 - p0, b = p1 - p0;
            double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
            auto dop = [](double a, double b, double c, double d) {
                double cd = c * d;
                double r = ::fma(a, b, -cd);
                double e = ::fma(-c, d, cd);
                return r + e;
            };
            ng = V3{(float)dop(ay, bz, az, by), (float)dop(az, bx, ax, bz), (float)dop(ax, by, ay, bx)};
        }
        CoordinateSystem(Normalize(ng), &dpdu, &dpdv);
    }
    V3 pHit = b0 * p0 + b1 * p1 + b2 * p2;
    V2 uvHit{b0 * uv0.x + b1 * uv1.x + b2 * uv2.x, b0 * uv0.y + b1 * uv1.y + b2 * uv2.y};
    bool flipNormal = (mesh.flags & WF_MESH_FLIP_NORMAL) != 0;
    V3 pAbsSum = Abs(b0 * p0) + Abs(b1 * p1) + Abs(b2 * p2);
    V3 pError = gamma(7) * pAbsSum;
    si->pi = MakeP3i(pHit, pError);
    si->uv = uvHit;
    si->dpdu = dpdu;
    si->dpdv = dpdv;
    si->dndu = si->dndv = N3{0, 0, 0};
    // SurfaceInteraction ctor (interaction.h:164-183) computes n from dpdu x dpdv and flips it; the
    // triangle then overrides n and shading.n (shapes.h:936-938)
    N3 n = toN(Normalize(Cross(dp02, dp12)));
    if (flipNormal) n = -n;
    si->n = n;
    si->ns = n;
    si->dpdus = dpdu;
    si->dpdvs = dpdv;
    si->dndus = si->dndvs = N3{0, 0, 0};
    if (mesh.flags & (WF_MESH_HAS_N | WF_MESH_HAS_S)) {   // shapes.h:940: mesh->n || mesh->s
        const bool hasN = mesh.flags & WF_MESH_HAS_N;
        N3 ns = si->n;
        if (hasN) {
            if (!deindexed) { n0 = LoadN(sv, v[0]); n1 = LoadN(sv, v[1]); n2 = LoadN(sv, v[2]); }
            ns = b0 * n0 + b1 * n1 + b2 * n2;
            ns = LengthSquared(ns) > 0 ? Normalize(ns) : si->n;
        }
        V3 ss = si->dpdu;
        if (mesh.flags & WF_MESH_HAS_S) {   // the interpolated "S" tangent (shapes.h:951-959)
            ss = b0 * LoadS(sv, mesh, v[0]) + b1 * LoadS(sv, mesh, v[1]) + b2 * LoadS(sv, mesh, v[2]);
            if (LengthSquared(ss) == 0) ss = si->dpdu;
        }
        V3 ts = Cross(ns, ss);
        if (LengthSquared(ts) > 0) ss = Cross(ts, ns);
        else CoordinateSystem(toV(ns), &ss, &ts);
        N3 dndu, dndv;
        N3 dn1 = n0 - n2, dn2 = n1 - n2;
        float det2 = DifferenceOfProducts(duv02.x, duv12.y, duv02.y, duv12.x);
        bool degUV = abs(det2) < 1e-9;  // double comparison in the reference (shapes.h:984)
        if (!hasN) dndu = dndv = N3{0, 0, 0};   // shapes.h:1003-1004
        else if (degUV) {
            V3 dn = Cross(toV(n2 - n0), toV(n1 - n0));
            if (LengthSquared(dn) == 0) dndu = dndv = N3{0, 0, 0};
            else {
                V3 dnu, dnv;
                CoordinateSystem(dn, &dnu, &dnv);
                dndu = toN(dnu);
                dndv = toN(dnv);
            }
        } else {
            float invDet = 1 / det2;
            dndu = DifferenceOfProductsN(duv12.y, dn1, duv02.y, dn2) * invDet;
            dndv = DifferenceOfProductsN(duv02.x, dn2, duv12.x, dn1) * invDet;
        }
        // SetShadingGeometry(ns, ss, ts, dndu, dndv, true), interaction.h:194-214
        si->ns = ns;
        si->n = FaceForward(si->n, si->ns);
        si->dpdus = ss;
        si->dpdvs = ts;
        si->dndus = dndu;
        si->dndvs = dndv;
        while (LengthSquared(si->dpdus) > 1e16f || LengthSquared(si->dpdvs) > 1e16f) {
            si->dpdus = si->dpdus / 1e8f;
            si->dpdvs = si->dpdvs / 1e8f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle sampling (shapes.h:845-880, 1013-1180)
WF_HD float TriangleArea(V3 p0, V3 p1, V3 p2) { return 0.5f * Length(Cross(p1 - p0, p2 - p0)); }
WF_HD float TriangleSolidAngle(V3 p0, V3 p1, V3 p2, V3 p) {
    return SphericalTriangleArea(Normalize(p0 - p), Normalize(p1 - p), Normalize(p2 - p));
}
constexpr float MinSphericalSampleArea = 3e-4;
constexpr float MaxSphericalSampleArea = 6.22;

struct ShapeSampleR { P3i pi; N3 n; V2 uv; float pdf; bool valid; };

WF_HD N3 TriSampleNormal(const SceneView &sv, const wf_mesh &mesh, const int32_t *v, V3 p0, V3 p1, V3 p2, float b0, float b1) {
    N3 n = Normalize(toN(Cross(p1 - p0, p2 - p0)));
    if (mesh.flags & WF_MESH_HAS_N) {
        N3 ns = b0 * LoadN(sv, v[0]) + b1 * LoadN(sv, v[1]) + (1 - b0 - b1) * LoadN(sv, v[2]);
        n = FaceForward(n, ns);
    } else if (mesh.flags & WF_MESH_FLIP_NORMAL) n = n * -1.f;
    return n;
}
WF_HD V2 TriSampleUV(const SceneView &sv, const wf_mesh &mesh, const int32_t *v, const float b[3]) {
    V2 uv0{0, 0}, uv1{1, 0}, uv2{1, 1};
    if (mesh.flags & WF_MESH_HAS_UV) { uv0 = LoadUV(sv, v[0]); uv1 = LoadUV(sv, v[1]); uv2 = LoadUV(sv, v[2]); }
    return V2{b[0] * uv0.x + b[1] * uv1.x + b[2] * uv2.x, b[0] * uv0.y + b[1] * uv1.y + b[2] * uv2.y};
}

// Triangle::Sample(const ShapeSampleContext &, Point2f u).  ctx: reference point interval, n, ns.
WF_HD ShapeSampleR TriangleSample(const SceneView &sv, int tri, const P3i &ctxPi, N3 ctxNs, V2 u) {
    ShapeSampleR r{};
    r.valid = false;
    const auto v = sv.triIndices + 3 * (size_t)tri;
    const wf_mesh mesh = sv.meshes[sv.triMesh[tri]];
    V3 p0 = LoadP(sv, v[0]), p1 = LoadP(sv, v[1]), p2 = LoadP(sv, v[2]);
    V3 rp = ctxPi.mid();
    float solidAngle = TriangleSolidAngle(p0, p1, p2, rp);
    if (solidAngle < MinSphericalSampleArea || solidAngle > MaxSphericalSampleArea) {
        // Triangle::Sample(Point2f u), shapes.h:1013-1047
        float b[3];
        SampleUniformTriangle(u, b);
        V3 p = b[0] * p0 + b[1] * p1 + b[2] * p2;
        N3 n = TriSampleNormal(sv, mesh, v, p0, p1, p2, b[0], b[1]);
        V2 uvS = TriSampleUV(sv, mesh, v, b);
        V3 pAbsSum = Abs(b[0] * p0) + Abs(b[1] * p1) + Abs((1 - b[0] - b[1]) * p2);
        V3 pError = gamma(6) * pAbsSum;
        r.pi = MakeP3i(p, pError);
        r.n = n;
        r.uv = uvS;
        r.pdf = 1 / TriangleArea(p0, p1, p2);
        V3 wi = r.pi.mid() - rp;
        if (LengthSquared(wi) == 0) return r;
        wi = Normalize(wi);
        r.pdf /= AbsDot(r.n, -wi) / DistanceSquared(rp, r.pi.mid());
        if (IsInf(r.pdf)) return r;
        r.valid = true;
        return r;
    }
    float pdf = 1;
    if (!IsZero(ctxNs)) {
        V3 wi0 = Normalize(p0 - rp), wi1 = Normalize(p1 - rp), wi2 = Normalize(p2 - rp);
        float w[4] = {fmax(0.01f, AbsDot(ctxNs, wi1)), fmax(0.01f, AbsDot(ctxNs, wi1)), fmax(0.01f, AbsDot(ctxNs, wi0)),
                      fmax(0.01f, AbsDot(ctxNs, wi2))};
        u = SampleBilinear(u, w);
        pdf = BilinearPDF(u, w);
    }
    float triPDF;
    float b[3];
    SampleSphericalTriangle(p0, p1, p2, rp, u, b, &triPDF);
    if (triPDF == 0) return r;
    pdf *= triPDF;
    V3 pAbsSum = Abs(b[0] * p0) + Abs(b[1] * p1) + Abs((1 - b[0] - b[1]) * p2);
    V3 pError = gamma(6) * pAbsSum;
    V3 p = b[0] * p0 + b[1] * p1 + b[2] * p2;
    r.n = TriSampleNormal(sv, mesh, v, p0, p1, p2, b[0], b[1]);
    r.uv = TriSampleUV(sv, mesh, v, b);
    r.pi = MakeP3i(p, pError);
    r.pdf = pdf;
    r.valid = true;
    return r;
}

// Triangle::PDF(const ShapeSampleContext &, Vector3f wi), shapes.h:1133-1171
WF_HD float TrianglePDF(const SceneView &sv, int tri, const P3i &ctxPi, N3 ctxN, N3 ctxNs, V3 wi) {
    const auto v = sv.triIndices + 3 * (size_t)tri;
    V3 p0 = LoadP(sv, v[0]), p1 = LoadP(sv, v[1]), p2 = LoadP(sv, v[2]);
    V3 rp = ctxPi.mid();
    float solidAngle = TriangleSolidAngle(p0, p1, p2, rp);
    if (solidAngle < MinSphericalSampleArea || solidAngle > MaxSphericalSampleArea) {
        // ctx.SpawnRay(wi) (shapes.h:63-90), Triangle::Intersect with tMax = Infinity
        V3 o = OffsetRayOrigin(ctxPi, ctxN, wi);
        TriHit h;
#if defined(WF_COUNT_TRI_TESTS) && !defined(__HIP_DEVICE_COMPILE__)
        WF_COUNT_TRI_TESTS;  // the CPU checker's statistics: this is one of the reference's "Ray-Triangle intersection tests"
#endif
        if (!IntersectTriangle(o, wi, WF_INFINITY, p0, p1, p2, &h)) return 0;
        SurfIntr si;
        TriangleInteraction(sv, tri, h.b0, h.b1, h.b2, &si);
        float pdf = (1 / TriangleArea(p0, p1, p2)) / (AbsDot(si.n, -wi) / DistanceSquared(rp, si.pi.mid()));
        if (IsInf(pdf)) pdf = 0;
        return pdf;
    }
    float pdf = 1 / solidAngle;
    if (!IsZero(ctxNs)) {
        V2 u = InvertSphericalTriangleSample(p0, p1, p2, rp, wi);
        V3 wi0 = Normalize(p0 - rp), wi1 = Normalize(p1 - rp), wi2 = Normalize(p2 - rp);
        float w[4] = {fmax(0.01f, AbsDot(ctxNs, wi1)), fmax(0.01f, AbsDot(ctxNs, wi1)), fmax(0.01f, AbsDot(ctxNs, wi0)),
                      fmax(0.01f, AbsDot(ctxNs, wi2))};
        pdf *= BilinearPDF(u, w);
    }
    return pdf;
}

// ---------------------------------------------------------------------------------------------
// Sphere::InteractionFromIntersection (shapes.h:241-289) + the SurfaceInteraction ctor (interaction.h:164-183) +
// Transform::operator()(SurfaceInteraction) (util/transform.cpp:229-261)
// (out of line, pointer / scalar arguments only: the quadric code stays out of the material and traversal kernels'
// register budgets, which scenes without spheres would otherwise pay for)
WF_NI void SphereInteractionP(const wf_quadric *sp, int meshFlags, float px, float py, float pz, SurfIntr *si) {
    const wf_quadric s = *sp;
    V3 pHit{px, py, pz};
    const float radius = s.radius, phiMax = s.phi_max, thetaZMin = s.theta_z_min, thetaZMax = s.theta_z_max;
    float phi = atan2(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * Pi;
    const bool flipN = (meshFlags & WF_MESH_FLIP_NORMAL) != 0;
    if (s.type != WF_QUADRIC_SPHERE) {
        float u = phi / phiMax, v;
        V3 dpdu{-phiMax * pHit.y, phiMax * pHit.x, 0}, dpdv, pError;
        N3 dndu{0, 0, 0}, dndv{0, 0, 0};
        if (s.type == WF_QUADRIC_DISK) {
            // Disk::InteractionFromIntersection, shapes.h:455-482
            float rHit = sqrt(Sqr(pHit.x) + Sqr(pHit.y));
            v = (radius - rHit) / (radius - s.inner_radius);
            dpdv = V3{pHit.x, pHit.y, 0} * (s.inner_radius - radius) / rHit;
            pHit.z = s.z_min;
            pError = V3{0, 0, 0};
        } else {
            // Cylinder::InteractionFromIntersection, shapes.h:662-698
            v = (pHit.z - s.z_min) / (s.z_max - s.z_min);
            dpdv = V3{0, 0, s.z_max - s.z_min};
            V3 d2Pduu = -phiMax * phiMax * V3{pHit.x, pHit.y, 0};
            V3 d2Pduv{0, 0, 0}, d2Pdvv{0, 0, 0};
            float E = Dot(dpdu, dpdu), F = Dot(dpdu, dpdv), G = Dot(dpdv, dpdv);
            V3 nn = Normalize(Cross(dpdu, dpdv));
            float e = Dot(nn, d2Pduu), f = Dot(nn, d2Pduv), g = Dot(nn, d2Pdvv);
            float EGF2 = DifferenceOfProducts(E, G, F, F);
            float invEGF2 = (EGF2 == 0) ? 0.f : 1 / EGF2;
            dndu = toN((f * F - e * G) * invEGF2 * dpdu + (e * F - f * E) * invEGF2 * dpdv);
            dndv = toN((g * F - f * G) * invEGF2 * dpdu + (f * F - g * E) * invEGF2 * dpdv);
            pError = gamma(3) * Abs(V3{pHit.x, pHit.y, 0});
        }
        N3 nObj = toN(Normalize(Cross(dpdu, dpdv)));
        if (flipN) nObj = -nObj;
        const float(*m)[4] = s.render_from_object.m;
        const float(*mi)[4] = s.render_from_object.mInv;
        si->pi = XfPointI(m, pHit, pError);
        si->n = Normalize(XfNormal3(mi, nObj));
        si->uv = V2{u, v};
        si->dpdu = XfVector3(m, dpdu);
        si->dpdv = XfVector3(m, dpdv);
        si->dndu = XfNormal3(mi, dndu);
        si->dndv = XfNormal3(mi, dndv);
        si->ns = FaceForward(Normalize(XfNormal3(mi, nObj)), si->n);
        si->dpdus = si->dpdu;
        si->dpdvs = si->dpdv;
        si->dndus = si->dndu;
        si->dndvs = si->dndv;
        si->mesh = s.mesh;
        return;
    }
    float u = phi / phiMax;
    float cosTheta = pHit.z / radius;
    float theta = SafeACos(cosTheta);
    float v = (theta - thetaZMin) / (thetaZMax - thetaZMin);
    float zRadius = sqrt(Sqr(pHit.x) + Sqr(pHit.y));
    float cosPhi = pHit.x / zRadius, sinPhi = pHit.y / zRadius;
    V3 dpdu{-phiMax * pHit.y, phiMax * pHit.x, 0};
    float sinTheta = SafeSqrt(1 - Sqr(cosTheta));
    V3 dpdv = (thetaZMax - thetaZMin) * V3{pHit.z * cosPhi, pHit.z * sinPhi, -radius * sinTheta};
    V3 d2Pduu = -phiMax * phiMax * V3{pHit.x, pHit.y, 0};
    V3 d2Pduv = (thetaZMax - thetaZMin) * pHit.z * phiMax * V3{-sinPhi, cosPhi, 0.f};
    V3 d2Pdvv = -Sqr(thetaZMax - thetaZMin) * V3{pHit.x, pHit.y, pHit.z};
    float E = Dot(dpdu, dpdu), F = Dot(dpdu, dpdv), G = Dot(dpdv, dpdv);
    V3 n = Normalize(Cross(dpdu, dpdv));
    float e = Dot(n, d2Pduu), f = Dot(n, d2Pduv), g = Dot(n, d2Pdvv);
    float EGF2 = DifferenceOfProducts(E, G, F, F);
    float invEGF2 = (EGF2 == 0) ? 0.f : 1 / EGF2;
    N3 dndu = toN((f * F - e * G) * invEGF2 * dpdu + (e * F - f * E) * invEGF2 * dpdv);
    N3 dndv = toN((g * F - f * G) * invEGF2 * dpdu + (f * F - g * E) * invEGF2 * dpdv);
    V3 pError = gamma(5) * Abs(pHit);
    bool flipNormal = (meshFlags & WF_MESH_FLIP_NORMAL) != 0;
    N3 nObj = toN(n);
    if (flipNormal) nObj = -nObj;
    const float(*m)[4] = s.render_from_object.m;
    const float(*mi)[4] = s.render_from_object.mInv;
    si->pi = XfPointI(m, pHit, pError);
    si->n = Normalize(XfNormal3(mi, nObj));
    si->uv = V2{u, v};
    si->dpdu = XfVector3(m, dpdu);
    si->dpdv = XfVector3(m, dpdv);
    si->dndu = XfNormal3(mi, dndu);
    si->dndv = XfNormal3(mi, dndv);
    si->ns = FaceForward(Normalize(XfNormal3(mi, nObj)), si->n);
    si->dpdus = si->dpdu;
    si->dpdvs = si->dpdv;
    si->dndus = si->dndu;
    si->dndvs = si->dndv;
    si->mesh = s.mesh;
}
// BilinearPatch::InteractionFromIntersection (shapes.h:1396-1497).  Out of line: pointer arguments only.
WF_NI void BilinearInteractionP(const wf_quadric *sp, int meshFlags, float u, float v, SurfIntr *si) {
    const BlpData d = LoadBlp(*sp);
    const V3 p00 = d.p00, p10 = d.p10, p01 = d.p01, p11 = d.p11;
    V3 p = LerpV(u, LerpV(v, p00, p01), LerpV(v, p10, p11));
    V3 dpdu = LerpV(v, p10, p11) - LerpV(v, p00, p01);
    V3 dpdv = LerpV(u, p01, p11) - LerpV(u, p00, p10);
    V2 st{u, v};
    float duds = 1, dudt = 0, dvds = 0, dvdt = 1;
    auto lerp2 = [](float t, V2 a, V2 b) { return V2{(1 - t) * a.x + t * b.x, (1 - t) * a.y + t * b.y}; };
    if (d.hasUV) {
        st = lerp2(u, lerp2(v, d.uv00, d.uv01), lerp2(v, d.uv10, d.uv11));
        V2 e0 = lerp2(v, d.uv10, d.uv11), e1 = lerp2(v, d.uv00, d.uv01), f0 = lerp2(u, d.uv01, d.uv11), f1 = lerp2(u, d.uv00, d.uv10);
        V2 dstdu{e0.x - e1.x, e0.y - e1.y}, dstdv{f0.x - f1.x, f0.y - f1.y};
        duds = abs(dstdu.x) < 1e-8f ? 0 : 1 / dstdu.x;
        dvds = abs(dstdv.x) < 1e-8f ? 0 : 1 / dstdv.x;
        dudt = abs(dstdu.y) < 1e-8f ? 0 : 1 / dstdu.y;
        dvdt = abs(dstdv.y) < 1e-8f ? 0 : 1 / dstdv.y;
        V3 dpds = dpdu * duds + dpdv * dvds;
        V3 dpdt = dpdu * dudt + dpdv * dvdt;
        V3 cr = Cross(dpds, dpdt);
        if (cr.x != 0 || cr.y != 0 || cr.z != 0) {
            if (Dot(Cross(dpdu, dpdv), cr) < 0) dpdt = -dpdt;
            dpdu = dpds;
            dpdv = dpdt;
        }
    }
    V3 d2Pduu{0, 0, 0}, d2Pdvv{0, 0, 0};
    V3 d2Pduv = (p00 - p01) + (p11 - p10);
    float E = Dot(dpdu, dpdu), F = Dot(dpdu, dpdv), G = Dot(dpdv, dpdv);
    V3 n = Normalize(Cross(dpdu, dpdv));
    float e = Dot(n, d2Pduu), f = Dot(n, d2Pduv), g = Dot(n, d2Pdvv);
    float EGF2 = DifferenceOfProducts(E, G, F, F);
    float invEGF2 = (EGF2 == 0) ? 0.f : 1 / EGF2;
    N3 dndu = toN((f * F - e * G) * invEGF2 * dpdu + (e * F - f * E) * invEGF2 * dpdv);
    N3 dndv = toN((g * F - f * G) * invEGF2 * dpdu + (f * F - g * E) * invEGF2 * dpdv);
    N3 dnds = dndu * duds + dndv * dvds;
    N3 dndt = dndu * dudt + dndv * dvdt;
    dndu = dnds;
    dndv = dndt;
    V3 pAbsSum = Abs(p00) + Abs(p01) + Abs(p10) + Abs(p11);
    V3 pError = gamma(6) * pAbsSum;
    // SurfaceInteraction ctor (interaction.h:140-162)
    si->pi = MakeP3i(p, pError);
    si->uv = st;
    si->dpdu = dpdu; si->dpdv = dpdv; si->dndu = dndu; si->dndv = dndv;
    N3 ng = toN(Normalize(Cross(dpdu, dpdv)));
    N3 ns = ng;
    if (meshFlags & WF_MESH_FLIP_NORMAL) { ng = -ng; ns = -ns; }
    si->n = ng; si->ns = ns;
    si->dpdus = dpdu; si->dpdvs = dpdv; si->dndus = dndu; si->dndvs = dndv;
    si->mesh = sp->mesh;
    if (d.hasN) {
        auto lerpN = [](float t, N3 a, N3 b) { return (1 - t) * a + t * b; };
        N3 nsI = lerpN(u, lerpN(v, d.n00, d.n01), lerpN(v, d.n10, d.n11));
        if (LengthSquared(nsI) > 0) {
            nsI = Normalize(nsI);
            N3 dnduS = lerpN(v, d.n10, d.n11) - lerpN(v, d.n00, d.n01);
            N3 dndvS = lerpN(u, d.n01, d.n11) - lerpN(u, d.n00, d.n10);
            N3 dndsS = dnduS * duds + dndvS * dvds;
            N3 dndtS = dnduS * dudt + dndvS * dvdt;
            // RotateFromTo(Normalize(isect.n), ns) (util/transform.h:249-270) applied to dpdu, dpdv
            V3 from = Normalize(toV(si->n)), to = toV(nsI);
            V3 refl;
            if (abs(from.x) < 0.72f && abs(to.x) < 0.72f) refl = V3{1, 0, 0};
            else if (abs(from.y) < 0.72f && abs(to.y) < 0.72f) refl = V3{0, 1, 0};
            else refl = V3{0, 0, 1};
            V3 uu = refl - from, vv = refl - to;
            float r[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    r[i][j] = ((i == j) ? 1 : 0) - 2 / Dot(uu, uu) * uu[i] * uu[j] - 2 / Dot(vv, vv) * vv[i] * vv[j] + 4 * Dot(uu, vv) / (Dot(uu, uu) * Dot(vv, vv)) * vv[i] * uu[j];
            auto rot = [&](V3 w) { return V3{r[0][0] * w.x + r[0][1] * w.y + r[0][2] * w.z, r[1][0] * w.x + r[1][1] * w.y + r[1][2] * w.z, r[2][0] * w.x + r[2][1] * w.y + r[2][2] * w.z}; };
            // SetShadingGeometry(ns, r(dpdu), r(dpdv), dndu, dndv, true) (interaction.h:186-200)
            si->ns = nsI;
            si->n = FaceForward(si->n, si->ns);
            si->dpdus = rot(dpdu);
            si->dpdvs = rot(dpdv);
            si->dndus = dndsS;
            si->dndvs = dndtS;
        }
    }
}


// The SurfaceInteraction Curve::RecursiveIntersect builds at its accepted leaf (shapes.cpp:683-727), rebuilt from the hit
// record (u, v, tHit) and the ray the primitive was intersected with (ro, rd: in the space the primitive lives in).
WF_NI void CurveInteractionP(const wf_quadric *sp, int meshFlags, float u, float v, float tHit, float rox, float roy, float roz,
                             float rdx, float rdy, float rdz, SurfIntr *si) {
    const wf_quadric &s = *sp;
    const CurveData c = LoadCurve(s);
    V3 o, d;
    XfRayOffset(s.render_from_object.mInv, V3{rox, roy, roz}, V3{rdx, rdy, rdz}, &o, &d);
    const float rayLength = Length(d);
    float hitWidth = Lerp(u, c.width0, c.width1);
    N3 nHit{0, 0, 0};
    if (c.type == 2) {
        nHit = CurveRibbonNormal(c, u);
        hitWidth *= AbsDot(nHit, d) / rayLength;
    }
    V3 dpdu, dpdv;
    EvaluateCubicBezierD(c.cp, u, &dpdu);
    if (c.type == 2) dpdv = Normalize(Cross(nHit, dpdu)) * hitWidth;
    else {
        V3 cpObj[4];
        M44 rayFromObject, objectFromRay;
        CurveRayFrame(c, o, d, cpObj, &rayFromObject, &objectFromRay);
        V3 dpduPlane = XfVector3(rayFromObject.m, dpdu);   // objectFromRay.ApplyInverse(dpdu)
        V3 dpdvPlane = Normalize(V3{-dpduPlane.y, dpduPlane.x, 0}) * hitWidth;
        if (c.type == 1) {
            // Rotate(-theta, dpduPlane) (util/transform.h:243-283) applied to a vector
            float theta = Lerp(v, -90.f, 90.f);
            float sinTheta = sin(Radians(-theta)), cosTheta = cos(Radians(-theta));
            V3 a = Normalize(dpduPlane);
            float m00 = a.x * a.x + (1 - a.x * a.x) * cosTheta, m01 = a.x * a.y * (1 - cosTheta) - a.z * sinTheta, m02 = a.x * a.z * (1 - cosTheta) + a.y * sinTheta;
            float m10 = a.x * a.y * (1 - cosTheta) + a.z * sinTheta, m11 = a.y * a.y + (1 - a.y * a.y) * cosTheta, m12 = a.y * a.z * (1 - cosTheta) - a.x * sinTheta;
            float m20 = a.x * a.z * (1 - cosTheta) - a.y * sinTheta, m21 = a.y * a.z * (1 - cosTheta) + a.x * sinTheta, m22 = a.z * a.z + (1 - a.z * a.z) * cosTheta;
            V3 q = dpdvPlane;
            dpdvPlane = V3{m00 * q.x + m01 * q.y + m02 * q.z, m10 * q.x + m11 * q.y + m12 * q.z, m20 * q.x + m21 * q.y + m22 * q.z};
        }
        dpdv = XfVector3(objectFromRay.m, dpdvPlane);
    }
    V3 pError{hitWidth, hitWidth, hitWidth};
    V3 pHit = o + d * tHit;   // ray(tHit)
    bool flipNormal = (meshFlags & WF_MESH_FLIP_NORMAL) != 0;
    N3 nObj = toN(Normalize(Cross(dpdu, dpdv)));   // SurfaceInteraction ctor (interaction.h:142-160)
    if (flipNormal) nObj = -nObj;
    const float(*m)[4] = s.render_from_object.m;
    const float(*mi)[4] = s.render_from_object.mInv;
    si->pi = XfPointI(m, pHit, pError);
    si->n = Normalize(XfNormal3(mi, nObj));
    si->uv = V2{u, v};
    si->dpdu = XfVector3(m, dpdu);
    si->dpdv = XfVector3(m, dpdv);
    si->dndu = XfNormal3(mi, N3{0, 0, 0});
    si->dndv = XfNormal3(mi, N3{0, 0, 0});
    si->ns = FaceForward(Normalize(XfNormal3(mi, nObj)), si->n);
    si->dpdus = si->dpdu;
    si->dpdvs = si->dpdv;
    si->dndus = si->dndu;
    si->dndvs = si->dndv;
    si->mesh = s.mesh;
}
WF_HD void SphereInteraction(const SceneView &sv, int prim, V3 pHit, SurfIntr *si) {
    const wf_quadric *s = sv.quadrics + (prim - sv.nTriangles);
    SurfIntr tmp;  // the out-of-line call's result lives in memory; *si stays in registers
    if (s->type == WF_QUADRIC_BILINEAR) BilinearInteractionP(s, sv.meshes[s->mesh].flags, pHit.x, pHit.y, &tmp);
    else
    SphereInteractionP(s, sv.meshes[s->mesh].flags, pHit.x, pHit.y, pHit.z, &tmp);
    *si = tmp;
}
// intr.wo: the Interaction ctor normalises -ray.d (interaction.h:40-43); a quadric builds its interaction in object
// space and transforms it back, so its wo is normalised there and again after the transform (shapes.h:286-288,
// util/transform.cpp:235)
WF_NI void SphereWoP(const wf_quadric *s, float x, float y, float z, float *ox, float *oy, float *oz) {
    if (s->type == WF_QUADRIC_BILINEAR) {  // built in render space like a triangle's: normalised once
        V3 w = Normalize(V3{x, y, z});
        *ox = w.x; *oy = w.y; *oz = w.z;
        return;
    }
    V3 w = Normalize(XfVector3(s->render_from_object.m, Normalize(XfVector3(s->render_from_object.mInv, V3{x, y, z}))));
    *ox = w.x; *oy = w.y; *oz = w.z;
}
// Transform::operator()(Point3fi) (util/transform.h:152-170) for an interval point
WF_HD P3i XfP3i(const float m[4][4], const P3i &in) {
    if (in.exact()) {
        Ivl3 r = XfPointExactI(m, in.mid());
        return P3i{V3{r.x.lo, r.y.lo, r.z.lo}, V3{r.x.hi, r.y.hi, r.z.hi}};
    }
    const V3 p = in.mid(), e = in.err();
    float x = p.x, y = p.y, z = p.z;
    float xp = (m[0][0] * x + m[0][1] * y) + (m[0][2] * z + m[0][3]);
    float yp = (m[1][0] * x + m[1][1] * y) + (m[1][2] * z + m[1][3]);
    float zp = (m[2][0] * x + m[2][1] * y) + (m[2][2] * z + m[2][3]);
    V3 pe;
    pe.x = (gamma(3) + 1) * (abs(m[0][0]) * e.x + abs(m[0][1]) * e.y + abs(m[0][2]) * e.z) +
           gamma(3) * (abs(m[0][0] * x) + abs(m[0][1] * y) + abs(m[0][2] * z) + abs(m[0][3]));
    pe.y = (gamma(3) + 1) * (abs(m[1][0]) * e.x + abs(m[1][1]) * e.y + abs(m[1][2]) * e.z) +
           gamma(3) * (abs(m[1][0] * x) + abs(m[1][1] * y) + abs(m[1][2] * z) + abs(m[1][3]));
    pe.z = (gamma(3) + 1) * (abs(m[2][0]) * e.x + abs(m[2][1]) * e.y + abs(m[2][2]) * e.z) +
           gamma(3) * (abs(m[2][0] * x) + abs(m[2][1] * y) + abs(m[2][2] * z) + abs(m[2][3]));
    P3i r = MakeP3i(V3{xp, yp, zp}, pe);
    const float wp = XfW(m, p);
    if (wp != 1) {
        Ivl rx = DivF(Ivl(r.lo.x, r.hi.x), wp), ry = DivF(Ivl(r.lo.y, r.hi.y), wp), rz = DivF(Ivl(r.lo.z, r.hi.z), wp);
        r = P3i{V3{rx.lo, ry.lo, rz.lo}, V3{rx.hi, ry.hi, rz.hi}};
    }
    return r;
}
// Transform::operator()(SurfaceInteraction) (util/transform.cpp:229-261): what TransformedPrimitive::Intersect applies to
// the interaction found in the instance's space.  Out of line: pointer arguments only (see WF_NI).
WF_HD void InstanceInteraction(const wf_instance *in, SurfIntr *si) {
    const float(*m)[4] = in->render_from_instance.m;
    const float(*mi)[4] = in->render_from_instance.mInv;
    SurfIntr r;
    r.pi = XfP3i(m, si->pi);
    r.n = Normalize(XfNormal3(mi, si->n));
    r.uv = si->uv;
    r.dpdu = XfVector3(m, si->dpdu);
    r.dpdv = XfVector3(m, si->dpdv);
    r.dndu = XfNormal3(mi, si->dndu);
    r.dndv = XfNormal3(mi, si->dndv);
    r.ns = Normalize(XfNormal3(mi, si->ns));
    r.dpdus = XfVector3(m, si->dpdus);
    r.dpdvs = XfVector3(m, si->dpdvs);
    r.dndus = XfNormal3(mi, si->dndus);
    r.dndvs = XfNormal3(mi, si->dndvs);
    r.ns = FaceForward(r.ns, r.n);
    r.mesh = si->mesh;
    *si = r;
}
WF_NI void InstanceInteractionP(const wf_instance *in, SurfIntr *si) { InstanceInteraction(in, si); }
WF_NI void InstanceWoP(const wf_instance *in, float x, float y, float z, float *ox, float *oy, float *oz) {
    // the instance-space interaction's wo = Normalize(-ApplyInverse(ray.d)) (interaction.h:40-43), transformed back and
    // normalised again (util/transform.cpp:235)
    V3 w = Normalize(XfVector3(in->render_from_instance.m, Normalize(XfVector3(in->render_from_instance.mInv, V3{x, y, z}))));
    *ox = w.x; *oy = w.y; *oz = w.z;
}
// GENERAL = false (the default in a lean unit, WF_DEV_LEAN): the scene has triangles only — the quadric / patch / curve callees are not
// reachable from the caller (see wf_scene.h "LEAN DEVICE VARIANTS")
template <bool GENERAL = !WF_DEV_LEAN, bool ANIM = false>
WF_HD V3 IntrWo(const SceneView &sv, int prim, int inst, V3 minusD, float time = 0) {
    if (inst >= 0) {
        V3 w;
        wf_instance inTmp;
        const wf_instance *in = &InstanceAt<ANIM>(sv, sv.instances[inst], time, &inTmp);
        if (GENERAL && prim >= sv.nTriangles) {
            // a quadric inside an instance: built in object space from the instance-space ray (normalised there and after the transform
            // back to instance space), then taken to render space by the instance transform (normalised again)
            V3 vI = XfVector3(in->render_from_instance.mInv, minusD);
            SphereWoP(sv.quadrics + (prim - sv.nTriangles), vI.x, vI.y, vI.z, &w.x, &w.y, &w.z);
            return Normalize(XfVector3(in->render_from_instance.m, w));
        }
        InstanceWoP(in, minusD.x, minusD.y, minusD.z, &w.x, &w.y, &w.z);
        return w;
    }
    if (!GENERAL || prim < sv.nTriangles) return Normalize(minusD);
    V3 w;
    SphereWoP(sv.quadrics + (prim - sv.nTriangles), minusD.x, minusD.y, minusD.z, &w.x, &w.y, &w.z);
    return w;
}
// the SurfaceInteraction of a hit record (prim, three floats): barycentrics for a triangle, pObj for a sphere
// inst >= 0: the primitive was reached through that object instance — the interaction is built in the definition's
// space and transformed (TransformedPrimitive::Intersect)
// the interaction of a curve hit (u, v, distance) found by the ray (ro, rd) in the space the curve lives in — with an alpha texture, through
// the replay of GeometricPrimitive::Intersect's alpha recursion (CurveAlphaIntersectP)
WF_NI void CurveHitInteractionP(const SceneView *svp, int prim, float b0, float b1, float b2, float rox, float roy, float roz, float rdx, float rdy, float rdz, SurfIntr *out) {
    const SceneView &sv = *svp;
    const wf_quadric *s = sv.quadrics + (prim - sv.nTriangles);
    if (sv.haveQuadricAlpha && sv.meshes[s->mesh].alpha_tex >= 0) {
        // GeometricPrimitive::Intersect's alpha recursion (cpu/primitive.cpp:56-72) found this hit with a ray respawned behind the
        // rejected ones: the same deterministic recursion again (tMax = infinity: every hit up to the accepted one lies inside the
        // bound the walk had) gives that ray's origin and the distance along it
        float t, u, v, lo[3], tl;
        if (CurveAlphaIntersectP(svp, prim, rox, roy, roz, rdx, rdy, rdz, WF_INFINITY, &t, &u, &v, lo, &tl)) {
            CurveInteractionP(s, sv.meshes[s->mesh].flags, u, v, tl, lo[0], lo[1], lo[2], rdx, rdy, rdz, out);
            return;
        }
    }
    CurveInteractionP(s, sv.meshes[s->mesh].flags, b0, b1, b2, rox, roy, roz, rdx, rdy, rdz, out);
}
WF_HD bool IsCurvePrim(const SceneView &sv, int prim) { return sv.haveCurves && prim >= sv.nTriangles && sv.quadrics[prim - sv.nTriangles].type == WF_QUADRIC_CURVE; }
// ro, rd: the render-space ray that found the hit — only a curve's interaction depends on it (its frame is ray-aligned)
// CURVE_ALPHA: the caller may meet a curve with an alpha texture, whose interaction needs the replay of the alpha recursion
// (CurveHitInteractionP: a deep out-of-line chain — curve intersector, texture evaluator).  Only the material stage's general shade kernels
// ask for it: the scene builder admits alpha textures on curves with ordinary materials only (no interface / mix / subsurface material,
// no emission), so no other consumer of a hit can meet one — and none of them pays for the chain's registers.
#ifndef WF_LEAN_INLINE_INSTANCE
#define WF_LEAN_INLINE_INSTANCE 1
#endif
template <bool GENERAL = !WF_DEV_LEAN, bool CURVE_ALPHA = false, bool ANIM = false>
WF_HD void HitInteraction(const SceneView &sv, int prim, int inst, float b0, float b1, float b2, SurfIntr *si, V3 ro, V3 rd, float time = 0) {
    wf_instance inTmp;
    const wf_instance *inp = inst >= 0 ? &InstanceAt<ANIM>(sv, sv.instances[inst], time, &inTmp) : nullptr;
    if constexpr (!GENERAL) TriangleInteraction(sv, prim, b0, b1, b2, si);
    else
    if (IsCurvePrim(sv, prim)) {
        if (inst >= 0) { float tm = WF_INFINITY; InstanceRay(*inp, ro, rd, &tm, &ro, &rd); }
        SurfIntr tmp;
        if constexpr (CURVE_ALPHA) CurveHitInteractionP(sv.self, prim, b0, b1, b2, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, &tmp);
        else { const wf_quadric *s_ = sv.quadrics + (prim - sv.nTriangles); CurveInteractionP(s_, sv.meshes[s_->mesh].flags, b0, b1, b2, ro.x, ro.y, ro.z, rd.x, rd.y, rd.z, &tmp); }
        *si = tmp;
    } else if (prim >= sv.nTriangles) SphereInteraction(sv, prim, V3{b0, b1, b2}, si);
    else TriangleInteraction(sv, prim, b0, b1, b2, si);
    if (inst >= 0) {
        // (a lean kernel — triangles only, 4 waves — applies the transform in line: the out-of-line call passes the 45-float interaction
        //  through scratch both ways, ~360 B of the 640 B per item k_mat_shade<diffuse> wrote; round 5)
        if constexpr (!GENERAL && WF_LEAN_INLINE_INSTANCE) InstanceInteraction(inp, si);
        else {
        SurfIntr tmp = *si;
        InstanceInteractionP(inp, &tmp);
        *si = tmp;
        }
    }
}

// GeometricPrimitive::Intersect (cpu/primitive.cpp:50-78) for a sphere / disk / cylinder / bilinear patch with an alpha texture: a hit
// that fails the (stochastic) alpha test is skipped by re-intersecting the same shape with the ray respawned behind it; the
// parametric distances add up innermost first, as the recursion returns.  (A triangle cannot be hit twice: AlphaTestPasses.)
WF_NI bool QuadricAlphaIntersectP(const SceneView *svp, int prim, float ox, float oy, float oz, float dx, float dy, float dz, float tMax,
                                  float *tHit, float *px, float *py, float *pz) {
    const SceneView &sv = *svp;
    const wf_quadric &s = sv.quadrics[prim - sv.nTriangles];
    if (s.type == WF_QUADRIC_CURVE) {
        float lo[3], tl;
        if (!CurveAlphaIntersectP(svp, prim, ox, oy, oz, dx, dy, dz, tMax, tHit, px, py, lo, &tl)) return false;
        *pz = *tHit;
        return true;
    }
    const int alphaTex = sv.meshes[s.mesh].alpha_tex;
    V3 o{ox, oy, oz};
    const V3 d{dx, dy, dz};
    // (the reference recurses without a bound, GeometricPrimitive::Intersect; here the chain of rejected hits is capped at MAXN - 1 and the
    //  next hit accepted.  Unreachable for a quadric — a ray meets one at most twice, and every respawned ray starts beyond the rejected
    //  hit — and for a curve only by a ray that crosses the ribbon's alpha-rejected parts fifteen times: ADVICE r5, documented, not a parity case)
    constexpr int MAXN = 16;
    float ts[MAXN];
    int n = 0;
    QuadricHit h;
    while (true) {
        if (!QuadricBasicIntersect<false>(s, o, d, tMax, &h)) return false;
        SurfIntr si;
        SphereInteraction(sv, prim, h.pObj, &si);
        TexCtx tc;
        tc.p = si.pi.mid(); tc.n = si.n; tc.uv = si.uv;
        const float a = EvalFloatTexture(sv, alphaTex, tc);
        bool accept = true;
        if (a < 1) {
            const float u = (a <= 0) ? 1.f : HashToFloat(Hash6f(o, d));
            if (u > a) accept = false;
        }
        if (accept || n == MAXN - 1) break;
        ts[n++] = h.tHit;
        o = OffsetRayOrigin(si.pi, si.n, d);   // si->intr.SpawnRay(r.d)
        tMax = tMax - h.tHit;
    }
    float t = h.tHit;
    for (int i = n - 1; i >= 0; --i) t += ts[i];   // siNext->tHit += si->tHit, innermost first
    *tHit = t; *px = h.pObj.x; *py = h.pObj.y; *pz = h.pObj.z;
    return true;
}

WF_NI bool CurveAlphaIntersectP(const SceneView *svp, int prim, float ox, float oy, float oz, float dx, float dy, float dz, float tMax,
                                float *tTotal, float *uOut, float *vOut, float *lastOrigin, float *tLocal) {
    const SceneView &sv = *svp;
    const wf_quadric &s = sv.quadrics[prim - sv.nTriangles];
    const int alphaTex = sv.meshes[s.mesh].alpha_tex;
    V3 o{ox, oy, oz};
    const V3 d{dx, dy, dz};
    constexpr int MAXN = 16;
    float ts[MAXN];
    int n = 0;
    QuadricHit h;
    while (true) {
        if (!CurveBasicIntersect(s, o, d, tMax, &h)) return false;
        SurfIntr si;
        CurveInteractionP(&s, sv.meshes[s.mesh].flags, h.pObj.x, h.pObj.y, h.pObj.z, o.x, o.y, o.z, d.x, d.y, d.z, &si);
        TexCtx tc;
        tc.p = si.pi.mid(); tc.n = si.n; tc.uv = si.uv;
        const float a = EvalFloatTexture(sv, alphaTex, tc);
        bool accept = true;
        if (a < 1) {
            const float u = (a <= 0) ? 1.f : HashToFloat(Hash6f(o, d));
            if (u > a) accept = false;
        }
        if (accept || n == MAXN - 1) break;
        ts[n++] = h.tHit;
        o = OffsetRayOrigin(si.pi, si.n, d);   // si->intr.SpawnRay(r.d)
        tMax = tMax - h.tHit;
    }
    float t = h.tHit;
    for (int i = n - 1; i >= 0; --i) t += ts[i];   // siNext->tHit += si->tHit, innermost first
    *tTotal = t; *uOut = h.pObj.x; *vOut = h.pObj.y; *tLocal = h.tHit;
    lastOrigin[0] = o.x; lastOrigin[1] = o.y; lastOrigin[2] = o.z;
    return true;
}

// Sphere::Sample(Point2f u), shapes.cpp:38-58
WF_HD ShapeSampleR SphereSampleArea(const wf_quadric &s, int meshFlags, V2 u) {
    ShapeSampleR r;
    if (s.type == WF_QUADRIC_DISK) {
        // Disk::Sample(Point2f), shapes.h:490-505
        V2 pd = SampleUniformDiskConcentric(u);
        V3 pObj{pd.x * s.radius, pd.y * s.radius, s.z_min};
        r.pi = XfPointI(s.render_from_object.m, pObj, V3{0, 0, 0});
        N3 n = Normalize(XfNormal3(s.render_from_object.mInv, N3{0, 0, 1}));
        if (meshFlags & WF_MESH_REVERSE_ORIENTATION) n = n * -1.f;
        float phi = atan2(pd.y, pd.x);
        if (phi < 0) phi += 2 * Pi;
        float radiusSample = sqrt(Sqr(pObj.x) + Sqr(pObj.y));
        r.uv = V2{phi / s.phi_max, (s.radius - radiusSample) / (s.radius - s.inner_radius)};
        r.n = n;
        r.pdf = 1 / QuadricArea(s);
        r.valid = true;
        return r;
    }
    if (s.type == WF_QUADRIC_CYLINDER) {
        // Cylinder::Sample(Point2f), shapes.h:701-717
        float z = Lerp(u.x, s.z_min, s.z_max);
        float phi = u.y * s.phi_max;
        V3 pObj{s.radius * cos(phi), s.radius * sin(phi), z};
        float hitRad = sqrt(Sqr(pObj.x) + Sqr(pObj.y));
        pObj.x *= s.radius / hitRad;
        pObj.y *= s.radius / hitRad;
        V3 pObjError = gamma(3) * Abs(V3{pObj.x, pObj.y, 0});
        r.pi = XfPointI(s.render_from_object.m, pObj, pObjError);
        N3 n = Normalize(XfNormal3(s.render_from_object.mInv, N3{pObj.x, pObj.y, 0}));
        if (meshFlags & WF_MESH_REVERSE_ORIENTATION) n = n * -1.f;
        r.uv = V2{phi / s.phi_max, (pObj.z - s.z_min) / (s.z_max - s.z_min)};
        r.n = n;
        r.pdf = 1 / QuadricArea(s);
        r.valid = true;
        return r;
    }
    V3 pObj = s.radius * SampleUniformSphere(u);
    pObj = pObj * (s.radius / Length(pObj));
    V3 pObjError = gamma(5) * Abs(pObj);
    N3 n = Normalize(XfNormal3(s.render_from_object.mInv, N3{pObj.x, pObj.y, pObj.z}));
    if (meshFlags & WF_MESH_REVERSE_ORIENTATION) n = n * -1.f;
    float theta = SafeACos(pObj.z / s.radius);
    float phi = atan2(pObj.y, pObj.x);
    if (phi < 0) phi += 2 * Pi;
    r.uv = V2{phi / s.phi_max, (theta - s.theta_z_min) / (s.theta_z_max - s.theta_z_min)};
    r.pi = XfPointI(s.render_from_object.m, pObj, pObjError);
    r.n = n;
    r.pdf = 1 / QuadricArea(s);
    r.valid = true;
    return r;
}
// Sphere::Sample(const ShapeSampleContext &, Point2f), shapes.h:300-372
WF_NI void SphereSampleP(const wf_quadric *sp, int meshFlags, const P3i *ctxPiP, float nx, float ny, float nz, float ux, float uy, ShapeSampleR *out) {
    const wf_quadric s = *sp;
    const P3i ctxPi = *ctxPiP;
    const N3 ctxN{nx, ny, nz};
    const V2 u{ux, uy};
    const float radius = s.radius;
    ShapeSampleR &r = *out;
    r.valid = false;
    r.pdf = 0;
    V3 pCenter = XfPoint3(s.render_from_object.m, V3{0, 0, 0});
    V3 rp = ctxPi.mid();
    V3 pOrigin = OffsetRayOrigin(ctxPi, ctxN, pCenter - rp);
    // Disk / Cylinder::Sample(ctx, u) (shapes.h:511-526, 723-738) are the area-sampling branch taken unconditionally
    if (s.type != WF_QUADRIC_SPHERE || DistanceSquared(pOrigin, pCenter) <= Sqr(radius)) {
        r = SphereSampleArea(s, meshFlags, u);
        r.valid = false;
        V3 wi = r.pi.mid() - rp;
        if (LengthSquared(wi) == 0) return;
        wi = Normalize(wi);
        r.pdf /= AbsDot(r.n, -wi) / DistanceSquared(rp, r.pi.mid());
        if (IsInf(r.pdf)) return;
        r.valid = true;
        return;
    }
    float sinThetaMax = radius / Distance(rp, pCenter);
    float sin2ThetaMax = Sqr(sinThetaMax);
    float cosThetaMax = SafeSqrt(1 - sin2ThetaMax);
    float oneMinusCosThetaMax = 1 - cosThetaMax;
    float cosTheta = (cosThetaMax - 1) * u.x + 1;
    float sin2Theta = 1 - Sqr(cosTheta);
    if (sin2ThetaMax < 0.00068523f /* sin^2(1.5 deg) */) {
        sin2Theta = sin2ThetaMax * u.x;
        cosTheta = sqrt(1 - sin2Theta);
        oneMinusCosThetaMax = sin2ThetaMax / 2;
    }
    float cosAlpha = sin2Theta / sinThetaMax + cosTheta * SafeSqrt(1 - sin2Theta / Sqr(sinThetaMax));
    float sinAlpha = SafeSqrt(1 - Sqr(cosAlpha));
    float phi = u.y * 2 * Pi;
    V3 w = SphericalDirection(sinAlpha, cosAlpha, phi);
    Frame samplingFrame = Frame::FromZ(Normalize(pCenter - rp));
    N3 n = toN(samplingFrame.FromLocal(-w));
    V3 p = pCenter + radius * V3{n.x, n.y, n.z};
    if (meshFlags & WF_MESH_REVERSE_ORIENTATION) n = n * -1.f;
    V3 pError = gamma(5) * Abs(p);
    V3 pObj = XfPoint3(s.render_from_object.mInv, p);
    float theta = SafeACos(pObj.z / radius);
    float spherePhi = atan2(pObj.y, pObj.x);
    if (spherePhi < 0) spherePhi += 2 * Pi;
    r.uv = V2{spherePhi / s.phi_max, (theta - s.theta_z_min) / (s.theta_z_max - s.theta_z_min)};
    r.pi = MakeP3i(p, pError);
    r.n = n;
    r.pdf = 1 / (2 * Pi * oneMinusCosThetaMax);
    r.valid = true;
}

// ---------------------------------------------------------------------------------------------
// BilinearPatch as an emitter: Sample / PDF (shapes.cpp:1155-1368), SampleSphericalRectangle / InvertSphericalRectangleSample
// (util/sampling.cpp:163-330), SphericalQuadArea (util/vecmath.h:1646-1664), InvertBilinear (util/vecmath.h:623-660).
// The patch's area (radius field) and its IsRectangle() verdict (pad[0] bit 2) are computed at load (scene_build.cpp).
WF_HD float SphericalQuadArea(V3 a, V3 b, V3 c, V3 d) {
    V3 axb = Cross(a, b), bxc = Cross(b, c);
    V3 cxd = Cross(c, d), dxa = Cross(d, a);
    if (LengthSquared(axb) == 0 || LengthSquared(bxc) == 0 || LengthSquared(cxd) == 0 || LengthSquared(dxa) == 0) return 0;
    axb = Normalize(axb); bxc = Normalize(bxc); cxd = Normalize(cxd); dxa = Normalize(dxa);
    float alpha = AngleBetween(dxa, -axb);
    float beta = AngleBetween(axb, -bxc);
    float gam = AngleBetween(bxc, -cxd);
    float delta = AngleBetween(cxd, -dxa);
    return abs(alpha + beta + gam + delta - 2 * Pi);
}
struct SphRect { V3 rx, ry, rz; float x0, y0, z0, x1, y1, g0, g1, g2, g3, b0, b1; };
WF_HD SphRect SphRectInit(V3 pRef, V3 s, V3 ex, V3 ey, float exl, float eyl) {
    SphRect q;
    q.rx = ex / exl; q.ry = ey / eyl; q.rz = Cross(q.rx, q.ry);  // Frame::FromXY
    V3 dd = s - pRef;
    V3 dLocal{Dot(dd, q.rx), Dot(dd, q.ry), Dot(dd, q.rz)};
    q.z0 = dLocal.z;
    if (q.z0 > 0) { q.rz = -q.rz; q.z0 *= -1; }
    q.x0 = dLocal.x; q.y0 = dLocal.y;
    q.x1 = q.x0 + exl; q.y1 = q.y0 + eyl;
    V3 v00{q.x0, q.y0, q.z0}, v01{q.x0, q.y1, q.z0}, v10{q.x1, q.y0, q.z0}, v11{q.x1, q.y1, q.z0};
    V3 n0 = Normalize(Cross(v00, v10)), n1 = Normalize(Cross(v10, v11));
    V3 n2 = Normalize(Cross(v11, v01)), n3 = Normalize(Cross(v01, v00));
    q.g0 = AngleBetween(-n0, n1); q.g1 = AngleBetween(-n1, n2);
    q.g2 = AngleBetween(-n2, n3); q.g3 = AngleBetween(-n3, n0);
    q.b0 = n0.z; q.b1 = n2.z;
    return q;
}
WF_HD V3 SampleSphericalRectangle(V3 pRef, V3 s, V3 ex, V3 ey, V2 u, float *pdf) {
    float exl = Length(ex), eyl = Length(ey);
    SphRect q = SphRectInit(pRef, s, ex, ey, exl, eyl);
    float solidAngle = q.g0 + q.g1 + q.g2 + q.g3 - 2 * Pi;
    if (solidAngle <= 0) { *pdf = 0; return s + u.x * ex + u.y * ey; }
    *pdf = fmax(0.f, 1 / solidAngle);
    if ((double)solidAngle < 1e-3) return s + u.x * ex + u.y * ey;
    float au = u.x * (q.g0 + q.g1 - 2 * Pi) + (u.x - 1) * (q.g2 + q.g3);
    float fu = (cos(au) * q.b0 - q.b1) / sin(au);
    float cu = copysign(1 / sqrt(Sqr(fu) + Sqr(q.b0)), fu);
    cu = Clamp(cu, -OneMinusEpsilon, OneMinusEpsilon);
    float xu = -(cu * q.z0) / SafeSqrt(1 - Sqr(cu));
    xu = Clamp(xu, q.x0, q.x1);
    float dd = sqrt(Sqr(xu) + Sqr(q.z0));
    float h0 = q.y0 / sqrt(Sqr(dd) + Sqr(q.y0));
    float h1 = q.y1 / sqrt(Sqr(dd) + Sqr(q.y1));
    float hv = h0 + u.y * (h1 - h0), hvsq = Sqr(hv);
    float yv = (hvsq < 1 - 1e-6f) ? (hv * dd) / sqrt(1 - hvsq) : q.y1;
    return pRef + (xu * q.rx + yv * q.ry + q.z0 * q.rz);
}
WF_HD V2 InvertSphericalRectangleSample(V3 pRef, V3 s, V3 ex, V3 ey, V3 pRect) {
    float exl = Length(ex), eyl = Length(ey);
    SphRect q = SphRectInit(pRef, s, ex, ey, exl, eyl);
    const float z0 = q.z0, x0 = q.x0, y0 = q.y0, x1 = q.x1, y1 = q.y1;
    float z0sq = Sqr(z0), y0sq = Sqr(y0), y1sq = Sqr(y1);
    float b0 = q.b0, b1 = q.b1, b0sq = Sqr(b0);
    float solidAngle = (float)((double)q.g0 + (double)q.g1 + (double)q.g2 + (double)q.g3 - 2. * (double)Pi);
    if ((double)solidAngle < 1e-3) {
        V3 pq = pRect - s;
        return V2{Dot(pq, ex) / LengthSquared(ex), Dot(pq, ey) / LengthSquared(ey)};
    }
    V3 vv = pRect - pRef;
    V3 v{Dot(vv, q.rx), Dot(vv, q.ry), Dot(vv, q.rz)};
    float xu = v.x, yv = v.y;
    xu = Clamp(xu, x0, x1);
    if (xu == 0) xu = 1e-10f;
    float invcusq = 1 + z0sq / Sqr(xu);
    float fusq = invcusq - b0sq;
    float fu = copysign(sqrt(fusq), xu);
    float sq = SafeSqrt(DifferenceOfProducts(b0, b0, b1, b1) + fusq);
    float au = atan2(-(b1 * fu) - copysign(b0 * sq, fu * b0), b0 * b1 - sq * abs(fu));
    if (au > 0) au -= 2 * Pi;
    if (fu == 0) au = Pi;
    float u0 = (au + q.g2 + q.g3) / solidAngle;
    float ddsq = Sqr(xu) + z0sq;
    float h0 = y0 / sqrt(ddsq + y0sq);
    float h1 = y1 / sqrt(ddsq + y1sq);
    float yvsq = Sqr(yv);
    float u1[2] = {(DifferenceOfProducts(h0, h0, h0, h1) - abs(h0 - h1) * sqrt(yvsq * (ddsq + yvsq)) / (ddsq + yvsq)) / Sqr(h0 - h1),
                   (DifferenceOfProducts(h0, h0, h0, h1) + abs(h0 - h1) * sqrt(yvsq * (ddsq + yvsq)) / (ddsq + yvsq)) / Sqr(h0 - h1)};
    // TODO: yuck is there a better way to figure out which is the right solution?
    float hv[2] = {Lerp(u1[0], h0, h1), Lerp(u1[1], h0, h1)};
    float hvsq[2] = {Sqr(hv[0]), Sqr(hv[1])};
    float yz[2] = {(hv[0] * sqrt(ddsq)) / sqrt(1 - hvsq[0]), (hv[1] * sqrt(ddsq)) / sqrt(1 - hvsq[1])};
    V2 u = (abs(yz[0] - yv) < abs(yz[1] - yv)) ? V2{Clamp(u0, 0.f, 1.f), u1[0]} : V2{Clamp(u0, 0.f, 1.f), u1[1]};
    return u;
}
WF_HD V2 InvertBilinear(V2 p, V2 v0, V2 v1, V2 v2, V2 v3) {
    V2 a = v0, b = v1, c = v3, d = v2;
    V2 e{b.x - a.x, b.y - a.y}, f{d.x - a.x, d.y - a.y}, g{(a.x - b.x) + (c.x - d.x), (a.y - b.y) + (c.y - d.y)}, h{p.x - a.x, p.y - a.y};
    auto cross2d = [](V2 a, V2 b) { return DifferenceOfProducts(a.x, b.y, a.y, b.x); };
    float k2 = cross2d(g, f);
    float k1 = cross2d(e, f) + cross2d(h, g);
    float k0 = cross2d(h, e);
    if (abs(k2) < 0.001f) {
        if (abs(e.x * k1 - g.x * k0) < 1e-5f) return V2{(h.y * k1 + f.y * k0) / (e.y * k1 - g.y * k0), -k0 / k1};
        else return V2{(h.x * k1 + f.x * k0) / (e.x * k1 - g.x * k0), -k0 / k1};
    }
    float v0q, v1q;
    if (!QuadraticF(k2, k1, k0, &v0q, &v1q)) return V2{0, 0};
    float u = (h.x - f.x * v0q) / (e.x + g.x * v0q);
    if (u < 0 || u > 1 || v0q < 0 || v0q > 1) return V2{(h.x - f.x * v1q) / (e.x + g.x * v1q), v1q};
    return V2{u, v0q};
}
// BilinearPatch::IsRectangle (shapes.h:1512-1533) and the area of the constructor (shapes.cpp:1036-1067); evaluated at load
WF_HD bool BlpIsRectangle(const BlpData &d) {
    const V3 p00 = d.p00, p10 = d.p10, p01 = d.p01, p11 = d.p11;
    auto eq = [](V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; };
    if (eq(p00, p01) || eq(p01, p11) || eq(p11, p10) || eq(p10, p00)) return false;
    N3 n = toN(Normalize(Cross(p10 - p00, p01 - p00)));
    if (AbsDot(toN(Normalize(p11 - p00)), n) > 1e-5f) return false;
    V3 pCenter = (p00 + p01 + p10 + p11) / 4;
    float d2[4] = {DistanceSquared(p00, pCenter), DistanceSquared(p01, pCenter), DistanceSquared(p10, pCenter), DistanceSquared(p11, pCenter)};
    for (int i = 1; i < 4; ++i)
        if (abs(d2[i] - d2[0]) / d2[0] > 1e-4f) return false;
    return true;
}
WF_HD float BlpArea(const BlpData &d, bool rectangle) {
    const V3 p00 = d.p00, p10 = d.p10, p01 = d.p01, p11 = d.p11;
    if (rectangle) return Distance(p00, p01) * Distance(p00, p10);
    constexpr int na = 3;
    V3 p[na + 1][na + 1];
    for (int i = 0; i <= na; ++i) {
        float u = float(i) / float(na);
        for (int j = 0; j <= na; ++j) {
            float v = float(j) / float(na);
            p[i][j] = LerpV(u, LerpV(v, p00, p01), LerpV(v, p10, p11));
        }
    }
    float area = 0;
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < na; ++j) area += 0.5f * Length(Cross(p[i + 1][j + 1] - p[i][j], p[i + 1][j] - p[i][j + 1]));
    return area;
}
WF_HD N3 BlpSampleNormal(const BlpData &d, int meshFlags, V3 dpdu, V3 dpdv, float u, float v) {
    N3 n = toN(Normalize(Cross(dpdu, dpdv)));
    if (d.hasN) {
        auto lerpN = [](float t, N3 a, N3 b) { return (1 - t) * a + t * b; };
        N3 ns = lerpN(u, lerpN(v, d.n00, d.n01), lerpN(v, d.n10, d.n11));
        n = FaceForward(n, ns);
    } else if (meshFlags & WF_MESH_FLIP_NORMAL) n = -n;
    return n;
}
WF_HD V2 BlpST(const BlpData &d, float u, float v) {
    if (!d.hasUV) return V2{u, v};
    auto lerp2 = [](float t, V2 a, V2 b) { return V2{(1 - t) * a.x + t * b.x, (1 - t) * a.y + t * b.y}; };
    return lerp2(u, lerp2(v, d.uv00, d.uv01), lerp2(v, d.uv10, d.uv11));
}
// BilinearPatch::Sample(Point2f u) (shapes.cpp:1155-1215)
// `dist` / D: the mesh's image distribution ("emissionfilename", shapes.cpp:1165-1166) and the table it lives in, or null
WF_HD ShapeSampleR BlpSampleArea(const BlpData &d, int meshFlags, bool rectangle, V2 u, const wf_pc2d *dist = nullptr, const float *D = nullptr) {
    ShapeSampleR r;
    r.valid = false;
    const V3 p00 = d.p00, p10 = d.p10, p01 = d.p01, p11 = d.p11;
    float pdf = 1;
    V2 uv;
    if (dist) uv = PC2DSample(D, *dist, u, &pdf);
    else if (!rectangle) {
        const float w[4] = {Length(Cross(p10 - p00, p01 - p00)), Length(Cross(p10 - p00, p11 - p10)), Length(Cross(p01 - p00, p11 - p01)), Length(Cross(p11 - p10, p11 - p01))};
        uv = SampleBilinear(u, w);
        pdf = BilinearPDF(uv, w);
    } else uv = u;
    V3 pu0 = LerpV(uv.y, p00, p01), pu1 = LerpV(uv.y, p10, p11);
    V3 p = LerpV(uv.x, pu0, pu1);
    V3 dpdu = pu1 - pu0;
    V3 dpdv = LerpV(uv.x, p01, p11) - LerpV(uv.x, p00, p10);
    if (LengthSquared(dpdu) == 0 || LengthSquared(dpdv) == 0) return r;
    V3 pAbsSum = Abs(p00) + Abs(p01) + Abs(p10) + Abs(p11);
    r.pi = MakeP3i(p, gamma(6) * pAbsSum);
    r.n = BlpSampleNormal(d, meshFlags, dpdu, dpdv, uv.x, uv.y);
    r.uv = BlpST(d, uv.x, uv.y);
    r.pdf = pdf / Length(Cross(dpdu, dpdv));
    r.valid = true;
    return r;
}
// BilinearPatch::Sample(const ShapeSampleContext &, Point2f) (shapes.cpp:1252-1327).  Out of line: pointer arguments only.
WF_NI void BilinearSampleP(const wf_quadric *sp, int meshFlags, const P3i *ctxPiP, float nsx, float nsy, float nsz, float ux, float uy, ShapeSampleR *out, const float *D) {
    const BlpData d = LoadBlp(*sp);
    const bool rectangle = ((int)sp->pad[0] & 4) != 0;
    const wf_pc2d *dist = ((int)sp->pad[0] & 8) ? (const wf_pc2d *)sp->ext : nullptr;   // "emissionfilename": always sampled by area (shapes.cpp:1265)
    const V3 p00 = d.p00, p10 = d.p10, p01 = d.p01, p11 = d.p11;
    const V3 ctxP = ctxPiP->mid();
    const N3 ctxNs{nsx, nsy, nsz};
    V2 u{ux, uy};
    V3 v00 = Normalize(p00 - ctxP), v10 = Normalize(p10 - ctxP);
    V3 v01 = Normalize(p01 - ctxP), v11 = Normalize(p11 - ctxP);
    if (!rectangle || dist || SphericalQuadArea(v00, v10, v11, v01) <= 1e-4f) {
        ShapeSampleR ss = BlpSampleArea(d, meshFlags, rectangle, u, dist, D);
        ss.valid = ss.valid;
        if (ss.valid) {
            V3 wi = ss.pi.mid() - ctxP;
            if (LengthSquared(wi) == 0) ss.valid = false;
            else {
                wi = Normalize(wi);
                ss.pdf /= AbsDot(ss.n, -wi) / DistanceSquared(ctxP, ss.pi.mid());
                if (IsInf(ss.pdf)) ss.valid = false;
            }
        }
        *out = ss;
        return;
    }
    float pdf = 1;
    if (ctxNs.x != 0 || ctxNs.y != 0 || ctxNs.z != 0) {
        const float w[4] = {fmax(0.01f, AbsDot(toN(v00), ctxNs)), fmax(0.01f, AbsDot(toN(v10), ctxNs)), fmax(0.01f, AbsDot(toN(v01), ctxNs)), fmax(0.01f, AbsDot(toN(v11), ctxNs))};
        u = SampleBilinear(u, w);
        pdf *= BilinearPDF(u, w);
    }
    V3 eu = p10 - p00, ev = p01 - p00;
    float quadPDF;
    V3 p = SampleSphericalRectangle(ctxP, p00, eu, ev, u, &quadPDF);
    pdf *= quadPDF;
    V2 uv{Dot(p - p00, eu) / DistanceSquared(p10, p00), Dot(p - p00, ev) / DistanceSquared(p01, p00)};
    ShapeSampleR r;
    r.pi = MakeP3i(p);
    r.n = BlpSampleNormal(d, meshFlags, eu, ev, uv.x, uv.y);
    r.uv = BlpST(d, uv.x, uv.y);
    r.pdf = pdf;
    r.valid = true;
    *out = r;
}
// BilinearPatch::PDF(const ShapeSampleContext &, Vector3f wi) (shapes.cpp:1329-1368) over PDF(const Interaction &) (:1217-1250)
WF_NI float BilinearPDFP(const wf_quadric *sp, int meshFlags, const P3i *ctxPiP, float nx, float ny, float nz, float nsx, float nsy, float nsz, float wx, float wy, float wz, const float *D) {
    const BlpData d = LoadBlp(*sp);
    const bool rectangle = ((int)sp->pad[0] & 4) != 0;
    const wf_pc2d *dist = ((int)sp->pad[0] & 8) ? (const wf_pc2d *)sp->ext : nullptr;
    const V3 p00 = d.p00, p10 = d.p10, p01 = d.p01, p11 = d.p11;
    const P3i ctxPi = *ctxPiP;
    const V3 ctxP = ctxPi.mid();
    const N3 ctxN{nx, ny, nz}, ctxNs{nsx, nsy, nsz};
    const V3 wi{wx, wy, wz};
    V3 o = OffsetRayOrigin(ctxPi, ctxN, wi);
    float hu, hv, ht;
    if (!IntersectBilinearPatch(o, wi, WF_INFINITY, p00, p10, p01, p11, &hu, &hv, &ht)) return 0;
    SurfIntr si;
    BilinearInteractionP(sp, meshFlags, hu, hv, &si);
    const V3 pHit = si.pi.mid();
    V3 v00 = Normalize(p00 - ctxP), v10 = Normalize(p10 - ctxP);
    V3 v01 = Normalize(p01 - ctxP), v11 = Normalize(p11 - ctxP);
    if (!rectangle || dist || SphericalQuadArea(v00, v10, v11, v01) <= 1e-4f) {
        // PDF(isect->intr): parametric (u, v) back from the interaction's (s, t)
        V2 uv = si.uv;
        if (d.hasUV) uv = InvertBilinear(uv, d.uv00, d.uv10, d.uv01, d.uv11);
        float pdfA;
        if (dist) pdfA = PC2DPDF(D, *dist, uv);   // shapes.cpp:1233-1234
        else if (!rectangle) {
            const float w[4] = {Length(Cross(p10 - p00, p01 - p00)), Length(Cross(p10 - p00, p11 - p10)), Length(Cross(p01 - p00, p11 - p01)), Length(Cross(p11 - p10, p11 - p01))};
            pdfA = BilinearPDF(uv, w);
        } else pdfA = 1;
        V3 pu0 = LerpV(uv.y, p00, p01), pu1 = LerpV(uv.y, p10, p11);
        V3 dpdu = pu1 - pu0;
        V3 dpdv = LerpV(uv.x, p01, p11) - LerpV(uv.x, p00, p10);
        pdfA = pdfA / Length(Cross(dpdu, dpdv));
        float pdf = pdfA * (DistanceSquared(ctxP, pHit) / AbsDot(si.n, -wi));
        return IsInf(pdf) ? 0.f : pdf;
    }
    float pdf = 1 / SphericalQuadArea(v00, v10, v11, v01);
    if (ctxNs.x != 0 || ctxNs.y != 0 || ctxNs.z != 0) {
        const float w[4] = {fmax(0.01f, AbsDot(toN(v00), ctxNs)), fmax(0.01f, AbsDot(toN(v10), ctxNs)), fmax(0.01f, AbsDot(toN(v01), ctxNs)), fmax(0.01f, AbsDot(toN(v11), ctxNs))};
        V2 u = InvertSphericalRectangleSample(ctxP, p00, p10 - p00, p01 - p00, pHit);
        return BilinearPDF(u, w) * pdf;
    }
    return pdf;
}

WF_HD ShapeSampleR SphereSample(const SceneView &sv, int prim, const P3i &ctxPi, N3 ctxN, N3 ctxNs, V2 u) {
    const wf_quadric *s = sv.quadrics + (prim - sv.nTriangles);
    const P3i pi = ctxPi;
    ShapeSampleR r;
    if (s->type == WF_QUADRIC_CURVE) {
        // Curve::Sample(const ShapeSampleContext &, Point2f): LOG_FATAL("Curve::Sample not implemented.") (shapes.cpp:748-752)
        RaiseFatal(sv, WF_FATAL_CURVE_SAMPLE);
        r.valid = false;
        return r;
    }
    if (s->type == WF_QUADRIC_BILINEAR) { BilinearSampleP(s, sv.meshes[s->mesh].flags, &pi, ctxNs.x, ctxNs.y, ctxNs.z, u.x, u.y, &r, sv.tableData); return r; }
    SphereSampleP(s, sv.meshes[s->mesh].flags, &pi, ctxN.x, ctxN.y, ctxN.z, u.x, u.y, &r);
    return r;
}
// Sphere::PDF(const ShapeSampleContext &, Vector3f wi), shapes.h:374-405
WF_NI float SpherePDFP(const wf_quadric *sp, int meshFlags, const P3i *ctxPiP, float nx, float ny, float nz, float wx, float wy, float wz) {
    const wf_quadric s = *sp;
    const P3i ctxPi = *ctxPiP;
    const N3 ctxN{nx, ny, nz};
    const V3 wi{wx, wy, wz};
    const float radius = s.radius;
    V3 pCenter = XfPoint3(s.render_from_object.m, V3{0, 0, 0});
    V3 rp = ctxPi.mid();
    V3 pOrigin = OffsetRayOrigin(ctxPi, ctxN, pCenter - rp);
    if (s.type != WF_QUADRIC_SPHERE || DistanceSquared(pOrigin, pCenter) <= Sqr(radius)) {
        V3 o = OffsetRayOrigin(ctxPi, ctxN, wi);
        QuadricHit qh;
        if (!QuadricBasicIntersect(s, o, wi, WF_INFINITY, &qh)) return 0;
        SurfIntr si;
        SphereInteractionP(sp, meshFlags, qh.pObj.x, qh.pObj.y, qh.pObj.z, &si);
        float area = QuadricArea(s);
        float pdf = (1 / area) / (AbsDot(si.n, -wi) / DistanceSquared(rp, si.pi.mid()));
        if (IsInf(pdf)) pdf = 0;
        return pdf;
    }
    float sin2ThetaMax = radius * radius / DistanceSquared(rp, pCenter);
    float cosThetaMax = SafeSqrt(1 - sin2ThetaMax);
    float oneMinusCosThetaMax = 1 - cosThetaMax;
    if (sin2ThetaMax < 0.00068523f /* sin^2(1.5 deg) */) oneMinusCosThetaMax = sin2ThetaMax / 2;
    return 1 / (2 * Pi * oneMinusCosThetaMax);
}
WF_HD float SpherePDF(const SceneView &sv, int prim, const P3i &ctxPi, N3 ctxN, N3 ctxNs, V3 wi) {
    const wf_quadric *s = sv.quadrics + (prim - sv.nTriangles);
    const P3i pi = ctxPi;
    if (s->type == WF_QUADRIC_CURVE) { RaiseFatal(sv, WF_FATAL_CURVE_PDF); return 0; }   // Curve::PDF: LOG_FATAL (shapes.cpp:754-757)
    if (s->type == WF_QUADRIC_BILINEAR) return BilinearPDFP(s, sv.meshes[s->mesh].flags, &pi, ctxN.x, ctxN.y, ctxN.z, ctxNs.x, ctxNs.y, ctxNs.z, wi.x, wi.y, wi.z, sv.tableData);
    return SpherePDFP(s, sv.meshes[s->mesh].flags, &pi, ctxN.x, ctxN.y, ctxN.z, wi.x, wi.y, wi.z);
}

}  // namespace wf
