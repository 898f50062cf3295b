// wf_media.h — participating media on the hot path: HomogeneousMedium / GridMedium majorant iteration,
// SampleT_maj (delta tracking driver), point sampling of the medium properties.  Restates media.h:40-352,
// 724-800 and util/containers.h:765-850 operation for operation (host branches; FastExp is the polynomial
// of util/math.h:450-474, not an intrinsic).
#pragma once

#include "wf_camera.h"

namespace wf {

// Hash(Point3f, Float) / Hash(Vector3f) as used for the medium RNGs (wavefront/media.cpp:44, intersect.h:170)
WF_HD uint64_t Hash3f1(V3 p, float t) {
    uint32_t w[4] = {FloatToBits(p.x), FloatToBits(p.y), FloatToBits(p.z), FloatToBits(t)};
    return HashWords(w, 4);
}
constexpr float WF_FLT_MAX = 3.402823466e+38f;

// general SampleDiscrete (util/sampling.h:79-113), weights in a small array
WF_HD int SampleDiscreteN(const float *weights, int n, float u) {
    float sumWeights = 0;
    for (int i = 0; i < n; ++i) sumWeights += weights[i];
    float up = u * sumWeights;
    if (up == sumWeights) up = NextFloatDown(up);
    int offset = 0;
    float sum = 0;
    while (sum + weights[offset] <= up) sum += weights[offset++];
    return offset;
}

struct MediumProps {  // MediumProperties, media.h:71-76 (phase = HG with parameter g)
    S4 sigma_a, sigma_s, Le;
    float g;
};
struct MajorantSeg { float tMin, tMax; S4 sigma_maj; };  // RayMajorantSegment, base/medium.h

// SampledGrid<Float>::Lookup (util/containers.h:797-826)
WF_HD float GridLookupI(const float *v, int nx, int ny, int nz, int x, int y, int z) {
    if (!(x >= 0 && x < nx && y >= 0 && y < ny && z >= 0 && z < nz)) return 0.f;
    return v[((size_t)z * ny + y) * nx + x];
}
WF_HD float GridLookup(const float *v, int nx, int ny, int nz, V3 p) {
    V3 ps{p.x * nx - .5f, p.y * ny - .5f, p.z * nz - .5f};
    int ix = (int)floor(ps.x), iy = (int)floor(ps.y), iz = (int)floor(ps.z);
    V3 d{ps.x - (float)ix, ps.y - (float)iy, ps.z - (float)iz};
    float d00 = Lerp(d.x, GridLookupI(v, nx, ny, nz, ix, iy, iz), GridLookupI(v, nx, ny, nz, ix + 1, iy, iz));
    float d10 = Lerp(d.x, GridLookupI(v, nx, ny, nz, ix, iy + 1, iz), GridLookupI(v, nx, ny, nz, ix + 1, iy + 1, iz));
    float d01 = Lerp(d.x, GridLookupI(v, nx, ny, nz, ix, iy, iz + 1), GridLookupI(v, nx, ny, nz, ix + 1, iy, iz + 1));
    float d11 = Lerp(d.x, GridLookupI(v, nx, ny, nz, ix, iy + 1, iz + 1), GridLookupI(v, nx, ny, nz, ix + 1, iy + 1, iz + 1));
    return Lerp(d.z, Lerp(d.y, d00, d10), Lerp(d.y, d01, d11));
}
// The same lookup over the corner-packed copy of the grid (SceneView::gridCorners): cell (ix + 1, iy + 1, iz + 1) of an
// (nx + 1)(ny + 1)(nz + 1) table holds v(ix.., iy.., iz..) for the eight corners in the order the lerps below take them, zeros outside the
// grid as GridLookupI returns them.  The same eight values through the same expressions: bit-identical.
// Where cell (cx, cy, cz) of the corner-packed table lives (round 6): the table is BRICKED — 8 x 8 x 8 cells of 32 bytes = one 16 KiB brick,
// bricks in row-major order — so that lookups that are close in space (neighbouring rays, successive steps of one ray) fall into the same pages and
// lines (in the row-major table a step along z moves 8.4 MB on a 512^3 grid).  Built on the suspicion that the delta-tracking kernel was TLB-bound; it was
// not (see WF_GRID_BRICKS).
#ifndef WF_GRID_BRICKS
#define WF_GRID_BRICKS 0   // 1: the bricked table (A/B builds) — measured once the kernel's real bound (the scatter counter's atomics) was gone: 12.2 against 12.1 ms, no gain: the row-major table of round 4 stays
#endif
WF_HD size_t GridCornerBricks(int n) { return (size_t)((n + 1 + 7) >> 3); }   // bricks along an axis of n voxels (n + 1 cells)
WF_HD size_t GridCornerCells(int nx, int ny, int nz) {   // cells the table holds, padding included
    if (!WF_GRID_BRICKS) return (size_t)(nx + 1) * (ny + 1) * (nz + 1);
    return GridCornerBricks(nx) * GridCornerBricks(ny) * GridCornerBricks(nz) * 512;
}
WF_HD size_t GridCornerIndex(int nx, int ny, int cx, int cy, int cz) {
    if (!WF_GRID_BRICKS) return ((size_t)cz * (ny + 1) + cy) * (nx + 1) + cx;
    const size_t brick = ((size_t)(cz >> 3) * GridCornerBricks(ny) + (size_t)(cy >> 3)) * GridCornerBricks(nx) + (size_t)(cx >> 3);
    return brick * 512 + (size_t)(((cz & 7) << 6) | ((cy & 7) << 3) | (cx & 7));
}
WF_HD float GridLookupPacked(const float *c, int nx, int ny, int nz, V3 p) {
    V3 ps{p.x * nx - .5f, p.y * ny - .5f, p.z * nz - .5f};
    int ix = (int)floor(ps.x), iy = (int)floor(ps.y), iz = (int)floor(ps.z);
    V3 d{ps.x - (float)ix, ps.y - (float)iy, ps.z - (float)iz};
    if (!(ix >= -1 && ix < nx && iy >= -1 && iy < ny && iz >= -1 && iz < nz)) {
        // every corner lies outside the grid: the lerps of zeros
        float z0 = Lerp(d.x, 0.f, 0.f);
        return Lerp(d.z, Lerp(d.y, z0, z0), Lerp(d.y, z0, z0));
    }
    struct alignas(16) G4 { float a, b, c, d; };
    const G4 *cell = reinterpret_cast<const G4 *>(c) + 2 * GridCornerIndex(nx, ny, ix + 1, iy + 1, iz + 1);
    const G4 lo = cell[0], hi = cell[1];
    float d00 = Lerp(d.x, lo.a, lo.b);
    float d10 = Lerp(d.x, lo.c, lo.d);
    float d01 = Lerp(d.x, hi.a, hi.b);
    float d11 = Lerp(d.x, hi.c, hi.d);
    return Lerp(d.z, Lerp(d.y, d00, d10), Lerp(d.y, d01, d11));
}
// Bounds3::Offset (util/vecmath.h)
WF_HD V3 BoundsOffset(const float b[6], V3 p) {
    V3 o{p.x - b[0], p.y - b[1], p.z - b[2]};
    if (b[3] > b[0]) o.x /= b[3] - b[0];
    if (b[4] > b[1]) o.y /= b[4] - b[1];
    if (b[5] > b[2]) o.z /= b[5] - b[2];
    return o;
}
// Bounds3::IntersectP(o, d, tMax, &t0, &t1), util/vecmath.h:1545-1572
WF_HD bool BoundsIntersectT(const float b[6], V3 o, V3 d, float tMax, float *hitt0, float *hitt1) {
    float t0 = 0, t1 = tMax;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    for (int i = 0; i < 3; ++i) {
        float invRayDir = 1 / dd[i];
        float tNear = (b[i] - oo[i]) * invRayDir;
        float tFar = (b[3 + i] - oo[i]) * invRayDir;
        if (tNear > tFar) { float t = tNear; tNear = tFar; tFar = t; }
        tFar *= 1 + 2 * gamma(3);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    *hitt0 = t0;
    *hitt1 = t1;
    return true;
}
// Transform::ApplyInverse(Point3f) (util/transform.h:387-399) and (Ray, tMax) (:350-364 with the Point3fi
// form of util/transform.cpp for an exact point)
WF_HD V3 XfInvPoint(const wf_transform &t, V3 p) {
    const float (*m)[4] = t.mInv;
    float x = p.x, y = p.y, z = p.z;
    float xp = (m[0][0] * x + m[0][1] * y) + (m[0][2] * z + m[0][3]);
    float yp = (m[1][0] * x + m[1][1] * y) + (m[1][2] * z + m[1][3]);
    float zp = (m[2][0] * x + m[2][1] * y) + (m[2][2] * z + m[2][3]);
    float wp = (m[3][0] * x + m[3][1] * y) + (m[3][2] * z + m[3][3]);
    if (wp == 1) return V3{xp, yp, zp};
    return V3{xp, yp, zp} / wp;
}
WF_HD void XfInvRay(const wf_transform &t, V3 *o, V3 *d, float *tMax) {
    const float (*m)[4] = t.mInv;
    float x = o->x, y = o->y, z = o->z;
    float xp = (m[0][0] * x + m[0][1] * y) + (m[0][2] * z + m[0][3]);
    float yp = (m[1][0] * x + m[1][1] * y) + (m[1][2] * z + m[1][3]);
    float zp = (m[2][0] * x + m[2][1] * y) + (m[2][2] * z + m[2][3]);
    V3 pe;
    pe.x = gamma(3) * (abs(m[0][0] * x) + abs(m[0][1] * y) + abs(m[0][2] * z));
    pe.y = gamma(3) * (abs(m[1][0] * x) + abs(m[1][1] * y) + abs(m[1][2] * z));
    pe.z = gamma(3) * (abs(m[2][0] * x) + abs(m[2][1] * y) + abs(m[2][2] * z));
    P3i oi = MakeP3i(V3{xp, yp, zp}, pe);  // affine transforms only (wp == 1)
    V3 dd = XfVector(t.mInv, *d);
    float lengthSquared = LengthSquared(dd);
    if (lengthSquared > 0) {
        float dt = Dot(Abs(dd), oi.err()) / lengthSquared;
        V3 off = dd * dt;
        o->x = IntervalAddMid(oi.lo.x, oi.hi.x, off.x);
        o->y = IntervalAddMid(oi.lo.y, oi.hi.y, off.y);
        o->z = IntervalAddMid(oi.lo.z, oi.hi.z, off.z);
        *tMax -= dt;
    } else *o = oi.mid();
    *d = dd;
}

// Medium::SamplePoint: HomogeneousMedium media.h:247-252, GridMedium media.h:283-317.  The spectra sampled at the
// ray's wavelengths (sigma_a_spec.Sample(lambda) ...) do not depend on the point: MediumAtLambda holds them once per
// ray instead of once per tracking event (same values).
struct MediumAtLambda { S4 sigma_a, sigma_s, Le, lam; };
// SampledGrid<RGB*Spectrum>::Lookup(p, convert) (util/containers.h:790-826): trilinear interpolation of the CONVERTED cells
WF_HD S4 RGBGridCell(const float *v, int nx, int ny, int nz, int x, int y, int z, const S4 &lam) {
    if (!(x >= 0 && x < nx && y >= 0 && y < ny && z >= 0 && z < nz)) return S4c(0.f);
    const float *c = v + 4 * (((size_t)z * ny + y) * nx + x);
    S4 s;
    for (int i = 0; i < 4; ++i) s[i] = c[3] * SigmoidPoly(lam[i], c[0], c[1], c[2]);
    return s;
}
WF_HD S4 LerpS(float t, const S4 &a, const S4 &b) { return (1 - t) * a + t * b; }
WF_HD S4 RGBGridLookup(const float *v, int nx, int ny, int nz, V3 p, const S4 &lam) {
    V3 ps{p.x * nx - .5f, p.y * ny - .5f, p.z * nz - .5f};
    int ix = (int)floor(ps.x), iy = (int)floor(ps.y), iz = (int)floor(ps.z);
    V3 d{ps.x - (float)ix, ps.y - (float)iy, ps.z - (float)iz};
    S4 d00 = LerpS(d.x, RGBGridCell(v, nx, ny, nz, ix, iy, iz, lam), RGBGridCell(v, nx, ny, nz, ix + 1, iy, iz, lam));
    S4 d10 = LerpS(d.x, RGBGridCell(v, nx, ny, nz, ix, iy + 1, iz, lam), RGBGridCell(v, nx, ny, nz, ix + 1, iy + 1, iz, lam));
    S4 d01 = LerpS(d.x, RGBGridCell(v, nx, ny, nz, ix, iy, iz + 1, lam), RGBGridCell(v, nx, ny, nz, ix + 1, iy, iz + 1, lam));
    S4 d11 = LerpS(d.x, RGBGridCell(v, nx, ny, nz, ix, iy + 1, iz + 1, lam), RGBGridCell(v, nx, ny, nz, ix + 1, iy + 1, iz + 1, lam));
    return LerpS(d.z, LerpS(d.y, d00, d10), LerpS(d.y, d01, d11));
}
// nanovdb::SampleFromVoxels<Tree, 1, false> on a dense block: Map::applyInverseMapF (fmaf chain of matMult), Floor, TrilinearSampler::sample
WF_HD float VdbSample(const float *data, const int32_t vmin[3], const int32_t vdim[3], float background, const float m[9], const float vec[3], V3 p) {
    const float x = p.x - vec[0], y = p.y - vec[1], z = p.z - vec[2];
    float u = fma(x, m[0], fma(y, m[1], z * m[2])), v = fma(x, m[3], fma(y, m[4], z * m[5])), w = fma(x, m[6], fma(y, m[7], z * m[8]));
    const float fu = floor(u), fv = floor(v), fw = floor(w);
    u -= fu; v -= fv; w -= fw;
    // (indices clamped to just outside the block before the integer conversion: a point far outside — or a NaN — reads the background,
    //  never an int overflow; inside the block nothing changes: ADVICE r3)
    auto idx = [](float f, int lo, int n) {
        const long long q = (long long)(int)fmin(fmax(f, -2147483000.f), 2147483000.f) - lo;
        return (int)(q < -2 ? -2 : (q > (long long)n + 1 ? (long long)n + 1 : q));
    };
    const int i = idx(fu, vmin[0], vdim[0]), j = idx(fv, vmin[1], vdim[1]), k = idx(fw, vmin[2], vdim[2]);
    auto at = [&](int a, int b, int c) -> float {
        const int xi = i + a, yj = j + b, zk = k + c;
        if (!(xi >= 0 && xi < vdim[0] && yj >= 0 && yj < vdim[1] && zk >= 0 && zk < vdim[2])) return background;
        return data[((size_t)zk * vdim[1] + yj) * vdim[0] + xi];
    };
    auto lerp = [](float a, float b, float t) { return a + t * (b - a); };
    return lerp(lerp(lerp(at(0, 0, 0), at(0, 0, 1), w), lerp(at(0, 1, 0), at(0, 1, 1), w), v),
                lerp(lerp(at(1, 0, 0), at(1, 0, 1), w), lerp(at(1, 1, 0), at(1, 1, 1), w), v), u);
}
WF_HD MediumAtLambda MediumSpectra(const SceneView &sv, const wf_medium &M, const Wavelengths &lambda) {
    MediumAtLambda ml;
    ml.sigma_a = DenseSample(sv, M.sigma_a_offset, lambda);
    ml.sigma_s = DenseSample(sv, M.sigma_s_offset, lambda);
    ml.Le = (M.type == WF_MEDIUM_HOMOGENEOUS || M.is_emissive) ? DenseSample(sv, M.le_offset, lambda) : S4c(0.f);
    for (int i = 0; i < 4; ++i) ml.lam[i] = lambda.lambda[i];
    return ml;
}
// LEAN (round 6, the device's k_medium_sample<true>): homogeneous media and non-emissive uniform-grid media only — the procedural cloud's
// noise, the NanoVDB sampler, the RGB grids and the blackbody emission are compiled out, so that the delta-tracking kernel of a scene
// without them is not allocated their registers ("a kernel is allocated what it can reach").  Same arithmetic on the types it keeps.
template <bool LEAN = false>
WF_HD MediumProps MediumSamplePoint(const SceneView &sv, const wf_medium &M, const MediumAtLambda &ml, V3 p) {
    MediumProps mp;
    mp.g = M.g;
    mp.sigma_a = ml.sigma_a;
    mp.sigma_s = ml.sigma_s;
    if (M.type == WF_MEDIUM_HOMOGENEOUS) {
        mp.Le = ml.Le;
        return mp;
    }
    p = XfInvPoint(M.render_from_medium, p);
    if constexpr (!LEAN)
    if (M.type == WF_MEDIUM_CLOUD) {
        // CloudMedium::SamplePoint + Density (media.h:464-471, 493-517)
        const int32_t *perm = sv.noisePerm;
        V3 pp = M.cloud_frequency * p;
        if (M.cloud_wispiness > 0) {
            float vomega = 0.05f * M.cloud_wispiness, vlambda = 10.f;
            for (int i = 0; i < 2; ++i) {
                // DNoise (util/noise.cpp:110-116)
                V3 q = vlambda * pp;
                const float delta = .01f;
                float n = Noise3(perm, q);
                V3 nd{Noise3(perm, q + V3{delta, 0, 0}), Noise3(perm, q + V3{0, delta, 0}), Noise3(perm, q + V3{0, 0, delta})};
                pp = pp + vomega * ((nd - V3{n, n, n}) / delta);
                vomega *= 0.5f;
                vlambda *= 1.99f;
            }
        }
        float d = 0;
        float omega = 0.5f, lambda = 1.f;
        for (int i = 0; i < 5; ++i) {
            d += omega * Noise3(perm, lambda * pp);
            omega *= 0.5f;
            lambda *= 1.99f;
        }
        d = Clamp((1 - p.y) * 4.5f * M.cloud_density * d, 0.f, 1.f);
        d += 2 * fmax(0.f, 0.5f - p.y);
        d = Clamp(d, 0.f, 1.f);
        mp.sigma_a = d * ml.sigma_a;
        mp.sigma_s = d * ml.sigma_s;
        mp.Le = S4c(0.f);
        return mp;
    }
    if constexpr (!LEAN)
    if (M.type == WF_MEDIUM_NANOVDB) {
        // NanoVDBMedium::SamplePoint + Le (media.h:615-632, 655-668): densityFloatGrid->worldToIndexF(p), then
        // SampleFromVoxels<TreeType, 1, false> = trilinear interpolation of the eight surrounding voxels (parity unpinned: see wf_abi.h)
        const float d = VdbSample(sv.mediumData + M.density_offset, M.vdb_min, M.vdb_dim, M.vdb_background, M.vdb_inv_mat, M.vdb_vec, p);
        mp.sigma_a = ml.sigma_a * d;
        mp.sigma_s = ml.sigma_s * d;
        mp.Le = S4c(0.f);
        if (M.is_emissive) {
            float temp = VdbSample(sv.mediumData + M.temperature_offset, M.vdbt_min, M.vdbt_dim, M.vdbt_background, M.vdbt_inv_mat, M.vdbt_vec, p);
            temp = (temp - M.temperature_shift) * M.temperature_scale;
            if (temp > 100.f) {
                float lambdaMax = 2.8977721e-3f / temp;
                float norm = 1 / Blackbody(lambdaMax * 1e9f, temp);
                S4 bb;
                for (int i = 0; i < 4; ++i) bb[i] = Blackbody(ml.lam[i], temp) * norm;
                mp.Le = M.le_scale * bb;
            }
        }
        return mp;
    }
    p = BoundsOffset(M.bounds, p);
    if constexpr (!LEAN)
    if (M.type == WF_MEDIUM_RGB_GRID) {
        // RGBGridMedium::SamplePoint, media.h:377-401 (ml.Le = the colour space's illuminant at lambda)
        mp.sigma_a = M.sigma_scale * (M.rgb_a_offset >= 0 ? RGBGridLookup(sv.mediumData + M.rgb_a_offset, M.nx, M.ny, M.nz, p, ml.lam) : S4c(1.f));
        mp.sigma_s = M.sigma_scale * (M.rgb_s_offset >= 0 ? RGBGridLookup(sv.mediumData + M.rgb_s_offset, M.nx, M.ny, M.nz, p, ml.lam) : S4c(1.f));
        mp.Le = S4c(0.f);
        if (M.is_emissive) {
            // the RGBIlluminantSpectrum cells: scale * sigmoid, times the illuminant, interpolated after conversion
            const float *v = sv.mediumData + M.rgb_le_offset;
            const int nx = M.nx, ny = M.ny, nz = M.nz;
            V3 ps{p.x * nx - .5f, p.y * ny - .5f, p.z * nz - .5f};
            int ix = (int)floor(ps.x), iy = (int)floor(ps.y), iz = (int)floor(ps.z);
            V3 d{ps.x - (float)ix, ps.y - (float)iy, ps.z - (float)iz};
            auto cell = [&](int x, int y, int z) { return RGBGridCell(v, nx, ny, nz, x, y, z, ml.lam) * ml.Le; };
            S4 d00 = LerpS(d.x, cell(ix, iy, iz), cell(ix + 1, iy, iz));
            S4 d10 = LerpS(d.x, cell(ix, iy + 1, iz), cell(ix + 1, iy + 1, iz));
            S4 d01 = LerpS(d.x, cell(ix, iy, iz + 1), cell(ix + 1, iy, iz + 1));
            S4 d11 = LerpS(d.x, cell(ix, iy + 1, iz + 1), cell(ix + 1, iy + 1, iz + 1));
            mp.Le = M.le_scale * LerpS(d.z, LerpS(d.y, d00, d10), LerpS(d.y, d01, d11));
        }
        return mp;
    }
    float d;
    {
        const long long cb = sv.gridCornerBase ? sv.gridCornerBase[&M - sv.media] : -1;
        d = cb >= 0 ? GridLookupPacked(sv.gridCorners + cb, M.nx, M.ny, M.nz, p) : GridLookup(sv.mediumData + M.density_offset, M.nx, M.ny, M.nz, p);
    }
    mp.sigma_a = mp.sigma_a * d;
    mp.sigma_s = mp.sigma_s * d;
    mp.Le = S4c(0.f);
    if constexpr (!LEAN)
    if (M.is_emissive) {
        float scale = GridLookup(sv.mediumData + M.le_scale_offset, M.le_nx, M.le_ny, M.le_nz, p);
        if (scale > 0) {
            if (M.temperature_offset >= 0) {
                // blackbody emission from the temperature grid (media.h:305-318, util/spectrum.h:486-520)
                float temp = GridLookup(sv.mediumData + M.temperature_offset, M.nx, M.ny, M.nz, p);
                temp = (temp - M.temperature_shift) * M.temperature_scale;
                if (temp > 100.f) {
                    float lambdaMax = 2.8977721e-3f / temp;
                    float norm = 1 / Blackbody(lambdaMax * 1e9f, temp);
                    S4 bb;
                    for (int i = 0; i < 4; ++i) bb[i] = Blackbody(ml.lam[i], temp) * norm;
                    mp.Le = scale * bb;
                }
            } else mp.Le = scale * ml.Le;
        }
    }
    return mp;
}

// HomogeneousMajorantIterator (media.h:79-102) and DDAMajorantIterator (media.h:136-214) behind one interface
struct MajorantIter {
    // homogeneous
    bool homogeneous, called;
    MajorantSeg seg;
    // DDA
    S4 sigma_t;
    float tMin, tMax;
    const float *voxels;
    int res[3];
    float nextCrossingT[3], deltaT[3];
    int step[3], voxelLimit[3], voxel[3];

    WF_HD bool Next(MajorantSeg *out) {
        if (homogeneous) {
            if (called) return false;
            called = true;
            *out = seg;
            return true;
        }
        // (the reference: `if (tMin >= tMax) return {}`, media.h:181.  Written so that a NaN interval ends the iteration as well: a ray with a NaN
        //  direction — it happens: a light sample that came out NaN, the path the library carries on with after WF_FATAL_CHECK_NAN_PDF — has
        //  tMax = NaN, every crossing time NaN, and would step through the grid's memory backwards for ever; the reference turns the NaN
        //  coordinate into voxel INT_MIN and segfaults (fuzz scenes s2400094, s3200123).  Same result for every ordered pair.)
        if (!(tMin < tMax)) return false;
        int bits = ((nextCrossingT[0] < nextCrossingT[1]) << 2) + ((nextCrossingT[0] < nextCrossingT[2]) << 1) +
                   ((nextCrossingT[1] < nextCrossingT[2]));
        // cmpToAxis = {2, 1, 2, 1, 2, 2, 0, 0}
        int stepAxis = (bits == 6 || bits == 7) ? 0 : ((bits == 1 || bits == 3) ? 1 : 2);
        float nct = stepAxis == 0 ? nextCrossingT[0] : (stepAxis == 1 ? nextCrossingT[1] : nextCrossingT[2]);
        float tVoxelExit = fmin(tMax, nct);
        S4 sigma_maj = sigma_t * voxels[voxel[0] + res[0] * (voxel[1] + res[1] * voxel[2])];
        *out = MajorantSeg{tMin, tVoxelExit, sigma_maj};
        tMin = tVoxelExit;
        if (nct > tMax) tMin = tMax;
        for (int a = 0; a < 3; ++a)
            if (a == stepAxis) {
                voxel[a] += step[a];
                if (voxel[a] == voxelLimit[a]) tMin = tMax;
                nextCrossingT[a] += deltaT[a];
            }
        return true;
    }
};

// Medium::SampleRay for a ray with unit-length direction (SampleT_maj normalises first)
template <bool LEAN = false>
WF_HD MajorantIter MediumSampleRay(const SceneView &sv, const wf_medium &M, const MediumAtLambda &ml, V3 o, V3 d, float raytMax) {
    MajorantIter it;
    S4 sigma_a = ml.sigma_a;
    S4 sigma_s = ml.sigma_s;
    if (M.type == WF_MEDIUM_HOMOGENEOUS) {
        it.homogeneous = true;
        it.called = false;
        it.seg = MajorantSeg{0, raytMax, sigma_a + sigma_s};
        return it;
    }
    it.homogeneous = false;
    it.called = true;
    it.tMin = WF_INFINITY;   // default-constructed DDA iterator: Next() returns nothing
    it.tMax = -WF_INFINITY;
    XfInvRay(M.render_from_medium, &o, &d, &raytMax);
    float tMin, tMax;
    if constexpr (!LEAN)
    if (M.type == WF_MEDIUM_CLOUD) {
        // CloudMedium::SampleRay (media.h:474-488): one HomogeneousMajorantIterator over the box overlap (none: the default iterator)
        it.homogeneous = true;
        it.called = !BoundsIntersectT(M.bounds, o, d, raytMax, &tMin, &tMax);
        if (!it.called) it.seg = MajorantSeg{tMin, tMax, sigma_a + sigma_s};
        return it;
    }
    if (!BoundsIntersectT(M.bounds, o, d, raytMax, &tMin, &tMax)) return it;
    it.sigma_t = M.type == WF_MEDIUM_RGB_GRID ? S4c(1.f) : sigma_a + sigma_s;  // RGBGridMedium::SampleRay: sigma_t(1), media.h:413
    it.tMin = tMin;
    it.tMax = tMax;
    it.voxels = sv.mediumData + M.maj_offset;
    for (int a = 0; a < 3; ++a) it.res[a] = M.maj_res[a];
    // DDAMajorantIterator ctor, media.h:141-178
    const float diag[3] = {M.bounds[3] - M.bounds[0], M.bounds[4] - M.bounds[1], M.bounds[5] - M.bounds[2]};
    V3 og = BoundsOffset(M.bounds, o);
    float rd[3] = {d.x / diag[0], d.y / diag[1], d.z / diag[2]};
    // rayGrid(tMin) = o + d * t
    const float gi[3] = {og.x + rd[0] * tMin, og.y + rd[1] * tMin, og.z + rd[2] * tMin};
    for (int a = 0; a < 3; ++a) {
        it.voxel[a] = (int)Clamp(gi[a] * it.res[a], 0.f, (float)(it.res[a] - 1));
        it.deltaT[a] = 1 / (abs(rd[a]) * it.res[a]);
        if (rd[a] == -0.f) rd[a] = 0.f;
        if (rd[a] >= 0) {
            float nextVoxelPos = float(it.voxel[a] + 1) / it.res[a];
            it.nextCrossingT[a] = tMin + (nextVoxelPos - gi[a]) / rd[a];
            it.step[a] = 1;
            it.voxelLimit[a] = it.res[a];
        } else {
            float nextVoxelPos = float(it.voxel[a]) / it.res[a];
            it.nextCrossingT[a] = tMin + (nextVoxelPos - gi[a]) / rd[a];
            it.step[a] = -1;
            it.voxelLimit[a] = -1;
        }
    }
    return it;
}

// (Round 6, measured and dropped for THIS function: the two nested loops flattened into one, as the medium-sample stage's state machine
//  does (MediumTrackStep, wf_kernels.h: -15 % there) — the transmittance walks that call this lose: 21.3 against 19.2 ms on the cloud scene,
//  same box, profiles/r06_sampleT_maj_flattened_for_transmittance_ab_cloud16.txt.  Their callback is a few multiplications: the nested form's
//  tight inner loop is worth more than the lanes the flattening keeps busy.)
// SampleT_maj, media.h:724-800.  callback(p, mp, sigma_maj, T_maj) -> continue?
#ifndef WF_TMAJ_SPEC
#define WF_TMAJ_SPEC 0   // measured and LEFT OFF (round 6, cloud scene: Intersect shadow (Tr) 19.2 against 19.0 ms, profiles/r06_sampleT_maj_two_collisions_ab_cloud16.txt)
#endif
// MLEAN: the lean medium code (MediumSamplePoint<true>: homogeneous and non-emissive uniform-grid media only), for the transmittance kernels of scenes with no other medium type
template <bool MLEAN = false, typename F>
WF_HD S4 SampleT_maj(const SceneView &sv, int mediumId, V3 o, V3 d, float tMax, float u, RNG &rng, const Wavelengths &lambda, F callback) {
    const wf_medium &M = sv.media[mediumId];
    tMax *= Length(d);
    d = Normalize(d);
    const MediumAtLambda ml = MediumSpectra(sv, M, lambda);
    MajorantIter iter = MediumSampleRay<MLEAN>(sv, M, ml, o, d, tMax);
    S4 T_maj = S4c(1.f);
    bool done = false;
    while (!done) {
        MajorantSeg seg;
        if (!iter.Next(&seg)) return T_maj;
        if (seg.sigma_maj[0] == 0) {
            float dt = seg.tMax - seg.tMin;
            if (IsInf(dt)) dt = WF_FLT_MAX;
            T_maj = T_maj * FastExp(-dt * seg.sigma_maj);
            continue;
        }
        float tMin = seg.tMin;
        while (true) {
            float t = tMin + SampleExponential(u, seg.sigma_maj[0]);
            u = rng.UniformFloat();
            if (t < seg.tMax) {
                T_maj = T_maj * FastExp(-(t - tMin) * seg.sigma_maj);
                V3 p = o + d * t;
#if WF_TMAJ_SPEC
                // TWO tentative collisions per iteration for the grid media (round 6): where the next exponential step lands depends only on
                // this one's position and on the sample value already drawn (u), not on the density here, so both points' density gathers are
                // issued together; the second event then runs exactly as the next iteration would (same operands, the RNG draws in the same
                // order).  Bit-identical; no gain on the cloud scene's transmittance walks either.
                const bool gridMedium = M.type == WF_MEDIUM_GRID || M.type == WF_MEDIUM_RGB_GRID || M.type == WF_MEDIUM_NANOVDB;
                const float t2 = t + SampleExponential(u, seg.sigma_maj[0]);
                const bool spec = gridMedium && t2 < seg.tMax;
                MediumProps mp = MediumSamplePoint<MLEAN>(sv, M, ml, p);
                V3 p2 = p;
                MediumProps mp2 = mp;
                if (spec) { p2 = o + d * t2; mp2 = MediumSamplePoint<MLEAN>(sv, M, ml, p2); }
                if (!callback(p, mp, seg.sigma_maj, T_maj)) {
                    done = true;
                    break;
                }
                T_maj = S4c(1.f);
                tMin = t;
                if (spec) {
                    u = rng.UniformFloat();
                    T_maj = T_maj * FastExp(-(t2 - tMin) * seg.sigma_maj);
                    if (!callback(p2, mp2, seg.sigma_maj, T_maj)) {
                        done = true;
                        break;
                    }
                    T_maj = S4c(1.f);
                    tMin = t2;
                }
#else
                MediumProps mp = MediumSamplePoint<MLEAN>(sv, M, ml, p);
                if (!callback(p, mp, seg.sigma_maj, T_maj)) {
                    done = true;
                    break;
                }
                T_maj = S4c(1.f);
                tMin = t;
#endif
            } else {
                float dt = seg.tMax - tMin;
                if (IsInf(dt)) dt = WF_FLT_MAX;
                T_maj = T_maj * FastExp(-dt * seg.sigma_maj);
                break;
            }
        }
    }
    return S4c(1.f);
}

}  // namespace wf
