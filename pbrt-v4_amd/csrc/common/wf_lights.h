// wf_lights.h — Light::SampleLi / PDF_Li / L / Le and the light samplers over the flat light table.
// Restates lights.h:205-233 (point), 262-290 (distant), 441-470 + lights.cpp:739-767 (diffuse area),
// 771-783 + lights.cpp:1360-1363 (spot), lights.cpp:950-972 (uniform infinite),
// lightsamplers.h:26-60 (uniform), 101-257 (CompactLightBounds), 260-358 (BVH sampler).
#pragma once

#include "wf_shapes.h"

namespace wf {

// LightSampleContext, base/light.h:120-160
struct LightCtx {
    P3i pi;
    N3 n, ns;
    WF_HD V3 p() const { return pi.mid(); }
};

struct LightLiSample {
    S4 L;
    V3 wi;
    float pdf;
    P3i pLightPi;  // pLight.pi
    N3 pLightN;    // pLight.n
    bool valid;
};

WF_HD bool IsDeltaLight(const wf_light &l) {
    const int type = l.type;
    if (l.flags & WF_LIGHTFLAG_DELTA_POSITION) return true;
    return type == WF_LIGHT_POINT || type == WF_LIGHT_SPOT || type == WF_LIGHT_DISTANT || type == WF_LIGHT_GONIOMETRIC || type == WF_LIGHT_PROJECTION;
}

// (SmoothStep: wf_noise.h)


// ---------------------------------------------------------------------------------------------
// ImageInfiniteLight (lights.h:566-662, lights.cpp:1042-1052)
// PiecewiseConstant2D::Sample / PDF (util/sampling.h:760-780) over [0,1]^2
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
// ImageInfiniteLight::ImageLe (lights.h:640-647): nearest texel with octahedral wrap (util/image.h:96-125,352-356),
// RGBIlluminantSpectrum of the clamped RGB (util/spectrum.cpp:235-246, util/spectrum.h:606-626)
// RGBIlluminantSpectrum(cs, ClampZero(rgb)).Sample(lambda) (util/spectrum.cpp:2674-2680, spectrum.h:620-640)
WF_HD S4 RGBIlluminantSample(const SceneView &sv, float r, float g, float b, const Wavelengths &lambda) {
    float rgb[3] = {fmax(0.f, r), fmax(0.f, g), fmax(0.f, b)};
    float m = fmax(fmax(rgb[0], rgb[1]), rgb[2]);
    float scale = 2 * m;
    float in[3] = {0, 0, 0};
    if (scale) { in[0] = rgb[0] / scale; in[1] = rgb[1] / scale; in[2] = rgb[2] / scale; }
    float c[3];
    RGBToSpectrumCoeffs(sv, in, c);
    S4 s;
    for (int i = 0; i < 4; ++i) s[i] = scale * SigmoidPoly(lambda.lambda[i], c[0], c[1], c[2]);
    return s * DenseSample(sv, sv.csIlluminantOffset, lambda);
}
WF_HD S4 ImageLightLe(const SceneView &sv, const wf_light &l, V2 uv, const Wavelengths &lambda) {
    const wf_image_light &im = sv.imageLights[l.image];
    const int res = im.res;
    int px = (int)(uv.x * res), py = (int)(uv.y * res);
    if (px < 0) { px = -px; py = res - 1 - py; }
    else if (px >= res) { px = 2 * res - 1 - px; py = res - 1 - py; }
    if (py < 0) { px = res - 1 - px; py = -py; }
    else if (py >= res) { px = res - 1 - px; py = 2 * res - 1 - py; }
    if (res == 1) { px = 0; py = 0; }
    const float *texel = sv.tableData + im.pixel_offset + 3 * ((size_t)py * res + px);
    return l.scale * RGBIlluminantSample(sv, texel[0], texel[1], texel[2], lambda);
}
// DiffuseAreaLight::AlphaMasked (lights.h:486-496): the alpha texture sees TextureEvalContext(Interaction(p, uv));
// a fractional alpha is resolved by HashFloat(p)
WF_HD bool AreaLightAlphaMasked(const SceneView &sv, const wf_light &l, V3 p, V2 uv) {
    if (l.alpha_tex_plus1 == 0) return false;
    TexCtx tc;
    tc.p = p;
    tc.uv = uv;
    float a = EvalFloatTexture(sv, l.alpha_tex_plus1 - 1, tc);
    if (a >= 1) return false;
    if (a <= 0) return true;
    return HashToFloat(Hash3f(p)) > a;
}
// DiffuseAreaLight::L, lights.h:441-463; with an image: Image::BilerpChannel at (u, 1 - v), clamp wrap
WF_HD S4 AreaLightL(const SceneView &sv, const wf_light &l, V3 p, N3 n, V2 uv, V3 w, const Wavelengths &lambda) {
    if (!(l.flags & WF_LIGHTFLAG_TWOSIDED) && Dot(n, w) < 0) return S4c(0.f);
    if (AreaLightAlphaMasked(sv, l, p, uv)) return S4c(0.f);
    if (l.image >= 0) {
        const wf_tex_image im = sv.texImages[l.image];
        V2 st{uv.x, 1 - uv.y};
        float r = ImageBilerpChannel(sv.tableData, im, 0, st, 0), g = ImageBilerpChannel(sv.tableData, im, 0, st, 1);
        float b = ImageBilerpChannel(sv.tableData, im, 0, st, 2);
        return l.scale * RGBIlluminantSample(sv, r, g, b, lambda);
    }
    return l.scale * DenseSample(sv, l.spectrum_offset, lambda);
}
WF_HD V3 XfApply3(const float m[4][4], V3 v) {
    return V3{m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
              m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
}

WF_HD LightLiSample LightSampleLi(const SceneView &sv, const wf_light &l, const LightCtx &ctx, V2 u,
                                  const Wavelengths &lambda, bool allowIncompletePDF) {
    LightLiSample ls{};
    ls.valid = false;
    switch (l.type) {
    case WF_LIGHT_DIFFUSE_AREA: {
        ShapeSampleR ss = l.tri >= sv.nTriangles ? SphereSample(sv, l.tri, ctx.pi, ctx.n, ctx.ns, u) : TriangleSample(sv, l.tri, ctx.pi, ctx.ns, u);
        if (!ss.valid || ss.pdf == 0 || LengthSquared(ss.pi.mid() - ctx.p()) == 0) return ls;
        V3 wi = Normalize(ss.pi.mid() - ctx.p());
        S4 Le = AreaLightL(sv, l, ss.pi.mid(), ss.n, ss.uv, -wi, lambda);
        if (!Le) return ls;
        ls.L = Le; ls.wi = wi; ls.pdf = ss.pdf; ls.pLightPi = ss.pi; ls.pLightN = ss.n; ls.valid = true;
        return ls;
    }
    case WF_LIGHT_POINT: {
        V3 p{l.pos[0], l.pos[1], l.pos[2]};
        V3 wi = Normalize(p - ctx.p());
        S4 Li = l.scale * DenseSample(sv, l.spectrum_offset, lambda) / DistanceSquared(p, ctx.p());
        ls.L = Li; ls.wi = wi; ls.pdf = 1; ls.pLightPi = MakeP3i(p); ls.pLightN = N3{0, 0, 0}; ls.valid = true;
        return ls;
    }
    case WF_LIGHT_SPOT: {
        V3 p{l.pos[0], l.pos[1], l.pos[2]};
        V3 wi = Normalize(p - ctx.p());
        const wf_transform &X = sv.lightXforms[l.xform];
        V3 mw = -wi;
        V3 wl{X.mInv[0][0] * mw.x + X.mInv[0][1] * mw.y + X.mInv[0][2] * mw.z,
              X.mInv[1][0] * mw.x + X.mInv[1][1] * mw.y + X.mInv[1][2] * mw.z,
              X.mInv[2][0] * mw.x + X.mInv[2][1] * mw.y + X.mInv[2][2] * mw.z};
        V3 wLight = Normalize(wl);
        S4 I = SmoothStep(wLight.z, l.cosFalloffEnd, l.cosFalloffStart) * l.scale * DenseSample(sv, l.spectrum_offset, lambda);
        S4 Li = I / DistanceSquared(p, ctx.p());
        if (!Li) return ls;
        ls.L = Li; ls.wi = wi; ls.pdf = 1; ls.pLightPi = MakeP3i(p); ls.pLightN = N3{0, 0, 0}; ls.valid = true;
        return ls;
    }
    case WF_LIGHT_GONIOMETRIC: {
        // GoniometricLight::SampleLi / I (lights.cpp:538-547, lights.h:393-396): Image::LookupNearestChannel, clamp wrap
        V3 p{l.pos[0], l.pos[1], l.pos[2]};
        V3 wi = Normalize(p - ctx.p());
        V3 w = XfApply3(sv.lightXforms[l.xform].mInv, -wi);
        V2 uv = EqualAreaSphereToSquare(w);
        const wf_tex_image &im = sv.texImages[l.image];
        int x = (int)(uv.x * im.res[0]), y = (int)(uv.y * im.res[1]);
        x = x < 0 ? 0 : (x > im.res[0] - 1 ? im.res[0] - 1 : x);
        y = y < 0 ? 0 : (y > im.res[1] - 1 ? im.res[1] - 1 : y);
        S4 I = l.scale * DenseSample(sv, l.spectrum_offset, lambda) * sv.tableData[im.level_offset[0] + (size_t)y * im.res[0] + x];
        ls.L = I / DistanceSquared(p, ctx.p());
        ls.wi = wi; ls.pdf = 1; ls.pLightPi = MakeP3i(p); ls.pLightN = N3{0, 0, 0}; ls.valid = true;
        return ls;
    }
    case WF_LIGHT_PROJECTION: {
        // ProjectionLight::SampleLi / I (lights.cpp:324-360)
        V3 p{l.pos[0], l.pos[1], l.pos[2]};
        V3 wi = Normalize(p - ctx.p());
        V3 wl = XfApply3(sv.lightXforms[l.xform].mInv, -wi);
        if (wl.z < 1e-3f) return ls;
        V3 ps = XfPoint(sv.lightXforms[l.xform2].m, wl);
        if (!(ps.x >= l.screen_bounds[0] && ps.x <= l.screen_bounds[2] && ps.y >= l.screen_bounds[1] && ps.y <= l.screen_bounds[3])) return ls;
        V2 uv{ps.x - l.screen_bounds[0], ps.y - l.screen_bounds[1]};
        if (l.screen_bounds[2] > l.screen_bounds[0]) uv.x /= l.screen_bounds[2] - l.screen_bounds[0];
        if (l.screen_bounds[3] > l.screen_bounds[1]) uv.y /= l.screen_bounds[3] - l.screen_bounds[1];
        const wf_tex_image &im = sv.texImages[l.image];
        int x = (int)(uv.x * im.res[0]), y = (int)(uv.y * im.res[1]);
        x = x < 0 ? 0 : (x > im.res[0] - 1 ? im.res[0] - 1 : x);
        y = y < 0 ? 0 : (y > im.res[1] - 1 ? im.res[1] - 1 : y);
        const float *texel = sv.tableData + im.level_offset[0] + 3 * ((size_t)y * im.res[0] + x);
        S4 Li = l.scale * RGBIlluminantSample(sv, texel[0], texel[1], texel[2], lambda) / DistanceSquared(p, ctx.p());
        if (!Li) return ls;
        ls.L = Li; ls.wi = wi; ls.pdf = 1; ls.pLightPi = MakeP3i(p); ls.pLightN = N3{0, 0, 0}; ls.valid = true;
        return ls;
    }
    case WF_LIGHT_DISTANT: {
        V3 wi{l.pos[0], l.pos[1], l.pos[2]};
        V3 pOutside = ctx.p() + wi * (2 * l.sceneRadius);
        ls.L = l.scale * DenseSample(sv, l.spectrum_offset, lambda);
        ls.wi = wi; ls.pdf = 1; ls.pLightPi = MakeP3i(pOutside); ls.pLightN = N3{0, 0, 0}; ls.valid = true;
        return ls;
    }
    case WF_LIGHT_UNIFORM_INFINITE: {
        if (allowIncompletePDF) return ls;
        V3 wi = SampleUniformSphere(u);
        ls.L = l.scale * DenseSample(sv, l.spectrum_offset, lambda);
        ls.wi = wi; ls.pdf = Inv4Pi;
        ls.pLightPi = MakeP3i(ctx.p() + wi * (2 * l.sceneRadius)); ls.pLightN = N3{0, 0, 0}; ls.valid = true;
        return ls;
    }
    case WF_LIGHT_IMAGE_INFINITE: {
        // lights.h:606-633
        const wf_image_light &im = sv.imageLights[l.image];
        float mapPDF = 0;
        V2 uv = PC2DSample(sv.tableData, allowIncompletePDF ? im.compensated : im.distribution, u, &mapPDF);
        if (mapPDF == 0) return ls;
        V3 wLight = EqualAreaSquareToSphere(uv);
        V3 wi = XfApply3(sv.lightXforms[l.xform].m, wLight);
        ls.L = ImageLightLe(sv, l, uv, lambda);
        ls.wi = wi; ls.pdf = mapPDF / (4 * Pi);
        ls.pLightPi = MakeP3i(ctx.p() + wi * (2 * l.sceneRadius)); ls.pLightN = N3{0, 0, 0}; ls.valid = true;
        return ls;
    }
    default: return ls;
    }
}

WF_HD float LightPDF_Li(const SceneView &sv, const wf_light &l, const LightCtx &ctx, V3 wi, bool allowIncompletePDF) {
    switch (l.type) {
    case WF_LIGHT_DIFFUSE_AREA:
        return l.tri >= sv.nTriangles ? SpherePDF(sv, l.tri, ctx.pi, ctx.n, ctx.ns, wi) : TrianglePDF(sv, l.tri, ctx.pi, ctx.n, ctx.ns, wi);
    case WF_LIGHT_UNIFORM_INFINITE: return allowIncompletePDF ? 0.f : Inv4Pi;
    case WF_LIGHT_IMAGE_INFINITE: {
        // lights.cpp:1042-1052
        const wf_image_light &im = sv.imageLights[l.image];
        V3 wLight = XfApply3(sv.lightXforms[l.xform].mInv, wi);
        V2 uv = EqualAreaSphereToSquare(wLight);
        return PC2DPDF(sv.tableData, allowIncompletePDF ? im.compensated : im.distribution, uv) / (4 * Pi);
    }
    default: return 0.f;
    }
}
// Light::Le for infinite lights (lights.h:172-174 for the others)
WF_HD S4 LightLe(const SceneView &sv, const wf_light &l, V3 rayd, const Wavelengths &lambda) {
    if (l.type == WF_LIGHT_UNIFORM_INFINITE) return l.scale * DenseSample(sv, l.spectrum_offset, lambda);
    if (l.type == WF_LIGHT_IMAGE_INFINITE) {
        // lights.h:597-601
        V3 wLight = Normalize(XfApply3(sv.lightXforms[l.xform].mInv, rayd));
        return ImageLightLe(sv, l, EqualAreaSphereToSquare(wLight), lambda);
    }
    return S4c(0.f);
}

// ---------------------------------------------------------------------------------------------
// CompactLightBounds::Importance, lightsamplers.h:144-201
WF_HD float LightBoundsImportance(const SceneView &sv, const wf_light_bvh_node &nd, V3 p, N3 n) {
    const float *ab = sv.allLightBounds;
    B3 bounds;
    bounds.pMin = V3{Lerp(nd.qb[0][0] / 65535.f, ab[0], ab[3]), Lerp(nd.qb[0][1] / 65535.f, ab[1], ab[4]), Lerp(nd.qb[0][2] / 65535.f, ab[2], ab[5])};
    bounds.pMax = V3{Lerp(nd.qb[1][0] / 65535.f, ab[0], ab[3]), Lerp(nd.qb[1][1] / 65535.f, ab[1], ab[4]), Lerp(nd.qb[1][2] / 65535.f, ab[2], ab[5])};
    uint32_t qo = nd.cos_bits & 0x7fffu, qe = (nd.cos_bits >> 15) & 0x7fffu;
    bool twoSided = (nd.cos_bits >> 30) & 1u;
    float cosTheta_o = 2 * (qo / 32767.f) - 1, cosTheta_e = 2 * (qe / 32767.f) - 1;
    V3 pc = (bounds.pMin + bounds.pMax) / 2;
    float d2 = DistanceSquared(p, pc);
    d2 = fmax(d2, Length(bounds.Diagonal()) / 2);
    auto cosSubClamped = [](float sinTheta_a, float cosTheta_a, float sinTheta_b, float cosTheta_b) -> float {
        if (cosTheta_a > cosTheta_b) return 1;
        return cosTheta_a * cosTheta_b + sinTheta_a * sinTheta_b;
    };
    auto sinSubClamped = [](float sinTheta_a, float cosTheta_a, float sinTheta_b, float cosTheta_b) -> float {
        if (cosTheta_a > cosTheta_b) return 0;
        return sinTheta_a * cosTheta_b - cosTheta_a * sinTheta_b;
    };
    V3 wi = Normalize(p - pc);
    V3 w = OctahedralToVector(nd.w_oct[0], nd.w_oct[1]);
    float cosTheta_w = Dot(w, wi);
    if (twoSided) cosTheta_w = abs(cosTheta_w);
    float sinTheta_w = SafeSqrt(1 - Sqr(cosTheta_w));
    float cosTheta_b = BoundSubtendedDirections(bounds, p).cosTheta;
    float sinTheta_b = SafeSqrt(1 - Sqr(cosTheta_b));
    float sinTheta_o = SafeSqrt(1 - Sqr(cosTheta_o));
    float cosTheta_x = cosSubClamped(sinTheta_w, cosTheta_w, sinTheta_o, cosTheta_o);
    float sinTheta_x = sinSubClamped(sinTheta_w, cosTheta_w, sinTheta_o, cosTheta_o);
    float cosThetap = cosSubClamped(sinTheta_x, cosTheta_x, sinTheta_b, cosTheta_b);
    if (cosThetap <= cosTheta_e) return 0;
    float importance = nd.phi * cosThetap / d2;
    if (!IsZero(n)) {
        float cosTheta_i = AbsDot(wi, n);
        float sinTheta_i = SafeSqrt(1 - Sqr(cosTheta_i));
        float cosThetap_i = cosSubClamped(sinTheta_i, cosTheta_i, sinTheta_b, cosTheta_b);
        importance *= cosThetap_i;
    }
    importance = fmax(importance, 0.f);
    return importance;
}

// LightSampler::Sample(ctx, u): returns light id or -1, and its pmf
WF_HD int LightSamplerSample(const SceneView &sv, const LightCtx &ctx, float u, float *pmfOut) {
    if (sv.lightSampler == WF_LS_UNIFORM) {
        // lightsamplers.h:33-38
        if (sv.nLights == 0) return -1;
        int lightIndex = (int)(u * sv.nLights);
        if (lightIndex > sv.nLights - 1) lightIndex = sv.nLights - 1;
        *pmfOut = 1.f / sv.nLights;
        return lightIndex;
    }
    if (sv.lightSampler == WF_LS_POWER) {
        // PowerLightSampler::Sample (lightsamplers.h:69-75) = AliasTable::Sample (util/sampling.cpp:88-113)
        if (sv.nLights == 0) return -1;
        int offset = (int)(u * sv.nLights);
        if (offset > sv.nLights - 1) offset = sv.nLights - 1;
        float up = fmin(u * sv.nLights - offset, OneMinusEpsilon);
        const float *bin = sv.powerAlias + 3 * offset;
        if (up < bin[0]) {
            *pmfOut = bin[1];
            return offset;
        }
        int alias = (int)FloatToBits(bin[2]);
        *pmfOut = sv.powerAlias[3 * alias + 1];
        return alias;
    }
    // BVHLightSampler::Sample, lightsamplers.h:266-320
    int nInf = sv.nInfiniteLights;
    bool nodesEmpty = sv.nLightBvhNodes == 0;
    float pInfinite = float(nInf) / float(nInf + (nodesEmpty ? 0 : 1));
    if (u < pInfinite) {
        u /= pInfinite;
        int index = (int)(u * nInf);
        if (index > nInf - 1) index = nInf - 1;
        *pmfOut = pInfinite / nInf;
        return sv.infiniteLights[index];
    }
    if (nodesEmpty) return -1;
    V3 p = ctx.p();
    N3 n = ctx.ns;
    u = fmin((u - pInfinite) / (1 - pInfinite), OneMinusEpsilon);
    int nodeIndex = 0;
    float pmf = 1 - pInfinite;
    while (true) {
        wf_light_bvh_node node = sv.lightBvh[nodeIndex];
        bool isLeaf = node.child_or_light >> 31;
        int childOrLight = (int)(node.child_or_light & 0x7fffffffu);
        if (!isLeaf) {
            float ci0 = LightBoundsImportance(sv, sv.lightBvh[nodeIndex + 1], p, n);
            float ci1 = LightBoundsImportance(sv, sv.lightBvh[childOrLight], p, n);
            if (ci0 == 0 && ci1 == 0) return -1;
            float nodePMF;
            int child = SampleDiscrete2(ci0, ci1, u, &nodePMF, &u);
            pmf *= nodePMF;
            nodeIndex = (child == 0) ? (nodeIndex + 1) : childOrLight;
        } else {
            if (nodeIndex > 0 || LightBoundsImportance(sv, node, p, n) > 0) {
                *pmfOut = pmf;
                return childOrLight;
            }
            return -1;
        }
    }
}

// LightSampler::PMF(ctx, light)
WF_HD float LightSamplerPMF(const SceneView &sv, const LightCtx &ctx, int lightId) {
    if (sv.lightSampler == WF_LS_UNIFORM) return sv.nLights == 0 ? 0.f : 1.f / sv.nLights;
    if (sv.lightSampler == WF_LS_POWER) return sv.nLights == 0 ? 0.f : sv.powerAlias[3 * lightId + 1];  // lightsamplers.h:78-82
    // BVHLightSampler::PMF, lightsamplers.h:323-358
    const wf_light &l = sv.lights[lightId];
    int nInf = sv.nInfiniteLights;
    bool nodesEmpty = sv.nLightBvhNodes == 0;
    if (l.bit_trail < 0) return 1.f / (nInf + (nodesEmpty ? 0 : 1));
    uint32_t bitTrail = (uint32_t)l.bit_trail;
    V3 p = ctx.p();
    N3 n = ctx.ns;
    float pInfinite = float(nInf) / float(nInf + (nodesEmpty ? 0 : 1));
    float pmf = 1 - pInfinite;
    int nodeIndex = 0;
    while (true) {
        const wf_light_bvh_node node = sv.lightBvh[nodeIndex];
        if (node.child_or_light >> 31) return pmf;
        int child1 = (int)(node.child_or_light & 0x7fffffffu);
        float ci[2] = {LightBoundsImportance(sv, sv.lightBvh[nodeIndex + 1], p, n), LightBoundsImportance(sv, sv.lightBvh[child1], p, n)};
        pmf *= ci[bitTrail & 1] / (ci[0] + ci[1]);
        nodeIndex = (bitTrail & 1) ? child1 : (nodeIndex + 1);
        bitTrail >>= 1;
    }
}

}  // namespace wf
