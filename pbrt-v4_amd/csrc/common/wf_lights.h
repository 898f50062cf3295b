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
// (PC2DSample / PC2DPDF: wf_shapes.h, where the bilinear patches' image distribution needs them too)
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
    // (a NaN or infinite direction — the tail of a path that already went wrong — gives coordinates the reflection above does not bring
    //  back: the reference reads outside its image there; no kernel may)
    px = Clamp(px, 0, res - 1); py = Clamp(py, 0, res - 1);
    const auto texel = sv.tableData + im.pixel_offset + 3 * ((size_t)py * res + px);
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
// (ALPHA = false: a caller that knows no emitter of the scene has an alpha texture — the texture-graph evaluator behind AlphaMasked is an
//  out-of-line callee of 160+ VGPRs, and a reachable callee sets the caller's allocation)
template <bool ALPHA = true>
WF_HD S4 AreaLightL(const SceneView &sv, const wf_light &l, V3 p, N3 n, V2 uv, V3 w, const Wavelengths &lambda) {
    if (!(l.flags & WF_LIGHTFLAG_TWOSIDED) && Dot(n, w) < 0) return S4c(0.f);
    if constexpr (ALPHA)
    if (AreaLightAlphaMasked(sv, l, p, uv)) return S4c(0.f);
    if (l.image >= 0) {
        const wf_tex_image &im = sv.texImages[l.image];
        V2 st{uv.x, 1 - uv.y};
        float r = ImageBilerpChannel(sv.tableData, im, 0, st, 0), g = ImageBilerpChannel(sv.tableData, im, 0, st, 1);
        float b = ImageBilerpChannel(sv.tableData, im, 0, st, 2);
        return l.scale * RGBIlluminantSample(sv, r, g, b, lambda);
    }
    return l.scale * DenseSample(sv, l.spectrum_offset, lambda);
}

// ---------------------------------------------------------------------------------------------
// PortalImageInfiniteLight (lights.h:631-731, lights.cpp:1109-1336): an environment map seen through a rectangular portal,
// re-parametrised so that the portal bounds an axis-aligned rectangle of the image for every reference point, sampled through a
// summed-area table (WindowedPiecewiseConstant2D, util/sampling.h:830-990)
struct B2 { V2 pMin, pMax; };
WF_HD Frame PortalFrame(const wf_image_light &im) {
    Frame f;
    f.x = V3{im.portal_frame[0][0], im.portal_frame[0][1], im.portal_frame[0][2]};
    f.y = V3{im.portal_frame[1][0], im.portal_frame[1][1], im.portal_frame[1][2]};
    f.z = V3{im.portal_frame[2][0], im.portal_frame[2][1], im.portal_frame[2][2]};
    return f;
}
WF_HD bool PortalImageFromRender(const wf_image_light &im, V3 wRender, V2 *uv, float *duv_dw) {
    V3 w = PortalFrame(im).ToLocal(wRender);
    if (w.z <= 0) return false;
    if (duv_dw) *duv_dw = Sqr(Pi) * (1 - Sqr(w.x)) * (1 - Sqr(w.y)) / w.z;
    float alpha = atan2(w.x, w.z), beta = atan2(w.y, w.z);
    *uv = V2{Clamp((alpha + Pi / 2) / Pi, 0.f, 1.f), Clamp((beta + Pi / 2) / Pi, 0.f, 1.f)};
    return true;
}
WF_HD V3 PortalRenderFromImage(const wf_image_light &im, V2 uv, float *duv_dw) {
    float alpha = -Pi / 2 + uv.x * Pi, beta = -Pi / 2 + uv.y * Pi;
    float x = tan(alpha), y = tan(beta);
    V3 w = Normalize(V3{x, y, 1});
    if (duv_dw) *duv_dw = Sqr(Pi) * (1 - Sqr(w.x)) * (1 - Sqr(w.y)) / w.z;
    return PortalFrame(im).FromLocal(w);
}
WF_HD bool PortalImageBounds(const wf_image_light &im, V3 p, B2 *b) {
    V3 c0{im.portal[0][0], im.portal[0][1], im.portal[0][2]}, c2{im.portal[2][0], im.portal[2][1], im.portal[2][2]};
    V2 p0, p1;
    if (!PortalImageFromRender(im, Normalize(c0 - p), &p0, nullptr)) return false;
    if (!PortalImageFromRender(im, Normalize(c2 - p), &p1, nullptr)) return false;
    b->pMin = V2{fmin(p0.x, p1.x), fmin(p0.y, p1.y)};
    b->pMax = V2{fmax(p0.x, p1.x), fmax(p0.y, p1.y)};
    return true;
}
WF_HD float SATLookupInt(const double *sum, int n, int x, int y) {
    if (x == 0 || y == 0) return 0;
    x = x - 1 < n - 1 ? x - 1 : n - 1;
    y = y - 1 < n - 1 ? y - 1 : n - 1;
    return (float)sum[(size_t)y * n + x];
}
WF_HD float SATLookup(const double *sum, int n, float x, float y) {
    x *= n; y *= n;
    int x0 = (int)x, y0 = (int)y;
    float v00 = SATLookupInt(sum, n, x0, y0), v10 = SATLookupInt(sum, n, x0 + 1, y0);
    float v01 = SATLookupInt(sum, n, x0, y0 + 1), v11 = SATLookupInt(sum, n, x0 + 1, y0 + 1);
    float dx = x - int(x), dy = y - int(y);
    return (1 - dx) * (1 - dy) * v00 + (1 - dx) * dy * v01 + dx * (1 - dy) * v10 + dx * dy * v11;
}
WF_HD float SATIntegral(const double *sum, int n, const B2 &e) {
    double s = (((double)SATLookup(sum, n, e.pMax.x, e.pMax.y) - (double)SATLookup(sum, n, e.pMin.x, e.pMax.y)) +
                ((double)SATLookup(sum, n, e.pMin.x, e.pMin.y) - (double)SATLookup(sum, n, e.pMax.x, e.pMin.y)));
    float r = (float)(s / (n * n));
    return r > 0 ? r : 0.f;
}
WF_HD float WindowedEval(const float *func, int n, V2 p) {
    int px = (int)(p.x * n), py = (int)(p.y * n);
    px = px < n - 1 ? px : n - 1;
    py = py < n - 1 ? py : n - 1;
    return func[(size_t)py * n + px];
}
// WindowedPiecewiseConstant2D::Sample / SampleBisection (util/sampling.h:903-975)
WF_HD bool WindowedSample(const float *func, const double *sum, int n, V2 u, const B2 &b, V2 *pOut, float *pdf) {
    const float bInt = SATIntegral(sum, n, b);
    if (bInt == 0) return false;
    V2 p;
    {
        auto Px = [&](float x) { B2 bx = b; bx.pMax.x = x; return SATIntegral(sum, n, bx) / bInt; };
        float lo = b.pMin.x, hi = b.pMax.x;
        while (ceil(n * hi) - floor(n * lo) > 1) {
            float mid = (lo + hi) / 2;
            if (Px(mid) > u.x) hi = mid; else lo = mid;
        }
        float t = (u.x - Px(lo)) / (Px(hi) - Px(lo));
        p.x = Clamp(Lerp(t, lo, hi), lo, hi);
    }
    B2 bCond{V2{floor(p.x * n) / n, b.pMin.y}, V2{ceil(p.x * n) / n, b.pMax.y}};
    if (bCond.pMin.x == bCond.pMax.x) bCond.pMax.x += 1.f / n;
    const float condIntegral = SATIntegral(sum, n, bCond);
    if (condIntegral == 0) return false;
    {
        auto Py = [&](float y) { B2 by = bCond; by.pMax.y = y; return SATIntegral(sum, n, by) / condIntegral; };
        float lo = b.pMin.y, hi = b.pMax.y;
        while (ceil(n * hi) - floor(n * lo) > 1) {
            float mid = (lo + hi) / 2;
            if (Py(mid) > u.y) hi = mid; else lo = mid;
        }
        float t = (u.y - Py(lo)) / (Py(hi) - Py(lo));
        p.y = Clamp(Lerp(t, lo, hi), lo, hi);
    }
    *pdf = WindowedEval(func, n, p) / bInt;
    *pOut = p;
    return true;
}
// PortalImageInfiniteLight::ImageLookup (lights.cpp:1217-1224): Image::LookupNearestChannel, clamp wrap
WF_HD S4 PortalImageLookup(const SceneView &sv, float scale, const wf_image_light &im, V2 uv, const Wavelengths &lambda) {
    const int res = im.res;
    int px = (int)(uv.x * res), py = (int)(uv.y * res);
    px = px < 0 ? 0 : (px > res - 1 ? res - 1 : px);
    py = py < 0 ? 0 : (py > res - 1 ? res - 1 : py);
    const auto texel = sv.tableData + im.pixel_offset + 3 * ((size_t)py * res + px);
    return scale * RGBIlluminantSample(sv, fmax(0.f, texel[0]), fmax(0.f, texel[1]), fmax(0.f, texel[2]), lambda);
}
// out of line: the sampling loops are long and only scenes with a portal light reach them
// (the light's fields come by value: a pointer to the caller's wf_light would pin its copy in scratch)
WF_NI void PortalSampleLiP(const SceneView *svp, int image, float scale, float sceneRadius, float px, float py, float pz, float u0, float u1,
                           const Wavelengths *lambda, LightLiSample *out) {
    const SceneView &sv = *svp;
    const wf_image_light &im = sv.imageLights[image];
    LightLiSample ls{};
    ls.valid = false;
    *out = ls;
    const V3 p{px, py, pz};
    B2 b;
    if (!PortalImageBounds(im, p, &b)) return;
    const float *func = sv.tableData + im.func_offset;
    const double *sum = (const double *)((const float *)sv.tableData + im.sat_offset);
    float mapPDF;
    V2 uv;
    if (!WindowedSample(func, sum, im.res, V2{u0, u1}, b, &uv, &mapPDF)) return;
    float duv_dw;
    V3 wi = PortalRenderFromImage(im, uv, &duv_dw);
    if (duv_dw == 0) return;
    ls.L = PortalImageLookup(sv, scale, im, uv, *lambda);
    ls.wi = wi; ls.pdf = mapPDF / duv_dw;
    ls.pLightPi = MakeP3i(p + wi * (2 * sceneRadius)); ls.pLightN = N3{0, 0, 0}; ls.valid = true;
    *out = ls;
}
WF_NI float PortalPDFLiP(const SceneView *svp, int image, float px, float py, float pz, float wx, float wy, float wz) {
    const SceneView &sv = *svp;
    const wf_image_light &im = sv.imageLights[image];
    float duv_dw;
    V2 uv;
    if (!PortalImageFromRender(im, V3{wx, wy, wz}, &uv, &duv_dw) || duv_dw == 0) return 0;
    B2 b;
    if (!PortalImageBounds(im, V3{px, py, pz}, &b)) return 0;
    const double *sum = (const double *)((const float *)sv.tableData + im.sat_offset);
    float funcInt = SATIntegral(sum, im.res, b);
    if (funcInt == 0) return 0;
    return WindowedEval(sv.tableData + im.func_offset, im.res, uv) / funcInt / duv_dw;
}
// PortalImageInfiniteLight::Le (lights.cpp:1208-1215)
WF_NI void PortalLeP(const SceneView *svp, int image, float scale, float ox, float oy, float oz, float dx, float dy, float dz, const Wavelengths *lambda, S4 *out) {
    const SceneView &sv = *svp;
    const wf_image_light &im = sv.imageLights[image];
    *out = S4c(0.f);
    V2 uv;
    B2 b;
    if (!PortalImageFromRender(im, Normalize(V3{dx, dy, dz}), &uv, nullptr)) return;
    if (!PortalImageBounds(im, V3{ox, oy, oz}, &b)) return;
    if (!(uv.x >= b.pMin.x && uv.x <= b.pMax.x && uv.y >= b.pMin.y && uv.y <= b.pMax.y)) return;   // Inside(Point2f, Bounds2f)
    *out = PortalImageLookup(sv, scale, im, uv, *lambda);
}
WF_HD V3 XfApply3(const float m[4][4], V3 v) {
    return V3{m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
              m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
}

template <bool RARE = true>
WF_HD LightLiSample LightSampleLi(const SceneView &sv, const wf_light &l, const LightCtx &ctx, V2 u,
                                  const Wavelengths &lambda, bool allowIncompletePDF) {
    LightLiSample ls{};
    ls.valid = false;
    switch (l.type) {
    case WF_LIGHT_DIFFUSE_AREA: {
        // (RARE = false: a kernel for scenes whose emitters are all triangles — the out-of-line sampler of sphere / disk / cylinder / patch
        //  emitters needs 214 VGPRs, and a callee that is merely reachable sets the kernel's allocation: round 5, tools/exp/light_vgprs.hip)
        ShapeSampleR ss = (RARE && l.tri >= sv.nTriangles) ? SphereSample(sv, l.tri, ctx.pi, ctx.n, ctx.ns, u) : TriangleSample(sv, l.tri, ctx.pi, ctx.ns, u);
        if (!ss.valid || ss.pdf == 0 || LengthSquared(ss.pi.mid() - ctx.p()) == 0) return ls;
        V3 wi = Normalize(ss.pi.mid() - ctx.p());
        S4 Le = AreaLightL<RARE>(sv, l, ss.pi.mid(), ss.n, ss.uv, -wi, lambda);
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
        const auto texel = sv.tableData + im.level_offset[0] + 3 * ((size_t)y * im.res[0] + x);
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
    case WF_LIGHT_PORTAL_INFINITE: if constexpr (RARE) {
        // locals for everything whose address the out-of-line callee gets: `ls` and `lambda` themselves stay in registers on the other paths
        const V3 p = ctx.p();
        const Wavelengths lam = lambda;
        LightLiSample tmp;
        PortalSampleLiP(sv.self, l.image, l.scale, l.sceneRadius, p.x, p.y, p.z, u.x, u.y, &lam, &tmp);
        return tmp;
    } else return ls;
    default: return ls;
    }
}

// AREA = false: the caller only ever asks for infinite lights (HandleEscapedRays); RARE = false: no portal infinite light, no emitter that
// is not a triangle — the callees a kernel cannot reach do not set its register allocation (wf_scene.h "LEAN DEVICE VARIANTS")
template <bool AREA = true, bool RARE = true>
WF_HD float LightPDF_Li(const SceneView &sv, const wf_light &l, const LightCtx &ctx, V3 wi, bool allowIncompletePDF) {
    switch (l.type) {
    case WF_LIGHT_DIFFUSE_AREA:
        if constexpr (!AREA) return 0.f;
        else return (RARE && l.tri >= sv.nTriangles) ? SpherePDF(sv, l.tri, ctx.pi, ctx.n, ctx.ns, wi) : TrianglePDF(sv, l.tri, ctx.pi, ctx.n, ctx.ns, wi);
    case WF_LIGHT_UNIFORM_INFINITE: return allowIncompletePDF ? 0.f : Inv4Pi;
    case WF_LIGHT_IMAGE_INFINITE: {
        // lights.cpp:1042-1052
        const wf_image_light &im = sv.imageLights[l.image];
        V3 wLight = XfApply3(sv.lightXforms[l.xform].mInv, wi);
        V2 uv = EqualAreaSphereToSquare(wLight);
        return PC2DPDF(sv.tableData, allowIncompletePDF ? im.compensated : im.distribution, uv) / (4 * Pi);
    }
    case WF_LIGHT_PORTAL_INFINITE: {
        if constexpr (!RARE) return 0.f;
        else {
        const V3 p = ctx.p();
        return PortalPDFLiP(sv.self, l.image, p.x, p.y, p.z, wi.x, wi.y, wi.z);
        }
    }
    default: return 0.f;
    }
}
// Light::Le for infinite lights (lights.h:172-174 for the others)
template <bool RARE = true>
WF_HD S4 LightLe(const SceneView &sv, const wf_light &l, V3 rayo, V3 rayd, const Wavelengths &lambda) {
    if (RARE && l.type == WF_LIGHT_PORTAL_INFINITE) {
        S4 Le;
        const Wavelengths lam = lambda;
        PortalLeP(sv.self, l.image, l.scale, rayo.x, rayo.y, rayo.z, rayd.x, rayd.y, rayd.z, &lam, &Le);
        return Le;
    }
    if (l.type == WF_LIGHT_UNIFORM_INFINITE) return l.scale * DenseSample(sv, l.spectrum_offset, lambda);
    if (l.type == WF_LIGHT_IMAGE_INFINITE) {
        // lights.h:597-601
        V3 wLight = Normalize(XfApply3(sv.lightXforms[l.xform].mInv, rayd));
        return ImageLightLe(sv, l, EqualAreaSphereToSquare(wLight), lambda);
    }
    return S4c(0.f);
}

// ---------------------------------------------------------------------------------------------
// CompactLightBounds::Importance, lightsamplers.h:144-201 — in two parts (round 5).  Everything the reference recomputes from the node's
// quantised fields at every visit is a CONSTANT of the node: the de-quantised bounds (six divisions by 65535 and six lerps), the two cone
// cosines (two divisions), sin(theta_o) (a square root), the octahedral axis (two divisions, a normalisation), the bounds' centre, half
// diagonal and bounding-sphere radius (BoundSubtendedDirections, util/vecmath.h; three more square roots).  ExpandLightNode evaluates
// them — the reference's expressions, in its order — and LightBoundsImportanceX does the point-dependent rest.  The HIP back end expands
// every node once at upload (SceneView::lightBvhX, 80 bytes per node instead of 32) and the descents of next-event estimation run the second
// part only: the two child evaluations of a level cost about half the instructions (IEEE division is ten instructions on gfx950).
// The CPU checker expands on the fly: the same two functions, the same bits.
#ifndef WF_LIGHT_NODES_EXPANDED
#define WF_LIGHT_NODES_EXPANDED 1   // 0: the device expands at every visit too (A/B builds)
#endif
struct alignas(16) LightNodeX {
    V3 pMin; float phi;
    V3 pMax; float radius;      // Bounds3::BoundingSphere: Inside(centre, b) ? Distance(centre, pMax) : 0
    V3 pc; float halfDiag;      // (pMin + pMax) / 2, Length(Diagonal()) / 2
    V3 w; float cosTheta_o;
    float sinTheta_o, cosTheta_e;
    uint32_t twoSided, child_or_light;
};
WF_HD LightNodeX ExpandLightNode(const float *ab, const wf_light_bvh_node &nd) {
    LightNodeX x;
    x.pMin = V3{Lerp(nd.qb[0][0] / 65535.f, ab[0], ab[3]), Lerp(nd.qb[0][1] / 65535.f, ab[1], ab[4]), Lerp(nd.qb[0][2] / 65535.f, ab[2], ab[5])};
    x.pMax = V3{Lerp(nd.qb[1][0] / 65535.f, ab[0], ab[3]), Lerp(nd.qb[1][1] / 65535.f, ab[1], ab[4]), Lerp(nd.qb[1][2] / 65535.f, ab[2], ab[5])};
    const uint32_t qo = nd.cos_bits & 0x7fffu, qe = (nd.cos_bits >> 15) & 0x7fffu;
    x.twoSided = (nd.cos_bits >> 30) & 1u;
    x.cosTheta_o = 2 * (qo / 32767.f) - 1;
    x.cosTheta_e = 2 * (qe / 32767.f) - 1;
    x.sinTheta_o = SafeSqrt(1 - Sqr(x.cosTheta_o));
    B3 bounds;
    bounds.pMin = x.pMin; bounds.pMax = x.pMax;
    x.pc = (bounds.pMin + bounds.pMax) / 2;
    x.halfDiag = Length(bounds.Diagonal()) / 2;
    x.radius = Inside(x.pc, bounds) ? Distance(x.pc, bounds.pMax) : 0;
    x.w = OctahedralToVector(nd.w_oct[0], nd.w_oct[1]);
    x.phi = nd.phi;
    x.child_or_light = nd.child_or_light;
    return x;
}
WF_HD float LightBoundsImportanceX(const LightNodeX &x, V3 p, N3 n) {
    const V3 pc = x.pc;
    float d2 = DistanceSquared(p, pc);
    d2 = fmax(d2, x.halfDiag);
    auto cosSubClamped = [](float sinTheta_a, float cosTheta_a, float sinTheta_b, float cosTheta_b) -> float {
        if (cosTheta_a > cosTheta_b) return 1;
        return cosTheta_a * cosTheta_b + sinTheta_a * sinTheta_b;
    };
    auto sinSubClamped = [](float sinTheta_a, float cosTheta_a, float sinTheta_b, float cosTheta_b) -> float {
        if (cosTheta_a > cosTheta_b) return 0;
        return sinTheta_a * cosTheta_b - cosTheta_a * sinTheta_b;
    };
    V3 wi = Normalize(p - pc);
    float cosTheta_w = Dot(x.w, wi);
    if (x.twoSided) cosTheta_w = abs(cosTheta_w);
    float sinTheta_w = SafeSqrt(1 - Sqr(cosTheta_w));
    // BoundSubtendedDirections(bounds, p).cosTheta (wf_math.h) with the node's centre and radius
    float cosTheta_b;
    if (DistanceSquared(p, pc) < Sqr(x.radius)) cosTheta_b = -1.f;
    else {
        float sin2ThetaMax = Sqr(x.radius) / DistanceSquared(pc, p);
        cosTheta_b = SafeSqrt(1 - sin2ThetaMax);
    }
    float sinTheta_b = SafeSqrt(1 - Sqr(cosTheta_b));
    float cosTheta_x = cosSubClamped(sinTheta_w, cosTheta_w, x.sinTheta_o, x.cosTheta_o);
    float sinTheta_x = sinSubClamped(sinTheta_w, cosTheta_w, x.sinTheta_o, x.cosTheta_o);
    float cosThetap = cosSubClamped(sinTheta_x, cosTheta_x, sinTheta_b, cosTheta_b);
    if (cosThetap <= x.cosTheta_e) return 0;
    float importance = x.phi * cosThetap / d2;
    if (!IsZero(n)) {
        float cosTheta_i = AbsDot(wi, n);
        float sinTheta_i = SafeSqrt(1 - Sqr(cosTheta_i));
        float cosThetap_i = cosSubClamped(sinTheta_i, cosTheta_i, sinTheta_b, cosTheta_b);
        importance *= cosThetap_i;
    }
    importance = fmax(importance, 0.f);
    return importance;
}
// the importance of node `index` for the point (p, n): from the expanded table on the device, expanded on the fly on the host
WF_HD float LightNodeImportance(const SceneView &sv, int index, V3 p, N3 n) {
#if defined(__HIP_DEVICE_COMPILE__) && WF_LIGHT_NODES_EXPANDED
    const LightNodeX x = sv.lightBvhX[index];
    return LightBoundsImportanceX(x, p, n);
#else
    return LightBoundsImportanceX(ExpandLightNode(sv.allLightBounds, sv.lightBvh[index]), p, n);
#endif
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
        const auto bin = sv.powerAlias + 3 * offset;
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
            float ci0 = LightNodeImportance(sv, nodeIndex + 1, p, n);
            float ci1 = LightNodeImportance(sv, childOrLight, p, n);
            if (ci0 == 0 && ci1 == 0) return -1;
            float nodePMF;
            int child = SampleDiscrete2(ci0, ci1, u, &nodePMF, &u);
            pmf *= nodePMF;
            nodeIndex = (child == 0) ? (nodeIndex + 1) : childOrLight;
        } else {
            if (nodeIndex > 0 || LightNodeImportance(sv, nodeIndex, p, n) > 0) {
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
        float ci[2] = {LightNodeImportance(sv, nodeIndex + 1, p, n), LightNodeImportance(sv, child1, p, n)};
        pmf *= ci[bitTrail & 1] / (ci[0] + ci[1]);
        nodeIndex = (bitTrail & 1) ? child1 : (nodeIndex + 1);
        bitTrail >>= 1;
    }
}

}  // namespace wf
