// wf_bssrdf.h — subsurface scattering (K12): the tabulated BSSRDF, its spline helpers and the probe-segment record.
// Restates, operation for operation:
//   util/math.cpp:157-285     CatmullRomWeights, InvertCatmullRom, IntegrateCatmullRom;  util/math.h:661-695 NewtonBisection
//   util/sampling.cpp:425-483 SampleCatmullRom2D
//   util/scattering.cpp:10-36 FresnelMoment1 / FresnelMoment2
//   bssrdf.h:110-281          TabulatedBSSRDF (Sr, SampleSr, PDF_Sr, SampleSp, PDF_Sp, ProbeIntersectionToSample), SubsurfaceFromDiffuse
//   bssrdf.cpp:26-128         BeamDiffusionMS / BeamDiffusionSS / ComputeBeamDiffusionBSSRDF (host, at scene load)
//   bxdfs.h:1073-1132         NormalizedFresnelBxDF
//   materials.h:747-765       SubsurfaceMaterial::GetBSSRDF
#pragma once

#include "wf_bxdf.h"

namespace wf {

// BSSRDFTable (bssrdf.h:73-96; SubsurfaceMaterial builds it with 100 albedo x 64 radius samples, materials.h:719) as one run
// of floats in the scene's table pool: rhoSamples[100] radiusSamples[64] profile[100*64] rhoEff[100] profileCDF[100*64]
constexpr int BSSRDF_NRHO = 100, BSSRDF_NRADIUS = 64;
constexpr int BSSRDF_TABLE_FLOATS = BSSRDF_NRHO + BSSRDF_NRADIUS + BSSRDF_NRHO * BSSRDF_NRADIUS + BSSRDF_NRHO + BSSRDF_NRHO * BSSRDF_NRADIUS;
struct BSSRDFTableView {
    const float *rhoSamples, *radiusSamples, *profile, *rhoEff, *profileCDF;
    WF_HD explicit BSSRDFTableView(const float *base)
        : rhoSamples(base), radiusSamples(base + BSSRDF_NRHO), profile(base + BSSRDF_NRHO + BSSRDF_NRADIUS),
          rhoEff(profile + BSSRDF_NRHO * BSSRDF_NRADIUS), profileCDF(rhoEff + BSSRDF_NRHO) {}
    WF_HD float EvalProfile(int rhoIndex, int radiusIndex) const { return profile[rhoIndex * BSSRDF_NRADIUS + radiusIndex]; }
};

// util/math.cpp:157-199
WF_HD bool CatmullRomWeights(const float *nodes, int n, float x, int *offset, float weights[4]) {
    if (!(x >= nodes[0] && x <= nodes[n - 1])) return false;
    int idx = FindInterval(n, [&](int i) { return nodes[i] <= x; });
    *offset = idx - 1;
    float x0 = nodes[idx], x1 = nodes[idx + 1];
    float t = (x - x0) / (x1 - x0), t2 = t * t, t3 = t2 * t;
    weights[1] = 2 * t3 - 3 * t2 + 1;
    weights[2] = -2 * t3 + 3 * t2;
    if (idx > 0) {
        float w0 = (t3 - 2 * t2 + t) * (x1 - x0) / (x1 - nodes[idx - 1]);
        weights[0] = -w0;
        weights[2] += w0;
    } else {
        float w0 = t3 - 2 * t2 + t;
        weights[0] = 0;
        weights[1] -= w0;
        weights[2] += w0;
    }
    if (idx + 2 < n) {
        float w3 = (t3 - t2) * (x1 - x0) / (nodes[idx + 2] - x0);
        weights[1] -= w3;
        weights[3] = w3;
    } else {
        float w3 = t3 - t2;
        weights[1] -= w3;
        weights[2] += w3;
        weights[3] = 0;
    }
    return true;
}

// util/math.h:661-695.  f(t, &value, &derivative)
template <typename Func>
WF_HD float NewtonBisection(float x0, float x1, Func f, float xEps = 1e-6f, float fEps = 1e-6f) {
    float fx0, fx1, d;
    f(x0, &fx0, &d);
    f(x1, &fx1, &d);
    if (abs(fx0) < fEps) return x0;
    if (abs(fx1) < fEps) return x1;
    bool startIsNegative = fx0 < 0;
    float xMid = x0 + (x1 - x0) * -fx0 / (fx1 - fx0);
    while (true) {
        if (!(x0 < xMid && xMid < x1)) xMid = (x0 + x1) / 2;
        float fMid, dMid;
        f(xMid, &fMid, &dMid);
        if (startIsNegative == (fMid < 0)) x0 = xMid;
        else x1 = xMid;
        if ((x1 - x0) < xEps || abs(fMid) < fEps) return xMid;
        xMid -= fMid / dMid;
    }
}

// util/math.cpp:227-265
WF_HD float InvertCatmullRom(const float *nodes, const float *f, int n, float u) {
    if (!(u > f[0])) return nodes[0];
    else if (!(u < f[n - 1])) return nodes[n - 1];
    int i = FindInterval(n, [&](int k) { return f[k] <= u; });
    float x0 = nodes[i], x1 = nodes[i + 1];
    float f0 = f[i], f1 = f[i + 1];
    float width = x1 - x0;
    float d0 = (i > 0) ? width * (f1 - f[i - 1]) / (x1 - nodes[i - 1]) : (f1 - f0);
    float d1 = (i + 2 < n) ? width * (f[i + 2] - f0) / (nodes[i + 2] - x0) : (f1 - f0);
    auto eval = [&](float t, float *Fv, float *fv) {
        float t2 = t * t, t3 = t2 * t;
        float Fhat = (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
        float fhat = (6 * t2 - 6 * t) * f0 + (-6 * t2 + 6 * t) * f1 + (3 * t2 - 4 * t + 1) * d0 + (3 * t2 - 2 * t) * d1;
        *Fv = Fhat - u;
        *fv = fhat;
    };
    float t = NewtonBisection(0.f, 1.f, eval);
    return x0 + t * width;
}

// util/math.cpp:267-285
WF_HD float IntegrateCatmullRom(const float *nodes, const float *f, int n, float *cdf) {
    float sum = 0;
    cdf[0] = 0;
    for (int i = 0; i < n - 1; ++i) {
        float x0 = nodes[i], x1 = nodes[i + 1];
        float f0 = f[i], f1 = f[i + 1];
        float width = x1 - x0;
        float d0 = (i > 0) ? width * (f1 - f[i - 1]) / (x1 - nodes[i - 1]) : (f1 - f0);
        float d1 = (i + 2 < n) ? width * (f[i + 2] - f0) / (nodes[i + 2] - x0) : (f1 - f0);
        sum += width * ((f0 + f1) / 2 + (d0 - d1) / 12);
        cdf[i + 1] = sum;
    }
    return sum;
}

// util/sampling.cpp:425-483 (fval / pdf outputs unused by the BSSRDF)
WF_HD float SampleCatmullRom2D(const float *nodes1, int n1, const float *nodes2, int n2, const float *values, const float *cdf, float alpha, float u) {
    int offset;
    float weights[4];
    if (!CatmullRomWeights(nodes1, n1, alpha, &offset, weights)) return 0;
    auto interpolate = [&](const float *array, int idx) {
        float v = 0;
        for (int i = 0; i < 4; ++i)
            if (weights[i] != 0) v += array[(offset + i) * n2 + idx] * weights[i];
        return v;
    };
    float maximum = interpolate(cdf, n2 - 1);
    u *= maximum;
    int idx = FindInterval(n2, [&](int i) { return interpolate(cdf, i) <= u; });
    float f0 = interpolate(values, idx), f1 = interpolate(values, idx + 1);
    float x0 = nodes2[idx], x1 = nodes2[idx + 1];
    float width = x1 - x0;
    float d0, d1;
    u = (u - interpolate(cdf, idx)) / width;
    if (idx > 0) d0 = width * (f1 - interpolate(values, idx - 1)) / (x1 - nodes2[idx - 1]);
    else d0 = f1 - f0;
    if (idx + 2 < n2) d1 = width * (interpolate(values, idx + 2) - f0) / (nodes2[idx + 2] - x0);
    else d1 = f1 - f0;
    auto eval = [&](float t, float *Fv, float *fv) {
        // EvaluatePolynomial(t, c0, c1, c2, c3[, c4]) = fma(t, EvaluatePolynomial(t, c1, ...), c0)  (util/math.h:329-337)
        float c3 = (1.f / 3.f) * (-2 * d0 - d1) + f1 - f0, c4 = 0.25f * (d0 + d1) + 0.5f * (f0 - f1);
        float Fhat = fmaf(t, fmaf(t, fmaf(t, fmaf(t, c4, c3), 0.5f * d0), f0), 0.f);
        float e2 = -2 * d0 - d1 + 3 * (f1 - f0), e3 = d0 + d1 + 2 * (f0 - f1);
        float fhat = fmaf(t, fmaf(t, fmaf(t, e3, e2), d0), f0);
        *Fv = Fhat - u;
        *fv = fhat;
    };
    float t = NewtonBisection(0.f, 1.f, eval);
    return x0 + width * t;
}

// util/scattering.cpp:10-36.  (The unsuffixed 3.904945 makes the tail of the first polynomial a double expression.)
WF_HD float FresnelMoment1(float eta) {
    float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1)
        return (float)((double)(0.45966f - 1.73965f * eta + 3.37668f * eta2) - 3.904945 * (double)eta3 + (double)(2.49277f * eta4) - (double)(0.68441f * eta5));
    else
        return -4.61686f + 11.1136f * eta - 10.4646f * eta2 + 5.11455f * eta3 - 1.27198f * eta4 + 0.12746f * eta5;
}
WF_HD float FresnelMoment2(float eta) {
    float eta2 = eta * eta, eta3 = eta2 * eta, eta4 = eta3 * eta, eta5 = eta4 * eta;
    if (eta < 1) {
        return 0.27614f - 0.87350f * eta + 1.12077f * eta2 - 0.65095f * eta3 + 0.07883f * eta4 + 0.04860f * eta5;
    } else {
        float r_eta = 1 / eta, r_eta2 = r_eta * r_eta, r_eta3 = r_eta2 * r_eta;
        return -547.033f + 45.3087f * r_eta3 - 218.725f * r_eta2 + 458.843f * r_eta + 404.557f * eta - 189.519f * eta2 + 54.9327f * eta3 - 9.00603f * eta4 +
               0.63942f * eta5;
    }
}

// bxdfs.h:1073-1132
struct NormalizedFresnelBxDF {
    float eta;
    WF_HD int Flags() const { return BXDF_REFLECTION | BXDF_DIFFUSE; }
    WF_HD S4 f(V3 wo, V3 wi, int mode) const {
        if (!SameHemisphere(wo, wi)) return S4c(0.f);
        float c = 1 - 2 * FresnelMoment1(1 / eta);
        S4 f = S4c((1 - FrDielectric(CosTheta(wi), eta)) / (c * Pi));
        if (mode == MODE_RADIANCE) f = f * Sqr(eta);
        return f;
    }
    WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags) const {
        if (!(sampleFlags & REFLTRANS_REFLECTION)) return 0;
        return SameHemisphere(wo, wi) ? AbsCosTheta(wi) * InvPi : 0;
    }
    WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags) const {
        if (!(sampleFlags & REFLTRANS_REFLECTION)) return {};
        V3 wi = SampleCosineHemisphere(u);
        if (wo.z < 0) wi.z *= -1;
        return MakeSample(f(wo, wi, mode), wi, PDF(wo, wi, mode, sampleFlags), BXDF_DIFFUSE_REFLECTION);
    }
    WF_HD void Regularize() {}
};

// bssrdf.h:110-281.  `table` = the material's BSSRDFTable in the scene's float pool.
struct TabulatedBSSRDF {
    V3 po, wo;
    N3 ns;
    float eta;
    S4 sigma_t, rho;
    const float *table;

    WF_HD TabulatedBSSRDF() : table(nullptr) {}
    WF_HD TabulatedBSSRDF(V3 po, N3 ns, V3 wo, float eta, S4 sigma_a, S4 sigma_s, const float *table) : po(po), wo(wo), ns(ns), eta(eta), table(table) {
        sigma_t = sigma_a + sigma_s;
        rho = SafeDiv(sigma_s, sigma_t);
    }
    WF_HD S4 Sr(float r) const {
        BSSRDFTableView t(table);
        S4 Srv = S4c(0.f);
        for (int i = 0; i < 4; ++i) {
            float rOptical = r * sigma_t[i];
            int rhoOffset, radiusOffset;
            float rhoWeights[4], radiusWeights[4];
            if (!CatmullRomWeights(t.rhoSamples, BSSRDF_NRHO, rho[i], &rhoOffset, rhoWeights) ||
                !CatmullRomWeights(t.radiusSamples, BSSRDF_NRADIUS, rOptical, &radiusOffset, radiusWeights))
                continue;
            float sr = 0;
            for (int j = 0; j < 4; ++j)
                for (int k = 0; k < 4; ++k) {
                    float weight = rhoWeights[j] * radiusWeights[k];
                    if (weight != 0) sr += weight * t.EvalProfile(rhoOffset + j, radiusOffset + k);
                }
            if (rOptical != 0) sr /= 2 * Pi * rOptical;
            Srv[i] = sr;
        }
        Srv = Srv * (sigma_t * sigma_t);
        return ClampZero(Srv);
    }
    WF_HD S4 Sp(V3 pi) const { return Sr(Distance(po, pi)); }
    WF_HD bool SampleSr(float u, float *r) const {
        if (sigma_t[0] == 0) return false;
        BSSRDFTableView t(table);
        *r = SampleCatmullRom2D(t.rhoSamples, BSSRDF_NRHO, t.radiusSamples, BSSRDF_NRADIUS, t.profile, t.profileCDF, rho[0], u) / sigma_t[0];
        return true;
    }
    WF_HD S4 PDF_Sr(float r) const {
        BSSRDFTableView t(table);
        S4 pdf = S4c(0.f);
        for (int i = 0; i < 4; ++i) {
            float rOptical = r * sigma_t[i];
            int rhoOffset, radiusOffset;
            float rhoWeights[4], radiusWeights[4];
            if (!CatmullRomWeights(t.rhoSamples, BSSRDF_NRHO, rho[i], &rhoOffset, rhoWeights) ||
                !CatmullRomWeights(t.radiusSamples, BSSRDF_NRADIUS, rOptical, &radiusOffset, radiusWeights))
                continue;
            float sr = 0, rhoEff = 0;
            for (int j = 0; j < 4; ++j)
                if (rhoWeights[j] != 0) {
                    rhoEff += t.rhoEff[rhoOffset + j] * rhoWeights[j];
                    for (int k = 0; k < 4; ++k)
                        if (radiusWeights[k] != 0) sr += t.EvalProfile(rhoOffset + j, radiusOffset + k) * rhoWeights[j] * radiusWeights[k];
                }
            if (rOptical != 0) sr /= 2 * Pi * rOptical;
            pdf[i] = sr * Sqr(sigma_t[i]) / rhoEff;
        }
        return ClampZero(pdf);
    }
    // bssrdf.h:207-236: the probe segment pStart -> pTarget
    WF_HD bool SampleSp(float u1, V2 u2, V3 *pStart, V3 *pTarget) const {
        Frame f;
        V3 nsv = toV(ns);
        if (u1 < 0.25f) { f.x = nsv; CoordinateSystem(nsv, &f.y, &f.z); }        // Frame::FromX
        else if (u1 < 0.5f) { f.y = nsv; CoordinateSystem(nsv, &f.z, &f.x); }    // Frame::FromY
        else f = Frame::FromZ(nsv);
        float r;
        if (!SampleSr(u2.x, &r)) return false;
        float phi = 2 * Pi * u2.y;
        float r_max;
        if (!SampleSr(0.999f, &r_max) || r >= r_max) return false;
        float l = 2 * sqrt(Sqr(r_max) - Sqr(r));
        *pStart = po + r * (f.x * cos(phi) + f.y * sin(phi)) - l * f.z / 2;
        *pTarget = *pStart + l * f.z;
        return true;
    }
    WF_HD S4 PDF_Sp(V3 pi, N3 ni) const {
        V3 d = pi - po;
        Frame f = Frame::FromZ(toV(ns));
        V3 dLocal = f.ToLocal(d);
        N3 nLocal = f.ToLocal(ni);
        float rProj[3] = {sqrt(Sqr(dLocal.y) + Sqr(dLocal.z)), sqrt(Sqr(dLocal.z) + Sqr(dLocal.x)), sqrt(Sqr(dLocal.x) + Sqr(dLocal.y))};
        S4 pdf = S4c(0.f);
        float axisProb[3] = {.25f, .25f, .5f};
        float nl[3] = {nLocal.x, nLocal.y, nLocal.z};
        for (int axis = 0; axis < 3; ++axis) pdf = pdf + PDF_Sr(rProj[axis]) * abs(nl[axis]) * axisProb[axis];
        return pdf;
    }
};

// bssrdf.h:284-294
WF_HD void SubsurfaceFromDiffuse(const float *table, S4 rhoEff, S4 mfp, S4 *sigma_a, S4 *sigma_s) {
    BSSRDFTableView t(table);
    for (int c = 0; c < 4; ++c) {
        float rho = InvertCatmullRom(t.rhoSamples, t.rhoEff, BSSRDF_NRHO, rhoEff[c]);
        (*sigma_s)[c] = rho / mfp[c];
        (*sigma_a)[c] = (1 - rho) / mfp[c];
    }
}

// SubsurfaceMaterial::GetBxDF (materials.h:731-745): a dielectric interface with the material's scalar eta
WF_HD DielectricBxDF GetSubsurfaceBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    float urough = EvalFloatTexture(sv, m.tex[WF_MT_UROUGH], tc), vrough = EvalFloatTexture(sv, m.tex[WF_MT_VROUGH], tc);
    if (m.flags & WF_MATFLAG_REMAP_ROUGHNESS) {
        urough = TrowbridgeReitz::RoughnessToAlpha(urough);
        vrough = TrowbridgeReitz::RoughnessToAlpha(vrough);
    }
    return DielectricBxDF{m.sss_eta, TrowbridgeReitz(urough, vrough), sv.fatal};
}
// SubsurfaceMaterial::GetBSSRDF (materials.h:747-765); ctx = (p, ns, wo, uv) of GetBSSRDFAndProbeRayWorkItem::GetMaterialEvalContext
WF_HD TabulatedBSSRDF GetBSSRDF(const SceneView &sv, const wf_material &m, const Wavelengths &lambda, const TexCtx &tc, N3 ns, V3 wo) {
    const float *table = sv.tableData + m.sss_table;
    S4 sig_a, sig_s;
    if (m.flags & WF_MATFLAG_SSS_COEFFICIENTS) {
        sig_a = ClampZero(m.scale * EvalSpectrumTexture(sv, m.tex[WF_MT_SIGMA_A], lambda, tc));
        sig_s = ClampZero(m.scale * EvalSpectrumTexture(sv, m.tex[WF_MT_SIGMA_S], lambda, tc));
    } else {
        S4 mfree = ClampZero(m.scale * EvalSpectrumTexture(sv, m.tex[WF_MT_MFP], lambda, tc));
        S4 r = ClampS(EvalSpectrumTexture(sv, m.tex[WF_MT_REFLECTANCE], lambda, tc), 0.f, 1.f);
        SubsurfaceFromDiffuse(table, r, mfree, &sig_a, &sig_s);
    }
    return TabulatedBSSRDF(tc.p, ns, wo, m.sss_eta, sig_a, sig_s, table);
}

#ifndef __HIP_DEVICE_COMPILE__
// ---- host only: the table (bssrdf.cpp:26-128), computed once per material at scene load ------------------------
inline float BeamDiffusionMS(float sigma_s, float sigma_a, float g, float eta, float r) {
    const int nSamples = 100;
    float Ed = 0;
    float sigmap_s = sigma_s * (1 - g);
    float sigmap_t = sigma_a + sigmap_s;
    float rhop = sigmap_s / sigmap_t;
    float D_g = (2 * sigma_a + sigmap_s) / (3 * sigmap_t * sigmap_t);
    float sigma_tr = SafeSqrt(sigma_a / D_g);
    float fm1 = FresnelMoment1(eta), fm2 = FresnelMoment2(eta);
    float ze = -2 * D_g * (1 + 3 * fm2) / (1 - 2 * fm1);
    float cPhi = 0.25f * (1 - 2 * fm1), cE = 0.5f * (1 - 3 * fm2);
    for (int i = 0; i < nSamples; ++i) {
        float zr = SampleExponential((i + 0.5f) / nSamples, sigmap_t);
        float zv = -zr + 2 * ze;
        float dr = sqrt(Sqr(r) + Sqr(zr)), dv = sqrt(Sqr(r) + Sqr(zv));
        float phiD = Inv4Pi / D_g * (FastExp(-sigma_tr * dr) / dr - FastExp(-sigma_tr * dv) / dv);
        float EDn = Inv4Pi * (zr * (1 + sigma_tr * dr) * FastExp(-sigma_tr * dr) / (dr * dr * dr) - zv * (1 + sigma_tr * dv) * FastExp(-sigma_tr * dv) / (dv * dv * dv));
        float E = phiD * cPhi + EDn * cE;
        float kappa = 1 - FastExp(-2 * sigmap_t * (dr + zr));
        Ed += kappa * rhop * rhop * E;
    }
    return Ed / nSamples;
}
inline float BeamDiffusionSS(float sigma_s, float sigma_a, float g, float eta, float r) {
    float sigma_t = sigma_a + sigma_s, rho = sigma_s / sigma_t;
    float tCrit = r * SafeSqrt(Sqr(eta) - 1);
    float Ess = 0;
    const int nSamples = 100;
    for (int i = 0; i < nSamples; ++i) {
        float ti = tCrit + SampleExponential((i + 0.5f) / nSamples, sigma_t);
        float d = sqrt(Sqr(r) + Sqr(ti));
        float cosTheta_o = ti / d;
        Ess += rho * FastExp(-sigma_t * (d + tCrit)) / Sqr(d) * HenyeyGreenstein(cosTheta_o, g) * (1 - FrDielectric(-cosTheta_o, eta)) * abs(cosTheta_o);
    }
    return Ess / nSamples;
}
inline void ComputeBeamDiffusionBSSRDF(float g, float eta, float *table) {
    float *rhoSamples = table, *radiusSamples = table + BSSRDF_NRHO, *profile = radiusSamples + BSSRDF_NRADIUS;
    float *rhoEff = profile + BSSRDF_NRHO * BSSRDF_NRADIUS, *profileCDF = rhoEff + BSSRDF_NRHO;
    radiusSamples[0] = 0;
    radiusSamples[1] = 2.5e-3f;
    for (int i = 2; i < BSSRDF_NRADIUS; ++i) radiusSamples[i] = radiusSamples[i - 1] * 1.2f;
    for (int i = 0; i < BSSRDF_NRHO; ++i) rhoSamples[i] = (1 - FastExp(-8 * i / (float)(BSSRDF_NRHO - 1))) / (1 - FastExp(-8));
    for (int i = 0; i < BSSRDF_NRHO; ++i) {
        for (int j = 0; j < BSSRDF_NRADIUS; ++j) {
            float rho = rhoSamples[i], r = radiusSamples[j];
            profile[i * BSSRDF_NRADIUS + j] = 2 * Pi * r * (BeamDiffusionSS(rho, 1 - rho, g, eta, r) + BeamDiffusionMS(rho, 1 - rho, g, eta, r));
        }
        rhoEff[i] = IntegrateCatmullRom(radiusSamples, profile + i * BSSRDF_NRADIUS, BSSRDF_NRADIUS, profileCDF + i * BSSRDF_NRADIUS);
    }
}
#endif

}  // namespace wf
