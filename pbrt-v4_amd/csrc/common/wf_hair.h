// wf_hair.h — HairBxDF (bxdfs.h:921-1019, bxdfs.cpp:272-568) and HairMaterial::GetBxDF (materials.h:379-404), operation for operation.
#pragma once

#include "wf_bxdf.h"

namespace wf {

// Pow<n> (util/math.h:288-309): n2 = Pow<n/2>(v); n2 * n2 * Pow<n&1>(v)
template <int n>
WF_HD float PowN(float v) {
    if constexpr (n == 0) return 1.f;
    else if constexpr (n == 1) return v;
    else {
        float n2 = PowN<n / 2>(v);
        return n2 * n2 * PowN<(n & 1)>(v);
    }
}
// util/math.h:793-815
WF_HD float BesselI0(float x) {
    float val = 0;
    float x2i = 1;
    int64_t ifact = 1;
    int i4 = 1;
    for (int i = 0; i < 10; ++i) {
        if (i > 1) ifact *= i;
        val += x2i / (float)(i4 * (ifact * ifact));
        x2i *= x * x;
        i4 *= 4;
    }
    return val;
}
WF_HD float LogI0(float x) {
    if (x > 12) return x + 0.5f * (-log(2 * Pi) + log(1 / x) + 1 / (8 * x));
    return log(BesselI0(x));
}
// util/math.h:489-501, util/sampling.h:256-278
WF_HD float Logistic(float x, float s) {
    x = abs(x);
    return exp(-x / s) / (s * Sqr(1 + exp(-x / s)));
}
WF_HD float LogisticCDF(float x, float s) { return 1 / (1 + exp(-x / s)); }
WF_HD float TrimmedLogistic(float x, float s, float a, float b) { return Logistic(x, s) / (LogisticCDF(b, s) - LogisticCDF(a, s)); }
WF_HD float SampleTrimmedLogistic(float u, float s, float a, float b) {
    auto P = [&](float x) { return 1 / (1 + exp(-x / s)); };  // InvertLogisticSample
    u = Lerp(u, P(a), P(b));
    float x = -s * log(1 / u - 1);  // SampleLogistic
    return Clamp(x, a, b);
}

struct HairBxDF {
    static constexpr int pMax = 3;
    float h, eta;
    S4 sigma_a;
    float beta_m, beta_n;
    float v[pMax + 1];
    float s;
    float sin2kAlpha[pMax], cos2kAlpha[pMax];

    WF_HD HairBxDF() {}
    WF_HD HairBxDF(float h, float eta, const S4 &sigma_a, float beta_m, float beta_n, float alpha)
        : h(h), eta(eta), sigma_a(sigma_a), beta_m(beta_m), beta_n(beta_n) {
        v[0] = Sqr(0.726f * beta_m + 0.812f * Sqr(beta_m) + 3.7f * PowN<20>(beta_m));
        v[1] = (float)(.25 * (double)v[0]);
        v[2] = 4 * v[0];
        for (int p = 3; p <= pMax; ++p) v[p] = v[2];
        const float SqrtPiOver8 = 0.626657069f;
        s = SqrtPiOver8 * (0.265f * beta_n + 1.194f * Sqr(beta_n) + 5.372f * PowN<22>(beta_n));
        sin2kAlpha[0] = sin((Pi / 180) * alpha);  // Radians()
        cos2kAlpha[0] = SafeSqrt(1 - Sqr(sin2kAlpha[0]));
        for (int i = 1; i < pMax; ++i) {
            sin2kAlpha[i] = 2 * cos2kAlpha[i - 1] * sin2kAlpha[i - 1];
            cos2kAlpha[i] = Sqr(cos2kAlpha[i - 1]) - Sqr(sin2kAlpha[i - 1]);
        }
    }
    WF_HD int Flags() const { return BXDF_GLOSSY_REFLECTION; }
    WF_HD void Regularize() {}

    WF_HD static float Mp(float cosTheta_i, float cosTheta_o, float sinTheta_i, float sinTheta_o, float v) {
        float a = cosTheta_i * cosTheta_o / v, b = sinTheta_i * sinTheta_o / v;
        return ((double)v <= .1) ? (FastExp(LogI0(a) - b - 1 / v + 0.6931f + log(1 / (2 * v)))) : (FastExp(-b) * BesselI0(a)) / (sinh(1 / v) * 2 * v);
    }
    WF_HD static void Ap(float cosTheta_o, float eta, float h, const S4 &T, S4 ap[pMax + 1]) {
        float cosGamma_o = SafeSqrt(1 - Sqr(h));
        float cosTheta = cosTheta_o * cosGamma_o;
        float f = FrDielectric(cosTheta, eta);
        ap[0] = S4c(f);
        ap[1] = Sqr(1 - f) * T;
        for (int p = 2; p < pMax; ++p) ap[p] = ap[p - 1] * T * f;
        ap[pMax] = S4c(0.f);
        if (S4c(1.f) - T * f) ap[pMax] = ap[pMax - 1] * f * T / (S4c(1.f) - T * f);
    }
    WF_HD static float Phi(int p, float gamma_o, float gamma_t) { return 2 * p * gamma_t - 2 * gamma_o + p * Pi; }
    WF_HD static float Np(float phi, int p, float s, float gamma_o, float gamma_t) {
        float dphi = phi - Phi(p, gamma_o, gamma_t);
        while (dphi > Pi) dphi -= 2 * Pi;
        while (dphi < -Pi) dphi += 2 * Pi;
        return TrimmedLogistic(dphi, s, -Pi, Pi);
    }
    WF_HD void TiltedO(int p, float sinTheta_o, float cosTheta_o, float *sinThetap_o, float *cosThetap_o) const {
        if (p == 0) {
            *sinThetap_o = sinTheta_o * cos2kAlpha[1] - cosTheta_o * sin2kAlpha[1];
            *cosThetap_o = cosTheta_o * cos2kAlpha[1] + sinTheta_o * sin2kAlpha[1];
        } else if (p == 1) {
            *sinThetap_o = sinTheta_o * cos2kAlpha[0] + cosTheta_o * sin2kAlpha[0];
            *cosThetap_o = cosTheta_o * cos2kAlpha[0] - sinTheta_o * sin2kAlpha[0];
        } else if (p == 2) {
            *sinThetap_o = sinTheta_o * cos2kAlpha[2] + cosTheta_o * sin2kAlpha[2];
            *cosThetap_o = cosTheta_o * cos2kAlpha[2] - sinTheta_o * sin2kAlpha[2];
        } else {
            *sinThetap_o = sinTheta_o;
            *cosThetap_o = cosTheta_o;
        }
        *cosThetap_o = abs(*cosThetap_o);
    }
    WF_HD S4 Transmittance(float sinTheta_o, float cosTheta_o, float *gamma_t) const {
        float sinTheta_t = sinTheta_o / eta;
        float cosTheta_t = SafeSqrt(1 - Sqr(sinTheta_t));
        float etap = SafeSqrt(Sqr(eta) - Sqr(sinTheta_o)) / cosTheta_o;
        float sinGamma_t = h / etap;
        float cosGamma_t = SafeSqrt(1 - Sqr(sinGamma_t));
        *gamma_t = SafeASin(sinGamma_t);
        S4 e = -sigma_a * (2 * cosGamma_t / cosTheta_t);
        return S4{{exp(e[0]), exp(e[1]), exp(e[2]), exp(e[3])}};
    }
    WF_HD S4 f(V3 wo, V3 wi, int mode) const {
        float sinTheta_o = wo.x;
        float cosTheta_o = SafeSqrt(1 - Sqr(sinTheta_o));
        float phi_o = atan2(wo.z, wo.y);
        float gamma_o = SafeASin(h);
        float sinTheta_i = wi.x;
        float cosTheta_i = SafeSqrt(1 - Sqr(sinTheta_i));
        float phi_i = atan2(wi.z, wi.y);
        float gamma_t;
        S4 T = Transmittance(sinTheta_o, cosTheta_o, &gamma_t);
        float phi = phi_i - phi_o;
        S4 ap[pMax + 1];
        Ap(cosTheta_o, eta, h, T, ap);
        S4 fsum = S4c(0.f);
        for (int p = 0; p < pMax; ++p) {
            float sinThetap_o, cosThetap_o;
            TiltedO(p, sinTheta_o, cosTheta_o, &sinThetap_o, &cosThetap_o);
            fsum = fsum + Mp(cosTheta_i, cosThetap_o, sinTheta_i, sinThetap_o, v[p]) * ap[p] * Np(phi, p, s, gamma_o, gamma_t);
        }
        fsum = fsum + Mp(cosTheta_i, cosTheta_o, sinTheta_i, sinTheta_o, v[pMax]) * ap[pMax] / (2 * Pi);
        if (AbsCosTheta(wi) > 0) fsum = fsum / AbsCosTheta(wi);
        return fsum;
    }
    WF_HD void ApPDF(float cosTheta_o, float apPDF[pMax + 1]) const {
        float sinTheta_o = SafeSqrt(1 - Sqr(cosTheta_o));
        float gamma_t;
        S4 T = Transmittance(sinTheta_o, cosTheta_o, &gamma_t);
        S4 ap[pMax + 1];
        Ap(cosTheta_o, eta, h, T, ap);
        float sumY = 0;
        for (int i = 0; i <= pMax; ++i) sumY += ap[i].Average();
        for (int i = 0; i <= pMax; ++i) apPDF[i] = ap[i].Average() / sumY;
    }
    WF_HD float PdfSum(float phi, float sinTheta_i, float cosTheta_i, float sinTheta_o, float cosTheta_o, float gamma_o, float gamma_t,
                       const float apPDF[pMax + 1]) const {
        float pdf = 0;
        for (int p = 0; p < pMax; ++p) {
            float sinThetap_o, cosThetap_o;
            TiltedO(p, sinTheta_o, cosTheta_o, &sinThetap_o, &cosThetap_o);
            pdf += Mp(cosTheta_i, cosThetap_o, sinTheta_i, sinThetap_o, v[p]) * apPDF[p] * Np(phi, p, s, gamma_o, gamma_t);
        }
        pdf += Mp(cosTheta_i, cosTheta_o, sinTheta_i, sinTheta_o, v[pMax]) * apPDF[pMax] * (1 / (2 * Pi));
        return pdf;
    }
    WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags) const {
        float sinTheta_o = wo.x;
        float cosTheta_o = SafeSqrt(1 - Sqr(sinTheta_o));
        float phi_o = atan2(wo.z, wo.y);
        float gamma_o = SafeASin(h);
        float apPDF[pMax + 1];
        ApPDF(cosTheta_o, apPDF);
        // SampleDiscrete(apPDF, uc, nullptr, &uc)  (util/sampling.h:79-111)
        int p;
        {
            float sumWeights = 0;
            for (int i = 0; i <= pMax; ++i) sumWeights += apPDF[i];
            float up = uc * sumWeights;
            if (up == sumWeights) up = NextFloatDown(up);
            int offset = 0;
            float sum = 0;
            while (sum + apPDF[offset] <= up) sum += apPDF[offset++];
            uc = fmin((up - sum) / apPDF[offset], OneMinusEpsilon);
            p = offset;
        }
        float sinThetap_o, cosThetap_o;
        TiltedO(p, sinTheta_o, cosTheta_o, &sinThetap_o, &cosThetap_o);
        float cosTheta = 1 + v[p] * log(fmax(u.x, 1e-5f) + (1 - u.x) * FastExp(-2 / v[p]));
        float sinTheta = SafeSqrt(1 - Sqr(cosTheta));
        float cosPhi = cos(2 * Pi * u.y);
        float sinTheta_i = -cosTheta * sinThetap_o + sinTheta * cosPhi * cosThetap_o;
        float cosTheta_i = SafeSqrt(1 - Sqr(sinTheta_i));
        float etap = SafeSqrt(Sqr(eta) - Sqr(sinTheta_o)) / cosTheta_o;
        float sinGamma_t = h / etap;
        float gamma_t = SafeASin(sinGamma_t);
        float dphi;
        if (p < pMax) dphi = Phi(p, gamma_o, gamma_t) + SampleTrimmedLogistic(uc, s, -Pi, Pi);
        else dphi = 2 * Pi * uc;
        float phi_i = phi_o + dphi;
        V3 wi{sinTheta_i, cosTheta_i * cos(phi_i), cosTheta_i * sin(phi_i)};
        float pdf = PdfSum(dphi, sinTheta_i, cosTheta_i, sinTheta_o, cosTheta_o, gamma_o, gamma_t, apPDF);
        return MakeSample(f(wo, wi, mode), wi, pdf, Flags());
    }
    WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags) const {
        float sinTheta_o = wo.x;
        float cosTheta_o = SafeSqrt(1 - Sqr(sinTheta_o));
        float phi_o = atan2(wo.z, wo.y);
        float gamma_o = SafeASin(h);
        float sinTheta_i = wi.x;
        float cosTheta_i = SafeSqrt(1 - Sqr(sinTheta_i));
        float phi_i = atan2(wi.z, wi.y);
        float etap = SafeSqrt(eta * eta - Sqr(sinTheta_o)) / cosTheta_o;
        float sinGamma_t = h / etap;
        float gamma_t = SafeASin(sinGamma_t);
        float apPDF[pMax + 1];
        ApPDF(cosTheta_o, apPDF);
        return PdfSum(phi_i - phi_o, sinTheta_i, cosTheta_i, sinTheta_o, cosTheta_o, gamma_o, gamma_t, apPDF);
    }
};

// HairMaterial::GetBxDF (materials.h:379-404)
WF_HD HairBxDF GetHairBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    float bm = fmax(1e-2f, fmin(1.0f, EvalFloatTexture(sv, m.tex[WF_MT_HAIR_BETA_M], tc)));
    float bn = fmax(1e-2f, fmin(1.0f, EvalFloatTexture(sv, m.tex[WF_MT_HAIR_BETA_N], tc)));
    float a = EvalFloatTexture(sv, m.tex[WF_MT_HAIR_ALPHA], tc);
    float e = EvalFloatTexture(sv, m.tex[WF_MT_HAIR_ETA], tc);
    S4 sig_a;
    if (m.tex[WF_MT_SIGMA_A] >= 0) sig_a = ClampZero(EvalSpectrumTexture(sv, m.tex[WF_MT_SIGMA_A], lambda, tc));
    else if (m.tex[WF_MT_REFLECTANCE] >= 0) {
        // HairBxDF::SigmaAFromReflectance (bxdfs.cpp:559-568)
        S4 c = ClampS(EvalSpectrumTexture(sv, m.tex[WF_MT_REFLECTANCE], lambda, tc), 0.f, 1.f);
        for (int i = 0; i < 4; ++i)
            sig_a[i] = Sqr(log(c[i]) / (5.969f - 0.215f * bn + 2.532f * Sqr(bn) - 10.73f * PowN<3>(bn) + 5.574f * PowN<4>(bn) + 0.245f * PowN<5>(bn)));
    } else {
        // HairBxDF::SigmaAFromConcentration (bxdfs.cpp:548-557) -> RGBUnboundedSpectrum(sRGB, rgb).Sample(lambda)
        float ce = fmax(0.f, m.tex[WF_MT_HAIR_EUMELANIN] >= 0 ? EvalFloatTexture(sv, m.tex[WF_MT_HAIR_EUMELANIN], tc) : 0.f);
        float cp = fmax(0.f, m.tex[WF_MT_HAIR_PHEOMELANIN] >= 0 ? EvalFloatTexture(sv, m.tex[WF_MT_HAIR_PHEOMELANIN], tc) : 0.f);
        float rgb[3] = {ce * 0.419f + cp * 0.187f, ce * 0.697f + cp * 0.4f, ce * 1.37f + cp * 1.05f};
        float mx = fmax(fmax(rgb[0], rgb[1]), rgb[2]);
        float scale = 2 * mx;
        float in[3] = {0, 0, 0};
        if (scale) { in[0] = rgb[0] / scale; in[1] = rgb[1] / scale; in[2] = rgb[2] / scale; }
        float cf[3];
        RGBToSpectrumCoeffs(sv, in, cf);
        for (int i = 0; i < 4; ++i) sig_a[i] = scale * SigmoidPoly(lambda.lambda[i], cf[0], cf[1], cf[2]);
    }
    float h = -1 + 2 * tc.uv.y;
    // HairBxDF ctor (bxdfs.cpp:278-280): CHECK(h >= -1 && h <= 1), CHECK(beta_m / beta_n in [0, 1]) — a hair material on a surface whose
    // v leaves [0, 1] (a (u, v)-scaled quad) aborts the reference
    if (!(h >= -1 && h <= 1) || !(bm >= 0 && bm <= 1) || !(bn >= 0 && bn <= 1)) RaiseFatal(sv, WF_FATAL_CHECK_HAIR);
    return HairBxDF(h, e, sig_a, bm, bn, a);
}

}  // namespace wf
