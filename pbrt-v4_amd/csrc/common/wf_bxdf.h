// wf_bxdf.h — BxDFs, microfacet distribution, Fresnel, and the BSDF shading-frame wrapper.
// Restates bxdfs.h:28-430, bxdfs.cpp:77-258, util/scattering.h:18-215, bsdf.h:19-152 and the
// Material::GetBxDF bodies (materials.h:182-203,226-240,465-469,491-511,815-821) operation for operation.
#pragma once

#include "wf_scene.h"

namespace wf {

enum BxDFFlags {
    BXDF_UNSET = 0,
    BXDF_REFLECTION = 1 << 0,
    BXDF_TRANSMISSION = 1 << 1,
    BXDF_DIFFUSE = 1 << 2,
    BXDF_GLOSSY = 1 << 3,
    BXDF_SPECULAR = 1 << 4,
    BXDF_DIFFUSE_REFLECTION = BXDF_DIFFUSE | BXDF_REFLECTION,
    BXDF_DIFFUSE_TRANSMISSION = BXDF_DIFFUSE | BXDF_TRANSMISSION,
    BXDF_GLOSSY_REFLECTION = BXDF_GLOSSY | BXDF_REFLECTION,
    BXDF_GLOSSY_TRANSMISSION = BXDF_GLOSSY | BXDF_TRANSMISSION,
    BXDF_SPECULAR_REFLECTION = BXDF_SPECULAR | BXDF_REFLECTION,
    BXDF_SPECULAR_TRANSMISSION = BXDF_SPECULAR | BXDF_TRANSMISSION,
    BXDF_ALL = 31
};
enum { REFLTRANS_REFLECTION = 1, REFLTRANS_TRANSMISSION = 2, REFLTRANS_ALL = 3 };
enum { MODE_RADIANCE = 0, MODE_IMPORTANCE = 1 };

WF_HD bool IsReflective(int f) { return f & BXDF_REFLECTION; }
WF_HD bool IsTransmissive(int f) { return f & BXDF_TRANSMISSION; }
WF_HD bool IsDiffuse(int f) { return f & BXDF_DIFFUSE; }
WF_HD bool IsGlossy(int f) { return f & BXDF_GLOSSY; }
WF_HD bool IsSpecular(int f) { return f & BXDF_SPECULAR; }
WF_HD bool IsNonSpecular(int f) { return f & (BXDF_DIFFUSE | BXDF_GLOSSY); }

struct BSDFSample {
    S4 f;
    V3 wi;
    float pdf = 0;
    int flags = 0;
    float eta = 1;
    bool pdfIsProportional = false;
    bool valid = false;
    WF_HD bool IsReflection() const { return IsReflective(flags); }
    WF_HD bool IsTransmission() const { return IsTransmissive(flags); }
    WF_HD bool IsSpecularS() const { return IsSpecular(flags); }
};
WF_HD BSDFSample MakeSample(S4 f, V3 wi, float pdf, int flags, float eta = 1, bool prop = false) {
    BSDFSample s;
    s.f = f; s.wi = wi; s.pdf = pdf; s.flags = flags; s.eta = eta; s.pdfIsProportional = prop; s.valid = true;
    return s;
}

// spherical geometry helpers (util/vecmath.h:1683-1725)
WF_HD float CosTheta(V3 w) { return w.z; }
WF_HD float Cos2Theta(V3 w) { return Sqr(w.z); }
WF_HD float AbsCosTheta(V3 w) { return abs(w.z); }
WF_HD float Sin2Theta(V3 w) { return fmax(0.f, 1 - Cos2Theta(w)); }
WF_HD float SinTheta(V3 w) { return sqrt(Sin2Theta(w)); }
WF_HD float Tan2Theta(V3 w) { return Sin2Theta(w) / Cos2Theta(w); }
WF_HD float CosPhi(V3 w) { float s = SinTheta(w); return (s == 0) ? 1 : Clamp(w.x / s, -1.f, 1.f); }
WF_HD float SinPhi(V3 w) { float s = SinTheta(w); return (s == 0) ? 0 : Clamp(w.y / s, -1.f, 1.f); }
WF_HD bool SameHemisphere(V3 w, V3 wp) { return w.z * wp.z > 0; }

// util/scattering.h:18-46
WF_HD V3 Reflect(V3 wo, V3 n) { return -wo + 2 * Dot(wo, n) * n; }
WF_HD bool Refract(V3 wi, N3 n, float eta, float *etap, V3 *wt) {
    float cosTheta_i = Dot(n, wi);
    if (cosTheta_i < 0) {
        eta = 1 / eta;
        cosTheta_i = -cosTheta_i;
        n = -n;
    }
    float sin2Theta_i = fmax(0.f, 1 - Sqr(cosTheta_i));
    float sin2Theta_t = sin2Theta_i / Sqr(eta);
    if (sin2Theta_t >= 1) return false;
    float cosTheta_t = sqrt(1 - sin2Theta_t);
    *wt = -wi / eta + (cosTheta_i / eta - cosTheta_t) * toV(n);
    if (etap) *etap = eta;
    return true;
}
// util/scattering.h:61-81
WF_HD float FrDielectric(float cosTheta_i, float eta) {
    cosTheta_i = Clamp(cosTheta_i, -1.f, 1.f);
    if (cosTheta_i < 0) {
        eta = 1 / eta;
        cosTheta_i = -cosTheta_i;
    }
    float sin2Theta_i = 1 - Sqr(cosTheta_i);
    float sin2Theta_t = sin2Theta_i / Sqr(eta);
    if (sin2Theta_t >= 1) return 1.f;
    float cosTheta_t = SafeSqrt(1 - sin2Theta_t);
    float r_parl = (eta * cosTheta_i - cosTheta_t) / (eta * cosTheta_i + cosTheta_t);
    float r_perp = (cosTheta_i - eta * cosTheta_t) / (cosTheta_i + eta * cosTheta_t);
    return (Sqr(r_parl) + Sqr(r_perp)) / 2;
}
// pstd::complex<float> (util/pstd.h:1066-1229), only what FrComplex needs
struct Cx { float re, im; };
WF_HD Cx cx(float re, float im = 0) { return Cx{re, im}; }
WF_HD Cx operator+(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
WF_HD Cx operator-(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
WF_HD Cx operator*(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
WF_HD Cx operator/(Cx a, Cx z) {
    float scale = 1 / (z.re * z.re + z.im * z.im);
    return {scale * (a.re * z.re + a.im * z.im), scale * (a.im * z.re - a.re * z.im)};
}
WF_HD float cnorm(Cx z) { return z.re * z.re + z.im * z.im; }
WF_HD Cx csqrt(Cx z) {
    float n = sqrt(cnorm(z)), t1 = sqrt(.5f * (n + abs(z.re))), t2 = .5f * z.im / t1;
    if (n == 0) return cx(0);
    if (z.re >= 0) return {t1, t2};
    return {abs(t2), copysign(t1, z.im)};
}
// util/scattering.h:83-103
WF_HD float FrComplex1(float cosTheta_i, Cx eta) {
    cosTheta_i = Clamp(cosTheta_i, 0.f, 1.f);
    float sin2Theta_i = 1 - Sqr(cosTheta_i);
    Cx sin2Theta_t = cx(sin2Theta_i) / (eta * eta);
    Cx cosTheta_t = csqrt(cx(1) - sin2Theta_t);
    Cx r_parl = (eta * cx(cosTheta_i) - cosTheta_t) / (eta * cx(cosTheta_i) + cosTheta_t);
    Cx r_perp = (cx(cosTheta_i) - eta * cosTheta_t) / (cx(cosTheta_i) + eta * cosTheta_t);
    return (cnorm(r_parl) + cnorm(r_perp)) / 2;
}
WF_HD S4 FrComplex(float cosTheta_i, S4 eta, S4 k) {
    S4 r;
    for (int i = 0; i < 4; ++i) r[i] = FrComplex1(cosTheta_i, Cx{eta[i], k[i]});
    return r;
}

WF_HD V2 SampleUniformDiskPolar(V2 u) {
    float r = sqrt(u.x);
    float theta = 2 * Pi * u.y;
    return {r * cos(theta), r * sin(theta)};
}

// TrowbridgeReitzDistribution, util/scattering.h:110-210
struct TrowbridgeReitz {
    float alpha_x, alpha_y;
    WF_HD TrowbridgeReitz() : alpha_x(0), alpha_y(0) {}
    WF_HD TrowbridgeReitz(float ax, float ay) : alpha_x(ax), alpha_y(ay) {
        if (!EffectivelySmooth()) {
            alpha_x = fmax(alpha_x, 1e-4f);
            alpha_y = fmax(alpha_y, 1e-4f);
        }
    }
    WF_HD bool EffectivelySmooth() const { return fmax(alpha_x, alpha_y) < 1e-3f; }
    WF_HD float D(V3 wm) const {
        float tan2Theta = Tan2Theta(wm);
        if (IsInf(tan2Theta)) return 0;
        float cos4Theta = Sqr(Cos2Theta(wm));
        if (cos4Theta < 1e-16f) return 0;
        float e = tan2Theta * (Sqr(CosPhi(wm) / alpha_x) + Sqr(SinPhi(wm) / alpha_y));
        return 1 / (Pi * alpha_x * alpha_y * cos4Theta * Sqr(1 + e));
    }
    WF_HD float Lambda(V3 w) const {
        float tan2Theta = Tan2Theta(w);
        if (IsInf(tan2Theta)) return 0;
        float alpha2 = Sqr(CosPhi(w) * alpha_x) + Sqr(SinPhi(w) * alpha_y);
        return (sqrt(1 + alpha2 * tan2Theta) - 1) / 2;
    }
    WF_HD float G1(V3 w) const { return 1 / (1 + Lambda(w)); }
    WF_HD float G(V3 wo, V3 wi) const { return 1 / (1 + Lambda(wo) + Lambda(wi)); }
    WF_HD float D(V3 w, V3 wm) const { return G1(w) / AbsCosTheta(w) * D(wm) * AbsDot(w, wm); }
    WF_HD float PDF(V3 w, V3 wm) const { return D(w, wm); }
    WF_HD V3 Sample_wm(V3 w, V2 u) const {
        V3 wh = Normalize(V3{alpha_x * w.x, alpha_y * w.y, w.z});
        if (wh.z < 0) wh = -wh;
        V3 T1 = (wh.z < 0.99999f) ? Normalize(Cross(V3{0, 0, 1}, wh)) : V3{1, 0, 0};
        V3 T2 = Cross(wh, T1);
        V2 p = SampleUniformDiskPolar(u);
        float h = sqrt(1 - Sqr(p.x));
        p.y = Lerp((1 + wh.z) / 2, h, p.y);
        float pz = sqrt(fmax(0.f, 1 - (Sqr(p.x) + Sqr(p.y))));
        V3 nh = p.x * T1 + p.y * T2 + pz * wh;
        return Normalize(V3{alpha_x * nh.x, alpha_y * nh.y, fmax(1e-6f, nh.z)});
    }
    WF_HD static float RoughnessToAlpha(float roughness) { return sqrt(roughness); }
    WF_HD void Regularize() {
        if (alpha_x < 0.3f) alpha_x = Clamp(2 * alpha_x, 0.1f, 0.3f);
        if (alpha_y < 0.3f) alpha_y = Clamp(2 * alpha_y, 0.1f, 0.3f);
    }
};

// ---------------------------------------------------------------------------------------------
// DiffuseBxDF, bxdfs.h:28-82
struct DiffuseBxDF {
    S4 R;
    WF_HD S4 f(V3 wo, V3 wi, int mode) const {
        if (!SameHemisphere(wo, wi)) return S4c(0.f);
        return R * InvPi;
    }
    WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags = REFLTRANS_ALL) const {
        if (!(sampleFlags & REFLTRANS_REFLECTION)) return {};
        V3 wi = SampleCosineHemisphere(u);
        if (wo.z < 0) wi.z *= -1;
        float pdf = CosineHemispherePDF(AbsCosTheta(wi));
        return MakeSample(R * InvPi, wi, pdf, BXDF_DIFFUSE_REFLECTION);
    }
    WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags = REFLTRANS_ALL) const {
        if (!(sampleFlags & REFLTRANS_REFLECTION) || !SameHemisphere(wo, wi)) return 0;
        return CosineHemispherePDF(AbsCosTheta(wi));
    }
    WF_HD void Regularize() {}
    WF_HD int Flags() const { return R ? BXDF_DIFFUSE_REFLECTION : BXDF_UNSET; }
};

// DiffuseTransmissionBxDF, bxdfs.h:84-160
struct DiffuseTransmissionBxDF {
    S4 R, T;
    WF_HD S4 f(V3 wo, V3 wi, int mode) const { return SameHemisphere(wo, wi) ? (R * InvPi) : (T * InvPi); }
    WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags = REFLTRANS_ALL) const {
        float pr = R.MaxComponentValue(), pt = T.MaxComponentValue();
        if (!(sampleFlags & REFLTRANS_REFLECTION)) pr = 0;
        if (!(sampleFlags & REFLTRANS_TRANSMISSION)) pt = 0;
        if (pr == 0 && pt == 0) return {};
        if (uc < pr / (pr + pt)) {
            V3 wi = SampleCosineHemisphere(u);
            if (wo.z < 0) wi.z *= -1;
            float pdf = CosineHemispherePDF(AbsCosTheta(wi)) * pr / (pr + pt);
            return MakeSample(f(wo, wi, mode), wi, pdf, BXDF_DIFFUSE_REFLECTION);
        } else {
            V3 wi = SampleCosineHemisphere(u);
            if (wo.z > 0) wi.z *= -1;
            float pdf = CosineHemispherePDF(AbsCosTheta(wi)) * pt / (pr + pt);
            return MakeSample(f(wo, wi, mode), wi, pdf, BXDF_DIFFUSE_TRANSMISSION);
        }
    }
    WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags = REFLTRANS_ALL) const {
        float pr = R.MaxComponentValue(), pt = T.MaxComponentValue();
        if (!(sampleFlags & REFLTRANS_REFLECTION)) pr = 0;
        if (!(sampleFlags & REFLTRANS_TRANSMISSION)) pt = 0;
        if (pr == 0 && pt == 0) return 0;
        if (SameHemisphere(wo, wi)) return pr / (pr + pt) * CosineHemispherePDF(AbsCosTheta(wi));
        else return pt / (pr + pt) * CosineHemispherePDF(AbsCosTheta(wi));
    }
    WF_HD void Regularize() {}
    WF_HD int Flags() const { return (R ? BXDF_DIFFUSE_REFLECTION : BXDF_UNSET) | (T ? BXDF_DIFFUSE_TRANSMISSION : BXDF_UNSET); }
};

// DielectricBxDF, bxdfs.h:162-201, bxdfs.cpp:77-258
struct DielectricBxDF {
    float eta;
    TrowbridgeReitz mfDistrib;
    int32_t *fatal = nullptr;   // SceneView::fatal (wf_scene.h) where a material kernel built the BxDF: the reference's CHECK below reports there
    WF_HD int Flags() const {
        int flags = (eta == 1) ? BXDF_TRANSMISSION : (BXDF_REFLECTION | BXDF_TRANSMISSION);
        return flags | (mfDistrib.EffectivelySmooth() ? BXDF_SPECULAR : BXDF_GLOSSY);
    }
    WF_HD void Regularize() { mfDistrib.Regularize(); }
    WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags = REFLTRANS_ALL) const {
        if (eta == 1 || mfDistrib.EffectivelySmooth()) {
            float R = FrDielectric(CosTheta(wo), eta), T = 1 - R;
            float pr = R, pt = T;
            if (!(sampleFlags & REFLTRANS_REFLECTION)) pr = 0;
            if (!(sampleFlags & REFLTRANS_TRANSMISSION)) pt = 0;
            if (pr == 0 && pt == 0) return {};
            if (uc < pr / (pr + pt)) {
                V3 wi{-wo.x, -wo.y, wo.z};
                S4 fr = S4c(R / AbsCosTheta(wi));
                return MakeSample(fr, wi, pr / (pr + pt), BXDF_SPECULAR_REFLECTION);
            } else {
                V3 wi;
                float etap;
                bool valid = Refract(wo, N3{0, 0, 1}, eta, &etap, &wi);
                if (!valid) return {};
                S4 ft = S4c(T / AbsCosTheta(wi));
                if (mode == MODE_RADIANCE) ft = ft / Sqr(etap);
                return MakeSample(ft, wi, pt / (pr + pt), BXDF_SPECULAR_TRANSMISSION, etap);
            }
        } else {
            V3 wm = mfDistrib.Sample_wm(wo, u);
            float R = FrDielectric(Dot(wo, wm), eta);
            float T = 1 - R;
            float pr = R, pt = T;
            if (!(sampleFlags & REFLTRANS_REFLECTION)) pr = 0;
            if (!(sampleFlags & REFLTRANS_TRANSMISSION)) pt = 0;
            if (pr == 0 && pt == 0) return {};
            float pdf;
            if (uc < pr / (pr + pt)) {
                V3 wi = Reflect(wo, wm);
                if (!SameHemisphere(wo, wi)) return {};
                pdf = mfDistrib.PDF(wo, wm) / (4 * AbsDot(wo, wm)) * pr / (pr + pt);
                S4 f = S4c(mfDistrib.D(wm) * mfDistrib.G(wo, wi) * R / (4 * CosTheta(wi) * CosTheta(wo)));
                return MakeSample(f, wi, pdf, BXDF_GLOSSY_REFLECTION);
            } else {
                float etap;
                V3 wi{0, 0, 0};
                bool tir = !Refract(wo, toN(wm), eta, &etap, &wi);
                if (tir) return {};  // the reference tests wi of a failed Refract too; it is unset there
                if (SameHemisphere(wo, wi) || wi.z == 0) return {};
                float denom = Sqr(Dot(wi, wm) + Dot(wo, wm) / etap);
                float dwm_dwi = AbsDot(wi, wm) / denom;
                pdf = mfDistrib.PDF(wo, wm) * dwm_dwi * pt / (pr + pt);
                // CHECK(!IsNaN(pdf)), bxdfs.cpp:158 (a negative roughness gets here); raised as RaiseFatal does (wf_scene.h)
#if defined(__HIP_DEVICE_COMPILE__)
                if (pdf != pdf) atomicOr(fatal, (int)WF_FATAL_CHECK_NAN_PDF);
#else
                if (pdf != pdf && fatal) *fatal |= WF_FATAL_CHECK_NAN_PDF;
#endif
                S4 ft = S4c(T * mfDistrib.D(wm) * mfDistrib.G(wo, wi) *
                            abs(Dot(wi, wm) * Dot(wo, wm) / (CosTheta(wi) * CosTheta(wo) * denom)));
                if (mode == MODE_RADIANCE) ft = ft / Sqr(etap);
                return MakeSample(ft, wi, pdf, BXDF_GLOSSY_TRANSMISSION, etap);
            }
        }
    }
    WF_HD S4 f(V3 wo, V3 wi, int mode) const {
        if (eta == 1 || mfDistrib.EffectivelySmooth()) return S4c(0.f);
        float cosTheta_o = CosTheta(wo), cosTheta_i = CosTheta(wi);
        bool reflect = cosTheta_i * cosTheta_o > 0;
        float etap = 1;
        if (!reflect) etap = cosTheta_o > 0 ? eta : (1 / eta);
        V3 wm = wi * etap + wo;
        if (cosTheta_i == 0 || cosTheta_o == 0 || LengthSquared(wm) == 0) return S4c(0.f);
        wm = FaceForward(Normalize(wm), N3{0, 0, 1});
        if (Dot(wm, wi) * cosTheta_i < 0 || Dot(wm, wo) * cosTheta_o < 0) return S4c(0.f);
        float F = FrDielectric(Dot(wo, wm), eta);
        if (reflect) {
            return S4c(mfDistrib.D(wm) * mfDistrib.G(wo, wi) * F / abs(4 * cosTheta_i * cosTheta_o));
        } else {
            float denom = Sqr(Dot(wi, wm) + Dot(wo, wm) / etap) * cosTheta_i * cosTheta_o;
            float ft = mfDistrib.D(wm) * (1 - F) * mfDistrib.G(wo, wi) * abs(Dot(wi, wm) * Dot(wo, wm) / denom);
            if (mode == MODE_RADIANCE) ft /= Sqr(etap);
            return S4c(ft);
        }
    }
    WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags = REFLTRANS_ALL) const {
        if (eta == 1 || mfDistrib.EffectivelySmooth()) return 0;
        float cosTheta_o = CosTheta(wo), cosTheta_i = CosTheta(wi);
        bool reflect = cosTheta_i * cosTheta_o > 0;
        float etap = 1;
        if (!reflect) etap = cosTheta_o > 0 ? eta : (1 / eta);
        V3 wm = wi * etap + wo;
        if (cosTheta_i == 0 || cosTheta_o == 0 || LengthSquared(wm) == 0) return 0;
        wm = FaceForward(Normalize(wm), N3{0, 0, 1});
        if (Dot(wm, wi) * cosTheta_i < 0 || Dot(wm, wo) * cosTheta_o < 0) return 0;
        float R = FrDielectric(Dot(wo, wm), eta);
        float T = 1 - R;
        float pr = R, pt = T;
        if (!(sampleFlags & REFLTRANS_REFLECTION)) pr = 0;
        if (!(sampleFlags & REFLTRANS_TRANSMISSION)) pt = 0;
        if (pr == 0 && pt == 0) return 0;
        float pdf;
        if (reflect) {
            pdf = mfDistrib.PDF(wo, wm) / (4 * AbsDot(wo, wm)) * pr / (pr + pt);
        } else {
            float denom = Sqr(Dot(wi, wm) + Dot(wo, wm) / etap);
            float dwm_dwi = AbsDot(wi, wm) / denom;
            pdf = mfDistrib.PDF(wo, wm) * dwm_dwi * pt / (pr + pt);
        }
        return pdf;
    }
};

// ThinDielectricBxDF, bxdfs.h:203-275
struct ThinDielectricBxDF {
    float eta;
    WF_HD S4 f(V3 wo, V3 wi, int mode) const { return S4c(0.f); }
    WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags = REFLTRANS_ALL) const {
        float R = FrDielectric(AbsCosTheta(wo), eta), T = 1 - R;
        if (R < 1) {
            R += Sqr(T) * R / (1 - Sqr(R));
            T = 1 - R;
        }
        float pr = R, pt = T;
        if (!(sampleFlags & REFLTRANS_REFLECTION)) pr = 0;
        if (!(sampleFlags & REFLTRANS_TRANSMISSION)) pt = 0;
        if (pr == 0 && pt == 0) return {};
        if (uc < pr / (pr + pt)) {
            V3 wi{-wo.x, -wo.y, wo.z};
            S4 fr = S4c(R / AbsCosTheta(wi));
            return MakeSample(fr, wi, pr / (pr + pt), BXDF_SPECULAR_REFLECTION);
        } else {
            V3 wi = -wo;
            S4 ft = S4c(T / AbsCosTheta(wi));
            return MakeSample(ft, wi, pt / (pr + pt), BXDF_SPECULAR_TRANSMISSION);
        }
    }
    WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags = REFLTRANS_ALL) const { return 0; }
    WF_HD void Regularize() {}
    WF_HD int Flags() const { return BXDF_REFLECTION | BXDF_TRANSMISSION | BXDF_SPECULAR; }
};

// ConductorBxDF, bxdfs.h:277-384
struct ConductorBxDF {
    TrowbridgeReitz mfDistrib;
    S4 eta, k;
    WF_HD int Flags() const { return mfDistrib.EffectivelySmooth() ? BXDF_SPECULAR_REFLECTION : BXDF_GLOSSY_REFLECTION; }
    WF_HD void Regularize() { mfDistrib.Regularize(); }
    WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags = REFLTRANS_ALL) const {
        if (!(sampleFlags & REFLTRANS_REFLECTION)) return {};
        if (mfDistrib.EffectivelySmooth()) {
            V3 wi{-wo.x, -wo.y, wo.z};
            S4 f = FrComplex(AbsCosTheta(wi), eta, k) / AbsCosTheta(wi);
            return MakeSample(f, wi, 1, BXDF_SPECULAR_REFLECTION);
        }
        if (wo.z == 0) return {};
        V3 wm = mfDistrib.Sample_wm(wo, u);
        V3 wi = Reflect(wo, wm);
        if (!SameHemisphere(wo, wi)) return {};
        float pdf = mfDistrib.PDF(wo, wm) / (4 * AbsDot(wo, wm));
        float cosTheta_o = AbsCosTheta(wo), cosTheta_i = AbsCosTheta(wi);
        if (cosTheta_i == 0 || cosTheta_o == 0) return {};
        S4 F = FrComplex(AbsDot(wo, wm), eta, k);
        S4 f = mfDistrib.D(wm) * F * mfDistrib.G(wo, wi) / (4 * cosTheta_i * cosTheta_o);
        return MakeSample(f, wi, pdf, BXDF_GLOSSY_REFLECTION);
    }
    WF_HD S4 f(V3 wo, V3 wi, int mode) const {
        if (!SameHemisphere(wo, wi)) return S4c(0.f);
        if (mfDistrib.EffectivelySmooth()) return S4c(0.f);
        float cosTheta_o = AbsCosTheta(wo), cosTheta_i = AbsCosTheta(wi);
        if (cosTheta_i == 0 || cosTheta_o == 0) return S4c(0.f);
        V3 wm = wi + wo;
        if (LengthSquared(wm) == 0) return S4c(0.f);
        wm = Normalize(wm);
        S4 F = FrComplex(AbsDot(wo, wm), eta, k);
        return mfDistrib.D(wm) * F * mfDistrib.G(wo, wi) / (4 * cosTheta_i * cosTheta_o);
    }
    WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags = REFLTRANS_ALL) const {
        if (!(sampleFlags & REFLTRANS_REFLECTION)) return 0;
        if (!SameHemisphere(wo, wi)) return 0;
        if (mfDistrib.EffectivelySmooth()) return 0;
        V3 wm = wo + wi;
        if (LengthSquared(wm) == 0) return 0;
        wm = FaceForward(Normalize(wm), N3{0, 0, 1});
        return mfDistrib.PDF(wo, wm) / (4 * AbsDot(wo, wm));
    }
};

// ---------------------------------------------------------------------------------------------
// Henyey-Greenstein phase function (util/scattering.h:48-58, util/sampling.cpp:348-374, media.h:43-70)
WF_HD float ClampG(float g) {
    // Clamp(g, -.99, .99) with double bounds (comparison in double, result converted back)
    if ((double)g < -.99) return (float)-.99;
    if ((double)g > .99) return (float).99;
    return g;
}
WF_HD float HenyeyGreenstein(float cosTheta, float g) {
    g = ClampG(g);
    float denom = 1 + Sqr(g) + 2 * g * cosTheta;
    return Inv4Pi * (1 - Sqr(g)) / (denom * SafeSqrt(denom));
}
WF_HD V3 SampleHenyeyGreenstein(V3 wo, float g, V2 u, float *pdf) {
    g = ClampG(g);
    float cosTheta;
    if (abs(g) < 1e-3f) cosTheta = 1 - 2 * u.x;
    else cosTheta = -1 / (2 * g) * (1 + Sqr(g) - Sqr((1 - Sqr(g)) / (1 + g - 2 * g * u.x)));
    float sinTheta = SafeSqrt(1 - Sqr(cosTheta));
    float phi = 2 * Pi * u.y;
    Frame wFrame = Frame::FromZ(wo);
    V3 wi = wFrame.FromLocal(SphericalDirection(sinTheta, cosTheta, phi));
    if (pdf) *pdf = HenyeyGreenstein(cosTheta, g);
    return wi;
}
WF_HD float SampleExponential(float u, float a) { return -log(1 - u) / a; }
WF_HD float PowerHeuristic(int nf, float fPdf, int ng, float gPdf) {
    float f = nf * fPdf, g = ng * gPdf;
    if (IsInf(Sqr(f))) return 1;
    return Sqr(f) / (Sqr(f) + Sqr(g));
}
// Hash(int, Vector3f) / Hash(Vector3f) / Hash(Float, Point2f): util/hash.h:100-107 packs the arguments back to back
WF_HD uint64_t HashIntV3(int a, V3 v) {
    uint32_t w[4] = {(uint32_t)a, FloatToBits(v.x), FloatToBits(v.y), FloatToBits(v.z)};
    return HashWords(w, 4);
}
WF_HD uint64_t HashF3(float a, V2 u) {
    uint32_t w[3] = {FloatToBits(a), FloatToBits(u.x), FloatToBits(u.y)};
    return HashWords(w, 3);
}

// LayeredBxDF<Top, Bottom, twoSided>, bxdfs.h:432-905
template <typename TopBxDF, typename BottomBxDF, bool twoSided>
struct LayeredBxDF {
    TopBxDF top;
    BottomBxDF bottom;
    float thickness, g;
    S4 albedo;
    int maxDepth, nSamples;
    int seed;  // GetOptions().seed

    // TopOrBottomBxDF (bxdfs.h:386-429) as a flag
    struct IF {
        const LayeredBxDF *l;
        bool isTop;
        WF_HD S4 f(V3 wo, V3 wi, int mode) const { return isTop ? l->top.f(wo, wi, mode) : l->bottom.f(wo, wi, mode); }
        WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags = REFLTRANS_ALL) const {
            return isTop ? l->top.Sample_f(wo, uc, u, mode, sampleFlags) : l->bottom.Sample_f(wo, uc, u, mode, sampleFlags);
        }
        WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags = REFLTRANS_ALL) const {
            return isTop ? l->top.PDF(wo, wi, mode, sampleFlags) : l->bottom.PDF(wo, wi, mode, sampleFlags);
        }
        WF_HD int Flags() const { return isTop ? l->top.Flags() : l->bottom.Flags(); }
    };
    WF_HD static bool Bad(const BSDFSample &b) { return !b.valid || !b.f || b.pdf == 0 || b.wi.z == 0; }
    WF_HD static float Tr(float dz, V3 w) {
        if (abs(dz) <= 1.17549435e-38f) return 1;  // numeric_limits<float>::min()
        return FastExp(-abs(dz / w.z));
    }
    WF_HD void Regularize() { top.Regularize(); bottom.Regularize(); }
    WF_HD int Flags() const {
        int topFlags = top.Flags(), bottomFlags = bottom.Flags();
        int flags = BXDF_REFLECTION;
        if (IsSpecular(topFlags)) flags |= BXDF_SPECULAR;
        if (IsDiffuse(topFlags) || IsDiffuse(bottomFlags) || albedo) flags |= BXDF_DIFFUSE;
        else if (IsGlossy(topFlags) || IsGlossy(bottomFlags)) flags |= BXDF_GLOSSY;
        if (IsTransmissive(topFlags) && IsTransmissive(bottomFlags)) flags |= BXDF_TRANSMISSION;
        return flags;
    }
    WF_HD S4 f(V3 wo, V3 wi, int mode) const {
        S4 f = S4c(0.f);
        if (twoSided && wo.z < 0) { wo = -wo; wi = -wi; }
        bool enteredTop = twoSided || wo.z > 0;
        IF enterInterface{this, enteredTop};
        bool exitIsBottom = SameHemisphere(wo, wi) ^ enteredTop;
        IF exitInterface{this, !exitIsBottom}, nonExitInterface{this, exitIsBottom};
        float exitZ = exitIsBottom ? 0 : thickness;
        if (SameHemisphere(wo, wi)) f = nSamples * enterInterface.f(wo, wi, mode);
        RNG rng(HashIntV3(seed, wo), Hash3f(wi));
        auto r = [&rng]() { return fmin(rng.UniformFloat(), OneMinusEpsilon); };
        for (int s = 0; s < nSamples; ++s) {
            // NOTE on sample order: the reference writes Point2f(r(), r()) / f(x, r(), {r(), r()}) — function
            // arguments, whose evaluation order C++ leaves unspecified.  The oracle is the reference as g++
            // compiles it: arguments right to left (braced lists left to right inside).  Restated explicitly.
            float uc = r();
            float uy = r(), ux = r();
            BSDFSample wos = enterInterface.Sample_f(wo, uc, V2{ux, uy}, mode, REFLTRANS_TRANSMISSION);
            if (Bad(wos)) continue;
            uc = r();
            uy = r(); ux = r();
            BSDFSample wis = exitInterface.Sample_f(wi, uc, V2{ux, uy}, !mode, REFLTRANS_TRANSMISSION);
            if (Bad(wis)) continue;
            S4 beta = wos.f * AbsCosTheta(wos.wi) / wos.pdf;
            float z = enteredTop ? thickness : 0;
            V3 w = wos.wi;
            for (int depth = 0; depth < maxDepth; ++depth) {
                if (depth > 3 && beta.MaxComponentValue() < 0.25f) {
                    float q = fmax(0.f, 1 - beta.MaxComponentValue());
                    if (r() < q) break;
                    beta = beta / (1 - q);
                }
                if (!albedo) {
                    z = (z == thickness) ? 0 : thickness;
                    beta = beta * Tr(thickness, w);
                } else {
                    float sigma_t = 1;
                    float dz = SampleExponential(r(), sigma_t / abs(w.z));
                    float zp = w.z > 0 ? (z + dz) : (z - dz);
                    if (z == zp) continue;
                    if (0 < zp && zp < thickness) {
                        float wt = 1;
                        if (!IsSpecular(exitInterface.Flags())) wt = PowerHeuristic(1, wis.pdf, 1, HenyeyGreenstein(Dot(-w, -wis.wi), g));
                        f = f + beta * albedo * HenyeyGreenstein(Dot(-w, -wis.wi), g) * wt * Tr(zp - exitZ, wis.wi) * wis.f / wis.pdf;
                        float u0 = r(), u1 = r();
                        float ppdf;
                        V3 pwi = SampleHenyeyGreenstein(-w, g, V2{u0, u1}, &ppdf);
                        if (ppdf == 0 || pwi.z == 0) continue;
                        beta = beta * (albedo * ppdf / ppdf);
                        w = pwi;
                        z = zp;
                        if (((z < exitZ && w.z > 0) || (z > exitZ && w.z < 0)) && !IsSpecular(exitInterface.Flags())) {
                            S4 fExit = exitInterface.f(-w, wi, mode);
                            if (fExit) {
                                float exitPDF = exitInterface.PDF(-w, wi, mode, REFLTRANS_TRANSMISSION);
                                float wt2 = PowerHeuristic(1, ppdf, 1, exitPDF);
                                f = f + beta * Tr(zp - exitZ, pwi) * fExit * wt2;
                            }
                        }
                        continue;
                    }
                    z = Clamp(zp, 0.f, thickness);
                }
                if (z == exitZ) {
                    float uc2 = r();
                    float vy = r(), vx = r();
                    BSDFSample bs = exitInterface.Sample_f(-w, uc2, V2{vx, vy}, mode, REFLTRANS_REFLECTION);
                    if (Bad(bs)) break;
                    beta = beta * (bs.f * AbsCosTheta(bs.wi) / bs.pdf);
                    w = bs.wi;
                } else {
                    if (!IsSpecular(nonExitInterface.Flags())) {
                        float wt = 1;
                        if (!IsSpecular(exitInterface.Flags())) wt = PowerHeuristic(1, wis.pdf, 1, nonExitInterface.PDF(-w, -wis.wi, mode));
                        f = f + beta * nonExitInterface.f(-w, -wis.wi, mode) * AbsCosTheta(wis.wi) * wt * Tr(thickness, wis.wi) * wis.f / wis.pdf;
                    }
                    float uc2 = r();
                    float vy = r(), vx = r();
                    BSDFSample bs = nonExitInterface.Sample_f(-w, uc2, V2{vx, vy}, mode, REFLTRANS_REFLECTION);
                    if (Bad(bs)) break;
                    beta = beta * (bs.f * AbsCosTheta(bs.wi) / bs.pdf);
                    w = bs.wi;
                    if (!IsSpecular(exitInterface.Flags())) {
                        S4 fExit = exitInterface.f(-w, wi, mode);
                        if (fExit) {
                            float wt = 1;
                            if (!IsSpecular(nonExitInterface.Flags())) {
                                float exitPDF = exitInterface.PDF(-w, wi, mode, REFLTRANS_TRANSMISSION);
                                wt = PowerHeuristic(1, bs.pdf, 1, exitPDF);
                            }
                            f = f + beta * Tr(thickness, bs.wi) * fExit * wt;
                        }
                    }
                }
            }
        }
        return f / (float)nSamples;
    }
    WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags = REFLTRANS_ALL) const {
        bool flipWi = false;
        if (twoSided && wo.z < 0) { wo = -wo; flipWi = true; }
        bool enteredTop = twoSided || wo.z > 0;
        BSDFSample bs = enteredTop ? top.Sample_f(wo, uc, u, mode) : bottom.Sample_f(wo, uc, u, mode);
        if (Bad(bs)) return {};
        if (bs.IsReflection()) {
            if (flipWi) bs.wi = -bs.wi;
            bs.pdfIsProportional = true;
            return bs;
        }
        V3 w = bs.wi;
        bool specularPath = bs.IsSpecularS();
        RNG rng(HashIntV3(seed, wo), HashF3(uc, u));
        auto r = [&rng]() { return fmin(rng.UniformFloat(), OneMinusEpsilon); };
        S4 f = bs.f * AbsCosTheta(bs.wi);
        float pdf = bs.pdf;
        float z = enteredTop ? thickness : 0;
        for (int depth = 0; depth < maxDepth; ++depth) {
            float rrBeta = f.MaxComponentValue() / pdf;
            if (depth > 3 && rrBeta < 0.25f) {
                float q = fmax(0.f, 1 - rrBeta);
                if (r() < q) return {};
                pdf *= 1 - q;
            }
            if (w.z == 0) return {};
            if (albedo) {
                float sigma_t = 1;
                float dz = SampleExponential(r(), sigma_t / AbsCosTheta(w));
                float zp = w.z > 0 ? (z + dz) : (z - dz);
                if (zp == z) return {};
                if (0 < zp && zp < thickness) {
                    float u1 = r(), u0 = r();
                    float ppdf;
                    V3 pwi = SampleHenyeyGreenstein(-w, g, V2{u0, u1}, &ppdf);
                    if (ppdf == 0 || pwi.z == 0) return {};
                    f = f * (albedo * ppdf);
                    pdf *= ppdf;
                    specularPath = false;
                    w = pwi;
                    z = zp;
                    continue;
                }
                z = Clamp(zp, 0.f, thickness);
            } else {
                z = (z == thickness) ? 0 : thickness;
                f = f * Tr(thickness, w);
            }
            IF interface{this, z != 0};
            float uc2 = r();
            float vy = r(), vx = r();
            BSDFSample b2 = interface.Sample_f(-w, uc2, V2{vx, vy}, mode);
            if (Bad(b2)) return {};
            f = f * b2.f;
            pdf *= b2.pdf;
            specularPath &= b2.IsSpecularS();
            w = b2.wi;
            if (b2.IsTransmission()) {
                int flags = SameHemisphere(wo, w) ? BXDF_REFLECTION : BXDF_TRANSMISSION;
                flags |= specularPath ? BXDF_SPECULAR : BXDF_GLOSSY;
                if (flipWi) w = -w;
                return MakeSample(f, w, pdf, flags, 1.f, true);
            }
            f = f * AbsCosTheta(b2.wi);
        }
        return {};
    }
    WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags = REFLTRANS_ALL) const {
        if (twoSided && wo.z < 0) { wo = -wo; wi = -wi; }
        RNG rng(HashIntV3(seed, wi), Hash3f(wo));
        auto r = [&rng]() { return fmin(rng.UniformFloat(), OneMinusEpsilon); };
        bool enteredTop = twoSided || wo.z > 0;
        float pdfSum = 0;
        if (SameHemisphere(wo, wi)) {
            pdfSum += enteredTop ? nSamples * top.PDF(wo, wi, mode, REFLTRANS_REFLECTION) : nSamples * bottom.PDF(wo, wi, mode, REFLTRANS_REFLECTION);
        }
        for (int s = 0; s < nSamples; ++s) {
            if (SameHemisphere(wo, wi)) {
                IF rInterface{this, !enteredTop}, tInterface{this, enteredTop};
                float a1 = r(), a2 = r(), a0 = r();
                BSDFSample wos = tInterface.Sample_f(wo, a0, V2{a1, a2}, mode, REFLTRANS_TRANSMISSION);
                float b1 = r(), b2 = r(), b0 = r();
                BSDFSample wis = tInterface.Sample_f(wi, b0, V2{b1, b2}, !mode, REFLTRANS_TRANSMISSION);
                if (wos.valid && wos.f && wos.pdf > 0 && wis.valid && wis.f && wis.pdf > 0) {
                    if (!IsNonSpecular(tInterface.Flags())) pdfSum += rInterface.PDF(-wos.wi, -wis.wi, mode);
                    else {
                        float c1 = r(), c2 = r(), c0 = r();
                        BSDFSample rs = rInterface.Sample_f(-wos.wi, c0, V2{c1, c2}, mode);
                        if (rs.valid && rs.f && rs.pdf > 0) {
                            if (!IsNonSpecular(rInterface.Flags())) pdfSum += tInterface.PDF(-rs.wi, wi, mode);
                            else {
                                float rPDF = rInterface.PDF(-wos.wi, -wis.wi, mode);
                                float wt = PowerHeuristic(1, wis.pdf, 1, rPDF);
                                pdfSum += wt * rPDF;
                                float tPDF = tInterface.PDF(-rs.wi, wi, mode);
                                wt = PowerHeuristic(1, rs.pdf, 1, tPDF);
                                pdfSum += wt * tPDF;
                            }
                        }
                    }
                }
            } else {
                IF toInterface{this, enteredTop}, tiInterface{this, !enteredTop};
                float uc = r();
                float uy = r(), ux = r();
                BSDFSample wos = toInterface.Sample_f(wo, uc, V2{ux, uy}, mode);
                if (Bad(wos) || wos.IsReflection()) continue;
                uc = r();
                uy = r(); ux = r();
                BSDFSample wis = tiInterface.Sample_f(wi, uc, V2{ux, uy}, !mode);
                if (Bad(wis) || wis.IsReflection()) continue;
                if (IsSpecular(toInterface.Flags())) pdfSum += tiInterface.PDF(-wos.wi, wi, mode);
                else if (IsSpecular(tiInterface.Flags())) pdfSum += toInterface.PDF(wo, -wis.wi, mode);
                else pdfSum += (toInterface.PDF(wo, -wis.wi, mode) + tiInterface.PDF(-wos.wi, wi, mode)) / 2;
            }
        }
        return Lerp(0.9f, 1 / (4 * Pi), pdfSum / nSamples);
    }
};
using CoatedDiffuseBxDF = LayeredBxDF<DielectricBxDF, DiffuseBxDF, true>;
using CoatedConductorBxDF = LayeredBxDF<DielectricBxDF, ConductorBxDF, true>;

// ---------------------------------------------------------------------------------------------
// BSDF (bsdf.h:19-152): shading frame + the concrete BxDF
template <typename BxDF>
struct BSDF {
    BxDF bxdf;
    Frame shadingFrame;
    WF_HD BSDF(N3 ns, V3 dpdus, const BxDF &b) : bxdf(b), shadingFrame(Frame::FromXZ(Normalize(dpdus), toV(ns))) {}
    WF_HD int Flags() const { return bxdf.Flags(); }
    WF_HD V3 RenderToLocal(V3 v) const { return shadingFrame.ToLocal(v); }
    WF_HD V3 LocalToRender(V3 v) const { return shadingFrame.FromLocal(v); }
    WF_HD S4 f(V3 woRender, V3 wiRender, int mode = MODE_RADIANCE) const {
        V3 wi = RenderToLocal(wiRender), wo = RenderToLocal(woRender);
        if (wo.z == 0) return S4c(0.f);
        return bxdf.f(wo, wi, mode);
    }
    WF_HD BSDFSample Sample_f(V3 woRender, float u, V2 u2, int mode = MODE_RADIANCE, int sampleFlags = REFLTRANS_ALL) const {
        V3 wo = RenderToLocal(woRender);
        if (wo.z == 0) return {};
        if (!(bxdf.Flags() & sampleFlags)) return {};
        BSDFSample bs = bxdf.Sample_f(wo, u, u2, mode, sampleFlags);
        if (!bs.valid || !bs.f || bs.pdf == 0 || bs.wi.z == 0) return {};
        bs.wi = LocalToRender(bs.wi);
        return bs;
    }
    WF_HD float PDF(V3 woRender, V3 wiRender, int mode = MODE_RADIANCE, int sampleFlags = REFLTRANS_ALL) const {
        V3 wo = RenderToLocal(woRender), wi = RenderToLocal(wiRender);
        if (wo.z == 0) return 0;
        return bxdf.PDF(wo, wi, mode, sampleFlags);
    }
    WF_HD void Regularize() { bxdf.Regularize(); }
};

// ---------------------------------------------------------------------------------------------
// Material::GetBxDF for constant/scale/mix textures (BasicTextureEvaluator)
WF_HD S4 ClampS01(S4 s) { return ClampS(s, 0.f, 1.f); }

WF_HD DiffuseBxDF GetDiffuseBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    // materials.h:465-469
    S4 r = ClampS01(EvalSpectrumTexture(sv, m.tex[WF_MT_REFLECTANCE], lambda, tc));
    return DiffuseBxDF{r};
}
WF_HD DiffuseTransmissionBxDF GetDiffuseTransmissionBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    // materials.h:815-821
    S4 r = ClampS01(m.scale * EvalSpectrumTexture(sv, m.tex[WF_MT_REFLECTANCE], lambda, tc));
    S4 t = ClampS01(m.scale * EvalSpectrumTexture(sv, m.tex[WF_MT_TRANSMITTANCE], lambda, tc));
    return DiffuseTransmissionBxDF{r, t};
}
WF_HD float SampledEta(const SceneView &sv, const wf_material &m, Wavelengths &lambda) {
    // materials.h:184-192
    float sampledEta = SpectrumEval(sv, m.eta_spectrum, lambda.lambda[0]);
    if (!SpectrumIsConstant(sv, m.eta_spectrum)) lambda.TerminateSecondary();
    if (sampledEta == 0) sampledEta = 1;
    return sampledEta;
}
WF_HD DielectricBxDF GetDielectricBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    // materials.h:182-203
    float sampledEta = SampledEta(sv, m, lambda);
    float urough = EvalFloatTexture(sv, m.tex[WF_MT_UROUGH], tc), vrough = EvalFloatTexture(sv, m.tex[WF_MT_VROUGH], tc);
    if (m.flags & WF_MATFLAG_REMAP_ROUGHNESS) {
        urough = TrowbridgeReitz::RoughnessToAlpha(urough);
        vrough = TrowbridgeReitz::RoughnessToAlpha(vrough);
    }
    return DielectricBxDF{sampledEta, TrowbridgeReitz(urough, vrough), sv.fatal};
}
WF_HD ThinDielectricBxDF GetThinDielectricBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    // materials.h:226-240
    return ThinDielectricBxDF{SampledEta(sv, m, lambda)};
}
WF_HD ConductorBxDF GetConductorBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    // materials.h:491-511
    float uRough = EvalFloatTexture(sv, m.tex[WF_MT_UROUGH], tc), vRough = EvalFloatTexture(sv, m.tex[WF_MT_VROUGH], tc);
    if (m.flags & WF_MATFLAG_REMAP_ROUGHNESS) {
        uRough = TrowbridgeReitz::RoughnessToAlpha(uRough);
        vRough = TrowbridgeReitz::RoughnessToAlpha(vRough);
    }
    S4 etas, ks;
    if (!(m.flags & WF_MATFLAG_CONDUCTOR_REFLECTANCE)) {
        etas = EvalSpectrumTexture(sv, m.tex[WF_MT_ETA], lambda, tc);
        ks = EvalSpectrumTexture(sv, m.tex[WF_MT_K], lambda, tc);
    } else {
        S4 r = ClampS(EvalSpectrumTexture(sv, m.tex[WF_MT_REFLECTANCE], lambda, tc), 0.f, .9999f);
        etas = S4c(1.f);
        ks = 2 * Sqrt(r) / Sqrt(ClampZero(S4c(1.f) - r));
    }
    return ConductorBxDF{TrowbridgeReitz(uRough, vRough), etas, ks};
}

WF_HD CoatedDiffuseBxDF GetCoatedDiffuseBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    // materials.cpp:255-284
    S4 r = ClampS01(EvalSpectrumTexture(sv, m.tex[WF_MT_REFLECTANCE], lambda, tc));
    float urough = EvalFloatTexture(sv, m.tex[WF_MT_UROUGH], tc);
    float vrough = EvalFloatTexture(sv, m.tex[WF_MT_VROUGH], tc);
    if (m.flags & WF_MATFLAG_REMAP_ROUGHNESS) {
        urough = TrowbridgeReitz::RoughnessToAlpha(urough);
        vrough = TrowbridgeReitz::RoughnessToAlpha(vrough);
    }
    TrowbridgeReitz distrib(urough, vrough);
    float thick = EvalFloatTexture(sv, m.tex[WF_MT_THICKNESS], tc);
    float sampledEta = SampledEta(sv, m, lambda);
    S4 a = ClampS01(EvalSpectrumTexture(sv, m.tex[WF_MT_ALBEDO], lambda, tc));
    float gg = Clamp(EvalFloatTexture(sv, m.tex[WF_MT_G], tc), -1.f, 1.f);
    return CoatedDiffuseBxDF{DielectricBxDF{sampledEta, distrib, sv.fatal}, DiffuseBxDF{r}, fmax(thick, 1.17549435e-38f), gg, a, m.maxdepth, m.nsamples,
                             sv.options.seed};
}
WF_HD CoatedConductorBxDF GetCoatedConductorBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    // materials.cpp:346-392
    float iurough = EvalFloatTexture(sv, m.tex[WF_MT_UROUGH], tc);
    float ivrough = EvalFloatTexture(sv, m.tex[WF_MT_VROUGH], tc);
    if (m.flags & WF_MATFLAG_REMAP_ROUGHNESS) {
        iurough = TrowbridgeReitz::RoughnessToAlpha(iurough);
        ivrough = TrowbridgeReitz::RoughnessToAlpha(ivrough);
    }
    TrowbridgeReitz interfaceDistrib(iurough, ivrough);
    float thick = EvalFloatTexture(sv, m.tex[WF_MT_THICKNESS], tc);
    float ieta = SampledEta(sv, m, lambda);
    S4 ce, ck;
    if (!(m.flags & WF_MATFLAG_CONDUCTOR_REFLECTANCE)) {
        ce = EvalSpectrumTexture(sv, m.tex[WF_MT_COND_ETA], lambda, tc);
        ck = EvalSpectrumTexture(sv, m.tex[WF_MT_COND_K], lambda, tc);
    } else {
        S4 r = ClampS(EvalSpectrumTexture(sv, m.tex[WF_MT_REFLECTANCE], lambda, tc), 0.f, .9999f);
        ce = S4c(1.f);
        ck = 2 * Sqrt(r) / Sqrt(ClampZero(S4c(1.f) - r));
    }
    ce = ce / ieta;
    ck = ck / ieta;
    float curough = EvalFloatTexture(sv, m.tex[WF_MT_COND_UROUGH], tc);
    float cvrough = EvalFloatTexture(sv, m.tex[WF_MT_COND_VROUGH], tc);
    if (m.flags & WF_MATFLAG_REMAP_ROUGHNESS) {
        curough = TrowbridgeReitz::RoughnessToAlpha(curough);
        cvrough = TrowbridgeReitz::RoughnessToAlpha(cvrough);
    }
    TrowbridgeReitz conductorDistrib(curough, cvrough);
    S4 a = ClampS01(EvalSpectrumTexture(sv, m.tex[WF_MT_ALBEDO], lambda, tc));
    float gg = Clamp(EvalFloatTexture(sv, m.tex[WF_MT_G], tc), -1.f, 1.f);
    return CoatedConductorBxDF{DielectricBxDF{ieta, interfaceDistrib, sv.fatal}, ConductorBxDF{conductorDistrib, ce, ck}, fmax(thick, 1.17549435e-38f), gg, a,
                               m.maxdepth, m.nsamples, sv.options.seed};
}

}  // namespace wf
