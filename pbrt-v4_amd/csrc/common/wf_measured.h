// wf_measured.h — MeasuredBxDF (bxdfs.h:1022-1069, bxdfs.cpp:998-1113) and the PiecewiseLinear2D<N> interpolant / warp it is built on
// (util/sampling.h:1298-1745: Sample :1450-1549, Invert :1552-1646, Evaluate :1652-1709, lookup :1722-1738), operation for operation.
// The tables are laid out in table_data by csrc/host/measured_io.cpp (wf_material::measured_table, include/wf_abi.h): a header of int32
// words stored as float bit patterns, then the arrays.  MeasuredMaterial::GetBxDF (materials.h:854-858) hands the BxDF the data and the
// path's wavelengths; nothing is textured.
#pragma once
#include "wf_bxdf.h"

namespace wf {

struct PLSample {
    V2 p;
    float pdf;
};

// DIM = the number of parameters the 2D function is conditioned on (0: ndf, sigma; 2: vndf, luminance over (phi_o, theta_o); 3: spectra,
// + wavelength); a view over the words at table[hdr ...]
template <int DIM>
struct PiecewiseLinear2D {
    const float *data, *marginalCdf, *conditionalCdf;
    const float *paramValues[DIM > 0 ? DIM : 1];
    int paramSize[DIM > 0 ? DIM : 1], paramStride[DIM > 0 ? DIM : 1];
    int sizeX, sizeY;
    float patchX, patchY, invPatchX, invPatchY;

    WF_HD PiecewiseLinear2D(const float *table, int hdr) {
        auto word = [&](int k) { return (int)FloatToBits(table[hdr + k]); };
        sizeX = word(0); sizeY = word(1);
        for (int i = 0; i < DIM; ++i) { paramSize[i] = word(2 + i); paramStride[i] = word(5 + i); paramValues[i] = table + word(8 + i); }
        data = table + word(11);
        marginalCdf = table + word(12);
        conditionalCdf = table + word(13);
        patchX = 1.f / (sizeX - 1); patchY = 1.f / (sizeY - 1);
        invPatchX = (float)(sizeX - 1); invPatchY = (float)(sizeY - 1);
    }
    // "Look up parameter-related indices and weights"; returns slice_offset
    WF_HD uint32_t Weights(const float *param, float *w) const {
        uint32_t sliceOffset = 0;
        for (int dim = 0; dim < DIM; ++dim) {
            if (paramSize[dim] == 1) {
                w[2 * dim] = 1.f;
                w[2 * dim + 1] = 0.f;
                continue;
            }
            const float *pv = paramValues[dim];
            const float pd = param[dim];
            int idx = FindInterval(paramSize[dim], [&](int i) { return pv[i] <= pd; });
            float p0 = pv[idx], p1 = pv[idx + 1];
            w[2 * dim + 1] = Clamp((pd - p0) / (p1 - p0), 0.f, 1.f);
            w[2 * dim] = 1.f - w[2 * dim + 1];
            sliceOffset += (uint32_t)paramStride[dim] * (uint32_t)idx;
        }
        return sliceOffset;
    }
    template <int D>
    WF_HD float Lookup(const float *a, uint32_t i0, uint32_t size, const float *w) const {
        if constexpr (D == 0) return a[i0];
        else {
            uint32_t i1 = i0 + (uint32_t)paramStride[D - 1] * size;
            float w0 = w[2 * D - 2], w1 = w[2 * D - 1];
            float v0 = Lookup<D - 1>(a, i0, size, w), v1 = Lookup<D - 1>(a, i1, size, w);
            return fma(v0, w0, v1 * w1);
        }
    }

    WF_HD PLSample Sample(V2 sample, const float *param) const {
        sample.x = Clamp(sample.x, 1 - OneMinusEpsilon, OneMinusEpsilon);
        sample.y = Clamp(sample.y, 1 - OneMinusEpsilon, OneMinusEpsilon);
        float w[2 * (DIM > 0 ? DIM : 1)];
        const uint32_t sliceOffset = Weights(param, w);
        // sample the row
        uint32_t offset = DIM != 0 ? sliceOffset * (uint32_t)sizeY : 0u;
        auto fetchMarginal = [&](uint32_t idx) { return Lookup<DIM>(marginalCdf, offset + idx, (uint32_t)sizeY, w); };
        const uint32_t row = (uint32_t)FindInterval(sizeY, [&](int idx) { return fetchMarginal((uint32_t)idx) < sample.y; });
        sample.y -= fetchMarginal(row);
        const uint32_t sliceSize = (uint32_t)(sizeX * sizeY);
        offset = row * (uint32_t)sizeX;
        if (DIM != 0) offset += sliceOffset * sliceSize;
        float r0 = Lookup<DIM>(conditionalCdf, offset + sizeX - 1, sliceSize, w), r1 = Lookup<DIM>(conditionalCdf, offset + (sizeX * 2 - 1), sliceSize, w);
        bool isConst = abs(r0 - r1) < 1e-4f * (r0 + r1);
        sample.y = isConst ? (2.f * sample.y) : (r0 - SafeSqrt(r0 * r0 - 2.f * sample.y * (r0 - r1)));
        sample.y /= isConst ? (r0 + r1) : (r0 - r1);
        // sample the column
        sample.x *= (1.f - sample.y) * r0 + sample.y * r1;
        auto fetchConditional = [&](uint32_t idx) {
            float v0 = Lookup<DIM>(conditionalCdf, offset + idx, sliceSize, w), v1 = Lookup<DIM>(conditionalCdf + sizeX, offset + idx, sliceSize, w);
            return (1.f - sample.y) * v0 + sample.y * v1;
        };
        const uint32_t col = (uint32_t)FindInterval(sizeX, [&](int idx) { return fetchConditional((uint32_t)idx) < sample.x; });
        sample.x -= fetchConditional(col);
        offset += col;
        float v00 = Lookup<DIM>(data, offset, sliceSize, w), v10 = Lookup<DIM>(data + 1, offset, sliceSize, w),
              v01 = Lookup<DIM>(data + sizeX, offset, sliceSize, w), v11 = Lookup<DIM>(data + sizeX + 1, offset, sliceSize, w);
        float c0 = fma(1.f - sample.y, v00, sample.y * v01), c1 = fma(1.f - sample.y, v10, sample.y * v11);
        isConst = abs(c0 - c1) < 1e-4f * (c0 + c1);
        sample.x = isConst ? (2.f * sample.x) : (c0 - SafeSqrt(c0 * c0 - 2.f * sample.x * (c0 - c1)));
        sample.x /= isConst ? (c0 + c1) : (c0 - c1);
        return PLSample{V2{(col + sample.x) * patchX, (row + sample.y) * patchY}, ((1.f - sample.x) * c0 + sample.x * c1) * (invPatchX * invPatchY)};
    }

    WF_HD PLSample Invert(V2 sample, const float *param) const {
        float w[2 * (DIM > 0 ? DIM : 1)];
        const uint32_t sliceOffset = Weights(param, w);
        sample.x *= invPatchX;
        sample.y *= invPatchY;
        int posX = (int)sample.x, posY = (int)sample.y;
        if (sizeX - 2 < posX) posX = sizeX - 2;
        if (sizeY - 2 < posY) posY = sizeY - 2;
        sample.x -= (float)posX;
        sample.y -= (float)posY;
        uint32_t offset = (uint32_t)(posX + posY * sizeX);
        const uint32_t sliceSize = (uint32_t)(sizeX * sizeY);
        if (DIM != 0) offset += sliceOffset * sliceSize;
        // invert the x component
        float v00 = Lookup<DIM>(data, offset, sliceSize, w), v10 = Lookup<DIM>(data + 1, offset, sliceSize, w),
              v01 = Lookup<DIM>(data + sizeX, offset, sliceSize, w), v11 = Lookup<DIM>(data + sizeX + 1, offset, sliceSize, w);
        float w1x = sample.x, w1y = sample.y, w0x = 1 - w1x, w0y = 1 - w1y;
        float c0 = fma(w0y, v00, w1y * v01), c1 = fma(w0y, v10, w1y * v11), pdf = fma(w0x, c0, w1x * c1);
        sample.x *= c0 + .5f * sample.x * (c1 - c0);
        float v0 = Lookup<DIM>(conditionalCdf, offset, sliceSize, w), v1 = Lookup<DIM>(conditionalCdf + sizeX, offset, sliceSize, w);
        sample.x += (1.f - sample.y) * v0 + sample.y * v1;
        offset = (uint32_t)(posY * sizeX);
        if (DIM != 0) offset += sliceOffset * sliceSize;
        float r0 = Lookup<DIM>(conditionalCdf, offset + sizeX - 1, sliceSize, w), r1 = Lookup<DIM>(conditionalCdf, offset + (sizeX * 2 - 1), sliceSize, w);
        sample.x /= (1.f - sample.y) * r0 + sample.y * r1;
        // invert the y component
        sample.y *= r0 + .5f * sample.y * (r1 - r0);
        offset = (uint32_t)posY;
        if (DIM != 0) offset += sliceOffset * (uint32_t)sizeY;
        sample.y += Lookup<DIM>(marginalCdf, offset, (uint32_t)sizeY, w);
        return PLSample{sample, pdf * (invPatchX * invPatchY)};
    }

    WF_HD float Evaluate(V2 pos, const float *param) const {
        float w[2 * (DIM > 0 ? DIM : 1)];
        const uint32_t sliceOffset = Weights(param, w);
        pos.x *= invPatchX;
        pos.y *= invPatchY;
        int offX = (int)pos.x, offY = (int)pos.y;
        if (sizeX - 2 < offX) offX = sizeX - 2;
        if (sizeY - 2 < offY) offY = sizeY - 2;
        float w1x = pos.x - (float)offX, w1y = pos.y - (float)offY, w0x = 1 - w1x, w0y = 1 - w1y;
        uint32_t index = (uint32_t)(offX + offY * sizeX);
        const uint32_t size = (uint32_t)(sizeX * sizeY);
        if (DIM != 0) index += sliceOffset * size;
        float v00 = Lookup<DIM>(data, index, size, w), v10 = Lookup<DIM>(data + 1, index, size, w), v01 = Lookup<DIM>(data + sizeX, index, size, w),
              v11 = Lookup<DIM>(data + sizeX + 1, index, size, w);
        return fma(w0y, fma(w0x, v00, w1x * v10), w1y * fma(w0x, v01, w1x * v11)) * (invPatchX * invPatchY);
    }
};

struct MeasuredBxDF {
    const float *table;   // the scene's table_data
    int hdr;              // wf_material::measured_table
    float lambda[4];

    WF_HD static float theta2u(float theta) { return sqrt(theta * (2 / Pi)); }
    WF_HD static float phi2u(float phi) { return phi * (1 / (2 * Pi)) + .5f; }
    WF_HD static float u2theta(float u) { return Sqr(u) * (Pi / 2.f); }
    WF_HD static float u2phi(float u) { return (2.f * u - 1.f) * Pi; }
    WF_HD bool Isotropic() const { return FloatToBits(table[hdr]) != 0; }
    WF_HD PiecewiseLinear2D<0> Ndf() const { return PiecewiseLinear2D<0>(table, hdr + 16); }
    WF_HD PiecewiseLinear2D<0> Sigma() const { return PiecewiseLinear2D<0>(table, hdr + 32); }
    WF_HD PiecewiseLinear2D<2> Vndf() const { return PiecewiseLinear2D<2>(table, hdr + 48); }
    WF_HD PiecewiseLinear2D<2> Luminance() const { return PiecewiseLinear2D<2>(table, hdr + 64); }
    WF_HD PiecewiseLinear2D<3> Spectra() const { return PiecewiseLinear2D<3>(table, hdr + 80); }

    // bxdfs.cpp:998-1033
    WF_HD S4 f(V3 wo, V3 wi, int mode) const {
        if (!SameHemisphere(wo, wi)) return S4c(0.f);
        if (wo.z < 0) { wo = -wo; wi = -wi; }
        V3 wm = wi + wo;
        if (LengthSquared(wm) == 0) return S4c(0.f);
        wm = Normalize(wm);
        float theta_o = SafeACos(wo.z), phi_o = atan2(wo.y, wo.x);
        float theta_m = SafeACos(wm.z), phi_m = atan2(wm.y, wm.x);
        V2 u_wo{theta2u(theta_o), phi2u(phi_o)};
        V2 u_wm{theta2u(theta_m), phi2u(Isotropic() ? (phi_m - phi_o) : phi_m)};
        u_wm.y = u_wm.y - floor(u_wm.y);
        const float p2[2] = {phi_o, theta_o};
        PLSample ui = Vndf().Invert(u_wm, p2);
        S4 fr;
        const PiecewiseLinear2D<3> spectra = Spectra();
        for (int i = 0; i < 4; ++i) {
            const float p3[3] = {phi_o, theta_o, lambda[i]};
            fr[i] = fmax(0.f, spectra.Evaluate(ui.p, p3));
        }
        return fr * Ndf().Evaluate(u_wm, nullptr) / (4 * Sigma().Evaluate(u_wo, nullptr) * CosTheta(wi));
    }
    // bxdfs.cpp:1035-1083
    WF_HD BSDFSample Sample_f(V3 wo, float uc, V2 u, int mode, int sampleFlags = REFLTRANS_ALL) const {
        if (!(sampleFlags & REFLTRANS_REFLECTION)) return {};
        bool flipWi = false;
        if (wo.z <= 0) { wo = -wo; flipWi = true; }
        float theta_o = SafeACos(wo.z), phi_o = atan2(wo.y, wo.x);
        const float p2[2] = {phi_o, theta_o};
        PLSample s = Luminance().Sample(u, p2);
        u = s.p;
        float lum_pdf = s.pdf;
        s = Vndf().Sample(u, p2);
        V2 u_wm = s.p;
        float pdf = s.pdf;
        float phi_m = u2phi(u_wm.y), theta_m = u2theta(u_wm.x);
        if (Isotropic()) phi_m += phi_o;
        float sinTheta_m = sin(theta_m), cosTheta_m = cos(theta_m);
        V3 wm = SphericalDirection(sinTheta_m, cosTheta_m, phi_m);
        V3 wi = Reflect(wo, wm);
        if (wi.z <= 0) return {};
        S4 fr = S4c(0.f);
        const PiecewiseLinear2D<3> spectra = Spectra();
        for (int i = 0; i < 4; ++i) {
            const float p3[3] = {phi_o, theta_o, lambda[i]};
            fr[i] = fmax(0.f, spectra.Evaluate(u, p3));
        }
        V2 u_wo{theta2u(theta_o), phi2u(phi_o)};
        fr = fr * (Ndf().Evaluate(u_wm, nullptr) / (4 * Sigma().Evaluate(u_wo, nullptr) * AbsCosTheta(wi)));
        pdf /= 4 * Dot(wo, wm) * fmax(2 * Sqr(Pi) * u_wm.x * sinTheta_m, 1e-6f);
        if (flipWi) wi = -wi;
        return MakeSample(fr, wi, pdf * lum_pdf, BXDF_GLOSSY_REFLECTION);
    }
    // bxdfs.cpp:1085-1113
    WF_HD float PDF(V3 wo, V3 wi, int mode, int sampleFlags = REFLTRANS_ALL) const {
        if (!(sampleFlags & REFLTRANS_REFLECTION)) return 0;
        if (!SameHemisphere(wo, wi)) return 0;
        if (wo.z < 0) { wo = -wo; wi = -wi; }
        V3 wm = wi + wo;
        if (LengthSquared(wm) == 0) return 0;
        wm = Normalize(wm);
        float theta_o = SafeACos(wo.z), phi_o = atan2(wo.y, wo.x);
        float theta_m = SafeACos(wm.z), phi_m = atan2(wm.y, wm.x);
        V2 u_wm{theta2u(theta_m), phi2u(Isotropic() ? (phi_m - phi_o) : phi_m)};
        u_wm.y = u_wm.y - floor(u_wm.y);
        const float p2[2] = {phi_o, theta_o};
        PLSample ui = Vndf().Invert(u_wm, p2);
        float vndfPDF = ui.pdf;
        float pdf = Luminance().Evaluate(ui.p, p2);
        float sinTheta_m = sqrt(Sqr(wm.x) + Sqr(wm.y));
        float jacobian = 4.f * Dot(wo, wm) * fmax(2 * Sqr(Pi) * u_wm.x * sinTheta_m, 1e-6f);
        return vndfPDF * pdf / jacobian;
    }
    WF_HD void Regularize() {}
    WF_HD int Flags() const { return BXDF_GLOSSY_REFLECTION; }
};

// MeasuredMaterial::GetBxDF (materials.h:854-858)
WF_HD MeasuredBxDF GetMeasuredBxDF(const SceneView &sv, const wf_material &m, Wavelengths &lambda, const TexCtx &tc) {
    MeasuredBxDF b;
    b.table = sv.tableData;
    b.hdr = m.measured_table;
    for (int i = 0; i < 4; ++i) b.lambda[i] = lambda.lambda[i];
    return b;
}

}  // namespace wf
