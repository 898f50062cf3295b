// wf_noise.h — Perlin noise and the procedural functions built on it (util/noise.cpp:59-145), operation for operation.
// `perm` = the 2 x 256-entry permutation table (data/noise_perm.txt, uploaded with the scene).
#pragma once

#include "wf_math.h"

namespace wf {

WF_HD float NoiseGrad(const int32_t *perm, int x, int y, int z, float dx, float dy, float dz) {  // :98-104
    int h = perm[perm[perm[x] + y] + z];
    h &= 15;
    float u = h < 8 || h == 12 || h == 13 ? dx : dy;
    float v = h < 4 || h == 12 || h == 13 ? dy : dz;
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
WF_HD float NoiseWeight(float t) {  // :106-108; Pow<n> (util/math.h:176-186): Pow<5>(t) = (t*t)*(t*t)*t, Pow<4> = (t*t)*(t*t), Pow<3> = t*t*t
    float t2 = t * t;
    return 6 * (t2 * t2 * t) - 15 * (t2 * t2) + 10 * (t * t * t);
}
WF_HD float Noise(const int32_t *perm, float x, float y = .5f, float z = .5f) {  // :59-92
    x = fmod(x, (float)(1 << 30));
    y = fmod(y, (float)(1 << 30));
    z = fmod(z, (float)(1 << 30));
    int ix = (int)floor(x), iy = (int)floor(y), iz = (int)floor(z);
    float dx = x - ix, dy = y - iy, dz = z - iz;
    ix &= 255; iy &= 255; iz &= 255;
    float w000 = NoiseGrad(perm, ix, iy, iz, dx, dy, dz);
    float w100 = NoiseGrad(perm, ix + 1, iy, iz, dx - 1, dy, dz);
    float w010 = NoiseGrad(perm, ix, iy + 1, iz, dx, dy - 1, dz);
    float w110 = NoiseGrad(perm, ix + 1, iy + 1, iz, dx - 1, dy - 1, dz);
    float w001 = NoiseGrad(perm, ix, iy, iz + 1, dx, dy, dz - 1);
    float w101 = NoiseGrad(perm, ix + 1, iy, iz + 1, dx - 1, dy, dz - 1);
    float w011 = NoiseGrad(perm, ix, iy + 1, iz + 1, dx, dy - 1, dz - 1);
    float w111 = NoiseGrad(perm, ix + 1, iy + 1, iz + 1, dx - 1, dy - 1, dz - 1);
    float wx = NoiseWeight(dx), wy = NoiseWeight(dy), wz = NoiseWeight(dz);
    float x00 = Lerp(wx, w000, w100);
    float x10 = Lerp(wx, w010, w110);
    float x01 = Lerp(wx, w001, w101);
    float x11 = Lerp(wx, w011, w111);
    float y0 = Lerp(wy, x00, x10);
    float y1 = Lerp(wy, x01, x11);
    return Lerp(wz, y0, y1);
}
WF_HD float Noise3(const int32_t *perm, V3 p) { return Noise(perm, p.x, p.y, p.z); }
WF_HD float SmoothStep(float x, float a, float b) {  // util/math.h:268-274
    if (a == b) return (x < a) ? 0.f : 1.f;
    float t = Clamp((x - a) / (b - a), 0.f, 1.f);
    return t * t * (3 - 2 * t);
}
WF_HD float Log2f(float x) { return log(x) * 1.442695040888963387004650940071f; }  // util/math.h:385-388
WF_HD float FBm(const int32_t *perm, V3 p, V3 dpdx, V3 dpdy, float omega, int maxOctaves) {  // :118-133
    float len2 = fmax(LengthSquared(dpdx), LengthSquared(dpdy));
    float n = Clamp(-1 - Log2f(len2) / 2, 0.f, (float)maxOctaves);
    int nInt = (int)floor(n);
    float sum = 0, lambda = 1, o = 1;
    for (int i = 0; i < nInt; ++i) {
        sum += o * Noise3(perm, lambda * p);
        lambda *= 1.99f;
        o *= omega;
    }
    float nPartial = n - nInt;
    sum += o * SmoothStep(nPartial, .3f, .7f) * Noise3(perm, lambda * p);
    return sum;
}
WF_HD float Turbulence(const int32_t *perm, V3 p, V3 dpdx, V3 dpdy, float omega, int maxOctaves) {  // :135-158
    float len2 = fmax(LengthSquared(dpdx), LengthSquared(dpdy));
    float n = Clamp(-1 - Log2f(len2) / 2, 0.f, (float)maxOctaves);
    int nInt = (int)floor(n);
    float sum = 0, lambda = 1, o = 1;
    for (int i = 0; i < nInt; ++i) {
        sum += o * abs(Noise3(perm, lambda * p));
        lambda *= 1.99f;
        o *= omega;
    }
    float nPartial = n - nInt;
    sum += o * Lerp(SmoothStep(nPartial, .3f, .7f), 0.2f, abs(Noise3(perm, lambda * p)));
    for (int i = nInt; i < maxOctaves; ++i) {
        sum += o * 0.2f;
        o *= omega;
    }
    return sum;
}

}  // namespace wf
