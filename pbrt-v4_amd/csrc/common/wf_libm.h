// wf_libm.h — glibc 2.35 (x86-64) float elementary functions, restated for the device.
//
// The reference calls std::sin/cos/atan2/acos/asin/exp/log/cosh/atanh on floats (e.g. util/math.h:1139-1157
// SphericalDirection, bxdfs.h:510 SampleUniformDiskPolar via sampling.h, util/spectrum.h:76-85 the visible-wavelength
// sampling, media.h:760 exp of the majorant transmittance) and thereby inherits whatever float libm it is linked
// against.  The oracle build (oracle/_ref) and the committed goldens use this image's glibc 2.35; its results are *not*
// correctly rounded, and one ulp re-seeds the hash-seeded random walks of the path (wavefront/media.cpp:44,
// cpu/primitive.cpp:60, the LayeredBxDF).  The device therefore evaluates the *same algorithms*:
//
//  * sinf / cosf   — sysdeps/ieee754/flt-32/s_sincosf.h (ARM optimized-routines): double-precision polynomials,
//                    reduce_fast / reduce_large with the 2/pi bit table.  Operation order and fusing follow the
//                    ifunc-selected FMA variant (__sinf_fma / __cosf_fma of libm-2.35) as disassembled: every
//                    `a + b*c` of the source is one fma.
//  * expf / logf   — sysdeps/ieee754/flt-32/e_expf.c / e_logf.c (same origin), __expf_fma / __logf_fma fusing.
//  * atanf, atan2f, acosf, asinf, coshf (+ expm1f on |x| <= ln2/2), atanhf (+ log1pf) — the fdlibm-derived float
//    routines of sysdeps/ieee754/flt-32, compiled for baseline x86-64 (SSE2, no fusing).
//
// The table constants are the ones of the installed libm (tools/extract_libm_tables.py).  Every function is compared
// with the live libm: exhaustively over all 2^32 arguments on the host (oracle/wf_cpu/libm_check.cpp), and on the GPU
// over seeded + edge-case vectors (tests/test_gpu_parity.py::test_device_libm_*).  Signalling (errno, FP exceptions) is
// not reproduced; results, including infinities / NaN-ness, are.
//
// The file has no dependencies so the host checker can include it alone; compile with -ffp-contract=off.
//
// LICENCE: a derived work of glibc 2.35's flt-32 sources (LGPL 2.1 or later; the fdlibm-derived routines also carry Sun's
// permissive notice) — see wf_libm.LICENSE next to this file, which must travel with it.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define WFLM_HD __host__ __device__ inline
#else
#define WFLM_HD inline
#endif
// the entry points (sinf, cosf, ...): WF_LIBM_NOINLINE makes them real calls on the device (code-size experiments on the material
// kernels, which reach them from hundreds of sites)
#if defined(__HIPCC__) && defined(WF_LIBM_NOINLINE) && defined(__HIP_DEVICE_COMPILE__)
#define WFLM_FN __host__ __device__ __attribute__((noinline))
#else
#define WFLM_FN WFLM_HD
#endif

namespace glibc235 {

WFLM_HD uint32_t asuint(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
#endif
}
WFLM_HD float asfloat(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    std::memcpy(&f, &u, 4);
    return f;
#endif
}
WFLM_HD uint64_t asuint64(double f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(f);
#else
    uint64_t u;
    std::memcpy(&u, &f, 8);
    return u;
#endif
}
WFLM_HD double asdouble(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double f;
    std::memcpy(&f, &u, 8);
    return f;
#endif
}
WFLM_HD double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
WFLM_HD float fsqrt(float x) { return __builtin_sqrtf(x); }
WFLM_HD float ffabs(float x) { return asfloat(asuint(x) & 0x7fffffffu); }
WFLM_HD float fnan() { return asfloat(0x7fc00000u); }

// ---------------------------------------------------------------------------------------------------------------
// sinf / cosf (s_sincosf.h).  __sincosf_table[1] is table[0] with the cosine coefficients negated; `neg` selects it.
struct SinCosPoly {
    double c0, c1, c2, c3, c4, s1, s2, s3;
};
WFLM_HD SinCosPoly sincos_table(bool neg) {
    const double sg = neg ? -1.0 : 1.0;
    return SinCosPoly{sg * 0x1.0000000000000p+0,  sg * -0x1.ffffffd0c621cp-2, sg * 0x1.55553e1068f19p-5,
                      sg * -0x1.6c087e89a359dp-10, sg * 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
                      0x1.1107605230bc4p-7,        -0x1.994eb3774cf24p-13};
}
// sign[n & 3] of the table: {1, -1, -1, 1}
WFLM_HD double sincos_sign(int n) { return ((n + 1) & 2) ? -1.0 : 1.0; }

WFLM_HD float sinf_poly(double x, double x2, const SinCosPoly &p, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = dfma(x2, p.s3, p.s2);
        double x7 = x3 * x2;
        double s = dfma(x3, p.s1, x);
        return (float)dfma(s1, x7, s);
    } else {
        double x4 = x2 * x2;
        double c2 = dfma(x2, p.c4, p.c3);
        double c1 = dfma(x2, p.c1, p.c0);
        double x6 = x4 * x2;
        double c = dfma(x4, p.c2, c1);
        return (float)dfma(c2, x6, c);
    }
}
WFLM_HD double reduce_fast(double x, int *np) {
    // hpi_inv = 2/pi * 2^24, hpi = pi/2
    double r = x * 0x1.45f306dc9c883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return dfma(-(double)n, 0x1.921fb54442d18p+0, x);
}
WFLM_HD uint32_t inv_pio4(int i) {
    // __inv_pio4[i] = bits [8i-24 .. 8i+7] of 4/pi: consecutive entries overlap by 24 bits
    const uint32_t t[24] = {0x000000a2u, 0x0000a2f9u, 0x00a2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
                            0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
                            0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
    return t[i];
}
WFLM_HD double reduce_large(uint32_t xi, int *np) {
    int i0 = (xi >> 26) & 15;
    int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = (uint32_t)(xi * inv_pio4(i0));
    res1 = (uint64_t)xi * inv_pio4(i0 + 4);
    res2 = (uint64_t)xi * inv_pio4(i0 + 8);
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}
WFLM_HD uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }

WFLM_FN float sinf(float y) {
    double x = y;
    int n;
    uint32_t top = abstop12(y);
    if (top < 0x3f4) {  // |y| < pi/4
        double x2 = x * x;
        if (top < 0x398) return y;  // |y| < 2^-12
        return sinf_poly(x, x2, sincos_table(false), 0);
    } else if (top < 0x42f) {  // |y| < 120
        x = reduce_fast(x, &n);
        double s = sincos_sign(n);
        return sinf_poly(x * s, x * x, sincos_table((n & 2) != 0), n);
    } else if (top < 0x7f8) {
        uint32_t xi = asuint(y);
        int sign = xi >> 31;
        x = reduce_large(xi, &n);
        double s = sincos_sign(n + sign);
        return sinf_poly(x * s, x * x, sincos_table(((n + sign) & 2) != 0), n);
    }
    return fnan();  // inf or NaN
}
WFLM_FN float cosf(float y) {
    double x = y;
    int n;
    uint32_t top = abstop12(y);
    if (top < 0x3f4) {
        double x2 = x * x;
        if (top < 0x398) return 1.0f;
        return sinf_poly(x, x2, sincos_table(false), 1);
    } else if (top < 0x42f) {
        x = reduce_fast(x, &n);
        double s = sincos_sign(n);
        return sinf_poly(x * s, x * x, sincos_table((n & 2) != 0), n ^ 1);
    } else if (top < 0x7f8) {
        uint32_t xi = asuint(y);
        int sign = xi >> 31;
        x = reduce_large(xi, &n);
        double s = sincos_sign(n + sign);
        return sinf_poly(x * s, x * x, sincos_table(((n + sign) & 2) != 0), n ^ 1);
    }
    return fnan();
}

// ---------------------------------------------------------------------------------------------------------------
// expf (e_expf.c, N = 32, non-TOINT path)
WFLM_HD uint64_t exp2f_tab(int i) {
    const uint64_t t[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull,
        0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull,
        0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull,
        0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull,
        0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
        0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    return t[i];
}
WFLM_FN float expf(float x) {
    double xd = (double)x;
    uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
    if (abstop >= 0x42b) {  // |x| >= 88 or NaN
        if (asuint(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return asfloat(0x7f800000u);  // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                   // underflow
        if (x < -0x1.9d1d9ep6f) return asfloat(1u);            // __math_may_uflowf: 0x1.4p-75f^2 -> 2^-149
    }
    const double Shift = 0x1.8000000000000p+52, InvLn2N = 0x1.71547652b82fep+5;
    double kd = dfma(InvLn2N, xd, Shift);
    uint64_t ki = asuint64(kd);
    kd -= Shift;
    double r = dfma(InvLn2N, xd, -kd);
    uint64_t t = exp2f_tab((int)(ki & 31));
    t += ki << (52 - 5);
    double s = asdouble(t);
    double z = dfma(0x1.c6af84b912394p-20, r, 0x1.ebfce50fac4f3p-13);
    double r2 = r * r;
    double y = dfma(0x1.62e42ff0c52d6p-6, r, 1.0);
    y = dfma(z, r2, y);
    y = y * s;
    return (float)y;
}

// ---------------------------------------------------------------------------------------------------------------
// logf (e_logf.c, N = 16)
WFLM_HD void logf_tab(int i, double *invc, double *logc) {
    const double t[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1.0000000000000p+0, 0x0.0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    *invc = t[i][0];
    *logc = t[i][1];
}
WFLM_FN float logf(float x) {
    uint32_t ix = asuint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return asfloat(0xff800000u);  // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return fnan();
        ix = asuint(x * 0x1p23f);  // subnormal: normalise
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    int i = (tmp >> 19) & 15;
    int k = (int32_t)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000u);
    double invc, logc;
    logf_tab(i, &invc, &logc);
    double z = (double)asfloat(iz);
    double r = dfma(z, invc, -1.0);
    double y0 = dfma((double)k, 0x1.62e42fefa39efp-1, logc);
    double r2 = r * r;
    double y = dfma(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = dfma(-0x1.00ea348b88334p-2, r2, y);
    y = dfma(y, r2, y0 + r);
    return (float)y;
}

// ---------------------------------------------------------------------------------------------------------------
// atanf (s_atanf.c), atan2f (e_atan2f.c)
WFLM_FN float atanf(float x) {
    const float atanhi[4] = {asfloat(0x3eed6338u), asfloat(0x3f490fdau), asfloat(0x3f7b985eu), asfloat(0x3fc90fdau)};
    const float atanlo[4] = {asfloat(0x31ac3769u), asfloat(0x33222168u), asfloat(0x33140fb4u), asfloat(0x33a22168u)};
    const float aT0 = asfloat(0x3eaaaaabu), aT1 = asfloat(0xbe4ccccdu), aT2 = asfloat(0x3e124925u), aT3 = asfloat(0xbde38e38u),
                aT4 = asfloat(0x3dba2e6eu), aT5 = asfloat(0xbd9d8795u), aT6 = asfloat(0x3d886b35u), aT7 = asfloat(0xbd6ef16bu),
                aT8 = asfloat(0x3d4bda59u), aT9 = asfloat(0xbd15a221u), aT10 = asfloat(0x3c8569d7u);
    int32_t hx = (int32_t)asuint(x);
    int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        if (hx > 0) return atanhi[3] + atanlo[3];
        return -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                // |x| < 0.4375
        if (ix < 0x31000000) return x;    // |x| < 2^-29
        id = -1;
    } else {
        x = ffabs(x);
        if (ix < 0x3f980000) {      // |x| < 1.1875
            if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000) {  // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    float z = x * x;
    float w = z * z;
    float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}
WFLM_FN float atan2f(float y, float x) {
    const float tiny = asfloat(0x0da24260u), pi_o_4 = asfloat(0x3f490fdbu), pi_o_2 = asfloat(0x3fc90fdbu), pi = asfloat(0x40490fdbu),
                pi_lo = asfloat(0xb3bbbd2eu);
    int32_t hx = (int32_t)asuint(x), hy = (int32_t)asuint(y);
    int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf(y);
    int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
        case 0:
        case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf(ffabs(y / x));
    switch (m) {
    case 0: return z;
    case 1: return asfloat(asuint(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// asinf (e_asinf.c, minimax variant of glibc >= 2.27) and acosf (e_acosf.c)
WFLM_FN float asinf(float x) {
    const float pio2_hi = asfloat(0x3fc90fdbu), pio2_lo = asfloat(0xb33bbd2eu), pio4_hi = asfloat(0x3f490fdbu);
    const float p0 = asfloat(0x3e2aaae4u), p1 = asfloat(0x3d9980f2u), p2 = asfloat(0x3d3a3f25u), p3 = asfloat(0x3cc6141eu),
                p4 = asfloat(0x3d2cb694u);
    int32_t hx = (int32_t)asuint(x);
    int32_t ix = hx & 0x7fffffff;
    float t, w, p, q, c, r, s;
    if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;
    if (ix > 0x3f800000) return fnan();
    if (ix < 0x3f000000) {               // |x| < 0.5
        if (ix < 0x32000000) return x;   // |x| < 2^-27
        t = x * x;
        w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
        return x + x * w;
    }
    w = 1.0f - ffabs(x);
    t = w * 0.5f;
    p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    s = fsqrt(t);
    if (ix >= 0x3f79999a) {  // |x| > 0.975
        t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    } else {
        w = asfloat(asuint(s) & 0xfffff000u);
        c = (t - w * w) / (s + w);
        r = p;
        p = 2.0f * s * r - (pio2_lo - 2.0f * c);
        q = pio4_hi - 2.0f * w;
        t = pio4_hi - (p - q);
    }
    return (hx > 0) ? t : -t;
}
WFLM_FN float acosf(float x) {
    const float pi = asfloat(0x40490fdau), pio2_hi = asfloat(0x3fc90fdau), pio2_lo = asfloat(0x33a22168u);
    const float pS0 = asfloat(0x3e2aaaabu), pS1 = asfloat(0xbea6b090u), pS2 = asfloat(0x3e4e0aa8u), pS3 = asfloat(0xbd241146u),
                pS4 = asfloat(0x3a4f7f04u), pS5 = asfloat(0x3811ef08u), qS1 = asfloat(0xc019d139u), qS2 = asfloat(0x4001572du),
                qS3 = asfloat(0xbf303361u), qS4 = asfloat(0x3d9dc62eu);
    int32_t hx = (int32_t)asuint(x);
    int32_t ix = hx & 0x7fffffff;
    float z, p, q, r, w, s, c, df;
    if (ix == 0x3f800000) {
        if (hx > 0) return 0.0f;
        return pi + 2.0f * pio2_lo;
    }
    if (ix > 0x3f800000) return fnan();
    if (ix < 0x3f000000) {  // |x| < 0.5
        if (ix <= 0x32800000) return pio2_hi + pio2_lo;
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (hx < 0) {  // x < -0.5
        z = (1.0f + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = fsqrt(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {  // x > 0.5
        z = (1.0f - x) * 0.5f;
        s = fsqrt(z);
        df = asfloat(asuint(s) & 0xfffff000u);
        c = (z - df * df) / (s + df);
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        w = r * s + c;
        return 2.0f * (df + w);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// coshf (e_coshf.c).  expm1f (s_expm1f.c) is reached only for |x| <= ln2/2, i.e. its k == 0 branch.
WFLM_HD float expm1f_k0(float x) {
    const float Q1 = asfloat(0xbd088889u), Q2 = asfloat(0x3ad00d01u), Q3 = asfloat(0xb8a670cdu), Q4 = asfloat(0x36867e54u),
                Q5 = asfloat(0xb457edbbu);
    if ((asuint(x) & 0x7fffffffu) < 0x33000000u) return x;  // |x| < 2^-25
    float hfx = 0.5f * x;
    float hxs = x * hfx;
    float r1 = 1.0f + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
    float t = 3.0f - r1 * hfx;
    float e = hxs * ((r1 - t) / (6.0f - x * t));
    return x - (x * e - hxs);
}
WFLM_FN float coshf(float x) {
    int32_t ix = (int32_t)asuint(x) & 0x7fffffff;
    if (ix < 0x41b00000) {  // |x| < 22
        if (ix < 0x3eb17218) {
            if (ix < 0x24000000) return 1.0f;
            float t = expm1f_k0(ffabs(x));
            float w = 1.0f + t;
            return 1.0f + (t * t) / (w + w);
        }
        float t = expf(ffabs(x));
        return 0.5f * t + 0.5f / t;
    }
    if (ix <= 0x42b1717f) return 0.5f * expf(ffabs(x));
    if (ix <= 0x42b2d4fc) {
        float w = expf(0.5f * ffabs(x));
        float t = 0.5f * w;
        return t * w;
    }
    if (ix >= 0x7f800000) return x * x;
    return asfloat(0x7f800000u);
}

// ---------------------------------------------------------------------------------------------------------------
// sinhf (e_sinhf.c) over the full expm1f (s_expm1f.c) and expf
WFLM_FN float expm1f(float x) {
    const float o_threshold = asfloat(0x42b17180u), ln2_hi = asfloat(0x3f317180u), ln2_lo = asfloat(0x3717f7d1u), invln2 = asfloat(0x3fb8aa3bu);
    const float Q1 = asfloat(0xbd088889u), Q2 = asfloat(0x3ad00d01u), Q3 = asfloat(0xb8a670cdu), Q4 = asfloat(0x36867e54u),
                Q5 = asfloat(0xb457edbbu);
    float y, hi, lo, c = 0, t, e, hxs, hfx, r1;
    int32_t k;
    uint32_t hx = asuint(x);
    const uint32_t xsb = hx & 0x80000000u;
    hx &= 0x7fffffffu;
    if (hx >= 0x4195b844u) {        // |x| >= 27 ln2
        if (hx >= 0x42b17218u) {    // |x| >= 88.72...
            if (hx > 0x7f800000u) return x + x;
            if (hx == 0x7f800000u) return xsb == 0 ? x : -1.0f;
            if (x > o_threshold) return asfloat(0x7f800000u);  // huge * huge
        }
        if (xsb != 0) return 1.0e-30f - 1.0f;  // x < -27 ln2
    }
    if (hx > 0x3eb17218u) {         // |x| > 0.5 ln2
        if (hx < 0x3f851592u) {     // |x| < 1.5 ln2
            if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
            else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
        } else {
            k = (int32_t)(invln2 * x + (xsb == 0 ? 0.5f : -0.5f));
            t = (float)k;
            hi = x - t * ln2_hi;
            lo = t * ln2_lo;
        }
        x = hi - lo;
        c = (hi - x) - lo;
    } else if (hx < 0x33000000u) {  // |x| < 2^-25
        return x;
    } else k = 0;
    hfx = 0.5f * x;
    hxs = x * hfx;
    r1 = 1.0f + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
    t = 3.0f - r1 * hfx;
    e = hxs * ((r1 - t) / (6.0f - x * t));
    if (k == 0) return x - (x * e - hxs);
    e = (x * (e - c) - c);
    e -= hxs;
    if (k == -1) return 0.5f * (x - e) - 0.5f;
    if (k == 1) {
        if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
        return 1.0f + 2.0f * (x - e);
    }
    if (k <= -2 || k > 56) {
        y = 1.0f - (e - x);
        y = asfloat((uint32_t)((int32_t)asuint(y) + (k << 23)));
        return y - 1.0f;
    }
    if (k < 23) {
        t = asfloat((uint32_t)(0x3f800000 - (0x1000000 >> k)));  // 1 - 2^-k
        y = t - (e - x);
        y = asfloat((uint32_t)((int32_t)asuint(y) + (k << 23)));
    } else {
        t = asfloat((uint32_t)((0x7f - k) << 23));  // 2^-k
        y = x - (e + t);
        y += 1.0f;
        y = asfloat((uint32_t)((int32_t)asuint(y) + (k << 23)));
    }
    return y;
}
WFLM_FN float sinhf(float x) {
    const int32_t jx = (int32_t)asuint(x), ix = jx & 0x7fffffff;
    if (ix >= 0x7f800000) return x + x;
    float h = jx < 0 ? -0.5f : 0.5f;
    if (ix < 0x41b00000) {  // |x| < 22
        if (ix < 0x31800000) return x;  // |x| < 2^-28
        float t = expm1f(ffabs(x));
        if (ix < 0x3f800000) return h * (2.0f * t - t * t / (t + 1.0f));
        return h * (t + t / (t + 1.0f));
    }
    if (ix <= 0x42b1717f) return h * expf(ffabs(x));
    if (ix <= 0x42b2d4fc) {
        float w = expf(0.5f * ffabs(x));
        float t = h * w;
        return t * w;
    }
    return x * 1.0e37f;
}

// ---------------------------------------------------------------------------------------------------------------
// atanhf (e_atanhf.c) over log1pf (s_log1pf.c)
WFLM_FN float log1pf(float x) {
    const float ln2_hi = asfloat(0x3f317180u), ln2_lo = asfloat(0x3717f7d1u);
    const float Lp1 = asfloat(0x3f2aaaabu), Lp2 = asfloat(0x3ecccccdu), Lp3 = asfloat(0x3e924925u), Lp4 = asfloat(0x3e638e29u),
                Lp5 = asfloat(0x3e3a3325u), Lp6 = asfloat(0x3e1cd04fu), Lp7 = asfloat(0x3e178897u);
    float hfsq, f = 0, c = 0, s, z, R, u;
    int32_t k, hx, hu = 0, ax;
    hx = (int32_t)asuint(x);
    ax = hx & 0x7fffffff;
    k = 1;
    if (hx < 0x3ed413d7) {  // x < 0.41422
        if (ax >= 0x3f800000) {
            if (x == -1.0f) return asfloat(0xff800000u);
            return fnan();
        }
        if (ax < 0x31000000) {  // |x| < 2^-29
            if (ax < 0x24800000) return x;
            return x - x * x * 0.5f;
        }
        if (hx > 0 || hx <= (int32_t)0xbe95f61f) {
            k = 0;
            f = x;
            hu = 1;
        }
    }
    if (hx >= 0x7f800000) return x + x;
    if (k != 0) {
        if (hx < 0x5a000000) {
            u = 1.0f + x;
            hu = (int32_t)asuint(u);
            k = (hu >> 23) - 127;
            c = (k > 0) ? 1.0f - (u - x) : x - (u - 1.0f);
            c /= u;
        } else {
            u = x;
            hu = (int32_t)asuint(u);
            k = (hu >> 23) - 127;
            c = 0;
        }
        hu &= 0x007fffff;
        if (hu < 0x3504f7) {
            u = asfloat((uint32_t)hu | 0x3f800000u);
        } else {
            k += 1;
            u = asfloat((uint32_t)hu | 0x3f000000u);
            hu = (0x00800000 - hu) >> 2;
        }
        f = u - 1.0f;
    }
    hfsq = 0.5f * f * f;
    if (hu == 0) {  // |f| < 2^-20
        if (f == 0.0f) {
            if (k == 0) return 0.0f;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        R = hfsq * (1.0f - Lp1 * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    s = f / (2.0f + f);
    z = s * s;
    R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}
WFLM_FN float atanhf(float x) {
    float xa = ffabs(x);
    float t;
    if (xa < 0.5f) {
        if (xa < 0x1.0p-28f) return x;
        t = xa + xa;
        t = 0.5f * log1pf(t + t * xa / (1.0f - xa));
    } else if (xa < 1.0f) {
        t = 0.5f * log1pf((xa + xa) / (1.0f - xa));
    } else {
        if (!(xa <= 1.0f)) return fnan();                          // |x| > 1 or NaN
        return asfloat((asuint(x) & 0x80000000u) | 0x7f800000u);  // x / 0
    }
    return asfloat((asuint(t) & 0x7fffffffu) | (asuint(x) & 0x80000000u));
}

// tanf: sysdeps/ieee754/flt-32/s_tanf.c of glibc 2.35 — the argument reduction of sincosf (reduce_fast / reduce_large above, here
// without fused multiply-add: tanf has no FMA variant in 2.35), the double remainder split into a float head and tail — over fdlibm's
// k_tanf.c kernel.  Verified against the live libm on all 2^32 arguments (oracle/wf_cpu/libm_check.cpp).
WFLM_HD float kernel_tanf(float x, float y, int iy) {
    const float one = 1.0f, pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f, T4 = 8.8632395491e-03f,
                T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f, T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f,
                T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f, T12 = 2.5907305826e-05f;
    float z, r, v, w, s;
    const int32_t hx = (int32_t)asuint(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix < 0x39000000) {  // |x| < 2^-13
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return one / ffabs(x);
            else if (iy == 1) return x;
            else return -1 / x;
        }
    }
    if (ix >= 0x3f2ca140) {  // |x| >= 0.6744
        if (hx < 0) { x = -x; y = -y; }
        z = pio4 - x;
        w = pio4lo - y;
        x = z + w;
        y = 0.0f;
        if (ffabs(x) < 0x1p-13f) return (1 - ((hx >> 30) & 2)) * iy * (1.0f - 2 * iy * x);
    }
    z = x * x;
    w = z * z;
    r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    // -1 / (x + r), accurately
    float a, t;
    z = asfloat(asuint(w) & 0xfffff000u);
    v = r - (z - x);
    t = a = -1.0f / w;
    t = asfloat(asuint(t) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}
WFLM_FN float tanf(float x) {
    const int32_t hx = (int32_t)asuint(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix <= 0x3f490fda) return kernel_tanf(x, 0.0f, 1);   // |x| ~< pi/4
    if (ix > 0x7f7fffff) return x - x;                      // tan(Inf or NaN) is NaN
    double xd = x;
    int n;
    if (abstop12(x) <= 0x42e) {  // |x| < 120: reduce_fast, multiply and subtract rounded separately
        double r = xd * 0x1.45f306dc9c883p+23;
        n = ((int32_t)r + 0x800000) >> 24;
        xd = xd - (double)n * 0x1.921fb54442d18p+0;
    } else {
        xd = reduce_large(asuint(x), &n);
        if (hx < 0) xd = -xd;
    }
    const float y0 = (float)xd, y1 = (float)(xd - (double)y0);
    return kernel_tanf(y0, y1, 1 - ((2 * n) & 2));   // 1: n even, -1: n odd
}

}  // namespace glibc235
