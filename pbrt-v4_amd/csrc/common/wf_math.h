// wf_math.h — scalar/vector/spectral arithmetic shared by the host scene builder (g++) and the HIP
// kernels (hipcc, gfx950).  Every function restates the arithmetic of the reference's CPU ("host")
// branch operation-for-operation — same association order, same use of fma — because hash-seeded
// decisions downstream amplify 1-ulp differences (SURVEY.md §8c caveats 3 and 5).  Build both sides
// with -ffp-contract=off: the only fused operations are the explicit wf::fma calls.
//
// Reference map (all under /root/reference/src/pbrt/):
//   util/float.h      NextFloatUp/Down, gamma, directed rounding (host branch: NextFloat*(a op b))
//   util/math.h       DifferenceOfProducts, SumOfProducts, Lerp, SafeSqrt, FastExp, EvaluatePolynomial
//   util/vecmath.h    Tuple ops, Dot (Vector: plain sum; with a Normal: FMA + SumOfProducts), Cross,
//                     CoordinateSystem, Frame, Interval-backed Point3fi, Bounds3, DirectionCone, Octahedral
//   util/hash.h       MurmurHash64A, MixBits;  util/rng.h  PCG32
//   util/spectrum.h   SampledSpectrum (4 samples), SampledWavelengths
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
// always_inline: a kernel must never fall back to real calls (the by-value scene struct would be spilled to
// scratch and the callee would run on a private stack)
#define WF_HD __host__ __device__ inline __attribute__((always_inline))
// the few deliberately out-of-line device functions (image-texture filtering): leaf functions whose arguments are
// plain pointers and scalars, so nothing of the by-value scene struct is forced into memory.  Inlining them at every
// node of the texture-graph template multiplies the material kernels' code by ~100 (40-minute compiles).
#define WF_NI __host__ __device__ inline __attribute__((noinline))
#else
#define WF_HD inline
#define WF_NI inline
#endif
#include "wf_libm.h"

namespace wf {

// ---------------------------------------------------------------------------------------------
// constants (util/math.h:24-42, util/float.h:43-50)
constexpr float Pi = 3.14159265358979323846f;
constexpr float InvPi = 0.31830988618379067154f;
constexpr float Inv2Pi = 0.15915494309189533577f;
constexpr float Inv4Pi = 0.07957747154594766788f;
constexpr float PiOver2 = 1.57079632679489661923f;
constexpr float PiOver4 = 0.78539816339744830961f;
constexpr float Sqrt2 = 1.41421356237309504880f;
constexpr float ShadowEpsilon = 0.0001f;
constexpr float OneMinusEpsilon = 0x1.fffffep-1f;
constexpr float MachineEpsilon = 0x1p-24f;  // numeric_limits<float>::epsilon() * 0.5
#define WF_INFINITY __builtin_huge_valf()

WF_HD constexpr float gamma(int n) { return (n * MachineEpsilon) / (1 - n * MachineEpsilon); }

// ---------------------------------------------------------------------------------------------
// bit casts / float stepping (util/float.h:101-196)
WF_HD uint32_t FloatToBits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
#endif
}
WF_HD float BitsToFloat(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    std::memcpy(&f, &u, 4);
    return f;
#endif
}
WF_HD int Exponent(float v) { return (int)(FloatToBits(v) >> 23) - 127; }
WF_HD bool IsNaN(float v) { return v != v; }
WF_HD bool IsInf(float v) { return (FloatToBits(v) & 0x7fffffffu) == 0x7f800000u; }
WF_HD bool IsFinite(float v) { return (FloatToBits(v) & 0x7f800000u) != 0x7f800000u; }

WF_HD float NextFloatUp(float v) {
    if (IsInf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = FloatToBits(v);
    if (v >= 0) ++ui; else --ui;
    return BitsToFloat(ui);
}
WF_HD float NextFloatDown(float v) {
    if (IsInf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = FloatToBits(v);
    if (v > 0) --ui; else ++ui;
    return BitsToFloat(ui);
}

// ---------------------------------------------------------------------------------------------
// elementary functions.  fma/sqrt/div are IEEE-exact on both sides.  For the transcendental
// functions the host uses libm's float routines (as the reference does); the device evaluates
// wf_libm.h, the operation-for-operation restatement of this image's glibc 2.35 float routines
// (bit-identical to the live libm on all 2^32 arguments: oracle/wf_cpu/libm_check.cpp), so that the
// device and the reference agree to the last bit, not "almost always".
WF_HD float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
WF_HD float sqrt(float x) { return ::sqrtf(x); }
WF_HD float abs(float x) { return ::fabsf(x); }
WF_HD float floor(float x) { return ::floorf(x); }
WF_HD float ceil(float x) { return ::ceilf(x); }
WF_HD float copysign(float a, float b) { return ::copysignf(a, b); }
WF_HD float fmin(float a, float b) { return b < a ? b : a; }  // std::min semantics
WF_HD float fmax(float a, float b) { return a < b ? b : a; }  // std::max semantics
#if defined(__HIP_DEVICE_COMPILE__)
WF_HD float sin(float x) { return glibc235::sinf(x); }
WF_HD float cos(float x) { return glibc235::cosf(x); }
WF_HD float asin(float x) { return glibc235::asinf(x); }
WF_HD float acos(float x) { return glibc235::acosf(x); }
WF_HD float atan(float x) { return glibc235::atanf(x); }
WF_HD float atan2(float y, float x) { return glibc235::atan2f(y, x); }
WF_HD float exp(float x) { return glibc235::expf(x); }
WF_HD float log(float x) { return glibc235::logf(x); }
WF_HD float cosh(float x) { return glibc235::coshf(x); }
WF_HD float sinh(float x) { return glibc235::sinhf(x); }
WF_HD float atanh(float x) { return glibc235::atanhf(x); }
WF_HD float tan(float x) { return glibc235::tanf(x); }
// pow has no call site in the device path (host-side scene set-up only)
WF_HD long lround(float x) { return (long)::roundf(x); }
#else
WF_HD float sin(float x) { return std::sin(x); }
WF_HD float cos(float x) { return std::cos(x); }
WF_HD float asin(float x) { return std::asin(x); }
WF_HD float acos(float x) { return std::acos(x); }
WF_HD float atan(float x) { return std::atan(x); }
WF_HD float atan2(float y, float x) { return std::atan2(y, x); }
WF_HD float exp(float x) { return std::exp(x); }
WF_HD float log(float x) { return std::log(x); }
WF_HD float cosh(float x) { return std::cosh(x); }
WF_HD float sinh(float x) { return std::sinh(x); }
WF_HD float atanh(float x) { return std::atanh(x); }
WF_HD float tan(float x) { return std::tan(x); }
WF_HD long lround(float x) { return std::lround(x); }
#endif

WF_HD float Sqr(float v) { return v * v; }
WF_HD float Lerp(float x, float a, float b) { return (1 - x) * a + x * b; }
WF_HD float Clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
WF_HD int Clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
WF_HD float SafeSqrt(float x) { return sqrt(fmax(0.f, x)); }
WF_HD float SafeASin(float x) { return asin(Clamp(x, -1.f, 1.f)); }
WF_HD float SafeACos(float x) { return acos(Clamp(x, -1.f, 1.f)); }
WF_HD float Radians(float deg) { return (Pi / 180) * deg; }
WF_HD float Degrees(float rad) { return (180 / Pi) * rad; }

// util/math.h:569-583
WF_HD float DifferenceOfProducts(float a, float b, float c, float d) {
    float cd = c * d;
    float dop = fma(a, b, -cd);
    float err = fma(-c, d, cd);
    return dop + err;
}
WF_HD float SumOfProducts(float a, float b, float c, float d) {
    float cd = c * d;
    float sop = fma(a, b, cd);
    float err = fma(c, d, -cd);
    return sop + err;
}
// EvaluatePolynomial(t, c0, c1, c2, c3) = fma(t, fma(t, fma(t, c3, c2), c1), c0)  (util/math.h:329-337)
WF_HD float Poly3(float t, float c0, float c1, float c2, float c3) {
    return fma(t, fma(t, fma(t, c3, c2), c1), c0);
}
WF_HD float Poly2(float t, float c0, float c1, float c2) { return fma(t, fma(t, c2, c1), c0); }

// util/math.h:450-474 (host branch: polynomial 2^f with exponent splice, NOT __expf)
WF_HD float FastExp(float x) {
    float xp = x * 1.442695041f;
    float fxp = floor(xp), f = xp - fxp;
    int i = (int)fxp;
    float twoToF = Poly3(f, 1.f, 0.695556856f, 0.226173572f, 0.0781455737f);
    int exponent = Exponent(twoToF) + i;
    if (exponent < -126) return 0;
    if (exponent > 127) return WF_INFINITY;
    uint32_t bits = FloatToBits(twoToF);
    bits &= 0b10000000011111111111111111111111u;
    bits |= (uint32_t)(exponent + 127) << 23;
    return BitsToFloat(bits);
}
WF_HD float Gaussian(float x, float mu, float sigma) {
    return 1 / sqrt(2 * Pi * sigma * sigma) * FastExp(-Sqr(x - mu) / (2 * sigma * sigma));
}

// util/math.h:506-519
template <typename Pred>
WF_HD int FindInterval(int sz, const Pred &pred) {
    int size = sz - 2, first = 1;
    while (size > 0) {
        int half = size >> 1, middle = first + half;
        bool r = pred(middle);
        first = r ? middle + 1 : first;
        size = r ? size - (half + 1) : half;
    }
    return Clamp(first - 1, 0, sz - 2);
}

// ---------------------------------------------------------------------------------------------
// vectors.  V3 = Vector3f/Point3f (plain arithmetic);  N3 = Normal3f (Dot uses FMA+SumOfProducts).
struct V2 { float x, y; };
struct V3 {
    float x, y, z;
    WF_HD float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    WF_HD float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
struct N3 { float x, y, z; };

WF_HD V3 mk3(float x, float y, float z) { return V3{x, y, z}; }
WF_HD N3 mkn(float x, float y, float z) { return N3{x, y, z}; }
WF_HD V3 toV(N3 n) { return V3{n.x, n.y, n.z}; }
WF_HD N3 toN(V3 v) { return N3{v.x, v.y, v.z}; }
WF_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
WF_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
WF_HD V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
WF_HD V3 operator*(V3 a, float s) { return {s * a.x, s * a.y, s * a.z}; }
WF_HD V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
WF_HD V3 operator/(V3 a, float d) { return {a.x / d, a.y / d, a.z / d}; }
WF_HD bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
WF_HD N3 operator+(N3 a, N3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
WF_HD N3 operator-(N3 a, N3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
WF_HD N3 operator-(N3 a) { return {-a.x, -a.y, -a.z}; }
WF_HD N3 operator*(N3 a, float s) { return {s * a.x, s * a.y, s * a.z}; }
WF_HD N3 operator*(float s, N3 a) { return {s * a.x, s * a.y, s * a.z}; }
WF_HD N3 operator/(N3 a, float d) { return {a.x / d, a.y / d, a.z / d}; }
WF_HD bool operator==(N3 a, N3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
WF_HD bool IsZero(N3 n) { return n.x == 0 && n.y == 0 && n.z == 0; }
WF_HD V3 Abs(V3 v) { return {abs(v.x), abs(v.y), abs(v.z)}; }
WF_HD N3 Abs(N3 v) { return {abs(v.x), abs(v.y), abs(v.z)}; }
WF_HD float Dot(V3 v, V3 w) { return v.x * w.x + v.y * w.y + v.z * w.z; }
WF_HD float Dot(N3 n, V3 v) { return fma(n.x, v.x, SumOfProducts(n.y, v.y, n.z, v.z)); }
WF_HD float Dot(V3 v, N3 n) { return fma(n.x, v.x, SumOfProducts(n.y, v.y, n.z, v.z)); }
WF_HD float Dot(N3 a, N3 b) { return fma(a.x, b.x, SumOfProducts(a.y, b.y, a.z, b.z)); }
WF_HD float AbsDot(V3 a, V3 b) { return abs(Dot(a, b)); }
WF_HD float AbsDot(N3 a, V3 b) { return abs(Dot(a, b)); }
WF_HD float AbsDot(V3 a, N3 b) { return abs(Dot(a, b)); }
WF_HD float AbsDot(N3 a, N3 b) { return abs(Dot(a, b)); }
WF_HD float LengthSquared(V3 v) { return Sqr(v.x) + Sqr(v.y) + Sqr(v.z); }
WF_HD float LengthSquared(N3 v) { return Sqr(v.x) + Sqr(v.y) + Sqr(v.z); }
WF_HD float Length(V3 v) { return sqrt(LengthSquared(v)); }
WF_HD float Length(N3 v) { return sqrt(LengthSquared(v)); }
WF_HD V3 Normalize(V3 v) { return v / Length(v); }
WF_HD N3 Normalize(N3 v) { return v / Length(v); }
WF_HD float DistanceSquared(V3 a, V3 b) { return LengthSquared(a - b); }
WF_HD float Distance(V3 a, V3 b) { return Length(a - b); }
WF_HD V3 Cross(V3 v, V3 w) {
    return {DifferenceOfProducts(v.y, w.z, v.z, w.y), DifferenceOfProducts(v.z, w.x, v.x, w.z),
            DifferenceOfProducts(v.x, w.y, v.y, w.x)};
}
WF_HD V3 Cross(N3 v, V3 w) { return Cross(toV(v), w); }
WF_HD V3 Cross(V3 v, N3 w) { return Cross(v, toV(w)); }
WF_HD N3 FaceForward(N3 n, V3 v) { return (Dot(n, v) < 0.f) ? -n : n; }
WF_HD N3 FaceForward(N3 n, N3 v) { return (Dot(n, v) < 0.f) ? -n : n; }
WF_HD V3 FaceForward(V3 n, V3 v) { return (Dot(n, v) < 0.f) ? -n : n; }
WF_HD V3 FaceForward(V3 n, N3 v) { return (Dot(n, v) < 0.f) ? -n : n; }
WF_HD float MaxComponentValue(V3 v) { return fmax(fmax(v.x, v.y), v.z); }
WF_HD int MaxComponentIndex(V3 t) { return (t.x > t.y) ? ((t.x > t.z) ? 0 : 2) : ((t.y > t.z) ? 1 : 2); }
WF_HD V3 Permute(V3 v, int x, int y, int z) { return {v[x], v[y], v[z]}; }
WF_HD V3 GramSchmidt(V3 v, V3 w) { return v - Dot(v, w) * w; }
// util/vecmath.h:1006-1022
WF_HD void CoordinateSystem(V3 v1, V3 *v2, V3 *v3) {
    float sign = copysign(1.f, v1.z);
    float a = -1 / (sign + v1.z);
    float b = v1.x * v1.y * a;
    *v2 = V3{1 + sign * Sqr(v1.x) * a, sign * b, -sign * v1.x};
    *v3 = V3{b, sign + Sqr(v1.y) * a, -v1.y};
}
// util/vecmath.h:970-991
WF_HD float AngleBetween(V3 v1, V3 v2) {
    if (Dot(v1, v2) < 0) return Pi - 2 * SafeASin(Length(v1 + v2) / 2);
    else return 2 * SafeASin(Length(v2 - v1) / 2);
}
WF_HD float SphericalTriangleArea(V3 a, V3 b, V3 c) {
    return abs(2 * atan2(Dot(a, Cross(b, c)), 1 + Dot(a, b) + Dot(a, c) + Dot(b, c)));
}

// EqualAreaSquareToSphere / EqualAreaSphereToSquare, util/math.cpp:23-44,47-94
WF_HD V3 EqualAreaSquareToSphere(V2 p) {
    float u = 2 * p.x - 1, v = 2 * p.y - 1;
    float up = abs(u), vp = abs(v);
    float signedDistance = 1 - (up + vp);
    float d = abs(signedDistance);
    float r = 1 - d;
    float phi = (r == 0 ? 1 : (vp - up) / r + 1) * Pi / 4;
    float z = copysign(1 - Sqr(r), signedDistance);
    float cosPhi = copysign(cos(phi), u);
    float sinPhi = copysign(sin(phi), v);
    return V3{cosPhi * r * SafeSqrt(2 - Sqr(r)), sinPhi * r * SafeSqrt(2 - Sqr(r)), z};
}
WF_HD V2 EqualAreaSphereToSquare(V3 d) {
    float x = abs(d.x), y = abs(d.y), z = abs(d.z);
    float r = SafeSqrt(1 - z);
    float a = fmax(x, y), b = fmin(x, y);
    b = a == 0 ? 0 : b / a;
    const float t1 = 0.406758566246788489601959989e-5f;
    const float t2 = 0.636226545274016134946890922156f;
    const float t3 = 0.61572017898280213493197203466e-2f;
    const float t4 = -0.247333733281268944196501420480f;
    const float t5 = 0.881770664775316294736387951347e-1f;
    const float t6 = 0.419038818029165735901852432784e-1f;
    const float t7 = -0.251390972343483509333252996350e-1f;
    // EvaluatePolynomial(b, t1..t7) = fma(b, EvaluatePolynomial(b, t2..t7), t1)  (util/math.h:329-337)
    float phi = fma(b, fma(b, fma(b, fma(b, fma(b, fma(b, t7, t6), t5), t4), t3), t2), t1);
    if (x < y) phi = 1 - phi;
    float v = phi * r;
    float u = r - v;
    if (d.z < 0) {
        float t = u; u = v; v = t;
        u = 1 - u;
        v = 1 - v;
    }
    u = copysign(u, d.x);
    v = copysign(v, d.y);
    return V2{0.5f * (u + 1), 0.5f * (v + 1)};
}

// Frame (util/vecmath.h:1847-1927)
struct Frame {
    V3 x, y, z;
    WF_HD static Frame FromZ(V3 z) { Frame f; f.z = z; CoordinateSystem(z, &f.x, &f.y); return f; }
    WF_HD static Frame FromXZ(V3 x, V3 z) { Frame f; f.x = x; f.y = Cross(z, x); f.z = z; return f; }
    WF_HD V3 ToLocal(V3 v) const { return {Dot(v, x), Dot(v, y), Dot(v, z)}; }
    WF_HD N3 ToLocal(N3 n) const { return {Dot(n, x), Dot(n, y), Dot(n, z)}; }
    WF_HD V3 FromLocal(V3 v) const { return v.x * x + v.y * y + v.z * z; }
};

// ---------------------------------------------------------------------------------------------
// Point3fi: per-component [lo,hi] interval (util/math.h:818-950 Interval, host rounding branch)
struct P3i {
    V3 lo, hi;
    WF_HD V3 mid() const { return {(lo.x + hi.x) / 2, (lo.y + hi.y) / 2, (lo.z + hi.z) / 2}; }
    WF_HD V3 err() const { return {(hi.x - lo.x) / 2, (hi.y - lo.y) / 2, (hi.z - lo.z) / 2}; }
    WF_HD bool exact() const { return lo.x == hi.x && lo.y == hi.y && lo.z == hi.z; }
};
WF_HD void IntervalFromValueAndError(float v, float e, float *lo, float *hi) {
    if (e == 0) { *lo = *hi = v; }
    else { *lo = NextFloatDown(v - e); *hi = NextFloatUp(v + e); }
}
WF_HD P3i MakeP3i(V3 p, V3 e) {
    P3i r;
    IntervalFromValueAndError(p.x, e.x, &r.lo.x, &r.hi.x);
    IntervalFromValueAndError(p.y, e.y, &r.lo.y, &r.hi.y);
    IntervalFromValueAndError(p.z, e.z, &r.lo.z, &r.hi.z);
    return r;
}
WF_HD P3i MakeP3i(V3 p) { return P3i{p, p}; }

WF_HD V3 SampleUniformSphere(V2 u) {
    float z = 1 - 2 * u.x;
    float r = SafeSqrt(1 - Sqr(z));
    float phi = 2 * Pi * u.y;
    return {r * cos(phi), r * sin(phi), z};
}
WF_HD V3 SphericalDirection(float sinTheta, float cosTheta, float phi) {
    return V3{Clamp(sinTheta, -1.f, 1.f) * cos(phi), Clamp(sinTheta, -1.f, 1.f) * sin(phi), Clamp(cosTheta, -1.f, 1.f)};
}

// Interval (util/math.h:818-1130) with the HOST rounding branch of util/float.h:204-305 (NextFloatUp/Down of the
// rounded-to-nearest result) on both sides, so the quadric tests decide exactly as the reference's CPU build does.
struct Ivl {
    float lo, hi;
    WF_HD Ivl() : lo(0), hi(0) {}
    WF_HD explicit Ivl(float v) : lo(v), hi(v) {}
    WF_HD Ivl(float a, float b) : lo(b < a ? b : a), hi(a < b ? b : a) {}  // std::min / std::max of the two
    WF_HD static Ivl FromValueAndError(float v, float e) {
        Ivl i;
        IntervalFromValueAndError(v, e, &i.lo, &i.hi);
        return i;
    }
    WF_HD float mid() const { return (lo + hi) / 2; }
    WF_HD bool contains0() const { return 0 >= lo && 0 <= hi; }
    WF_HD bool operator==(Ivl o) const { return lo == o.lo && hi == o.hi; }
    WF_HD Ivl operator-() const { return Ivl(-hi, -lo); }
    WF_HD Ivl operator+(Ivl i) const { return Ivl(NextFloatDown(lo + i.lo), NextFloatUp(hi + i.hi)); }
    WF_HD Ivl operator-(Ivl i) const { return Ivl(NextFloatDown(lo + -i.hi), NextFloatUp(hi + -i.lo)); }
    WF_HD static float Min4(float a, float b, float c, float d) { float m = a; if (b < m) m = b; if (c < m) m = c; if (d < m) m = d; return m; }
    WF_HD static float Max4(float a, float b, float c, float d) { float m = a; if (m < b) m = b; if (m < c) m = c; if (m < d) m = d; return m; }
    WF_HD Ivl operator*(Ivl i) const {
        float p0 = lo * i.lo, p1 = hi * i.lo, p2 = lo * i.hi, p3 = hi * i.hi;
        return Ivl(Min4(NextFloatDown(p0), NextFloatDown(p1), NextFloatDown(p2), NextFloatDown(p3)),
                   Max4(NextFloatUp(p0), NextFloatUp(p1), NextFloatUp(p2), NextFloatUp(p3)));
    }
    WF_HD Ivl operator/(Ivl i) const {
        if (i.contains0()) return Ivl(-WF_INFINITY, WF_INFINITY);
        float q0 = lo / i.lo, q1 = hi / i.lo, q2 = lo / i.hi, q3 = hi / i.hi;
        return Ivl(Min4(NextFloatDown(q0), NextFloatDown(q1), NextFloatDown(q2), NextFloatDown(q3)),
                   Max4(NextFloatUp(q0), NextFloatUp(q1), NextFloatUp(q2), NextFloatUp(q3)));
    }
};
WF_HD Ivl operator*(float f, Ivl i) {
    if (f > 0) return Ivl(NextFloatDown(f * i.lo), NextFloatUp(f * i.hi));
    return Ivl(NextFloatDown(f * i.hi), NextFloatUp(f * i.lo));
}
WF_HD Ivl Sqr(Ivl i) {
    float alow = abs(i.lo), ahigh = abs(i.hi);
    if (alow > ahigh) { float t = alow; alow = ahigh; ahigh = t; }
    if (i.contains0()) return Ivl(0, NextFloatUp(ahigh * ahigh));
    return Ivl(NextFloatDown(alow * alow), NextFloatUp(ahigh * ahigh));
}
WF_HD Ivl Sqrt(Ivl i) {
    float l = NextFloatDown(sqrt(i.lo));
    return Ivl(0 < l ? l : 0.f, NextFloatUp(sqrt(i.hi)));  // SqrtRoundDown clamps at 0 (std::max<Float>(0, .))
}
struct Ivl3 { Ivl x, y, z; };

// ray.h:75-113
WF_HD V3 OffsetRayOrigin(const P3i &pi, N3 n, V3 w) {
    float d = Dot(Abs(n), pi.err());
    V3 offset = d * toV(n);
    if (Dot(w, n) < 0) offset = -offset;
    V3 po = pi.mid() + offset;
    for (int i = 0; i < 3; ++i) {
        if (offset[i] > 0) po[i] = NextFloatUp(po[i]);
        else if (offset[i] < 0) po[i] = NextFloatDown(po[i]);
    }
    return po;
}
struct RayOD { V3 o, d; };
WF_HD RayOD SpawnRay(const P3i &pi, N3 n, V3 d) { return {OffsetRayOrigin(pi, n, d), d}; }
WF_HD RayOD SpawnRayTo(const P3i &pFrom, N3 n, V3 pTo) {
    V3 d = pTo - pFrom.mid();
    return SpawnRay(pFrom, n, d);
}
WF_HD RayOD SpawnRayTo(const P3i &pFrom, N3 nFrom, const P3i &pTo, N3 nTo) {
    V3 pf = OffsetRayOrigin(pFrom, nFrom, pTo.mid() - pFrom.mid());
    V3 pt = OffsetRayOrigin(pTo, nTo, pf - pTo.mid());
    return {pf, pt - pf};
}

// ---------------------------------------------------------------------------------------------
// Bounds3f (util/vecmath.h:1251-1420)
struct B3 {
    // default = empty box: numeric_limits max()/lowest() (util/vecmath.h:1257-1262)
    V3 pMin{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, pMax{-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    WF_HD V3 Diagonal() const { return pMax - pMin; }
    WF_HD float SurfaceArea() const {
        V3 d = Diagonal();
        return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    WF_HD int MaxDimension() const {
        V3 d = Diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        else return 2;
    }
    WF_HD V3 Offset(V3 p) const {
        V3 o = p - pMin;
        if (pMax.x > pMin.x) o.x /= pMax.x - pMin.x;
        if (pMax.y > pMin.y) o.y /= pMax.y - pMin.y;
        if (pMax.z > pMin.z) o.z /= pMax.z - pMin.z;
        return o;
    }
    WF_HD bool IsEmpty() const { return pMin.x >= pMax.x || pMin.y >= pMax.y || pMin.z >= pMax.z; }
};
WF_HD B3 Union(const B3 &b, V3 p) {
    B3 r;
    r.pMin = {fmin(b.pMin.x, p.x), fmin(b.pMin.y, p.y), fmin(b.pMin.z, p.z)};
    r.pMax = {fmax(b.pMax.x, p.x), fmax(b.pMax.y, p.y), fmax(b.pMax.z, p.z)};
    return r;
}
WF_HD B3 Union(const B3 &a, const B3 &b) {
    B3 r;
    r.pMin = {fmin(a.pMin.x, b.pMin.x), fmin(a.pMin.y, b.pMin.y), fmin(a.pMin.z, b.pMin.z)};
    r.pMax = {fmax(a.pMax.x, b.pMax.x), fmax(a.pMax.y, b.pMax.y), fmax(a.pMax.z, b.pMax.z)};
    return r;
}
WF_HD bool Inside(V3 p, const B3 &b) {
    return p.x >= b.pMin.x && p.x <= b.pMax.x && p.y >= b.pMin.y && p.y <= b.pMax.y && p.z >= b.pMin.z &&
           p.z <= b.pMax.z;
}

// DirectionCone (util/vecmath.h:1785-1840)
struct DirectionCone { V3 w; float cosTheta; };
WF_HD DirectionCone BoundSubtendedDirections(const B3 &b, V3 p) {
    V3 pCenter = (b.pMin + b.pMax) / 2;
    float radius = Inside(pCenter, b) ? Distance(pCenter, b.pMax) : 0;
    if (DistanceSquared(p, pCenter) < Sqr(radius)) return DirectionCone{V3{0, 0, 1}, -1.f};
    V3 w = Normalize(pCenter - p);
    float sin2ThetaMax = Sqr(radius) / DistanceSquared(pCenter, p);
    float cosThetaMax = SafeSqrt(1 - sin2ThetaMax);
    return DirectionCone{Normalize(w), cosThetaMax};
}

// OctahedralVector (util/vecmath.h:1733-1782)
WF_HD uint16_t OctEncode(float f) { return (uint16_t)::roundf(Clamp((f + 1) / 2, 0.f, 1.f) * 65535.f); }
WF_HD void OctahedralFromVector(V3 v, uint16_t *ox, uint16_t *oy) {
    v = v / (abs(v.x) + abs(v.y) + abs(v.z));
    if (v.z >= 0) { *ox = OctEncode(v.x); *oy = OctEncode(v.y); }
    else {
        *ox = OctEncode((1 - abs(v.y)) * copysign(1.f, v.x));
        *oy = OctEncode((1 - abs(v.x)) * copysign(1.f, v.y));
    }
}
WF_HD V3 OctahedralToVector(uint16_t ox, uint16_t oy) {
    V3 v;
    v.x = -1 + 2 * (ox / 65535.f);
    v.y = -1 + 2 * (oy / 65535.f);
    v.z = 1 - (abs(v.x) + abs(v.y));
    if (v.z < 0) {
        float xo = v.x;
        v.x = (1 - abs(v.y)) * copysign(1.f, xo);
        v.y = (1 - abs(xo)) * copysign(1.f, v.y);
    }
    return Normalize(v);
}

// ---------------------------------------------------------------------------------------------
// hashing (util/hash.h)
WF_HD uint64_t MurmurHash64A(const unsigned char *key, size_t len, uint64_t seed) {
    const uint64_t m = 0xc6a4a7935bd1e995ull;
    const int r = 47;
    uint64_t h = seed ^ (len * m);
    const unsigned char *end = key + 8 * (len / 8);
    while (key != end) {
        uint64_t k = 0;
        for (int i = 0; i < 8; ++i) k |= (uint64_t)key[i] << (8 * i);
        key += 8;
        k *= m; k ^= k >> r; k *= m;
        h ^= k; h *= m;
    }
    switch (len & 7) {
    case 7: h ^= uint64_t(key[6]) << 48;
    case 6: h ^= uint64_t(key[5]) << 40;
    case 5: h ^= uint64_t(key[4]) << 32;
    case 4: h ^= uint64_t(key[3]) << 24;
    case 3: h ^= uint64_t(key[2]) << 16;
    case 2: h ^= uint64_t(key[1]) << 8;
    case 1: h ^= uint64_t(key[0]); h *= m;
    };
    h ^= h >> r; h *= m; h ^= h >> r;
    return h;
}
// Hash(args...) packs the arguments back to back; these helpers cover the argument shapes the hot path
// uses: (int,int) samplers.h:261; (Point3f) lights.h:498; (Point3f,Vector3f) cpu/primitive.cpp:60; etc.
WF_HD uint64_t HashWords(const uint32_t *w, int nwords) {
    unsigned char buf[64];
    for (int i = 0; i < nwords; ++i)
        for (int b = 0; b < 4; ++b) buf[4 * i + b] = (unsigned char)(w[i] >> (8 * b));
    return MurmurHash64A(buf, 4 * (size_t)nwords, 0);
}
WF_HD uint64_t Hash2i(int a, int b) { uint32_t w[2] = {(uint32_t)a, (uint32_t)b}; return HashWords(w, 2); }
WF_HD uint64_t Hash3f(V3 p) { uint32_t w[3] = {FloatToBits(p.x), FloatToBits(p.y), FloatToBits(p.z)}; return HashWords(w, 3); }
WF_HD uint64_t Hash6f(V3 a, V3 b) {
    uint32_t w[6] = {FloatToBits(a.x), FloatToBits(a.y), FloatToBits(a.z), FloatToBits(b.x), FloatToBits(b.y), FloatToBits(b.z)};
    return HashWords(w, 6);
}
WF_HD uint64_t MixBits(uint64_t v) {
    v ^= (v >> 31); v *= 0x7fb5d329728ea185ull;
    v ^= (v >> 27); v *= 0x81dadef4bc2dd44dull;
    v ^= (v >> 33);
    return v;
}
// util/math.h:727-755
WF_HD int PermutationElement(uint32_t i, uint32_t l, uint32_t p) {
    uint32_t w = l - 1;
    w |= w >> 1;
    w |= w >> 2;
    w |= w >> 4;
    w |= w >> 8;
    w |= w >> 16;
    do {
        i ^= p;
        i *= 0xe170893d;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8;
        i *= 0x0929eb3f;
        i ^= p >> 23;
        i ^= (i & w) >> 1;
        i *= 1 | p >> 27;
        i *= 0x6935fa69;
        i ^= (i & w) >> 11;
        i *= 0x74dcb303;
        i ^= (i & w) >> 2;
        i *= 0x9e501cc3;
        i ^= (i & w) >> 2;
        i *= 0xc860a3df;
        i &= w;
        i ^= i >> 5;
    } while (i >= l);
    return (int)((i + p) % l);
}
WF_HD float HashToFloat(uint64_t h) { return uint32_t(h) * 0x1p-32f; }

// PCG32 (util/rng.h:22-172)
struct RNG {
    uint64_t state, inc;
    WF_HD RNG() : state(0x853c49e6748fea9bULL), inc(0xda3e39cb94b95bdbULL) {}
    WF_HD RNG(uint64_t seqIndex, uint64_t offset) { SetSequence(seqIndex, offset); }
    WF_HD void SetSequence(uint64_t sequenceIndex) { SetSequence(sequenceIndex, MixBits(sequenceIndex)); }  // util/rng.h:43-45
    // util/rng.h:137-150
    WF_HD void Advance(int64_t idelta) {
        uint64_t curMult = 0x5851f42d4c957f2dULL, curPlus = inc, accMult = 1u;
        uint64_t accPlus = 0u, delta = (uint64_t)idelta;
        while (delta > 0) {
            if (delta & 1) {
                accMult *= curMult;
                accPlus = accPlus * curMult + curPlus;
            }
            curPlus = (curMult + 1) * curPlus;
            curMult *= curMult;
            delta /= 2;
        }
        state = accMult * state + accPlus;
    }
    WF_HD void SetSequence(uint64_t sequenceIndex, uint64_t seed) {
        state = 0u;
        inc = (sequenceIndex << 1u) | 1u;
        Uniform32();
        state += seed;
        Uniform32();
    }
    WF_HD uint32_t Uniform32() {
        uint64_t oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
        uint32_t rot = (uint32_t)(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    WF_HD float UniformFloat() { return fmin(OneMinusEpsilon, Uniform32() * 0x1p-32f); }
    // RNG::operator- (util/rng.h:152-170): how many draws this generator is ahead of `other` (same sequence)
    WF_HD int64_t operator-(const RNG &other) const {
        uint64_t curMult = 0x5851f42d4c957f2dULL, curPlus = inc, curState = other.state;
        uint64_t theBit = 1u, distance = 0u;
        while (state != curState) {
            if ((state & theBit) != (curState & theBit)) {
                curState = curState * curMult + curPlus;
                distance |= theBit;
            }
            theBit <<= 1;
            curPlus = (curMult + 1ULL) * curPlus;
            curMult *= curMult;
        }
        return (int64_t)distance;
    }
};

// bit tricks (util/math.h:55-153)
WF_HD uint32_t ReverseBits32(uint32_t n) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
    return n;
}
WF_HD uint64_t LeftShift2(uint64_t x) {
    x &= 0xffffffff;
    x = (x ^ (x << 16)) & 0x0000ffff0000ffff;
    x = (x ^ (x << 8)) & 0x00ff00ff00ff00ff;
    x = (x ^ (x << 4)) & 0x0f0f0f0f0f0f0f0f;
    x = (x ^ (x << 2)) & 0x3333333333333333;
    x = (x ^ (x << 1)) & 0x5555555555555555;
    return x;
}
WF_HD uint64_t EncodeMorton2(uint32_t x, uint32_t y) { return (LeftShift2(y) << 1) | LeftShift2(x); }

// ---------------------------------------------------------------------------------------------
// SampledSpectrum / SampledWavelengths (util/spectrum.h:88-330), NSpectrumSamples = 4
struct S4 {
    float v[4];
    WF_HD float operator[](int i) const { return v[i]; }
    WF_HD float &operator[](int i) { return v[i]; }
    WF_HD explicit operator bool() const { return v[0] != 0 || v[1] != 0 || v[2] != 0 || v[3] != 0; }
    WF_HD float Average() const { float s = v[0]; s += v[1]; s += v[2]; s += v[3]; return s / 4; }
    WF_HD float MaxComponentValue() const { float m = v[0]; m = fmax(m, v[1]); m = fmax(m, v[2]); m = fmax(m, v[3]); return m; }
    WF_HD float MinComponentValue() const { float m = v[0]; m = fmin(m, v[1]); m = fmin(m, v[2]); m = fmin(m, v[3]); return m; }
};
WF_HD S4 S4c(float c) { return S4{{c, c, c, c}}; }
WF_HD S4 operator+(S4 a, S4 b) { return S4{{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}}; }
WF_HD S4 operator-(S4 a, S4 b) { return S4{{a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]}}; }
WF_HD S4 operator*(S4 a, S4 b) { return S4{{a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]}}; }
WF_HD S4 operator/(S4 a, S4 b) { return S4{{a[0] / b[0], a[1] / b[1], a[2] / b[2], a[3] / b[3]}}; }
WF_HD S4 operator*(S4 a, float s) { return S4{{a[0] * s, a[1] * s, a[2] * s, a[3] * s}}; }
WF_HD S4 operator*(float s, S4 a) { return S4{{a[0] * s, a[1] * s, a[2] * s, a[3] * s}}; }
WF_HD S4 operator/(S4 a, float s) { return S4{{a[0] / s, a[1] / s, a[2] / s, a[3] / s}}; }
WF_HD S4 operator-(float a, S4 s) { return S4{{a - s[0], a - s[1], a - s[2], a - s[3]}}; }
WF_HD S4 operator-(S4 s) { return S4{{-s[0], -s[1], -s[2], -s[3]}}; }
WF_HD S4 SafeDiv(S4 a, S4 b) {
    S4 r;
    for (int i = 0; i < 4; ++i) r[i] = (b[i] != 0) ? a[i] / b[i] : 0.f;
    return r;
}
WF_HD S4 Sqrt(S4 s) { return S4{{sqrt(s[0]), sqrt(s[1]), sqrt(s[2]), sqrt(s[3])}}; }
WF_HD S4 ClampS(S4 s, float lo, float hi) { return S4{{Clamp(s[0], lo, hi), Clamp(s[1], lo, hi), Clamp(s[2], lo, hi), Clamp(s[3], lo, hi)}}; }
WF_HD S4 ClampZero(S4 s) { return S4{{fmax(0.f, s[0]), fmax(0.f, s[1]), fmax(0.f, s[2]), fmax(0.f, s[3])}}; }
WF_HD S4 FastExp(S4 s) { return S4{{FastExp(s[0]), FastExp(s[1]), FastExp(s[2]), FastExp(s[3])}}; }
WF_HD S4 Exp(S4 s) { return S4{{exp(s[0]), exp(s[1]), exp(s[2]), exp(s[3])}}; }

struct Wavelengths {
    float lambda[4], pdf[4];
    WF_HD S4 PDF() const { return S4{{pdf[0], pdf[1], pdf[2], pdf[3]}}; }
    WF_HD bool SecondaryTerminated() const { return pdf[1] == 0 && pdf[2] == 0 && pdf[3] == 0; }
    WF_HD void TerminateSecondary() {
        if (SecondaryTerminated()) return;
        pdf[1] = pdf[2] = pdf[3] = 0;
        pdf[0] /= 4;
    }
};
// util/sampling.h:163-171
WF_HD float VisibleWavelengthsPDF(float lambda) {
    if (lambda < 360 || lambda > 830) return 0;
    return 0.0039398042f / Sqr(cosh(0.0072f * (lambda - 538)));
}
WF_HD float SampleVisibleWavelengths(float u) { return 538 - 138.888889f * atanh(0.85691062f - 1.82750197f * u); }
WF_HD Wavelengths SampleVisible(float u) {
    Wavelengths swl;
    for (int i = 0; i < 4; ++i) {
        float up = u + float(i) / 4;
        if (up > 1) up -= 1;
        swl.lambda[i] = SampleVisibleWavelengths(up);
        swl.pdf[i] = VisibleWavelengthsPDF(swl.lambda[i]);
    }
    return swl;
}
WF_HD Wavelengths SampleUniformWavelengths(float u, float lmin = 360.f, float lmax = 830.f) {
    Wavelengths swl;
    swl.lambda[0] = Lerp(u, lmin, lmax);
    float delta = (lmax - lmin) / 4;
    for (int i = 1; i < 4; ++i) {
        swl.lambda[i] = swl.lambda[i - 1] + delta;
        if (swl.lambda[i] > lmax) swl.lambda[i] = lmin + (swl.lambda[i] - lmax);
    }
    for (int i = 0; i < 4; ++i) swl.pdf[i] = 1 / (lmax - lmin);
    return swl;
}

// RGBSigmoidPolynomial (util/color.h:332-364)
WF_HD float SigmoidS(float x) {
    if (IsInf(x)) return x > 0 ? 1.f : 0.f;
    return .5f + x / (2 * sqrt(1 + Sqr(x)));
}
WF_HD float SigmoidPoly(float lambda, float c0, float c1, float c2) { return SigmoidS(Poly2(lambda, c2, c1, c0)); }

// Blackbody (util/spectrum.h:69-83)
WF_HD float Pow5(float v) { float n2 = v * v; return n2 * n2 * v; }
WF_HD float Blackbody(float lambda, float T) {
    if (T <= 0) return 0;
    const float c = 299792458.f, h = 6.62606957e-34f, kb = 1.3806488e-23f;
    float l = lambda * 1e-9f;
    return (2 * h * c * c) / (Pow5(l) * (FastExp((h * c) / (l * kb * T)) - 1));
}

// ---------------------------------------------------------------------------------------------
// sampling routines (util/sampling.h, util/sampling.cpp)
WF_HD V2 SampleUniformDiskConcentric(V2 u) {
    V2 uo{2 * u.x - 1, 2 * u.y - 1};
    if (uo.x == 0 && uo.y == 0) return {0, 0};
    float theta, r;
    if (abs(uo.x) > abs(uo.y)) { r = uo.x; theta = PiOver4 * (uo.y / uo.x); }
    else { r = uo.y; theta = PiOver2 - PiOver4 * (uo.x / uo.y); }
    return {r * cos(theta), r * sin(theta)};
}
WF_HD V3 SampleCosineHemisphere(V2 u) {
    V2 d = SampleUniformDiskConcentric(u);
    float z = SafeSqrt(1 - Sqr(d.x) - Sqr(d.y));
    return {d.x, d.y, z};
}
WF_HD float CosineHemispherePDF(float cosTheta) { return cosTheta * InvPi; }
WF_HD void SampleUniformTriangle(V2 u, float b[3]) {
    float b0, b1;
    if (u.x < u.y) { b0 = u.x / 2; b1 = u.y - b0; }
    else { b1 = u.y / 2; b0 = u.x - b1; }
    b[0] = b0; b[1] = b1; b[2] = 1 - b0 - b1;
}
WF_HD float SampleLinear(float u, float a, float b) {
    if (u == 0 && a == 0) return 0;
    float x = u * (a + b) / (a + sqrt(Lerp(u, Sqr(a), Sqr(b))));
    return fmin(x, OneMinusEpsilon);
}
WF_HD float BilinearPDF(V2 p, const float w[4]) {
    if (p.x < 0 || p.x > 1 || p.y < 0 || p.y > 1) return 0;
    if (w[0] + w[1] + w[2] + w[3] == 0) return 1;
    return 4 * ((1 - p.x) * (1 - p.y) * w[0] + p.x * (1 - p.y) * w[1] + (1 - p.x) * p.y * w[2] + p.x * p.y * w[3]) /
           (w[0] + w[1] + w[2] + w[3]);
}
WF_HD V2 SampleBilinear(V2 u, const float w[4]) {
    V2 p;
    p.y = SampleLinear(u.y, w[0] + w[1], w[2] + w[3]);
    p.x = SampleLinear(u.x, Lerp(p.y, w[0], w[2]), Lerp(p.y, w[1], w[3]));
    return p;
}
// SampleDiscrete for exactly two weights (lightsamplers.h:312), util/sampling.h:79-113
WF_HD int SampleDiscrete2(float w0, float w1, float u, float *pmf, float *uRemapped) {
    float sumWeights = 0;
    sumWeights += w0;
    sumWeights += w1;
    float up = u * sumWeights;
    if (up == sumWeights) up = NextFloatDown(up);
    int offset = 0;
    float sum = 0;
    float w[2] = {w0, w1};
    while (sum + w[offset] <= up) { sum += w[offset++]; }
    *pmf = w[offset] / sumWeights;
    *uRemapped = fmin((up - sum) / w[offset], OneMinusEpsilon);
    return offset;
}
// util/sampling.cpp:28-107
WF_HD void SampleSphericalTriangle(V3 v0, V3 v1, V3 v2, V3 p, V2 u, float b3[3], float *pdf) {
    *pdf = 0;
    b3[0] = b3[1] = b3[2] = 0;
    V3 a = v0 - p, b = v1 - p, c = v2 - p;
    a = Normalize(a); b = Normalize(b); c = Normalize(c);
    V3 n_ab = Cross(a, b), n_bc = Cross(b, c), n_ca = Cross(c, a);
    if (LengthSquared(n_ab) == 0 || LengthSquared(n_bc) == 0 || LengthSquared(n_ca) == 0) return;
    n_ab = Normalize(n_ab); n_bc = Normalize(n_bc); n_ca = Normalize(n_ca);
    float alpha = AngleBetween(n_ab, -n_ca);
    float beta = AngleBetween(n_bc, -n_ab);
    float gam = AngleBetween(n_ca, -n_bc);
    float A_pi = alpha + beta + gam;
    float Ap_pi = Lerp(u.x, Pi, A_pi);
    float A = A_pi - Pi;
    *pdf = (A <= 0) ? 0 : 1 / A;
    float cosAlpha = cos(alpha), sinAlpha = sin(alpha);
    float sinPhi = sin(Ap_pi) * cosAlpha - cos(Ap_pi) * sinAlpha;
    float cosPhi = cos(Ap_pi) * cosAlpha + sin(Ap_pi) * sinAlpha;
    float k1 = cosPhi + cosAlpha;
    float k2 = sinPhi - sinAlpha * Dot(a, b);
    float cosBp = (k2 + (DifferenceOfProducts(k2, cosPhi, k1, sinPhi)) * cosAlpha) /
                  ((SumOfProducts(k2, sinPhi, k1, cosPhi)) * sinAlpha);
    cosBp = Clamp(cosBp, -1.f, 1.f);
    float sinBp = SafeSqrt(1 - Sqr(cosBp));
    V3 cp = cosBp * a + sinBp * Normalize(GramSchmidt(c, a));
    float cosTheta = 1 - u.y * (1 - Dot(cp, b));
    float sinTheta = SafeSqrt(1 - Sqr(cosTheta));
    V3 w = cosTheta * b + sinTheta * Normalize(GramSchmidt(cp, b));
    V3 e1 = v1 - v0, e2 = v2 - v0;
    V3 s1 = Cross(w, e2);
    float divisor = Dot(s1, e1);
    if (divisor == 0) { b3[0] = b3[1] = b3[2] = 1.f / 3.f; return; }
    float invDivisor = 1 / divisor;
    V3 s = p - v0;
    float b1 = Dot(s, s1) * invDivisor;
    float b2 = Dot(w, Cross(s, e1)) * invDivisor;
    b1 = Clamp(b1, 0.f, 1.f);
    b2 = Clamp(b2, 0.f, 1.f);
    if (b1 + b2 > 1) { b1 /= b1 + b2; b2 /= b1 + b2; }
    b3[0] = float(1 - b1 - b2); b3[1] = b1; b3[2] = b2;
}
// util/sampling.cpp:110-160
WF_HD V2 InvertSphericalTriangleSample(V3 v0, V3 v1, V3 v2, V3 p, V3 w) {
    V3 a = v0 - p, b = v1 - p, c = v2 - p;
    a = Normalize(a); b = Normalize(b); c = Normalize(c);
    V3 n_ab = Cross(a, b), n_bc = Cross(b, c), n_ca = Cross(c, a);
    if (LengthSquared(n_ab) == 0 || LengthSquared(n_bc) == 0 || LengthSquared(n_ca) == 0) return {0, 0};
    n_ab = Normalize(n_ab); n_bc = Normalize(n_bc); n_ca = Normalize(n_ca);
    float alpha = AngleBetween(n_ab, -n_ca);
    float beta = AngleBetween(n_bc, -n_ab);
    float gam = AngleBetween(n_ca, -n_bc);
    V3 cp = Normalize(Cross(Cross(b, w), Cross(c, a)));
    if (Dot(cp, a + c) < 0) cp = -cp;
    float u0;
    if (Dot(a, cp) > 0.99999847691f) u0 = 0;
    else {
        V3 n_cpb = Cross(cp, b), n_acp = Cross(a, cp);
        if (LengthSquared(n_cpb) == 0 || LengthSquared(n_acp) == 0) return {0.5f, 0.5f};
        n_cpb = Normalize(n_cpb); n_acp = Normalize(n_acp);
        float Ap = alpha + AngleBetween(n_ab, n_cpb) + AngleBetween(n_acp, -n_cpb) - Pi;
        float A = alpha + beta + gam - Pi;
        u0 = Ap / A;
    }
    float u1 = (1 - Dot(w, b)) / (1 - Dot(cp, b));
    return {Clamp(u0, 0.f, 1.f), Clamp(u1, 0.f, 1.f)};
}

}  // namespace wf
