// hmath.h — host-only linear algebra for scene construction: 3x3/4x4 matrices with the reference's
// compensated arithmetic and the Transform factories.  The values computed here (camera matrices,
// render-space vertex positions) are inputs to every kernel, so the arithmetic restates
//   util/math.h:556-610,1405-1632   TwoProd/TwoSum/InnerProduct, SquareMatrix ops, Inverse<3>, Inverse<4>
//   util/transform.h / transform.cpp Transform, Translate/Scale/Rotate*/LookAt/Perspective/Orthographic
// operation for operation.
#pragma once

#include "../common/wf_math.h"
#include "../../../include/wf_abi.h"

#include <cstdio>
#include <cstdlib>

namespace wf {

struct CFloat { float v, err; };
inline CFloat TwoProd(float a, float b) { float ab = a * b; return {ab, fma(a, b, -ab)}; }
inline CFloat TwoSum(float a, float b) {
    float s = a + b, delta = s - a;
    return {s, (a - (s - delta)) + (b - delta)};
}
inline CFloat IP(float a, float b) { return TwoProd(a, b); }
template <typename... T>
inline CFloat IP(float a, float b, T... terms) {
    CFloat ab = TwoProd(a, b);
    CFloat tp = IP(terms...);
    CFloat sum = TwoSum(ab.v, tp.v);
    return {sum.v, ab.err + (tp.err + sum.err)};
}
template <typename... T>
inline float InnerProduct(T... terms) { CFloat ip = IP(terms...); return ip.v + ip.err; }

struct Mat3 {
    float m[3][3];
    static Mat3 Identity() { Mat3 r{}; r.m[0][0] = r.m[1][1] = r.m[2][2] = 1; return r; }
};
struct Mat4 {
    float m[4][4];
    static Mat4 Identity() { Mat4 r{}; for (int i = 0; i < 4; ++i) r.m[i][i] = 1; return r; }
    bool IsIdentity() const {
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (m[i][j] != (i == j ? 1.f : 0.f)) return false;
        return true;
    }
    bool operator==(const Mat4 &o) const { return std::memcmp(m, o.m, sizeof(m)) == 0 || eq(o); }
    bool eq(const Mat4 &o) const { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (m[i][j] != o.m[i][j]) return false; return true; }
};
inline Mat4 Transpose(const Mat4 &a) { Mat4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i]; return r; }
// SquareMatrix<4> product as Transform::operator* (util/transform.cpp:141-143) evaluates it: the generic FMA
// accumulation of util/math.h:1497-1508.  (math.h:1475-1484 also specialises the 4x4 product with the compensated
// InnerProduct, but the out-of-line Transform::operator* of the reference build does not use it: the translation
// column of renderFromCamera's inverse shows the FMA rounding — checked with a harness against libpbrt_ref.a.)
inline Mat4 operator*(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0;
            for (int k = 0; k < 4; ++k) acc = fma(a.m[i][k], b.m[k][j], acc);
            r.m[i][j] = acc;
        }
    return r;
}
// SquareMatrix<3> product: like the 4x4 one above, what the reference build evaluates for `m1 * m2` is the generic FMA accumulation
// (util/math.h:1497-1508), not the InnerProduct specialisation — pinned by the whitebalance golden (film_whitebalance), whose sensor
// matrix is a product of three 3x3 matrices
inline Mat3 operator*(const Mat3 &a, const Mat3 &b) {
    Mat3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float acc = 0;
            for (int k = 0; k < 3; ++k) acc = fma(a.m[i][k], b.m[k][j], acc);
            r.m[i][j] = acc;
        }
    return r;
}
inline float Determinant(const Mat3 &a) {
    const auto &m = a.m;
    float minor12 = DifferenceOfProducts(m[1][1], m[2][2], m[1][2], m[2][1]);
    float minor02 = DifferenceOfProducts(m[1][0], m[2][2], m[1][2], m[2][0]);
    float minor01 = DifferenceOfProducts(m[1][0], m[2][1], m[1][1], m[2][0]);
    return fma(m[0][2], minor01, DifferenceOfProducts(m[0][0], minor12, m[0][1], minor02));
}
inline bool Inverse(const Mat3 &a, Mat3 *out) {
    const auto &m = a.m;
    float det = Determinant(a);
    if (det == 0) return false;
    float invDet = 1 / det;
    auto &r = out->m;
    r[0][0] = invDet * DifferenceOfProducts(m[1][1], m[2][2], m[1][2], m[2][1]);
    r[1][0] = invDet * DifferenceOfProducts(m[1][2], m[2][0], m[1][0], m[2][2]);
    r[2][0] = invDet * DifferenceOfProducts(m[1][0], m[2][1], m[1][1], m[2][0]);
    r[0][1] = invDet * DifferenceOfProducts(m[0][2], m[2][1], m[0][1], m[2][2]);
    r[1][1] = invDet * DifferenceOfProducts(m[0][0], m[2][2], m[0][2], m[2][0]);
    r[2][1] = invDet * DifferenceOfProducts(m[0][1], m[2][0], m[0][0], m[2][1]);
    r[0][2] = invDet * DifferenceOfProducts(m[0][1], m[1][2], m[0][2], m[1][1]);
    r[1][2] = invDet * DifferenceOfProducts(m[0][2], m[1][0], m[0][0], m[1][2]);
    r[2][2] = invDet * DifferenceOfProducts(m[0][0], m[1][1], m[0][1], m[1][0]);
    return true;
}
// Mul<T>(m, v): result[i] = sum_j m[i][j]*v[j], accumulated left to right from 0 (util/math.h:1404-1413)
inline void Mul3(const Mat3 &a, const float v[3], float out[3]) {
    for (int i = 0; i < 3; ++i) {
        float r = 0;
        for (int j = 0; j < 3; ++j) r += a.m[i][j] * v[j];
        out[i] = r;
    }
}
inline bool Inverse(const Mat4 &a, Mat4 *out) {
    const auto &m = a.m;
    float s0 = DifferenceOfProducts(m[0][0], m[1][1], m[1][0], m[0][1]);
    float s1 = DifferenceOfProducts(m[0][0], m[1][2], m[1][0], m[0][2]);
    float s2 = DifferenceOfProducts(m[0][0], m[1][3], m[1][0], m[0][3]);
    float s3 = DifferenceOfProducts(m[0][1], m[1][2], m[1][1], m[0][2]);
    float s4 = DifferenceOfProducts(m[0][1], m[1][3], m[1][1], m[0][3]);
    float s5 = DifferenceOfProducts(m[0][2], m[1][3], m[1][2], m[0][3]);
    float c0 = DifferenceOfProducts(m[2][0], m[3][1], m[3][0], m[2][1]);
    float c1 = DifferenceOfProducts(m[2][0], m[3][2], m[3][0], m[2][2]);
    float c2 = DifferenceOfProducts(m[2][0], m[3][3], m[3][0], m[2][3]);
    float c3 = DifferenceOfProducts(m[2][1], m[3][2], m[3][1], m[2][2]);
    float c4 = DifferenceOfProducts(m[2][1], m[3][3], m[3][1], m[2][3]);
    float c5 = DifferenceOfProducts(m[2][2], m[3][3], m[3][2], m[2][3]);
    float determinant = InnerProduct(s0, c5, -s1, c4, s2, c3, s3, c2, s5, c0, -s4, c1);
    if (determinant == 0) return false;
    float s = 1 / determinant;
    float inv[4][4] = {{s * InnerProduct(m[1][1], c5, m[1][3], c3, -m[1][2], c4),
                        s * InnerProduct(-m[0][1], c5, m[0][2], c4, -m[0][3], c3),
                        s * InnerProduct(m[3][1], s5, m[3][3], s3, -m[3][2], s4),
                        s * InnerProduct(-m[2][1], s5, m[2][2], s4, -m[2][3], s3)},
                       {s * InnerProduct(-m[1][0], c5, m[1][2], c2, -m[1][3], c1),
                        s * InnerProduct(m[0][0], c5, m[0][3], c1, -m[0][2], c2),
                        s * InnerProduct(-m[3][0], s5, m[3][2], s2, -m[3][3], s1),
                        s * InnerProduct(m[2][0], s5, m[2][3], s1, -m[2][2], s2)},
                       {s * InnerProduct(m[1][0], c4, m[1][3], c0, -m[1][1], c2),
                        s * InnerProduct(-m[0][0], c4, m[0][1], c2, -m[0][3], c0),
                        s * InnerProduct(m[3][0], s4, m[3][3], s0, -m[3][1], s2),
                        s * InnerProduct(-m[2][0], s4, m[2][1], s2, -m[2][3], s0)},
                       {s * InnerProduct(-m[1][0], c3, m[1][1], c1, -m[1][2], c0),
                        s * InnerProduct(m[0][0], c3, m[0][2], c0, -m[0][1], c1),
                        s * InnerProduct(-m[3][0], s3, m[3][1], s1, -m[3][2], s0),
                        s * InnerProduct(m[2][0], s3, m[2][2], s0, -m[2][1], s1)}};
    std::memcpy(out->m, inv, sizeof(inv));
    return true;
}

struct Transform {
    Mat4 m = Mat4::Identity(), mInv = Mat4::Identity();
    Transform() = default;
    explicit Transform(const Mat4 &mm) : m(mm) {
        if (!Inverse(mm, &mInv)) {
            float nan = __builtin_nanf("");
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) mInv.m[i][j] = nan;
        }
    }
    Transform(const Mat4 &mm, const Mat4 &mi) : m(mm), mInv(mi) {}
    bool IsIdentity() const { return m.IsIdentity(); }
    bool operator==(const Transform &t) const { return m.eq(t.m); }
    bool operator!=(const Transform &t) const { return !m.eq(t.m); }
    Transform operator*(const Transform &t2) const { return Transform(m * t2.m, t2.mInv * mInv); }
    // util/transform.h:303-330
    V3 Point(V3 p) const {
        const auto &a = m.m;
        float xp = a[0][0] * p.x + a[0][1] * p.y + a[0][2] * p.z + a[0][3];
        float yp = a[1][0] * p.x + a[1][1] * p.y + a[1][2] * p.z + a[1][3];
        float zp = a[2][0] * p.x + a[2][1] * p.y + a[2][2] * p.z + a[2][3];
        float wp = a[3][0] * p.x + a[3][1] * p.y + a[3][2] * p.z + a[3][3];
        if (wp == 1) return V3{xp, yp, zp};
        return V3{xp, yp, zp} / wp;
    }
    // Transform::ApplyInverse(Point3) (util/transform.h:386-398): the inverse matrix, and the sum associated in PAIRS — not the
    // left-to-right sum of operator()(Point3) above; the two differ in the last bit
    V3 ApplyInversePoint(V3 p) const {
        const auto &a = mInv.m;
        float xp = (a[0][0] * p.x + a[0][1] * p.y) + (a[0][2] * p.z + a[0][3]);
        float yp = (a[1][0] * p.x + a[1][1] * p.y) + (a[1][2] * p.z + a[1][3]);
        float zp = (a[2][0] * p.x + a[2][1] * p.y) + (a[2][2] * p.z + a[2][3]);
        float wp = (a[3][0] * p.x + a[3][1] * p.y) + (a[3][2] * p.z + a[3][3]);
        if (wp == 1) return V3{xp, yp, zp};
        return V3{xp, yp, zp} / wp;
    }
    V3 Vector(V3 v) const {
        const auto &a = m.m;
        return V3{a[0][0] * v.x + a[0][1] * v.y + a[0][2] * v.z, a[1][0] * v.x + a[1][1] * v.y + a[1][2] * v.z,
                  a[2][0] * v.x + a[2][1] * v.y + a[2][2] * v.z};
    }
    N3 Normal(N3 n) const {
        const auto &a = mInv.m;
        float x = n.x, y = n.y, z = n.z;
        return N3{a[0][0] * x + a[1][0] * y + a[2][0] * z, a[0][1] * x + a[1][1] * y + a[2][1] * z,
                  a[0][2] * x + a[1][2] * y + a[2][2] * z};
    }
    bool SwapsHandedness() const {
        Mat3 s;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s.m[i][j] = m.m[i][j];
        return Determinant(s) < 0;
    }
    wf_transform abi() const {
        wf_transform t;
        std::memcpy(t.m, m.m, sizeof(t.m));
        std::memcpy(t.mInv, mInv.m, sizeof(t.mInv));
        return t;
    }
};
inline Transform Inverse(const Transform &t) { return Transform(t.mInv, t.m); }
inline Transform TransposeT(const Transform &t) { return Transform(Transpose(t.m), Transpose(t.mInv)); }

inline Mat4 M4(float a00, float a01, float a02, float a03, float a10, float a11, float a12, float a13, float a20,
               float a21, float a22, float a23, float a30, float a31, float a32, float a33) {
    Mat4 r = {{{a00, a01, a02, a03}, {a10, a11, a12, a13}, {a20, a21, a22, a23}, {a30, a31, a32, a33}}};
    return r;
}
inline Transform Translate(V3 d) {
    return Transform(M4(1, 0, 0, d.x, 0, 1, 0, d.y, 0, 0, 1, d.z, 0, 0, 0, 1), M4(1, 0, 0, -d.x, 0, 1, 0, -d.y, 0, 0, 1, -d.z, 0, 0, 0, 1));
}
inline Transform Scale(float x, float y, float z) {
    return Transform(M4(x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1), M4(1 / x, 0, 0, 0, 0, 1 / y, 0, 0, 0, 0, 1 / z, 0, 0, 0, 0, 1));
}
// util/transform.h:260-283
inline Transform Rotate(float sinTheta, float cosTheta, V3 axis) {
    V3 a = Normalize(axis);
    Mat4 m = Mat4::Identity();
    m.m[0][0] = a.x * a.x + (1 - a.x * a.x) * cosTheta;
    m.m[0][1] = a.x * a.y * (1 - cosTheta) - a.z * sinTheta;
    m.m[0][2] = a.x * a.z * (1 - cosTheta) + a.y * sinTheta;
    m.m[0][3] = 0;
    m.m[1][0] = a.x * a.y * (1 - cosTheta) + a.z * sinTheta;
    m.m[1][1] = a.y * a.y + (1 - a.y * a.y) * cosTheta;
    m.m[1][2] = a.y * a.z * (1 - cosTheta) - a.x * sinTheta;
    m.m[1][3] = 0;
    m.m[2][0] = a.x * a.z * (1 - cosTheta) - a.y * sinTheta;
    m.m[2][1] = a.y * a.z * (1 - cosTheta) + a.x * sinTheta;
    m.m[2][2] = a.z * a.z + (1 - a.z * a.z) * cosTheta;
    m.m[2][3] = 0;
    return Transform(m, Transpose(m));
}
inline Transform Rotate(float theta, V3 axis) {
    float sinTheta = std::sin(Radians(theta));
    float cosTheta = std::cos(Radians(theta));
    return Rotate(sinTheta, cosTheta, axis);
}
// util/transform.cpp:81-113
inline Transform LookAt(V3 pos, V3 look, V3 up) {
    Mat4 wfc = Mat4::Identity();
    wfc.m[0][3] = pos.x; wfc.m[1][3] = pos.y; wfc.m[2][3] = pos.z; wfc.m[3][3] = 1;
    V3 dir = Normalize(look - pos);
    if (Length(Cross(Normalize(up), dir)) == 0) {
        fprintf(stderr, "LookAt: \"up\" vector and viewing direction are pointing in the same direction.\n");
        exit(1);
    }
    V3 right = Normalize(Cross(Normalize(up), dir));
    V3 newUp = Cross(dir, right);
    wfc.m[0][0] = right.x; wfc.m[1][0] = right.y; wfc.m[2][0] = right.z; wfc.m[3][0] = 0.;
    wfc.m[0][1] = newUp.x; wfc.m[1][1] = newUp.y; wfc.m[2][1] = newUp.z; wfc.m[3][1] = 0.;
    wfc.m[0][2] = dir.x; wfc.m[1][2] = dir.y; wfc.m[2][2] = dir.z; wfc.m[3][2] = 0.;
    Mat4 cfw;
    if (!Inverse(wfc, &cfw)) { fprintf(stderr, "LookAt: singular matrix\n"); exit(1); }
    return Transform(cfw, wfc);
}
inline Transform Orthographic(float zNear, float zFar) {
    return Scale(1, 1, 1 / (zFar - zNear)) * Translate(V3{0, 0, -zNear});
}
inline Transform Perspective(float fov, float n, float f) {
    Mat4 persp = M4(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, f / (f - n), -f * n / (f - n), 0, 0, 1, 0);
    float invTanAng = 1 / std::tan(Radians(fov) / 2);
    return Scale(invTanAng, invTanAng, 1) * Transform(persp);
}

}  // namespace wf
