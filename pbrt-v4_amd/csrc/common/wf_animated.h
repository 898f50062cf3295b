// wf_animated.h — AnimatedTransform::Interpolate (util/transform.cpp:1062-1081) over the decomposition the host makes at load
// (csrc/host/hanimated.h): what a moving camera (wf_camera.h) and an AnimatedPrimitive (wf_shapes.h, round 5) evaluate per ray.
// A FRAGMENT: included by wf_shapes.h inside namespace wf, right after M44 / Inverse44 (what it needs); nothing else includes it.
#pragma once

// Camera motion blur: CameraTransform::renderFromCamera is an AnimatedTransform (cameras.h:27-110).
// AnimatedTransform::Interpolate (util/transform.cpp:1062-1081) over the decomposition made at load: the translation and the scale
// matrix interpolated linearly, the rotation by Slerp (util/vecmath.h:1138-1151), recomposed as Translate(trans) * Transform(rotate) *
// Transform(scale) — Transform::operator* multiplies m and mInv separately with the generic FMA-accumulated product
// (util/transform.cpp:141-143, util/math.h:1497-1508), Transform(SquareMatrix<4>) inverts numerically (util/transform.h:44-57).
struct Quat { float x, y, z, w; };
WF_HD float QDot(Quat a, Quat b) { return (a.x * b.x + a.y * b.y + a.z * b.z) + a.w * b.w; }   // Dot(q1.v, q2.v) + q1.w * q2.w
WF_HD float SinXOverX(float x) {   // util/math.h:340-344
    if (1 - x * x == 1) return 1;
    return sin(x) / x;
}
WF_HD Quat Slerp(float t, Quat q1, Quat q2) {
    // AngleBetween(Quaternion, Quaternion), util/vecmath.h:1138-1143
    float theta;
    if (QDot(q1, q2) < 0) {
        const Quat s{q1.x + q2.x, q1.y + q2.y, q1.z + q2.z, q1.w + q2.w};
        theta = Pi - 2 * SafeASin(sqrt(QDot(s, s)) / 2);
    } else {
        const Quat d{q2.x - q1.x, q2.y - q1.y, q2.z - q1.z, q2.w - q1.w};
        theta = 2 * SafeASin(sqrt(QDot(d, d)) / 2);
    }
    const float sinThetaOverTheta = SinXOverX(theta);
    // q1 * (1 - t) * SinXOverX((1 - t) * theta) / sinThetaOverTheta + q2 * t * SinXOverX(t * theta) / sinThetaOverTheta, left to right
    const float a = 1 - t, sa = SinXOverX((1 - t) * theta), sb = SinXOverX(t * theta);
    const Quat u{q1.x * a * sa / sinThetaOverTheta, q1.y * a * sa / sinThetaOverTheta, q1.z * a * sa / sinThetaOverTheta, q1.w * a * sa / sinThetaOverTheta};
    const Quat v{q2.x * t * sb / sinThetaOverTheta, q2.y * t * sb / sinThetaOverTheta, q2.z * t * sb / sinThetaOverTheta, q2.w * t * sb / sinThetaOverTheta};
    return Quat{u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w};
}
WF_HD void MulFMA44(const float a[4][4], const float b[4][4], float r[4][4]) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0;
            for (int k = 0; k < 4; ++k) acc = fma(a[i][k], b[k][j], acc);
            r[i][j] = acc;
        }
}
// out of line (pointer arguments): reached only by scenes whose camera moves
WF_NI void AnimatedInterpolateP(const wf_animated_transform *A, float time, wf_transform *out) {
    if (!A->actually_animated || time <= A->start_time) { *out = A->start; return; }
    if (time >= A->end_time) { *out = A->end; return; }
    const float dt = (time - A->start_time) / (A->end_time - A->start_time);
    const float trans[3] = {(1 - dt) * A->T[0][0] + dt * A->T[1][0], (1 - dt) * A->T[0][1] + dt * A->T[1][1], (1 - dt) * A->T[0][2] + dt * A->T[1][2]};
    const Quat q = Slerp(dt, Quat{A->R[0][0], A->R[0][1], A->R[0][2], A->R[0][3]}, Quat{A->R[1][0], A->R[1][1], A->R[1][2], A->R[1][3]});
    M44 scale, scaleInv;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) scale.m[i][j] = A->S[0][i][j] * (1 - dt) + A->S[1][i][j] * dt;
    if (!Inverse44(scale, &scaleInv)) {
        const float nan = BitsToFloat(0x7fc00000u);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) scaleInv.m[i][j] = nan;
    }
    // Transform(Quaternion), util/transform.h:367-384
    const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    const float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
    const float wx = q.x * q.w, wy = q.y * q.w, wz = q.z * q.w;
    float rInv[4][4] = {{1 - 2 * (yy + zz), 2 * (xy + wz), 2 * (xz - wy), 0}, {2 * (xy - wz), 1 - 2 * (xx + zz), 2 * (yz + wx), 0},
                        {2 * (xz + wy), 2 * (yz - wx), 1 - 2 * (xx + yy), 0}, {0, 0, 0, 1}};
    float r[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r[i][j] = rInv[j][i];
    // Translate(trans), util/transform.cpp:21-31
    const float t[4][4] = {{1, 0, 0, trans[0]}, {0, 1, 0, trans[1]}, {0, 0, 1, trans[2]}, {0, 0, 0, 1}};
    const float tInv[4][4] = {{1, 0, 0, -trans[0]}, {0, 1, 0, -trans[1]}, {0, 0, 1, -trans[2]}, {0, 0, 0, 1}};
    float tr[4][4], trInv[4][4];
    MulFMA44(t, r, tr);          // (Translate * Rotate).m
    MulFMA44(rInv, tInv, trInv); // (Translate * Rotate).mInv = Rotate.mInv * Translate.mInv
    MulFMA44(tr, scale.m, out->m);
    MulFMA44(scaleInv.m, trInv, out->mInv);
}
