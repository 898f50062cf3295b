// hanimated.h — host side of camera motion blur: the AnimatedTransform constructor (util/transform.cpp:375-395 — the part
// Interpolate needs; the motion-derivative coefficients serve only MotionBounds of animated shapes, which this build does not have)
// with Transform::Decompose (util/transform.cpp:191-227) and Transform::operator Quaternion (:154-189), operation for operation.
// Interpolation itself is the common code the kernels run (csrc/common/wf_camera.h: AnimatedInterpolateP).
#pragma once

#include "hmath.h"
#include <stdexcept>
#include "../common/wf_camera.h"

namespace wf {

inline wf_animated_transform MakeAnimatedTransform(const Transform &startTransform, float startTime, const Transform &endTransform, float endTime) {
    wf_animated_transform A{};
    A.start = startTransform.abi();
    A.end = endTransform.abi();
    A.start_time = startTime;
    A.end_time = endTime;
    A.actually_animated = startTransform != endTransform;
    if (!A.actually_animated) return A;
    auto decompose = [](const Transform &t, float T[3], Quat *Rq, float S[4][4]) {
        const Mat4 &m = t.m;
        T[0] = m.m[0][3]; T[1] = m.m[1][3]; T[2] = m.m[2][3];
        Mat4 M = m;
        for (int i = 0; i < 3; ++i) M.m[i][3] = M.m[3][i] = 0.f;
        M.m[3][3] = 1.f;
        // polar decomposition: R <- (R + (R^T)^-1) / 2 until the rows stop moving
        float norm;
        int count = 0;
        Mat4 R = M;
        do {
            Mat4 Rit;
            if (!Inverse(Transpose(R), &Rit)) throw std::runtime_error("Unable to invert matrix (AnimatedTransform decomposition of a singular camera transformation)");
            Mat4 Rnext;
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Rnext.m[i][j] = (R.m[i][j] + Rit.m[i][j]) / 2;
            norm = 0;
            for (int i = 0; i < 3; ++i) {
                float n = std::abs(R.m[i][0] - Rnext.m[i][0]) + std::abs(R.m[i][1] - Rnext.m[i][1]) + std::abs(R.m[i][2] - Rnext.m[i][2]);
                norm = std::max(norm, n);
            }
            R = Rnext;
        } while (++count < 100 && norm > .0001);   // (a double comparison in the reference: float promoted)
        Mat4 Rinv;
        if (!Inverse(R, &Rinv)) throw std::runtime_error("Unable to invert matrix (AnimatedTransform decomposition of a singular camera transformation)");
        const Mat4 Sm = Rinv * M;
        std::memcpy(S, Sm.m, sizeof(Sm.m));
        // Transform::operator Quaternion() on Transform(R): only m is read
        const auto &r = R.m;
        const float trace = r[0][0] + r[1][1] + r[2][2];
        Quat q;
        if (trace > 0.f) {
            float s = std::sqrt(trace + 1.0f);
            q.w = s / 2.0f;
            s = 0.5f / s;
            q.x = (r[2][1] - r[1][2]) * s;
            q.y = (r[0][2] - r[2][0]) * s;
            q.z = (r[1][0] - r[0][1]) * s;
        } else {
            const int nxt[3] = {1, 2, 0};
            float qq[3];
            int i = 0;
            if (r[1][1] > r[0][0]) i = 1;
            if (r[2][2] > r[i][i]) i = 2;
            const int j = nxt[i], k = nxt[j];
            float s = SafeSqrt((r[i][i] - (r[j][j] + r[k][k])) + 1.0f);
            qq[i] = s * 0.5f;
            if (s != 0.f) s = 0.5f / s;
            q.w = (r[k][j] - r[j][k]) * s;
            qq[j] = (r[j][i] + r[i][j]) * s;
            qq[k] = (r[k][i] + r[i][k]) * s;
            q.x = qq[0]; q.y = qq[1]; q.z = qq[2];
        }
        *Rq = q;
    };
    Quat R[2];
    decompose(startTransform, A.T[0], &R[0], A.S[0]);
    decompose(endTransform, A.T[1], &R[1], A.S[1]);
    if (QDot(R[0], R[1]) < 0) R[1] = Quat{-R[1].x, -R[1].y, -R[1].z, -R[1].w};   // the shortest path
    A.has_rotation = QDot(R[0], R[1]) < 0.9995f;
    for (int e = 0; e < 2; ++e) { A.R[e][0] = R[e].x; A.R[e][1] = R[e].y; A.R[e][2] = R[e].z; A.R[e][3] = R[e].w; }
    return A;
}

// AnimatedTransform::Interpolate(time) as a host Transform
inline Transform AnimatedAt(const wf_animated_transform &A, float time) {
    wf_transform t;
    AnimatedInterpolateP(&A, time, &t);
    Transform r;
    std::memcpy(r.m.m, t.m, sizeof(t.m));
    std::memcpy(r.mInv.m, t.mInv, sizeof(t.mInv));
    return r;
}

}  // namespace wf
