// gate.h -- the uncertainty-ellipse gate shared by both matchers (descriptor mode A, NCC mode B):
// matrix2x2ToUncertaintyEllipse2D (Core/EKFMath.cpp:271-298) and the foci test of pointIsInsideEllipse
// (Core/EKFMath.cpp:302-351).
#pragma once
#include "engine.h"

namespace ekf {

// cv::eigen of a symmetric 2x2 (Jacobi, eigenvalues descending, eigenvectors as rows) + the ellipse of
// matrix2x2ToUncertaintyEllipse2D: float semi-axes, angle = atan(V[1][0] / V[0][0]).
__device__ inline void ellipse_from_cov(const double *S, float *axes, double *angle)
{
    double A01 = S[1], W0 = S[0], W1 = S[3];
    double V[4] = {1, 0, 0, 1};
    for (int it = 0; it < 120; ++it) {
        const double p = A01;
        if (fabs(p) <= 2.220446049250313e-16) break;
        const double y = (W1 - W0) * 0.5;
        double t = fabs(y) + hypot(p, y);
        double s = hypot(p, t);
        const double c = t / s;
        s = p / s;
        t = (p / t) * p;
        if (y < 0) { s = -s; t = -t; }
        A01 = 0;
        W0 -= t;
        W1 += t;
        for (int i = 0; i < 2; ++i) {
            const double a0 = V[i], b0 = V[2 + i];
            V[i] = a0 * c - b0 * s;
            V[2 + i] = a0 * s + b0 * c;
        }
    }
    if (W0 < W1) {
        double t = W0; W0 = W1; W1 = t;
        for (int i = 0; i < 2; ++i) { t = V[i]; V[i] = V[2 + i]; V[2 + i] = t; }
    }
    axes[0] = (float)(2.0 * sqrt(W0 * EKF_CHISQ_95_2));
    axes[1] = (float)(2.0 * sqrt(W1 * EKF_CHISQ_95_2));
    *angle = atan(V[2] / V[0]);
}

struct Gate {
    double f1x, f1y, f2x, f2y, two_major;
};

// foci of the integer-axes ellipse (pointIsInsideEllipse, Core/EKFMath.cpp:302-334)
__device__ inline void gate_from_ellipse(float cx, float cy, int aw, int ah, double angle, Gate *g)
{
    const double major = aw > ah ? aw : ah;
    const double minor = aw < ah ? aw : ah;
    const double fo = sqrt(major * major - minor * minor);
    if (ah < aw) {
        g->f1x = fo * cos(angle) + cx;  g->f1y = fo * sin(angle) + cy;
        g->f2x = -fo * cos(angle) + cx; g->f2y = -fo * sin(angle) + cy;
    } else {
        g->f1x = fo * (-sin(angle)) + cx;  g->f1y = fo * cos(angle) + cy;
        g->f2x = -fo * (-sin(angle)) + cx; g->f2y = -fo * cos(angle) + cy;
    }
    g->two_major = 2 * major;
}

__device__ inline bool gate_contains(const Gate &g, double px, double py)
{
    const double a1x = px - g.f1x, a1y = py - g.f1y, a2x = px - g.f2x, a2y = py - g.f2y;
    return sqrt(a1x * a1x + a1y * a1y) + sqrt(a2x * a2x + a2y * a2y) <= g.two_major;
}

} // namespace ekf
