// device_math.h -- fp64 device functions of the camera / feature measurement model (gfx950).
//
// Values follow the reference's per-feature helpers (citations relative to
// /root/reference/kalmanFilter/modules/, EKF/ = 1PointRansacEKF/), including the two Jacobian quirks that
// change numbers (EKF/MeasurementPrediction.cpp:371-373,392-394 and :579).  Everything here is fp64 and
// register-resident: one thread evaluates one feature.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ekf_types.h"

namespace ekf {

struct CamD {
    double fx, fy, k1, k2, cx, cy, dx, dy, pixelErrorX, pixelErrorY, avx, avy;
    int W, H;
};

struct ParD {
    double linearAccelSD, angularAccelSD, coef2nd, ransacThr, ransacProb, chi2;
};

// R(q): Core/EKFMath.cpp:133-155
__device__ __forceinline__ void quat_to_rot(const double *q, double *M)
{
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    const double r2 = r * r, x2 = x * x, y2 = y * y, z2 = z * z;
    M[0] = r2 + x2 - y2 - z2;
    M[1] = 2 * (x * y - r * z);
    M[2] = 2 * (z * x + r * y);
    M[3] = 2 * (x * y + r * z);
    M[4] = r2 - x2 + y2 - z2;
    M[5] = 2 * (y * z - r * x);
    M[6] = 2 * (z * x - r * y);
    M[7] = 2 * (y * z + r * x);
    M[8] = r2 - x2 - y2 + z2;
}

// 3x3 inverse by the adjugate, the closed form cv::invert uses for 3x3 (R.inv(), MeasurementPrediction.cpp:216,672)
__device__ __forceinline__ void inv3(const double *S, double *D)
{
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
               S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.0) {
        for (int i = 0; i < 9; ++i) D[i] = 0.0;
        return;
    }
    d = 1.0 / d;
    D[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    D[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    D[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    D[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    D[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    D[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    D[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    D[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    D[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}

// 2x2 inverse, closed form of cv::invert for 2x2 (MeasurementPrediction.cpp:352-353, Update.cpp:108 with m=2,
// EKF.cpp:94)
__device__ __forceinline__ bool inv2(const double *S, double *D)
{
    double d = S[0] * S[3] - S[1] * S[2];
    if (d == 0.0) {
        D[0] = D[1] = D[2] = D[3] = 0.0;
        return false;
    }
    d = 1.0 / d;
    D[3] = S[0] * d;
    D[0] = S[3] * d;
    D[1] = -S[1] * d;
    D[2] = -S[2] * d;
    return true;
}

__device__ __forceinline__ void mat3_vec(const double *M, const double *v, double *o)
{
    const double x = v[0], y = v[1], z = v[2];
    o[0] = M[0] * x + M[1] * y + M[2] * z;
    o[1] = M[3] * x + M[4] * y + M[5] * z;
    o[2] = M[6] * x + M[7] * y + M[8] * z;
}

// m(theta, phi): Core/EKFMath.cpp:159-166
__device__ __forceinline__ void dir_vec(double theta, double phi, double *m)
{
    const double cp = cos(phi);
    m[0] = cp * sin(theta);
    m[1] = -sin(phi);
    m[2] = cp * cos(theta);
}

// Point of a feature in "camera axes" using rotation M (R' or inv(R)):
// MeasurementPrediction.cpp:127-140 (inverse depth), :147-156 (depth)
__device__ __forceinline__ void to_camera(const double *fp, int type, const double *r, const double *M, double *h)
{
    double t[3];
    if (type == EKF_FEATURE_INVERSE_DEPTH) {
        double m[3];
        const double rho = fp[5];
        dir_vec(fp[3], fp[4], m);
        t[0] = rho * (fp[0] - r[0]) + m[0];
        t[1] = rho * (fp[1] - r[1]) + m[1];
        t[2] = rho * (fp[2] - r[2]) + m[2];
    } else {
        t[0] = fp[0] - r[0];
        t[1] = fp[1] - r[1];
        t[2] = fp[2] - r[2];
    }
    mat3_vec(M, t, h);
}

// Radial distortion by 10 Newton iterations: MeasurementPrediction.cpp:47-83
__device__ __forceinline__ void distort(const CamD &c, double u, double v, double *out)
{
    const double pdx = u - c.cx, pdy = v - c.cy;
    const double mx = c.dx * pdx, my = c.dy * pdy;
    const double d2 = mx * mx + my * my;
    const double ru = sqrt(d2);
    double rd = ru / (1.0 + c.k1 * d2 + c.k2 * d2 * d2);
#pragma unroll 1
    for (int k = 0; k < 10; ++k) {
        const double r2 = rd * rd, r3 = r2 * rd, r4 = r2 * r2, r5 = r4 * rd;
        const double f = rd + c.k1 * r3 + c.k2 * r5 - ru;
        const double fp = 1 + 3 * c.k1 * r2 + 5 * c.k2 * r4;
        rd = rd - f / fp;
    }
    const double rd2 = rd * rd, rd4 = rd2 * rd2;
    const double d = 1.0 + c.k1 * rd2 + c.k2 * rd4;
    out[0] = c.cx + pdx / d;
    out[1] = c.cy + pdy / d;
}

// One feature of predictMeasurementState (MeasurementPrediction.cpp:203-265): returns true and the distorted
// pixel when the feature is inside the field of view (:162-171) and strictly inside the frame (:176-181).
// Rt = R', Rinv = inv(R): inverse-depth features use R', depth features use inv(R) (:226,:230).
__device__ __forceinline__ bool predict_pixel(const CamD &c, const double *r, const double *Rt, const double *Rinv,
                                              const double *fp, int type, double *uv)
{
    double h[3];
    to_camera(fp, type, r, type == EKF_FEATURE_INVERSE_DEPTH ? Rt : Rinv, h);
    const double ax = atan2(h[0], h[2]) * 180.0 / EKF_PI;
    const double ay = atan2(h[1], h[2]) * 180.0 / EKF_PI;
    if (!(-c.avx < ax && ax < c.avx && -c.avy < ay && ay < c.avy)) return false;
    const double u = c.cx + (c.fx * h[0] / h[2]);
    const double v = c.cy + (c.fy * h[1] / h[2]);
    distort(c, u, v, uv);
    return uv[0] > 0 && uv[0] < c.W && uv[1] > 0 && uv[1] < c.H;
}

// d(R(q) a)/dq as a 3x4 matrix: EKF/CommonFunctions.cpp:87-145
__device__ __forceinline__ void jac_rot_by_quat(const double *q, const double *a, double *J)
{
    const double q0 = 2 * q[0], qx = 2 * q[1], qy = 2 * q[2], qz = 2 * q[3];
    const double x = a[0], y = a[1], z = a[2];
    J[0] = q0 * x - qz * y + qy * z;  J[4] = qz * x + q0 * y - qx * z;  J[8] = -qy * x + qx * y + q0 * z;
    J[1] = qx * x + qy * y + qz * z;  J[5] = qy * x - qx * y - q0 * z;  J[9] = qz * x + q0 * y - qx * z;
    J[2] = -qy * x + qx * y + q0 * z; J[6] = qx * x + qy * y + qz * z;  J[10] = -q0 * x + qz * y - qy * z;
    J[3] = -qz * x - q0 * y + qx * z; J[7] = q0 * x - qz * y + qy * z;  J[11] = qx * x + qy * y + qz * z;
}

// Measurement Jacobian blocks of one feature: Hs = d h/d(r,q) (2x7; the reference's 2x13 block has columns
// 7..12 structurally zero, MeasurementPrediction.cpp:498-499,603) and Hf = d h/d(feature) (2x6, first d used).
// x13 = camera state, Rinv = inv(R(q)), uvd = the predicted distorted pixel.  MeasurementPrediction.cpp:273-589.
__device__ __forceinline__ void measurement_jacobians(const CamD &c, const double *x13, const double *Rinv,
                                                      const double *fp, int type, const double *uvd, double *Hs,
                                                      double *Hf)
{
    const bool invd = (type == EKF_FEATURE_INVERSE_DEPTH);
    // projection o distortion Jacobian, 2x3 (:343-362)
    double pj[6];
    {
        const double pdx = uvd[0] - c.cx, pdy = uvd[1] - c.cy;
        const double mx = c.dx * pdx, my = c.dy * pdy;
        const double d2 = mx * mx + my * my;
        const double rad = 1 + c.k1 * d2 + c.k2 * d2 * d2;
        const double kk = c.k1 + 2 * c.k2 * d2;
        double dj[4], idj[4];
        dj[0] = rad + pdx * kk * (2 * pdx * c.dx * c.dx);
        dj[3] = rad + pdy * kk * (2 * pdy * c.dy * c.dy);
        dj[1] = pdx * kk * (2 * pdy * c.dy * c.dy);
        dj[2] = pdy * kk * (2 * pdx * c.dx * c.dx);
        inv2(dj, idj);
        double h[3];
        to_camera(fp, type, x13, Rinv, h); // note: inv(R) for both feature kinds here (:284,:288)
        const double f0 = c.fx / h[2], f2 = -h[0] * c.fx / (h[2] * h[2]);
        const double f4 = c.fy / h[2], f5 = -h[1] * c.fy / (h[2] * h[2]);
        pj[0] = idj[0] * f0 + idj[1] * 0.0;
        pj[1] = idj[0] * 0.0 + idj[1] * f4;
        pj[2] = idj[0] * f2 + idj[1] * f5;
        pj[3] = idj[2] * f0 + idj[3] * 0.0;
        pj[4] = idj[2] * 0.0 + idj[3] * f4;
        pj[5] = idj[2] * f2 + idj[3] * f5;
    }
    const double rho = invd ? fp[5] : 1.0;
    // d h/d r = pj * carp, carp = -rho inv(R) except element [0][1] = 0 and [0][2] = -rho^2 inv(R)[0][2]
    // (element 1 never written, element 2 written/scaled twice: :371-373, :392-394)
    {
        double carp[9];
        carp[0] = -Rinv[0] * rho;
        carp[1] = 0.0;
        carp[2] = invd ? (-Rinv[2] * rho) * rho : -Rinv[2];
        for (int i = 3; i < 9; ++i) carp[i] = -Rinv[i] * rho;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j)
                Hs[i * 7 + j] = pj[i * 3] * carp[j] + pj[i * 3 + 1] * carp[3 + j] + pj[i * 3 + 2] * carp[6 + j];
    }
    // d h/d q (:442-484)
    double ymr[3] = {fp[0] - x13[0], fp[1] - x13[1], fp[2] - x13[2]};
    {
        double a[3] = {ymr[0], ymr[1], ymr[2]};
        if (invd) {
            double m[3];
            dir_vec(fp[3], fp[4], m);
            a[0] = a[0] * rho + m[0];
            a[1] = a[1] * rho + m[1];
            a[2] = a[2] * rho + m[2];
        }
        const double qc[4] = {x13[3], -x13[4], -x13[5], -x13[6]};
        double J[12];
        jac_rot_by_quat(qc, a, J);
        for (int r3 = 0; r3 < 3; ++r3) {
            J[r3 * 4 + 1] = -J[r3 * 4 + 1];
            J[r3 * 4 + 2] = -J[r3 * 4 + 2];
            J[r3 * 4 + 3] = -J[r3 * 4 + 3];
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 4; ++j)
                Hs[i * 7 + 3 + j] = pj[i * 3] * J[j] + pj[i * 3 + 1] * J[4 + j] + pj[i * 3 + 2] * J[8 + j];
    }
    // d h/d feature
    if (invd) {
        const double theta = fp[3], phi = fp[4];
        const double cp = cos(phi), ct = cos(theta), st = sin(theta), sp = sin(phi);
        const double dth[3] = {cp * ct, 0.0, -cp * st};
        const double dph[3] = {-sp * st, -cp, -sp * ct};
        double rth[3], rph[3];
        mat3_vec(Rinv, dth, rth);
        mat3_vec(Rinv, dph, rph);
        double D[18];
        for (int i = 0; i < 3; ++i) {
            D[i * 6 + 0] = rho * Rinv[3 * i + 0];
            D[i * 6 + 1] = rho * Rinv[3 * i + 1];
            D[i * 6 + 2] = rho * Rinv[3 * i + 2];
            D[i * 6 + 3] = rth[i];
            D[i * 6 + 4] = rph[i];
            D[i * 6 + 5] = ymr[i]; // un-rotated (y - r): :559 computes the rotated one and drops it, :579 uses this
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 6; ++j)
                Hf[i * 6 + j] = pj[i * 3] * D[j] + pj[i * 3 + 1] * D[6 + j] + pj[i * 3 + 2] * D[12 + j];
    } else {
        for (int i = 0; i < 2; ++i) {
            for (int j = 0; j < 3; ++j)
                Hf[i * 6 + j] = pj[i * 3] * Rinv[j] + pj[i * 3 + 1] * Rinv[3 + j] + pj[i * 3 + 2] * Rinv[6 + j];
            Hf[i * 6 + 3] = Hf[i * 6 + 4] = Hf[i * 6 + 5] = 0.0;
        }
    }
}

__device__ __forceinline__ int feat_dim(int type) { return type == EKF_FEATURE_INVERSE_DEPTH ? 6 : 3; }

// wave64 sum via DPP/shuffles
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

} // namespace ekf
