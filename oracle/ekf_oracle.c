/*
 * ekf_oracle.c -- TEST INFRASTRUCTURE ONLY (see ekf_oracle.h).  PARITY UNPINNED.
 *
 * Plain C99 restatement of the reference hot path.  Every function cites the reference file:line it follows;
 * paths are relative to /root/reference/kalmanFilter/modules/ with EKF/ = 1PointRansacEKF/.
 * All arithmetic is IEEE double like the reference (Matd = cv::Mat_<double>, Core/Base.h:172).
 *
 * OpenCV 2.4 (pin: OpenCV-2.4.3.2-android-sdk, android/EKFMonoSlam/jni/Android.mk:33) is not available; the
 * operations the path takes from it are restated here from their published algorithms:
 *   - Mat_ operator*  -> cv::gemm, k-sequential accumulation               (gemm_ikj / small fixed loops)
 *   - Mat::inv()      -> closed-form adjugate for 2x2 and 3x3, LU with partial pivoting otherwise
 *   - cv::eigen       -> cyclic Jacobi on the symmetric matrix, eigenvalues sorted descending, eigenvectors
 *                        as rows (jacobi_eigen_2x2)
 *   - cv::Size(Size2f)-> saturate_cast<int>(float) = round-half-to-even
 */
#define _POSIX_C_SOURCE 200809L
#define _DEFAULT_SOURCE
#include "ekf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct OrcFilter {
    EkfCamera cam;
    EkfParams par;
    int cap;       /* max features */
    int N;         /* features in map */
    int n;         /* state dimension */
    double x[13];  /* r q v w */
    double R[9];   /* State::orientationRotationMatrix */
    double *fpos;  /* 6 per feature */
    int32_t *ftype;
    int32_t *fcovpos;
    uint8_t *fdesc;
    int desc_bytes; /* bytes per descriptor row: 32 (CV_8U, Hamming) or 4 * cols (CV_32F, L2) */
    int desc_f32;
    uint32_t *ftimes_predicted;
    uint32_t *ftimes_matched;
    double *P;     /* n x n, leading dimension n */
    size_t Pcap;   /* allocated doubles */
};

/* ------------------------------------------------------------------------------------------------ helpers */

/* quaternionToRotationMatrix: Core/EKFMath.cpp:133-155 */
static void quat_to_rot(const double *q, double *M)
{
    double r = q[0], x = q[1], y = q[2], z = q[3];
    double r2 = r * r, x2 = x * x, y2 = y * y, z2 = z * z;
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

/* State::setOrientation: EKF/State.cpp:131-139 (copies q, recomputes R, never normalises) */
static void set_orientation(double *x13, double *R, const double *q)
{
    for (int i = 0; i < 4; ++i) x13[3 + i] = q[i];
    quat_to_rot(&x13[3], R);
}

/* euclideanNorm3: Core/EKFMath.cpp:36-43 */
static double norm3(const double *v) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* anglesToQuaternion: Core/EKFMath.cpp:58-78 */
static void angles_to_quat(const double *v, double *quat)
{
    double norm = norm3(v);
    if (norm < EKF_EPSILON) {
        quat[0] = 1; quat[1] = 0; quat[2] = 0; quat[3] = 0;
    } else {
        double nd2 = norm / 2;
        double s = sin(nd2);
        quat[0] = cos(nd2);
        quat[1] = s * v[0] / norm;
        quat[2] = s * v[1] / norm;
        quat[3] = s * v[2] / norm;
    }
}

/* multiplyQuaternions: Core/EKFMath.cpp:82-98 */
static void quat_mul(const double *q1, const double *q2, double *q)
{
    double q1w = q1[0], q1x = q1[1], q1y = q1[2], q1z = q1[3];
    double q2w = q2[0], q2x = q2[1], q2y = q2[2], q2z = q2[3];
    q[0] = q1w * q2w - q1x * q2x - q1y * q2y - q1z * q2z;
    q[1] = q1w * q2x + q1x * q2w + q1y * q2z - q1z * q2y;
    q[2] = q1w * q2y - q1x * q2z + q1y * q2w + q1z * q2x;
    q[3] = q1w * q2z + q1x * q2y - q1y * q2x + q1z * q2w;
}

/* makeDirectionalVector: Core/EKFMath.cpp:159-166 */
static void dir_vector(double theta, double fi, double *m)
{
    double cosfi = cos(fi);
    m[0] = cosfi * sin(theta);
    m[1] = -sin(fi);
    m[2] = cosfi * cos(theta);
}

/* multiplyRotationMatrixByVector: Core/EKFMath.cpp:170-179 (reads the vector before writing: alias-safe) */
static void rot_mul_vec(const double *M, const double *v, double *out)
{
    double x = v[0], y = v[1], z = v[2];
    out[0] = M[0] * x + M[1] * y + M[2] * z;
    out[1] = M[3] * x + M[4] * y + M[5] * z;
    out[2] = M[6] * x + M[7] * y + M[8] * z;
}

/* C(MxN) = A(MxK) * B(KxN), row-major with leading dimensions; k-sequential accumulation (cv::gemm class) */
static void gemm_ikj(int M, int K, int N, const double *A, int lda, const double *B, int ldb, double *C, int ldc)
{
    for (int i = 0; i < M; ++i) {
        double *c = C + (size_t)i * ldc;
        for (int j = 0; j < N; ++j) c[j] = 0.0;
        for (int k = 0; k < K; ++k) {
            double a = A[(size_t)i * lda + k];
            const double *b = B + (size_t)k * ldb;
            for (int j = 0; j < N; ++j) c[j] += a * b[j];
        }
    }
}

/* cv::invert, 2x2 closed form (OpenCV 2.4 modules/core/src/lapack.cpp, cv::invert, n == 2 branch) */
static int inv2x2(const double *S, double *D)
{
    double d = S[0] * S[3] - S[1] * S[2];
    if (d == 0.0) { D[0] = D[1] = D[2] = D[3] = 0.0; return 0; }
    d = 1.0 / d;
    double t0 = S[0] * d, t1 = S[3] * d;
    D[3] = t0; D[0] = t1;
    t0 = -S[1] * d; t1 = -S[2] * d;
    D[1] = t0; D[2] = t1;
    return 1;
}

/* cv::invert, 3x3 closed form (same file, n == 3 branch) */
static int inv3x3(const double *S, double *D)
{
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) +
               S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0.0) { memset(D, 0, 9 * sizeof(double)); return 0; }
    d = 1.0 / d;
    double t[9];
    t[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    t[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    t[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    t[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    t[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    t[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    t[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    t[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    t[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    memcpy(D, t, sizeof(t));
    return 1;
}

/* cv::invert DECOMP_LU for n > 3: Gaussian elimination with partial pivoting on [A | I] (cv::LU). A is destroyed. */
static int inv_lu(double *A, int m, double *X)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) X[(size_t)i * m + j] = (i == j) ? 1.0 : 0.0;
    for (int i = 0; i < m; ++i) {
        int k = i;
        for (int j = i + 1; j < m; ++j)
            if (fabs(A[(size_t)j * m + i]) > fabs(A[(size_t)k * m + i])) k = j;
        if (fabs(A[(size_t)k * m + i]) < 2.220446049250313e-16) { /* DBL_EPSILON: singular -> zeros */
            memset(X, 0, (size_t)m * m * sizeof(double));
            return 0;
        }
        if (k != i) {
            for (int j = i; j < m; ++j) { double t = A[(size_t)i * m + j]; A[(size_t)i * m + j] = A[(size_t)k * m + j]; A[(size_t)k * m + j] = t; }
            for (int j = 0; j < m; ++j) { double t = X[(size_t)i * m + j]; X[(size_t)i * m + j] = X[(size_t)k * m + j]; X[(size_t)k * m + j] = t; }
        }
        double d = -1.0 / A[(size_t)i * m + i];
        for (int j = i + 1; j < m; ++j) {
            double alpha = A[(size_t)j * m + i] * d;
            double *aj = A + (size_t)j * m; const double *ai = A + (size_t)i * m;
            for (int c = i + 1; c < m; ++c) aj[c] += alpha * ai[c];
            double *xj = X + (size_t)j * m; const double *xi = X + (size_t)i * m;
            for (int c = 0; c < m; ++c) xj[c] += alpha * xi[c];
        }
        A[(size_t)i * m + i] = -d;
    }
    for (int i = m - 1; i >= 0; --i) {
        double *xi = X + (size_t)i * m;
        for (int k = i + 1; k < m; ++k) {
            double a = A[(size_t)i * m + k];
            const double *xk = X + (size_t)k * m;
            for (int j = 0; j < m; ++j) xi[j] -= a * xk[j];
        }
        double p = A[(size_t)i * m + i];
        for (int j = 0; j < m; ++j) xi[j] *= p;
    }
    return 1;
}

static int feat_dim(int type) { return type == EKF_FEATURE_INVERSE_DEPTH ? 6 : 3; }

/* ------------------------------------------------------------------------------------------- life cycle */

OrcFilter *orc_create(const EkfCamera *cam, const EkfParams *par, int max_features)
{
    OrcFilter *f = (OrcFilter *)calloc(1, sizeof(OrcFilter));
    if (!f) return NULL;
    f->cam = *cam;
    f->par = *par;
    f->cap = max_features;
    f->fpos = (double *)calloc((size_t)max_features * 6 + 6, sizeof(double));
    f->ftype = (int32_t *)calloc((size_t)max_features + 1, sizeof(int32_t));
    f->fcovpos = (int32_t *)calloc((size_t)max_features + 1, sizeof(int32_t));
    f->desc_bytes = EKF_DESC_BYTES;
    f->desc_f32 = 0;
    f->fdesc = (uint8_t *)calloc((size_t)max_features * EKF_DESC_BYTES + EKF_DESC_BYTES, 1);
    f->ftimes_predicted = (uint32_t *)calloc((size_t)max_features + 1, sizeof(uint32_t));
    f->ftimes_matched = (uint32_t *)calloc((size_t)max_features + 1, sizeof(uint32_t));
    size_t nmax = 13 + (size_t)6 * max_features;
    f->Pcap = nmax * nmax;
    f->P = (double *)calloc(f->Pcap, sizeof(double));
    if (!f->fpos || !f->ftype || !f->fcovpos || !f->fdesc || !f->P) { orc_destroy(f); return NULL; }
    orc_reset(f);
    return f;
}

/* descriptor format = the branch of computeDistance (EKF/Matching.cpp:47-92) the matcher takes; flags as
 * EkfEngineConfig.flags (EKF_DESCRIPTOR_*).  Call before any descriptor is stored. */
int orc_set_descriptor_format(OrcFilter *f, int flags)
{
    int bytes = EKF_DESC_BYTES, f32 = 0;
    if ((flags & 0xff) == 1) {
        int cols = (flags >> 8) & 0xffff;
        if (cols < 1 || cols > 1024) return EKF_ERR_INVALID_ARG;
        bytes = 4 * cols;
        f32 = 1;
    } else if ((flags & 0xff) != 0) return EKF_ERR_INVALID_ARG;
    uint8_t *d = (uint8_t *)calloc((size_t)(f->cap + 1) * bytes, 1);
    if (!d) return EKF_ERR_CAPACITY;
    free(f->fdesc);
    f->fdesc = d;
    f->desc_bytes = bytes;
    f->desc_f32 = f32;
    return EKF_OK;
}

void orc_destroy(OrcFilter *f)
{
    if (!f) return;
    free(f->fpos); free(f->ftype); free(f->fcovpos); free(f->fdesc); free(f->P);
    free(f->ftimes_predicted); free(f->ftimes_matched);
    free(f);
}

/* initState EKF/CommonFunctions.cpp:39-55 ; initCovariance :59-80 */
void orc_reset(OrcFilter *f)
{
    f->N = 0;
    f->n = 13;
    memset(f->x, 0, sizeof(f->x));
    double q[4] = {1.0, 0.0, 0.0, 0.0};
    set_orientation(f->x, f->R, q);
    f->x[10] = EKF_EPSILON; f->x[11] = EKF_EPSILON; f->x[12] = EKF_EPSILON;
    memset(f->P, 0, 169 * sizeof(double));
    for (int i = 0; i < 7; ++i) f->P[i * 13 + i] = EKF_EPSILON;
    double l2 = f->par.initLinearAccelSD * f->par.initLinearAccelSD;
    double a2 = f->par.initAngularAccelSD * f->par.initAngularAccelSD;
    for (int i = 0; i < 3; ++i) {
        f->P[(i + 7) * 13 + i + 7] = l2;
        f->P[(i + 10) * 13 + i + 10] = a2;
    }
}

int orc_state_dim(const OrcFilter *f) { return f->n; }
int orc_descriptor_bytes(const OrcFilter *f) { return f->desc_bytes; }
int orc_num_features(const OrcFilter *f) { return f->N; }
double *orc_x13(OrcFilter *f) { return f->x; }
double *orc_rotation(OrcFilter *f) { return f->R; }
double *orc_feature_pos(OrcFilter *f) { return f->fpos; }
int32_t *orc_feature_type(OrcFilter *f) { return f->ftype; }
int32_t *orc_feature_covpos(OrcFilter *f) { return f->fcovpos; }
uint8_t *orc_feature_desc(OrcFilter *f) { return f->fdesc; }
uint32_t *orc_feature_times_predicted(OrcFilter *f) { return f->ftimes_predicted; }
uint32_t *orc_feature_times_matched(OrcFilter *f) { return f->ftimes_matched; }
double *orc_P(OrcFilter *f) { return f->P; }

int orc_set_state(OrcFilter *f, const double x13[13], int n_features, const double *feature_pos,
                  const int32_t *feature_type, const uint8_t *desc32, const double *P)
{
    if (n_features > f->cap) return EKF_ERR_CAPACITY;
    memcpy(f->x, x13, 13 * sizeof(double));
    quat_to_rot(&f->x[3], f->R);
    int pos = 13;
    for (int i = 0; i < n_features; ++i) {
        int t = feature_type ? feature_type[i] : EKF_FEATURE_INVERSE_DEPTH;
        if (t != EKF_FEATURE_DEPTH && t != EKF_FEATURE_INVERSE_DEPTH) return EKF_ERR_INVALID_ARG;
        f->ftype[i] = t;
        f->fcovpos[i] = pos;
        memcpy(&f->fpos[6 * i], &feature_pos[6 * i], 6 * sizeof(double));
        pos += feat_dim(t);
    }
    if (desc32) memcpy(f->fdesc, desc32, (size_t)n_features * f->desc_bytes);
    else memset(f->fdesc, 0, (size_t)n_features * f->desc_bytes);
    memset(f->ftimes_predicted, 0, (size_t)n_features * sizeof(uint32_t));
    memset(f->ftimes_matched, 0, (size_t)n_features * sizeof(uint32_t));
    f->N = n_features;
    f->n = pos;
    if (P) memcpy(f->P, P, (size_t)pos * pos * sizeof(double));
    return EKF_OK;
}

/* ----------------------------------------------------------------------------------------- add a feature */

/* undistortPoint: EKF/AddMapFeature.cpp:42-58 */
static void undistort_point(const EkfCamera *c, const double *p, double *u)
{
    double mx = p[0] - c->cx, my = p[1] - c->cy;
    double dx = c->dx * mx, dy = c->dy * my;
    double rd = dx * dx + dy * dy;
    double dist = 1 + c->k1 * rd + c->k2 * rd * rd;
    u[0] = c->cx + mx * dist;
    u[1] = c->cy + my * dist;
}

/* computeUndistortPointJacobian: EKF/AddMapFeature.cpp:65-90 */
static void undistort_jacobian(const EkfCamera *c, const double *px, double *J)
{
    double ud = px[0], vd = px[1];
    double xd = (ud - c->cx) * c->dx, yd = (vd - c->cy) * c->dy;
    double rd2 = xd * xd + yd * yd;
    double a = c->k1 + 2.0 * c->k2 * rd2;
    double b = 1.0 + c->k1 * rd2 + c->k2 * rd2 * rd2;
    double dx2 = 2.0 * c->dx * c->dx, dy2 = 2.0 * c->dy * c->dy;
    J[0] = b + (ud - c->cx) * a * ((ud - c->cx) * dx2);
    J[1] = (ud - c->cx) * a * ((vd - c->cy) * dy2);
    J[2] = (vd - c->cy) * a * ((ud - c->cx) * dx2);
    J[3] = (vd - c->cy) * a * ((vd - c->cy) * dy2) + b;
}

/* matrixMult: Core/EKFMath.cpp:217-232 */
static void matrix_mult(const double *L, int lr, int lc, const double *Rm, int rc, double *out)
{
    for (int i = 0; i < lr; ++i)
        for (int j = 0; j < rc; ++j) {
            out[i * rc + j] = 0.0;
            for (int k = 0; k < lc; ++k) out[i * rc + j] += L[i * lc + k] * Rm[k * rc + j];
        }
}

/* makeJacobianOfQuaternionToRotationMatrix: EKF/CommonFunctions.cpp:87-145.  jac is 3x4 row-major. */
static void jac_quat_to_rot(const double *q, const double *a, double *jac)
{
    double q0 = q[0], qx = q[1], qy = q[2], qz = q[3];
    double T[9], t[3];
    T[0] = 2 * q0; T[1] = -2 * qz; T[2] = 2 * qy;
    T[3] = 2 * qz; T[4] = 2 * q0; T[5] = -2 * qx;
    T[6] = -2 * qy; T[7] = 2 * qx; T[8] = 2 * q0;
    rot_mul_vec(T, a, t); jac[0] = t[0]; jac[4] = t[1]; jac[8] = t[2];
    T[0] = 2 * qx; T[1] = 2 * qy; T[2] = 2 * qz;
    T[3] = 2 * qy; T[4] = -2 * qx; T[5] = -2 * q0;
    T[6] = 2 * qz; T[7] = 2 * q0; T[8] = -2 * qx;
    rot_mul_vec(T, a, t); jac[1] = t[0]; jac[5] = t[1]; jac[9] = t[2];
    T[0] = -2 * qy; T[1] = 2 * qx; T[2] = 2 * q0;
    T[3] = 2 * qx; T[4] = 2 * qy; T[5] = 2 * qz;
    T[6] = -2 * q0; T[7] = 2 * qz; T[8] = -2 * qy;
    rot_mul_vec(T, a, t); jac[2] = t[0]; jac[6] = t[1]; jac[10] = t[2];
    T[0] = -2 * qz; T[1] = -2 * q0; T[2] = 2 * qx;
    T[3] = 2 * q0; T[4] = -2 * qz; T[5] = 2 * qy;
    T[6] = 2 * qx; T[7] = 2 * qy; T[8] = 2 * qz;
    rot_mul_vec(T, a, t); jac[3] = t[0]; jac[7] = t[1]; jac[11] = t[2];
}

/* computeAddFeatureJacobian: EKF/AddMapFeature.cpp:109-216.  Jpo 6x7, Jhr 6x3 (zero-initialised by caller). */
static void add_feature_jacobian(const EkfCamera *c, const double *q, const double *Rwc, const double *px,
                                 double *Jpo, double *Jhr)
{
    double up[2];
    undistort_point(c, px, up);
    double xyz_c[3] = {-(c->cx - up[0]) / c->fx, -(c->cy - up[1]) / c->fy, 1.0};
    double xyz_w[3];
    rot_mul_vec(Rwc, xyz_c, xyz_w);
    double xw = xyz_w[0], yw = xyz_w[1], zw = xyz_w[2];
    double xxzz = xw * xw + zw * zw;
    double dth[3] = {zw / xxzz, 0, -xw / xxzz};
    double sq = sqrt(xw * xw + zw * zw);
    double nsq = xxzz + yw * yw;
    double dph[3] = {xw * yw / (nsq * sq), -sq / nsq, zw * yw / (nsq * sq)};
    double dgw_dq[12];
    jac_quat_to_rot(q, xyz_c, dgw_dq);
    double dth_dq[4], dph_dq[4];
    matrix_mult(dth, 1, 3, dgw_dq, 4, dth_dq);
    matrix_mult(dph, 1, 3, dgw_dq, 4, dph_dq);
    for (int i = 0; i < 3; ++i) Jpo[i * 7 + i] = 1.0;
    for (int i = 0; i < 4; ++i) {
        Jpo[3 * 7 + i + 3] = dth_dq[i];
        Jpo[4 * 7 + i + 3] = dph_dq[i];
    }
    double sub[6];
    matrix_mult(dth, 1, 3, Rwc, 3, &sub[0]);
    matrix_mult(dph, 1, 3, Rwc, 3, &sub[3]);
    double dgc_dhu[6] = {1.0 / c->fx, 0, 0, 1.0 / c->fy, 0, 0};
    double s2[4];
    matrix_mult(sub, 2, 3, dgc_dhu, 2, s2);
    double dhu_dhd[4];
    undistort_jacobian(c, px, dhu_dhd);
    double s3[4];
    matrix_mult(s2, 2, 2, dhu_dhd, 2, s3);
    Jhr[9] = s3[0]; Jhr[10] = s3[1]; Jhr[12] = s3[2]; Jhr[13] = s3[3]; Jhr[17] = 1.0;
}

/* addFeatureToStateAndCovariance: EKF/AddMapFeature.cpp:293-344 ; addFeatureToCovarianceMatrix :221-289 */
int orc_add_feature(OrcFilter *f, const double uv[2], const uint8_t *desc32)
{
    if (f->N >= f->cap) return -EKF_ERR_CAPACITY;
    const EkfCamera *c = &f->cam;
    double fp[6] = {f->x[0], f->x[1], f->x[2], 0, 0, 0};
    double up[2];
    undistort_point(c, uv, up);
    double rp[3] = {-(c->cx - up[0]) / c->fx, -(c->cy - up[1]) / c->fy, 1.0};
    rot_mul_vec(f->R, rp, rp);
    fp[3] = atan2(rp[0], rp[2]);
    fp[4] = atan2(-rp[1], sqrt(rp[0] * rp[0] + rp[2] * rp[2]));
    fp[5] = f->par.initInvDepthRho;

    int idx = f->N;
    int n0 = f->n, n1 = n0 + 6;
    memcpy(&f->fpos[6 * idx], fp, sizeof(fp));
    f->ftype[idx] = EKF_FEATURE_INVERSE_DEPTH;
    f->fcovpos[idx] = n0;
    if (desc32) memcpy(&f->fdesc[(size_t)idx * f->desc_bytes], desc32, (size_t)f->desc_bytes);
    else memset(&f->fdesc[(size_t)idx * f->desc_bytes], 0, (size_t)f->desc_bytes);
    f->N = idx + 1;

    double Jpo[42] = {0}, Jhr[18] = {0};
    add_feature_jacobian(c, &f->x[3], f->R, uv, Jpo, Jhr);
    double noise[9] = {0};
    noise[0] = c->pixelErrorX * c->pixelErrorX;
    noise[4] = c->pixelErrorY * c->pixelErrorY;
    noise[8] = f->par.inverseDepthRhoSD * f->par.inverseDepthRhoSD;

    /* expand P in place from leading dimension n0 to n1 (walk backwards) */
    double *P = f->P;
    for (int i = n0 - 1; i >= 1; --i) memmove(P + (size_t)i * n1, P + (size_t)i * n0, (size_t)n0 * sizeof(double));
    /* new rows: Jpo (6x7) * P[0:7, 0:n0] */
    for (int a = 0; a < 6; ++a)
        for (int j = 0; j < n0; ++j) {
            double s = 0.0;
            for (int k = 0; k < 7; ++k) s += Jpo[a * 7 + k] * P[(size_t)k * n1 + j];
            P[(size_t)(n0 + a) * n1 + j] = s;
        }
    /* new cols: P[0:n0, 0:7] * Jpo' */
    for (int i = 0; i < n0; ++i)
        for (int a = 0; a < 6; ++a) {
            double s = 0.0;
            for (int k = 0; k < 7; ++k) s += P[(size_t)i * n1 + k] * Jpo[a * 7 + k];
            P[(size_t)i * n1 + n0 + a] = s;
        }
    /* corner: newRows[:,0:7] * Jpo' + Jhr * noise * Jhr' */
    double JN[18], JNJ[36];
    matrix_mult(Jhr, 6, 3, noise, 3, JN);
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += JN[a * 3 + k] * Jhr[b * 3 + k];
            JNJ[a * 6 + b] = s;
        }
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
            double s = 0.0;
            for (int k = 0; k < 7; ++k) s += P[(size_t)(n0 + a) * n1 + k] * Jpo[b * 7 + k];
            P[(size_t)(n0 + a) * n1 + n0 + b] = s + JNJ[a * 6 + b];
        }
    f->n = n1;
    return idx;
}


/* --------------------------------------------------------------------------------------- map management */

/* removeFeaturesFromStateAndCovariance: EKF/MapManagement.cpp:212-259 (+ removeRowsAndColumnsFromMat :168-208).
 * idx: ascending feature indices.  P is compacted in place (row-major, moving up/left only). */
int orc_remove_features(OrcFilter *f, const int32_t *idx, int count)
{
    if (count <= 0) return EKF_OK;
    int n = f->n, N = f->N;
    uint8_t *drop = (uint8_t *)calloc((size_t)n, 1);
    uint8_t *dropf = (uint8_t *)calloc((size_t)N, 1);
    for (int i = 0; i < count; ++i) {
        int fi = idx[i];
        if (fi < 0 || fi >= N || (i > 0 && idx[i] <= idx[i - 1])) { free(drop); free(dropf); return EKF_ERR_INVALID_ARG; }
        dropf[fi] = 1;
        for (int k = 0; k < feat_dim(f->ftype[fi]); ++k) drop[f->fcovpos[fi] + k] = 1;
    }
    int *keep = (int *)malloc((size_t)n * sizeof(int));
    int nn = 0;
    for (int i = 0; i < n; ++i)
        if (!drop[i]) keep[nn++] = i;
    for (int i = 0; i < nn; ++i)
        for (int j = 0; j < nn; ++j) f->P[(size_t)i * nn + j] = f->P[(size_t)keep[i] * n + keep[j]];
    int w = 0, pos = 13;
    for (int i = 0; i < N; ++i) {
        if (dropf[i]) continue;
        if (w != i) {
            memcpy(&f->fpos[6 * w], &f->fpos[6 * i], 6 * sizeof(double));
            f->ftype[w] = f->ftype[i];
            memcpy(&f->fdesc[(size_t)w * f->desc_bytes], &f->fdesc[(size_t)i * f->desc_bytes], (size_t)f->desc_bytes);
            f->ftimes_predicted[w] = f->ftimes_predicted[i];
            f->ftimes_matched[w] = f->ftimes_matched[i];
        }
        f->fcovpos[w] = pos;
        pos += feat_dim(f->ftype[w]);
        ++w;
    }
    f->N = w;
    f->n = nn;
    free(drop); free(dropf); free(keep);
    return EKF_OK;
}

/* removeBadMapFeatures: EKF/MapManagement.cpp:279-308.  Returns how many were removed. */
int orc_remove_bad_features(OrcFilter *f)
{
    int32_t *idx = (int32_t *)malloc((size_t)(f->N + 1) * sizeof(int32_t));
    int c = 0;
    for (int i = 0; i < f->N; ++i) {
        float pct = (float)f->ftimes_matched[i] / (float)f->ftimes_predicted[i]; /* 0/0 = NaN compares false */
        if (pct < f->par.goodFeatureMatchingPercent) idx[c++] = i;
    }
    orc_remove_features(f, idx, c);
    free(idx);
    return c;
}

/* computeLinearityIndex: EKF/MapManagement.cpp:312-341 */
double orc_linearity_index(const OrcFilter *f, int fi)
{
    int d = feat_dim(f->ftype[fi]);
    int ii = f->fcovpos[fi] + d - 1;
    const double *p = &f->fpos[6 * fi];
    double inv_err = sqrt(f->P[(size_t)ii * f->n + ii]);
    double inv_val = p[d - 1];
    double sigma = inv_err / (inv_val * inv_val);
    double m[3], xyz[3];
    dir_vector(p[3], p[4], m); /* changeInverseDepthToDepth, EKF/CommonFunctions.cpp:149-159 */
    xyz[0] = p[0] + m[0] / p[5];
    xyz[1] = p[1] + m[1] / p[5];
    xyz[2] = p[2] + m[2] / p[5];
    double to_cam[3] = {xyz[0] - f->x[0], xyz[1] - f->x[1], xyz[2] - f->x[2]};
    double to_first[3] = {xyz[0] - p[0], xyz[1] - p[1], xyz[2] - p[2]};
    double dot = 0.0;
    for (int k = 0; k < 3; ++k) dot += to_cam[k] * to_first[k];
    double d_first = norm3(to_first), d_cam = norm3(to_cam);
    double cos_div = dot / (d_first * d_cam);
    return 4.0 * sigma * cos_div / d_cam;
}

/* convertToDepth: EKF/MapManagement.cpp:343-490 */
int orc_convert_to_depth(OrcFilter *f, int fi)
{
    if (fi < 0 || fi >= f->N || f->ftype[fi] != EKF_FEATURE_INVERSE_DEPTH) return EKF_ERR_INVALID_ARG;
    int n = f->n, pos = f->fcovpos[fi];
    double *p = &f->fpos[6 * fi];
    double theta = p[3], phi = p[4], rho = p[5], mi[3];
    dir_vector(theta, phi, mi);
    double xyz[3] = {p[0] + mi[0] / rho, p[1] + mi[1] / rho, p[2] + mi[2] / rho};
    double J[18];
    memset(J, 0, sizeof(J));
    J[0 * 6 + 0] = 1.0; J[1 * 6 + 1] = 1.0; J[2 * 6 + 2] = 1.0;
    J[0 * 6 + 3] = cos(phi) * cos(theta) / rho;  J[1 * 6 + 3] = 0.0;             J[2 * 6 + 3] = -cos(phi) * sin(theta) / rho;
    J[0 * 6 + 4] = -sin(phi) * sin(theta) / rho; J[1 * 6 + 4] = -cos(phi) / rho; J[2 * 6 + 4] = -sin(phi) * cos(theta) / rho;
    J[0 * 6 + 5] = -mi[0] / (rho * rho);         J[1 * 6 + 5] = -mi[1] / (rho * rho); J[2 * 6 + 5] = -mi[2] / (rho * rho);
    /* subResult_3xn = J * P[pos:pos+6, :] ; new columns = P[:, pos:pos+6] * J' ; block = subResult[:, pos:pos+6] * J' */
    double *T = (double *)malloc((size_t)3 * n * sizeof(double));
    gemm_ikj(3, 6, n, J, 6, f->P + (size_t)pos * n, n, T, n);
    double *C = (double *)malloc((size_t)n * 3 * sizeof(double));
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            double s2 = 0.0;
            for (int k = 0; k < 6; ++k) s2 += f->P[(size_t)i * n + pos + k] * J[a * 6 + k];
            C[(size_t)i * 3 + a] = s2;
        }
    double B[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double s2 = 0.0;
            for (int k = 0; k < 6; ++k) s2 += T[(size_t)a * n + pos + k] * J[b * 6 + k];
            B[a * 3 + b] = s2;
        }
    int nn = n - 3;
    double *Pn = (double *)malloc((size_t)nn * nn * sizeof(double));
    for (int i = 0; i < nn; ++i) {
        int oi = i < pos ? i : (i < pos + 3 ? -1 - (i - pos) : i + 3); /* <0: new feature row a = -1-oi */
        for (int j = 0; j < nn; ++j) {
            int oj = j < pos ? j : (j < pos + 3 ? -1 - (j - pos) : j + 3);
            double v;
            if (oi >= 0 && oj >= 0) v = f->P[(size_t)oi * n + oj];
            else if (oi < 0 && oj < 0) v = B[(-1 - oi) * 3 + (-1 - oj)];
            else if (oi < 0) v = T[(size_t)(-1 - oi) * n + oj];
            else v = C[(size_t)oi * 3 + (-1 - oj)];
            Pn[(size_t)i * nn + j] = v;
        }
    }
    memcpy(f->P, Pn, (size_t)nn * nn * sizeof(double));
    free(T); free(C); free(Pn);
    p[0] = xyz[0]; p[1] = xyz[1]; p[2] = xyz[2]; p[3] = p[4] = p[5] = 0.0;
    f->ftype[fi] = EKF_FEATURE_DEPTH;
    for (int i = 0; i < f->N; ++i)
        if (f->fcovpos[i] > pos) f->fcovpos[i] -= 3;
    f->n = nn;
    return EKF_OK;
}

/* convertMapFeaturesInverseDepthToDepth: EKF/MapManagement.cpp:494-521 -- at most one conversion per call, the
 * first inverse-depth feature (map order) whose linearity index is below the threshold.  Returns its index or -1. */
int orc_convert_inverse_depth_to_depth(OrcFilter *f)
{
    for (int i = 0; i < f->N; ++i) {
        if (f->ftype[i] != EKF_FEATURE_INVERSE_DEPTH) continue;
        if (orc_linearity_index(f, i) < f->par.inverseDepthLinearityIndexThreshold) {
            orc_convert_to_depth(f, i);
            return i;
        }
    }
    return -1;
}

/* -------------------------------------------------------------------------------------------- prediction */

/* derivQuat*: EKF/StateAndCovariancePrediction.cpp:100-119 */
static double dq_w_by_wa(double wa, double w, double dt) { return (-dt / 2.0) * (wa / w) * sin(w * dt / 2.0); }
static double dq_a_by_wa(double wa, double w, double dt)
{
    return (dt / 2.0) * wa * wa / (w * w) * cos(w * dt / 2.0) +
           (1.0 / w) * (1.0 - wa * wa / (w * w)) * sin(w * dt / 2.0);
}
static double dq_a_by_wb(double wa, double wb, double w, double dt)
{
    return (wa * wb / (w * w)) * ((dt / 2.0) * cos(w * dt / 2.0) - (1.0 / w) * sin(w * dt / 2.0));
}

/* builds F (13x13) and G Q G' (13x13): EKF/StateAndCovariancePrediction.cpp:154-225 */
static void build_F_GQG(const OrcFilter *f, double dt, double *F, double *GQG)
{
    const double *x = f->x;
    memset(F, 0, 169 * sizeof(double));
    for (int i = 0; i < 13; ++i) F[i * 13 + i] = 1.0;
    for (int i = 0; i < 3; ++i) F[i * 13 + i + 7] = dt;
    /* jacobianDynmodelEq3to7Quat :71-92 */
    double w[3] = {x[10] * dt, x[11] * dt, x[12] * dt}, qr[4];
    angles_to_quat(w, qr);
    double qw = qr[0], qx = qr[1], qy = qr[2], qz = qr[3];
    double Fq[16] = {qw, -qx, -qy, -qz, qx, qw, qz, -qy, qy, -qz, qw, qx, qz, qy, -qx, qw};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) F[(3 + i) * 13 + 3 + j] = Fq[i * 4 + j];

    double G[13 * 6];
    memset(G, 0, sizeof(G));
    if (fabs(x[10]) < EKF_EPSILON && fabs(x[11]) < EKF_EPSILON && fabs(x[12]) < EKF_EPSILON) {
        /* :176-184 ; G's quaternion block stays zero: the 4x4 -> 4x3 copyTo at :211-212 detaches */
        for (int i = 0; i < 3; ++i) F[(i + 10) * 13 + i + 10] = 0.0;
    } else {
        /* jacobianDynmodelEq3to7Omega :123-148 */
        double nw = norm3(&x[10]);
        double ox = x[10], oy = x[11], oz = x[12];
        const double *q = &x[3];
        double Qm[16] = {q[0], -q[1], -q[2], -q[3], q[1], q[0], -q[3], q[2],
                         q[2], q[3], q[0], -q[1], q[3], -q[2], q[1], q[0]}; /* Core/EKFMath.cpp:102-113 */
        double D[12] = {dq_w_by_wa(ox, nw, dt),     dq_w_by_wa(oy, nw, dt),     dq_w_by_wa(oz, nw, dt),
                        dq_a_by_wa(ox, nw, dt),     dq_a_by_wb(ox, oy, nw, dt), dq_a_by_wb(ox, oz, nw, dt),
                        dq_a_by_wb(oy, ox, nw, dt), dq_a_by_wa(oy, nw, dt),     dq_a_by_wb(oy, oz, nw, dt),
                        dq_a_by_wb(oz, ox, nw, dt), dq_a_by_wb(oz, oy, nw, dt), dq_a_by_wa(oz, nw, dt)};
        double QD[12];
        gemm_ikj(4, 4, 3, Qm, 4, D, 3, QD, 3);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j) {
                F[(3 + i) * 13 + 10 + j] = QD[i * 3 + j];
                G[(3 + i) * 6 + 3 + j] = QD[i * 3 + j];
            }
    }
    for (int i = 0; i < 3; ++i) {
        G[(i + 7) * 6 + i] = 1.0;
        G[(i + 10) * 6 + i + 3] = 1.0;
        G[i * 6 + i] = 1.0 * dt;
    }
    double Q[36];
    memset(Q, 0, sizeof(Q));
    double ln = f->par.linearAccelSD * f->par.linearAccelSD * dt * dt;
    double an = f->par.angularAccelSD * f->par.angularAccelSD * dt * dt;
    for (int i = 0; i < 3; ++i) { Q[i * 6 + i] = ln; Q[(i + 3) * 6 + i + 3] = an; }
    double GQ[13 * 6], Gt[6 * 13];
    gemm_ikj(13, 6, 6, G, 6, Q, 6, GQ, 6);
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 6; ++j) Gt[j * 13 + i] = G[i * 6 + j];
    gemm_ikj(13, 6, 13, GQ, 6, Gt, 13, GQG, 13);
}

/* predictCovariance :154-240 then predictState :43-65 (that order, :251-252; dt = 1, :246) */
void orc_predict(OrcFilter *f, double *F13, double *GQG13)
{
    const double dt = 1.0;
    int n = f->n;
    double F[169], GQG[169], Ft[169];
    build_F_GQG(f, dt, F, GQG);
    if (F13) memcpy(F13, F, sizeof(F));
    if (GQG13) memcpy(GQG13, GQG, sizeof(GQG));
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 13; ++j) Ft[j * 13 + i] = F[i * 13 + j];
    double *P = f->P;
    /* P[0:13,0:13] = F P13 F' + G Q G'  :226-227 */
    double FP[169], FPF[169];
    gemm_ikj(13, 13, 13, F, 13, P, n, FP, 13);
    gemm_ikj(13, 13, 13, FP, 13, Ft, 13, FPF, 13);
    /* P[0:13,13:] = F P[0:13,13:]  :229-233 */
    if (n > 13) {
        double *tmp = (double *)malloc((size_t)13 * (n - 13) * sizeof(double));
        gemm_ikj(13, 13, n - 13, F, 13, P + 13, n, tmp, n - 13);
        for (int i = 0; i < 13; ++i) memcpy(P + (size_t)i * n + 13, tmp + (size_t)i * (n - 13), (size_t)(n - 13) * sizeof(double));
        free(tmp);
        /* P[13:,0:13] = P[13:,0:13] F'  :235-239 */
        for (int i = 13; i < n; ++i) {
            double row[13], out[13];
            memcpy(row, P + (size_t)i * n, sizeof(row));
            for (int j = 0; j < 13; ++j) out[j] = 0.0;
            for (int k = 0; k < 13; ++k)
                for (int j = 0; j < 13; ++j) out[j] += row[k] * Ft[k * 13 + j];
            memcpy(P + (size_t)i * n, out, sizeof(out));
        }
    }
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 13; ++j) P[(size_t)i * n + j] = FPF[i * 13 + j] + GQG[i * 13 + j];

    /* predictState :43-65 */
    for (int i = 0; i < 3; ++i) f->x[i] += f->x[7 + i] * dt;
    double w[3] = {f->x[10] * dt, f->x[11] * dt, f->x[12] * dt}, q2[4], q[4];
    angles_to_quat(w, q2);
    quat_mul(&f->x[3], q2, q);
    set_orientation(f->x, f->R, q);
}

/* ---------------------------------------------------------------------------------- measurement prediction */

/* distortPoint_matlab: EKF/MeasurementPrediction.cpp:47-83 (in-place safe) */
static void distort_point(const EkfCamera *c, const double *ip, double *out)
{
    double ppx = c->cx, ppy = c->cy;
    double pdx = ip[0] - ppx, pdy = ip[1] - ppy;
    double mx = c->dx * pdx, my = c->dy * pdy;
    double d2 = mx * mx + my * my;
    double ru = sqrt(d2);
    double rd = ru / (1.0 + c->k1 * d2 + c->k2 * d2 * d2);
    for (int k = 0; k < 10; ++k) {
        double r2 = rd * rd, r3 = r2 * rd, r4 = r2 * r2, r5 = r4 * rd;
        double fv = rd + c->k1 * r3 + c->k2 * r5 - ru;
        double fp = 1 + 3 * c->k1 * r2 + 5 * c->k2 * r4;
        rd = rd - fv / fp;
    }
    double rd2 = rd * rd, rd4 = rd2 * rd2;
    double d = (1.0 + c->k1 * rd2 + c->k2 * rd4);
    out[0] = ppx + pdx / d;
    out[1] = ppy + pdy / d;
}

/* changeToCameraReferenceAxisInverseDepth :127-140 */
static void to_camera_axis_invdepth(const double *p, const double *cam, const double *M, double *out)
{
    double rho = p[5], m[3], t[3];
    dir_vector(p[3], p[4], m);
    t[0] = rho * (p[0] - cam[0]) + m[0];
    t[1] = rho * (p[1] - cam[1]) + m[1];
    t[2] = rho * (p[2] - cam[2]) + m[2];
    rot_mul_vec(M, t, out);
}

/* changeToCameraReferenceAxis :147-156 */
static void to_camera_axis(const double *p, const double *cam, const double *M, double *out)
{
    double t[3] = {p[0] - cam[0], p[1] - cam[1], p[2] - cam[2]};
    rot_mul_vec(M, t, out);
}

/* isInFrontOfCamera :162-171 */
static int in_front_of_camera(const EkfCamera *c, const double *h)
{
    double ax = (atan2(h[0], h[2])) * 180.0 / EKF_PI;
    double ay = (atan2(h[1], h[2])) * 180.0 / EKF_PI;
    return -c->angularVisionX < ax && ax < c->angularVisionX && -c->angularVisionY < ay && ay < c->angularVisionY;
}

/* predictMeasurementState :203-265 */
int orc_predict_measurement_state(const OrcFilter *f, const double x13[13], const double R[9],
                                  const double *feature_pos, const int32_t *feat_idx, int count,
                                  EkfPrediction *out)
{
    const EkfCamera *c = &f->cam;
    int total = (feat_idx && count > 0) ? count : f->N;
    if (total == 0) return 0;
    double Rinv[9], Rt[9];
    inv3x3(R, Rinv); /* :216 */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rt[j * 3 + i] = R[i * 3 + j]; /* :217 */
    int np = 0;
    for (int i = 0; i < total; ++i) {
        int fi = (feat_idx && count > 0) ? feat_idx[i] : i;
        const double *p = &feature_pos[6 * fi];
        double h[3], proj[2];
        if (f->ftype[fi] == EKF_FEATURE_INVERSE_DEPTH) to_camera_axis_invdepth(p, x13, Rt, h);
        else to_camera_axis(p, x13, Rinv, h);
        if (in_front_of_camera(c, h)) {
            proj[0] = c->cx + (c->fx * h[0] / h[2]); /* projectToCameraFrame :110-120 */
            proj[1] = c->cy + (c->fy * h[1] / h[2]);
            distort_point(c, proj, proj);
            if (proj[0] > 0 && proj[0] < c->pixelsX && proj[1] > 0 && proj[1] < c->pixelsY) { /* :176-181 */
                out[np].featureIndex = fi;
                out[np]._pad = 0;
                out[np].imagePos[0] = proj[0];
                out[np].imagePos[1] = proj[1];
                out[np].covarianceMatrix[0] = out[np].covarianceMatrix[1] = 0.0;
                out[np].covarianceMatrix[2] = out[np].covarianceMatrix[3] = 0.0;
                ++np;
            }
        }
    }
    return np;
}

/* makeJacobianOfProjection :343-362 (with :273-297 and :308-337 inlined). pj is 2x3. */
static void jac_projection(const OrcFilter *f, const double *Rinv, const double *dpt, const double *wp, int invd,
                           double *pj)
{
    const EkfCamera *c = &f->cam;
    /* makeJacobianOfDistortionFunction :308-337 */
    double pdx = dpt[0] - c->cx, pdy = dpt[1] - c->cy;
    double mx = c->dx * pdx, my = c->dy * pdy;
    double d2 = mx * mx + my * my;
    double rad = 1 + c->k1 * d2 + c->k2 * d2 * d2;
    double dj[4];
    dj[0] = rad + pdx * (c->k1 + 2 * c->k2 * d2) * (2 * pdx * c->dx * c->dx);
    dj[3] = rad + pdy * (c->k1 + 2 * c->k2 * d2) * (2 * pdy * c->dy * c->dy);
    dj[1] = pdx * (c->k1 + 2 * c->k2 * d2) * (2 * pdy * c->dy * c->dy);
    dj[2] = pdy * (c->k1 + 2 * c->k2 * d2) * (2 * pdx * c->dx * c->dx);
    /* makeJacobianOfFrameProjectionFunction :273-297 */
    double h[3];
    if (invd) to_camera_axis_invdepth(wp, f->x, Rinv, h);
    else to_camera_axis(wp, f->x, Rinv, h);
    double fpj[6];
    fpj[0] = c->fx / h[2]; fpj[1] = 0; fpj[2] = -h[0] * c->fx / (h[2] * h[2]);
    fpj[3] = 0; fpj[4] = c->fy / h[2]; fpj[5] = -h[1] * c->fy / (h[2] * h[2]);
    double idj[4];
    inv2x2(dj, idj); /* :352-353 */
    pj[0] = idj[0] * fpj[0] + idj[1] * fpj[3];
    pj[1] = idj[0] * fpj[1] + idj[1] * fpj[4];
    pj[2] = idj[0] * fpj[2] + idj[1] * fpj[5];
    pj[3] = idj[2] * fpj[0] + idj[3] * fpj[3];
    pj[4] = idj[2] * fpj[1] + idj[3] * fpj[4];
    pj[5] = idj[2] * fpj[2] + idj[3] * fpj[5];
}

/* makeJacobianOfMeasurementByState :491-503 -> Hs (2x13, cols 7..12 stay zero) */
static void jac_by_state(const OrcFilter *f, const double *Rinv, const double *dpt, const double *wp, int invd,
                         double *Hs)
{
    memset(Hs, 0, 26 * sizeof(double));
    /* makeJacobianOfProjectionAndChangeToCameraAxis :411-436 */
    double carp[9] = {0};
    /* makeJacobianOfChangeToCameraAxisRightPart :369-380: element [1] is never written, [2] written twice */
    carp[0] = -Rinv[0];
    carp[2] = -Rinv[1];
    carp[2] = -Rinv[2];
    carp[3] = -Rinv[3]; carp[4] = -Rinv[4]; carp[5] = -Rinv[5];
    carp[6] = -Rinv[6]; carp[7] = -Rinv[7]; carp[8] = -Rinv[8];
    if (invd) { /* :388-401: [2] scaled twice, [1] never */
        double rho = wp[5];
        carp[0] *= rho;
        carp[2] *= rho;
        carp[2] *= rho;
        carp[3] *= rho; carp[4] *= rho; carp[5] *= rho;
        carp[6] *= rho; carp[7] *= rho; carp[8] *= rho;
    }
    double pj[6];
    jac_projection(f, Rinv, dpt, wp, invd, pj);
    double Jr[6];
    gemm_ikj(2, 3, 3, pj, 3, carp, 3, Jr, 3);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) Hs[i * 13 + j] = Jr[i * 3 + j];
    /* makeJacobianOfChangeToCameraAxisByQuaternion :442-484 */
    const double *q = &f->x[3];
    double qc[4] = {q[0], -q[1], -q[2], -q[3]};
    double ca[3] = {wp[0] - f->x[0], wp[1] - f->x[1], wp[2] - f->x[2]};
    if (invd) {
        double m[3], rho = wp[5];
        dir_vector(wp[3], wp[4], m);
        ca[0] = ca[0] * rho + m[0];
        ca[1] = ca[1] * rho + m[1];
        ca[2] = ca[2] * rho + m[2];
    }
    double qj[12];
    jac_quat_to_rot(qc, ca, qj);
    for (int r = 0; r < 3; ++r)
        for (int cidx = 1; cidx < 4; ++cidx) qj[r * 4 + cidx] = -qj[r * 4 + cidx]; /* :468-476 */
    double pj2[6];
    jac_projection(f, Rinv, dpt, wp, invd, pj2);
    double Jq[8];
    gemm_ikj(2, 3, 4, pj2, 3, qj, 4, Jq, 4);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) Hs[i * 13 + 3 + j] = Jq[i * 4 + j];
}

/* makeJacobianOfMeasurementByFeatureiInverseDepth :530-589 / ...Depth :510-523 -> Hf (2x6, first d cols) */
static void jac_by_feature(const OrcFilter *f, const double *Rinv, const double *dpt, const double *wp, int invd,
                           double *Hf)
{
    memset(Hf, 0, 12 * sizeof(double));
    double pj[6];
    if (!invd) {
        jac_projection(f, Rinv, dpt, wp, 0, pj);
        double J[6];
        gemm_ikj(2, 3, 3, pj, 3, Rinv, 3, J, 3);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) Hf[i * 6 + j] = J[i * 3 + j];
        return;
    }
    double theta = wp[3], phi = wp[4], rho = wp[5];
    double cphi = cos(phi), cth = cos(theta), sth = sin(theta), sphi = sin(phi);
    double dth[3] = {cphi * cth, 0, -cphi * sth};
    double dph[3] = {-sphi * sth, -cphi, -sphi * cth};
    double pc[3] = {wp[0] - f->x[0], wp[1] - f->x[1], wp[2] - f->x[2]};
    double rth[3], rph[3];
    rot_mul_vec(Rinv, dth, rth);
    rot_mul_vec(Rinv, dph, rph);
    /* :559 computes R*(y - r) and discards it; the UN-rotated (y - r) goes into the rho column (:579) */
    double D[18];
    for (int i = 0; i < 3; ++i) {
        D[i * 6 + 0] = rho * Rinv[3 * i + 0];
        D[i * 6 + 1] = rho * Rinv[3 * i + 1];
        D[i * 6 + 2] = rho * Rinv[3 * i + 2];
        D[i * 6 + 3] = rth[i];
        D[i * 6 + 4] = rph[i];
        D[i * 6 + 5] = pc[i];
    }
    jac_projection(f, Rinv, dpt, wp, 1, pj);
    gemm_ikj(2, 3, 6, pj, 3, D, 6, Hf, 6);
}

/* predictCameraMeasurements :705-719 = predictMeasurementState + predictMeasurementCovariance :666-700 */
int orc_predict_measurements(OrcFilter *f, const int32_t *feat_idx, int count, EkfPrediction *preds, double *Hs,
                             double *Hf, double *HP)
{
    int np = orc_predict_measurement_state(f, f->x, f->R, f->fpos, feat_idx, count, preds);
    int n = f->n;
    double Rinv[9];
    inv3x3(f->R, Rinv); /* :672 */
    double *hp = (double *)malloc((size_t)2 * n * sizeof(double));
    for (int k = 0; k < np; ++k) {
        int fi = preds[k].featureIndex;
        int invd = f->ftype[fi] == EKF_FEATURE_INVERSE_DEPTH;
        int d = feat_dim(f->ftype[fi]);
        int pos = f->fcovpos[fi];
        double *hs = Hs + (size_t)26 * k, *hf = Hf + (size_t)12 * k;
        /* makeMeasurementCovariance :595-658 */
        jac_by_state(f, Rinv, preds[k].imagePos, &f->fpos[6 * fi], invd, hs);
        jac_by_feature(f, Rinv, preds[k].imagePos, &f->fpos[6 * fi], invd, hf);
        /* hiByP = Hf * P[pos:pos+d,:] + Hs * P[0:13,:]  :644 */
        for (int r = 0; r < 2; ++r) {
            double *o = hp + (size_t)r * n;
            for (int j = 0; j < n; ++j) o[j] = 0.0;
            for (int a = 0; a < d; ++a) {
                double h = hf[r * 6 + a];
                const double *prow = f->P + (size_t)(pos + a) * n;
                for (int j = 0; j < n; ++j) o[j] += h * prow[j];
            }
            /* second product accumulated separately then added, as the expression template does */
            double *t = (double *)malloc((size_t)n * sizeof(double));
            for (int j = 0; j < n; ++j) t[j] = 0.0;
            for (int a = 0; a < 13; ++a) {
                double h = hs[r * 13 + a];
                const double *prow = f->P + (size_t)a * n;
                for (int j = 0; j < n; ++j) t[j] += h * prow[j];
            }
            for (int j = 0; j < n; ++j) o[j] += t[j];
            free(t);
        }
        /* S_i = hiByP[:,0:13] Hs' + hiByP[:,pos:pos+d] Hf' + I  :651-653 */
        for (int r = 0; r < 2; ++r)
            for (int c2 = 0; c2 < 2; ++c2) {
                double s1 = 0.0, s2 = 0.0;
                for (int a = 0; a < 13; ++a) s1 += hp[(size_t)r * n + a] * hs[c2 * 13 + a];
                for (int a = 0; a < d; ++a) s2 += hp[(size_t)r * n + pos + a] * hf[c2 * 6 + a];
                preds[k].covarianceMatrix[r * 2 + c2] = s1 + s2 + (r == c2 ? 1.0 : 0.0);
            }
        if (HP) memcpy(HP + (size_t)2 * n * k, hp, (size_t)2 * n * sizeof(double));
    }
    free(hp);
    return np;
}

/* ---------------------------------------------------------------------------------------------- matching */

/* cv::eigen on a symmetric 2x2 (OpenCV 2.4 modules/core/src/lapack.cpp JacobiImpl_): V starts as I, the single
 * pivot (0,1) is rotated away, eigenvalues sorted descending, eigenvectors are the ROWS of V. */
static void jacobi_eigen_2x2(const double *S, double *W, double *V)
{
    double A01 = S[1];
    W[0] = S[0]; W[1] = S[3];
    V[0] = 1; V[1] = 0; V[2] = 0; V[3] = 1;
    for (int iters = 0; iters < 2 * 2 * 30; ++iters) {
        double p = A01;
        if (fabs(p) <= 2.220446049250313e-16) break;
        double y = (W[1] - W[0]) * 0.5;
        double t = fabs(y) + hypot(p, y);
        double s = hypot(p, t);
        double c = t / s;
        s = p / s;
        t = (p / t) * p;
        if (y < 0) { s = -s; t = -t; }
        A01 = 0;
        W[0] -= t;
        W[1] += t;
        for (int i = 0; i < 2; ++i) { /* rotate eigenvector rows 0 and 1 */
            double a0 = V[0 * 2 + i], b0 = V[1 * 2 + i];
            V[0 * 2 + i] = a0 * c - b0 * s;
            V[1 * 2 + i] = a0 * s + b0 * c;
        }
    }
    if (W[0] < W[1]) {
        double t = W[0]; W[0] = W[1]; W[1] = t;
        for (int i = 0; i < 2; ++i) { t = V[i]; V[i] = V[2 + i]; V[2 + i] = t; }
    }
}

/* matrix2x2ToUncertaintyEllipse2D: Core/EKFMath.cpp:271-298 */
void orc_ellipse(const double S[4], float axes[2], double *angle)
{
    double W[2], V[4];
    jacobi_eigen_2x2(S, W, V);
    axes[0] = (float)(2.0 * sqrt(W[0] * EKF_CHISQ_95_2));
    axes[1] = (float)(2.0 * sqrt(W[1] * EKF_CHISQ_95_2));
    double tn = V[2] / V[0]; /* eigenVectors[1][0] / eigenVectors[0][0] */
    *angle = atan(tn);
}

/* pointIsInsideEllipse: Core/EKFMath.cpp:302-351 */
int orc_point_in_ellipse(float px, float py, float cx, float cy, int aw, int ah, double angle)
{
    double major = aw > ah ? aw : ah;
    double minor = aw < ah ? aw : ah;
    double fo = sqrt(major * major - minor * minor);
    double f1x, f1y, f2x, f2y;
    if (ah < aw) {
        f1x = fo * cos(angle) + cx;  f1y = fo * sin(angle) + cy;
        f2x = -fo * cos(angle) + cx; f2y = -fo * sin(angle) + cy;
    } else {
        f1x = fo * (-sin(angle)) + cx;  f1y = fo * cos(angle) + cy;
        f2x = -fo * (-sin(angle)) + cx; f2y = -fo * cos(angle) + cy;
    }
    double a1x = px - f1x, a1y = py - f1y, a2x = px - f2x, a2y = py - f2y;
    double ns = sqrt(a1x * a1x + a1y * a1y) + sqrt(a2x * a2x + a2y * a2y);
    return ns <= 2 * major;
}

static int popcount8(uint8_t v)
{
    int c = 0;
    while (v) { c += v & 1; v >>= 1; }
    return c; /* == popCountTable[v], Core/EKFMath.h:48-58 */
}

/* computeDistance, CV_8U branch: EKF/Matching.cpp:74-90 */
static double hamming32(const uint8_t *a, const uint8_t *b)
{
    unsigned d = 0;
    for (int j = 0; j < EKF_DESC_BYTES; ++j) d += (unsigned)popcount8((uint8_t)(a[j] ^ b[j]));
    return (double)d;
}

/* computeDistance, CV_32F branch: EKF/Matching.cpp:60-73 -- float difference, float square, double sum, sqrt */
static double l2_f32(const uint8_t *a, const uint8_t *b, int cols)
{
    const float *x = (const float *)a, *y = (const float *)b;
    double distance = 0.0f;
    for (int j = 0; j < cols; ++j) {
        float subs = x[j] - y[j];
        distance += subs * subs;
    }
    return sqrt(distance);
}

/* EKF/Matching.cpp:217-262 (gate), :148-177 (matchICDescriptors), :116-144 (findBestNMatches) */
int orc_match(const OrcFilter *f, const EkfPrediction *preds, int n_pred, const EkfKeypoint *kps,
              const uint8_t *desc32, int n_kp, EkfMatch *out)
{
    int nm = 0;
    double coef = f->par.matchingCompCoefSecondBestVSFirst;
    for (int i = 0; i < n_pred; ++i) {
        int fi = preds[i].featureIndex;
        float axes[2];
        double angle;
        orc_ellipse(preds[i].covarianceMatrix, axes, &angle);
        /* cv::Size(cv::Size2f): saturate_cast<int>(float) = round half to even */
        int aw = (int)lrintf(axes[0]), ah = (int)lrintf(axes[1]);
        float cx = (float)preds[i].imagePos[0], cy = (float)preds[i].imagePos[1]; /* Point2d -> Point2f */
        const uint8_t *qd = &f->fdesc[(size_t)fi * f->desc_bytes];
        /* findBestNMatches with nBest = 2: a 2-element list, newest in front */
        int list_n = 0, idx_front = -1, idx_back = -1;
        float d_front = 0.f, d_back = 0.f;
        double min_distance = -1.0;
        for (int j = 0; j < n_kp; ++j) {
            if (!orc_point_in_ellipse(kps[j].x, kps[j].y, cx, cy, aw, ah, angle)) continue;
            const uint8_t *cd = &desc32[(size_t)j * f->desc_bytes];
            double dist = f->desc_f32 ? l2_f32(qd, cd, f->desc_bytes / 4) : hamming32(qd, cd);
            if (dist < min_distance || list_n < 2) {
                min_distance = min_distance < 0 ? dist : (min_distance < dist ? min_distance : dist);
                idx_back = idx_front; d_back = d_front; /* push_front, pop_back beyond 2 */
                idx_front = j; d_front = (float)dist;
                if (list_n < 2) ++list_n;
            }
        }
        (void)idx_back;
        if (list_n == 1 || (list_n >= 2 && (double)d_front <= (double)d_back * coef)) {
            out[nm].featureIndex = fi;
            out[nm].keypointIndex = idx_front;
            out[nm].imagePos[0] = (double)kps[idx_front].x;
            out[nm].imagePos[1] = (double)kps[idx_front].y;
            out[nm].distance = d_front;
            out[nm]._pad = 0.f;
            ++nm;
        }
    }
    return nm;
}


/* ------------------------------------------------------------------------------------ NCC matcher (mode B) */
/* The north star's patch matcher (ZNCC of an 11x11 template inside the predicted ellipse, coarse-to-fine over a
 * 3-level pyramid) has NO counterpart in the reference (its matcher is detector + descriptor + gate,
 * EKF/Matching.cpp:181-264).  This is the build's own definition, restated on the CPU so the HIP kernels can be
 * checked bit-exactly: all image arithmetic is integer, the score is one fp64 multiply and one fp64 divide of
 * exactly representable or identically rounded operands. */

#define NCC_LEVELS 3
#define NCC_R 5              /* template half size: 11 x 11 */
#define NCC_T (2 * NCC_R + 1)
#define NCC_MAXRAD 16        /* coarse search radius, pixels at the coarse level */

typedef struct OrcImage {
    int w[NCC_LEVELS], h[NCC_LEVELS];
    uint8_t *px[NCC_LEVELS];
} OrcImage;

static OrcImage g_img; /* one current frame per process is enough for a test oracle */
static uint8_t *g_tmpl; /* cap x levels x 121 */
static int g_tmpl_cap;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* gray = (77 R + 150 G + 29 B + 128) >> 8 for 3/4-channel input (BGR / RGBA byte order as the reference's two
 * front ends deliver it: Img/FileSequenceImageGenerator.cpp:82, android jni/EKFNative.cpp:163-166), copy for 1 */
int orc_image_set(const uint8_t *image, int w, int h, int stride, int channels)
{
    for (int l = 0; l < NCC_LEVELS; ++l) { free(g_img.px[l]); g_img.px[l] = NULL; }
    g_img.w[0] = w; g_img.h[0] = h;
    g_img.px[0] = (uint8_t *)malloc((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t *p = image + (size_t)y * stride + (size_t)x * channels;
            int g;
            if (channels == 1) g = p[0];
            else if (channels == 3) g = (77 * p[2] + 150 * p[1] + 29 * p[0] + 128) >> 8;      /* B G R */
            else g = (77 * p[0] + 150 * p[1] + 29 * p[2] + 128) >> 8;                          /* R G B A */
            g_img.px[0][(size_t)y * w + x] = (uint8_t)g;
        }
    for (int l = 1; l < NCC_LEVELS; ++l) {
        int pw = g_img.w[l - 1], ww = pw / 2, hh = g_img.h[l - 1] / 2;
        g_img.w[l] = ww; g_img.h[l] = hh;
        g_img.px[l] = (uint8_t *)malloc((size_t)ww * hh + 1);
        const uint8_t *s2 = g_img.px[l - 1];
        for (int y = 0; y < hh; ++y)
            for (int x = 0; x < ww; ++x)
                g_img.px[l][(size_t)y * ww + x] = (uint8_t)((s2[(size_t)(2 * y) * pw + 2 * x] + s2[(size_t)(2 * y) * pw + 2 * x + 1] +
                                                             s2[(size_t)(2 * y + 1) * pw + 2 * x] + s2[(size_t)(2 * y + 1) * pw + 2 * x + 1] + 2) >> 2);
    }
    return EKF_OK;
}

const uint8_t *orc_image_level(int level, int *w, int *h)
{
    *w = g_img.w[level]; *h = g_img.h[level];
    return g_img.px[level];
}

static int img_at(int l, int x, int y)
{
    x = clampi(x, 0, g_img.w[l] - 1);
    y = clampi(y, 0, g_img.h[l] - 1);
    return g_img.px[l][(size_t)y * g_img.w[l] + x];
}

/* level-l pixel that contains level-0 coordinate u: floor((u + 0.5) / 2^l) */
static int to_level(double u, int l) { return (int)floor((u + 0.5) / (double)(1 << l)); }

/* templates of the listed features from the CURRENT image, centred on the pixel containing uv at each level */
int orc_capture_templates(OrcFilter *f, const int32_t *feat_idx, const double *uv, int count)
{
    if (g_tmpl_cap < f->cap) {
        free(g_tmpl);
        g_tmpl = (uint8_t *)calloc((size_t)f->cap * NCC_LEVELS * NCC_T * NCC_T, 1);
        g_tmpl_cap = f->cap;
    }
    for (int i = 0; i < count; ++i)
        for (int l = 0; l < NCC_LEVELS; ++l) {
            int cx = to_level(uv[2 * i], l), cy = to_level(uv[2 * i + 1], l);
            uint8_t *t = g_tmpl + ((size_t)feat_idx[i] * NCC_LEVELS + l) * NCC_T * NCC_T;
            for (int dy = -NCC_R; dy <= NCC_R; ++dy)
                for (int dx = -NCC_R; dx <= NCC_R; ++dx) t[(dy + NCC_R) * NCC_T + dx + NCC_R] = (uint8_t)img_at(l, cx + dx, cy + dy);
        }
    return EKF_OK;
}

const uint8_t *orc_templates(void) { return g_tmpl; }

/* key of a candidate: num^2 / den if the correlation numerator is positive, -1 otherwise (den == 0: flat window) */
static double ncc_key(int l, const uint8_t *t, int cx, int cy)
{
    long long sw = 0, sww = 0, swt = 0, st = 0, stt = 0;
    for (int dy = -NCC_R; dy <= NCC_R; ++dy)
        for (int dx = -NCC_R; dx <= NCC_R; ++dx) {
            int w = img_at(l, cx + dx, cy + dy), tv = t[(dy + NCC_R) * NCC_T + dx + NCC_R];
            sw += w; sww += w * w; swt += w * tv; st += tv; stt += tv * tv;
        }
    const long long n = NCC_T * NCC_T;
    long long num = n * swt - sw * st;
    long long den = (n * sww - sw * sw) * (n * stt - st * st);
    if (num <= 0 || den <= 0) return -1.0;
    double dn = (double)num;
    return dn * dn / (double)den;
}

/* One prediction.  Returns 1 and the level-0 pixel + key when matched. */
static int ncc_match_one(const OrcFilter *f, const EkfPrediction *p, int *mx, int *my, double *mkey)
{
    (void)f;
    float axes[2];
    double angle;
    orc_ellipse(p->covarianceMatrix, axes, &angle);
    int aw = (int)lrintf(axes[0]), ah = (int)lrintf(axes[1]);
    float cxf = (float)p->imagePos[0], cyf = (float)p->imagePos[1];
    const uint8_t *tb = g_tmpl + (size_t)p->featureIndex * NCC_LEVELS * NCC_T * NCC_T;
    /* coarse level: every pixel of the (clamped) window whose centre, mapped to level 0, lies inside the ellipse */
    int L = NCC_LEVELS - 1;
    int c2x = to_level(p->imagePos[0], L), c2y = to_level(p->imagePos[1], L);
    int major = aw > ah ? aw : ah;
    int rad = (major >> L) + 1;
    if (rad > NCC_MAXRAD) rad = NCC_MAXRAD;
    int bx = c2x, by = c2y;
    double bkey = -3.0;
    for (int y = c2y - rad; y <= c2y + rad; ++y)
        for (int x = c2x - rad; x <= c2x + rad; ++x) {
            if (x < 0 || y < 0 || x >= g_img.w[L] || y >= g_img.h[L]) continue;
            float x0 = (float)((x + 0.5) * (1 << L) - 0.5), y0 = (float)((y + 0.5) * (1 << L) - 0.5);
            if (!(x == c2x && y == c2y) && !orc_point_in_ellipse(x0, y0, cxf, cyf, aw, ah, angle)) continue;
            double k = ncc_key(L, tb + (size_t)L * NCC_T * NCC_T, x, y);
            if (k > bkey) { bkey = k; bx = x; by = y; }
        }
    /* refinement: the 4 x 4 block of children around the best parent, no gate */
    for (int l = L - 1; l >= 0; --l) {
        int px = bx, py = by;
        bkey = -3.0;
        for (int y = 2 * py - 1; y <= 2 * py + 2; ++y)
            for (int x = 2 * px - 1; x <= 2 * px + 2; ++x) {
                if (x < 0 || y < 0 || x >= g_img.w[l] || y >= g_img.h[l]) continue;
                double k = ncc_key(l, tb + (size_t)l * NCC_T * NCC_T, x, y);
                if (k > bkey) { bkey = k; bx = x; by = y; }
            }
    }
    *mx = bx; *my = by; *mkey = bkey;
    /* accept: ZNCC >= 0.8 (key = zncc^2 >= 0.64) and the pixel inside the gate ellipse */
    return bkey >= 0.64 && orc_point_in_ellipse((float)bx, (float)by, cxf, cyf, aw, ah, angle);
}

int orc_match_ncc(const OrcFilter *f, const EkfPrediction *preds, int n_pred, EkfMatch *out)
{
    int nm = 0;
    for (int i = 0; i < n_pred; ++i) {
        int x, y;
        double key;
        if (ncc_match_one(f, &preds[i], &x, &y, &key)) {
            out[nm].featureIndex = preds[i].featureIndex;
            out[nm].keypointIndex = -1;
            out[nm].imagePos[0] = (double)x;
            out[nm].imagePos[1] = (double)y;
            out[nm].distance = (float)(1.0 - sqrt(key));
            out[nm]._pad = 0.f;
            ++nm;
        }
    }
    return nm;
}

/* ------------------------------------------------------------------------- new-feature detection (mode B) */
/* detectNewImageFeatures, EKF/DetectNewImageFeatures.cpp:337-367, on the current image (orc_image_set): mask =
 * uncertainty ellipses of the predictions (buildImageMask :102-127); detector = the build's own integer Harris-type
 * measure (the reference's OpenCV STAR is absent third-party code), best unmasked pixel of every 16x16 cell; zone
 * heuristic = searchFeaturesByZone :171-330 with its two nondeterministic choices fixed (stable zone order, strongest
 * candidate instead of rand()). */
static long long corner_response(int x, int y)
{
    long long sxx = 0, syy = 0, sxy = 0;
    for (int dy = -2; dy <= 2; ++dy)
        for (int dx = -2; dx <= 2; ++dx) {
            int cx = x + dx, cy = y + dy;
            int a = img_at(0, cx - 1, cy - 1), b = img_at(0, cx, cy - 1), c = img_at(0, cx + 1, cy - 1);
            int d = img_at(0, cx - 1, cy), e2 = img_at(0, cx + 1, cy);
            int g = img_at(0, cx - 1, cy + 1), h = img_at(0, cx, cy + 1), k = img_at(0, cx + 1, cy + 1);
            long long ix = (c + 2 * e2 + k) - (a + 2 * d + g);
            long long iy = (g + 2 * h + k) - (a + 2 * b + c);
            sxx += ix * ix; syy += iy * iy; sxy += ix * iy;
        }
    long long tr = sxx + syy;
    return 16 * (sxx * syy - sxy * sxy) - tr * tr;
}

typedef struct { int x, y; long long r; } OrcCand;

int orc_detect_new_features(const EkfPrediction *preds, int n_pred, int max_new, int divide_times,
                            double mask_ellipse_size, double min_response, double *uv_out)
{
    const int w = g_img.w[0], h = g_img.h[0];
    const int cells_x = w / 16, cells_y = h / 16, ncell = cells_x * cells_y;
    if (ncell <= 0 || max_new <= 0) return 0;
    float *axes = (float *)malloc(sizeof(float) * 2 * (size_t)(n_pred + 1));
    double *ang = (double *)malloc(sizeof(double) * (size_t)(n_pred + 1));
    for (int k = 0; k < n_pred; ++k) orc_ellipse(preds[k].covarianceMatrix, &axes[2 * k], &ang[k]);
    OrcCand *cands = (OrcCand *)malloc(sizeof(OrcCand) * (size_t)ncell);
    int nc = 0;
    const long long thr = min_response >= 9.2e18 ? 0x7fffffffffffffffLL : (min_response <= 0 ? 0 : (long long)min_response);
    for (int c = 0; c < ncell; ++c) {
        const int cx0 = (c % cells_x) * 16, cy0 = (c / cells_x) * 16;
        long long best = -1;
        int bx = cx0, by = cy0;
        for (int ly = 0; ly < 16; ++ly)
            for (int lx = 0; lx < 16; ++lx) {
                const int x = cx0 + lx, y = cy0 + ly;
                if (x < 16 || y < 16 || x >= w - 16 || y >= h - 16) continue;
                int masked = 0;
                for (int k = 0; k < n_pred && !masked; ++k)
                    masked = orc_point_in_ellipse((float)x, (float)y, (float)preds[k].imagePos[0], (float)preds[k].imagePos[1],
                                                  (int)lrintf(axes[2 * k]), (int)lrintf(axes[2 * k + 1]), ang[k]);
                if (masked) continue;
                long long r = corner_response(x, y);
                if (r > best) { best = r; bx = x; by = y; }
            }
        if (best >= 0 && best >= thr) { cands[nc].x = bx; cands[nc].y = by; cands[nc].r = best; ++nc; }
    }
    free(axes); free(ang);
    int n_out = 0;
    if (nc <= max_new) {
        for (int i = 0; i < nc; ++i) { uv_out[2 * i] = cands[i].x; uv_out[2 * i + 1] = cands[i].y; }
        n_out = nc;
        free(cands);
        return n_out;
    }
    const int zones_row = 1 << divide_times, nzone = zones_row * zones_row;
    const int zw = w / zones_row > 0 ? w / zones_row : 1, zh = h / zones_row > 0 ? h / zones_row : 1;
    int *zcount = (int *)calloc((size_t)nzone, sizeof(int));
    int *order = (int *)malloc(sizeof(int) * (size_t)nzone);
    int *czone = (int *)malloc(sizeof(int) * (size_t)nc);
    char *used = (char *)calloc((size_t)nc, 1);
#define ZONE_OF(px, py) clampi(((int)(py) / zh) * (w / zw) + (int)(px) / zw, 0, nzone - 1)
    for (int i = 0; i < nc; ++i) czone[i] = ZONE_OF(cands[i].x, cands[i].y);
    for (int k = 0; k < n_pred; ++k) zcount[ZONE_OF((float)preds[k].imagePos[0], (float)preds[k].imagePos[1])]++;
    for (int z = 0; z < nzone; ++z) order[z] = z;
    for (int a = 1; a < nzone; ++a) { /* stable insertion sort by population */
        int v = order[a], b = a - 1;
        while (b >= 0 && zcount[order[b]] > zcount[v]) { order[b + 1] = order[b]; --b; }
        order[b + 1] = v;
    }
    const int radius = (int)lrintf((float)(2.0 * sqrt(mask_ellipse_size * EKF_CHISQ_95_2)));
    int *ax = (int *)malloc(sizeof(int) * 2 * (size_t)max_new);
    int left = max_new, head = 0;
    while (head < nzone && left > 0) {
        const int z = order[head];
        int best = -1;
        for (int i = 0; i < nc; ++i)
            if (!used[i] && czone[i] == z && (best < 0 || cands[i].r > cands[best].r)) best = i;
        if (best < 0) { ++head; continue; }
        int free_px = 1;
        for (int a = 0; a < n_out && free_px; ++a) {
            double dx = (double)(float)cands[best].x - (double)(float)ax[2 * a], dy = (double)(float)cands[best].y - (double)(float)ax[2 * a + 1];
            if (2.0 * sqrt(dx * dx + dy * dy) <= 2.0 * radius) free_px = 0;
        }
        if (free_px) {
            uv_out[2 * n_out] = cands[best].x; uv_out[2 * n_out + 1] = cands[best].y;
            ax[2 * n_out] = cands[best].x; ax[2 * n_out + 1] = cands[best].y;
            ++n_out;
            zcount[z]++;
            for (int p = head; p + 1 < nzone; ++p) {
                if (zcount[order[p]] >= zcount[order[p + 1]]) { int t = order[p]; order[p] = order[p + 1]; order[p + 1] = t; }
                else break;
            }
            --left;
        }
        used[best] = 1;
    }
#undef ZONE_OF
    free(zcount); free(order); free(czone); free(used); free(ax); free(cands);
    return n_out;
}

/* ------------------------------------------------------------------------------------------ state update */

/* stateUpdate: EKF/Update.cpp:147-204 -- apply K*nu with the DELTA dead-band on every component */
static void apply_delta(double *x13, double *R, double *fpos, const int32_t *ftype, int N, const double *dx)
{
    for (int i = 0; i < 3; ++i)
        if (fabs(dx[i]) > EKF_DELTA) x13[i] += dx[i];
    for (int i = 0; i < 4; ++i)
        if (fabs(dx[i + 3]) > EKF_DELTA) x13[3 + i] += dx[i + 3];
    quat_to_rot(&x13[3], R); /* state.setOrientation(state.orientation) :168 */
    for (int i = 0; i < 3; ++i)
        if (fabs(dx[i + 7]) > EKF_DELTA) x13[7 + i] += dx[i + 7];
    for (int i = 0; i < 3; ++i)
        if (fabs(dx[i + 10]) > EKF_DELTA) x13[10 + i] += dx[i + 10];
    int acc = 13;
    for (int i = 0; i < N; ++i) {
        int d = feat_dim(ftype[i]);
        for (int j = 0; j < d; ++j) {
            if (fabs(dx[acc]) > EKF_DELTA) fpos[6 * i + j] += dx[acc];
            ++acc;
        }
    }
}

/* innovation with dead-band: EKF/Update.cpp:125-135 */
static void innovation(const EkfMatch *m, const EkfPrediction *p, double *nu)
{
    double ax = m->imagePos[0] - p->imagePos[0];
    double ay = m->imagePos[1] - p->imagePos[1];
    nu[0] = fabs(ax) > EKF_DELTA ? ax : 0.0;
    nu[1] = fabs(ay) > EKF_DELTA ? ay : 0.0;
}

/* G column pair of one feature: (P H_i')  (n x 2), using only H_i's structurally non-zero columns.
 * EKF/Update.cpp:105 with H_i dense 2 x n whose only non-zeros are Hs[:,0:13] and Hf[:,0:d] at pos. */
static void pht_feature(const OrcFilter *f, int fi, const double *hs, const double *hf, double *G /* n x 2 */)
{
    int n = f->n, d = feat_dim(f->ftype[fi]), pos = f->fcovpos[fi];
    for (int r = 0; r < n; ++r) {
        const double *prow = f->P + (size_t)r * n;
        for (int c = 0; c < 2; ++c) {
            double s = 0.0;
            for (int a = 0; a < 13; ++a) s += prow[a] * hs[c * 13 + a];
            for (int a = 0; a < d; ++a) s += prow[pos + a] * hf[c * 6 + a];
            G[(size_t)r * 2 + c] = s;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ RANSAC */

/* matchesBelowAThreshold: EKF/1PointRansac.cpp:48-84.  Returns support size; writes match indexes. */
static int matches_below_threshold(const EkfMatch *matches, int M, const EkfPrediction *preds, int np, double thr,
                                   int32_t *support)
{
    int ns = 0;
    for (int i = 0; i < np; ++i) {
        int found = 0, mi = 0;
        while (!found && mi < M) {
            if (matches[mi].featureIndex == preds[i].featureIndex) {
                double xd = matches[mi].imagePos[0] - preds[i].imagePos[0];
                double yd = matches[mi].imagePos[1] - preds[i].imagePos[1];
                double dist = sqrt(xd * xd + yd * yd);
                if (dist < thr) support[ns++] = mi;
                found = 1;
            }
            ++mi;
        }
    }
    return ns;
}

/* ransac: EKF/1PointRansac.cpp:101-234 */
int orc_ransac(OrcFilter *f, const EkfPrediction *preds, const double *Hs, const double *Hf,
               const EkfMatch *matches, int M, uint8_t *inlier_mask, int32_t *support_counts)
{
    if (M == 0) return 0;
    int n = f->n, N = f->N;
    unsigned n_hyp = 1000;
    double thr = f->par.ransacThresholdPredictDistance;
    int32_t *best = (int32_t *)malloc((size_t)(M + 1) * sizeof(int32_t));
    int32_t *sup = (int32_t *)malloc((size_t)(M + N + 1) * sizeof(int32_t));
    int n_best = 0;
    double *G = (double *)malloc((size_t)n * 2 * sizeof(double));
    double *dx = (double *)malloc((size_t)n * sizeof(double));
    double *tpos = (double *)malloc((size_t)(6 * N + 6) * sizeof(double));
    EkfPrediction *tp = (EkfPrediction *)malloc((size_t)(N + 1) * sizeof(EkfPrediction));
    unsigned i;
    for (i = 0; i < n_hyp && i < (unsigned)M; ++i) {
        int fi = preds[i].featureIndex;
        const double *hs = Hs + (size_t)26 * i, *hf = Hf + (size_t)12 * i;
        int d = feat_dim(f->ftype[fi]), pos = f->fcovpos[fi];
        /* State temporalState(state) :147 ; updateOnlyState :149 -> EKF/Update.cpp:237-265 with one match */
        double tx[13], tR[9];
        memcpy(tx, f->x, sizeof(tx));
        memcpy(tR, f->R, sizeof(tR));
        memcpy(tpos, f->fpos, (size_t)6 * N * sizeof(double));
        pht_feature(f, fi, hs, hf, G);
        double S[4], Si[4];
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 2; ++c) {
                double s = 0.0;
                for (int a = 0; a < 13; ++a) s += hs[r * 13 + a] * G[(size_t)a * 2 + c];
                for (int a = 0; a < d; ++a) s += hf[r * 6 + a] * G[(size_t)(pos + a) * 2 + c];
                S[r * 2 + c] = s + (r == c ? f->cam.pixelErrorX : 0.0); /* R = I * pixelErrorX, Update.cpp:95-97 */
            }
        inv2x2(S, Si);
        double nu[2];
        innovation(&matches[i], &preds[i], nu);
        for (int r = 0; r < n; ++r) {
            double k0 = G[(size_t)r * 2] * Si[0] + G[(size_t)r * 2 + 1] * Si[2];
            double k1 = G[(size_t)r * 2] * Si[1] + G[(size_t)r * 2 + 1] * Si[3];
            dx[r] = k0 * nu[0] + k1 * nu[1];
        }
        apply_delta(tx, tR, tpos, f->ftype, N, dx);
        /* predictMeasurementState(temporalState, all features) :156 */
        int np = orc_predict_measurement_state(f, tx, tR, tpos, NULL, 0, tp);
        int ns = matches_below_threshold(matches, M, tp, np, thr, sup);
        if (support_counts) support_counts[i] = ns;
        if (ns > n_best) { /* strict: ties keep the earlier hypothesis :164 */
            memcpy(best, sup, (size_t)ns * sizeof(int32_t));
            n_best = ns;
            double e = 1.0 - (double)n_best / (double)M;
            n_hyp = (unsigned)(int)(log(1.0 - f->par.ransacAllInliersProbability) / log(1.0 - (1.0 - e)));
        }
    }
    memset(inlier_mask, 0, (size_t)M);
    for (int k = 0; k < n_best; ++k) inlier_mask[best[k]] = 1;
    free(best); free(sup); free(G); free(dx); free(tpos); free(tp);
    return (int)i;
}

/* ------------------------------------------------------------------------------------------------ update */

/* normalizeQuaternionJacobian: EKF/Update.cpp:45-60 */
static double quat_norm_jacobian(const double *q, double *J)
{
    double r = q[0], x = q[1], y = q[2], z = q[3];
    double norm = sqrt(r * r + x * x + y * y + z * z);
    double a = 1.0 / (pow(norm, 3));
    double M[16] = {x * x + y * y + z * z, -r * x, -r * y, -r * z,
                    -x * r, r * r + y * y + z * z, -x * y, -x * z,
                    -y * r, -y * x, r * r + x * x + z * z, -y * z,
                    -z * r, -z * x, -z * y, r * r + x * x + y * y};
    for (int i = 0; i < 16; ++i) J[i] = M[i] * a;
    return norm;
}

/* normalizeCovariance: EKF/Update.cpp:64-85 (five disjoint blocks, so the in-place order is immaterial) */
static void normalize_covariance(double *P, int n, const double *J)
{
    double Jt[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) Jt[j * 4 + i] = J[i * 4 + j];
    double t[16], u[16];
    /* P[0:3,3:7] = P[0:3,3:7] J' */
    double b1[12];
    gemm_ikj(3, 4, 4, P + 3, n, Jt, 4, b1, 4);
    /* P[3:7,0:3] = J P[3:7,0:3] */
    double b2[12];
    gemm_ikj(4, 4, 3, J, 4, P + (size_t)3 * n, n, b2, 3);
    /* P[3:7,3:7] = J P[3:7,3:7] J' */
    gemm_ikj(4, 4, 4, J, 4, P + (size_t)3 * n + 3, n, t, 4);
    gemm_ikj(4, 4, 4, t, 4, Jt, 4, u, 4);
    /* P[3:7,7:] = J P[3:7,7:] */
    int w = n - 7;
    double *b4 = (double *)malloc((size_t)4 * (w > 0 ? w : 1) * sizeof(double));
    if (w > 0) gemm_ikj(4, 4, w, J, 4, P + (size_t)3 * n + 7, n, b4, w);
    /* P[7:,3:7] = P[7:,3:7] J' */
    for (int i = 7; i < n; ++i) {
        double row[4], o[4] = {0, 0, 0, 0};
        memcpy(row, P + (size_t)i * n + 3, sizeof(row));
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) o[j] += row[k] * Jt[k * 4 + j];
        memcpy(P + (size_t)i * n + 3, o, sizeof(o));
    }
    for (int i = 0; i < 3; ++i) memcpy(P + (size_t)i * n + 3, b1 + i * 4, 4 * sizeof(double));
    for (int i = 0; i < 4; ++i) memcpy(P + (size_t)(3 + i) * n, b2 + i * 3, 3 * sizeof(double));
    for (int i = 0; i < 4; ++i) memcpy(P + (size_t)(3 + i) * n + 3, u + i * 4, 4 * sizeof(double));
    if (w > 0)
        for (int i = 0; i < 4; ++i) memcpy(P + (size_t)(3 + i) * n + 7, b4 + (size_t)i * w, (size_t)w * sizeof(double));
    free(b4);
}

/* updateStateAndCovariance, literal: EKF/Update.cpp:237-265 (joinJacobians :222, determineKalmanGain :92,
 * stateUpdate :116, covarianceUpdate :214) */
static int update_literal(OrcFilter *f, const EkfMatch *matches, const EkfPrediction *preds, const double *Hs,
                          const double *Hf, int M)
{
    int n = f->n, m = 2 * M;
    size_t nn = (size_t)n * n, nm = (size_t)n * m;
    double *H = (double *)calloc((size_t)m * n, sizeof(double));
    double *Ht = (double *)malloc(nm * sizeof(double));
    double *G = (double *)malloc(nm * sizeof(double));
    double *S = (double *)malloc((size_t)m * m * sizeof(double));
    double *Si = (double *)malloc((size_t)m * m * sizeof(double));
    double *K = (double *)malloc(nm * sizeof(double));
    double *KH = (double *)malloc(nn * sizeof(double));
    double *Pn = (double *)malloc(nn * sizeof(double));
    double *nu = (double *)malloc((size_t)m * sizeof(double));
    double *dx = (double *)malloc((size_t)n * sizeof(double));
    if (!H || !Ht || !G || !S || !Si || !K || !KH || !Pn || !nu || !dx) return EKF_ERR_CAPACITY;
    for (int i = 0; i < M; ++i) { /* joinJacobians */
        int fi = preds[i].featureIndex;
        int d = feat_dim(f->ftype[fi]), pos = f->fcovpos[fi];
        for (int r = 0; r < 2; ++r) {
            double *row = H + (size_t)(2 * i + r) * n;
            for (int a = 0; a < 13; ++a) row[a] = Hs[(size_t)26 * i + r * 13 + a];
            for (int a = 0; a < d; ++a) row[pos + a] = Hf[(size_t)12 * i + r * 6 + a];
        }
    }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) Ht[(size_t)j * m + i] = H[(size_t)i * n + j];
    gemm_ikj(n, n, m, f->P, n, Ht, m, G, m);          /* stateCovMultPredJacob = P H'   :105 */
    gemm_ikj(m, n, m, H, n, G, m, S, m);              /* H (P H')                       :107 */
    for (int i = 0; i < m; ++i) S[(size_t)i * m + i] += f->cam.pixelErrorX; /* + R, R = I*pixelErrorX :95-97 */
    int ok;
    if (m == 2) ok = inv2x2(S, Si);
    else if (m == 3) ok = inv3x3(S, Si);
    else ok = inv_lu(S, m, Si);                        /* S.inv()                        :108 */
    gemm_ikj(n, m, m, G, m, Si, m, K, m);             /* K = (P H') inv(S)              :108 */
    for (int i = 0; i < M; ++i) innovation(&matches[i], &preds[i], &nu[2 * i]);
    for (int r = 0; r < n; ++r) {                     /* K * nu                         :144 */
        double s = 0.0;
        for (int k = 0; k < m; ++k) s += K[(size_t)r * m + k] * nu[k];
        dx[r] = s;
    }
    apply_delta(f->x, f->R, f->fpos, f->ftype, f->N, dx);
    gemm_ikj(n, m, n, K, m, H, n, KH, n);             /* K H                            :216 */
    for (size_t i = 0; i < nn; ++i) KH[i] = -KH[i];
    for (int i = 0; i < n; ++i) KH[(size_t)i * n + i] += 1.0; /* I - K H */
    gemm_ikj(n, n, n, KH, n, f->P, n, Pn, n);         /* (I - K H) P                    :216-217 */
    memcpy(f->P, Pn, nn * sizeof(double));
    free(H); free(Ht); free(G); free(S); free(Si); free(K); free(KH); free(Pn); free(nu); free(dx);
    return ok ? EKF_OK : EKF_ERR_NOT_POSITIVE_DEFINITE;
}

/* Same mathematics without the dense detours: A = H P (M row pairs, 2 x n each), S = A H' + R, Cholesky
 * S = L L', B = inv(L) A, z = inv(L) nu, dx = B' z, P <- P - B' B.  Used for map sizes where the literal form
 * needs minutes (goldens at N >= 1000) and as the secondary CPU baseline. */
static int update_algorithmic(OrcFilter *f, const EkfMatch *matches, const EkfPrediction *preds, const double *Hs,
                              const double *Hf, int M)
{
    int n = f->n, m = 2 * M;
    double *A = (double *)malloc((size_t)m * n * sizeof(double));
    double *S = (double *)malloc((size_t)m * m * sizeof(double));
    double *z = (double *)malloc((size_t)m * sizeof(double));
    double *dx = (double *)malloc((size_t)n * sizeof(double));
    if (!A || !S || !z || !dx) return EKF_ERR_CAPACITY;
    for (int i = 0; i < M; ++i) {
        int fi = preds[i].featureIndex;
        int d = feat_dim(f->ftype[fi]), pos = f->fcovpos[fi];
        for (int r = 0; r < 2; ++r) {
            double *o = A + (size_t)(2 * i + r) * n;
            for (int j = 0; j < n; ++j) o[j] = 0.0;
            for (int a = 0; a < 7; ++a) { /* Hs columns 7..12 are structurally zero */
                double h = Hs[(size_t)26 * i + r * 13 + a];
                const double *prow = f->P + (size_t)a * n;
                for (int j = 0; j < n; ++j) o[j] += h * prow[j];
            }
            for (int a = 0; a < d; ++a) {
                double h = Hf[(size_t)12 * i + r * 6 + a];
                const double *prow = f->P + (size_t)(pos + a) * n;
                for (int j = 0; j < n; ++j) o[j] += h * prow[j];
            }
        }
    }
    for (int i = 0; i < m; ++i) {
        const double *ai = A + (size_t)i * n;
        for (int jb = 0; jb < M; ++jb) {
            int fj = preds[jb].featureIndex;
            int d = feat_dim(f->ftype[fj]), pos = f->fcovpos[fj];
            for (int r = 0; r < 2; ++r) {
                double s = 0.0;
                for (int a = 0; a < 7; ++a) s += ai[a] * Hs[(size_t)26 * jb + r * 13 + a];
                for (int a = 0; a < d; ++a) s += ai[pos + a] * Hf[(size_t)12 * jb + r * 6 + a];
                S[(size_t)i * m + 2 * jb + r] = s;
            }
        }
        S[(size_t)i * m + i] += f->cam.pixelErrorX;
    }
    for (int i = 0; i < M; ++i) innovation(&matches[i], &preds[i], &z[2 * i]);
    /* Cholesky (lower), in place, fused with the forward substitutions on A and nu */
    int status = EKF_OK;
    for (int j = 0; j < m && status == EKF_OK; ++j) {
        double djj = S[(size_t)j * m + j];
        for (int k = 0; k < j; ++k) djj -= S[(size_t)j * m + k] * S[(size_t)j * m + k];
        if (!(djj > 0.0)) { status = EKF_ERR_NOT_POSITIVE_DEFINITE; break; }
        djj = sqrt(djj);
        S[(size_t)j * m + j] = djj;
        for (int i = j + 1; i < m; ++i) {
            double s = S[(size_t)i * m + j];
            for (int k = 0; k < j; ++k) s -= S[(size_t)i * m + k] * S[(size_t)j * m + k];
            S[(size_t)i * m + j] = s / djj;
        }
    }
    if (status == EKF_OK) {
        enum { JB = 256, KB = 256 };
        for (int i = 0; i < m; ++i) { /* z = inv(L) nu */
            double zi = z[i];
            for (int k = 0; k < i; ++k) zi -= S[(size_t)i * m + k] * z[k];
            z[i] = zi / S[(size_t)i * m + i];
        }
        for (int jb = 0; jb < n; jb += JB) { /* B = inv(L) A, one cache-sized column panel at a time */
            int je = jb + JB < n ? jb + JB : n;
            for (int i = 0; i < m; ++i) {
                double *bi = A + (size_t)i * n;
                for (int k = 0; k < i; ++k) {
                    double l = S[(size_t)i * m + k];
                    const double *bk = A + (size_t)k * n;
                    for (int j = jb; j < je; ++j) bi[j] -= l * bk[j];
                }
                double inv = 1.0 / S[(size_t)i * m + i];
                for (int j = jb; j < je; ++j) bi[j] *= inv;
            }
        }
        for (int j = 0; j < n; ++j) dx[j] = 0.0;
        for (int k = 0; k < m; ++k) {
            const double *bk = A + (size_t)k * n;
            double zk = z[k];
            for (int j = 0; j < n; ++j) dx[j] += bk[j] * zk;
        }
        apply_delta(f->x, f->R, f->fpos, f->ftype, f->N, dx);
        /* P <- sym(P) - B' B.  (I-KH)P followed by 0.5(P+P') equals 0.5(P+P') - B'B when B'B is symmetric,
         * so average first, subtract on the upper triangle, mirror. */
        double *P = f->P;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) {
                double v = 0.5 * P[(size_t)i * n + j] + 0.5 * P[(size_t)j * n + i];
                P[(size_t)i * n + j] = v;
            }
        for (int jb = 0; jb < n; jb += JB) {
            int je = jb + JB < n ? jb + JB : n;
            for (int kb = 0; kb < m; kb += KB) {
                int ke = kb + KB < m ? kb + KB : m;
                for (int i = 0; i < je; ++i) {
                    double *prow = P + (size_t)i * n;
                    int j0 = jb > i ? jb : i;
                    for (int k = kb; k < ke; ++k) {
                        const double *bk = A + (size_t)k * n;
                        double bi = bk[i];
                        for (int j = j0; j < je; ++j) prow[j] -= bi * bk[j];
                    }
                }
            }
        }
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) P[(size_t)j * n + i] = P[(size_t)i * n + j];
    }
    free(A); free(S); free(z); free(dx);
    return status;
}

/* update: EKF/Update.cpp:282-319 */
int orc_update(OrcFilter *f, const EkfMatch *matches, const EkfPrediction *preds, const double *Hs,
               const double *Hf, int M, int variant)
{
    if (M <= 0) return EKF_OK; /* :292 */
    int n = f->n;
    int st = (variant == ORC_UPDATE_LITERAL) ? update_literal(f, matches, preds, Hs, Hf, M)
                                             : update_algorithmic(f, matches, preds, Hs, Hf, M);
    /* P = 0.5 P + 0.5 P'  :303 */
    double *P = f->P;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            double v = 0.5 * P[(size_t)i * n + j] + 0.5 * P[(size_t)j * n + i];
            P[(size_t)i * n + j] = v;
            P[(size_t)j * n + i] = v;
        }
    double J[16];
    double qn = quat_norm_jacobian(&f->x[3], J); /* from the un-normalised q :305 */
    double q[4] = {f->x[3] / qn, f->x[4] / qn, f->x[5] / qn, f->x[6] / qn};
    set_orientation(f->x, f->R, q);
    normalize_covariance(P, n, J);
    return st;
}

/* rescueOutliers: EKF/EKF.cpp:68-119 */
int orc_rescue(const OrcFilter *f, const EkfMatch *matches, const EkfPrediction *preds, int M,
               uint8_t *rescued_mask)
{
    int nr = 0;
    for (int i = 0; i < M; ++i) {
        double d0 = matches[i].imagePos[0] - preds[i].imagePos[0];
        double d1 = matches[i].imagePos[1] - preds[i].imagePos[1];
        double Si[4];
        inv2x2(preds[i].covarianceMatrix, Si);
        /* dist' * inv(S_i) * dist  :94 : (1x2 * 2x2) then * 2x1 */
        double t0 = d0 * Si[0] + d1 * Si[2];
        double t1 = d0 * Si[1] + d1 * Si[3];
        double v = t0 * d0 + t1 * d1;
        rescued_mask[i] = v < f->par.ransacChi2Threshold ? 1 : 0;
        nr += rescued_mask[i];
    }
    return nr;
}

/* -------------------------------------------------------------------------------------------------- step */

/* EKF::step: EKF/EKF.cpp:242-556 with a fixed map (map management :572-612 and logging are out of scope) */
static int orc_step_impl(OrcFilter *f, const EkfKeypoint *kps, const uint8_t *desc32, int n_kp, int variant,
                         OrcStepInfo *info, int use_ncc)
{
    int N = f->N, status = EKF_OK;
    OrcStepInfo li;
    memset(&li, 0, sizeof(li));
    EkfPrediction *preds = (EkfPrediction *)malloc((size_t)(N + 1) * sizeof(EkfPrediction));
    double *Hs = (double *)malloc((size_t)(N + 1) * 26 * sizeof(double));
    double *Hf = (double *)malloc((size_t)(N + 1) * 12 * sizeof(double));
    EkfMatch *matches = (EkfMatch *)malloc((size_t)(N + 1) * sizeof(EkfMatch));
    EkfPrediction *mp = (EkfPrediction *)malloc((size_t)(N + 1) * sizeof(EkfPrediction));
    double *mHs = (double *)malloc((size_t)(N + 1) * 26 * sizeof(double));
    double *mHf = (double *)malloc((size_t)(N + 1) * 12 * sizeof(double));
    EkfMatch *sel = (EkfMatch *)malloc((size_t)(N + 1) * sizeof(EkfMatch));
    uint8_t *mask = (uint8_t *)malloc((size_t)N + 1);
    int32_t *oidx = (int32_t *)malloc((size_t)(N + 1) * sizeof(int32_t));

    orc_predict(f, NULL, NULL);                                              /* :273 */
    int np = orc_predict_measurements(f, NULL, 0, preds, Hs, Hf, NULL);      /* :278 */
    li.n_predicted = np;
    /* updateMapFeatures, first loop (EKF/MapManagement.cpp:81-86): every predicted feature counts */
    for (int k = 0; k < np; ++k) f->ftimes_predicted[preds[k].featureIndex]++;
    int M = use_ncc ? orc_match_ncc(f, preds, np, matches) : orc_match(f, preds, np, kps, desc32, n_kp, matches); /* :337 */
    li.n_matches = M;
    /* predictions/Jacobians re-ordered to match order :368-392 */
    for (int i = 0; i < M; ++i) {
        for (int k = 0; k < np; ++k)
            if (preds[k].featureIndex == matches[i].featureIndex) {
                mp[i] = preds[k];
                memcpy(mHs + (size_t)26 * i, Hs + (size_t)26 * k, 26 * sizeof(double));
                memcpy(mHf + (size_t)12 * i, Hf + (size_t)12 * k, 12 * sizeof(double));
                break;
            }
    }
    li.n_hypotheses = orc_ransac(f, mp, mHs, mHf, matches, M, mask, NULL);   /* :402 */
    /* inliers (original order) :213-227 */
    int ni = 0, no = 0;
    EkfMatch *outl = (EkfMatch *)malloc((size_t)(N + 1) * sizeof(EkfMatch));
    for (int i = 0; i < M; ++i) {
        if (mask[i]) {
            sel[ni] = matches[i];
            preds[ni] = mp[i]; /* preds[] no longer needed: reuse as inlier predictions */
            memmove(Hs + (size_t)26 * ni, mHs + (size_t)26 * i, 26 * sizeof(double));
            memmove(Hf + (size_t)12 * ni, mHf + (size_t)12 * i, 12 * sizeof(double));
            ++ni;
        } else {
            outl[no] = matches[i];
            oidx[no] = matches[i].featureIndex;
            ++no;
        }
    }
    li.n_inliers = ni;
    li.n_outliers = no;
    /* updateMapFeatures for the low-innovation inliers (EKF.cpp:572, MapManagement.cpp:88-113): matched counter and
     * the map descriptor replaced by the matched keypoint's; it runs after both updates in the reference, which is
     * equivalent because nothing in between reads descriptors or counters */
    for (int i = 0; i < ni; ++i) {
        int fi = sel[i].featureIndex;
        f->ftimes_matched[fi]++;
        if (!use_ncc) memcpy(&f->fdesc[(size_t)fi * f->desc_bytes], &desc32[(size_t)sel[i].keypointIndex * f->desc_bytes], (size_t)f->desc_bytes);
    }
    int st = orc_update(f, sel, preds, Hs, Hf, ni, variant);                 /* :430 */
    if (st != EKF_OK) status = st;
    /* re-predict the outliers with the updated state and covariance :473 */
    int nop = 0;
    if (no > 0) nop = orc_predict_measurements(f, oidx, no, mp, mHs, mHf, NULL);
    /* keep only the outlier matches whose prediction fell inside the frame :483-499 */
    if (0 < nop && nop < no) {
        int j = 0, kept = 0;
        for (int i = 0; i < no && j < nop; ++i)
            if (outl[i].featureIndex == mp[j].featureIndex) { outl[kept++] = outl[i]; ++j; }
        no = kept;
    }
    int nr = 0;
    /* the reference calls rescueOutliers whenever outlierMatches is non-empty (:501) and indexes the
     * prediction vector out of bounds if it is empty; that undefined case is treated as "nothing rescued" */
    if (no > 0 && nop > 0) {
        orc_rescue(f, outl, mp, no, mask);                                   /* :504 */
        for (int i = 0; i < no; ++i)
            if (mask[i]) {
                sel[nr] = outl[i];
                preds[nr] = mp[i];
                memmove(Hs + (size_t)26 * nr, mHs + (size_t)26 * i, 26 * sizeof(double));
                memmove(Hf + (size_t)12 * nr, mHf + (size_t)12 * i, 12 * sizeof(double));
                ++nr;
            }
    }
    li.n_rescued = nr;
    for (int i = 0; i < nr; ++i) { /* rescued matches join the inliers (EKF.cpp:552-556) */
        int fi = sel[i].featureIndex;
        f->ftimes_matched[fi]++;
        if (!use_ncc) memcpy(&f->fdesc[(size_t)fi * f->desc_bytes], &desc32[(size_t)sel[i].keypointIndex * f->desc_bytes], (size_t)f->desc_bytes);
    }
    if (nr > 0) {                                                            /* :529-532 */
        st = orc_update(f, sel, preds, Hs, Hf, nr, variant);
        if (st != EKF_OK) status = st;
    }
    li.status = status;
    if (info) *info = li;
    free(preds); free(Hs); free(Hf); free(matches); free(mp); free(mHs); free(mHf); free(sel); free(mask);
    free(oidx); free(outl);
    return status;
}

int orc_step(OrcFilter *f, const EkfKeypoint *kps, const uint8_t *desc32, int n_kp, int variant, OrcStepInfo *info)
{
    return orc_step_impl(f, kps, desc32, n_kp, variant, info, 0);
}

/* EKF::step with the NCC matcher on the current image (orc_image_set); templates stay fixed */
int orc_step_image(OrcFilter *f, int variant, OrcStepInfo *info)
{
    return orc_step_impl(f, NULL, NULL, 0, variant, info, 1);
}

/* --------------------------------------------------------------------------------------- timing helper */

double orc_time_literal_rows(int n, int m, int rows, const double *K, const double *H, const double *P,
                             double *out_rows)
{
    struct timespec t0, t1;
    double *KH = (double *)malloc((size_t)rows * n * sizeof(double));
    clock_gettime(CLOCK_MONOTONIC, &t0);
    gemm_ikj(rows, m, n, K, m, H, n, KH, n);
    for (size_t i = 0; i < (size_t)rows * n; ++i) KH[i] = -KH[i];
    for (int i = 0; i < rows; ++i) KH[(size_t)i * n + i] += 1.0;
    gemm_ikj(rows, n, n, KH, n, P, n, out_rows, n);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(KH);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
