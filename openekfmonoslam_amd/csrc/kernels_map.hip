// kernels_map.hip -- map management on the device, i.e. the part of EKF::step that changes the
// state dimension between frames (EKF/EKF.cpp:574-612):
//   addFeaturesToStateAndCovariance   EKF/AddMapFeature.cpp:221-359   -> k_addfeat_prepare / _rows / _corner
//   removeFeaturesFromStateAndCovariance EKF/MapManagement.cpp:212-259 -> k_compact_P (+ host-side SoA compaction)
//   computeLinearityIndex / convertToDepth EKF/MapManagement.cpp:312-490 -> k_linearity, k_convert_*
// The reference re-allocates and copies all of P for every added feature (O(n^2) each); here P lives in a
// capacity-strided buffer, an append is two 6 x n strip products per feature (all new features in one launch),
// a removal is one gather pass into a second buffer.  All strip kernels are HBM/latency-bound.
#include "engine.h"

namespace ekf {

// ---------------------------------------------------------------------------------------------- add features
// One thread per new feature: initial 6-vector, d f/d (r, q) (6x7) and d f/d (h, rho) (6x3).
// EKF/AddMapFeature.cpp:42-58 (undistort), :65-90, :109-216, :293-344.
__global__ void __launch_bounds__(64)
k_addfeat_prepare(const double *st, CamD c, double rho0, const double *uv, int count, int N_old, int n_old,
                  double *feat_pos, int *feat_type, int *feat_covpos, double *Jpo, double *Jhr)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= count) return;
    const double *x = st + ST_X, *R = st + ST_R, *q = x + 3;
    const double ud = uv[2 * j], vd = uv[2 * j + 1];
    // undistortPoint
    const double mx = ud - c.cx, my = vd - c.cy;
    const double dxm = c.dx * mx, dym = c.dy * my;
    const double rd2 = dxm * dxm + dym * dym;
    const double dist = 1 + c.k1 * rd2 + c.k2 * rd2 * rd2;
    const double uu = c.cx + mx * dist, vu = c.cy + my * dist;
    const double xyz_c[3] = {-(c.cx - uu) / c.fx, -(c.cy - vu) / c.fy, 1.0};
    double g[3];
    mat3_vec(R, xyz_c, g);
    double *fp = feat_pos + 6 * (size_t)(N_old + j);
    fp[0] = x[0]; fp[1] = x[1]; fp[2] = x[2];
    fp[3] = atan2(g[0], g[2]);
    fp[4] = atan2(-g[1], sqrt(g[0] * g[0] + g[2] * g[2]));
    fp[5] = rho0;
    feat_type[N_old + j] = EKF_FEATURE_INVERSE_DEPTH;
    feat_covpos[N_old + j] = n_old + 6 * j;
    // Jacobians
    const double xw = g[0], yw = g[1], zw = g[2];
    const double xxzz = xw * xw + zw * zw, sq = sqrt(xxzz), nsq = xxzz + yw * yw;
    const double dth[3] = {zw / xxzz, 0.0, -xw / xxzz};
    const double dph[3] = {xw * yw / (nsq * sq), -sq / nsq, zw * yw / (nsq * sq)};
    double dg[12];
    jac_rot_by_quat(q, xyz_c, dg);
    double *J = Jpo + 42 * (size_t)j;
    for (int i = 0; i < 42; ++i) J[i] = 0.0;
    J[0] = J[8] = J[16] = 1.0;
    for (int i = 0; i < 4; ++i) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            s1 += dth[k] * dg[k * 4 + i];
            s2 += dph[k] * dg[k * 4 + i];
        }
        J[3 * 7 + 3 + i] = s1;
        J[4 * 7 + 3 + i] = s2;
    }
    double sub[6]; // [dtheta; dphi] * R
    for (int i = 0; i < 3; ++i) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            s1 += dth[k] * R[k * 3 + i];
            s2 += dph[k] * R[k * 3 + i];
        }
        sub[i] = s1;
        sub[3 + i] = s2;
    }
    const double s2m[4] = {sub[0] / c.fx, sub[1] / c.fy, sub[3] / c.fx, sub[4] / c.fy}; // * dgc_dhu
    // computeUndistortPointJacobian :65-90
    const double a = c.k1 + 2.0 * c.k2 * rd2, b = 1.0 + c.k1 * rd2 + c.k2 * rd2 * rd2;
    const double dx2 = 2.0 * c.dx * c.dx, dy2 = 2.0 * c.dy * c.dy;
    const double dh[4] = {b + mx * a * (mx * dx2), mx * a * (my * dy2), my * a * (mx * dx2), my * a * (my * dy2) + b};
    double *H = Jhr + 18 * (size_t)j;
    for (int i = 0; i < 18; ++i) H[i] = 0.0;
    H[9] = s2m[0] * dh[0] + s2m[1] * dh[2];
    H[10] = s2m[0] * dh[1] + s2m[1] * dh[3];
    H[12] = s2m[2] * dh[0] + s2m[3] * dh[2];
    H[13] = s2m[2] * dh[1] + s2m[3] * dh[3];
    H[17] = 1.0;
}

// New rows  P[n_old + 6j + a, c] = sum_k Jpo_j[a][k] P[k][c]  for the old columns c < n_old (:272), mirrored into the
// new columns (:273; P[0:n_old, 0:7] Jpo' is the transpose of the same products because P is symmetric).
template <typename T>
__global__ void __launch_bounds__(256) k_addfeat_rows(T *P, int ld, int n_old, const double *Jpo)
{
    __shared__ double sJ[42];
    const int j = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 42) sJ[threadIdx.x] = Jpo[42 * (size_t)j + threadIdx.x];
    __syncthreads();
    if (c >= n_old) return;
    double p[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) p[k] = (double)P[(size_t)k * ld + c];
    const int r0 = n_old + 6 * j;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) s += sJ[a * 7 + k] * p[k];
        P[(size_t)(r0 + a) * ld + c] = (T)s;
        P[(size_t)c * ld + r0 + a] = (T)s;
    }
}

// New x new blocks (j, i), i <= j:  Jpo_j P77 Jpo_i'  (+ Jhr N Jhr' on the diagonal, :279-281), upper + mirror.
template <typename T>
__global__ void __launch_bounds__(64)
k_addfeat_corner(T *P, int ld, int n_old, int count, const double *Jpo, const double *Jhr, double nx, double ny, double nr)
{
    const int j = blockIdx.x, i = blockIdx.y;
    if (i > j) return;
    __shared__ double C[49], W[42]; // P77, Jpo_j * P77 (6x7)
    const int t = threadIdx.x;
    if (t < 49) C[t] = (double)P[(size_t)(t / 7) * ld + t % 7];
    __syncthreads();
    const double *Jj = Jpo + 42 * (size_t)j, *Ji = Jpo + 42 * (size_t)i;
    if (t < 42) {
        const int a = t / 7, k = t % 7;
        double s = 0.0;
        for (int l = 0; l < 7; ++l) s += Jj[a * 7 + l] * C[l * 7 + k];
        W[t] = s;
    }
    __syncthreads();
    if (t < 36) {
        const int a = t / 6, b = t % 6;
        if (i == j && b < a) return;
        double s = 0.0;
        for (int k = 0; k < 7; ++k) s += W[a * 7 + k] * Ji[b * 7 + k];
        if (i == j) {
            const double *H = Jhr + 18 * (size_t)j;
            s += H[a * 3 + 0] * nx * H[b * 3 + 0] + H[a * 3 + 1] * ny * H[b * 3 + 1] + H[a * 3 + 2] * nr * H[b * 3 + 2];
        }
        const int r = n_old + 6 * j + a, cc = n_old + 6 * i + b;
        P[(size_t)r * ld + cc] = (T)s;
        P[(size_t)cc * ld + r] = (T)s;
    }
}

// ------------------------------------------------------------------------------------------ compaction of P
// P2[i][j] = P[new2old[i]][new2old[j]]  (removeRowsAndColumnsFromMat, EKF/MapManagement.cpp:168-208)
template <typename T>
__global__ void __launch_bounds__(256) k_compact_P(const T *P, T *P2, int ld, int n_new, const int *new2old)
{
    const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_new) return;
    P2[(size_t)i * ld + j] = P[(size_t)new2old[i] * ld + new2old[j]];
}

// -------------------------------------------------------------------------------- inverse depth -> depth
// computeLinearityIndex, EKF/MapManagement.cpp:312-341 (one thread per feature; depth features get +inf)
template <typename T>
__global__ void __launch_bounds__(256)
k_linearity(const double *st, const double *feat_pos, const int *feat_type, const int *feat_covpos, int N, const T *P, int ld,
            double *out)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= N) return;
    if (feat_type[f] != EKF_FEATURE_INVERSE_DEPTH) {
        out[f] = 1e300;
        return;
    }
    const double *p = feat_pos + 6 * (size_t)f, *x = st + ST_X;
    const int ii = feat_covpos[f] + 5;
    const double inv_err = sqrt((double)P[(size_t)ii * ld + ii]);
    const double sigma = inv_err / (p[5] * p[5]);
    double m[3];
    dir_vec(p[3], p[4], m);
    const double xyz[3] = {p[0] + m[0] / p[5], p[1] + m[1] / p[5], p[2] + m[2] / p[5]};
    const double tc[3] = {xyz[0] - x[0], xyz[1] - x[1], xyz[2] - x[2]};
    const double tf[3] = {xyz[0] - p[0], xyz[1] - p[1], xyz[2] - p[2]};
    double dot = 0.0;
    for (int k = 0; k < 3; ++k) dot += tc[k] * tf[k];
    const double df = sqrt(tf[0] * tf[0] + tf[1] * tf[1] + tf[2] * tf[2]);
    const double dc = sqrt(tc[0] * tc[0] + tc[1] * tc[1] + tc[2] * tc[2]);
    out[f] = 4.0 * sigma * (dot / (df * dc)) / dc;
}

// convertToDepth, EKF/MapManagement.cpp:343-392: XYZ position and the 3x6 Jacobian; one thread.
__global__ void k_convert_prepare(double *feat_pos, int *feat_type, int fi, double *J)
{
    if (threadIdx.x != 0) return;
    double *p = feat_pos + 6 * (size_t)fi;
    const double theta = p[3], phi = p[4], rho = p[5];
    double mi[3];
    dir_vec(theta, phi, mi);
    for (int i = 0; i < 18; ++i) J[i] = 0.0;
    J[0] = J[7] = J[14] = 1.0;
    J[0 * 6 + 3] = cos(phi) * cos(theta) / rho;  J[2 * 6 + 3] = -cos(phi) * sin(theta) / rho;
    J[0 * 6 + 4] = -sin(phi) * sin(theta) / rho; J[1 * 6 + 4] = -cos(phi) / rho; J[2 * 6 + 4] = -sin(phi) * cos(theta) / rho;
    J[0 * 6 + 5] = -mi[0] / (rho * rho); J[1 * 6 + 5] = -mi[1] / (rho * rho); J[2 * 6 + 5] = -mi[2] / (rho * rho);
    p[0] += mi[0] / rho; p[1] += mi[1] / rho; p[2] += mi[2] / rho;
    p[3] = p[4] = p[5] = 0.0;
    feat_type[fi] = EKF_FEATURE_DEPTH;
}

// T3[a][c] = sum_k J[a][k] P[pos+k][c]   (subResult_3xn, :436)
template <typename T>
__global__ void __launch_bounds__(256) k_convert_rows(const T *P, int ld, int n, int pos, const double *J, double *T3)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    double p[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) p[k] = (double)P[(size_t)(pos + k) * ld + c];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s += J[a * 6 + k] * p[k];
        T3[(size_t)a * ld + c] = s;
    }
}

// rows/columns pos..pos+2 <- T3 (columns outside the feature's own block; :441-451), 3x3 block <- T3[:,blk] J' (:439)
template <typename T>
__global__ void __launch_bounds__(256) k_convert_apply(T *P, int ld, int n, int pos, const double *J, const double *T3)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    if (c >= pos && c < pos + 6) {
        if (c < pos + 3) {
            const int a = c - pos;
            for (int b = a; b < 3; ++b) {
                double s = 0.0;
                for (int k = 0; k < 6; ++k) s += T3[(size_t)a * ld + pos + k] * J[b * 6 + k];
                P[(size_t)(pos + a) * ld + pos + b] = (T)s;
                P[(size_t)(pos + b) * ld + pos + a] = (T)s;
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const T v = (T)T3[(size_t)a * ld + c];
        P[(size_t)(pos + a) * ld + c] = v;
        P[(size_t)c * ld + pos + a] = v;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
void launch_add_features(EkfEngine *e, const double *d_uv, int count, double *d_Jpo, double *d_Jhr)
{
    hipStream_t s = e->stream;
    const int n_old = e->n, N_old = e->N;
    k_addfeat_prepare<<<(count + 63) / 64, 64, 0, s>>>(e->d.state, e->cam, e->cfg.par.initInvDepthRho, d_uv, count, N_old, n_old,
                                                       e->d.feat_pos, e->d.feat_type, e->d.feat_covpos, d_Jpo, d_Jhr);
    const double nx = e->cfg.cam.pixelErrorX * e->cfg.cam.pixelErrorX, ny = e->cfg.cam.pixelErrorY * e->cfg.cam.pixelErrorY;
    const double nr = e->cfg.par.inverseDepthRhoSD * e->cfg.par.inverseDepthRhoSD;
    dim3 g1((n_old + 255) / 256, count), g2(count, count);
    if (e->f32) {
        k_addfeat_rows<float><<<g1, 256, 0, s>>>((float *)e->d.P, e->ldP, n_old, d_Jpo);
        k_addfeat_corner<float><<<g2, 64, 0, s>>>((float *)e->d.P, e->ldP, n_old, count, d_Jpo, d_Jhr, nx, ny, nr);
    } else {
        k_addfeat_rows<double><<<g1, 256, 0, s>>>((double *)e->d.P, e->ldP, n_old, d_Jpo);
        k_addfeat_corner<double><<<g2, 64, 0, s>>>((double *)e->d.P, e->ldP, n_old, count, d_Jpo, d_Jhr, nx, ny, nr);
    }
}

void launch_compact_P(EkfEngine *e, int n_new, const int *d_new2old)
{
    if (n_new <= 0) return;
    dim3 g((n_new + 255) / 256, n_new);
    if (e->f32) k_compact_P<float><<<g, 256, 0, e->stream>>>((const float *)e->d.P, (float *)e->d.P2, e->ldP, n_new, d_new2old);
    else k_compact_P<double><<<g, 256, 0, e->stream>>>((const double *)e->d.P, (double *)e->d.P2, e->ldP, n_new, d_new2old);
    void *t = e->d.P;
    e->d.P = e->d.P2;
    e->d.P2 = t;
}

void launch_linearity(EkfEngine *e, double *d_out)
{
    if (e->N <= 0) return;
    const int nb = (e->N + 255) / 256;
    if (e->f32)
        k_linearity<float><<<nb, 256, 0, e->stream>>>(e->d.state, e->d.feat_pos, e->d.feat_type, e->d.feat_covpos, e->N,
                                                     (const float *)e->d.P, e->ldP, d_out);
    else
        k_linearity<double><<<nb, 256, 0, e->stream>>>(e->d.state, e->d.feat_pos, e->d.feat_type, e->d.feat_covpos, e->N,
                                                      (const double *)e->d.P, e->ldP, d_out);
}

void launch_convert(EkfEngine *e, int fi, int pos, double *d_J, double *d_T3)
{
    hipStream_t s = e->stream;
    k_convert_prepare<<<1, 64, 0, s>>>(e->d.feat_pos, e->d.feat_type, fi, d_J);
    const int nb = (e->n + 255) / 256;
    if (e->f32) {
        k_convert_rows<float><<<nb, 256, 0, s>>>((const float *)e->d.P, e->ldP, e->n, pos, d_J, d_T3);
        k_convert_apply<float><<<nb, 256, 0, s>>>((float *)e->d.P, e->ldP, e->n, pos, d_J, d_T3);
    } else {
        k_convert_rows<double><<<nb, 256, 0, s>>>((const double *)e->d.P, e->ldP, e->n, pos, d_J, d_T3);
        k_convert_apply<double><<<nb, 256, 0, s>>>((double *)e->d.P, e->ldP, e->n, pos, d_J, d_T3);
    }
}

// Measurement aid (ekf_round_covariance_to_f32): every stored entry of an fp64 covariance replaced by its nearest fp32 value.
__global__ void __launch_bounds__(256) k_round_P_f32(double *P, int ld, int n, int rows)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (i < rows && j < n) P[(size_t)i * ld + j] = (double)(float)P[(size_t)i * ld + j];
}

void launch_round_P_f32(EkfEngine *e)
{
    const int rows = e->shard_world > 1 ? local_row(e->rm, e->rm.r1 - 1) + 1 : e->n;
    k_round_P_f32<<<dim3((e->n + 255) / 256, rows), 256, 0, e->stream>>>((double *)e->d.P, e->ldP, e->n, rows);
}

} // namespace ekf
