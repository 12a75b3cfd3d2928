// kernels_update.hip -- A7, the Kalman update (EKF/Update.cpp:92-319), in the algebraically equivalent
// "square-root downdate" form that never materialises the reference's dense n x n temporaries:
//
//     A  = H P            (m x n; rows gathered from the cached H_f P pairs, HBM-bound copy)
//     S  = A H' + R       (m x m fp64, lower triangle; block-sparse H => 13-term dot products)
//     S  = L L'           (blocked right-looking Cholesky, fp64)
//     B  = inv(L) A,  z = inv(L) nu      (fused into the same panel sweep)
//     dx = B' z           (= K nu,  K = P H' inv(S))
//     P <- sym(P) - B' B  (= 0.5 ((I-KH)P + ((I-KH)P)'), kernels_pupdate.hip -- the MFMA kernel)
//     q normalisation and its Jacobian on the rows/columns 3..6 of P (Update.cpp:45-85, 303-317)
//
// The reference forms K = P H' inv(S) with an LU inverse and multiplies the dense (I - K H) by P (2 n^3 flops);
// both give the same x and P up to rounding (S is symmetric positive definite: R = pixelErrorX * I).
#include "engine.h"

namespace ekf {

// ---------------------------------------------------------------------------------------------------- gather
// Per selected match i: A[2i..2i+1, :] = HP[2f..2f+1, :], its Jacobian blocks, position/dimension and the
// dead-banded innovation nu (Update.cpp:125-135).  Rows m..m_pad of A are zero-filled for the k-tiled kernels.
template <typename T>
__global__ void __launch_bounds__(256)
k_gather(const EkfMatch *matches, int M, int m_pad, const T *HP, T *A, int ld, int n_pad, const double *uv_tab,
         const double *Hs_tab, const double *Hf_tab, const int *feat_type, const int *feat_covpos, double *nu,
         double *mHs, double *mHf, int *mpos, int *mdim)
{
    const int row = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int m = 2 * M;
    if (row < m) {
        const int i = row >> 1, r = row & 1;
        const int fi = matches[i].featureIndex;
        if (j < n_pad) A[(size_t)row * ld + j] = HP[(size_t)(2 * fi + r) * ld + j];
        if (blockIdx.x == 0 && r == 0) {
            const int t = threadIdx.x;
            if (t < 14) mHs[14 * i + t] = Hs_tab[14 * fi + t];
            else if (t < 26) mHf[12 * i + t - 14] = Hf_tab[12 * fi + t - 14];
            else if (t == 26) mpos[i] = feat_covpos[fi];
            else if (t == 27) mdim[i] = feat_dim(feat_type[fi]);
            else if (t == 28 || t == 29) {
                const int c = t - 28;
                const double a = matches[i].imagePos[c] - uv_tab[2 * fi + c];
                nu[2 * i + c] = fabs(a) > EKF_DELTA ? a : 0.0;
            }
        }
    } else if (row < m_pad) {
        if (j < n_pad) A[(size_t)row * ld + j] = (T)0;
        if (blockIdx.x == 0 && threadIdx.x == 0) nu[row] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------------ S = A H' + R
// One thread per 2x2 block (a, b), b <= a, of the lower triangle.
template <typename T>
__global__ void __launch_bounds__(256)
k_assemble_S(const T *A, int ld, int M, const double *mHs, const double *mHf, const int *mpos, const int *mdim,
             double pixel_err, double *S, int ldS)
{
    const int b = blockIdx.x * 16 + (threadIdx.x & 15);
    const int a = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (a >= M || b > a) return;
    const int pos = mpos[b], d = mdim[b];
    const double *hs = mHs + 14 * b, *hf = mHf + 12 * b;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const T *ar = A + (size_t)(2 * a + r) * ld;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double v = (double)ar[k];
            s0 += v * hs[k];
            s1 += v * hs[7 + k];
        }
        for (int k = 0; k < d; ++k) {
            const double v = (double)ar[pos + k];
            s0 += v * hf[k];
            s1 += v * hf[6 + k];
        }
        if (a == b) {
            if (r == 0) s0 += pixel_err;
            else s1 += pixel_err;
        }
        S[(size_t)(2 * a + r) * ldS + 2 * b] = s0;
        S[(size_t)(2 * a + r) * ldS + 2 * b + 1] = s1;
    }
}

// -------------------------------------------------------------------------------------- blocked Cholesky sweep
// Step k of the right-looking sweep over [ S | nu | A ], panel width NB = 32, rows k0..k0+kb-1:
//   diag   : L_kk = chol(S_kk), Linv = inv(L_kk)                                     (one workgroup, LDS)
//   panel  : L_ik = S_ik Linv' (i > k),  B_k = Linv A_k,  z_k = Linv nu_k
//   trail  : S_ij -= L_ik L_jk' (i >= j > k),  A_i -= L_ik B_k,  nu_i -= L_ik z_k  (i > k)
__global__ void __launch_bounds__(64) k_chol_diag(double *S, int ldS, int k0, int kb, double *Linv, int *counts)
{
    __shared__ double a[NB][NB + 1];
    __shared__ double li[NB][NB + 1];
    const int t = threadIdx.x;
    for (int i = t; i < NB * NB; i += 64) {
        const int r = i / NB, c = i % NB;
        a[r][c] = (r < kb && c <= r) ? S[(size_t)(k0 + r) * ldS + k0 + c] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int j = 0; j < kb; ++j) {
        const double djj = a[j][j];
        if (!(djj > 0.0)) {
            if (t == 0) counts[CNT_ERR] = EKF_ERR_NOT_POSITIVE_DEFINITE;
            return;
        }
        const double dj = sqrt(djj);
        __syncthreads();
        if (t == j) a[j][j] = dj;
        if (t > j && t < kb) a[t][j] = a[t][j] / dj;
        __syncthreads();
        // trailing update of the lower triangle: element (r, c), j < c <= r < kb
        for (int idx = t; idx < NB * NB; idx += 64) {
            const int r = idx / NB, c = idx % NB;
            if (c > j && c <= r && r < kb) a[r][c] -= a[r][j] * a[c][j];
        }
        __syncthreads();
    }
    // inverse of the lower-triangular factor: column c by forward substitution (lane c)
    if (t < NB) {
        const int c = t;
        for (int r = 0; r < NB; ++r) {
            double s = (r == c) ? 1.0 : 0.0;
            for (int k2 = c; k2 < r; ++k2) s -= a[r][k2] * li[k2][c];
            li[r][c] = (r >= c) ? s / a[r][r] : 0.0;
        }
    }
    __syncthreads();
    for (int i = t; i < NB * NB; i += 64) {
        const int r = i / NB, c = i % NB;
        if (r < kb && c <= r) S[(size_t)(k0 + r) * ldS + k0 + c] = a[r][c];
        Linv[i] = li[r][c];
    }
}

// grid.x = row blocks of S below the panel + column chunks of A (256 wide) + 1 block for nu
template <typename T>
__global__ void __launch_bounds__(256)
k_chol_panel(double *S, int ldS, int m, int k0, int kb, const double *Linv, T *A, int ld, int n_pad, double *nu,
             int n_sblocks)
{
    __shared__ double sL[NB][NB + 1];
    __shared__ double sS[NB][NB + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i < NB * NB; i += 256) sL[i / NB][i % NB] = Linv[i];
    const int k1 = k0 + kb;
    if ((int)blockIdx.x < n_sblocks) {
        // L_ik = S_ik Linv'  : 32 x kb tile, 4 outputs per thread
        const int i0 = k1 + blockIdx.x * NB;
        for (int i = tid; i < NB * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            sS[r][c] = (i0 + r < m && c < kb) ? S[(size_t)(i0 + r) * ldS + k0 + c] : 0.0;
        }
        __syncthreads();
        for (int i = tid; i < NB * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            if (i0 + r < m && c < kb) {
                double s = 0.0;
                for (int k2 = 0; k2 <= c; ++k2) s += sS[r][k2] * sL[c][k2];
                S[(size_t)(i0 + r) * ldS + k0 + c] = s;
            }
        }
    } else if ((int)blockIdx.x == n_sblocks) {
        // z_k = Linv nu_k
        __syncthreads();
        __shared__ double sn[NB];
        if (tid < NB) sn[tid] = tid < kb ? nu[k0 + tid] : 0.0;
        __syncthreads();
        if (tid < kb) {
            double s = 0.0;
            for (int c = 0; c <= tid; ++c) s += sL[tid][c] * sn[c];
            nu[k0 + tid] = s;
        }
    } else {
        // B_k = Linv A_k : one column per thread, 32 rows in registers
        __syncthreads();
        const int j = (blockIdx.x - n_sblocks - 1) * 256 + tid;
        if (j >= n_pad) return;
        double v[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) v[r] = r < kb ? (double)A[(size_t)(k0 + r) * ld + j] : 0.0;
#pragma unroll
        for (int r = NB - 1; r >= 0; --r) {
            if (r < kb) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c <= r; ++c) s += sL[r][c] * v[c];
                A[(size_t)(k0 + r) * ld + j] = (T)s;
            }
        }
    }
}

// grid.x : [0, n_tiles) lower-triangular 32x32 tiles of the trailing S; then row-block x column-chunk tiles of A;
// then one block for nu.
template <typename T>
__global__ void __launch_bounds__(256)
k_chol_trailing(double *S, int ldS, int m, int k0, int kb, T *A, int ld, int n_pad, double *nu, int nrb,
                int n_stiles, int n_cchunks)
{
    __shared__ double sA[NB][NB + 1];
    __shared__ double sB[NB][NB + 1];
    const int tid = threadIdx.x;
    const int k1 = k0 + kb;
    int b = blockIdx.x;
    if (b < n_stiles) {
        // decode (ti >= tj) from the linear lower-triangle index
        int ti = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
        while (ti * (ti + 1) / 2 > b) --ti;
        const int tj = b - ti * (ti + 1) / 2;
        const int i0 = k1 + ti * NB, j0 = k1 + tj * NB;
        for (int i = tid; i < NB * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            sA[r][c] = (i0 + r < m && c < kb) ? S[(size_t)(i0 + r) * ldS + k0 + c] : 0.0;
            sB[r][c] = (j0 + r < m && c < kb) ? S[(size_t)(j0 + r) * ldS + k0 + c] : 0.0;
        }
        __syncthreads();
        for (int i = tid; i < NB * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            if (i0 + r < m && j0 + c < m && j0 + c <= i0 + r) {
                double s = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < NB; ++k2) s += sA[r][k2] * sB[c][k2];
                S[(size_t)(i0 + r) * ldS + j0 + c] -= s;
            }
        }
        return;
    }
    b -= n_stiles;
    if (b < nrb * n_cchunks) {
        const int rb = b / n_cchunks, cc = b % n_cchunks;
        const int i0 = k1 + rb * NB;
        for (int i = tid; i < NB * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            sA[r][c] = (i0 + r < m && c < kb) ? S[(size_t)(i0 + r) * ldS + k0 + c] : 0.0;
        }
        __syncthreads();
        const int j = cc * 256 + tid;
        if (j >= n_pad) return;
        double bk[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) bk[c] = c < kb ? (double)A[(size_t)(k0 + c) * ld + j] : 0.0;
        for (int r = 0; r < NB; ++r) {
            if (i0 + r >= m) break;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) s += sA[r][c] * bk[c];
            A[(size_t)(i0 + r) * ld + j] = (T)((double)A[(size_t)(i0 + r) * ld + j] - s);
        }
        return;
    }
    // nu_i -= L_ik z_k
    for (int i = k1 + tid; i < m; i += 256) {
        double s = 0.0;
        for (int c = 0; c < kb; ++c) s += S[(size_t)i * ldS + k0 + c] * nu[k0 + c];
        nu[i] -= s;
    }
}

// ------------------------------------------------------------------------------------------------- dx = B' z
template <typename T>
__global__ void __launch_bounds__(256)
k_dx_partial(const T *B, int ld, int m, int n, const double *z, double *part, int ldpart)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int ks = blockIdx.y;
    if (j >= n) return;
    const int per = (m + DX_SPLIT - 1) / DX_SPLIT;
    const int kb = ks * per, ke = min(m, kb + per);
    double s = 0.0;
    for (int k = kb; k < ke; ++k) s += (double)B[(size_t)k * ld + j] * z[k];
    part[(size_t)ks * ldpart + j] = s;
}

// stateUpdate (Update.cpp:147-204): x += dx with the DELTA dead-band on every component; R(q) recomputed from
// the un-normalised q (:168).
__global__ void __launch_bounds__(256)
k_state_apply(double *st, double *feat_pos, const int *feat_type, const int *feat_covpos, int N, const double *part,
              int ldpart)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) {
        double *x = st + ST_X;
        for (int i = 0; i < 13; ++i) {
            double s = 0.0;
            for (int ks = 0; ks < DX_SPLIT; ++ks) s += part[(size_t)ks * ldpart + i];
            if (fabs(s) > EKF_DELTA) x[i] += s;
        }
        quat_to_rot(x + 3, st + ST_R);
    }
    if (t >= N * 6) return;
    const int f = t / 6, a = t % 6;
    if (a >= feat_dim(feat_type[f])) return;
    const int j = feat_covpos[f] + a;
    double s = 0.0;
    for (int ks = 0; ks < DX_SPLIT; ++ks) s += part[(size_t)ks * ldpart + j];
    if (fabs(s) > EKF_DELTA) feat_pos[6 * f + a] += s;
}

// ------------------------------------------------------------------------------ quaternion normalisation tail
// normalizeQuaternionJacobian from the un-normalised q (Update.cpp:45-60, 305), then q /= |q| (:308-313).
__global__ void k_quat_norm(double *st)
{
    if (threadIdx.x != 0) return;
    double *q = st + ST_X + 3;
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    const double nrm = sqrt(r * r + x * x + y * y + z * z);
    const double a = 1.0 / (nrm * nrm * nrm);
    double *J = st + ST_JN;
    const double M[16] = {x * x + y * y + z * z, -r * x, -r * y, -r * z,
                          -x * r, r * r + y * y + z * z, -x * y, -x * z,
                          -y * r, -y * x, r * r + x * x + z * z, -y * z,
                          -z * r, -z * x, -z * y, r * r + x * x + y * y};
    for (int i = 0; i < 16; ++i) J[i] = M[i] * a;
    q[0] = r / nrm; q[1] = x / nrm; q[2] = y / nrm; q[3] = z / nrm;
    quat_to_rot(q, st + ST_R);
}

// normalizeCovariance (Update.cpp:64-85): P <- D P D', D = diag(I3, J, I).  Five disjoint blocks; block 0 owns
// the 7x7 corner pieces, every other thread owns column j of the row strip 3..6 and row j of the column strip.
template <typename T>
__global__ void __launch_bounds__(256) k_normalize_cov(T *P, int ld, int n, const double *st)
{
    __shared__ double J[16];
    __shared__ double C[7][7];
    const int tid = threadIdx.x;
    if (tid < 16) J[tid] = st[ST_JN + tid];
    if (blockIdx.x == 0) {
        if (tid < 49) C[tid / 7][tid % 7] = (double)P[(size_t)(tid / 7) * ld + tid % 7];
        __syncthreads();
        if (tid < 12) { // P[0:3,3:7] J'
            const int i = tid / 4, a = tid % 4;
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += C[i][3 + k] * J[a * 4 + k];
            P[(size_t)i * ld + 3 + a] = (T)s;
        } else if (tid < 24) { // J P[3:7,0:3]
            const int t = tid - 12, a = t / 3, j = t % 3;
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += J[a * 4 + k] * C[3 + k][j];
            P[(size_t)(3 + a) * ld + j] = (T)s;
        } else if (tid < 40) { // J P[3:7,3:7] J' : upper triangle, mirrored (keeps P bitwise symmetric)
            const int t = tid - 24, a = t / 4, b = t % 4;
            if (a <= b) {
                double s = 0.0;
                for (int l = 0; l < 4; ++l) {
                    double u = 0.0;
                    for (int k = 0; k < 4; ++k) u += J[a * 4 + k] * C[3 + k][3 + l];
                    s += u * J[b * 4 + l];
                }
                P[(size_t)(3 + a) * ld + 3 + b] = (T)s;
                P[(size_t)(3 + b) * ld + 3 + a] = (T)s;
            }
        }
        return;
    }
    __syncthreads();
    const int j = 7 + (blockIdx.x - 1) * 256 + tid;
    if (j >= n) return;
    double col[4], row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        col[k] = (double)P[(size_t)(3 + k) * ld + j];
        row[k] = (double)P[(size_t)j * ld + 3 + k];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double s = 0.0, t = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s += J[a * 4 + k] * col[k];
            t += row[k] * J[a * 4 + k];
        }
        P[(size_t)(3 + a) * ld + j] = (T)s;
        P[(size_t)j * ld + 3 + a] = (T)t;
    }
}

// P-update launcher lives in kernels_pupdate.hip
void launch_p_update(EkfEngine *e, int m_pad);

template <typename T>
static void update_impl(EkfEngine *e, int M, bool update_cov)
{
    hipStream_t s = e->stream;
    const int m = 2 * M, n = e->n, ld = e->ldP, ldS = e->ldS;
    const int m_pad = round_up(m, NB);
    const int n_pad = round_up(n, LD_ALIGN);
    T *A = (T *)e->d.A;
    {
        dim3 grid((n_pad + 255) / 256, m_pad);
        k_gather<T><<<grid, 256, 0, s>>>(e->d.matches, M, m_pad, (const T *)e->d.HP, A, ld, n_pad, e->d.pred_uv,
                                         e->d.Hs, e->d.Hf, e->d.feat_type, e->d.feat_covpos, e->d.nu, e->d.mHs,
                                         e->d.mHf, e->d.mpos, e->d.mdim);
    }
    {
        dim3 grid((M + 15) / 16, (M + 15) / 16);
        k_assemble_S<T><<<grid, 256, 0, s>>>(A, ld, M, e->d.mHs, e->d.mHf, e->d.mpos, e->d.mdim,
                                             e->cfg.cam.pixelErrorX, e->d.S, ldS);
    }
    const int n_cchunks = (n_pad + 255) / 256;
    for (int k0 = 0; k0 < m; k0 += NB) {
        const int kb = min(NB, m - k0);
        const int k1 = k0 + kb;
        k_chol_diag<<<1, 64, 0, s>>>(e->d.S, ldS, k0, kb, e->d.Linv, e->d.counts);
        const int nrb = (m - k1 + NB - 1) / NB; // row blocks below the panel
        k_chol_panel<T><<<nrb + 1 + n_cchunks, 256, 0, s>>>(e->d.S, ldS, m, k0, kb, e->d.Linv, A, ld, n_pad, e->d.nu,
                                                           nrb);
        if (nrb > 0) {
            const int n_stiles = nrb * (nrb + 1) / 2;
            k_chol_trailing<T><<<n_stiles + nrb * n_cchunks + 1, 256, 0, s>>>(e->d.S, ldS, m, k0, kb, A, ld, n_pad,
                                                                              e->d.nu, nrb, n_stiles, n_cchunks);
        }
    }
    {
        dim3 grid((n + 255) / 256, DX_SPLIT);
        k_dx_partial<T><<<grid, 256, 0, s>>>(A, ld, m, n, e->d.nu, e->d.dx_part, ld);
        const int nt = max(e->N * 6, 1);
        k_state_apply<<<(nt + 255) / 256, 256, 0, s>>>(e->d.state, e->d.feat_pos, e->d.feat_type, e->d.feat_covpos,
                                                       e->N, e->d.dx_part, ld);
    }
    if (!update_cov) return;
    launch_p_update(e, m_pad);
    k_quat_norm<<<1, 64, 0, s>>>(e->d.state);
    const int nb = 1 + (n > 7 ? (n - 7 + 255) / 256 : 0);
    k_normalize_cov<T><<<nb, 256, 0, s>>>((T *)e->d.P, ld, n, e->d.state);
}

void launch_update(EkfEngine *e, int M, bool update_cov)
{
    if (M <= 0) return;
    if (e->f32) update_impl<float>(e, M, update_cov);
    else update_impl<double>(e, M, update_cov);
}

} // namespace ekf
