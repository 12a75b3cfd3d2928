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
#include "chol32.h"

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
// Right-looking sweep over [ S | nu ], panel width NB = 32.  The only serial piece is the 32x32 diagonal block:
// its Cholesky factor and the inverse of that factor.  It is computed by ONE workgroup with all 256 threads working
// in LDS (block_chol_inv32_bp in chol32.h: 4x4 block pivots, one barrier per block column) and -- look-ahead -- inside the trailing-update launch of the
// PREVIOUS panel, by the workgroup that owns tile (k+1, k+1): while the other workgroups of that launch update
// their tiles, this one finishes tile (k+1, k+1), factorises it and stores inv(L_{k+1,k+1}) into Dinv.  Per panel:
//   panel : L_ik = S_ik inv(L_kk)' for the row blocks below (inverse read from Dinv), z_k = inv(L_kk) nu_k
//   trail : S_ij -= L_ik L_jk' (i >= j > k), nu_i -= L_ik z_k, + the look-ahead factorisation of block k+1
// B = inv(L) A is NOT part of the sweep: it is independent per column of A and runs afterwards as one launch
// (k_trsm) on the MFMA pipe.

// its 256-block's row
__device__ __forceinline__ void store_linv(double *Dinv, int k0, const double (*x)[NB + 1])
{
    const int cb = k0 % TB;
    for (int i = threadIdx.x; i < NB * NB; i += 256) {
        const int r = i / NB, c = i % NB;
        Dinv[(size_t)(k0 + r) * TB + cb + c] = x[r][c];
    }
}

// first panel only: factorise block 0
__global__ void __launch_bounds__(256) k_chol_diag0(const double *S, int ldS, int kb, double *Dinv, int *counts)
{
    __shared__ double sa[NB][NB + 1], sx[NB][NB + 1];
    for (int i = threadIdx.x; i < NB * NB; i += 256) {
        const int r = i / NB, c = i % NB;
        sa[r][c] = (r < kb && c <= r) ? S[(size_t)r * ldS + c] : ((r == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (!block_chol_inv32_bp(sa, sx) && threadIdx.x == 0) counts[CNT_ERR] = EKF_ERR_NOT_POSITIVE_DEFINITE;
    store_linv(Dinv, 0, sx);
}

// grid.x = nrb row blocks below the panel + 1 (z_k)
__global__ void __launch_bounds__(256)
k_chol_panel(double *S, int ldS, int m, int k0, int kb, const double *Dinv, double *nu, int nrb)
{
    __shared__ double sLi[NB][NB + 1]; // inv(L_kk)
    __shared__ double sS[NB][NB + 1];
    const int tid = threadIdx.x;
    {
        const int cb = k0 % TB;
        for (int i = tid; i < NB * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            sLi[r][c] = Dinv[(size_t)(k0 + r) * TB + cb + c];
        }
    }
    const int k1 = k0 + kb;
    if ((int)blockIdx.x < nrb) {
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
                for (int k2 = 0; k2 <= c; ++k2) s += sS[r][k2] * sLi[c][k2];
                S[(size_t)(i0 + r) * ldS + k0 + c] = s;
            }
        }
    } else {
        __shared__ double sn[NB];
        if (tid < NB) sn[tid] = tid < kb ? nu[k0 + tid] : 0.0;
        __syncthreads();
        if (tid < kb) {
            double s = 0.0;
            for (int c = 0; c <= tid; ++c) s += sLi[tid][c] * sn[c];
            nu[k0 + tid] = s;
        }
    }
}

// grid.x : [0, n_stiles) lower-triangular 32x32 tiles of the trailing S (tile 0 = block (k+1, k+1): look-ahead
// factorisation), then one block for nu.
__global__ void __launch_bounds__(256)
k_chol_trailing(double *S, int ldS, int m, int k0, int kb, double *nu, int n_stiles, double *Dinv, int *counts)
{
    __shared__ double sA[NB][NB + 1];
    __shared__ double sB[NB][NB + 1];
    const int tid = threadIdx.x;
    const int k1 = k0 + kb;
    const int b = blockIdx.x;
    if (b < n_stiles) {
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
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256;
            const int r = i / NB, c = i % NB;
            v[q] = 0.0;
            if (i0 + r < m && j0 + c < m && j0 + c <= i0 + r) {
                double s = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < NB; ++k2) s += sA[r][k2] * sB[c][k2];
                v[q] = S[(size_t)(i0 + r) * ldS + j0 + c] - s;
                if (b != 0) S[(size_t)(i0 + r) * ldS + j0 + c] = v[q];
            }
        }
        if (b != 0) return;
        // look-ahead: this workgroup owns block (k+1, k+1); finish it in LDS, factorise, publish its inverse
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256;
            const int r = i / NB, c = i % NB;
            const bool live = (k1 + r < m) && (c <= r);
            sA[r][c] = live ? v[q] : ((r == c) ? 1.0 : 0.0);
        }
        __syncthreads();
        if (!block_chol_inv32_bp(sA, sB) && tid == 0) counts[CNT_ERR] = EKF_ERR_NOT_POSITIVE_DEFINITE;
        store_linv(Dinv, k1, sB);
        return;
    }
    for (int i = k1 + tid; i < m; i += 256) {
        double s = 0.0;
        for (int c = 0; c < kb; ++c) s += S[(size_t)i * ldS + k0 + c] * nu[k0 + c];
        nu[i] -= s;
    }
}

// ------------------------------------------------------------------------------ inverse of the 256-blocks of L
// Doubling step s -> 2s inside each TB = 256 diagonal block of L:  inv([L11 0; L21 L22]) has the off-diagonal
// block X21 = -X22 L21 X11.  Two batched small products per level (T = L21 X11, X21 = -X22 T), 32x32 output tile
// per workgroup.  O(m s^2) flops per level: negligible.
// mode 0: T[pair] = L21 * X11 ;  mode 1: X21 = -X22 * T
__global__ void __launch_bounds__(256)
k_triinv_level(const double *S, int ldS, int m, double *Dinv, double *Tbuf, int s, int mode)
{
    __shared__ double sA[NB][NB + 1];
    __shared__ double sB[NB][NB + 1];
    const int tiles = s / NB;
    const int pair = blockIdx.x / (tiles * tiles);
    const int t = blockIdx.x % (tiles * tiles);
    const int tr = t / tiles, tc = t % tiles;
    const int r0 = pair * 2 * s; // first row of the pair's 2s x 2s diagonal block
    if (r0 + s >= m) return;     // no second half
    const int cb = r0 % TB;      // column offset of the pair inside its 256-block
    const int tid = threadIdx.x;
    double acc[4] = {0, 0, 0, 0};
    for (int kk = 0; kk < s; kk += NB) {
        for (int i = tid; i < NB * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            double av, bv;
            if (mode == 0) {
                const int gr = r0 + s + tr * NB + r, gc = r0 + kk + c; // L21
                av = (gr < m) ? S[(size_t)gr * ldS + gc] : 0.0;
                bv = Dinv[(size_t)(r0 + kk + r) * TB + cb + tc * NB + c]; // X11[kk + r][tc*32 + c]
            } else {
                av = Dinv[(size_t)(r0 + s + tr * NB + r) * TB + cb + s + kk + c]; // X22[tr*32 + r][kk + c]
                bv = Tbuf[(size_t)(r0 + s + kk + r) * (TB / 2) + tc * NB + c];   // T[kk + r][tc*32 + c]
            }
            sA[r][c] = av;
            sB[r][c] = bv;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256;
            const int r = i / NB, c = i % NB;
            double sacc = 0.0;
#pragma unroll
            for (int k2 = 0; k2 < NB; ++k2) sacc += sA[r][k2] * sB[k2][c];
            acc[q] += sacc;
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + q * 256;
        const int r = i / NB, c = i % NB;
        if (mode == 0) Tbuf[(size_t)(r0 + s + tr * NB + r) * (TB / 2) + tc * NB + c] = acc[q];
        else Dinv[(size_t)(r0 + s + tr * NB + r) * TB + cb + tc * NB + c] = -acc[q];
    }
}

// ------------------------------------------------------------------------------------------ B = inv(L) A
// Forward substitution in TB = 256 row blocks, independent per column of A: one workgroup owns a CT-column strip
// (CT = MFMA block width) and walks the row blocks K = 0, 1, ...:
//     C   = A_K - L[K, 0:K] B[0:K]      (GEMM, k-depth 256 K, MFMA; L read as fp64 and converted)
//     B_K = inv(L_KK) C                  (GEMM, k-depth 256, triangular)
// No inter-workgroup dependency, so the whole solve is ONE launch.  Each wavefront owns 64 of the 256 rows.
template <typename T>
struct MmaU;
template <>
struct MmaU<float> {
    static constexpr int MB = 32, NACC = 16;
    typedef float acc_t __attribute__((ext_vector_type(16)));
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
    __device__ static __forceinline__ int col(int lane) { return lane & 31; }
};
template <>
struct MmaU<double> {
    static constexpr int MB = 16, NACC = 4;
    typedef double acc_t __attribute__((ext_vector_type(4)));
    __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int reg, int lane) { return (lane >> 4) + 4 * reg; }
    __device__ static __forceinline__ int col(int lane) { return lane & 15; }
};

constexpr int TR_KS = 16; // k-slab

template <typename T>
__global__ void __launch_bounds__(256)
k_trsm(const double *L, int ldS, int m, int m_pad, const double *Dinv, T *A, int ld)
{
    using M = MmaU<T>;
    constexpr int MB = M::MB, CT = MB, KI = 64 / MB, RB = 64 / MB; // RB row blocks per wavefront
    __shared__ T sL[TR_KS][TB];
    __shared__ T sB[TR_KS][CT];
    __shared__ T sC[TB][CT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int klane = lane / MB, idx = lane % MB;
    const int col0 = blockIdx.x * CT;
    const int nK = (m_pad + TB - 1) / TB;
    for (int K = 0; K < nK; ++K) {
        const int rows0 = K * TB;
        typename M::acc_t acc[RB];
#pragma unroll
        for (int x = 0; x < RB; ++x)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) acc[x][r] = (T)0;
        // phase 1: acc = L[rows0 + r, 0:rows0] * B[0:rows0, strip]
        for (int kk = 0; kk < rows0; kk += TR_KS) {
            {
                const int gr = rows0 + tid;
                const double *src = L + (size_t)gr * ldS + kk;
#pragma unroll
                for (int k = 0; k < TR_KS; ++k) sL[k][tid] = (gr < m) ? (T)src[k] : (T)0;
            }
            for (int i = tid; i < TR_KS * CT; i += 256) {
                const int k = i / CT, c = i % CT;
                sB[k][c] = A[(size_t)(kk + k) * ld + col0 + c];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < TR_KS; k += KI) {
                const T b = sB[k + klane][idx];
#pragma unroll
                for (int x = 0; x < RB; ++x) acc[x] = M::mma(sL[k + klane][wv * 64 + x * MB + idx], b, acc[x]);
            }
            __syncthreads();
        }
        // C = A_K - acc
#pragma unroll
        for (int x = 0; x < RB; ++x)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) {
                const int lr = wv * 64 + x * MB + M::row(r, lane), lc = M::col(lane);
                const int gr = rows0 + lr;
                const T a = (gr < m_pad) ? A[(size_t)gr * ld + col0 + lc] : (T)0;
                sC[lr][lc] = a - acc[x][r];
            }
        __syncthreads();
        // phase 2: B_K = inv(L_KK) C  (lower triangular: this wavefront's rows need k < its last row)
#pragma unroll
        for (int x = 0; x < RB; ++x)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) acc[x][r] = (T)0;
        for (int kk = 0; kk < TB; kk += TR_KS) {
            {
                const double *src = Dinv + (size_t)(rows0 + tid) * TB + kk;
                const bool live = rows0 + tid < m_pad;
#pragma unroll
                for (int k = 0; k < TR_KS; ++k) sL[k][tid] = live ? (T)src[k] : (T)0;
            }
            __syncthreads();
            if (kk < wv * 64 + 64) {
#pragma unroll
                for (int k = 0; k < TR_KS; k += KI) {
                    const T b = sC[kk + k + klane][idx];
#pragma unroll
                    for (int x = 0; x < RB; ++x) acc[x] = M::mma(sL[k + klane][wv * 64 + x * MB + idx], b, acc[x]);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int x = 0; x < RB; ++x)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) {
                const int gr = rows0 + wv * 64 + x * MB + M::row(r, lane);
                if (gr < m_pad) A[(size_t)gr * ld + col0 + M::col(lane)] = acc[x][r];
            }
        __syncthreads();
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
__global__ void __launch_bounds__(256) k_normalize_cov(T *P, int ld, int n, const double *st, RowMap rm)
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
    const bool mine = owns_row(rm, j); // sharded storage: the column strip exists only for owned rows
    T *prow = P + (size_t)local_row(rm, j) * ld;
    double col[4], row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        col[k] = (double)P[(size_t)(3 + k) * ld + j];
        row[k] = mine ? (double)prow[3 + k] : 0.0;
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
        if (mine) prow[3 + a] = (T)t;
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
    k_chol_diag0<<<1, 256, 0, s>>>(e->d.S, ldS, min(NB, m), e->d.Dinv, e->d.counts);
    for (int k0 = 0; k0 < m; k0 += NB) {
        const int kb = min(NB, m - k0);
        const int k1 = k0 + kb;
        const int nrb = (m - k1 + NB - 1) / NB; // row blocks below the panel
        k_chol_panel<<<nrb + 1, 256, 0, s>>>(e->d.S, ldS, m, k0, kb, e->d.Dinv, e->d.nu, nrb);
        if (nrb > 0) {
            const int n_stiles = nrb * (nrb + 1) / 2;
            k_chol_trailing<<<n_stiles + 1, 256, 0, s>>>(e->d.S, ldS, m, k0, kb, e->d.nu, n_stiles, e->d.Dinv,
                                                         e->d.counts);
        }
    }
    for (int sz = NB; sz < TB; sz *= 2) {
        const int npairs = (m_pad + 2 * sz - 1) / (2 * sz);
        const int tiles = (sz / NB) * (sz / NB);
        k_triinv_level<<<npairs * tiles, 256, 0, s>>>(e->d.S, ldS, m, e->d.Dinv, e->d.Tbuf, sz, 0);
        k_triinv_level<<<npairs * tiles, 256, 0, s>>>(e->d.S, ldS, m, e->d.Dinv, e->d.Tbuf, sz, 1);
    }
    {
        const int CT = sizeof(T) == 4 ? 32 : 16;
        k_trsm<T><<<n_pad / CT, 256, 0, s>>>(e->d.S, ldS, m, m_pad, e->d.Dinv, A, ld);
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
    k_normalize_cov<T><<<nb, 256, 0, s>>>((T *)e->d.P, ld, n, e->d.state, e->rm);
}

void launch_update(EkfEngine *e, int M, bool update_cov)
{
    if (M <= 0) return;
    if (e->f32) update_impl<float>(e, M, update_cov);
    else update_impl<double>(e, M, update_cov);
}

} // namespace ekf
