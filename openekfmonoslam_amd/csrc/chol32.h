// chol32.h -- the serial kernel of the blocked Cholesky sweep: factor a 32x32 SPD block and invert the factor, one
// workgroup of 256 threads.  Two formulations of the same elimination on the augmented block [A | I]:
//   block_chol_inv32     scalar pivots, one barrier per column (32 barriers, a division on every step's path)
//   block_chol_inv32_bp  4x4 block pivots, one barrier per block column (8 barriers); the 4x4 pivot block is
//                        factorised (LDL') redundantly by every thread, the within-block triangular solves are
//                        deferred to the end.  Same pivots in the same order, so the results agree to rounding.
// Thread (r, g) keeps columns 4g..4g+3 of row r of both halves in REGISTERS for the whole sweep; only what a step
// broadcasts travels through LDS (double-buffered).
//   a : 32x33 doubles in LDS, lower triangle of the SPD block (identity-padded rows beyond the live size)
//   x : 32x33 doubles in LDS, receives inv(L) (zeros above the diagonal)
// Both return false (uniformly) on a non-positive pivot.
#pragma once
#include <hip/hip_runtime.h>

namespace ekf {

constexpr int CH_NB = 32;

__device__ __forceinline__ bool block_chol_inv32(double (*a)[CH_NB + 1], double (*x)[CH_NB + 1], double *rs)
{
    __shared__ double colbuf[2][CH_NB], rowbuf[2][CH_NB];
    const int t = threadIdx.x;
    const int r = t >> 3, c0 = (t & 7) * 4;
    double av[4], xv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        av[e] = (c0 + e <= r) ? a[r][c0 + e] : 0.0;
        xv[e] = (r == c0 + e) ? 1.0 : 0.0;
    }
    if (c0 == 0) colbuf[0][r] = av[0];
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) rowbuf[0][c0 + e] = xv[e];
    }
    __syncthreads();
    bool ok = true;
    for (int j = 0; j < CH_NB; ++j) {
        const int p = j & 1;
        const double djj = colbuf[p][j];
        ok = ok && (djj > 0.0);
        const double inv = 1.0 / (djj > 0.0 ? djj : 1.0);
        if (t == 0) rs[j] = inv;
        if (r > j) {
            const double arj = colbuf[p][r] * inv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + e;
                if (c > j && c <= r) av[e] -= arj * colbuf[p][c];
                if (c <= j) xv[e] -= arj * rowbuf[p][c];
            }
        }
        const int jn = j + 1;
        if (jn < CH_NB) { // publish column j+1 of the A half and row j+1 of the identity half for the next step
            const int e = jn - c0;
            if (e >= 0 && e < 4 && r >= jn) colbuf[p ^ 1][r] = e == 0 ? av[0] : (e == 1 ? av[1] : (e == 2 ? av[2] : av[3]));
            if (r == jn) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rowbuf[p ^ 1][c0 + q] = xv[q];
            }
        }
        __syncthreads();
    }
    const double sr = sqrt(rs[r]);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[r][c0 + e] = (c0 + e <= r) ? xv[e] * sr : 0.0;
    __syncthreads();
    return ok;
}

// store inv(L_kk) (32x32, in LDS) into the block-diagonal inverse: Dinv is [m_pad256 x 256], row (k0 + r) holds

// 1/d to ~1 ulp: hardware estimate + two Newton steps (an IEEE division is ~3x longer and sits on the serial path)
__device__ __forceinline__ double fast_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

__device__ __forceinline__ bool block_chol_inv32_bp(double (*a)[CH_NB + 1], double (*x)[CH_NB + 1])
{
    __shared__ __attribute__((aligned(32))) double panel[2][CH_NB][4]; // panel[buf][row][k] = A[row][4J + k]
    __shared__ __attribute__((aligned(32))) double xrow[2][4][CH_NB];  // xrow[buf][k][c]    = X[4J + k][c]
    __shared__ double fac[8][10];                                      // per block: l10 l20 l30 l21 l31 l32 i0..i3
    const int t = threadIdx.x;
    const int r = t >> 3, g = t & 7, c0 = g * 4;
    double av0, av1, av2, av3, xv0, xv1, xv2, xv3;
    av0 = (c0 + 0 <= r) ? a[r][c0 + 0] : 0.0;
    av1 = (c0 + 1 <= r) ? a[r][c0 + 1] : 0.0;
    av2 = (c0 + 2 <= r) ? a[r][c0 + 2] : 0.0;
    av3 = (c0 + 3 <= r) ? a[r][c0 + 3] : 0.0;
    xv0 = (r == c0 + 0) ? 1.0 : 0.0;
    xv1 = (r == c0 + 1) ? 1.0 : 0.0;
    xv2 = (r == c0 + 2) ? 1.0 : 0.0;
    xv3 = (r == c0 + 3) ? 1.0 : 0.0;
    if (g == 0) { panel[0][r][0] = av0; panel[0][r][1] = av1; panel[0][r][2] = av2; panel[0][r][3] = av3; }
    if (r < 4) { xrow[0][r][c0] = xv0; xrow[0][r][c0 + 1] = xv1; xrow[0][r][c0 + 2] = xv2; xrow[0][r][c0 + 3] = xv3; }
    __syncthreads();
    bool ok = true;
#pragma unroll 1
    for (int J = 0; J < 8; ++J) {
        const int p = J & 1, j0 = 4 * J;
        const double(*pn)[4] = panel[p];
        // LDL' of the pivot block (every thread, same values)
        const double d00 = pn[j0][0];
        const double d10 = pn[j0 + 1][0], d11 = pn[j0 + 1][1];
        const double d20 = pn[j0 + 2][0], d21 = pn[j0 + 2][1], d22 = pn[j0 + 2][2];
        const double d30 = pn[j0 + 3][0], d31 = pn[j0 + 3][1], d32 = pn[j0 + 3][2], d33 = pn[j0 + 3][3];
        const double i0 = fast_rcp(d00);
        const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
        const double e11 = d11 - l10 * d10;
        const double e21 = d21 - l20 * d10, e31 = d31 - l30 * d10;
        const double i1 = fast_rcp(e11);
        const double l21 = e21 * i1, l31 = e31 * i1;
        const double e22 = d22 - l20 * d20 - l21 * e21;
        const double e32 = d32 - l30 * d20 - l31 * e21;
        const double i2 = fast_rcp(e22);
        const double l32 = e32 * i2;
        const double e33 = d33 - l30 * d30 - l31 * e31 - l32 * e32;
        const double i3 = fast_rcp(e33);
        ok = ok && d00 > 0.0 && e11 > 0.0 && e22 > 0.0 && e33 > 0.0;
        if (t == 0) {
            double *f = fac[J];
            f[0] = l10; f[1] = l20; f[2] = l30; f[3] = l21; f[4] = l31; f[5] = l32;
            f[6] = i0; f[7] = i1; f[8] = i2; f[9] = i3;
        }
        if (r >= j0 + 4) {
            // w D = a_r (row of the panel): forward, scale, back
            const double y0 = pn[r][0];
            const double y1 = pn[r][1] - l10 * y0;
            const double y2 = pn[r][2] - l20 * y0 - l21 * y1;
            const double y3 = pn[r][3] - l30 * y0 - l31 * y1 - l32 * y2;
            const double w3 = y3 * i3;
            const double w2 = y2 * i2 - l32 * w3;
            const double w1 = y1 * i1 - l21 * w2 - l31 * w3;
            const double w0 = y0 * i0 - l10 * w1 - l20 * w2 - l30 * w3;
            const double(*xr)[CH_NB] = xrow[p];
#define CH_UPD(E, AV, XV)                                                                                     \
    {                                                                                                         \
        const int c = c0 + E;                                                                                 \
        if (c >= j0 + 4 && c <= r) AV -= w0 * pn[c][0] + w1 * pn[c][1] + w2 * pn[c][2] + w3 * pn[c][3];         \
        XV -= w0 * xr[0][c] + w1 * xr[1][c] + w2 * xr[2][c] + w3 * xr[3][c];                                    \
    }
            CH_UPD(0, av0, xv0)
            CH_UPD(1, av1, xv1)
            CH_UPD(2, av2, xv2)
            CH_UPD(3, av3, xv3)
#undef CH_UPD
        }
        if (J < 7) { // publish block column J+1 of the A half and the pivot rows of the identity half
            const int q = p ^ 1;
            if (g == J + 1) { panel[q][r][0] = av0; panel[q][r][1] = av1; panel[q][r][2] = av2; panel[q][r][3] = av3; }
            if ((r >> 2) == J + 1) {
                xrow[q][r & 3][c0] = xv0; xrow[q][r & 3][c0 + 1] = xv1; xrow[q][r & 3][c0 + 2] = xv2; xrow[q][r & 3][c0 + 3] = xv3;
            }
        }
        __syncthreads();
    }
    // inv(L) = blockdiag(inv(chol(D_J))) X : unit-lower solve inside each 4-row group, then sqrt of the pivots
    x[r][c0] = xv0; x[r][c0 + 1] = xv1; x[r][c0 + 2] = xv2; x[r][c0 + 3] = xv3;
    __syncthreads();
    const int base = r & ~3, q = r & 3;
    const double *f = fac[r >> 2];
    double res[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        const double u0 = x[base][c];
        const double u1 = x[base + 1][c] - f[0] * u0;
        const double u2 = x[base + 2][c] - f[1] * u0 - f[3] * u1;
        const double u3 = x[base + 3][c] - f[2] * u0 - f[4] * u1 - f[5] * u2;
        res[e] = q == 0 ? u0 : (q == 1 ? u1 : (q == 2 ? u2 : u3));
    }
    const double sr = sqrt(f[6 + q]);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) x[r][c0 + e] = (c0 + e <= r) ? res[e] * sr : 0.0;
    __syncthreads();
    return ok;
}

// Same elimination with the rank-4 updates on the fp64 MFMA.  [A | X] (32 x 64) lives in accumulator layout for the
// whole sweep: wavefront w owns columns 16w..16w+15 (w = 0, 1: A; w = 2, 3: X), two 16x16 blocks (rows 0..15, 16..31),
// lane l holds rows (l >> 4) + 4 v, column l & 15.  Per 4x4 pivot step a lane publishes at most 9 values and reads
// 19 (pivot block, its two panel rows, its pivot-row element); the update M -= W [Ar | Xr] is ONE
// v_mfma_f64_16x16x4 per block (k = 4 = the pivot width).  A must be given as a full symmetric matrix (the pivot rows
// are read as rows).
__device__ __forceinline__ bool block_chol_inv32_mf(double (*a)[CH_NB + 1], double (*x)[CH_NB + 1])
{
    typedef double acc4 __attribute__((ext_vector_type(4)));
    __shared__ double pc[2][CH_NB][4];      // pc[buf][r][k] = A[r][4J + k]
    __shared__ double pr[2][4][2 * CH_NB];  // pr[buf][k][c] = [A | X][4J + k][c]
    __shared__ double fac[8][10];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int lc = lane & 15, lq = lane >> 4;
    acc4 m0, m1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r0 = lq + 4 * v, r1 = 16 + lq + 4 * v, c = 16 * w + lc;
        if (w < 2) {
            m0[v] = c <= r0 ? a[r0][c] : a[c][r0];
            m1[v] = c <= r1 ? a[r1][c] : a[c][r1];
        } else {
            m0[v] = (c - CH_NB == r0) ? 1.0 : 0.0;
            m1[v] = (c - CH_NB == r1) ? 1.0 : 0.0;
        }
    }
    bool ok = true;
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        const int p = J & 1, j0 = 4 * J;
        // publish the panel column (owner wavefront, 4 lanes columns) and the pivot rows (every wavefront, its columns)
        if (w == j0 / 16) {
            const int k = lc - (j0 % 16);
            if (k >= 0 && k < 4) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    pc[p][lq + 4 * v][k] = m0[v];
                    pc[p][16 + lq + 4 * v][k] = m1[v];
                }
            }
        }
        {
            const int v0 = (j0 % 16) / 4;
            const acc4 &src = (j0 < 16) ? m0 : m1;
            pr[p][lq][16 * w + lc] = src[v0];
        }
        __syncthreads();
        const double(*pn)[4] = pc[p];
        const double d00 = pn[j0][0];
        const double d10 = pn[j0 + 1][0], d11 = pn[j0 + 1][1];
        const double d20 = pn[j0 + 2][0], d21 = pn[j0 + 2][1], d22 = pn[j0 + 2][2];
        const double d30 = pn[j0 + 3][0], d31 = pn[j0 + 3][1], d32 = pn[j0 + 3][2], d33 = pn[j0 + 3][3];
        const double i0 = fast_rcp(d00);
        const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
        const double e11 = d11 - l10 * d10;
        const double e21 = d21 - l20 * d10, e31 = d31 - l30 * d10;
        const double i1 = fast_rcp(e11);
        const double l21 = e21 * i1, l31 = e31 * i1;
        const double e22 = d22 - l20 * d20 - l21 * e21;
        const double e32 = d32 - l30 * d20 - l31 * e21;
        const double i2 = fast_rcp(e22);
        const double l32 = e32 * i2;
        const double e33 = d33 - l30 * d30 - l31 * e31 - l32 * e32;
        const double i3 = fast_rcp(e33);
        ok = ok && d00 > 0.0 && e11 > 0.0 && e22 > 0.0 && e33 > 0.0;
        if (t == 0) {
            double *f = fac[J];
            f[0] = l10; f[1] = l20; f[2] = l30; f[3] = l21; f[4] = l31; f[5] = l32;
            f[6] = i0; f[7] = i1; f[8] = i2; f[9] = i3;
        }
        // column lq of inv(D): solve D e = unit(lq)
        const double u0 = lq == 0 ? 1.0 : 0.0, u1 = lq == 1 ? 1.0 : 0.0, u2 = lq == 2 ? 1.0 : 0.0, u3 = lq == 3 ? 1.0 : 0.0;
        const double y0 = u0;
        const double y1 = u1 - l10 * y0;
        const double y2 = u2 - l20 * y0 - l21 * y1;
        const double y3 = u3 - l30 * y0 - l31 * y1 - l32 * y2;
        const double q3 = y3 * i3;
        const double q2 = y2 * i2 - l32 * q3;
        const double q1 = y1 * i1 - l21 * q2 - l31 * q3;
        const double q0 = y0 * i0 - l10 * q1 - l20 * q2 - l30 * q3;
        // w[r][lq] for this lane's two rows (rows at or above the pivot block are not touched)
        const int r0 = lc, r1 = 16 + lc;
        double w0 = pn[r0][0] * q0 + pn[r0][1] * q1 + pn[r0][2] * q2 + pn[r0][3] * q3;
        double w1 = pn[r1][0] * q0 + pn[r1][1] * q1 + pn[r1][2] * q2 + pn[r1][3] * q3;
        if (r0 < j0 + 4) w0 = 0.0;
        if (r1 < j0 + 4) w1 = 0.0;
        const double bv = pr[p][lq][16 * w + lc];
        m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-w0, bv, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-w1, bv, m1, 0, 0, 0);
    }
    // X (wavefronts 2, 3) to LDS, then the unit-lower solve inside each 4-row group and the sqrt of the pivots
    __syncthreads();
    if (w >= 2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            x[lq + 4 * v][16 * (w - 2) + lc] = m0[v];
            x[16 + lq + 4 * v][16 * (w - 2) + lc] = m1[v];
        }
    }
    __syncthreads();
    const int r = t >> 3, c0 = (t & 7) * 4;
    const int base = r & ~3, q = r & 3;
    const double *f = fac[r >> 2];
    double res[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        const double u0 = x[base][c];
        const double u1 = x[base + 1][c] - f[0] * u0;
        const double u2 = x[base + 2][c] - f[1] * u0 - f[3] * u1;
        const double u3 = x[base + 3][c] - f[2] * u0 - f[4] * u1 - f[5] * u2;
        res[e] = q == 0 ? u0 : (q == 1 ? u1 : (q == 2 ? u2 : u3));
    }
    const double sr = sqrt(f[6 + q]);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) x[r][c0 + e] = (c0 + e <= r) ? res[e] * sr : 0.0;
    __syncthreads();
    return ok;
}

// block_chol_inv32_mf with less LDS traffic and no idle MFMAs (`a` is overwritten: it serves as scratch at the end).  The trailing block stays symmetric under the elimination,
// so the panel column A[r][4J + k] is read as the pivot row's element A[4J + k][r]: only the four pivot rows of the A
// half are published (one value per lane of wavefronts 0, 1); a lane's B operand is its own accumulator element (the
// pivot row it holds), not a round trip through LDS.  Updates that cannot change a live element are not issued: rows
// 0..15 once the pivot has passed them (J >= 3), columns of A the elimination has already left behind (wavefront 0 from
// J = 3), columns of X the pivot rows cannot reach yet (wavefront 3 before J = 4), everything at the last step.
__device__ __forceinline__ bool block_chol_inv32_v4(double (*a)[CH_NB + 1], double (*x)[CH_NB + 1])
{
    typedef double acc4 __attribute__((ext_vector_type(4)));
    __shared__ double pr[2][4][CH_NB]; // pr[buf][k][c] = A[4J + k][c]
    __shared__ double fac[8][10];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int lc = lane & 15, lq = lane >> 4;
    acc4 m0, m1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r0 = lq + 4 * v, r1 = 16 + lq + 4 * v, c = 16 * w + lc;
        if (w < 2) {
            m0[v] = c <= r0 ? a[r0][c] : a[c][r0];
            m1[v] = c <= r1 ? a[r1][c] : a[c][r1];
        } else {
            m0[v] = (c - CH_NB == r0) ? 1.0 : 0.0;
            m1[v] = (c - CH_NB == r1) ? 1.0 : 0.0;
        }
    }
    bool ok = true;
#pragma unroll
    for (int J = 0; J < 8; ++J) {
        const int p = J & 1, j0 = 4 * J;
        const int v0 = (j0 % 16) / 4;
        const double bv = (j0 < 16) ? m0[v0] : m1[v0]; // [A | X][4J + lq][16 w + lc]: this lane's pivot-row element
        if (w < 2) pr[p][lq][16 * w + lc] = bv;
        __syncthreads();
        const double(*pp)[CH_NB] = pr[p];
        const double d00 = pp[0][j0];
        const double d10 = pp[1][j0], d11 = pp[1][j0 + 1];
        const double d20 = pp[2][j0], d21 = pp[2][j0 + 1], d22 = pp[2][j0 + 2];
        const double d30 = pp[3][j0], d31 = pp[3][j0 + 1], d32 = pp[3][j0 + 2], d33 = pp[3][j0 + 3];
        const double i0 = fast_rcp(d00);
        const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
        const double e11 = d11 - l10 * d10;
        const double e21 = d21 - l20 * d10, e31 = d31 - l30 * d10;
        const double i1 = fast_rcp(e11);
        const double l21 = e21 * i1, l31 = e31 * i1;
        const double e22 = d22 - l20 * d20 - l21 * e21;
        const double e32 = d32 - l30 * d20 - l31 * e21;
        const double i2 = fast_rcp(e22);
        const double l32 = e32 * i2;
        const double e33 = d33 - l30 * d30 - l31 * e31 - l32 * e32;
        const double i3 = fast_rcp(e33);
        ok = ok && d00 > 0.0 && e11 > 0.0 && e22 > 0.0 && e33 > 0.0;
        if (t == 0) {
            double *f = fac[J];
            f[0] = l10; f[1] = l20; f[2] = l30; f[3] = l21; f[4] = l31; f[5] = l32;
            f[6] = i0; f[7] = i1; f[8] = i2; f[9] = i3;
        }
        if (J == 7) break; // no rows below the last pivot block
        const bool live = (w == 0 && J < 3) || (w == 1) || (w == 2) || (w == 3 && J >= 4); // wave-uniform, static per J
        if (!live) continue;
        // column lq of inv(D): solve D e = unit(lq)
        const double u0 = lq == 0 ? 1.0 : 0.0, u1 = lq == 1 ? 1.0 : 0.0, u2 = lq == 2 ? 1.0 : 0.0, u3 = lq == 3 ? 1.0 : 0.0;
        const double y0 = u0;
        const double y1 = u1 - l10 * y0;
        const double y2 = u2 - l20 * y0 - l21 * y1;
        const double y3 = u3 - l30 * y0 - l31 * y1 - l32 * y2;
        const double q3 = y3 * i3;
        const double q2 = y2 * i2 - l32 * q3;
        const double q1 = y1 * i1 - l21 * q2 - l31 * q3;
        const double q0 = y0 * i0 - l10 * q1 - l20 * q2 - l30 * q3;
        // w[r][lq] for this lane's rows (rows at or above the pivot block are not touched)
        const int r1 = 16 + lc;
        double w1 = pp[0][r1] * q0 + pp[1][r1] * q1 + pp[2][r1] * q2 + pp[3][r1] * q3;
        if (r1 < j0 + 4) w1 = 0.0;
        if (J < 3) {
            const int r0 = lc;
            double w0 = pp[0][r0] * q0 + pp[1][r0] * q1 + pp[2][r0] * q2 + pp[3][r0] * q3;
            if (r0 < j0 + 4) w0 = 0.0;
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-w0, bv, m0, 0, 0, 0);
        }
        m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-w1, bv, m1, 0, 0, 0);
    }
    // X (wavefronts 2, 3) to LDS -- into `a`, which nobody reads any more: no barrier before, none between the solve's
    // reads and the result's stores --, then the unit-lower solve inside each 4-row group and the sqrt of the pivots
    if (w >= 2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[lq + 4 * v][16 * (w - 2) + lc] = m0[v];
            a[16 + lq + 4 * v][16 * (w - 2) + lc] = m1[v];
        }
    }
    __syncthreads();
    const int r = t >> 3, c0 = (t & 7) * 4;
    const int base = r & ~3, q = r & 3;
    const double *f = fac[r >> 2];
    const double sr = sqrt(f[6 + q]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        const double u0 = a[base][c];
        const double u1 = a[base + 1][c] - f[0] * u0;
        const double u2 = a[base + 2][c] - f[1] * u0 - f[3] * u1;
        const double u3 = a[base + 3][c] - f[2] * u0 - f[4] * u1 - f[5] * u2;
        const double res = q == 0 ? u0 : (q == 1 ? u1 : (q == 2 ? u2 : u3));
        x[r][c] = (c <= r) ? res * sr : 0.0;
    }
    __syncthreads();
    return ok;
}

} // namespace ekf
