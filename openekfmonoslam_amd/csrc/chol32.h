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

// ---------------------------------------------------------------------------------------------------------------------------
// block_chol_inv32_w2: the same elimination of [A | I] with 4x4 block pivots, WITHOUT a workgroup barrier on the pivot chain.
// Two free-running wavefronts: wavefront 0 owns the A half (four 16x16 accumulator blocks of the full symmetric matrix),
// wavefront 1 the identity half (three blocks: X01 stays zero); the other wavefronts of the workgroup wait at the final barrier.
//   * the 4x4 pivot block reaches every lane of wavefront 0 through v_readlane (20 scalar reads of the lane's own accumulator
//     registers: no LDS round trip, no barrier);
//   * the panel column A[r][4J + k] = A[4J + k][r] (symmetry) is gathered across the four 16-lane rows through a private LDS
//     image written [column][k] (512 contiguous bytes, read back as 32 contiguous bytes per lane): a single wavefront's LDS
//     operations complete in order, so only s_waitcnt stands between the store and the load, and both run beside the
//     scalar LDL' chain of the pivot block instead of in front of it;
//   * the multipliers -W (already in MFMA A-operand layout) go to wavefront 1 through LDS with one flag word per step (eight
//     distinct slots: nothing is ever reused, so wavefront 0 never waits for wavefront 1), which trails by less than a step and
//     applies them to its own pivot rows;
//   * MFMAs that cannot change a live element are not issued (16 for the A half, 13 for the identity half), and the block that
//     holds the next pivot rows is issued first.
// v4 above costs ~1270 cycles per pivot step (LDS publish, barrier over four wavefronts, broadcast reads, chain, MFMA); here the
// step is the chain alone: readlane -> LDL' -> column of inv(D) -> dot with the gathered panel -> MFMA.
// `a`: lower triangle valid on entry, overwritten (scratch); `x`: inv(L).  256 threads must call it; returns uniformly.
__device__ __forceinline__ double w2_readlane(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

template <int NEWTON = 2>
__device__ __forceinline__ double w2_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int i = 0; i < NEWTON; ++i) r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

#ifndef W2_STAMP
#define W2_STAMP(slot) // scripts/micro/chol32_micro.hip defines it: shader-clock stamps of the two wavefronts
#endif
struct W2Idle {
    __device__ __forceinline__ void operator()(int) const {}
};
// `other(h)`: what wavefronts 2 and 3 (h = 0, 1) do while wavefronts 0 and 1 factorise -- e.g. the persistent sweep's critical
// workgroup fetches the next two tiles of S (chol_persist.h); it must not touch `a`, `x` or a workgroup barrier.
// LDS scratch of block_chol_inv32_w2 (17 KB): the caller owns it (a kernel whose other roles need the space overlays it)
struct __attribute__((aligned(32))) W2Scratch {
    double prow[8][CH_NB][4]; // prow[J][c][k] = A[4J + k][c] (= A[c][4J + k])
    double wbuf[8][2][64];    // -W of step J in A-operand layout, block rows 0 / 1
    double fac[8][10];
    int wflag[8];
    int okflag;
};

template <int NEWTON = 2, class OTHER = W2Idle>
__device__ __forceinline__ bool block_chol_inv32_w2(double (*a)[CH_NB + 1], double (*x)[CH_NB + 1], W2Scratch *ws, OTHER &&other = W2Idle())
{
    typedef double acc4 __attribute__((ext_vector_type(4)));
    double(*prow)[CH_NB][4] = ws->prow;
    double(*wbuf)[2][64] = ws->wbuf;
    double(*fac)[10] = ws->fac;
    int *wflag = ws->wflag;
    int &okflag = ws->okflag;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int lc = lane & 15, lq = lane >> 4;
    if (t < 8) wflag[t] = 0;
    if (t == 8) okflag = 1;
    __syncthreads();
    if (w == 0) {
        acc4 m[2][2];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int r = 16 * bi + lq + 4 * v, c = 16 * bj + lc;
                    m[bi][bj][v] = c <= r ? a[r][c] : a[c][r];
                }
        bool ok = true;
        double nw1_pend = 0.0, bv1_pend = 0.0; // the m[1][1] update of the previous step, issued behind this step's LDL' chain
#pragma unroll
        for (int J = 0; J < 8; ++J) {
            const int j0 = 4 * J, bp = J >> 2, v0 = J & 3, cc = j0 & 15;
            W2_STAMP(4 * J)
            const double bv0 = m[bp][0][v0], bv1 = m[bp][1][v0]; // pivot row 4J + lq at columns lc and 16 + lc
            // panel image for the gather (columns left of the pivot block are dead: J >= 4 needs columns 16.. only), requested back
            // at once: the LDS round trip runs beside the scalar chain below, not behind it
            if (J < 4) prow[J][lc][lq] = bv0;
            prow[J][16 + lc][lq] = bv1;
            asm volatile("" ::: "memory"); // (one wavefront: its LDS stores and loads complete in order; the compiler must keep that order too)
            double p0[4] = {0.0, 0.0, 0.0, 0.0}, p1[4];
            if (J < 7) {
#pragma unroll
                for (int k = 0; k < 4; ++k) p1[k] = prow[J][16 + lc][k];
                if (J < 3) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) p0[k] = prow[J][lc][k];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the pivot block, to every lane: D[k][k'] = A[4J + k][4J + k'] sits in lane 16 k + cc + k' of bv_{bp}
            const double bvp = bp ? bv1 : bv0;
            const double d00 = w2_readlane(bvp, cc);
            const double d10 = w2_readlane(bvp, 16 + cc), d11 = w2_readlane(bvp, 16 + cc + 1);
            const double d20 = w2_readlane(bvp, 32 + cc), d21 = w2_readlane(bvp, 32 + cc + 1), d22 = w2_readlane(bvp, 32 + cc + 2);
            const double d30 = w2_readlane(bvp, 48 + cc), d31 = w2_readlane(bvp, 48 + cc + 1), d32 = w2_readlane(bvp, 48 + cc + 2),
                         d33 = w2_readlane(bvp, 48 + cc + 3);
            const double i0 = w2_rcp<NEWTON>(d00);
            const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
            const double e11 = d11 - l10 * d10;
            const double e21 = d21 - l20 * d10, e31 = d31 - l30 * d10;
            const double i1 = w2_rcp<NEWTON>(e11);
            const double l21 = e21 * i1, l31 = e31 * i1;
            const double e22 = d22 - l20 * d20 - l21 * e21;
            const double e32 = d32 - l30 * d20 - l31 * e21;
            const double i2 = w2_rcp<NEWTON>(e22);
            const double l32 = e32 * i2;
            const double e33 = d33 - l30 * d30 - l31 * e31 - l32 * e32;
            const double i3 = w2_rcp<NEWTON>(e33);
            W2_STAMP(4 * J + 1)
            ok = ok && d00 > 0.0 && e11 > 0.0 && e22 > 0.0 && e33 > 0.0;
            if (lane == 0) {
                double *f = fac[J];
                f[0] = l10; f[1] = l20; f[2] = l30; f[3] = l21; f[4] = l31; f[5] = l32;
                f[6] = i0; f[7] = i1; f[8] = i2; f[9] = i3;
            }
            if (J == 7) break;
            // the previous step's update of block (1, 1) -- nobody needs it before the fifth pivot step -- goes into the matrix
            // pipe here, behind the chain that only needed blocks (0, 0) and (0, 1)
            if (J >= 1 && J <= 3) {
                __builtin_amdgcn_sched_barrier(0);
                m[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(nw1_pend, bv1_pend, m[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // column lq of inv(D): solve D e = unit(lq)
            const double u1 = lq == 1 ? 1.0 : 0.0, u2 = lq == 2 ? 1.0 : 0.0, u3 = lq == 3 ? 1.0 : 0.0;
            const double y0 = lq == 0 ? 1.0 : 0.0;
            const double y1 = u1 - l10 * y0;
            const double y2 = u2 - l20 * y0 - l21 * y1;
            const double y3 = u3 - l30 * y0 - l31 * y1 - l32 * y2;
            const double q3 = y3 * i3;
            const double q2 = y2 * i2 - l32 * q3;
            const double q1 = y1 * i1 - l21 * q2 - l31 * q3;
            const double q0 = y0 * i0 - l10 * q1 - l20 * q2 - l30 * q3;
            // W[r][lq] = sum_k' A[r][4J + k'] inv(D)[k'][lq] for the lane's rows lc (block row 0, live while J < 3) and 16 + lc
            double nw0 = 0.0;
            double nw1 = -((p1[0] * q0 + p1[1] * q1) + (p1[2] * q2 + p1[3] * q3));
            if (16 + lc < j0 + 4) nw1 = 0.0;
            if (J < 3) {
                nw0 = -((p0[0] * q0 + p0[1] * q1) + (p0[2] * q2 + p0[3] * q3));
                if (lc < j0 + 4) nw0 = 0.0;
            }
            W2_STAMP(4 * J + 2)
            wbuf[J][0][lane] = nw0;
            wbuf[J][1][lane] = nw1;
            // data, then flag: a wavefront's LDS operations are processed in issue order, so no wait is needed between them (a
            // workgroup-scope release fence would also drain this wavefront's global-memory counter)
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(&wflag[J], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            // the blocks with the next pivot rows now; block (1, 1) of the steps before the fourth waits (see above)
            if (J < 3) {
                m[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(nw0, bv0, m[0][0], 0, 0, 0);
                m[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(nw0, bv1, m[0][1], 0, 0, 0);
                nw1_pend = nw1;
                bv1_pend = bv1;
            } else {
                m[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(nw1, bv1, m[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            W2_STAMP(4 * J + 3)
        }
        if (!ok && lane == 0) okflag = 0;
    } else if (w == 1) {
        acc4 x00, x10, x11;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = lq + 4 * v;
            x00[v] = r == lc ? 1.0 : 0.0;
            x10[v] = 0.0;
            x11[v] = r == lc ? 1.0 : 0.0;
        }
#pragma unroll
        for (int J = 0; J < 7; ++J) {
            const int v0 = J & 3;
            while (__hip_atomic_load(&wflag[J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            W2_STAMP(32 + J)
            const double nw0 = wbuf[J][0][lane], nw1 = wbuf[J][1][lane];
            if (J < 3) {
                const double b0 = x00[v0];
                x00 = __builtin_amdgcn_mfma_f64_16x16x4f64(nw0, b0, x00, 0, 0, 0);
                x10 = __builtin_amdgcn_mfma_f64_16x16x4f64(nw1, b0, x10, 0, 0, 0);
            } else if (J == 3) {
                x10 = __builtin_amdgcn_mfma_f64_16x16x4f64(nw1, x00[v0], x10, 0, 0, 0);
            } else {
                const double b0 = x10[v0], b1 = x11[v0];
                x10 = __builtin_amdgcn_mfma_f64_16x16x4f64(nw1, b0, x10, 0, 0, 0);
                x11 = __builtin_amdgcn_mfma_f64_16x16x4f64(nw1, b1, x11, 0, 0, 0);
            }
        }
        // X to LDS (into `a`, which wavefront 0 has read completely before its first flag: this wavefront passed flag 6)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            a[lq + 4 * v][lc] = x00[v];
            a[lq + 4 * v][16 + lc] = 0.0;
            a[16 + lq + 4 * v][lc] = x10[v];
            a[16 + lq + 4 * v][16 + lc] = x11[v];
        }
    } else {
        other(w - 2);
    }
    W2_STAMP(40 + w)
    __syncthreads();
    W2_STAMP(44)
    // inv(L) = blockdiag(inv(chol(D_J))) X : unit-lower solve inside each 4-row group, then sqrt of the pivots
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
    const bool okr = okflag != 0;
    __syncthreads();
    W2_STAMP(45)
    return okr;
}

} // namespace ekf
