// Micro-benchmark of the 32x32 diagonal factor+inverse routines in csrc/chol32.h (one workgroup, one block):
// shader cycles (clock64) and wall time (wall_clock64, 100 MHz) around the routine, numerics against the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/chol32_micro.hip -o /tmp/chol32_micro && /tmp/chol32_micro
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#ifdef W2_TRACE // -DW2_TRACE: shader-clock stamps inside block_chol_inv32_w2 (lane 0 of each wavefront), printed for the last call
__device__ long long g_w2_stamps[64];
#define W2_STAMP(slot) if ((threadIdx.x & 63) == 0) g_w2_stamps[(slot)] = clock64();
#endif
#include "../../openekfmonoslam_amd/csrc/chol32.h"

using namespace ekf;

// ablations of block_chol_inv32_mf (timing only, results are wrong): 1 = rank-4 update without the MFMA, 2 = no reciprocals,
// 4 = no barrier per pivot step
template <int ABL>
__device__ __forceinline__ bool abl_chol(double (*a)[CH_NB + 1], double (*x)[CH_NB + 1])
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
        if (ABL != 4) __syncthreads();
        const double(*pn)[4] = pc[p];
        const double d00 = pn[j0][0];
        const double d10 = pn[j0 + 1][0], d11 = pn[j0 + 1][1];
        const double d20 = pn[j0 + 2][0], d21 = pn[j0 + 2][1], d22 = pn[j0 + 2][2];
        const double d30 = pn[j0 + 3][0], d31 = pn[j0 + 3][1], d32 = pn[j0 + 3][2], d33 = pn[j0 + 3][3];
        const double i0 = ABL == 2 ? d00 : fast_rcp(d00);
        const double l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
        const double e11 = d11 - l10 * d10;
        const double e21 = d21 - l20 * d10, e31 = d31 - l30 * d10;
        const double i1 = ABL == 2 ? e11 : fast_rcp(e11);
        const double l21 = e21 * i1, l31 = e31 * i1;
        const double e22 = d22 - l20 * d20 - l21 * e21;
        const double e32 = d32 - l30 * d20 - l31 * e21;
        const double i2 = ABL == 2 ? e22 : fast_rcp(e22);
        const double l32 = e32 * i2;
        const double e33 = d33 - l30 * d30 - l31 * e31 - l32 * e32;
        const double i3 = ABL == 2 ? e33 : fast_rcp(e33);
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
        if (ABL == 1) { m0[0] -= w0 * bv; m1[0] -= w1 * bv; }
        else {
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-w0, bv, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-w1, bv, m1, 0, 0, 0);
        }
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



template <int V>
__global__ void __launch_bounds__(256) k_abl(const double *A, double *Linv, long long *ticks, int reps)
{
    __shared__ double sa[CH_NB][CH_NB + 1], sx[CH_NB][CH_NB + 1];
    long long c_acc = 0;
    for (int it = 0; it < reps; ++it) {
        for (int i = threadIdx.x; i < CH_NB * CH_NB; i += 256) {
            const int r = i / CH_NB, c = i % CH_NB;
            sa[r][c] = c <= r ? A[r * CH_NB + c] : 0.0;
        }
        __syncthreads();
        const long long c0 = clock64();
        abl_chol<V>(sa, sx);
        c_acc += clock64() - c0;
    }
    for (int i = threadIdx.x; i < CH_NB * CH_NB; i += 256) Linv[i] = sx[i / CH_NB][i % CH_NB];
    if (threadIdx.x == 0) ticks[0] = c_acc;
}

template <int V>
__global__ void __launch_bounds__(256) k_run(const double *A, double *Linv, long long *ticks, int reps)
{
    __shared__ double sa[CH_NB][CH_NB + 1], sx[CH_NB][CH_NB + 1], srs[CH_NB];
    __shared__ W2Scratch w2s;
    long long c_acc = 0, w_acc = 0;
    bool ok = true;
    for (int it = 0; it < reps; ++it) {
        for (int i = threadIdx.x; i < CH_NB * CH_NB; i += 256) {
            const int r = i / CH_NB, c = i % CH_NB;
            sa[r][c] = c <= r ? A[r * CH_NB + c] : 0.0;
        }
        __syncthreads();
        const long long c0 = clock64(), w0 = wall_clock64();
        if (V == 0) ok = block_chol_inv32(sa, sx, srs) && ok;
        else if (V == 1) ok = block_chol_inv32_bp(sa, sx) && ok;
        else if (V == 2) ok = block_chol_inv32_mf(sa, sx) && ok;
        else if (V == 3) ok = block_chol_inv32_v4(sa, sx) && ok;
        else if (V == 4) ok = block_chol_inv32_w2<2>(sa, sx, &w2s) && ok;
        else ok = block_chol_inv32_w2<1>(sa, sx, &w2s) && ok;
        c_acc += clock64() - c0;
        w_acc += wall_clock64() - w0;
    }
    for (int i = threadIdx.x; i < CH_NB * CH_NB; i += 256) Linv[i] = sx[i / CH_NB][i % CH_NB];
    if (threadIdx.x == 0) { ticks[0] = c_acc; ticks[1] = w_acc; ticks[2] = ok; }
}

// accuracy of the hardware reciprocal estimate and of its Newton refinements (relative error, max over 64 K random arguments)
__global__ void k_rcp_err(double *out)
{
    double e0 = 0, e1 = 0, e2 = 0;
    unsigned s = 1234567u + threadIdx.x * 7919u;
    for (int i = 0; i < 256; ++i) {
        s = s * 1664525u + 1013904223u;
        const double d = ldexp(1.0 + (double)(s >> 8) / (1 << 24), (int)(s & 63) - 20);
        const double ex = 1.0 / d;
        double r = __builtin_amdgcn_rcp(d);
        e0 = fmax(e0, fabs(r - ex) / ex);
        r = fma(fma(-d, r, 1.0), r, r);
        e1 = fmax(e1, fabs(r - ex) / ex);
        r = fma(fma(-d, r, 1.0), r, r);
        e2 = fmax(e2, fabs(r - ex) / ex);
    }
    out[threadIdx.x * 3] = e0; out[threadIdx.x * 3 + 1] = e1; out[threadIdx.x * 3 + 2] = e2;
}

int main()
{
    {
        double *dE;
        hipMalloc(&dE, 256 * 3 * 8);
        k_rcp_err<<<1, 256>>>(dE);
        std::vector<double> hE(256 * 3);
        hipMemcpy(hE.data(), dE, hE.size() * 8, hipMemcpyDeviceToHost);
        double m[3] = {0, 0, 0};
        for (int i = 0; i < 256; ++i) for (int k = 0; k < 3; ++k) m[k] = std::fmax(m[k], hE[i * 3 + k]);
        std::printf("v_rcp_f64 relative error: raw %.2e, one Newton step %.2e, two %.2e\n", m[0], m[1], m[2]);
    }
    const int n = CH_NB;
    std::vector<double> M(n * n), A(n * n, 0.0);
    unsigned s = 12345;
    for (auto &v : M) { s = s * 1664525u + 1013904223u; v = (double)(s >> 8) / (1 << 24) - 0.5; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double t = i == j ? 0.5 : 0.0;
            for (int k = 0; k < n; ++k) t += M[i * n + k] * M[j * n + k];
            A[i * n + j] = t;
        }
    double *dA, *dL;
    long long *dT;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dT, 64);
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    for (int v = 0; v < 6; ++v) {
        const int reps = 50;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int warm = 0; warm < 2; ++warm) {
            hipEventRecord(e0);
            if (v == 0) k_run<0><<<1, 256>>>(dA, dL, dT, reps);
            else if (v == 1) k_run<1><<<1, 256>>>(dA, dL, dT, reps);
            else if (v == 2) k_run<2><<<1, 256>>>(dA, dL, dT, reps);
            else if (v == 3) k_run<3><<<1, 256>>>(dA, dL, dT, reps);
            else if (v == 4) k_run<4><<<1, 256>>>(dA, dL, dT, reps);
            else k_run<5><<<1, 256>>>(dA, dL, dT, reps);
            hipEventRecord(e1);
            hipDeviceSynchronize();
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<double> L(n * n);
        long long t[3];
        hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(t, dT, 24, hipMemcpyDeviceToHost);
        // || Linv A Linv' - I ||_max
        double err = 0;
        std::vector<double> T(n * n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double q = 0;
                for (int k = 0; k < n; ++k) q += L[i * n + k] * A[k * n + j];
                T[i * n + j] = q;
            }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double q = 0;
                for (int k = 0; k < n; ++k) q += T[i * n + k] * L[j * n + k];
                err = std::fmax(err, std::fabs(q - (i == j ? 1.0 : 0.0)));
            }
        double upper = 0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) upper = std::fmax(upper, std::fabs(L[i * n + j]));
        std::printf("variant %d: ok=%lld  cycles/call %.0f  wall us/call %.2f  (kernel %.1f us / %d reps = %.2f us)  |LinvALinv'-I| %.2e  upper %.1e\n",
                    v, t[2], (double)t[0] / reps, (double)t[1] / reps / 100.0, ms * 1e3, reps, ms * 1e3 / reps, err, upper);
    }
#ifdef W2_TRACE
    {
        long long st[64];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_w2_stamps), sizeof(st));
        std::printf("w2 stamps (cycles after step 0 start): per step J: start, LDL done, W ready, MFMAs issued | wavefront 1: flag seen\n");
        for (int J = 0; J < 8; ++J)
            std::printf("  J=%d  %6lld %6lld %6lld %6lld | %6lld\n", J, st[4 * J] - st[0], st[4 * J + 1] - st[0], st[4 * J + 2] - st[0],
                        st[4 * J + 3] - st[0], J < 7 ? st[32 + J] - st[0] : 0ll);
        std::printf("  wavefronts 0..3 at the barrier: %lld %lld %lld %lld, after it %lld, end %lld\n", st[40] - st[0], st[41] - st[0], st[42] - st[0],
                    st[43] - st[0], st[44] - st[0], st[45] - st[0]);
    }
#endif
    {
        const int reps = 50;
        long long t[1];
        k_abl<0><<<1, 256>>>(dA, dL, dT, reps); hipDeviceSynchronize(); hipMemcpy(t, dT, 8, hipMemcpyDeviceToHost); std::printf("ablation 0 (nothing removed): %.0f cycles/call\n", (double)t[0] / reps);
        k_abl<1><<<1, 256>>>(dA, dL, dT, reps); hipDeviceSynchronize(); hipMemcpy(t, dT, 8, hipMemcpyDeviceToHost); std::printf("ablation 1 (no MFMA):        %.0f cycles/call\n", (double)t[0] / reps);
        k_abl<2><<<1, 256>>>(dA, dL, dT, reps); hipDeviceSynchronize(); hipMemcpy(t, dT, 8, hipMemcpyDeviceToHost); std::printf("ablation 2 (no reciprocals): %.0f cycles/call\n", (double)t[0] / reps);
        k_abl<4><<<1, 256>>>(dA, dL, dT, reps); hipDeviceSynchronize(); hipMemcpy(t, dT, 8, hipMemcpyDeviceToHost); std::printf("ablation 4 (no barrier):     %.0f cycles/call\n", (double)t[0] / reps);
    }
    return 0;
}
