// Micro-benchmark of the 32x32 diagonal factor+inverse routines in csrc/chol32.h (one workgroup, one block):
// shader cycles (clock64) and wall time (wall_clock64, 100 MHz) around the routine, numerics against the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/chol32_micro.hip -o /tmp/chol32_micro && /tmp/chol32_micro
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../../openekfmonoslam_amd/csrc/chol32.h"

using namespace ekf;

template <int V>
__global__ void __launch_bounds__(256) k_run(const double *A, double *Linv, long long *ticks, int reps)
{
    __shared__ double sa[CH_NB][CH_NB + 1], sx[CH_NB][CH_NB + 1], srs[CH_NB];
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
        else ok = block_chol_inv32_mf(sa, sx) && ok;
        c_acc += clock64() - c0;
        w_acc += wall_clock64() - w0;
    }
    for (int i = threadIdx.x; i < CH_NB * CH_NB; i += 256) Linv[i] = sx[i / CH_NB][i % CH_NB];
    if (threadIdx.x == 0) { ticks[0] = c_acc; ticks[1] = w_acc; ticks[2] = ok; }
}

int main()
{
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
    for (int v = 0; v < 3; ++v) {
        const int reps = 50;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int warm = 0; warm < 2; ++warm) {
            hipEventRecord(e0);
            if (v == 0) k_run<0><<<1, 256>>>(dA, dL, dT, reps);
            else if (v == 1) k_run<1><<<1, 256>>>(dA, dL, dT, reps);
            else k_run<2><<<1, 256>>>(dA, dL, dT, reps);
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
    return 0;
}
