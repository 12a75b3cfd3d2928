// Timeline of one downdate launch (debug build of the kernel with -DPU_STAMPS): per unit, when it started, left its k-loop and
// ended.  Prints, per 10 us bin, how many workgroups are in their k-loop and how many in their epilogue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPU_STAMPS -I openekfmonoslam_amd/csrc scripts/micro/pu_stamps.hip -o scripts/micro/pu_stamps
#include "../../openekfmonoslam_amd/csrc/kernels_pupdate.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ekf;

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 1000, m = argc > 2 ? atoi(argv[2]) : 1014;
    const int n = 13 + 6 * N, ld = round_up(n, LD_ALIGN), m_pad = round_up(m, NB);
    std::vector<float> hP((size_t)(n + 128) * ld, 0.f), hB((size_t)(m_pad + NB) * ld, 0.f);
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (1 << 24) * 2.f - 1.f; };
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) hP[(size_t)i * ld + j] = hP[(size_t)j * ld + i] = (i == j ? 2.f : 0.f) + 0.25f * rnd();
    for (int k = 0; k < m; ++k)
        for (int j = 0; j < n; ++j) hB[(size_t)k * ld + j] = 0.5f / sqrtf((float)m) * rnd();
    float *dP, *dB;
    hipMalloc(&dP, hP.size() * 4); hipMalloc(&dB, hB.size() * 4);
    hipMemcpy(dP, hP.data(), hP.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    EkfEngine e;
    e.n = n; e.N = N; e.ldP = ld; e.f32 = true; e.shard_world = 1; e.rm = RowMap{13, n, 13}; e.p_exact_sym = true;
    hipStreamCreate(&e.stream);
    e.d.P = dP; e.d.A = dB;
    for (int w = 0; w < 3; ++w) launch_p_update(&e, m_pad, m); // warm up, builds the unit list
    const int grid = e.pu_per_xcd * 8;
    unsigned long long *dS;
    hipMalloc(&dS, (size_t)grid * 32);
    hipMemset(dS, 0, (size_t)grid * 32);
    hipMemcpyToSymbol(HIP_SYMBOL(g_pu_stamps), &dS, sizeof(dS));
    hipDeviceSynchronize();
    launch_p_update(&e, m_pad, m);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st((size_t)grid * 4);
    hipMemcpy(st.data(), dS, st.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < grid; ++b)
        if (st[4 * b]) { t0 = std::min(t0, st[4 * b]); t1 = std::max(t1, st[4 * b + 2]); }
    printf("N=%d m=%d grid %d: launch spans %.1f us (first unit start -> last unit end)\n", N, m, grid, (t1 - t0) * 0.01);
    const int bins = (int)((t1 - t0) / 1000) + 1;
    std::vector<int> loop(bins, 0), epi(bins, 0);
    double epi_full = 0, epi_half = 0, loop_full = 0, loop_half = 0;
    int nf = 0, nh = 0;
    for (int b = 0; b < grid; ++b) {
        if (!st[4 * b]) continue;
        const bool full = st[4 * b + 3] & 1;
        const double tl = (st[4 * b + 1] - st[4 * b]) * 0.01, te = (st[4 * b + 2] - st[4 * b + 1]) * 0.01;
        if (full) { loop_full += tl; epi_full += te; ++nf; } else { loop_half += tl; epi_half += te; ++nh; }
        for (unsigned long long t = st[4 * b]; t < st[4 * b + 2]; t += 1000) {
            const int bi = (int)((t - t0) / 1000);
            if (t < st[4 * b + 1]) ++loop[bi]; else ++epi[bi];
        }
    }
    printf("whole tiles: %d, k-loop %.1f us, epilogue %.1f us on average; half units: %d, k-loop %.1f us, epilogue %.1f us\n", nf,
           loop_full / std::max(nf, 1), epi_full / std::max(nf, 1), nh, loop_half / std::max(nh, 1), epi_half / std::max(nh, 1));
    printf("time [us]: workgroups in their k-loop / in their epilogue (sampled every 10 us)\n");
    for (int i = 0; i < bins; ++i) printf("  %4d: %4d / %4d\n", i * 10, loop[i], epi[i]);
    return 0;
}
