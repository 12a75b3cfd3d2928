// Stand-alone bench + check of the EXACT covariance downdate (csrc/kernels_pexact.hip: column scales, digit planes,
// k_p_update_i8) on random data, beside the fp32 MFMA downdate (csrc/kernels_pupdate.hip) on the same B rounded to fp32.
//   * sampled entries of the result are compared BITWISE with a CPU evaluation of the same integer algorithm (digits, level
//     sums in int64, the same fp64 combination, one rounding to fp32), and against an fp64 evaluation of P - B'B;
//   * the whole result is checked for bitwise symmetry;
//   * timing: HIP events per kernel, interleaved rounds, median and min.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I openekfmonoslam_amd/csrc scripts/micro/pu_i8_bench.hip -o scripts/micro/pu_i8_bench
//   scripts/micro/pu_i8_bench [N=1000] [m list, e.g. 298,1014,2000] [rounds]
#define PX_BENCH
#define PU_BENCH
#include "../../openekfmonoslam_amd/csrc/kernels_pupdate.hip"
#include "../../openekfmonoslam_amd/csrc/kernels_pexact.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace ekf;

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static inline double drand()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)((rng_state >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53) * 2.0 - 1.0;
}

static float evt_ms(hipEvent_t a, hipEvent_t b) { float ms = 0.f; hipEventElapsedTime(&ms, a, b); return ms; }

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 1000;
    std::vector<int> ms;
    {
        const char *s = argc > 2 ? argv[2] : "298,1014,2000";
        char *dup = strdup(s);
        for (char *t = strtok(dup, ","); t; t = strtok(nullptr, ",")) ms.push_back(atoi(t));
        free(dup);
    }
    const int rounds = argc > 3 ? atoi(argv[3]) : 7;
    const int variant = argc > 4 ? atoi(argv[4]) : 0; // 0 persistent kernel; 1 one workgroup per unit; 2 the AVG instance (first update after an upload)
    g_px_variant = (variant == 1 || variant == 3 || variant == 4) ? variant : 0; // 3: four wavefronts with 64 x 64 each (k_p_update_i8q); 4: two workgroups per CU (k_p_update_i8d)
    const int n = 13 + 6 * N, ld = round_up(n, LD_ALIGN);
    int m_max = 0;
    for (int m : ms) m_max = std::max(m_max, m);
    const int m_cap = round_up(m_max, 64) + 64;
    std::vector<float> hP((size_t)(n + 128) * ld, 0.f);
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            const float v = (float)((i == j ? 2.0 : 0.0) + 0.25 * drand());
            hP[(size_t)i * ld + j] = v;
            hP[(size_t)j * ld + i] = v;
        }
    // B: column scales over six decades, a few "spikes" per column (a feature's own measurement rows), the rest small
    std::vector<double> hB((size_t)m_cap * ld, 0.0);
    std::vector<float> hBf((size_t)m_cap * ld, 0.f);
    std::vector<double> cscale(ld);
    for (int j = 0; j < ld; ++j) cscale[j] = pow(10.0, -3.0 + 3.0 * drand());
    float *dP, *dP0, *dBf;
    double *dB;
    hipMalloc(&dP, hP.size() * 4); hipMalloc(&dP0, hP.size() * 4);
    hipMalloc(&dB, hB.size() * 8); hipMalloc(&dBf, hBf.size() * 4);
    hipMemcpy(dP0, hP.data(), hP.size() * 4, hipMemcpyHostToDevice);

    EkfEngine e;
    e.n = n; e.N = N; e.ldP = ld; e.f32 = true; e.shard_world = 1;
    e.rm = RowMap{13, n, 13};
    {
        hipDeviceProp_t prop;
        hipGetDeviceProperties(&prop, 0);
        e.n_cus = prop.multiProcessorCount;
    }
    hipStreamCreate(&e.stream);
    e.d.P = dP;
    e.bq_rows = m_cap;
    hipMalloc(&e.d.Bq, (size_t)PX_S * m_cap * ld);
    hipMalloc(&e.d.Bexp, sizeof(int) * ld);
    hipMemset(e.d.Bq, 0, (size_t)PX_S * m_cap * ld);
    e.bz_stride = m_cap / 16; // the table of non-zero pieces of plane 0 (k_slice_B writes it, k_p_update_i8p skips by it)
    hipMalloc(&e.d.Bz, (size_t)(ld / 32 + 8) * e.bz_stride);
    hipMemset(e.d.Bz, 0, (size_t)(ld / 32 + 8) * e.bz_stride);
    // PX_BENCH_RATIO: a column's typical entry = its largest / ratio (the engine's frames: ~300; default sqrt(m): a dense plane 0)
    const double ratio_env = getenv("PX_BENCH_RATIO") ? atof(getenv("PX_BENCH_RATIO")) : 0.0;
    e.timing = true;

    int rc = 0;
    for (int m : ms) {
        const double amp = ratio_env > 0.0 ? 1.0 / ratio_env : 1.0 / sqrt((double)m);
        for (int k = 0; k < m_cap; ++k)
            for (int j = 0; j < ld; ++j) {
                double v = 0.0;
                if (k < m && j < n) {
                    v = cscale[j] * amp * drand();
                    if (ratio_env > 0.0) { // a feature's own measurement rows: along the band k ~ m j / n
                        if (std::abs(k - (int)((long long)j * m / n)) <= 1) v = cscale[j] * (0.5 + 0.5 * fabs(drand()));
                    } else if ((k * 131 + j * 17) % 97 == 0) v = cscale[j] * drand(); // spike
                }
                hB[(size_t)k * ld + j] = v;
                hBf[(size_t)k * ld + j] = (float)v;
            }
        hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dBf, hBf.data(), hBf.size() * 4, hipMemcpyHostToDevice);

        // ---- correctness of the exact path
        hipMemcpy(dP, dP0, hP.size() * 4, hipMemcpyDeviceToDevice);
        e.p_exact_sym = variant != 2; // (the input is symmetric: the averaging instance must give the same bits)
        e.d.A = dB;
        launch_p_update_exact(&e, m, false);
        hipStreamSynchronize(e.stream);
        hipError_t st = hipGetLastError();
        if (st != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(st)); return 2; }
        std::vector<float> out(hP.size());
        hipMemcpy(out.data(), dP, out.size() * 4, hipMemcpyDeviceToHost);
        std::vector<int> hexp(ld);
        hipMemcpy(hexp.data(), e.d.Bexp, sizeof(int) * ld, hipMemcpyDeviceToHost);
        long long asym = 0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j)
                if (memcmp(&out[(size_t)i * ld + j], &out[(size_t)j * ld + i], 4) != 0) ++asym;
        // sampled entries: the integer algorithm on the CPU
        auto digits = [&](int k, int j, int *d) {
            const int sh = 8 * PX_S - 2 - (hexp[j] - 1022);
            long long X = llrint(ldexp(hB[(size_t)k * ld + j], sh));
            for (int s = PX_S - 1; s >= 0; --s) {
                const int dd = (int)((X + 128) & 255) - 128;
                X = (X - dd) >> 8;
                d[s] = dd;
            }
        };
        int n_bad = 0, n_chk = 0;
        double max_err64 = 0.0, max_err32 = 0.0, max_ref = 0.0;
        std::vector<float> out32; // filled below for the fp32 kernel's error on the same samples
        std::vector<std::pair<int, int>> samples;
        for (int t = 0; t < 3000; ++t) {
            int i = (int)((drand() * 0.5 + 0.5) * n) % n, j = (int)((drand() * 0.5 + 0.5) * n) % n;
            if (t < 200) j = i;                      // diagonal entries
            else if (t < 400) i = (i / 128) * 128 + (j % 128), j = j; // same tile row region
            if (t >= 400 && t < 600) { i = n - 1 - (t % 50); j = n - 1 - ((t * 7) % 120); } // the ragged last tile
            samples.emplace_back(i, j);
        }
        for (auto &sp : samples) {
            const int i = sp.first, j = sp.second;
            long long lv[PX_S] = {};
            double exact = 0.0;
            for (int k = 0; k < m; ++k) {
                int di[PX_S], dj[PX_S];
                digits(k, i, di);
                digits(k, j, dj);
                for (int s = 0; s < PX_S; ++s)
                    for (int u = 0; u < PX_S - s; ++u) lv[s + u] += (long long)di[s] * dj[u];
                exact += hB[(size_t)k * ld + i] * hB[(size_t)k * ld + j];
            }
            double tsum = (double)lv[PX_S - 1];
            for (int L = PX_S - 2; L >= 0; --L) tsum = tsum / 256.0 + (double)lv[L];
            const double v = ldexp(tsum, (hexp[i] - 1022) + (hexp[j] - 1022) - 12);
            const float want = (float)((double)hP[(size_t)i * ld + j] - v);
            const float got = out[(size_t)i * ld + j];
            ++n_chk;
            if (memcmp(&want, &got, 4) != 0) {
                if (n_bad < 5) printf("  MISMATCH (%d,%d): want %.9g got %.9g (old %.9g)\n", i, j, want, got, hP[(size_t)i * ld + j]);
                ++n_bad;
            }
            const double ref = (double)hP[(size_t)i * ld + j] - exact;
            max_err64 = std::max(max_err64, fabs(v - exact));
            max_ref = std::max(max_ref, fabs(exact));
            (void)ref;
        }
        // the fp32 kernel on the same data, for its error on the same samples
        hipMemcpy(dP, dP0, hP.size() * 4, hipMemcpyDeviceToDevice);
        e.d.A = dBf;
        e.p_exact_sym = true;
        e.pu_slots = 0;
        launch_p_update(&e, round_up(m, NB), m);
        hipStreamSynchronize(e.stream);
        out32.resize(hP.size());
        hipMemcpy(out32.data(), dP, out32.size() * 4, hipMemcpyDeviceToHost);
        double max_e32_new = 0.0, max_e32_old = 0.0;
        for (auto &sp : samples) {
            const int i = sp.first, j = sp.second;
            double exact = 0.0;
            for (int k = 0; k < m; ++k) exact += hB[(size_t)k * ld + i] * hB[(size_t)k * ld + j];
            const double ref = (double)hP[(size_t)i * ld + j] - exact;
            max_e32_new = std::max(max_e32_new, fabs((double)out[(size_t)i * ld + j] - ref));
            max_e32_old = std::max(max_e32_old, fabs((double)out32[(size_t)i * ld + j] - ref));
        }
        printf("m=%d n=%d: exact path: %d/%d sampled entries bitwise equal to the CPU integer algorithm, asymmetric pairs %lld; "
               "max |v - fp64 B'B| %.3e (max |B'B| %.3e); max |P_new - fp64 result|: exact path %.3e, fp32 MFMA kernel %.3e\n",
               m, n, n_chk - n_bad, n_chk, asym, max_err64, max_ref, max_e32_new, max_e32_old);
        if (n_bad || asym) rc = 1;
        if (variant == 3 || variant == 4) { // the whole matrix against the eight-wavefront kernel's bits
            const int keep = g_px_variant;
            g_px_variant = 0;
            hipMemcpy(dP, dP0, hP.size() * 4, hipMemcpyDeviceToDevice);
            e.d.A = dB;
            launch_p_update_exact(&e, m, false);
            hipStreamSynchronize(e.stream);
            g_px_variant = keep;
            std::vector<float> ref(hP.size());
            hipMemcpy(ref.data(), dP, ref.size() * 4, hipMemcpyDeviceToHost);
            long long diff = 0;
            for (int i = 0; i < n; ++i)
                if (memcmp(&ref[(size_t)i * ld], &out[(size_t)i * ld], 4 * (size_t)n) != 0)
                    for (int j = 0; j < n; ++j) diff += memcmp(&ref[(size_t)i * ld + j], &out[(size_t)i * ld + j], 4) != 0;
            printf("m=%d: entries that differ from k_p_update_i8p's: %lld of %lld\n", m, diff, (long long)n * n);
            if (diff) rc = 1;
        }
        for (auto &pr : e.pu_events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
        for (auto &pr : e.px_events) { hipEventDestroy(pr.first); }
        e.pu_events.clear(); e.px_events.clear(); e.pu_work.clear(); e.pu_m.clear();

        // ---- timing, interleaved
        std::vector<float> t_px, t_slice, t_f32;
        for (int r = 0; r < rounds + 1; ++r) {
            hipMemcpyAsync(dP, dP0, hP.size() * 4, hipMemcpyDeviceToDevice, e.stream);
            e.d.A = dB;
            launch_p_update_exact(&e, m, false);
            hipMemcpyAsync(dP, dP0, hP.size() * 4, hipMemcpyDeviceToDevice, e.stream);
            e.d.A = dBf;
            e.pu_slots = 0;
            launch_p_update(&e, round_up(m, NB), m);
            hipStreamSynchronize(e.stream);
            if (r > 0) {
                t_px.push_back(evt_ms(e.pu_events[0].first, e.pu_events[0].second));
                t_slice.push_back(evt_ms(e.px_events[0].first, e.px_events[0].second));
                t_f32.push_back(evt_ms(e.pu_events[1].first, e.pu_events[1].second));
            }
            for (auto &pr : e.pu_events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
            for (auto &pr : e.px_events) { hipEventDestroy(pr.first); }
            e.pu_events.clear(); e.px_events.clear(); e.pu_work.clear(); e.pu_m.clear();
        }
        auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mn = [](std::vector<float> v) { return *std::min_element(v.begin(), v.end()); };
        const double flop = (double)n * n * m;
        printf("m=%d: k_p_update_i8 median %.1f us (min %.1f) = %.1f TFLOP/s fp32-equivalent, %.0f int8 TOPS (15/2 n^2 m MACs x 2); "
               "column scales + digit planes %.1f us (min %.1f); fp32 MFMA kernel %.1f us (min %.1f) = %.1f TFLOP/s\n",
               m, med(t_px) * 1e3, mn(t_px) * 1e3, flop / (med(t_px) * 1e-3) * 1e-12, 15.0 * flop / (med(t_px) * 1e-3) * 1e-12,
               med(t_slice) * 1e3, mn(t_slice) * 1e3, med(t_f32) * 1e3, mn(t_f32) * 1e3, flop / (med(t_f32) * 1e-3) * 1e-12);
    }
    return rc;
}
