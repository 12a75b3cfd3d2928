// Stand-alone A/B bench of the covariance downdate kernel  P <- P - B'B  (csrc/kernels_pupdate.hip) on random data:
// every variant listed in main() is checked against an fp64 CPU evaluation of sampled entries (and for a bitwise symmetric
// result), then timed with HIP events, interleaved rounds, median and min per variant (guide rule 24).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I openekfmonoslam_amd/csrc scripts/micro/pu_bench.hip -o scripts/micro/pu_bench
//   scripts/micro/pu_bench [N=1000] [m list, e.g. 298,1014,2000]
#define PU_BENCH
#include "../../openekfmonoslam_amd/csrc/kernels_pupdate.hip"
#ifdef PU_BENCH_ABLATIONS
#include "pu_v3.h"
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

using namespace ekf;

struct Variant {
    std::string name;
    std::function<void(EkfEngine *, int /*m_pad*/, int /*m*/)> launch;
};

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static inline float frand()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) & 0xFFFFFF) / (float)(1 << 24) * 2.f - 1.f;
}

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
    const int n = 13 + 6 * N, ld = round_up(n, LD_ALIGN);
    int m_max = 0;
    for (int m : ms) m_max = std::max(m_max, m);
    const int m_cap = round_up(m_max, NB) + NB;
    std::vector<float> hP((size_t)(n + 128) * ld, 0.f), hB((size_t)m_cap * ld, 0.f);
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            const float v = (i == j ? 2.f : 0.f) + 0.25f * frand();
            hP[(size_t)i * ld + j] = v;
            hP[(size_t)j * ld + i] = v;
        }
    float *dP, *dP0, *dB;
    hipMalloc(&dP, hP.size() * 4); hipMalloc(&dP0, hP.size() * 4); hipMalloc(&dB, hB.size() * 4);
    hipMemcpy(dP0, hP.data(), hP.size() * 4, hipMemcpyHostToDevice);

    EkfEngine e;
    e.n = n; e.N = N; e.ldP = ld; e.f32 = true; e.shard_world = 1;
    e.rm = RowMap{13, n, 13};
    e.p_exact_sym = true;
    hipStreamCreate(&e.stream);
    e.d.P = dP; e.d.A = dB;

    std::vector<Variant> variants;
    // shipped: launch_p_update's own choice of split and order; the others force one
    variants.push_back({"shipped", [](EkfEngine *en, int m_pad, int m) { en->pu_slots = 0; launch_p_update(en, m_pad, m); }});
    variants.push_back({"legacy_split", [](EkfEngine *en, int m_pad, int m) { en->pu_slots = -1; g_pu_order_override = m_pad >= 512 ? 1 : 0; launch_p_update(en, m_pad, m); g_pu_order_override = -1; }});
    if (getenv("PU_STAGGER")) {
        static const int vals[] = {0, 40, 60, 120, 160, 240};
        for (int v : vals)
            variants.push_back({"stagger" + std::to_string(v), [v](EkfEngine *en, int m_pad, int m) { en->pu_slots = 0; g_pu_stagger_override = v; launch_p_update(en, m_pad, m); g_pu_stagger_override = -1; }});
    }
    if (getenv("PU_SPLITS")) { // unit lists balanced against other slot counts (4096: half units only), with and without the stagger
        static const int slots[] = {768, 1024, 1536, 4096};
        for (int sl : slots)
            for (int st : {0, 80}) {
                variants.push_back({"slots" + std::to_string(sl) + (st ? "_st80" : "_st0"), [sl, st](EkfEngine *en, int m_pad, int m) {
                    en->pu_slots = 0; g_pu_force_slots = sl; g_pu_stagger_override = st; launch_p_update(en, m_pad, m); g_pu_force_slots = 0; g_pu_stagger_override = -1; }});
            }
    }
    if (getenv("PU_ORDERS")) {
        variants.push_back({"bal_halves_first", [](EkfEngine *en, int m_pad, int m) { en->pu_slots = 0; g_pu_order_override = 1; launch_p_update(en, m_pad, m); g_pu_order_override = -1; }});
        variants.push_back({"bal_mixed_HFH", [](EkfEngine *en, int m_pad, int m) { en->pu_slots = 0; g_pu_order_override = 2; launch_p_update(en, m_pad, m); g_pu_order_override = -1; }});
        variants.push_back({"bal_mixed_FHF", [](EkfEngine *en, int m_pad, int m) { en->pu_slots = 0; g_pu_order_override = 3; launch_p_update(en, m_pad, m); g_pu_order_override = -1; }});
    }
#ifdef PU_BENCH_ABLATIONS
#define ABLV(name, A) variants.push_back({name, [](EkfEngine *en, int m_pad, int m) { en->pu_slots = 768; launch_pu_abl<A>(en, m_pad, m); }});
    ABLV("abl3_noepi_noload(read2_b32 x16)", 3)
    ABLV("abl7_no_lds_reads", 7)
    ABLV("abl19_b128x8_upfront", 19)
    ABLV("abl67_b64x16_interleaved", 67)
    ABLV("abl195_b64x16_upfront", 195)
    ABLV("abl259_b128x8_interleaved", 259)
    ABLV("abl15_mfma_only", 15)
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int m : ms) {
        const int m_pad = round_up(m, NB);
        std::fill(hB.begin(), hB.end(), 0.f);
        const float sc = 0.5f / sqrtf((float)m);
        for (int k = 0; k < m; ++k)
            for (int j = 0; j < n; ++j) hB[(size_t)k * ld + j] = sc * frand();
        hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        // correctness: one launch from P0, sampled entries against fp64
        std::vector<float> out(hP.size());
        for (auto &v : variants) {
            hipMemcpy(dP, dP0, hP.size() * 4, hipMemcpyDeviceToDevice);
            e.p_exact_sym = true;
            v.launch(&e, m_pad, m);
            hipStreamSynchronize(e.stream);
            hipMemcpy(out.data(), dP, hP.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0.0;
            long asym = 0;
            unsigned long long s = 12345;
            for (int t = 0; t < 6000; ++t) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                int i = (int)((s >> 33) % (unsigned)n);
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                int j = (int)((s >> 33) % (unsigned)n);
                if (t < 200) { i = n - 1 - (t % 20); j = n - 1 - (t / 20); }       // last tile
                else if (t < 400) { i = (t * 37) % n; j = i; }                      // diagonal
                else if (t < 600) { i = t % 13; }                                   // camera rows
                double acc = 0.0;
                for (int k = 0; k < m; ++k) acc += (double)hB[(size_t)k * ld + i] * (double)hB[(size_t)k * ld + j];
                const double ref = (double)hP[(size_t)i * ld + j] - acc;
                worst = std::max(worst, std::fabs(ref - (double)out[(size_t)i * ld + j]));
                if (out[(size_t)i * ld + j] != out[(size_t)j * ld + i]) ++asym;
            }
            // full symmetry scan + untouched padding columns
            long asym_full = 0;
            for (int i = 0; i < n; i += 7)
                for (int j = 0; j < n; ++j) asym_full += out[(size_t)i * ld + j] != out[(size_t)j * ld + i];
            printf("check m=%d %-12s max|err| %.3e  asym(sample) %ld asym(rows%%7) %ld %s\n", m, v.name.c_str(), worst, asym, asym_full,
                   (worst < 2e-5 && asym == 0 && asym_full == 0) ? "OK" : "FAIL");
        }
        // timing: interleaved rounds, `inner` launches per bracket
        const int inner = 5;
        std::vector<std::vector<float>> t(variants.size());
        for (int r = 0; r < rounds + 1; ++r)
            for (size_t vi = 0; vi < variants.size(); ++vi) {
                hipMemcpyAsync(dP, dP0, hP.size() * 4, hipMemcpyDeviceToDevice, e.stream);
                hipEventRecord(e0, e.stream);
                for (int it = 0; it < inner; ++it) variants[vi].launch(&e, m_pad, m);
                hipEventRecord(e1, e.stream);
                hipStreamSynchronize(e.stream);
                float msec = 0.f;
                hipEventElapsedTime(&msec, e0, e1);
                if (r > 0) t[vi].push_back(msec / inner);
            }
        for (size_t vi = 0; vi < variants.size(); ++vi) {
            std::sort(t[vi].begin(), t[vi].end());
            const double med = t[vi][t[vi].size() / 2], mn = t[vi][0];
            const double fl = (double)n * n * m;
            printf("time  m=%d %-12s median %.1f us (%.1f TF, %.3f of 157.3)  min %.1f us (%.1f TF)\n", m, variants[vi].name.c_str(),
                   med * 1e3, fl / (med * 1e-3) / 1e12, fl / (med * 1e-3) / 1e12 / 157.3, mn * 1e3, fl / (mn * 1e-3) / 1e12);
        }
        fflush(stdout);
    }
    return 0;
}
