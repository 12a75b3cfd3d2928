// How long is a chain of dependent small launches on this GPU: stream launches against one hipGraph of the same kernel nodes
// (is the kernel boundary of the Cholesky sweep cheaper inside a graph?).  The kernel mimics a sweep launch: 256 workgroups, one of
// which spins ~6 us (the look-ahead factorisation), the others return at once.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/launch_gap.hip -o scripts/micro/launch_gap && scripts/micro/launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_step(unsigned long long *sink, int spin_ticks)
{
    if (blockIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)spin_ticks) {}
        if (threadIdx.x == 0) sink[0] += 1;
    }
}

int main()
{
    unsigned long long *sink;
    hipMalloc(&sink, 64);
    hipMemset(sink, 0, 64);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int L = 32;
    for (int spin : {0, 300, 600}) { // 100 MHz ticks: 0, 3, 6 us
        std::vector<float> ts, tg;
        for (int rep = 0; rep < 12; ++rep) {
            hipEventRecord(e0, s);
            for (int i = 0; i < L; ++i) k_step<<<256, 256, 0, s>>>(sink, spin);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ts.push_back(ms * 1e3f / L);
        }
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < L; ++i) k_step<<<256, 256, 0, s>>>(sink, spin);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int rep = 0; rep < 12; ++rep) {
            hipEventRecord(e0, s);
            hipGraphLaunch(ge, s);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            tg.push_back(ms * 1e3f / L);
        }
        std::sort(ts.begin(), ts.end());
        std::sort(tg.begin(), tg.end());
        printf("spin %.0f us: %d dependent launches: stream %.2f us per launch (median), graph %.2f us per launch\n", spin / 100.0, L, ts[ts.size() / 2], tg[tg.size() / 2]);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    return 0;
}
