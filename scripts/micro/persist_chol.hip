// Per-panel floor of a PERSISTENT blocked Cholesky sweep: S = L L' of an m x m SPD matrix (fp64, 32-wide
// panels), ONE launch.  The engine's sweep is one launch per panel (k_chol_step: ~9.6 us per panel at m = 1014, of which
// ~4.2 us are the 32x32 factor-and-invert and ~4.5 us kernel boundary + cold loads + publish of the look-ahead workgroup).
// Here the critical chain lives in ONE resident workgroup (block 0):
//     factor + invert A_kk  ->  publish inv(L_kk)  ->  take S(k+1,k), S(k+1,k+1) with panels < k applied from their owners
//     ->  L(k+1,k) = S(k+1,k) inv(L_kk)',  A(k+1,k+1) -= L L'  ->  next k
// and every other tile (i, j) has a fixed owner workgroup that applies panel k to it as soon as inv(L_kk) and the two panel
// blocks S(i,k), S(j,k) are published.  All hand-offs are tile payloads stored write-through (sc1) + one relaxed agent-scope
// flag per tile (panels applied), polled by one lane; loads of handed-off data are sc1 (no L1) -- the R1 recipe of the guide.
// Every spin is bounded (error code instead of a hang); the result is checked against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I openekfmonoslam_amd/csrc scripts/micro/persist_chol.hip -o scripts/micro/persist_chol
//   scripts/micro/persist_chol [m=1024] [workgroups=128] [reps=20] [variant: 0 plain, 1 prefetch, 2 prefetch + late flag]
// Measured (profiles/r03_persist_chol_micro.txt, m = 1024, 255 workgroups = one per CU, S only -- no B / right-hand-side roles):
//   7.3 us per panel = 4.8 us factor-and-invert + write-through publish, + 2.1 us (flags, 16 KB of tiles through L2, two products);
//   the owners of the two tiles the chain needs next take ~5-6 us from the publish of inv(L_kk) to their own publish (poll 1,
//   inverse 1, three tiles 1, products 1, drain + flag 1), i.e. just longer than one factorisation, so the prefetch variant
//   rarely finds them ready and the late flag only delays them (8.1 us).  Against the engine's 9.6 us per launched panel that
//   is -24 % for the chain alone; with the rows of B and the right-hand sides sharing the CUs the hand-offs get slower
//   (guide: 1.5 -> 2.9 us under streaming load), so the integrated gain is estimated at <= 2 us per panel (~85 us per frame at
//   N = 1000).  m = 2048: 13.2 us per panel, bound by the tile owners in the first third (as the launched sweep is).
#include "chol32.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace ekf;
typedef unsigned long long u64;
typedef double acc4 __attribute__((ext_vector_type(4)));
constexpr int NBP = 32;
constexpr unsigned SPIN_LIMIT = 4000000u;

struct Ctl {
    unsigned inv_ready; // panels whose inv(L_kk) is published
    unsigned err;
    unsigned pad[30];
    unsigned done[1]; // [nbk * nbk]: panels applied to tile (i, j) by its owner (and the write is visible)
};

__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double *p, double x)
{
    __hip_atomic_store((u64 *)p, (u64)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every storing wavefront drains its stores, then ONE lane raises the flag
__device__ __forceinline__ void publish(unsigned *flag, unsigned value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane polls (relaxed, with s_sleep), bounded; returns false (uniformly) when the launch is failing
__device__ __forceinline__ bool wait_ge(unsigned *flag, unsigned value, Ctl *ctl, unsigned code)
{
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned n = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
            __builtin_amdgcn_s_sleep(1);
            if ((++n & 1023u) == 0 && __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
            if (n > SPIN_LIMIT) {
                __hip_atomic_store(&ctl->err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    const int r = ok_s;
    __syncthreads();
    return r != 0;
}

// out = A B' (32x32 blocks in LDS, [32][33]); wavefront w owns the 16x16 quadrant (w >> 1, w & 1); result in accumulator layout:
// element v of the lane is row 16 bi + (lane >> 4) + 4 v, column 16 bj + (lane & 15)
__device__ __forceinline__ acc4 prod_abt(const double (*A)[NBP + 1], const double (*B)[NBP + 1])
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, bi = w >> 1, bj = w & 1, lr = lane & 15, lk = lane >> 4;
    acc4 c = {0, 0, 0, 0};
#pragma unroll
    for (int k4 = 0; k4 < NBP; k4 += 4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * bi + lr][k4 + lk], B[16 * bj + lr][k4 + lk], c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ void tile_to_lds(const double *S, int ld, int i, int j, double (*dst)[NBP + 1])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + 256 * q, r = e >> 5, c = e & 31;
        dst[r][c] = ld_sc1(S + (size_t)(NBP * i + r) * ld + NBP * j + c);
    }
}

template <bool PREFETCH, bool LATE_FLAG>
__global__ void __launch_bounds__(256) k_persist_chol(double *S, int ld, int nbk, double *V, double *LL, Ctl *ctl, const int2 *tiles,
                                                       int ntiles, unsigned long long *stamps)
{
    __shared__ double sA[NBP][NBP + 1], sX[NBP][NBP + 1], sP[NBP][NBP + 1], sQ[NBP][NBP + 1], sLi[NBP][NBP + 1], sLj[NBP][NBP + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, bi = w >> 1, bj = w & 1, lr = lane & 15, lk = lane >> 4;
    const int nworkers = gridDim.x - 1;
    unsigned *done = ctl->done;
    if (blockIdx.x == 0) {
        // ------------------------------------------------------------------------------------------ critical chain
        tile_to_lds(S, ld, 0, 0, sA);
        __syncthreads();
        for (int k = 0; k < nbk; ++k) {
            if (stamps && tid == 0) stamps[2 * k] = wall_clock64();
            // PREFETCH: the two tiles the chain needs after this factorisation were handed the previous panel ~one
            // factorisation ago; if their owners are done (one relaxed look, no spinning) their values travel to registers
            // while block k is factorised
            double pf[4], v[4];
            bool have = false;
            if (PREFETCH && k + 1 < nbk) {
                __shared__ int both;
                if (tid == 0)
                    both = __hip_atomic_load(&done[(k + 1) * nbk + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)k &&
                           __hip_atomic_load(&done[(k + 1) * nbk + k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)k;
                __syncthreads();
                have = both != 0;
                if (have) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int e = tid + 256 * q;
                        pf[q] = ld_sc1(S + (size_t)(NBP * (k + 1) + (e >> 5)) * ld + NBP * k + (e & 31));
                        v[q] = ld_sc1(S + (size_t)(NBP * (k + 1) + 16 * bi + lk + 4 * q) * ld + NBP * (k + 1) + 16 * bj + lr);
                    }
                }
            }
            if (!block_chol_inv32_v4(sA, sX) && tid == 0) __hip_atomic_store(&ctl->err, 100u + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + 256 * q;
                st_sc1(V + (size_t)k * NBP * NBP + e, sX[e >> 5][e & 31]);
            }
            if (!LATE_FLAG) publish(&ctl->inv_ready, k + 1);
            if (stamps && tid == 0) stamps[2 * k + 1] = wall_clock64();
            if (k + 1 >= nbk) {
                if (LATE_FLAG) publish(&ctl->inv_ready, k + 1);
                break;
            }
            if (!have) {
                // the next diagonal tile and the block left of it, with panels 0 .. k-1 applied by their owners
                if (!wait_ge(&done[(k + 1) * nbk + k], k, ctl, 1000u + k)) return;
                if (!wait_ge(&done[(k + 1) * nbk + k + 1], k, ctl, 2000u + k)) return;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + 256 * q;
                    pf[q] = ld_sc1(S + (size_t)(NBP * (k + 1) + (e >> 5)) * ld + NBP * k + (e & 31));
                    v[q] = ld_sc1(S + (size_t)(NBP * (k + 1) + 16 * bi + lk + 4 * q) * ld + NBP * (k + 1) + 16 * bj + lr);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + 256 * q;
                sP[e >> 5][e & 31] = pf[q];
            }
            __syncthreads();
            const acc4 l = prod_abt(sP, sX); // L(k+1,k) = S(k+1,k) inv(L_kk)'
#pragma unroll
            for (int q = 0; q < 4; ++q) sLi[16 * bi + lk + 4 * q][16 * bj + lr] = l[q];
            __syncthreads();
            const acc4 t = prod_abt(sLi, sLi);
#pragma unroll
            for (int q = 0; q < 4; ++q) sA[16 * bi + lk + 4 * q][16 * bj + lr] = v[q] - t[q];
            // the flag of inv(L_kk) is raised AFTER the chain's own products: its owners have a whole factorisation of slack
            if (LATE_FLAG) publish(&ctl->inv_ready, k + 1);
            // L(k+1,k) is final: off the chain (nobody waits for it here; the engine's B role would)
#pragma unroll
            for (int q = 0; q < 4; ++q) LL[(size_t)(NBP * (k + 1) + 16 * bi + lk + 4 * q) * ld + NBP * k + 16 * bj + lr] = l[q];
            __syncthreads();
        }
        return;
    }
    // ------------------------------------------------------------------------------------------------- tile owners
    const int me = blockIdx.x - 1;
    for (int k = 0; k + 1 < nbk; ++k) {
        bool have_inv = false;
        for (int t = me; t < ntiles; t += nworkers) {
            const int i = tiles[t].x, j = tiles[t].y; // column-major order: the columns needed soonest come first
            if (j < k || i <= k) continue;
            if (i == k + 1 && j == k + 1) continue; // the critical workgroup applies panel k to the next diagonal tile itself
            if (!have_inv) {
                if (!wait_ge(&ctl->inv_ready, k + 1, ctl, 3000u + k)) return;
                const double *X = V + (size_t)k * NBP * NBP;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + 256 * q;
                    sX[e >> 5][e & 31] = ld_sc1(X + e);
                }
                have_inv = true;
            }
            if (j == k) { // own tile has become a block of panel k (panels < k applied): L(i,k) = S(i,k) inv(L_kk)', final
                if (i == k + 1) continue; // written by the critical workgroup
                tile_to_lds(S, ld, i, k, sP);
                __syncthreads();
                const acc4 l = prod_abt(sP, sX);
#pragma unroll
                for (int q = 0; q < 4; ++q) LL[(size_t)(NBP * i + 16 * bi + lk + 4 * q) * ld + NBP * k + 16 * bj + lr] = l[q];
                __syncthreads();
                continue;
            }
            // trailing tile (i, j), i >= j > k: needs the panel blocks S(i,k), S(j,k) with panels < k applied
            if (!wait_ge(&done[i * nbk + k], k, ctl, 4000u + k)) return;
            if (i != j && !wait_ge(&done[j * nbk + k], k, ctl, 5000u + k)) return;
            tile_to_lds(S, ld, i, k, sP);
            if (i != j) tile_to_lds(S, ld, j, k, sQ);
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ld_sc1(S + (size_t)(NBP * i + 16 * bi + lk + 4 * q) * ld + NBP * j + 16 * bj + lr);
            __syncthreads();
            const acc4 li = prod_abt(sP, sX);
            acc4 lj = li;
            if (i != j) lj = prod_abt(sQ, sX);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sLi[16 * bi + lk + 4 * q][16 * bj + lr] = li[q];
                sLj[16 * bi + lk + 4 * q][16 * bj + lr] = lj[q];
            }
            __syncthreads();
            const acc4 u = prod_abt(sLi, sLj);
#pragma unroll
            for (int q = 0; q < 4; ++q) st_sc1(S + (size_t)(NBP * i + 16 * bi + lk + 4 * q) * ld + NBP * j + 16 * bj + lr, v[q] - u[q]);
            publish(&done[i * nbk + j], k + 1);
        }
    }
}

int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 1024, grid = argc > 2 ? atoi(argv[2]) : 128, reps = argc > 3 ? atoi(argv[3]) : 20, variant = argc > 4 ? atoi(argv[4]) : 2;
    const int nbk = m / NBP, ld = m;
    std::vector<double> M((size_t)m * m), A((size_t)m * m), L((size_t)m * m, 0.0);
    unsigned s = 12345;
    for (auto &v : M) { s = s * 1664525u + 1013904223u; v = ((double)(s >> 8) / (1 << 24) - 0.5) * 0.1; }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
            double t = i == j ? 1.0 : 0.0;
            for (int k = 0; k < 64; ++k) t += M[(size_t)i * m + k] * M[(size_t)j * m + k]; // rank-64 + I: well conditioned, like S = H P H' + R
            A[(size_t)i * m + j] = A[(size_t)j * m + i] = t;
        }
    // host Cholesky
    L = A;
    for (int j = 0; j < m; ++j) {
        double d = L[(size_t)j * m + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * m + k] * L[(size_t)j * m + k];
        d = sqrt(d);
        L[(size_t)j * m + j] = d;
        for (int i = j + 1; i < m; ++i) {
            double t = L[(size_t)i * m + j];
            for (int k = 0; k < j; ++k) t -= L[(size_t)i * m + k] * L[(size_t)j * m + k];
            L[(size_t)i * m + j] = t / d;
        }
    }
    std::vector<int2> tiles;
    for (int j = 0; j < nbk; ++j)
        for (int i = j; i < nbk; ++i) tiles.push_back(make_int2(i, j));
    double *dS, *dS0, *dV, *dLL;
    Ctl *ctl;
    int2 *dT;
    unsigned long long *dStamps;
    const size_t ctl_bytes = sizeof(Ctl) + sizeof(unsigned) * nbk * nbk;
    hipMalloc(&dS, A.size() * 8); hipMalloc(&dS0, A.size() * 8); hipMalloc(&dV, (size_t)nbk * NBP * NBP * 8); hipMalloc(&dLL, A.size() * 8);
    hipMalloc(&ctl, ctl_bytes); hipMalloc(&dT, tiles.size() * sizeof(int2)); hipMalloc(&dStamps, 2 * nbk * 8);
    hipMemcpy(dS0, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dT, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice);
    int per_cu = 0, dev = 0, cus = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_persist_chol<true, true>, 256, 0);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (grid > per_cu * cus) { printf("grid %d exceeds the resident capacity %d x %d\n", grid, per_cu, cus); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> times;
    Ctl hc;
    for (int r = 0; r < reps + 2; ++r) {
        hipMemcpyAsync(dS, dS0, A.size() * 8, hipMemcpyDeviceToDevice, 0);
        hipMemsetAsync(ctl, 0, ctl_bytes, 0);
        hipMemsetAsync(dLL, 0, A.size() * 8, 0);
        hipEventRecord(e0, 0);
        unsigned long long *stp = r == reps + 1 ? dStamps : nullptr;
        if (variant == 0) k_persist_chol<false, false><<<grid, 256, 0, 0>>>(dS, ld, nbk, dV, dLL, ctl, dT, (int)tiles.size(), stp);
        else if (variant == 1) k_persist_chol<true, false><<<grid, 256, 0, 0>>>(dS, ld, nbk, dV, dLL, ctl, dT, (int)tiles.size(), stp);
        else k_persist_chol<true, true><<<grid, 256, 0, 0>>>(dS, ld, nbk, dV, dLL, ctl, dT, (int)tiles.size(), stp);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
        if (hc.err) { printf("rep %d: device error code %u (inv_ready %u)\n", r, hc.err, hc.inv_ready); return 2; }
        if (r >= 2 && r <= reps) times.push_back(ms);
    }
    // check: off-diagonal blocks of L, and inv(L_kk) against the host factor
    std::vector<double> hLL(A.size()), hV((size_t)nbk * NBP * NBP);
    hipMemcpy(hLL.data(), dLL, A.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hV.data(), dV, hV.size() * 8, hipMemcpyDeviceToHost);
    double e_l = 0.0, e_v = 0.0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < (i / NBP) * NBP; ++j) e_l = std::max(e_l, fabs(hLL[(size_t)i * m + j] - L[(size_t)i * m + j]));
    for (int k = 0; k < nbk; ++k)
        for (int r = 0; r < NBP; ++r)
            for (int c = 0; c < NBP; ++c) { // (inv(L_kk) L_kk)[r][c]
                double t = 0.0;
                for (int q = c; q < NBP; ++q) t += hV[(size_t)k * 1024 + r * NBP + q] * L[(size_t)(NBP * k + q) * m + NBP * k + c];
                e_v = std::max(e_v, fabs(t - (r == c ? 1.0 : 0.0)));
            }
    std::sort(times.begin(), times.end());
    const double med = times[times.size() / 2] * 1e3, mn = times[0] * 1e3;
    printf("persistent sweep (variant %d: 0 plain, 1 prefetch, 2 prefetch + late flag) m=%d (%d panels), %d workgroups: median %.1f us = %.2f us/panel, min %.1f us = %.2f us/panel;  max|L - L_host| %.2e, max|inv(Lkk) Lkk - I| %.2e %s\n",
           variant, m, nbk, grid, med, med / nbk, mn, mn / nbk, e_l, e_v, (e_l < 1e-10 && e_v < 1e-10) ? "OK" : "FAIL");
    std::vector<unsigned long long> st(2 * nbk);
    hipMemcpy(st.data(), dStamps, st.size() * 8, hipMemcpyDeviceToHost);
    printf("critical workgroup, 10 ns ticks: panel start -> published, published -> next start (wait for the owners + own update)\n");
    for (int k = 0; k + 1 < nbk; k += std::max(1, nbk / 8))
        printf("  k=%2d factor+publish %.2f us, hand-off + update %.2f us\n", k, (st[2 * k + 1] - st[2 * k]) * 0.01, (st[2 * k + 2] - st[2 * k + 1]) * 0.01);
    return 0;
}
