// kernels_pupdate.hip -- the covariance downdate  P <- sym(P) - B' B  (B = inv(L) H P, m x n, k-major).
//
// Replaces covarianceUpdate + symmetrisation (EKF/Update.cpp:214-218, 303): the reference multiplies the dense
// n x n (I - K H) by P (2 n^3 + 2 n^2 m flops); (I - K H) P = P - (P H') inv(S) (H P) = P - B' B, a rank-m
// symmetric downdate: n^2 m flops on the upper triangle, 2 n^2 w bytes of P traffic.  At m = 2000 that is
// ~500 flop/B, so the kernel is MFMA-bound (fp32: v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s peak; fp64:
// v_mfma_f64_16x16x4_f64).
//
// Layout / tiling (gfx950):
//   * B is stored k-major (row k = all n columns), so BOTH MFMA operands of a tile (I, J) are row slabs
//     B[k0:k0+BK, I] and B[k0:k0+BK, J]: coalesced 16-byte global loads, LDS image [k][column], and the
//     operand fetch "lane l reads [k + l / MB][c + l % MB]" is a conflict-free ds_read (consecutive lanes,
//     consecutive banks).
//   * workgroup = 256 threads = 2 x 2 wavefronts, each wavefront owns 2 x 2 MFMA blocks (fp32: 64 x 64 outputs,
//     64 accumulator VGPRs), workgroup tile TM x TM with TM = 4 MB (fp32 128, fp64 64), k-slab BK = 16,
//     double buffering through LDS (32 KB): fp32 register-staged, fp64 by LDS-DMA (PU_LDS_DMA / PU_LDS_DMA_F64).
//   * only tiles with I <= J are launched; the epilogue subtracts from P, writes the tile and its mirror image,
//     so P stays bitwise symmetric.
#include "engine.h"
#include "mma_tile.h"

#include <algorithm>

namespace ekf {

#ifndef PU_LDS_DMA
#define PU_LDS_DMA 0 // 1: slabs of B by LDS-DMA instead of staging registers (same results; measured, not faster: DESIGN 4.1)
#endif
#ifndef PU_LDS_DMA_F64
#define PU_LDS_DMA_F64 1 // the fp64 instance needs 86 VGPRs either way; without the staging stores it is 2 % faster at N = 1000
#endif
#ifndef PU_F32_PAIRS
#define PU_F32_PAIRS 1 // fp32: k_p_update_f32 (operand pairs by 8-byte LDS reads); 0: the generic kernel below
#endif
#ifndef PU_MIN_WAVES
#define PU_MIN_WAVES 3 // with PU_LDS_DMA 4 fits (128 VGPRs, 4 x 32 KB LDS): +2.4 % at N = 1000, -3 % at N = 2000 / 5000
#endif
// RECT (row-sharded storage): the rank owns row tiles, not a triangle.  Tile row 0 is the replicated
// camera block (13 live rows), tile row t >= 1 holds the owned global rows rm.r0 + (t-1) TM ..., stored from local
// row rm.base + (t-1) TM; every (row tile, column tile) pair is computed and written in place, nothing is mirrored.
// Swapping the operands of an MFMA product changes no bit, so P[i][j] here equals P[j][i] on the rank that owns j.
template <typename T, bool AVG, bool RECT>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? PU_MIN_WAVES : 3)
k_p_update(T *P, int ldp, int n, const T *B, int ldb, int m_pad, int per_xcd, const int4 *units, RowMap rm, int stagger, const int *counts)
{
    if (filter_frozen(counts)) return; // the update's sweep failed: P stays as it was (engine.h)
    using M = Mma<T>;
    constexpr int MB = M::MB, TM = 4 * MB, KI = 64 / MB, VEC = M::VEC;
    constexpr int LOADS = PU_BK * TM / (256 * VEC);
    // one LDS block: [I-slab x2 | J-slab x2], re-used by the epilogue as per-wavefront transpose scratch
    __shared__ __attribute__((aligned(16))) T smem[4 * PU_BK * TM];
    T(*sI)[PU_BK][TM] = reinterpret_cast<T(*)[PU_BK][TM]>(smem);
    T(*sJ)[PU_BK][TM] = reinterpret_cast<T(*)[PU_BK][TM]>(smem + 2 * PU_BK * TM);
    static_assert(4 * MB * (MB + 1) <= 4 * PU_BK * TM, "transpose scratch must fit");

    // XCD-aware work order: workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on it).  Each
    // XCD walks its own list: a contiguous chunk of a super-tiled (8 x 8 tiles) enumeration of the upper triangle, so
    // the workgroups resident on one XCD at a time share their B row-slabs through that XCD's L2, followed by the
    // HALF units (64 of the 128 tile rows) into which the last tiles are split so that the tail of the launch
    // spreads over all CUs instead of leaving most of them idle for a whole tile time.
    const int4 unit = units[(size_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)];
    if (unit.x < 0) return;
    if (stagger > 0) {
        // Short k-loops (m < 512): a tile is mostly its 64 KB read-modify-write of P, and the three workgroups resident on a
        // CU would hit that HBM-bound phase together.  The second and third workgroup of the first round start a fraction
        // of a tile time late (s_sleep costs no MFMA slots; the others keep the pipe busy): 0.146 -> 0.125 ms at m = 320.
        // For long k-loops the half-units-first order does the de-phasing and this is off (it costs 3 % there).
        const int slot = (blockIdx.x >> 3) / 32; // 32 CUs per XCD: the position this workgroup takes on its CU
        if (slot == 1 || slot == 2) {
            const int n_sleep = slot * m_pad / stagger;
            for (int i = 0; i < n_sleep; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
    const int ti = unit.x, tj = unit.y;
    const bool full = unit.z < 0;
    const bool diag = RECT || (ti == tj);
    // I0: first global row of the tile (= column of B for the row operand); p_off: local minus global row
    const int I0 = RECT ? (ti == 0 ? 0 : rm.r0 + (ti - 1) * TM) : ti * TM, J0 = tj * TM;
    const int p_off = RECT ? (ti == 0 ? 0 : rm.base - rm.r0) : 0;
    const int ilim = RECT ? (ti == 0 ? 13 : rm.r1) : n;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    // row offset of this wavefront inside the tile: full unit 64 rows per wavefront row, half unit 32
    const int rbase = full ? wr * 2 * MB : unit.z * 2 * MB + wr * MB;
    const int klane = lane / MB, idx = lane % MB;

    typename M::acc_t c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) c00[r] = c01[r] = c10[r] = c11[r] = (T)0;

    using V = typename M::vec_t;
    static_assert(LOADS == 2 || LOADS == 4, "2 or 4 16-byte pieces per thread and slab");
    const int nk = m_pad / PU_BK;
    const size_t slab = (size_t)PU_BK * ldb;
    // Per-thread pieces of a slab, in named registers (arrays here end up in scratch: hipcc cannot promote them).
#define PU_PIECE(q) const int lk##q = ((tid + q * 256) * VEC) / TM, lc##q = ((tid + q * 256) * VEC) % TM; \
                    const T *gI##q = B + (size_t)lk##q * ldb + I0 + lc##q; const T *gJ##q = B + (size_t)lk##q * ldb + J0 + lc##q; \
                    V rI##q = *(const V *)gI##q, rJ##q = *(const V *)gJ##q;
    if constexpr (PU_LDS_DMA || (PU_LDS_DMA_F64 && sizeof(T) == 8)) {
    // Slabs travel from global memory straight into LDS (global_load_lds_dwordx4: lane l of a wavefront writes 16 bytes at
    // the wavefront's LDS base + 16 l, which is exactly this kernel's [k][column] image: a wavefront's 64 pieces of a slab
    // are 1 KB of consecutive LDS).  No staging registers, no ds_write; completion is counted by vmcnt.
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
#define PU_PIECE_DMA(q) const int lk##q = ((tid + q * 256) * VEC) / TM, lc##q = ((tid + q * 256) * VEC) % TM; \
                    const T *gI##q = B + (size_t)lk##q * ldb + I0 + lc##q; const T *gJ##q = B + (size_t)lk##q * ldb + J0 + lc##q;
    PU_PIECE_DMA(0)
    PU_PIECE_DMA(1)
    PU_PIECE_DMA(2)
    PU_PIECE_DMA(3)
#undef PU_PIECE_DMA
    const int wbase = __builtin_amdgcn_readfirstlane(wv * 64 * VEC); // first element of this wavefront's pieces in a slab image
#define PU_DMA(q, off, b)                                                                                               \
    __builtin_amdgcn_global_load_lds((gptr_t)(gI##q + (off)), (lptr_t)(&sI[b][0][0] + wbase + q * 256 * VEC), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((gptr_t)(gJ##q + (off)), (lptr_t)(&sJ[b][0][0] + wbase + q * 256 * VEC), 16, 0, 0);
    PU_DMA(0, (size_t)0, 0)
    PU_DMA(1, (size_t)0, 0)
    if (LOADS == 4) {
        PU_DMA(2, (size_t)0, 0)
        PU_DMA(3, (size_t)0, 0)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) { // the next slab lands in the other buffer while this one is consumed (every wavefront left it at the barrier)
            const size_t off = (size_t)(kt + 1) * slab;
            PU_DMA(0, off, buf ^ 1)
            PU_DMA(1, off, buf ^ 1)
            if (LOADS == 4) {
                PU_DMA(2, off, buf ^ 1)
                PU_DMA(3, off, buf ^ 1)
            }
        }
        if (full) pu_slab<T, true, TM>(sI[buf], sJ[buf], klane, rbase + idx, wc * 2 * MB + idx, c00, c01, c10, c11);
        else pu_slab<T, false, TM>(sI[buf], sJ[buf], klane, rbase + idx, wc * 2 * MB + idx, c00, c01, c10, c11);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef PU_DMA
    } else {
    PU_PIECE(0)
    PU_PIECE(1)
    PU_PIECE(2)
    PU_PIECE(3)
#undef PU_PIECE
#define PU_STORE(q, b) *(V *)(&sI[b][lk##q][lc##q]) = rI##q; *(V *)(&sJ[b][lk##q][lc##q]) = rJ##q;
#define PU_LOAD(q, off) rI##q = *(const V *)(gI##q + off); rJ##q = *(const V *)(gJ##q + off);
    PU_STORE(0, 0)
    PU_STORE(1, 0)
    if (LOADS == 4) {
        PU_STORE(2, 0)
        PU_STORE(3, 0)
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) { // prefetch the next slab into registers while this one is consumed from LDS
            const size_t off = (size_t)(kt + 1) * slab;
            PU_LOAD(0, off)
            PU_LOAD(1, off)
            if (LOADS == 4) {
                PU_LOAD(2, off)
                PU_LOAD(3, off)
            }
        }
        if (full) pu_slab<T, true, TM>(sI[buf], sJ[buf], klane, rbase + idx, wc * 2 * MB + idx, c00, c01, c10, c11);
        else pu_slab<T, false, TM>(sI[buf], sJ[buf], klane, rbase + idx, wc * 2 * MB + idx, c00, c01, c10, c11);
        if (more) {
            PU_STORE(0, buf ^ 1)
            PU_STORE(1, buf ^ 1)
            if (LOADS == 4) {
                PU_STORE(2, buf ^ 1)
                PU_STORE(3, buf ^ 1)
            }
        }
        __syncthreads();
    }
#undef PU_STORE
#undef PU_LOAD
    }

    // epilogue.  P is bitwise symmetric on entry (engine invariant) unless AVG.
    //  - diagonal tiles: every element (i, j) of the tile is computed (acc is bitwise symmetric), written in place;
    //  - off-diagonal tiles: the tile is written with row-contiguous stores, and its mirror image through a
    //    per-wavefront LDS transpose so that the mirrored stores are row-contiguous too (full 128-byte segments
    //    instead of scattered 4-byte stores).
    //  - AVG (first update after an arbitrary upload): P(i,j) <- 0.5 (P(i,j) + P(j,i)) - acc on i <= j, mirrored.
    T *sT = smem + wv * MB * (MB + 1);
    // the P values of BOTH blocks of a block row are requested before the first one is needed: the epilogue of a tile
    // is a 64 KB read-modify-write whose latency was exposed once per MFMA block
    typename M::acc_t pv[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            if (x == 1 && !full) continue;
            if (!AVG && y == 0) {
#pragma unroll
                for (int yy = 0; yy < 2; ++yy) {
                    const int pbi = I0 + rbase + x * MB, pbj = J0 + wc * 2 * MB + yy * MB;
#pragma unroll
                    for (int r = 0; r < M::NACC; ++r) {
                        const int gi = pbi + M::row(r, lane), gj = pbj + M::col(lane);
                        pv[yy][r] = (gi < ilim && gj < n) ? P[(size_t)(gi + p_off) * ldp + gj] : (T)0;
                    }
                }
            }
            const int bi = I0 + rbase + x * MB, bj = J0 + wc * 2 * MB + y * MB;
            const typename M::acc_t &cc = x == 0 ? (y == 0 ? c00 : c01) : (y == 0 ? c10 : c11);
            if (AVG) {
#pragma unroll
                for (int r = 0; r < M::NACC; ++r) {
                    const int gi = bi + M::row(r, lane), gj = bj + M::col(lane);
                    if (!RECT && gi < n && gj < n && gi <= gj) {
                        T *pu = P + (size_t)gi * ldp + gj;
                        T *pl = P + (size_t)gj * ldp + gi;
                        const T v = ((T)0.5 * (*pu) + (T)0.5 * (*pl)) - cc[r];
                        *pu = v;
                        *pl = v;
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) {
                const int li = M::row(r, lane), lj = M::col(lane);
                const int gi = bi + li, gj = bj + lj;
                T v = (T)0;
                if (gi < ilim && gj < n) {
                    v = pv[y][r] - cc[r];
                    P[(size_t)(gi + p_off) * ldp + gj] = v;
                }
                if (!diag) sT[li * (MB + 1) + lj] = v;
            }
            if (!diag) {
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < MB / KI; ++it) {
                    const int c = it * KI + klane; // column of the block = row of the mirror
                    const int gi = bi + idx, gj = bj + c;
                    const T v = sT[idx * (MB + 1) + c];
                    if (gi < n && gj < n) P[(size_t)gj * ldp + gi] = v;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
}

// ------------------------------------------------------------------------------------------------ fp32 instance
// The same tiles, units and slab pipeline as k_p_update (register-staged, double-buffered [k][column] slabs, one barrier per
// slab), with the operands of TWO MFMA blocks fetched by ONE 8-byte LDS read.  On this chip the return traffic of the
// operand reads takes cycles from the matrix pipe: with one 4-byte read per operand and k-step (32 per wavefront and slab) the
// pure loop at three workgroups per CU runs 292 us at m = 1014, with 16-byte reads 266 us, with none 261 us
// (scripts/micro/pu_bench.hip, profiles/r03_pu_bench.txt).  Lane (klane, idx) therefore reads the PAIR of adjacent columns
// 2 idx, 2 idx + 1 of its wavefront's 64 rows (and of its 64 columns) at k-row k + klane: block (ea, eb) of the wavefront's
// 2 x 2 MFMA blocks then holds the outputs (row 2 i + ea, column 2 j + eb) instead of a contiguous 32 x 32 square.  Nothing
// else moves: B stays k-major in HBM, slabs are copied as before, and the epilogue's accesses become 8 bytes wide (the two
// column blocks of a register are adjacent columns), which also halves its load / store instruction count.
#ifdef PU_STAMPS
__device__ unsigned long long *g_pu_stamps = nullptr;
#endif

template <bool AVG, bool RECT>
__global__ void __launch_bounds__(256, 3)
k_p_update_f32(float *P, int ldp, int n, const float *B, int ldb, int m_k, int per_xcd, const int4 *units, RowMap rm, int stagger)
{
    using M = Mma<float>;
    constexpr int MB = 32, TM = 128, BK = PU_BK, VEC = 4;
    constexpr int LOADS = BK * TM / (256 * VEC);
    constexpr int TS = 2 * MB + 2;         // transpose scratch: [32 column pairs][64 rows + 2]
    static_assert(LOADS == 2, "two 16-byte pieces per thread, operand and slab");
    // [I-slab x2 | J-slab x2]; re-used by the epilogue as per-wavefront transpose scratch (32 x 66 floats each)
    __shared__ __attribute__((aligned(16))) float smem[(4 * BK * TM > 4 * MB * TS) ? 4 * BK * TM : 4 * MB * TS];
    float(*sI)[BK][TM] = reinterpret_cast<float(*)[BK][TM]>(smem);
    float(*sJ)[BK][TM] = reinterpret_cast<float(*)[BK][TM]>(smem + 2 * BK * TM);

    const int4 unit = units[(size_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)]; // XCD-aware work order, see k_p_update
    if (unit.x < 0) return;
#ifdef PU_STAMPS // scripts/micro/pu_bench.hip only: when a unit starts, leaves its k-loop and ends (100 MHz clock)
    if (g_pu_stamps && threadIdx.x == 0) g_pu_stamps[4 * blockIdx.x] = wall_clock64();
#endif
    if (stagger > 0) { // short k-loops: de-phase the three workgroups of a CU, see k_p_update
        const int slot = (blockIdx.x >> 3) / 32;
        if (slot == 1 || slot == 2) {
            const int n_sleep = slot * m_k / stagger;
            for (int i = 0; i < n_sleep; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
    const int ti = unit.x, tj = unit.y;
    const bool full = unit.z < 0;
    const bool diag = RECT || (ti == tj);
    const int I0 = RECT ? (ti == 0 ? 0 : rm.r0 + (ti - 1) * TM) : ti * TM, J0 = tj * TM;
    const int p_off = RECT ? (ti == 0 ? 0 : rm.base - rm.r0) : 0;
    const int ilim = RECT ? (ti == 0 ? 13 : rm.r1) : n;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int rbase = full ? wr * 2 * MB : unit.z * 2 * MB + wr * MB; // first tile row of the wavefront (64 or 32 rows)
    const int cbase = wc * 2 * MB;
    const int klane = lane >> 5, idx = lane & 31;

    // c<ea><eb>: rows rbase + 2 i + ea (whole unit) or rbase + i (half unit, ea = 0 only), columns cbase + 2 j + eb
    typename M::acc_t c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) c00[r] = c01[r] = c10[r] = c11[r] = 0.f;

    const int nk = m_k / BK;
    const size_t slab = (size_t)BK * ldb;
#define PUF_PIECE(q) const int lk##q = ((tid + q * 256) * VEC) / TM, lc##q = ((tid + q * 256) * VEC) % TM; \
                     const float *gI##q = B + (size_t)lk##q * ldb + I0 + lc##q; const float *gJ##q = B + (size_t)lk##q * ldb + J0 + lc##q; \
                     float4 rI##q = *(const float4 *)gI##q, rJ##q = *(const float4 *)gJ##q;
    PUF_PIECE(0)
    PUF_PIECE(1)
#undef PUF_PIECE
#define PUF_STORE(q, b) *(float4 *)(&sI[b][lk##q][lc##q]) = rI##q; *(float4 *)(&sJ[b][lk##q][lc##q]) = rJ##q;
#define PUF_LOAD(q, off) rI##q = *(const float4 *)(gI##q + off); rJ##q = *(const float4 *)(gJ##q + off);
    PUF_STORE(0, 0)
    PUF_STORE(1, 0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) { // the next slab travels to registers while this one is multiplied
            const size_t off = (size_t)(kt + 1) * slab;
            PUF_LOAD(0, off)
            PUF_LOAD(1, off)
        }
        // operands of k-step kk + 2 are requested before the MFMAs of k-step kk are issued (a wavefront issues in order).
        // Issuing all sixteen reads of a slab before its first MFMA instead was measured slower (350 against 332 us at
        // m = 1014): every wavefront of the workgroup then waits out the LDS latency right after the barrier.
        if (full) {
            const float *pa = &sI[buf][klane][rbase + 2 * idx], *pb = &sJ[buf][klane][cbase + 2 * idx];
            float2 a = *(const float2 *)pa, b = *(const float2 *)pb;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float2 na = a, nb = b;
                if (kk + 2 < BK) {
                    na = *(const float2 *)(pa + (kk + 2) * TM);
                    nb = *(const float2 *)(pb + (kk + 2) * TM);
                }
                __builtin_amdgcn_sched_barrier(0);
                c00 = M::mma(a.x, b.x, c00);
                c01 = M::mma(a.x, b.y, c01);
                c10 = M::mma(a.y, b.x, c10);
                c11 = M::mma(a.y, b.y, c11);
                __builtin_amdgcn_sched_barrier(0);
                a = na;
                b = nb;
            }
        } else {
            const float *pa = &sI[buf][klane][rbase + idx], *pb = &sJ[buf][klane][cbase + 2 * idx];
            float a = *pa;
            float2 b = *(const float2 *)pb;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float na = a;
                float2 nb = b;
                if (kk + 2 < BK) {
                    na = pa[(kk + 2) * TM];
                    nb = *(const float2 *)(pb + (kk + 2) * TM);
                }
                __builtin_amdgcn_sched_barrier(0);
                c00 = M::mma(a, b.x, c00);
                c01 = M::mma(a, b.y, c01);
                __builtin_amdgcn_sched_barrier(0);
                a = na;
                b = nb;
            }
        }
        if (more) {
            PUF_STORE(0, buf ^ 1)
            PUF_STORE(1, buf ^ 1)
        }
        __syncthreads();
    }
#undef PUF_STORE
#undef PUF_LOAD

#ifdef PU_STAMPS
    if (g_pu_stamps && threadIdx.x == 0) {
        g_pu_stamps[4 * blockIdx.x + 1] = wall_clock64();
        g_pu_stamps[4 * blockIdx.x + 3] = (unsigned long long)(full ? 1 : 0) | ((unsigned long long)(diag ? 1 : 0) << 1);
    }
#endif
#ifdef PU_F32_ABL // timing ablations of scripts/micro/pu_bench.hip only (wrong results): 1 no epilogue, 4 direct part only, 8 mirror part only
    if (PU_F32_ABL & 1) {
        if (c00[0] + c01[1] + c10[2] + c11[3] == 12345.f) P[0] = 0.f;
        return;
    }
#endif
    // epilogue.  P is bitwise symmetric on entry (engine invariant) unless AVG; every (row, column pair) of a register is one
    // 8-byte access.  Off-diagonal tiles also write the mirror image: the two row blocks of one column parity go through
    // a per-wavefront LDS transpose ([column pair][row]) and leave as 8-byte stores of adjacent mirror columns.
    const int gjp = J0 + cbase + 2 * M::col(lane); // first column of this lane's pair
    if (AVG) { // first update after an arbitrary upload: P(i,j) <- 0.5 (P(i,j) + P(j,i)) - acc on i <= j, mirrored
#pragma unroll
        for (int ea = 0; ea < 2; ++ea)
#pragma unroll
            for (int eb = 0; eb < 2; ++eb) {
                if (ea == 1 && !full) continue;
                const typename M::acc_t &cc = ea == 0 ? (eb == 0 ? c00 : c01) : (eb == 0 ? c10 : c11);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gi = I0 + rbase + (full ? 2 * M::row(r, lane) + ea : M::row(r, lane)), gj = gjp + eb;
                    if (!RECT && gi < n && gj < n && gi <= gj) {
                        float *pu = P + (size_t)gi * ldp + gj, *pl = P + (size_t)gj * ldp + gi;
                        const float v = (0.5f * (*pu) + 0.5f * (*pl)) - cc[r];
                        *pu = v;
                        *pl = v;
                    }
                }
            }
        return;
    }
    float *sT = smem + wv * MB * TS;
#ifdef PU_F32_ABL
    if (!(PU_F32_ABL & 8))
#endif
#pragma unroll
    for (int ea = 0; ea < 2; ++ea) {
        if (ea == 1 && !full) continue;
        typename M::acc_t &ca = ea == 0 ? c00 : c10, &cb = ea == 0 ? c01 : c11;
        // All sixteen row values are requested before the first is used.  Measured and not kept: requesting the first row
        // block's values before the k-loop (they do not depend on it) costs 32 live registers, the kernel spills: 133 against
        // 113 us at m = 298; an instance for four workgroups per CU (128 registers, 108 bytes of scratch): 163 against 115 us.
        constexpr int RH = 16;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += RH) {
            float2 pv[RH];
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                const int gi = I0 + rbase + (full ? 2 * M::row(r0 + r, lane) + ea : M::row(r0 + r, lane));
                pv[r] = (gi < ilim && gjp < n) ? *(const float2 *)(P + (size_t)(gi + p_off) * ldp + gjp) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int r = 0; r < RH; ++r) {
                const int gi = I0 + rbase + (full ? 2 * M::row(r0 + r, lane) + ea : M::row(r0 + r, lane));
                ca[r0 + r] = pv[r].x - ca[r0 + r]; // the accumulators now hold the new values of P (the mirror pass reads them)
                cb[r0 + r] = pv[r].y - cb[r0 + r];
                if (gi < ilim) {
                    float *dst = P + (size_t)(gi + p_off) * ldp + gjp;
                    if (gjp + 1 < n) *(float2 *)dst = make_float2(ca[r0 + r], cb[r0 + r]);
                    else if (gjp < n) *dst = ca[r0 + r]; // n is odd: the last column has no partner (the padding stays untouched)
                }
            }
        }
    }
#ifdef PU_STAMPS
    if (diag) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (g_pu_stamps && threadIdx.x == 0) g_pu_stamps[4 * blockIdx.x + 2] = wall_clock64();
    }
#endif
    if (diag) return;
#ifdef PU_F32_ABL
    if (PU_F32_ABL & 4) return;
#endif
    // mirror image: P[column][row].  Rows of an off-diagonal tile are all < n (its row range ends before its column range starts).
    const int ni = full ? 2 * MB : MB; // rows of the wavefront
#pragma unroll
    for (int eb = 0; eb < 2; ++eb) {
        const typename M::acc_t &c0 = eb == 0 ? c00 : c01, &c1 = eb == 0 ? c10 : c11;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int li = M::row(r, lane), lj = M::col(lane);
            if (full) {
                sT[lj * TS + 2 * li] = c0[r];
                sT[lj * TS + 2 * li + 1] = c1[r];
            } else {
                sT[lj * TS + li] = c0[r];
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        const int hp = ni / 2;                 // row pairs per mirror row: 32 or 16
        const int ip = lane % hp, cq = lane / hp, cstep = 64 / hp;
#pragma unroll
        for (int it = 0; it < MB / 2; ++it) {
            if (it * cstep >= MB) break;
            const int c = it * cstep + cq;     // column pair of the wavefront = mirror row 2 c + eb
            const int gj = J0 + cbase + 2 * c + eb, gi = I0 + rbase + 2 * ip;
            const float2 v = *(const float2 *)(sT + c * TS + 2 * ip);
            if (gj < n) *(float2 *)(P + (size_t)gj * ldp + gi) = v;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
#ifdef PU_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the unit ends when its stores have left the wavefront
    __syncthreads();
    if (g_pu_stamps && threadIdx.x == 0) g_pu_stamps[4 * blockIdx.x + 2] = wall_clock64();
#endif
}

// host-built work list.  Tiles: super-tiles of 8 x 8 tiles, row-major inside; upper triangle only (whole matrix on
// one GPU) or all nrt x nt tiles of the owned row tiles (RECT).  The tail (ntiles mod #CUs tiles, i.e. what would
// occupy only part of the chip for a whole tile time) is split into half units.
#ifdef PU_BENCH // tuning overrides of scripts/micro/pu_bench.hip: not in the product library (process-global, not thread-safe)
int g_pu_order_override = -1;
#else
constexpr int g_pu_order_override = -1;
#endif

// order: 0 half units last, 1 half units first, 2 mixed first round (see below)
void build_units(EkfEngine *e, int nt, int nrt, bool rect, int order)
{
    if (g_pu_order_override >= 0) order = g_pu_order_override;
    const long long key = ((long long)(rect ? -(nt * 4096 + nrt) : nt) * 4 + order) * 4096 + e->pu_slots;
    if (e->pu_tilemap_nt == key && e->d.pu_tilemap) return;
    {   // the two orders of one geometry alternate every frame (LI / HI update): keep every list once built
        auto it = e->pu_tables.find(key);
        if (it != e->pu_tables.end()) {
            e->d.pu_tilemap = it->second.first;
            e->pu_per_xcd = it->second.second;
            e->pu_tilemap_nt = key;
            return;
        }
    }
    std::vector<int4> tiles;
    const int ST = 8, NCU = 256, NX = 8;
    if (rect) {
        for (int si = 0; si < nrt; si += ST)
            for (int sj = 0; sj < nt; sj += ST)
                for (int i = si; i < si + ST && i < nrt; ++i)
                    for (int j = sj; j < sj + ST && j < nt; ++j) tiles.push_back(make_int4(i, j, -1, 0));
    } else {
        for (int si = 0; si < nt; si += ST)
            for (int sj = si; sj < nt; sj += ST)
                for (int i = si; i < si + ST && i < nt; ++i)
                    for (int j = (sj > i ? sj : i); j < sj + ST && j < nt; ++j) tiles.push_back(make_int4(i, j, -1, 0));
    }
    const int ntiles = (int)tiles.size();
    // How many tiles stay whole.  The launch is one workgroup per unit and the device keeps `slots` of them resident
    // (workgroups per CU x CUs); a freed slot takes the next unit of its XCD's list.  Whole tiles take the same time T and a
    // half unit T / 2, so the launch is `rounds` half-rounds long when every slot is dealt the same number of half units:
    // floor(rounds / 2) whole tiles and, for an odd count, one half.  At N = 1000 (1128 tiles, 768 slots): 752 slots x (one
    // whole tile + one half).  The rule of rounds 1-2 (whole tiles in multiples of the CU count: 1024 + 208 halves) left a
    // third round of whole tiles on a third of the CUs: the MFMA stream alone took 318 us where the balanced list takes 260
    // (scripts/micro/pu_bench.hip, profiles/r03_pu_bench.txt).
    int n_full;
    if (e->pu_slots > 0) {
        const int rounds = (2 * ntiles + e->pu_slots - 1) / e->pu_slots; // half-rounds of the whole launch
        const int used = (2 * ntiles + rounds - 1) / rounds;              // slots that get `rounds` half units
        n_full = rounds <= 1 ? 0 : std::min(ntiles, (rounds / 2) * used);
    } else {
        n_full = ntiles >= NCU ? (ntiles / NCU) * NCU : 0;
    }
    std::vector<int4> halves;
    for (int t = n_full; t < ntiles; ++t) {
        halves.push_back(make_int4(tiles[t].x, tiles[t].y, 0, 0));
        halves.push_back(make_int4(tiles[t].x, tiles[t].y, 1, 0));
    }
    const int fchunk = (n_full + NX - 1) / NX, hchunk = ((int)halves.size() + NX - 1) / NX;
    const int per = fchunk + hchunk;
    std::vector<int4> table((size_t)NX * per, make_int4(-1, -1, -1, 0));
    for (int x = 0; x < NX; ++x) {
        // order 0 (what launch_p_update asks for): whole tiles first, half units last.  Orders 1-3 are kept for the bench
        // (round 2 ran long k-loops with the half units first; against a balanced list that is the slower order).
        std::vector<int4> hl, fl, out;
        for (int k = 0; k < hchunk && x * hchunk + k < (int)halves.size(); ++k) hl.push_back(halves[x * hchunk + k]);
        for (int k = 0; k < fchunk && x * fchunk + k < n_full; ++k) fl.push_back(tiles[x * fchunk + k]);
        size_t ih = 0, jf = 0;
        if (order == 2 || order == 3) {
            // mixed: the slots of the first round alternate between a half unit and a whole tile, so that with a balanced
            // list (every slot: one of each) half of the slots run [half, tile] and the others [tile, half]: the epilogues of
            // the first units fall at T/3 and 2T/3 under the other slots' MFMA phases, and the launch ends with a mix of
            // tile and half-unit epilogues instead of every slot's tile epilogue at once
            const int first = e->pu_slots > 0 ? e->pu_slots / NX : 0;
            for (int i = 0; i < first && (ih < hl.size() || jf < fl.size()); ++i) {
                const bool want_h = (((i / 32) & 1) == 0) == (order == 2); // 32 CUs per XCD, handed one workgroup each in turn
                if ((want_h && ih < hl.size()) || jf >= fl.size()) out.push_back(hl[ih++]);
                else out.push_back(fl[jf++]);
            }
            while (jf < fl.size()) out.push_back(fl[jf++]);
            while (ih < hl.size()) out.push_back(hl[ih++]);
        } else if (order == 1) {
            out = hl;
            out.insert(out.end(), fl.begin(), fl.end());
        } else {
            out = fl;
            out.insert(out.end(), hl.begin(), hl.end());
        }
        for (size_t k = 0; k < out.size(); ++k) table[(size_t)x * per + k] = out[k];
    }
    e->d.pu_tilemap = nullptr;
    if (hipMalloc((void **)&e->d.pu_tilemap, table.size() * sizeof(int4)) != hipSuccess ||
        hipMemcpyAsync(e->d.pu_tilemap, table.data(), table.size() * sizeof(int4), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) {
        // no work list: the caller's launch is skipped (grid 0) and the update reports the failure instead of downdating garbage
        if (e->d.pu_tilemap) (void)hipFree(e->d.pu_tilemap);
        e->d.pu_tilemap = nullptr;
        e->pu_per_xcd = 0;
        e->pu_tilemap_nt = -1;
        e->err = "work list of the covariance downdate: allocation or upload failed";
        e->hook_rc = EKF_ERR_HIP;
        return;
    }
    e->pu_tilemap_nt = key;
    e->pu_per_xcd = per;
    e->pu_tables[key] = std::make_pair(e->d.pu_tilemap, per);
}

#ifdef PU_BENCH
int g_pu_stagger_override = -1;
int g_pu_force_slots = 0;       // the slot count the unit list is balanced against
#else
constexpr int g_pu_stagger_override = -1, g_pu_force_slots = 0;
#endif

template <typename T>
static void launch_p_update_t(EkfEngine *e, int m_pad, int grid, const int4 *tm, bool avg, bool rect)
{
    hipStream_t s = e->stream;
    T *P = (T *)e->d.P;
    const T *B = (const T *)e->d.A;
    const int stagger = g_pu_stagger_override >= 0 ? g_pu_stagger_override : ((m_pad < 512 && grid >= 768) ? 80 : 0);
    const int *frz = sizeof(T) == 8 ? e->d.counts : nullptr; // fp64 configuration: a failed sweep leaves P alone (engine.h: filter_frozen)
    if constexpr (sizeof(T) == 4 && PU_F32_PAIRS) {
        if (rect) k_p_update_f32<false, true><<<grid, 256, 0, s>>>(P, e->ldP, e->n, B, e->ldP, m_pad, e->pu_per_xcd, tm, e->rm, stagger);
        else if (avg) k_p_update_f32<true, false><<<grid, 256, 0, s>>>(P, e->ldP, e->n, B, e->ldP, m_pad, e->pu_per_xcd, tm, e->rm, stagger);
        else k_p_update_f32<false, false><<<grid, 256, 0, s>>>(P, e->ldP, e->n, B, e->ldP, m_pad, e->pu_per_xcd, tm, e->rm, stagger);
        return;
    }
    if (rect)
        k_p_update<T, false, true><<<grid, 256, 0, s>>>(P, e->ldP, e->n, B, e->ldP, m_pad, e->pu_per_xcd, tm, e->rm, stagger, frz);
    else if (avg)
        k_p_update<T, true, false><<<grid, 256, 0, s>>>(P, e->ldP, e->n, B, e->ldP, m_pad, e->pu_per_xcd, tm, e->rm, stagger, frz);
    else
        k_p_update<T, false, false><<<grid, 256, 0, s>>>(P, e->ldP, e->n, B, e->ldP, m_pad, e->pu_per_xcd, tm, e->rm, stagger, frz);
}

void launch_p_update(EkfEngine *e, int m_pad, int m)
{
    hipStream_t s = e->stream;
    const int n = e->n;
    const int TM = e->f32 ? 128 : 64;
    const int nt = (n + TM - 1) / TM;
    const bool rect = e->shard_world > 1;
    const int owned = e->rm.r1 - e->rm.r0;
    const int nrt = 1 + (owned + TM - 1) / TM; // camera tile + owned row tiles
    if (e->pu_slots == 0) { // workgroups of this kernel the device keeps resident: the unit list is balanced against it
        int per_cu = 0;
        hipError_t st;
        if (e->f32 && PU_F32_PAIRS) st = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_p_update_f32<false, false>, 256, 0);
        else if (e->f32) st = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_p_update<float, false, false>, 256, 0);
        else st = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_p_update<double, false, false>, 256, 0);
        e->pu_slots = (st == hipSuccess && per_cu > 0) ? per_cu * e->n_cus : -1; // -1: the CU-count rule of rounds 1-2
    }
    // the kernels walk B in slabs of PU_BK rows: rows m .. m_pad of B are zero, so the k-loop may stop at the next multiple
    // of the slab depth instead of the Cholesky panel width (m = 298: 304 rows instead of 320)
    m_pad = round_up(m, PU_BK);
    // Whole tiles first, half units last, in both regimes (scripts/micro/pu_bench.hip, 15 interleaved rounds, m = 1014: half
    // units first 332 us, mixed first round 323-347 us, whole tiles first 323 us).  Short k-loops keep the CU-count split of
    // rounds 1-2 (m = 298: 113 against 116.5 us with the balanced list): there the launch is mostly epilogue traffic.
    const int slots_saved = e->pu_slots;
    if (g_pu_force_slots) e->pu_slots = g_pu_force_slots;
    else if (m_pad < 512 && !rect) e->pu_slots = -1;
    build_units(e, nt, nrt, rect, 0);
    e->pu_slots = slots_saved;
    const int grid = e->pu_per_xcd * 8;
    const int4 *tm = (const int4 *)e->d.pu_tilemap;
    if (grid == 0 || !tm) return; // build_units failed (e->hook_rc is set)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e->timing) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, s);
    }
    const bool avg = !e->p_exact_sym;
    if (e->f32) launch_p_update_t<float>(e, m_pad, grid, tm, avg, rect);
    else launch_p_update_t<double>(e, m_pad, grid, tm, avg, rect);
    if (e->timing) {
        (void)hipEventRecord(e1, s);
        e->pu_events.emplace_back(e0, e1);
        // flops of this launch / 1 (n^2 m counts the symmetric downdate; a rank computes owned x n x m x 2 / 2)
        // algorithmic work of the launch from the UN-padded m (the kernel runs m rounded up to 16 rows of B, the rest zero)
        e->pu_work.push_back(rect ? (double)(owned + 13) * (double)n * (double)m * 2.0 : (double)n * (double)n * (double)m);
        e->pu_m.push_back(m);
    }
    e->p_exact_sym = true;
}

} // namespace ekf
