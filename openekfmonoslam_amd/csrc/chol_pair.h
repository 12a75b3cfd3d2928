// chol_pair.h -- the Cholesky sweep with TWO 32-row panels per launch (included by kernels_update.hip, inside namespace ekf,
// after k_chol_step and its helpers).
//
// k_chol_step's launch is as long as its look-ahead workgroup: kernel boundary ~1.5 us, cold loads ~2, own-tile update ~1,
// 32x32 factor-and-invert 4.2, publish 0.7 -- 9.3 us per panel, of which 3.5 are paid per LAUNCH, not per panel.  Here a
// launch eliminates the panels A = [k0, k0 + 32) and B = [k0 + 32, k0 + 32 + kbB) together, and its look-ahead workgroup
// prepares the NEXT PAIR P, Q: the 64 x 64 inverse of the factor's diagonal block
//
//      inv [ L_AA   0   ]  =  [ Linv_A     0    ]          C = - Linv_B L_BA Linv_A
//          [ L_BA  L_BB ]     [   C     Linv_B  ]
//
// is in V when the launch starts (Linv_A, Linv_B on the diagonal of V, C below: exactly where inv(L) has it), so no role
// waits for anything inside a launch -- the algebra of one 64-wide panel, executed on 32 x 32 blocks:
//
//      [ L_iA  L_iB ] = [ S_iA  S_iB ] inv(L_KK)'   :   L_iA = S_iA Linv_A',   L_iB = S_iA C' + S_iB Linv_B'
//      tiles            S_ij -= L_iA L_jA' + L_iB L_jB'
//      rows of B        B_A = Linv_A R_A,   B_B = C R_A + Linv_B R_B,      R_X = G_X - sum_{j < A} L_Xj B_j  (B_j read once for both)
//      right-hand sides Z likewise;  R_i -= S_iA (Linv_A' Z_A + C' Z_B) + S_iB (Linv_B' Z_B)
//      look-ahead       T_PP, T_QP, T_QQ updated with the pair, T_PP factorised (Linv_P), L_QP = T_QP Linv_P',
//                       T_QQ - L_QP L_QP' factorised (Linv_Q), C' = - Linv_Q L_QP Linv_P; all three published
//
// The first launch of a sweep has only Linv_0 (k_assemble_S): it runs with kbB = 0 (every B term vanishes) and prepares the
// pair (1, 2).
//
// MEASURED (round 3, N = 1000, scripts/sweep_trace.py, profiles/r03_sweep_trace_pairs_n1000_f32.txt): NOT faster, and therefore
// not the default (ekf_set_sweep_mode).  The look-ahead workgroup of a pair launch needs 17.6 us -- loads 2.2, the twelve
// 32^3 fp64 products of the pair update 3.7 (the matrix pipes of ONE CU: 96 x v_mfma_f64_16x16x4 per wavefront), factor P 4.8,
// L_QP / its update / C 1.5, factor Q 4.6, publish 0.8 -- against 2 x 7.6 for two one-panel launches: the extra algebra of
// the 64-wide inverse costs what the saved kernel boundary (1.7 us) and cold-load round (2) bring, 9.65 against 9.29 us per panel.
#pragma once

typedef double acc4_t __attribute__((ext_vector_type(4)));

// 16 x 16 quadrant (bi, bj) of the 32 x 32 x 32 product A Bm' (BT) or A Bm, operands in LDS, one wavefront
template <bool BT>
__device__ __forceinline__ acc4_t quad_prod(acc4_t c, const double (*A)[NB + 1], const double (*Bm)[NB + 1], int bi, int bj, int lr, int lk)
{
#pragma unroll
    for (int k4 = 0; k4 < NB; k4 += 4) {
        const double a = A[16 * bi + lr][k4 + lk];
        const double b = BT ? Bm[16 * bj + lr][k4 + lk] : Bm[k4 + lk][16 * bj + lr];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    return c;
}

// whole 32 x 32 block A Bm' by one wavefront: c[bi][bj]
__device__ __forceinline__ void block_prod_bt(acc4_t (&c)[2][2], const double (*A)[NB + 1], const double (*Bm)[NB + 1], int lr, int lk)
{
#pragma unroll
    for (int k4 = 0; k4 < NB; k4 += 4) {
        const double a0 = A[lr][k4 + lk], a1 = A[16 + lr][k4 + lk];
        const double b0 = Bm[lr][k4 + lk], b1 = Bm[16 + lr][k4 + lk];
        c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c[0][0], 0, 0, 0);
        c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c[0][1], 0, 0, 0);
        c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c[1][0], 0, 0, 0);
        c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c[1][1], 0, 0, 0);
    }
}

// partial sums of the rows of B over the panels j = wv, wv + 4, ... < kp, for NP row panels at once (columns k0 + 32 p of L'):
// the operands of the next two blocks are in flight while one is multiplied (as in k_chol_step); B_j is read once for all NP
#include "chol_bplanes.h" // BPlanes, store_l_planes, b_pair_rows_planes (EKF_PRECISION_F32_EXACT: the rows of B from digit planes)

template <typename T, int NP>
__device__ __forceinline__ void b_row_sums(const T *Lt, const T *Bout, int ldS, int ld, int k0, int c0, int kp, int wv, int lm, int lq,
                                           typename Mma<T>::acc_t (&acc0)[NB / Mma<T>::MB][NB / Mma<T>::MB],
                                           typename Mma<T>::acc_t (&acc1)[NB / Mma<T>::MB][NB / Mma<T>::MB])
{
    using M = Mma<T>;
    constexpr int MB = M::MB, NBLK = NB / MB, KS = 64 / MB, NSTEP = NB / KS;
    T la[3][NP][NSTEP][NBLK], lb[3][NSTEP][NBLK];
#define CP_LOAD(S_, J_)                                                                                           \
    _Pragma("unroll") for (int st = 0; st < NSTEP; ++st) {                                                        \
        const size_t kr = (size_t)(J_) * NB + st * KS + lq;                                                       \
        _Pragma("unroll") for (int bb = 0; bb < NBLK; ++bb) {                                                     \
            _Pragma("unroll") for (int p = 0; p < NP; ++p) la[S_][p][st][bb] = Lt[kr * ldS + k0 + NB * p + MB * bb + lm]; \
            lb[S_][st][bb] = Bout[kr * ld + c0 + MB * bb + lm];                                                   \
        }                                                                                                         \
    }
#define CP_MMA(S_)                                                                                                \
    _Pragma("unroll") for (int st = 0; st < NSTEP; ++st)                                                          \
        _Pragma("unroll") for (int bi = 0; bi < NBLK; ++bi)                                                       \
            _Pragma("unroll") for (int bj = 0; bj < NBLK; ++bj) {                                                 \
                acc0[bi][bj] = M::mma(la[S_][0][st][bi], lb[S_][st][bj], acc0[bi][bj]);                           \
                if (NP > 1) acc1[bi][bj] = M::mma(la[S_][NP - 1][st][bi], lb[S_][st][bj], acc1[bi][bj]);          \
            }
    const int cnt = kp > wv ? (kp - wv + 3) / 4 : 0;
    if (cnt > 0) { CP_LOAD(0, wv) }
    if (cnt > 1) { CP_LOAD(1, wv + 4) }
    for (int i = 0; i < cnt; i += 3) {
        if (i + 2 < cnt) { CP_LOAD(2, wv + 4 * (i + 2)) }
        CP_MMA(0)
        if (i + 1 < cnt) {
            if (i + 3 < cnt) { CP_LOAD(0, wv + 4 * (i + 3)) }
            CP_MMA(1)
        }
        if (i + 2 < cnt) {
            if (i + 4 < cnt) { CP_LOAD(1, wv + 4 * (i + 4)) }
            CP_MMA(2)
        }
    }
#undef CP_LOAD
#undef CP_MMA
}

// Rows of B for a PAIR of panels, 64 columns per workgroup (fp32 covariance, large maps).  With 32-column workgroups the finished
// rows of B are read as 128-byte pieces, one per 48 KB row and workgroup: at n = 12013 that strided stream -- up to 92 MB per launch
// -- runs at a fraction of the HBM rate and the rows of B outlast the look-ahead workgroup 1.5 - 2.5x (50 against 18 us at panel 45,
// N = 2000).  Here a lane takes the column PAIR (2 lm, 2 lm + 1) with one 8-byte load (256 contiguous bytes per row and wavefront,
// half the workgroups, no second workgroup on the CU) and feeds the two values to two MFMA column blocks -- block q holds the columns
// c0 + 2 c + q -- so one operand of L' serves four products.  Two operand stages in flight (three do not fit the register file).
__device__ __forceinline__ void b_pair_rows_wide(const float *Lt, const float *G, float *Bout, int ldS, int ld, int k0, int c0, double (*pool)[NB][NB + 1],
                                                 double (*sLi)[NB + 1], const double (&gv)[4], const double (&gC)[4], const double (&gB)[4])
{
    using M = Mma<float>;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lm = lane & 31, lq = lane >> 5;
    const int kB0 = k0 + NB, kp = k0 / NB;
    typename M::acc_t acc[2][2]; // [panel][column block]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.f;
    // this thread's elements of G_A, G_B: row r, logical columns cg .. cg + 3 of both column blocks = 8 consecutive floats
    const int r = tid >> 3, cg = (tid & 7) * 4;
    float2 gA[4], gBv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        gA[e] = *(const float2 *)(G + (size_t)(k0 + r) * ld + c0 + 2 * (cg + e));
        gBv[e] = *(const float2 *)(G + (size_t)(kB0 + r) * ld + c0 + 2 * (cg + e));
    }
    float la[2][2][16];
    float2 lb[2][16];
#define BW_LOAD(S_, J_)                                                                                \
    _Pragma("unroll") for (int st = 0; st < 16; ++st) {                                                \
        const size_t kr = (size_t)(J_) * NB + st * 2 + lq;                                             \
        la[S_][0][st] = Lt[kr * ldS + k0 + lm];                                                        \
        la[S_][1][st] = Lt[kr * ldS + kB0 + lm];                                                       \
        lb[S_][st] = *(const float2 *)(Bout + kr * ld + c0 + 2 * lm);                                  \
    }
#define BW_MMA(S_)                                                                                     \
    _Pragma("unroll") for (int st = 0; st < 16; ++st) {                                                \
        acc[0][0] = M::mma(la[S_][0][st], lb[S_][st].x, acc[0][0]);                                    \
        acc[0][1] = M::mma(la[S_][0][st], lb[S_][st].y, acc[0][1]);                                    \
        acc[1][0] = M::mma(la[S_][1][st], lb[S_][st].x, acc[1][0]);                                    \
        acc[1][1] = M::mma(la[S_][1][st], lb[S_][st].y, acc[1][1]);                                    \
    }
    const int cnt = kp > wv ? (kp - wv + 3) / 4 : 0; // this wavefront's blocks j = wv, wv + 4, ...
    if (cnt > 0) { BW_LOAD(0, wv) }
    for (int i = 0; i < cnt; i += 2) {
        if (i + 1 < cnt) { BW_LOAD(1, wv + 4 * (i + 1)) }
        BW_MMA(0)
        if (i + 1 < cnt) {
            if (i + 2 < cnt) { BW_LOAD(0, wv + 4 * (i + 2)) }
            BW_MMA(1)
        }
    }
#undef BW_LOAD
#undef BW_MMA
    // partial sums: wavefronts 2, 3 through LDS to wavefronts 0, 1; red[(panel * 2 + half) * 2 + block], 8 x 4224 bytes = pool[0..3]
    float(*red)[NB][NB + 1] = reinterpret_cast<float(*)[NB][NB + 1]>(&pool[0][0][0]);
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) sLi[(tid + q4 * 256) / NB][(tid + q4 * 256) % NB] = gv[q4];
    if (wv >= 2) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) red[(p * 2 + wv - 2) * 2 + q][M::row(rr, lane)][M::col(lane)] = acc[p][q][rr];
    }
    __syncthreads();
    if (wv < 2) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) red[(p * 2 + wv) * 2 + q][M::row(rr, lane)][M::col(lane)] += acc[p][q][rr];
    }
    __syncthreads();
    double ra[2][4], rb[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ra[0][e] = (double)gA[e].x - ((double)red[0][r][cg + e] + (double)red[2][r][cg + e]);
        ra[1][e] = (double)gA[e].y - ((double)red[1][r][cg + e] + (double)red[3][r][cg + e]);
        rb[0][e] = (double)gBv[e].x - ((double)red[4][r][cg + e] + (double)red[6][r][cg + e]);
        rb[1][e] = (double)gBv[e].y - ((double)red[5][r][cg + e] + (double)red[7][r][cg + e]);
    }
    __syncthreads(); // red is dead: pool[0..3] take R_A, R_B of both column blocks, pool[4..5] C and Linv_B
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pool[q][r][cg + e] = ra[q][e];
            pool[2 + q][r][cg + e] = rb[q][e];
        }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        pool[4][(tid + q4 * 256) / NB][(tid + q4 * 256) % NB] = gC[q4];
        pool[5][(tid + q4 * 256) / NB][(tid + q4 * 256) % NB] = gB[q4];
    }
    __syncthreads();
    {   // B_A = Linv_A R_A, B_B = C R_A + Linv_B R_B on the fp64 MFMA, one 16 x 16 quadrant per wavefront and column block
        const int bi = wv >> 1, bj = wv & 1, lr = lane & 15, lk = lane >> 4;
        const acc4_t z4 = {0, 0, 0, 0};
        const acc4_t oa0 = quad_prod<false>(z4, sLi, pool[0], bi, bj, lr, lk);
        const acc4_t oa1 = quad_prod<false>(z4, sLi, pool[1], bi, bj, lr, lk);
        acc4_t ob0 = quad_prod<false>(z4, pool[4], pool[0], bi, bj, lr, lk);
        acc4_t ob1 = quad_prod<false>(z4, pool[4], pool[1], bi, bj, lr, lk);
        ob0 = quad_prod<false>(ob0, pool[5], pool[2], bi, bj, lr, lk);
        ob1 = quad_prod<false>(ob1, pool[5], pool[3], bi, bj, lr, lk);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int row = 16 * bi + lk + 4 * q4, col = c0 + 2 * (16 * bj + lr);
            *(float2 *)(Bout + (size_t)(k0 + row) * ld + col) = make_float2((float)oa0[q4], (float)oa1[q4]);
            *(float2 *)(Bout + (size_t)(kB0 + row) * ld + col) = make_float2((float)ob0[q4], (float)ob1[q4]);
        }
    }
}

template <typename T, typename TG, bool PL = false> // T: type of B and of its MFMA; TG: storage type of the gathered rows G (see k_chol_step); PL: rows of B from digit planes
__global__ void __launch_bounds__(256, (sizeof(T) == 4 || PL) ? 2 : 1) // fp32 / planes: two workgroups per CU (LDS: 70 KB each)
k_chol_pair(double *S, double *LL, float *LLf, int ldS, int m, int m_pad, int k0, int kbA, int kbB, int k2, double *nu, int n_stiles,
            double *V, double *W, float *Wf, int ldw, int *counts, double *Gc, double *zout, double *Bc, const TG *G, T *Bout, int ld,
            int n_bblocks, int n_rhs, int tiles_first, int spacer, unsigned long long *trace, int b_wide, BPlanes bp = BPlanes{})
{
#ifdef EKF_SWEEP_TRACE // debug builds only (scripts/sweep_trace.py): slot 0 first start, 1..4 end of role 0..3, 8.. milestones
    const unsigned long long t_in = trace ? wall_clock64() : 0ull;
#define PAIR_TRACE(slot)                                                                       \
    if (trace && threadIdx.x == 0) {                                                           \
        atomicMin(trace, t_in);                                                                \
        atomicMax(trace + (slot), (unsigned long long)wall_clock64());                         \
    }
#else
#define PAIR_TRACE(slot)
    (void)trace;
#endif
    constexpr int NR = 14; // right-hand sides: nu + 13 camera columns
    __shared__ double pool[7][NB][NB + 1];
    __shared__ double sLi[NB][NB + 1]; // inv(L_AA)
    const int tid = threadIdx.x;
    const int kB0 = k0 + NB;       // first row of panel B (kbB = 0: there is none)
    const bool two = kbB > 0;
    // block order: as k_chol_step (look-ahead first; B before or after the tile groups; empty blocks on the look-ahead's CU)
    int b = blockIdx.x;
    int bcol = -1;
    if (spacer > 0) {
        if (b > 0 && b % spacer == 0) return;
        b -= b / spacer;
    }
    if (tiles_first) {
        if (b >= n_stiles + n_rhs) {
            bcol = b - n_stiles - n_rhs;
            b = n_stiles + 1;
        }
    } else {
        const int f = n_stiles > 0 ? 1 : 0;
        if (b >= f && b < f + n_bblocks) {
            bcol = b - f;
            b = n_stiles + 1;
        } else if (b >= f + n_bblocks && b < f + n_bblocks + n_rhs) {
            b = n_stiles + (b - f - n_bblocks);
        } else if (b >= f + n_bblocks + n_rhs) {
            b -= n_bblocks + n_rhs;
        }
    }
    // the 64 x 64 inverse: requested before anything else (cold), stored to LDS inside each role after its own loads
    double gv[4], gC[4], gB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + q * 256, r = i / NB, c = i % NB;
        gv[q] = V[(size_t)(k0 + r) * ldw + k0 + c];
        gC[q] = two ? V[(size_t)(kB0 + r) * ldw + k0 + c] : 0.0;
        gB[q] = two ? V[(size_t)(kB0 + r) * ldw + kB0 + c] : (r == c ? 1.0 : 0.0);
    }
    const int lane = tid & 63, wv = tid >> 6;

    if (bcol >= 0 && b_wide) { // (the host asks for it only with both panels present and the fp32 covariance)
        if constexpr (sizeof(T) == 4) {
            b_pair_rows_wide((const float *)LLf, (const float *)G, (float *)Bout, ldS, ld, k0, bcol * 2 * NB, pool, sLi, gv, gC, gB);
            PAIR_TRACE(2)
        }
        return;
    }
    if constexpr (PL) {
        if (bcol >= 0) { // EKF_PRECISION_F32_EXACT: the same role from int8 digit planes (chol_bplanes.h)
            b_pair_rows_planes(bp, (const double *)G, ld, m, k0, two, bcol + bp.bcol0, pool, sLi, gv, gC, gB);
            PAIR_TRACE(2)
            return;
        }
    }
    if (bcol >= 0) {
        // ---- rows of B: B_A = Linv_A R_A, B_B = C R_A + Linv_B R_B, columns 32 bcol ..
        using M = Mma<T>;
        constexpr int MB = M::MB, NBLK = NB / MB;
        T(*red)[NB][NB + 1] = reinterpret_cast<T(*)[NB][NB + 1]>(&pool[0][0][0]); // [panel][half] partial sums (pool[0..3] at most)
        double(*sRA)[NB + 1] = pool[4];
        double(*sRB)[NB + 1] = pool[5];
        double(*sC)[NB + 1] = pool[0];  // after the partial sums have been consumed
        double(*sLB)[NB + 1] = pool[1];
        const int lm = lane % MB, lq = lane / MB;
        const int c0 = bcol * NB;
        const int kp = k0 / NB;
        const T *Lt = sizeof(T) == 4 ? (const T *)LLf : (const T *)LL;
        const int r = tid >> 3, cg = (tid & 7) * 4;
        T gA4[4], gB4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gA4[e] = G[(size_t)(k0 + r) * ld + c0 + cg + e];
            gB4[e] = two ? G[(size_t)(kB0 + r) * ld + c0 + cg + e] : (T)0;
        }
        typename M::acc_t accA[NBLK][NBLK], accB[NBLK][NBLK];
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
            for (int bj = 0; bj < NBLK; ++bj)
#pragma unroll
                for (int q = 0; q < M::NACC; ++q) accA[bi][bj][q] = accB[bi][bj][q] = (T)0;
        if (sizeof(T) == 4 && two) {
            b_row_sums<T, 2>(Lt, Bout, ldS, ld, k0, c0, kp, wv, lm, lq, accA, accB);
        } else { // fp64: the operands of both panels at once do not fit the register file
            b_row_sums<T, 1>(Lt, Bout, ldS, ld, k0, c0, kp, wv, lm, lq, accA, accA);
            if (two) b_row_sums<T, 1>(Lt, Bout, ldS, ld, kB0, c0, kp, wv, lm, lq, accB, accB);
        }
        PAIR_TRACE(14)
#pragma unroll
        for (int q = 0; q < 4; ++q) sLi[(tid + q * 256) / NB][(tid + q * 256) % NB] = gv[q];
        if (wv >= 2) {
#pragma unroll
            for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
                for (int bj = 0; bj < NBLK; ++bj)
#pragma unroll
                    for (int q = 0; q < M::NACC; ++q) {
                        red[wv - 2][MB * bi + M::row(q, lane)][MB * bj + M::col(lane)] = accA[bi][bj][q];
                        if (two) red[wv][MB * bi + M::row(q, lane)][MB * bj + M::col(lane)] = accB[bi][bj][q];
                    }
        }
        __syncthreads();
        if (wv < 2) {
#pragma unroll
            for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
                for (int bj = 0; bj < NBLK; ++bj)
#pragma unroll
                    for (int q = 0; q < M::NACC; ++q) {
                        red[wv][MB * bi + M::row(q, lane)][MB * bj + M::col(lane)] += accA[bi][bj][q];
                        if (two) red[2 + wv][MB * bi + M::row(q, lane)][MB * bj + M::col(lane)] += accB[bi][bj][q];
                    }
        }
        __syncthreads();
        double ra[4], rb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ra[e] = (double)gA4[e] - ((double)red[0][r][cg + e] + (double)red[1][r][cg + e]);
            rb[e] = two ? (double)gB4[e] - ((double)red[2][r][cg + e] + (double)red[3][r][cg + e]) : 0.0;
        }
        __syncthreads(); // red is dead: its space takes C and Linv_B
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sRA[r][cg + e] = ra[e];
            sRB[r][cg + e] = rb[e];
        }
        if (two) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sC[(tid + q * 256) / NB][(tid + q * 256) % NB] = gC[q];
                sLB[(tid + q * 256) / NB][(tid + q * 256) % NB] = gB[q];
            }
        }
        __syncthreads();
        // B_A = Linv_A R_A, B_B = C R_A + Linv_B R_B on the fp64 MFMA, one 16 x 16 quadrant per wavefront (as scalar loops of
        // LDS reads the three products took 4 of this role's 6 us after the sums)
        {
            const int bi = wv >> 1, bj = wv & 1, lr = lane & 15, lk = lane >> 4;
            const acc4_t z4 = {0, 0, 0, 0};
            const acc4_t oa = quad_prod<false>(z4, sLi, sRA, bi, bj, lr, lk);
#pragma unroll
            for (int q = 0; q < 4; ++q) Bout[(size_t)(k0 + 16 * bi + lk + 4 * q) * ld + c0 + 16 * bj + lr] = (T)oa[q];
            if (two) {
                acc4_t ob = quad_prod<false>(z4, sC, sRA, bi, bj, lr, lk);
                ob = quad_prod<false>(ob, sLB, sRB, bi, bj, lr, lk);
#pragma unroll
                for (int q = 0; q < 4; ++q) Bout[(size_t)(kB0 + 16 * bi + lk + 4 * q) * ld + c0 + 16 * bj + lr] = (T)ob[q];
            }
        }
        PAIR_TRACE(2)
        return;
    }

    const int lr = lane & 15, lk = lane >> 4;
    if (b > 0 && b < n_stiles) {
        // ---- 2 x 2 group (TI, TJ) of trailing tiles (block rows / columns counted from k2); group (0, 0) is the look-ahead
        // workgroup's region P, Q: it only forms and stores its rows of L
        const int sidx = b - 1;
        int TI = (int)((sqrt(8.0 * sidx + 1.0) - 1.0) * 0.5);
        while ((TI + 1) * (TI + 2) / 2 <= sidx) ++TI;
        while (TI * (TI + 1) / 2 > sidx) --TI;
        const int TJ = sidx - TI * (TI + 1) / 2;
        const bool diag = TI == TJ;
        const int qi = wv >> 1, qj = wv & 1;
        const int i0 = k2 + (2 * TI + qi) * NB, j0 = k2 + (2 * TJ + qj) * NB;
        const bool tile_live = i0 < m && j0 <= i0 && !(TI == 0 && TJ == 0);
        double(*sP)[NB][NB + 1] = pool; // [0], [1]: rows 2 TI, 2 TI + 1; [2], [3]: columns 2 TJ, 2 TJ + 1 (off the diagonal)
        double(*sC)[NB + 1] = pool[4];
        double(*sLB)[NB + 1] = pool[5];
        double v[2][2][4];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = i0 + 16 * bi + lk + 4 * q, c = j0 + 16 * bj + lr;
                    v[bi][bj][q] = (tile_live && r < m && c <= r) ? S[(size_t)r * ldS + c] : 0.0;
                }
        double gpA[4][4], gpB[4][4];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int r0 = k2 + (blk < 2 ? 2 * TI + blk : 2 * TJ + blk - 2) * NB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + q * 256, r = i / NB, c = i % NB;
                const bool in = (blk < 2 || !diag) && r0 + r < m;
                gpA[blk][q] = (in && c < kbA) ? S[(size_t)(r0 + r) * ldS + k0 + c] : 0.0;
                gpB[blk][q] = (in && c < kbB) ? S[(size_t)(r0 + r) * ldS + kB0 + c] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256, r = i / NB, c = i % NB;
            sLi[r][c] = gv[q];
            sC[r][c] = gC[q];
            sLB[r][c] = gB[q];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
                if (blk < 2 || !diag) sP[blk][r][c] = gpA[blk][q];
        }
        __syncthreads();
        // wavefront w: L_wA = S_wA Linv_A' (over S_wA) and the first half of L_wB, S_wA C' (kept in registers)
        acc4_t cc[2][2];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj) cc[bi][bj] = acc4_t{0, 0, 0, 0};
        if (wv < 2 || !diag) {
            acc4_t c[2][2];
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj) c[bi][bj] = acc4_t{0, 0, 0, 0};
            block_prod_bt(c, sP[wv], sLi, lr, lk);
            if (two) block_prod_bt(cc, sP[wv], sC, lr, lk);
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) sP[wv][16 * bi + lk + 4 * q][16 * bj + lr] = c[bi][bj][q];
        }
        __syncthreads();
        acc4_t u[2][2];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj) u[bi][bj] = acc4_t{0, 0, 0, 0};
        if (tile_live) block_prod_bt(u, sP[qi], sP[diag ? qj : 2 + qj], lr, lk);
        if (TJ == 0) { // L leaves the groups of the first group column (mirrored; row-major too on the inverse + GEMM path)
            store_l_block(LL, LLf, W != nullptr, ldS, m_pad, k2 + 2 * TI * NB, k0, kbA, sP[0]);
            store_l_block(LL, LLf, W != nullptr, ldS, m_pad, k2 + (2 * TI + 1) * NB, k0, kbA, sP[1]);
            if constexpr (PL) store_l_planes(bp, m, 2, k2 + 2 * TI * NB, sP[0], k2 + (2 * TI + 1) * NB, sP[1], k0, kbA);
        }
        if (two) {
            __syncthreads(); // every L_xA has been read
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + q * 256, r = i / NB, c = i % NB;
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
                    if (blk < 2 || !diag) sP[blk][r][c] = gpB[blk][q];
            }
            __syncthreads();
            if (wv < 2 || !diag) {
                block_prod_bt(cc, sP[wv], sLB, lr, lk); // L_wB = S_wA C' + S_wB Linv_B'
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                        for (int q = 0; q < 4; ++q) sP[wv][16 * bi + lk + 4 * q][16 * bj + lr] = cc[bi][bj][q];
            }
            __syncthreads();
            if (tile_live) block_prod_bt(u, sP[qi], sP[diag ? qj : 2 + qj], lr, lk);
            if (TJ == 0) {
                store_l_block(LL, LLf, W != nullptr, ldS, m_pad, k2 + 2 * TI * NB, kB0, kbB, sP[0]);
                store_l_block(LL, LLf, W != nullptr, ldS, m_pad, k2 + (2 * TI + 1) * NB, kB0, kbB, sP[1]);
                if constexpr (PL) store_l_planes(bp, m, 2, k2 + 2 * TI * NB, sP[0], k2 + (2 * TI + 1) * NB, sP[1], kB0, kbB);
            }
        }
        if (tile_live) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = i0 + 16 * bi + lk + 4 * q, c = j0 + 16 * bj + lr;
                        if (r < m && c <= r) S[(size_t)r * ldS + c] = v[bi][bj][q] - u[bi][bj][q];
                    }
        }
        PAIR_TRACE(3)
        return;
    }

    if (b < n_stiles) {
        // ---- b == 0, the look-ahead workgroup: the next pair P = [k2, k2 + 32), Q = [k2 + 32, ..) -- its three tiles updated with
        // this launch's pair, factorised, the 64 x 64 inverse published.  Each wavefront owns quadrant (bi, bj) of every block.
        const int bi = wv >> 1, bj = wv & 1;
        const int kQ = k2 + NB;
        const bool hasQ = kQ < m;
        double vpp[4], vqp[4], vqq[4];
        bool lpp[4], lqq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
            lpp[q] = k2 + r < m && c <= r;
            lqq[q] = kQ + r < m && c <= r;
            vpp[q] = lpp[q] ? S[(size_t)(k2 + r) * ldS + k2 + c] : 0.0;
            vqp[q] = (kQ + r < m) ? S[(size_t)(kQ + r) * ldS + k2 + c] : 0.0;
            vqq[q] = lqq[q] ? S[(size_t)(kQ + r) * ldS + kQ + c] : 0.0;
        }
        double g[4][4]; // S_pA, S_pB, S_qA, S_qB
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256, r = i / NB, c = i % NB;
            g[0][q] = (k2 + r < m && c < kbA) ? S[(size_t)(k2 + r) * ldS + k0 + c] : 0.0;
            g[1][q] = (k2 + r < m && c < kbB) ? S[(size_t)(k2 + r) * ldS + kB0 + c] : 0.0;
            g[2][q] = (kQ + r < m && c < kbA) ? S[(size_t)(kQ + r) * ldS + k0 + c] : 0.0;
            g[3][q] = (kQ + r < m && c < kbB) ? S[(size_t)(kQ + r) * ldS + kB0 + c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256, r = i / NB, c = i % NB;
            sLi[r][c] = gv[q];
            pool[4][r][c] = gC[q];
            pool[5][r][c] = gB[q];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) pool[blk][r][c] = g[blk][q];
        }
        __syncthreads();
        PAIR_TRACE(8)
        const acc4_t z4 = {0, 0, 0, 0};
        // (one product after the other: a fused k-loop over six independent accumulators was measured SLOWER, 4.9 against 3.7 us)
        acc4_t lpa = quad_prod<true>(z4, pool[0], sLi, bi, bj, lr, lk);
        acc4_t lpb = z4, lqa = z4, lqb = z4;
        if (two) {
            lpb = quad_prod<true>(lpb, pool[0], pool[4], bi, bj, lr, lk);
            lpb = quad_prod<true>(lpb, pool[1], pool[5], bi, bj, lr, lk);
        }
        if (hasQ) {
            lqa = quad_prod<true>(lqa, pool[2], sLi, bi, bj, lr, lk);
            if (two) {
                lqb = quad_prod<true>(lqb, pool[2], pool[4], bi, bj, lr, lk);
                lqb = quad_prod<true>(lqb, pool[3], pool[5], bi, bj, lr, lk);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
            pool[0][r][c] = lpa[q];
            pool[1][r][c] = lpb[q];
            pool[2][r][c] = lqa[q];
            pool[3][r][c] = lqb[q];
        }
        __syncthreads();
        {
            acc4_t u = quad_prod<true>(z4, pool[0], pool[0], bi, bj, lr, lk);
            if (two) u = quad_prod<true>(u, pool[1], pool[1], bi, bj, lr, lk);
#pragma unroll
            for (int q = 0; q < 4; ++q) vpp[q] -= u[q];
            if (hasQ) {
                acc4_t u1 = quad_prod<true>(z4, pool[2], pool[0], bi, bj, lr, lk);
                acc4_t u2 = quad_prod<true>(z4, pool[2], pool[2], bi, bj, lr, lk);
                if (two) {
                    u1 = quad_prod<true>(u1, pool[3], pool[1], bi, bj, lr, lk);
                    u2 = quad_prod<true>(u2, pool[3], pool[3], bi, bj, lr, lk);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    vqp[q] -= u1[q];
                    vqq[q] -= u2[q];
                }
            }
        }
        // pool[4..6] and sLi are free from here (their last readers passed the barrier above)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
            pool[4][r][c] = lpp[q] ? vpp[q] : ((r == c) ? 1.0 : 0.0);
            pool[6][r][c] = vqp[q];
        }
        __syncthreads();
        PAIR_TRACE(9)
        bool ok = block_chol_inv32_v4(pool[4], pool[5]); // Linv_P in pool[5]
        PAIR_TRACE(10)
        // everything that does not need Linv_Q leaves before the second factorisation (only C' and Linv_Q follow it)
        store_linv(V, W, Wf, ldw, k2, pool[5]);
        if (n_stiles == 1) { // P's rows of L when there is no tile group to store them
            store_l_block(LL, LLf, W != nullptr, ldS, m_pad, k2, k0, kbA, pool[0]);
            if (two) store_l_block(LL, LLf, W != nullptr, ldS, m_pad, k2, kB0, kbB, pool[1]);
            if constexpr (PL) {
                store_l_planes(bp, m, 1, k2, pool[0], k2, pool[0], k0, kbA);
                if (two) store_l_planes(bp, m, 1, k2, pool[1], k2, pool[1], kB0, kbB);
            }
        }
        if (hasQ) {
            const acc4_t lqp = quad_prod<true>(z4, pool[6], pool[5], bi, bj, lr, lk); // L_QP = T_QP Linv_P'
#pragma unroll
            for (int q = 0; q < 4; ++q) sLi[16 * bi + lk + 4 * q][16 * bj + lr] = lqp[q];
            __syncthreads();
            const acc4_t u = quad_prod<true>(z4, sLi, sLi, bi, bj, lr, lk);
            const acc4_t m1 = quad_prod<false>(z4, sLi, pool[5], bi, bj, lr, lk); // L_QP Linv_P
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
                pool[4][r][c] = lqq[q] ? vqq[q] - u[q] : ((r == c) ? 1.0 : 0.0);
                pool[2][r][c] = m1[q];
            }
            // L_QP itself: only the inverse + GEMM path reads it (row-major, k_inv_diag); the rows of B never need it (C does its work)
            if (W) store_l_block(LL, LLf, true, ldS, m_pad, kQ, k2, NB, sLi);
            __syncthreads();
            PAIR_TRACE(11)
            ok = block_chol_inv32_v4(pool[4], pool[6]) && ok; // Linv_Q in pool[6]
            PAIR_TRACE(12)
            const acc4_t cq = quad_prod<false>(z4, pool[6], pool[2], bi, bj, lr, lk); // Linv_Q (L_QP Linv_P)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                V[(size_t)(kQ + 16 * bi + lk + 4 * q) * ldw + k2 + 16 * bj + lr] = -cq[q];
            store_linv(V, W, Wf, ldw, kQ, pool[6]);
        }
        if (!ok && tid == 0) counts[CNT_ERR] = EKF_ERR_NOT_POSITIVE_DEFINITE;
        PAIR_TRACE(1)
        return;
    }

    // ---- right-hand sides [ nu | Gc ]: Z_A = Linv_A R_A, Z_B = C R_A + Linv_B R_B (final rows of z and Bc), then for 64 rows each
    // R_i -= S_iA W_A + S_iB W_B with W_A = Linv_A' Z_A + C' Z_B, W_B = Linv_B' Z_B
    constexpr int NRP = 16;
    typedef double(*rhs_t)[NRP + 1];
    // six 32 x 17 matrices back to back in pool[0..3] (3264 of 4224 doubles)
    rhs_t sRa = reinterpret_cast<rhs_t>(&pool[0][0][0]), sRb = sRa + NB, sZa = sRb + NB, sZb = sZa + NB, sWa = sZb + NB, sWb = sWa + NB;
    static_assert(6 * NB * (NRP + 1) <= 4 * NB * (NB + 1), "right-hand-side scratch fits pool[0..3]");
    double(*sC)[NB + 1] = pool[4];
    double(*sLB)[NB + 1] = pool[5];
    const int nb = b - n_stiles;
    const int nrhs = Gc ? NR : 1;
    double rva[2], rvb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = tid + q * 256, r = idx / NRP, c = idx % NRP;
        rva[q] = rvb[q] = 0.0;
        if (r < kbA && c < nrhs) rva[q] = c == 0 ? nu[k0 + r] : Gc[(size_t)(k0 + r) * 16 + c - 1];
        if (r < kbB && c < nrhs) rvb[q] = c == 0 ? nu[kB0 + r] : Gc[(size_t)(kB0 + r) * 16 + c - 1];
    }
    const int i = k2 + nb * 64 + (tid >> 2), part = tid & 3;
    double sva[8], svb[8];
    {
        const double *srow = S + (size_t)min(i, m - 1) * ldS + 8 * part;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            const double2 ta = *(const double2 *)(srow + k0 + q);
            sva[q] = i < m ? ta.x : 0.0;
            sva[q + 1] = i < m ? ta.y : 0.0;
            svb[q] = svb[q + 1] = 0.0;
            if (two) { // columns kB0 + kbB .. of S are allocated (ldS is padded) but not part of the panel: masked below
                const double2 tb = *(const double2 *)(srow + kB0 + q);
                svb[q] = (i < m && 8 * part + q < kbB) ? tb.x : 0.0;
                svb[q + 1] = (i < m && 8 * part + q + 1 < kbB) ? tb.y : 0.0;
            }
        }
    }
    double old[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = part + 4 * q;
        old[q] = 0.0;
        if (i < m && c < nrhs) old[q] = c == 0 ? nu[i] : Gc[(size_t)i * 16 + c - 1];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (tid + q * 256) / NB, c = (tid + q * 256) % NB;
        sLi[r][c] = gv[q];
        sC[r][c] = gC[q];
        sLB[r][c] = gB[q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        sRa[(tid + q * 256) / NRP][(tid + q * 256) % NRP] = rva[q];
        sRb[(tid + q * 256) / NRP][(tid + q * 256) % NRP] = rvb[q];
    }
    __syncthreads();
    const int lm = lane & 15, lq = lane >> 4, wh = wv & 1;
    if (wv < 2) { // Z_A, rows 16 wh ..
        acc4_t z = {0, 0, 0, 0};
#pragma unroll
        for (int k4 = 0; k4 < NB; k4 += 4) z = __builtin_amdgcn_mfma_f64_16x16x4f64(sLi[16 * wh + lm][k4 + lq], sRa[k4 + lq][lm], z, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * wh + lq + 4 * q;
            sZa[r][lm] = r < kbA ? z[q] : 0.0;
        }
    } else { // Z_B
        acc4_t z = {0, 0, 0, 0};
        if (two) {
#pragma unroll
            for (int k4 = 0; k4 < NB; k4 += 4) {
                z = __builtin_amdgcn_mfma_f64_16x16x4f64(sC[16 * wh + lm][k4 + lq], sRa[k4 + lq][lm], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f64_16x16x4f64(sLB[16 * wh + lm][k4 + lq], sRb[k4 + lq][lm], z, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * wh + lq + 4 * q;
            sZb[r][lm] = r < kbB ? z[q] : 0.0;
        }
    }
    __syncthreads();
    if (wv < 2) { // W_A = Linv_A' Z_A + C' Z_B
        acc4_t w = {0, 0, 0, 0};
#pragma unroll
        for (int k4 = 0; k4 < NB; k4 += 4) {
            w = __builtin_amdgcn_mfma_f64_16x16x4f64(sLi[k4 + lq][16 * wh + lm], sZa[k4 + lq][lm], w, 0, 0, 0);
            if (two) w = __builtin_amdgcn_mfma_f64_16x16x4f64(sC[k4 + lq][16 * wh + lm], sZb[k4 + lq][lm], w, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sWa[16 * wh + lq + 4 * q][lm] = w[q];
    } else { // W_B = Linv_B' Z_B
        acc4_t w = {0, 0, 0, 0};
        if (two) {
#pragma unroll
            for (int k4 = 0; k4 < NB; k4 += 4) w = __builtin_amdgcn_mfma_f64_16x16x4f64(sLB[k4 + lq][16 * wh + lm], sZb[k4 + lq][lm], w, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sWb[16 * wh + lq + 4 * q][lm] = w[q];
    }
    if (nb == 0) { // Z is final: rows of z and of Bc = inv(L) Gc
        for (int idx = tid; idx < 2 * NB * NR; idx += 256) {
            const int p = idx / (NB * NR), r = (idx / NR) % NB, c = idx % NR;
            if (r < (p ? kbB : kbA) && c < nrhs) {
                const double zv = p ? sZb[r][c] : sZa[r][c];
                const int row = (p ? kB0 : k0) + r;
                if (c == 0) zout[row] = zv;
                else Bc[(size_t)row * 16 + c - 1] = zv;
            }
        }
    }
    __syncthreads();
    {
        double acc[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) acc[c] = 0.0;
        if (Gc) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double wrow[NR];
#pragma unroll
                for (int c = 0; c < NR; ++c) wrow[c] = sWa[8 * part + q][c];
#pragma unroll
                for (int c = 0; c < NR; ++c) acc[c] += sva[q] * wrow[c];
            }
            if (two) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    double wrow[NR];
#pragma unroll
                    for (int c = 0; c < NR; ++c) wrow[c] = sWb[8 * part + q][c];
#pragma unroll
                    for (int c = 0; c < NR; ++c) acc[c] += svb[q] * wrow[c];
                }
            }
#pragma unroll
            for (int c = 0; c < NR; ++c) acc[c] = quad_sum(acc[c]);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[0] += sva[q] * sWa[8 * part + q][0] + svb[q] * sWb[8 * part + q][0];
            acc[0] = quad_sum(acc[0]);
        }
        if (i < m) {
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                if ((c & 3) != part) continue;
                if (c == 0) nu[i] = old[0] - acc[0];
                else if (Gc) Gc[(size_t)i * 16 + c - 1] = old[c >> 2] - acc[c];
            }
        }
    }
    PAIR_TRACE(4)
#undef PAIR_TRACE
}
