// chol_bplanes.h -- the rows of B = inv(L) G of the Cholesky sweep on the int8 MFMA (EKF_PRECISION_F32_EXACT, one GPU).
// Included by kernels_update.hip inside namespace ekf, before k_chol_step.
//
// In fp64 the row block of a late panel is bound by the fp64 matrix pipe of ONE CU: a workgroup owns 32 columns and multiplies
// every finished 32-row block of B by the panel's block of L' -- 32 v_mfma_f64_16x16x4 (64 cycles each) per finished block
// and wavefront, 6.6 us at the last panel of an m = 1014 update, where the look-ahead workgroup the launch waits for needs
// 7.5: the sweep ran 11.7 us per panel against 9.1 with the fp32 rows of B.  The exact downdate (kernels_pexact.hip) wants B
// as int8 digit planes anyway, so the rows of B are formed FROM those planes:
//
//   * column scales of B are known before B exists: sum_k B_kj^2 = P_jj - P_jj(new) <= P_jj, so |B_kj| <= sqrt(P_jj)
//     (k_gather stores the exponent of sqrt(P_jj) per column); a finished row block is cut into PX_S digit planes by the
//     workgroup that computed it, in the layout the downdate reads ([k / 16][column][k % 16]);
//   * row scales of L likewise: sum_k L_rk^2 = S_rr, so |L_rk| <= sqrt(S_rr) (k_assemble_S stores the exponent per row); the
//     workgroups that store a block of L also store its digit planes, one contiguous KB per (plane, block, k-half);
//   * the left-looking sum  sum_{j<k} L_kj B_j  is then 15 v_mfma_i32_32x32x32_i8 (32 cycles each) per finished block, exact
//     in int32, the five levels combined in fp64 once per launch: matrix time vanishes and the role streams 5 instead of 8
//     bytes per element of B;
//   * the rest of the role is unchanged and fp64: R = G_k - sum, B_k = inv(L_kk) R on the fp64 MFMA, stored in fp64 (for
//     dx = B'z) and as digit planes (for the later panels and for the downdate).
//
// Accuracy of the a-priori scales against the column's true maximum (which is only known when B is complete): 5 digits with
// the bound sqrt(P_jj) leave the cross-feature entries of B'B with 7e-13 rms absolute error against 1.2e-13 with the true
// maximum (N = 1000, scripts/diag_accum.py, profiles/r04_accumulation_schemes.txt) -- the fp32 rounding of the result is 3e-11.
#pragma once
// (digit_planes.h is included by kernels_update.hip before this file)

typedef int bp_v4i __attribute__((ext_vector_type(4)));
typedef int bp_v16i __attribute__((ext_vector_type(16)));

struct BPlanes {
    int8_t *Bq = nullptr;     // digit planes of B: [PX_S][rows / 16][ldq][16]
    size_t b_stride = 0;      // bytes per plane of B
    int ldq = 0;              // columns per plane row
    const int *bexp = nullptr; // biased exponent of the column scale (|B_kj| < 2^(bexp - 1022))
    int8_t *Lq = nullptr;     // digit planes of L: [PX_S][row block][column block][k half][row][16]
    size_t l_stride = 0;      // bytes per plane of L
    int nbk = 0;              // blocks per side of the L planes
    const int *lexp = nullptr; // biased exponent of the row scale of L
    const int *grow = nullptr; // row of the H P table behind every row of G (-1: a zero row), see k_gather; null: G is a copy
    int bcol0 = 0;             // first column block of this rank (row-sharded engines form their own blocks only)
    int no_fp64 = 0;           // the rows of B are not stored in fp64 (dx = B'z comes from the planes, k_dx_planes)
    int *counts = nullptr;     // the engine's counter block: a row of B that does not fit its a-priori column scale raises CNT_ERR (digit_planes.h)
    int c_live0 = 0, n_live = 0; // ... checked for the columns [c_live0, n_live) only: the state's columns this engine forms B for (the padding columns of a
                               // block, and on a row-sharded rank the neighbour's columns of a straddling block, may hold stale rows of H P: never used)
};

// Digit planes of TWO 32 x 32 blocks of L (rows i0a.. and i0b.., columns k0 .. k0 + kb - 1, in LDS; n_blk = 1: the first
// only), by all 256 threads: thread (block, k half, 8-byte half, row) cuts eight values and stores eight bytes per plane.
// (As sixty-four threads cutting sixteen values each with a carry chain this took the tile groups, which the launch waits for
// on the early panels, from ~6 to 10.4 us.)
__device__ __forceinline__ void store_l_planes(const BPlanes &bp, int m, int n_blk, int i0a, const double (*sLa)[NB + 1], int i0b,
                                               const double (*sLb)[NB + 1], int k0, int kb)
{
    if (!bp.Lq) return;
    const int tid = threadIdx.x;
    const int blk = tid >> 7, r = tid & 31, kg = (tid >> 5) & 1, hf = (tid >> 6) & 1;
    if (blk >= n_blk) return;
    const int i0 = blk ? i0b : i0a;
    const double(*sL)[NB + 1] = blk ? sLb : sLa;
    const int er = i0 + r < m ? bp.lexp[i0 + r] - 1022 : 0;
    const int sh = 8 * PX_S - 2 - er;
    unsigned w[PX_S][2];
#pragma unroll
    for (int s = 0; s < PX_S; ++s) w[s][0] = w[s][1] = 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int col = 16 * kg + 8 * hf + c;
        const double v = (i0 + r < m && col < kb) ? sL[r][col] : 0.0;
        const unsigned long long dw = px_digit_word(v, sh);
#pragma unroll
        for (int s = 0; s < PX_S; ++s) w[s][c >> 2] |= px_digit_byte(dw, s) << (8 * (c & 3));
    }
    const size_t off = ((size_t)(i0 / NB) * bp.nbk + k0 / NB) * 1024 + kg * 512 + r * 16 + hf * 8;
#pragma unroll
    for (int s = 0; s < PX_S; ++s) *(uint2 *)(bp.Lq + (size_t)s * bp.l_stride + off) = make_uint2(w[s][0], w[s][1]);
}

// Row block k of B, columns 32 bcol ..: B_k = inv(L_kk) (G_k - sum_{j<k} L_kj B_j), the sum from the digit planes.
// pool: the launch's LDS blocks (k_chol_step): [0], [1] partial sums, [2] right-hand side, [3] the finished block; sLi = inv(L_kk)
__device__ __forceinline__ void b_rows_planes(const BPlanes &bp, const double *G, double *Bout, int ld, int m, int k0, int bcol,
                                              double (*pool)[NB][NB + 1], double (*sLi)[NB + 1], const double (&gv)[4])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kg = lane >> 5, idx = lane & 31;
    const int c0 = bcol * NB, kp = k0 / NB;
    double(*red)[NB][NB + 1] = pool; // [0], [1]
    double(*sR)[NB + 1] = pool[2];
    double(*sO)[NB + 1] = pool[3];
    // this thread's four elements of G_k, requested first: they are cold and only needed at the end
    const int r4 = tid >> 3, cg = (tid & 7) * 4;
    double g4[4];
    {
        const int gr = bp.grow ? bp.grow[k0 + r4] : k0 + r4; // (grow: G is the H P table; rows m .. m_pad are zero)
#pragma unroll
        for (int e = 0; e < 4; ++e) g4[e] = gr >= 0 ? G[(size_t)gr * ld + c0 + cg + e] : 0.0;
    }
    bp_v16i acc[PX_S];
#pragma unroll
    for (int L = 0; L < PX_S; ++L)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[L][r] = 0;
    // the wavefront's finished blocks j = wv, wv + 4, ...: the operands of the next block are in flight while one is multiplied
    const int8_t *pl = bp.Lq + ((size_t)kp * bp.nbk * 2 + kg) * 512 + idx * 16;            // + j * 1024 + s * l_stride
    const int8_t *pb = bp.Bq + ((size_t)kg * bp.ldq + c0 + idx) * 16;                       // + 2 j * ldq * 16 + s * b_stride
    const size_t bstep = (size_t)2 * bp.ldq * 16;
#ifndef BP_STAGES
#define BP_STAGES 2 // operand blocks in flight besides the one being multiplied + 1 (3 needs 13 spilled registers at two workgroups per CU)
#endif
    bp_v4i la[BP_STAGES][PX_S], lb[BP_STAGES][PX_S];
#define BP_LOAD(S_, J_)                                                                                           \
    _Pragma("unroll") for (int s = 0; s < PX_S; ++s) {                                                            \
        la[S_][s] = *(const bp_v4i *)(pl + (size_t)(J_) * 1024 + (size_t)s * bp.l_stride);                        \
        lb[S_][s] = *(const bp_v4i *)(pb + (size_t)(J_) * bstep + (size_t)s * bp.b_stride);                       \
    }
#define BP_MMA(S_)                                                                                                \
    _Pragma("unroll") for (int s = 0; s < PX_S; ++s)                                                              \
        _Pragma("unroll") for (int t = 0; t < PX_S - s; ++t)                                                      \
            acc[s + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(la[S_][s], lb[S_][t], acc[s + t], 0, 0, 0);
    const int cnt = kp > wv ? (kp - wv + 3) / 4 : 0;
#if BP_STAGES == 3
    if (cnt > 0) { BP_LOAD(0, wv) }
    if (cnt > 1) { BP_LOAD(1, wv + 4) }
    for (int i = 0; i < cnt; i += 3) {
        if (i + 2 < cnt) { BP_LOAD(2, wv + 4 * (i + 2)) }
        BP_MMA(0)
        if (i + 1 < cnt) {
            if (i + 3 < cnt) { BP_LOAD(0, wv + 4 * (i + 3)) }
            BP_MMA(1)
        }
        if (i + 2 < cnt) {
            if (i + 4 < cnt) { BP_LOAD(1, wv + 4 * (i + 4)) }
            BP_MMA(2)
        }
    }
#else
    if (cnt > 0) { BP_LOAD(0, wv) }
    for (int i = 0; i < cnt; i += 2) {
        if (i + 1 < cnt) { BP_LOAD(1, wv + 4 * (i + 1)) }
        BP_MMA(0)
        if (i + 1 < cnt) {
            if (i + 2 < cnt) { BP_LOAD(0, wv + 4 * (i + 2)) }
            BP_MMA(1)
        }
    }
#endif
#undef BP_LOAD
#undef BP_MMA
#pragma unroll
    for (int q = 0; q < 4; ++q) sLi[(tid + q * 256) / NB][(tid + q * 256) % NB] = gv[q];
    // levels -> fp64 (value = 2^(e_row + e_col - 12) sum_L acc_L 256^-L, see kernels_pexact.hip), partial sums of the four
    // wavefronts through LDS: 2, 3 store, 0, 1 add
    const int ec = bp.bexp[c0 + idx] - 1022;
    double part[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
        double tsum = (double)acc[PX_S - 1][r];
#pragma unroll
        for (int L = PX_S - 2; L >= 0; --L) tsum = fma(tsum, 1.0 / 256.0, (double)acc[L][r]);
        const int er = k0 + row < m ? bp.lexp[k0 + row] - 1022 : 0;
        part[r] = cnt > 0 ? ldexp(tsum, er + ec - 12) : 0.0;
    }
    if (wv >= 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wv - 2][(r & 3) + 8 * (r >> 2) + 4 * kg][idx] = part[r];
    }
    __syncthreads();
    if (wv < 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wv][(r & 3) + 8 * (r >> 2) + 4 * kg][idx] += part[r];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) sR[r4][cg + e] = g4[e] - (red[0][r4][cg + e] + red[1][r4][cg + e]);
    __syncthreads();
    {   // B_k = Linv_k R on the fp64 MFMA, one 16 x 16 quadrant per wavefront
        const int bi = wv >> 1, bj = wv & 1, lr = lane & 15, lk = lane >> 4;
        const acc4_t o = quad_prod<false>(acc4_t{0, 0, 0, 0}, sLi, sR, bi, bj, lr, lk);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!bp.no_fp64) Bout[(size_t)(k0 + 16 * bi + lk + 4 * q) * ld + c0 + 16 * bj + lr] = o[q];
            sO[16 * bi + lk + 4 * q][16 * bj + lr] = o[q];
        }
    }
    __syncthreads();
    {   // the block's digit planes: thread (k half, 4-row quarter, column) cuts four rows of its column: four bytes per plane
        const int col = tid & 31, qr = (tid >> 5) & 3, kh = tid >> 7;
        const int sh = 8 * PX_S - 2 - (bp.bexp[c0 + col] - 1022);
        unsigned w[PX_S];
#pragma unroll
        for (int s = 0; s < PX_S; ++s) w[s] = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * kh + 4 * qr + i;
            const unsigned long long dw = px_digit_word_checked(k0 + row < m ? sO[row][col] : 0.0, sh, (c0 + col >= bp.c_live0 && c0 + col < bp.n_live) ? bp.counts : nullptr);
#pragma unroll
            for (int s = 0; s < PX_S; ++s) w[s] |= px_digit_byte(dw, s) << (8 * i);
        }
        int8_t *dst = bp.Bq + ((size_t)(2 * kp + kh) * bp.ldq + c0 + col) * 16 + 4 * qr;
#pragma unroll
        for (int s = 0; s < PX_S; ++s) *(unsigned *)(dst + (size_t)s * bp.b_stride) = w[s];
    }
}

// The same role for a launch that eliminates TWO panels A = [k0, k0 + 32) and B = [k0 + 32, ..) (chol_pair.h):
//     R_X = G_X - sum_{j < A} L_Xj B_j   (X = A, B),     B_A = Linv_A R_A,     B_B = C R_A + Linv_B R_B
// Every finished block B_j is fetched from the memory-side cache ONCE per pair instead of once per panel: wavefronts 0, 1 form the
// sum of panel A over the blocks j = 0, 2, .. / 1, 3, .., wavefronts 2, 3 the sum of panel B over the same blocks (their loads
// of B_j's planes hit the L1 / L2 lines their neighbours just pulled).  `two` false (the first launch of a sweep, or a short last
// launch): panel A alone, its blocks dealt to all four wavefronts as in b_rows_planes.
// pool: [0..3] partial sums, then C and Linv_B in [0], [1]; [4], [5] the right-hand sides R_A, R_B; [2], [3] the finished blocks.
__device__ __forceinline__ void b_pair_rows_planes(const BPlanes &bp, const double *G, int ld, int m, int k0, bool two, int bcol,
                                                   double (*pool)[NB][NB + 1], double (*sLi)[NB + 1], const double (&gv)[4],
                                                   const double (&gC)[4], const double (&gB)[4])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kg = lane >> 5, idx = lane & 31;
    const int c0 = bcol * NB, kp = k0 / NB, kB0 = k0 + NB;
    double(*red)[NB][NB + 1] = pool; // [0..3]
    double(*sRA)[NB + 1] = pool[4];
    double(*sRB)[NB + 1] = pool[5];
    const int r4 = tid >> 3, cg = (tid & 7) * 4;
    double gA4[4], gB4[4];
    {
        const int ra = bp.grow ? bp.grow[k0 + r4] : k0 + r4;
        const int rb = two ? (bp.grow ? bp.grow[kB0 + r4] : kB0 + r4) : -1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            gA4[e] = ra >= 0 ? G[(size_t)ra * ld + c0 + cg + e] : 0.0;
            gB4[e] = rb >= 0 ? G[(size_t)rb * ld + c0 + cg + e] : 0.0;
        }
    }
    bp_v16i acc[PX_S];
#pragma unroll
    for (int L = 0; L < PX_S; ++L)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[L][r] = 0;
    // this wavefront's panel (row block of L) and its share of the finished blocks
    const int side = two ? (wv >> 1) : 0;            // 0: panel A, 1: panel B
    const int j_first = two ? (wv & 1) : wv, j_step = two ? 2 : 4;
    const int kprow = kp + side;
    const int8_t *pl = bp.Lq + ((size_t)kprow * bp.nbk * 2 + kg) * 512 + idx * 16; // + j * 1024 + s * l_stride
    const int8_t *pb = bp.Bq + ((size_t)kg * bp.ldq + c0 + idx) * 16;               // + 2 j * ldq * 16 + s * b_stride
    const size_t bstep = (size_t)2 * bp.ldq * 16;
    bp_v4i la[2][PX_S], lb[2][PX_S];
#define BPP_LOAD(S_, J_)                                                                                          \
    _Pragma("unroll") for (int s = 0; s < PX_S; ++s) {                                                            \
        la[S_][s] = *(const bp_v4i *)(pl + (size_t)(J_) * 1024 + (size_t)s * bp.l_stride);                        \
        lb[S_][s] = *(const bp_v4i *)(pb + (size_t)(J_) * bstep + (size_t)s * bp.b_stride);                       \
    }
#define BPP_MMA(S_)                                                                                               \
    _Pragma("unroll") for (int s = 0; s < PX_S; ++s)                                                              \
        _Pragma("unroll") for (int t = 0; t < PX_S - s; ++t)                                                      \
            acc[s + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(la[S_][s], lb[S_][t], acc[s + t], 0, 0, 0);
    const int cnt = kp > j_first ? (kp - j_first + j_step - 1) / j_step : 0; // blocks j < kp (the pair's own blocks go through C)
    if (cnt > 0) { BPP_LOAD(0, j_first) }
    for (int i = 0; i < cnt; i += 2) {
        if (i + 1 < cnt) { BPP_LOAD(1, j_first + j_step * (i + 1)) }
        BPP_MMA(0)
        if (i + 1 < cnt) {
            if (i + 2 < cnt) { BPP_LOAD(0, j_first + j_step * (i + 2)) }
            BPP_MMA(1)
        }
    }
#undef BPP_LOAD
#undef BPP_MMA
#pragma unroll
    for (int q = 0; q < 4; ++q) sLi[(tid + q * 256) / NB][(tid + q * 256) % NB] = gv[q];
    const int ec = bp.bexp[c0 + idx] - 1022;
    const int krow0 = side ? kB0 : k0;
    double part[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
        double tsum = (double)acc[PX_S - 1][r];
#pragma unroll
        for (int L = PX_S - 2; L >= 0; --L) tsum = fma(tsum, 1.0 / 256.0, (double)acc[L][r]);
        const int er = krow0 + row < m ? bp.lexp[krow0 + row] - 1022 : 0;
        part[r] = cnt > 0 ? ldexp(tsum, er + ec - 12) : 0.0;
    }
    if (two) { // every wavefront stores its partial sum: [0], [1] panel A, [2], [3] panel B
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wv][(r & 3) + 8 * (r >> 2) + 4 * kg][idx] = part[r];
        __syncthreads();
    } else { // four partial sums of panel A: 2, 3 store, 0, 1 add
        if (wv >= 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wv - 2][(r & 3) + 8 * (r >> 2) + 4 * kg][idx] = part[r];
        }
        __syncthreads();
        if (wv < 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wv][(r & 3) + 8 * (r >> 2) + 4 * kg][idx] += part[r];
        }
        __syncthreads();
    }
    double ra[4], rb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ra[e] = gA4[e] - (red[0][r4][cg + e] + red[1][r4][cg + e]);
        rb[e] = two ? gB4[e] - (red[2][r4][cg + e] + red[3][r4][cg + e]) : 0.0;
    }
    __syncthreads(); // the partial sums are dead: their space takes C, Linv_B and the finished blocks
    double(*sC)[NB + 1] = pool[0];
    double(*sLB)[NB + 1] = pool[1];
    double(*sOA)[NB + 1] = pool[2];
    double(*sOB)[NB + 1] = pool[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sRA[r4][cg + e] = ra[e];
        sRB[r4][cg + e] = rb[e];
    }
    if (two) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sC[(tid + q * 256) / NB][(tid + q * 256) % NB] = gC[q];
            sLB[(tid + q * 256) / NB][(tid + q * 256) % NB] = gB[q];
        }
    }
    __syncthreads();
    {   // B_A = Linv_A R_A, B_B = C R_A + Linv_B R_B on the fp64 MFMA, one 16 x 16 quadrant per wavefront
        const int bi = wv >> 1, bj = wv & 1, lr = lane & 15, lk = lane >> 4;
        const acc4_t z4 = {0, 0, 0, 0};
        const acc4_t oa = quad_prod<false>(z4, sLi, sRA, bi, bj, lr, lk);
#pragma unroll
        for (int q = 0; q < 4; ++q) sOA[16 * bi + lk + 4 * q][16 * bj + lr] = oa[q];
        if (two) {
            acc4_t ob = quad_prod<false>(z4, sC, sRA, bi, bj, lr, lk);
            ob = quad_prod<false>(ob, sLB, sRB, bi, bj, lr, lk);
#pragma unroll
            for (int q = 0; q < 4; ++q) sOB[16 * bi + lk + 4 * q][16 * bj + lr] = ob[q];
        }
    }
    __syncthreads();
    {   // digit planes of the finished block(s): thread (k half, 4-row quarter, column) cuts four rows of its column
        const int col = tid & 31, qr = (tid >> 5) & 3, kh = tid >> 7;
        const int sh = 8 * PX_S - 2 - (bp.bexp[c0 + col] - 1022);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (x == 1 && !two) break;
            const double(*sO)[NB + 1] = x ? sOB : sOA;
            const int kr0 = x ? kB0 : k0;
            unsigned w[PX_S];
#pragma unroll
            for (int s = 0; s < PX_S; ++s) w[s] = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * kh + 4 * qr + i;
                const unsigned long long dw = px_digit_word_checked(kr0 + row < m ? sO[row][col] : 0.0, sh, (c0 + col >= bp.c_live0 && c0 + col < bp.n_live) ? bp.counts : nullptr);
#pragma unroll
                for (int s = 0; s < PX_S; ++s) w[s] |= px_digit_byte(dw, s) << (8 * i);
            }
            int8_t *dst = bp.Bq + ((size_t)(2 * (kp + x) + kh) * bp.ldq + c0 + col) * 16 + 4 * qr;
#pragma unroll
            for (int s = 0; s < PX_S; ++s) *(unsigned *)(dst + (size_t)s * bp.b_stride) = w[s];
        }
    }
}
