// kernels_update.hip -- A7, the Kalman update (EKF/Update.cpp:92-319), in the algebraically equivalent
// "square-root downdate" form that never materialises the reference's dense n x n temporaries:
//
//     G  = H P            (m x n; rows gathered from the cached H_f P pairs, HBM-bound copy)
//     S  = G H' + R       (m x m fp64, lower triangle; block-sparse H => 13-term dot products)
//     S  = L L'           (blocked right-looking Cholesky, fp64, one launch per 32-wide panel; z = inv(L) nu and, in the
//                          fp32 configuration, the fp64 camera columns of B ride along as right-hand sides)
//     B  = inv(L) G       up to 2048 rows: row block k inside the launch of panel k (forward substitution on the MFMA
//                          beside the look-ahead factorisation); above: inv(L) explicitly (128 x 128 diagonal chunks,
//                          then doubling levels, fp64 MFMA) and ONE GEMM against it (kernels_gemm.hip)
//     dx = B' z           (= K nu,  K = P H' inv(S))
//     P <- sym(P) - B' B  (= 0.5 ((I-KH)P + ((I-KH)P)'), kernels_pupdate.hip -- the MFMA kernel)
//     q normalisation and its Jacobian on the rows/columns 3..6 of P (Update.cpp:45-85, 303-317)
//
// The reference forms K = P H' inv(S) with an LU inverse and multiplies the dense (I - K H) by P (2 n^3 flops);
// both give the same x and P up to rounding (S is symmetric positive definite: R = pixelErrorX * I).
#include "engine.h"
#include <vector>
#include <cstdio>
#include <mutex>
#include <cstdlib>
#include <atomic>
#ifdef EKF_SWEEP_TRACE // when each of the four wavefronts of the chain workgroup reaches the join of the factorisation (scripts/persist_trace.py)
namespace ekf { __device__ unsigned long long g_w2_clock[4]; }
#define W2_STAMP(slot) { if ((slot) >= 40 && (slot) < 44 && (threadIdx.x & 63) == 0) ekf::g_w2_clock[(slot) - 40] = wall_clock64(); }
#endif
#include "chol32.h"
#include "mma_tile.h"
#include "digit_planes.h"

namespace ekf {

// ---------------------------------------------------------------------------------------------------- gather
// Per selected match i: A[2i..2i+1, :] = HP[2f..2f+1, :], its Jacobian blocks, position/dimension and the
// dead-banded innovation nu (Update.cpp:125-135).  Rows m..m_pad of A are zero-filled for the k-tiled kernels.
// (the body takes its workgroup coordinates as arguments: k_gather_assemble runs it beside the assembly of S in ONE launch)
template <typename T, typename TP> // TP: type of the diagonal table (the covariance itself on one GPU, the float table of a sharded engine)
__device__ __forceinline__ void
gather_body(const int bx, const int by, const EkfMatch *matches, int M, int m_pad, const T *HP, T *A, int ld, int n_pad, const double *uv_tab,
            const double *Hs_tab, const double *Hf_tab, const int *feat_type, const int *feat_covpos, double *nu,
            double *mHs, double *mHf, int *mpos, int *mdim, const double *HPc, double *Gc, int *bexp, const TP *Pdiag, int ldpd,
            int n, int *grow, int scale_shift)
{
    // grow != nullptr (exact configuration, rows of B from digit planes): the rows are NOT copied -- their consumers (k_assemble_S,
    // b_rows_planes) read H P through the row map grow[row] = row of H P, -1 for the zero rows m .. m_pad; the launch then has
    // one workgroup column (grid.x = 1 covers the per-match bookkeeping; the column scales are written by grid.y = 0's columns)
    const int row = by;
    if (bexp && Pdiag) {
        // rows of B from digit planes (chol_bplanes.h): the column scales are known before B exists, |B_kj| <= sqrt(P_jj); the
        // m_pad workgroups of this launch (one per row, no copy) share the columns
        if (bx == 0) {
            const int per = (n_pad + m_pad - 1) / m_pad;
            for (int j = row * per + threadIdx.x; j < min((row + 1) * per, n_pad); j += 256) {
                const double pjj = j < n ? (double)Pdiag[(size_t)j * ldpd + j] : 0.0;
                // (+ scale_shift bits of head-room, engine.h px_scale_shift: the top digit plane of B is then almost all zeros and the
                // downdate skips its products, px_flag_plane0)
                bexp[j] = pjj > 0.0 ? ilogb(sqrt(pjj) * 1.001) + 1 + 1022 + scale_shift : 1022;
            }
        }
    } else if (bexp && row == 0) {
        // exact downdate, column scales of B collected by k_dx_partial (atomicMax of the entries' exponents): start from zero
        constexpr int VWz = 16 / sizeof(T);
        const int jz = (bx * 256 + threadIdx.x) * VWz;
#pragma unroll
        for (int v = 0; v < VWz; ++v)
            if (jz + v < n_pad) bexp[jz + v] = 0;
    }
    // 16 bytes per thread: the copy is pure HBM traffic (n_pad and ld are multiples of 128 elements)
    constexpr int VW = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    const int j = (bx * 256 + threadIdx.x) * VW;
    const int m = 2 * M;
    if (row < m) {
        const int i = row >> 1, r = row & 1;
        const int fi = matches[i].featureIndex;
        if (grow) {
            if (bx == 0 && threadIdx.x == 0) grow[row] = 2 * fi + r;
        } else if (j < n_pad) *(vec_t *)(A + (size_t)row * ld + j) = *(const vec_t *)(HP + (size_t)(2 * fi + r) * ld + j);
        if (bx == 0 && threadIdx.x < 16) Gc[(size_t)row * 16 + threadIdx.x] = threadIdx.x < 13 ? HPc[(size_t)(2 * fi + r) * 16 + threadIdx.x] : 0.0;
        if (bx == 0 && r == 0) {
            const int t = threadIdx.x;
            if (t < 14) mHs[14 * i + t] = Hs_tab[14 * fi + t];
            else if (t < 26) mHf[12 * i + t - 14] = Hf_tab[12 * fi + t - 14];
            else if (t == 26) mpos[i] = feat_covpos[fi];
            else if (t == 27) mdim[i] = feat_dim(feat_type[fi]);
            else if (t == 28 || t == 29) {
                const int c = t - 28;
                const double a = matches[i].imagePos[c] - uv_tab[2 * fi + c];
                nu[2 * i + c] = fabs(a) > EKF_DELTA ? a : 0.0;
            }
        }
    } else if (row < m_pad) {
        if (grow) {
            if (bx == 0 && threadIdx.x == 0) grow[row] = -1;
        } else if (j < n_pad) {
            vec_t zero;
#pragma unroll
            for (int v = 0; v < VW; ++v) zero[v] = (T)0;
            *(vec_t *)(A + (size_t)row * ld + j) = zero;
        }
        if (bx == 0 && threadIdx.x < 16) Gc[(size_t)row * 16 + threadIdx.x] = 0.0;
        if (bx == 0 && threadIdx.x == 0) nu[row] = 0.0;
    }
}

template <typename T, typename TP = float>
__global__ void __launch_bounds__(256)
k_gather(const EkfMatch *matches, int M, int m_pad, const T *HP, T *A, int ld, int n_pad, const double *uv_tab,
         const double *Hs_tab, const double *Hf_tab, const int *feat_type, const int *feat_covpos, double *nu,
         double *mHs, double *mHf, int *mpos, int *mdim, const double *HPc, double *Gc, int *bexp, const TP *Pdiag, int ldpd,
         int n, int *grow, int scale_shift)
{
    gather_body<T, TP>((int)blockIdx.x, (int)blockIdx.y, matches, M, m_pad, HP, A, ld, n_pad, uv_tab, Hs_tab, Hf_tab, feat_type, feat_covpos, nu, mHs, mHf,
                       mpos, mdim, HPc, Gc, bexp, Pdiag, ldpd, n, grow, scale_shift);
}

// sum over the four lanes of a quad, in every lane: two DPP quad permutes (lane ^ 1, lane ^ 2) -- register moves, where
// __shfl_xor goes through the LDS crossbar (ds_bpermute: the 28 dependent pairs of the right-hand-side role cost 4 us)
__device__ __forceinline__ double quad_sum(double x)
{
    double y = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xF, 0xF, true),
                                __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xF, 0xF, true)); // quad_perm [1,0,3,2]
    x += y;
    y = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), 0x4E, 0xF, 0xF, true),
                         __builtin_amdgcn_mov_dpp(__double2loint(x), 0x4E, 0xF, 0xF, true)); // quad_perm [2,3,0,1]
    return x + y;
}

// one 32 x kb block of L (rows i0.., columns k0..) from LDS to LL / LLf, see k_chol_step
__device__ __forceinline__ void store_l_block(double *LL, float *LLf, bool full, int ldS, int m_pad, int i0, int k0, int kb,
                                              const double (*sL)[NB + 1])
{
    if (full) {
        for (int i = threadIdx.x; i < NB * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            if (i0 + r < m_pad && c < kb) LL[(size_t)(i0 + r) * ldS + k0 + c] = sL[r][c];
        }
    }
    for (int i = threadIdx.x; i < NB * NB; i += 256) {
        const int c = i / NB, r = i % NB;
        if (i0 + r < m_pad && c < kb) {
            if (full || !LLf) LL[(size_t)(k0 + c) * ldS + i0 + r] = sL[r][c];
            if (LLf) LLf[(size_t)(k0 + c) * ldS + i0 + r] = (float)sL[r][c];
        }
    }
}

// inv(L_kk) goes to V (row-major inv(L)), its transpose to W (and to the fp32 copy of W, when present)
__device__ __forceinline__ void store_linv(double *V, double *W, float *Wf, int ldw, int k0, const double (*x)[NB + 1])
{
    for (int i = threadIdx.x; i < NB * NB; i += 256) {
        const int r = i / NB, c = i % NB;
        V[(size_t)(k0 + r) * ldw + k0 + c] = x[r][c];
        if (!W) continue; // only the inverse + GEMM path reads W
        const double t = x[c][r]; // W[k0 + r][k0 + c] = inv(L)[k0 + c][k0 + r]
        W[(size_t)(k0 + r) * ldw + k0 + c] = t;
        if (Wf) Wf[(size_t)(k0 + r) * ldw + k0 + c] = (float)t;
    }
}

// ------------------------------------------------------------------------------------------------ S = A H' + R
// One thread per 2x2 block (a, b), b <= a, of the lower triangle.
// direct != null (k_gather_assemble): nothing gathered is read -- the Jacobian blocks, positions and the rows of H P come straight
// from the prediction tables through the matches' feature indices (the same numbers as their gathered copies: same S, bit for bit)
struct AssembleDirect {
    const EkfMatch *matches;
    const double *Hs_tab, *Hf_tab;
    const int *feat_type, *feat_covpos;
};
template <typename T>
__device__ __forceinline__ void
assemble_body(const int bx, const int by, const T *A, int ld, int M, const double *mHs, const double *mHf, const int *mpos, const int *mdim,
              double pixel_err, double *S, int ldS, double *V, double *W, float *Wf, int ldw, int *counts, int *lexp, const int *grow,
              int b_lo, int b_hi, double *St, int ldst, int duties, const AssembleDirect *direct)
{
    // [b_lo, b_hi), St: a rank of a row-sharded filter forms the block COLUMNS of its own matches only (it holds G[:, own
    // columns], k_g_cols) and writes them a second time transposed (St: row = column of S) for the exchange; duties: the first
    // diagonal block is factorised here (complete S only)
    const int b = bx * 16 + (threadIdx.x & 15);
    const int a = by * 16 + (threadIdx.x >> 4);
    if (a < M && b <= a && b >= b_lo && b < b_hi) {
        const int fb = direct ? direct->matches[b].featureIndex : 0, fa = direct ? direct->matches[a].featureIndex : 0;
        const int pos = direct ? direct->feat_covpos[fb] : mpos[b], d = direct ? feat_dim(direct->feat_type[fb]) : mdim[b];
        const double *hs = direct ? direct->Hs_tab + 14 * fb : mHs + 14 * b, *hf = direct ? direct->Hf_tab + 12 * fb : mHf + 12 * b;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const T *ar = A + (size_t)(direct ? 2 * fa + r : (grow ? grow[2 * a + r] : 2 * a + r)) * ld; // (grow, direct: A is the H P table itself)
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const double v = (double)ar[k];
                s0 += v * hs[k];
                s1 += v * hs[7 + k];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) { // fixed trip count (d = 3 or 6): the six loads are requested together
                const double v = (double)ar[pos + min(k, d - 1)];
                s0 += k < d ? v * hf[k] : 0.0;
                s1 += k < d ? v * hf[6 + k] : 0.0;
            }
            if (a == b) {
                if (r == 0) s0 += pixel_err;
                else s1 += pixel_err;
            }
            S[(size_t)(2 * a + r) * ldS + 2 * b] = s0;
            S[(size_t)(2 * a + r) * ldS + 2 * b + 1] = s1;
            if (St) {
                St[(size_t)(2 * b) * ldst + 2 * a + r] = s0;
                St[(size_t)(2 * b + 1) * ldst + 2 * a + r] = s1;
            }
            // row scale of L for its digit planes (chol_bplanes.h): sum_k L_rk^2 = S_rr, so |L_rk| <= sqrt(S_rr)
            if (lexp && a == b) lexp[2 * a + r] = ilogb(sqrt(r == 0 ? s0 : s1) * 1.001) + 1 + 1022;
        }
    }
    if (bx != 0 || by != 0 || !duties) return;
    // Block (0, 0) has just written the first 32 x 32 diagonal block of S: factorise it here (what the sweep's
    // look-ahead does for every later block), which saves the separate launch that used to start the sweep.
    __shared__ double sa[NB][NB + 1], sx[NB][NB + 1];
    __threadfence_block();
    __syncthreads();
    const int kb = min(NB, 2 * M);
    for (int i = threadIdx.x; i < NB * NB; i += 256) {
        const int r = i / NB, c = i % NB;
        sa[r][c] = (r < kb && c <= r) ? S[(size_t)r * ldS + c] : ((r == c) ? 1.0 : 0.0);
    }
    __syncthreads();
    if (!block_chol_inv32_v4(sa, sx) && threadIdx.x == 0) counts[CNT_ERR] = EKF_ERR_NOT_POSITIVE_DEFINITE;
    store_linv(V, W, Wf, ldw, 0, sx);
}

template <typename T>
__global__ void __launch_bounds__(256)
k_assemble_S(const T *A, int ld, int M, const double *mHs, const double *mHf, const int *mpos, const int *mdim,
             double pixel_err, double *S, int ldS, double *V, double *W, float *Wf, int ldw, int *counts, int *lexp, const int *grow,
             int b_lo, int b_hi, double *St, int ldst, int duties)
{
    assemble_body<T>((int)blockIdx.x, (int)blockIdx.y, A, ld, M, mHs, mHf, mpos, mdim, pixel_err, S, ldS, V, W, Wf, ldw, counts, lexp, grow, b_lo, b_hi,
                     St, ldst, duties, nullptr);
}

// The gather's bookkeeping (and copy, where there is one) and the assembly of S in ONE launch (one GPU, nothing exchanged between the
// two): workgroups 0 .. mb^2 - 1 assemble the 16 x 16 groups of 2 x 2 blocks (reading the prediction tables directly), the rest are
// the gather's grid, row-major.  One launch fewer per update.
template <typename T, typename TP>
__global__ void __launch_bounds__(256)
k_gather_assemble(int mb, int gxg, const EkfMatch *matches, int M, int m_pad, const T *HP, T *A, int ld, int n_pad, const double *uv_tab,
                  const double *Hs_tab, const double *Hf_tab, const int *feat_type, const int *feat_covpos, double *nu,
                  double *mHs, double *mHf, int *mpos, int *mdim, const double *HPc, double *Gc, int *bexp, const TP *Pdiag, int ldpd,
                  int n, int *grow, double pixel_err, double *S, int ldS, double *V, double *W, float *Wf, int ldw, int *counts, int *lexp,
                  int duties, int scale_shift)
{
    const int id = (int)blockIdx.x;
    if (id < mb * mb) {
        const AssembleDirect dr{matches, Hs_tab, Hf_tab, feat_type, feat_covpos};
        assemble_body<T>(id % mb, id / mb, HP, ld, M, nullptr, nullptr, nullptr, nullptr, pixel_err, S, ldS, V, W, Wf, ldw, counts, lexp, nullptr, 0, M,
                         nullptr, 0, duties, &dr);
    } else {
        const int g = id - mb * mb;
        gather_body<T, TP>(g % gxg, g / gxg, matches, M, m_pad, HP, A, ld, n_pad, uv_tab, Hs_tab, Hf_tab, feat_type, feat_covpos, nu, mHs, mHf, mpos,
                           mdim, HPc, Gc, bexp, Pdiag, ldpd, n, grow, scale_shift);
    }
}

// ------------------------------------------------------------------------- row-sharded filter: G by symmetry, S by columns
// a rank holds the rows of P of its own features, all columns.  G = H P restricted to the rank's OWN columns needs no
// other rank: G[r][c] = sum_j H_r[j] P[j][c] = sum_j H_r[j] P[c][j] (P is bitwise symmetric), i.e. row c of P -- an own row --
// against the 7 + d non-zeros of row r of H.  Columns [c_lo, c_hi) (multiples of 32; the camera block is formed by every rank from
// the replicated camera rows); columns in that range the rank does not own are written as zeros (the 32-column blocks of the rows
// of B straddle the ownership boundaries; their owner's digit planes replace them after the exchange).
// Block: 64 columns x 4 row lanes, 64 rows per block.y.
template <typename TP>
__global__ void __launch_bounds__(256)
k_g_cols(const TP *P, int ldp, RowMap rm, int n, int M, const double *mHs, const double *mHf, const int *mpos, const int *mdim,
         double *G, int ld, int c_lo, int c_hi, int m_pad)
{
    const int c = c_lo + blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    if (c >= c_hi) return;
    const bool valid = c < n && owns_row(rm, c);
    const TP *pr = P + (size_t)(valid ? local_row(rm, c) : 0) * ldp;
    double pc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) pc[k] = valid ? (double)pr[k] : 0.0;
    const int row0 = blockIdx.y * 64;
    for (int q = 0; q < 16; ++q) {
        const int row = row0 + 4 * q + rl;
        if (row >= m_pad) break;
        double g = 0.0;
        if (valid && row < 2 * M) {
            const int i = row >> 1, a = row & 1;
            const int pos = mpos[i], d = mdim[i];
            const double *hs = mHs + 14 * i + 7 * a, *hf = mHf + 12 * i + 6 * a;
#pragma unroll
            for (int k = 0; k < 7; ++k) g += hs[k] * pc[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) g += k < d ? hf[k] * (double)pr[pos + min(k, d - 1)] : 0.0;
        }
        G[(size_t)row * ld + c] = g;
    }
}

// After the exchange of St (row cb = column cb of S, all ranks' columns complete): the columns of the other ranks into S (lower
// triangle) and the row scales of L of their rows (k_assemble_S wrote the own ones).  grid (row chunks of 256, 2 M columns)
__global__ void __launch_bounds__(256)
k_s_unpack(const double *St, int ldst, double *S, int ldS, int m, int own_lo, int own_hi, int *lexp)
{
    const int cb = blockIdx.y, ra = blockIdx.x * 256 + threadIdx.x;
    if (cb >= own_lo && cb < own_hi) return;
    if (ra >= m || ra < cb) return;
    const double v = St[(size_t)cb * ldst + ra];
    S[(size_t)ra * ldS + cb] = v;
    if (lexp && ra == cb) lexp[ra] = ilogb(sqrt(v) * 1.001) + 1 + 1022;
}

#include "chol_pair.h" // the two-panels-per-launch variant of the sweep below, and the 32^3 product helpers both use
#include "chol_bplanes.h" // EKF_PRECISION_F32_EXACT: the rows of B from int8 digit planes
#include "chol_persist.h" // the whole sweep as one persistent launch

// -------------------------------------------------------------------------------------- blocked Cholesky sweep
// Right-looking sweep over [ S | nu ], panel width NB = 32, ONE launch per panel (k_chol_step).  The only serial
// piece is the 32x32 diagonal block: its Cholesky factor and the inverse of that factor.  It is computed by one
// workgroup with all 256 threads working in LDS (block_chol_inv32_v4 in chol32.h: 4x4 block pivots, rank-4 updates on
// the fp64 MFMA, one barrier per block column) and -- look-ahead -- inside the launch of the PREVIOUS panel, by the
// workgroup that owns tile (k+1, k+1): while the other workgroups of that launch do their work, this one finishes its
// tile, factorises it and stores inv(L_{k+1,k+1}) into V.  That workgroup needs ~7.5 us and one CU; everything else a
// panel has to do is arranged to fit in its shadow on the other 255 (scripts/sweep_trace.py shows when each role ends):
//
//   grid.x, one launch per panel k, Linv_k = inv(L_kk) already in V (look-ahead of the previous launch):
//     [0]                the look-ahead workgroup: tile (k+1, k+1) updated, factorised, its inverse published
//     [1 .. n_bblocks]   rows of B (m <= B_SWEEP_MAX): B_k = Linv_k (G_k - sum_{j<k} L_kj B_j), one 32x32 block each
//     [.. + n_rhs]       right-hand sides: z_k = Linv_k nu_k (and the fp64 camera columns Bc_k), then for 64 rows each
//                        nu_i -= L_ik z_k = S_ik (Linv_k' z_k)
//     [the rest]         the trailing tiles in 2 x 2 groups: S_ij -= L_ik L_jk' with L_ik = S_ik Linv_k' formed in the
//                        group (the panel solve is not a launch of its own); the groups of the first group column also
//                        store L_ik into LL / LLf (mirrored: L' above the diagonal blocks, the A operand of the rows of
//                        B; on the inverse + GEMM path also row-major, the operands of the inverse's doubling levels)
//   S itself is only read in column k, B only in rows above k, nu / Gc only in rows k: nothing races inside a launch.
// Above B_SWEEP_MAX rows B = inv(L) G is not part of the sweep: the factor is inverted explicitly (k_inv_diag,
// k_triinv_level) and B is one GEMM against it (k_xty, kernels_gemm.hip).
// (two workgroups per CU: the early panels of a sweep have more workgroups than the chip has CUs -- tile groups + rows of B --
// and the second round must not wait for the first: 14 against 8 us per launch on the first eight panels of an m = 920 sweep)
template <typename T, typename TG> // T: type of B and of its MFMA; TG: storage type of the gathered rows G
__global__ void __launch_bounds__(256, 2)
k_chol_step(double *S, double *LL, float *LLf, int ldS, int m, int m_pad, int k0, int kb, double *nu, int n_stiles, double *V,
            double *W, float *Wf, int ldw, int *counts, double *Gc, double *zout, double *Bc, const TG *G, T *Bout, int ld,
            int n_bblocks, int n_rhs, int tiles_first, int spacer, unsigned long long *trace, int abl, BPlanes bp)
{
#ifdef EKF_SWEEP_TRACE // debug builds only (scripts/sweep_trace.py): per-role time stamps and role ablations
    const unsigned long long t_in = trace ? wall_clock64() : 0ull;
#define SWEEP_TRACE(role)                                                                      \
    if (trace && threadIdx.x == 0) {                                                           \
        atomicMin(trace, t_in);                                                                \
        atomicMax(trace + 1 + (role), (unsigned long long)wall_clock64());                     \
        if ((role) == 2) atomicMax(trace + 6, t_in);                                           \
        if ((role) == 3) atomicMax(trace + 7, t_in);                                           \
    }
#else
#define SWEEP_TRACE(role)
    (void)trace;
    (void)abl;
#endif
    constexpr int NR = 14; // right-hand sides: nu + 13 camera columns
    // tile role: S_ik, S_jk, L_ik, L_jk; B role: two partial sums, G_k - sum; right-hand sides: R, Z, W
    __shared__ double pool[4][NB][NB + 1];
    __shared__ double sLi[NB][NB + 1];       // inv(L_kk)
    double(*sA)[NB + 1] = pool[0];
    double(*sB)[NB + 1] = pool[1];
    const int tid = threadIdx.x;
    const int k1 = k0 + kb;
    // block order: the look-ahead workgroup first.  A launch that fits one workgroup per CU: then the row block of B (the
    // longest units of a late panel; a column block keeps its place in the order, hence its XCD and the L2 that holds its
    // rows of B), the right-hand sides, the tile groups.  A launch with more workgroups than CUs (early panels): the
    // tile groups before B, so that the workgroup which ends up sharing the look-ahead workgroup's CU is a short B unit
    // and not a tile group (the look-ahead publishes 1.5 us later beside one: 9.1 against 7.5 us).
    int b = blockIdx.x;
    int bcol = -1;
    if (spacer > 0) {
        // Workgroups are handed to the CUs in turn, so block `spacer` (= the number of CUs), 2 spacer, ... would land on
        // the CU of block 0, the look-ahead workgroup, which then publishes 1.5 us later.  Those blocks are empty.
        if (b > 0 && b % spacer == 0) return;
        b -= b / spacer;
    }
    if (tiles_first) { // [look-ahead][tile groups][right-hand sides][B]: b is already the tile / right-hand-side index
        if (b >= n_stiles + n_rhs) {
            bcol = b - n_stiles - n_rhs;
            b = n_stiles + 1; // not a tile
        }
    } else {
        const int f = n_stiles > 0 ? 1 : 0;
        if (b >= f && b < f + n_bblocks) {
            bcol = b - f;
            b = n_stiles + 1; // not a tile
        } else if (b >= f + n_bblocks && b < f + n_bblocks + n_rhs) {
            b = n_stiles + (b - f - n_bblocks);
        } else if (b >= f + n_bblocks + n_rhs) {
            b -= n_bblocks + n_rhs;
        }
    }
#ifdef EKF_SWEEP_TRACE
    if (abl) { // timing experiments only (scripts/sweep_trace.py): a role returns at once -- WRONG filter results
        if ((abl & 1) && bcol >= 0) return;
        if ((abl & 2) && bcol < 0 && b >= n_stiles) return;
        if ((abl & 4) && bcol < 0 && b > 0 && b < n_stiles) return;
    }
#endif
    // every global load of the prologue is issued before the first LDS store waits on one of them (the look-ahead
    // workgroup's inputs are cold: they were written by the previous launch on other XCDs)
    double gv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = tid + q * 256;
        gv[q] = V[(size_t)(k0 + i / NB) * ldw + k0 + i % NB];
    }
    // (gv lands in sLi inside each role, after the role's own loads have been requested: one round trip, not two)
    if constexpr (sizeof(T) == 8 && sizeof(TG) == 8) {
        if (bcol >= 0 && bp.Bq) { // EKF_PRECISION_F32_EXACT: the same role from int8 digit planes (chol_bplanes.h)
            b_rows_planes(bp, (const double *)G, (double *)Bout, ld, m, k0, bcol + bp.bcol0, pool, sLi, gv);
            SWEEP_TRACE(1)
            return;
        }
    }
    if (bcol >= 0) {
        // Row block k of B = inv(L) G, columns 32 bcol ..: B_k = inv(L_kk) (G_k - sum_{j<k} L_kj B_j) -- every operand
        // is final when this launch starts (L_kj: stored by the panels before, B_j: by their launches), so the forward
        // substitution costs no launch of its own and runs beside the look-ahead factorisation, which leaves most CUs
        // idle.  One 32 x 32 block per workgroup, the j-range dealt to the four wavefronts (their partial sums meet
        // in LDS); operands go straight from L2 into the MFMA operand layout: the mirrored copy of L (L' above the
        // diagonal blocks of LL) makes the lanes of an A operand read consecutive addresses.
        using M = Mma<T>;
        constexpr int MB = M::MB, NBLK = NB / MB, KS = 64 / MB, NSTEP = NB / KS;
        T(*red)[NB][NB + 1] = reinterpret_cast<T(*)[NB][NB + 1]>(&pool[0][0][0]); // two partial sums (pool[0..1] at most)
        double(*sR)[NB + 1] = pool[2];
        const int lane = tid & 63, wv = tid >> 6;
        const int lm = lane % MB, lq = lane / MB;
        const int c0 = bcol * NB;
        const int kp = k0 / NB; // panels before this one
        const T *Lt = sizeof(T) == 4 ? (const T *)LLf : (const T *)LL; // L' above the diagonal blocks, in T
        // this thread's four elements of G_k, requested first: they are cold and only needed at the end
        const int r = tid >> 3, cg = (tid & 7) * 4;
        TG g4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) g4[e] = G[(size_t)(k0 + r) * ld + c0 + cg + e];
        typename M::acc_t acc[NBLK][NBLK];
#pragma unroll
        for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
            for (int bj = 0; bj < NBLK; ++bj)
#pragma unroll
                for (int r = 0; r < M::NACC; ++r) acc[bi][bj][r] = (T)0;
        // the wavefront's blocks j = wv, wv + 4, ...: operands of the next two blocks are in flight while one is multiplied
        // (a block's MFMAs take 0.45 us, a load round trip under load over 1 us)
        T la[3][NSTEP][NBLK], lb[3][NSTEP][NBLK];
#define CB_LOAD(S_, J_)                                                                                           \
    _Pragma("unroll") for (int st = 0; st < NSTEP; ++st) {                                                        \
        const size_t kr = (size_t)(J_) * NB + st * KS + lq;                                                       \
        _Pragma("unroll") for (int bb = 0; bb < NBLK; ++bb) {                                                     \
            la[S_][st][bb] = Lt[kr * ldS + k0 + MB * bb + lm];                                                    \
            lb[S_][st][bb] = Bout[kr * ld + c0 + MB * bb + lm];                                                   \
        }                                                                                                         \
    }
#define CB_MMA(S_)                                                                                                \
    _Pragma("unroll") for (int st = 0; st < NSTEP; ++st)                                                          \
        _Pragma("unroll") for (int bi = 0; bi < NBLK; ++bi)                                                       \
            _Pragma("unroll") for (int bj = 0; bj < NBLK; ++bj)                                                   \
                acc[bi][bj] = M::mma(la[S_][st][bi], lb[S_][st][bj], acc[bi][bj]);
        const int cnt = kp > wv ? (kp - wv + 3) / 4 : 0;
        if (cnt > 0) { CB_LOAD(0, wv) }
        if (cnt > 1) { CB_LOAD(1, wv + 4) }
        for (int i = 0; i < cnt; i += 3) {
            if (i + 2 < cnt) { CB_LOAD(2, wv + 4 * (i + 2)) }
            CB_MMA(0)
            if (i + 1 < cnt) {
                if (i + 3 < cnt) { CB_LOAD(0, wv + 4 * (i + 3)) }
                CB_MMA(1)
            }
            if (i + 2 < cnt) {
                if (i + 4 < cnt) { CB_LOAD(1, wv + 4 * (i + 4)) }
                CB_MMA(2)
            }
        }
#undef CB_LOAD
#undef CB_MMA
#pragma unroll
        for (int q = 0; q < 4; ++q) sLi[(tid + q * 256) / NB][(tid + q * 256) % NB] = gv[q];
        // partial sums: wavefronts 2, 3 through LDS to wavefronts 0, 1, whose sums meet in red[0..1]
        if (wv >= 2) {
#pragma unroll
            for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
                for (int bj = 0; bj < NBLK; ++bj)
#pragma unroll
                    for (int r = 0; r < M::NACC; ++r) red[wv - 2][MB * bi + M::row(r, lane)][MB * bj + M::col(lane)] = acc[bi][bj][r];
        }
        __syncthreads();
        if (wv < 2) {
#pragma unroll
            for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
                for (int bj = 0; bj < NBLK; ++bj)
#pragma unroll
                    for (int r = 0; r < M::NACC; ++r) red[wv][MB * bi + M::row(r, lane)][MB * bj + M::col(lane)] += acc[bi][bj][r];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e)
            sR[r][cg + e] = (double)g4[e] - ((double)red[0][r][cg + e] + (double)red[1][r][cg + e]);
        __syncthreads();
        {   // B_k = Linv_k R on the fp64 MFMA, one 16 x 16 quadrant per wavefront (as a scalar loop of LDS reads this product
            // took 1.5 of the role's 8-9 us on the late panels, where the rows of B outlast the look-ahead workgroup)
            const int bi = wv >> 1, bj = wv & 1, lr = lane & 15, lk = lane >> 4;
            const acc4_t o = quad_prod<false>(acc4_t{0, 0, 0, 0}, sLi, sR, bi, bj, lr, lk);
#pragma unroll
            for (int q = 0; q < 4; ++q) Bout[(size_t)(k0 + 16 * bi + lk + 4 * q) * ld + c0 + 16 * bj + lr] = (T)o[q];
        }
        SWEEP_TRACE(1)
        return;
    }
    if (b > 0 && b < n_stiles) {
        // Super-tile (TI, TJ): the 2 x 2 group of 32 x 32 tiles with block rows 2 TI, 2 TI + 1 and block columns 2 TJ, 2 TJ + 1
        // of the trailing matrix (lower triangle; block (0, 0) is the look-ahead workgroup's).  One wavefront per tile.
        // The four panel blocks L_ik = S_ik Linv' the group needs (two on the diagonal) are formed once, one per
        // wavefront, and overwrite S_ik in LDS: five blocks loaded and four products for four tiles, where one tile per
        // workgroup loaded four and made two -- with hundreds of tiles in flight that traffic was the early panels' tail.
        typedef double acc4 __attribute__((ext_vector_type(4)));
        const int sidx = b - 1;
        int TI = (int)((sqrt(8.0 * sidx + 1.0) - 1.0) * 0.5);
        while ((TI + 1) * (TI + 2) / 2 <= sidx) ++TI;
        while (TI * (TI + 1) / 2 > sidx) --TI;
        const int TJ = sidx - TI * (TI + 1) / 2;
        const bool diag = TI == TJ;
        const int lane = tid & 63, wv = tid >> 6;
        const int lr = lane & 15, lk = lane >> 4;
        const int qi = wv >> 1, qj = wv & 1; // this wavefront's tile of the group
        const int i0 = k1 + (2 * TI + qi) * NB, j0 = k1 + (2 * TJ + qj) * NB;
        const bool tile_live = i0 < m && j0 <= i0 && !(TI == 0 && TJ == 0 && qi == 0 && qj == 0);
        double(*sP)[NB][NB + 1] = pool; // [0], [1]: rows 2 TI, 2 TI + 1; [2], [3]: columns 2 TJ, 2 TJ + 1 (off the diagonal)
        // the tile's own values (accumulator layout of four 16 x 16 blocks), then the panel blocks
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
        double gp[4][4];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int r0 = k1 + (blk < 2 ? 2 * TI + blk : 2 * TJ + blk - 2) * NB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + q * 256, r = i / NB, c = i % NB;
                gp[blk][q] = ((blk < 2 || !diag) && r0 + r < m && c < kb) ? S[(size_t)(r0 + r) * ldS + k0 + c] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256, r = i / NB, c = i % NB;
            sLi[r][c] = gv[q];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
                if (blk < 2 || !diag) sP[blk][r][c] = gp[blk][q];
        }
        __syncthreads();
        // wavefront w: block w of the panel, L = S_blk Linv', written back over S_blk (only this wavefront reads it)
        if (wv < 2 || !diag) {
            acc4 c[2][2];
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj) c[bi][bj] = acc4{0, 0, 0, 0};
#pragma unroll
            for (int k4 = 0; k4 < NB; k4 += 4) {
                const double a0 = sP[wv][lr][k4 + lk], a1 = sP[wv][16 + lr][k4 + lk];
                const double l0 = sLi[lr][k4 + lk], l1 = sLi[16 + lr][k4 + lk]; // b[k][c] = Linv[c][k]
                c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, l0, c[0][0], 0, 0, 0);
                c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, l1, c[0][1], 0, 0, 0);
                c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, l0, c[1][0], 0, 0, 0);
                c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, l1, c[1][1], 0, 0, 0);
            }
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) sP[wv][16 * bi + lk + 4 * q][16 * bj + lr] = c[bi][bj][q];
        }
        __syncthreads();
        if (tile_live) {
            const double(*Li)[NB + 1] = sP[qi];
            const double(*Lj)[NB + 1] = sP[diag ? qj : 2 + qj];
            acc4 c[2][2];
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj) c[bi][bj] = acc4{0, 0, 0, 0};
#pragma unroll
            for (int k4 = 0; k4 < NB; k4 += 4) {
                const double a0 = Li[lr][k4 + lk], a1 = Li[16 + lr][k4 + lk];
                const double b0 = Lj[lr][k4 + lk], b1 = Lj[16 + lr][k4 + lk];
                c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c[0][0], 0, 0, 0);
                c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c[0][1], 0, 0, 0);
                c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c[1][0], 0, 0, 0);
                c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c[1][1], 0, 0, 0);
            }
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = i0 + 16 * bi + lk + 4 * q, cc = j0 + 16 * bj + lr;
                        if (r < m && cc <= r) S[(size_t)r * ldS + cc] = v[bi][bj][q] - c[bi][bj][q];
                    }
        }
        // L_ik leaves the groups of the first group column (see store_l_block), the look-ahead workgroup's block with it
        if (TJ == 0) {
            store_l_block(LL, LLf, W != nullptr, ldS, m_pad, k1 + 2 * TI * NB, k0, kb, sP[0]);
            store_l_block(LL, LLf, W != nullptr, ldS, m_pad, k1 + (2 * TI + 1) * NB, k0, kb, sP[1]);
            store_l_planes(bp, m, 2, k1 + 2 * TI * NB, sP[0], k1 + (2 * TI + 1) * NB, sP[1], k0, kb);
        }
        SWEEP_TRACE(2)
        return;
    }
    if (b < n_stiles) { // b == 0: the look-ahead workgroup, tile (0, 0) of the trailing matrix
        double(*sLI)[NB + 1] = pool[2];
        double(*sLJ)[NB + 1] = pool[3];
        const int ti = 0, tj = 0;
        const int i0 = k1, j0 = k1;
        // Each wavefront owns one 16x16 block (bi, bj) of the 32x32 tile; the three 32^3 products run on the fp64 MFMA
        // (v_mfma_f64_16x16x4: lane l feeds a[l % 16][l / 16] and b[l / 16][l % 16], holds rows (l >> 4) + 4 v of column
        // l & 15) -- as plain FMA loops they were LDS-bound and cost 4 of the 19 us of this launch's critical path.
        typedef double acc4 __attribute__((ext_vector_type(4)));
        const int lane = tid & 63, wv = tid >> 6;
        const int bi = wv >> 1, bj = wv & 1;
        const int lr = lane & 15, lk = lane >> 4;
        const bool diag_tile = ti == tj;
        // the tile's own values first (accumulator layout): their latency hides behind the panel products
        double v[4];
        bool live[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
            live[q] = i0 + r < m && j0 + c < m && j0 + c <= i0 + r;
            v[q] = live[q] ? S[(size_t)(i0 + r) * ldS + j0 + c] : 0.0;
        }
        double ga[4], gb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256;
            const int r = i / NB, c = i % NB;
            ga[q] = (i0 + r < m && c < kb) ? S[(size_t)(i0 + r) * ldS + k0 + c] : 0.0;
            gb[q] = (!diag_tile && j0 + r < m && c < kb) ? S[(size_t)(j0 + r) * ldS + k0 + c] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256;
            const int r = i / NB, c = i % NB;
            sLi[r][c] = gv[q];
            sA[r][c] = ga[q];
            if (!diag_tile) sB[r][c] = gb[q];
        }
        __syncthreads();
        // L_ik = S_ik Linv', L_jk = S_jk Linv'
        {
            acc4 ci = {0, 0, 0, 0}, cj = {0, 0, 0, 0};
#pragma unroll
            for (int k4 = 0; k4 < NB; k4 += 4) {
                const double l = sLi[16 * bj + lr][k4 + lk]; // b[k][c] = Linv[c][k]
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[16 * bi + lr][k4 + lk], l, ci, 0, 0, 0);
                if (!diag_tile) cj = __builtin_amdgcn_mfma_f64_16x16x4f64(sB[16 * bi + lr][k4 + lk], l, cj, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sLI[16 * bi + lk + 4 * q][16 * bj + lr] = ci[q];
                sLJ[16 * bi + lk + 4 * q][16 * bj + lr] = diag_tile ? ci[q] : cj[q];
            }
        }
        __syncthreads();
        {
            acc4 cu = {0, 0, 0, 0};
#pragma unroll
            for (int k4 = 0; k4 < NB; k4 += 4)
                cu = __builtin_amdgcn_mfma_f64_16x16x4f64(sLI[16 * bi + lr][k4 + lk], sLJ[16 * bj + lr][k4 + lk], cu, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (live[q]) {
                    v[q] -= cu[q];
                }
        }
        if (b == 0) {
            // look-ahead: this workgroup owns block (k+1, k+1); finish it in LDS, factorise, publish its inverse
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
                const bool lv = (k1 + r < m) && (c <= r);
                sA[r][c] = lv ? v[q] : ((r == c) ? 1.0 : 0.0);
            }
            __syncthreads();
            if (!block_chol_inv32_v4(sA, sB) && tid == 0) counts[CNT_ERR] = EKF_ERR_NOT_POSITIVE_DEFINITE;
            store_linv(V, W, Wf, ldw, k1, sB);
            SWEEP_TRACE(0)
        }
        // L_ik leaves the tiles of the first tile column: mirrored (L' above the diagonal blocks; zero rows m..m_pad) for
        // the rows of B -- in fp32 when the covariance is --, and with W set (inverse + GEMM path) also row-major and in
        // fp64 for the inverse's levels.  The look-ahead workgroup's own block is stored by the first group of tiles, which
        // holds the same block, when there is one: the look-ahead workgroup is the one the launch waits for.
        if (n_stiles == 1) {
            store_l_block(LL, LLf, W != nullptr, ldS, m_pad, i0, k0, kb, sLI);
            store_l_planes(bp, m, 1, i0, sLI, i0, sLI, k0, kb);
        }
        SWEEP_TRACE(2)
        return;
    }
    // right-hand-side blocks: [ nu | Gc ] (Gc = fp64 camera columns of the gathered H P rows, fp32 configuration only) are
    // carried through the sweep like extra columns of S:  Z_k = Linv_k R_k (final rows of z and of Bc = inv(L) Gc), then
    // R_i -= L_ik Z_k = S_ik (Linv_k' Z_k) for the rows below.  Block nb handles rows k1 + 64 nb ...; every block forms
    // the 32 x 14 matrices itself (a 32^2 x 14 product) from the rows k0.. of the WORKING arrays nu / Gc, which nobody
    // writes in this launch; block 0 stores Z_k into the RESULT arrays zout / Bc (writing it back in place would race
    // with the other blocks' reads of those rows).
    constexpr int NRP = 16; // right-hand sides padded to one MFMA block
    double(*sR)[NRP + 1] = reinterpret_cast<double(*)[NRP + 1]>(&pool[0][0][0]);
    double(*sZ)[NRP + 1] = reinterpret_cast<double(*)[NRP + 1]>(&pool[1][0][0]);
    double(*sW)[NRP + 1] = reinterpret_cast<double(*)[NRP + 1]>(&pool[2][0][0]);
    const int nb = b - n_stiles;
    const int nrhs = Gc ? NR : 1;
    // the panel's rows of [ nu | Gc ] first (the products below wait for them), then this block's rows of S: 64 rows per
    // block, four lanes to a row (each a quarter of the row's 32 columns: the lanes of a wavefront read 16 whole rows,
    // 256 contiguous bytes each), in flight during the products
    double rv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = tid + q * 256, r = idx / NRP, c = idx % NRP;
        rv[q] = 0.0;
        if (r < kb && c < nrhs) rv[q] = c == 0 ? nu[k0 + r] : Gc[(size_t)(k0 + r) * 16 + c - 1];
    }
    const int i = k1 + nb * 64 + (tid >> 2), part = tid & 3;
    double sv[8];
    {
        const double *srow = S + (size_t)min(i, m - 1) * ldS + k0 + 8 * part;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            const double2 t = *(const double2 *)(srow + q);
            sv[q] = i < m ? t.x : 0.0;
            sv[q + 1] = i < m ? t.y : 0.0;
        }
    }
    // ... and the values this lane will update (lane `part` of a row owns the right-hand sides part, part + 4, ...)
    double old[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = part + 4 * q;
        old[q] = 0.0;
        if (i < m && c < nrhs) old[q] = c == 0 ? nu[i] : Gc[(size_t)i * 16 + c - 1];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) sLi[(tid + q * 256) / NB][(tid + q * 256) % NB] = gv[q];
#pragma unroll
    for (int q = 0; q < 2; ++q) sR[(tid + q * 256) / NRP][(tid + q * 256) % NRP] = rv[q];
    __syncthreads();
    // Z = Linv_k R_k and W = Linv_k' Z on the fp64 MFMA (32 x 32 x 16: one 16 x 16 block per wavefront 0, 1; as scalar
    // loops of dependent LDS reads the two products took 5 of this block's 9 us)
    typedef double acc4 __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wv = tid >> 6, lm = lane & 15, lq = lane >> 4;
    if (wv < 2) {
        acc4 z = {0, 0, 0, 0};
#pragma unroll
        for (int k4 = 0; k4 < NB; k4 += 4) z = __builtin_amdgcn_mfma_f64_16x16x4f64(sLi[16 * wv + lm][k4 + lq], sR[k4 + lq][lm], z, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * wv + lq + 4 * q;
            sZ[r][lm] = r < kb ? z[q] : 0.0;
        }
    }
    __syncthreads();
    if (wv < 2) {
        acc4 w = {0, 0, 0, 0};
#pragma unroll
        for (int k4 = 0; k4 < NB; k4 += 4) w = __builtin_amdgcn_mfma_f64_16x16x4f64(sLi[k4 + lq][16 * wv + lm], sZ[k4 + lq][lm], w, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) sW[16 * wv + lq + 4 * q][lm] = w[q];
    } else if (nb == 0) {
        for (int idx = tid - 128; idx < NB * NR; idx += 128) {
            const int r = idx / NR, c = idx % NR;
            if (r < kb && c < nrhs) {
                if (c == 0) zout[k0 + r] = sZ[r][0];
                else Bc[(size_t)(k0 + r) * 16 + c - 1] = sZ[r][c];
            }
        }
    }
    __syncthreads();
    {
        double acc[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) acc[c] = 0.0;
        if (Gc) { // one branch around branch-free loops: with the test inside, every LDS read waited for the one before it
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                double wrow[NR];
#pragma unroll
                for (int c = 0; c < NR; ++c) wrow[c] = sW[8 * part + q][c];
#pragma unroll
                for (int c = 0; c < NR; ++c) acc[c] += sv[q] * wrow[c];
            }
#pragma unroll
            for (int c = 0; c < NR; ++c) acc[c] = quad_sum(acc[c]);
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[0] += sv[q] * sW[8 * part + q][0];
            acc[0] = quad_sum(acc[0]);
        }
        if (i < m) {
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                if ((c & 3) != part) continue; // the four lanes of a row share its stores
                if (c == 0) nu[i] = old[0] - acc[0];
                else if (Gc) Gc[(size_t)i * 16 + c - 1] = old[c >> 2] - acc[c];
            }
        }
    }
    SWEEP_TRACE(3)
#undef SWEEP_TRACE
}

// Debug aid (scripts/sweep_trace.py), compiled only with -DEKF_SWEEP_TRACE (EKF_EXTRA_FLAGS of openekfmonoslam_amd/build.py):
// per launch of the sweep, the earliest workgroup start and the latest end of each role (0 look-ahead published, 1 row
// block of B, 2 tiles, 3 right-hand sides), in 10 ns ticks of the constant clock.  The state below is process-global and
// the ablation bits (enable >> 8) make roles return early: it is NOT part of the product library.
constexpr int TRACE_SLOTS = 16, TRACE_MAX = 4096;
#ifdef EKF_SWEEP_TRACE
static unsigned long long *g_trace = nullptr;
static int g_trace_n = 0;
static int g_trace_abl = 0;
extern "C" int ekf_debug_sweep_trace(int enable, unsigned long long *out, int *count)
{
    g_trace_abl = enable >> 8;
    if (enable && !g_trace) {
        if (hipMalloc(&g_trace, sizeof(unsigned long long) * TRACE_SLOTS * TRACE_MAX) != hipSuccess) return -1;
    }
    if (out && g_trace) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(out, g_trace, sizeof(unsigned long long) * TRACE_SLOTS * g_trace_n, hipMemcpyDeviceToHost);
        if (count) *count = g_trace_n;
    }
    if (g_trace) {
        std::vector<unsigned long long> init(TRACE_SLOTS * TRACE_MAX, 0ull);
        for (int i = 0; i < TRACE_MAX; ++i) init[(size_t)i * TRACE_SLOTS] = ~0ull;
        (void)hipMemcpy(g_trace, init.data(), sizeof(unsigned long long) * init.size(), hipMemcpyHostToDevice);
    }
    g_trace_n = 0;
    if (!enable && g_trace) { (void)hipFree(g_trace); g_trace = nullptr; }
    return 0;
}
#endif

// ------------------------------------------------------------------------------- inverse of a diagonal chunk of L
// X_cc = inverse of the 128 x 128 diagonal block of L that a chunk of 4 panels covers, by block columns:
//     X_jj = inv(L_jj)  (published by the sweep),   X_aj = -inv(L_aa) sum_{c=j}^{a-1} L_ac X_cj   for a > j.
// Block column j is independent of the others and row a of it needs only the rows above it of the SAME column: one
// workgroup per column (3 at most), its rows in sequence, the X_aj kept in LDS for the rows after them.  Each wavefront
// owns one 16x16 quadrant of the 32x32 blocks; products on the fp64 MFMA, operands fetched straight into the MFMA operand
// layout (lane l: a[l % 16][l / 16], b[l / 16][l % 16]).  Latency is all that matters here (the launch sits between the
// sweep and the chunk's solve): EVERY global operand the column can need -- six L tiles, three inv(L_aa), X_jj -- is
// requested unconditionally before the first product (one round trip instead of one per product: 14 -> ~4 us); tiles of
// block rows past the chunk's end are read (allocated, finite) and never used.  V = inv(L) row-major, W = V' (+ fp32 copy).
// One launch for ALL chunks (blockIdx.y = chunk) right after the sweep: it replaces the first two doubling levels
// (32 -> 64 -> 128: four dependent launches of k_triinv_level) by one.
constexpr int INV_CH = 4; // block rows per chunk

__global__ void __launch_bounds__(256)
k_inv_diag(const double *LL, int ldS, double *V, double *W, float *Wf, int ldw, int nbk)
{
    const int a_first = blockIdx.y * INV_CH, a_count = min(INV_CH, nbk - a_first);
    if ((int)blockIdx.x >= a_count - 1) return; // a column needs at least one block row below its diagonal block
    typedef double acc4 __attribute__((ext_vector_type(4)));
    __shared__ double sX[INV_CH][NB][NB + 1]; // X_aj of the chunk's rows (operand of the rows below them)
    __shared__ double sT[NB][NB + 1];
    const int jl = blockIdx.x; // local column block 0..2
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qi = wv >> 1, qj = wv & 1, lr = lane & 15, lk = lane >> 4;
    // operands: L_{al,cl} (A layout) for 1 <= al <= 3, cl < al; inv(L_aa) (A layout) for al = 1..3; X_jj (B layout)
    double pa[3][3][8], da[3][8], pb[8];
#pragma unroll
    for (int al = 1; al < INV_CH; ++al) {
#pragma unroll
        for (int cl = 0; cl < al; ++cl) {
            const double *Lp = LL + (size_t)(NB * (a_first + al) + 16 * qi + lr) * ldS + NB * (a_first + cl) + lk;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) pa[al - 1][cl][kk] = Lp[4 * kk];
        }
        const double *Dp = V + (size_t)(NB * (a_first + al) + 16 * qi + lr) * ldw + NB * (a_first + al) + lk;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) da[al - 1][kk] = Dp[4 * kk];
    }
    {
        const double *Xp = V + (size_t)(NB * (a_first + jl) + lk) * ldw + NB * (a_first + jl) + 16 * qj + lr;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) pb[kk] = Xp[(size_t)4 * kk * ldw];
    }
#pragma unroll
    for (int al = 1; al < INV_CH; ++al) {
        if (al <= jl || al >= a_count) continue; // uniform: rows at or above the column's diagonal block, rows past the chunk
        acc4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int cl = 0; cl < al; ++cl) {
            if (cl < jl) continue;
            if (cl == jl) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[al - 1][cl][kk], pb[kk], acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[al - 1][cl][kk], sX[cl][4 * kk + lk][16 * qj + lr], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) sT[16 * qi + lk + 4 * v][16 * qj + lr] = acc[v];
        __syncthreads();
        acc4 x = {0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(da[al - 1][kk], sT[4 * kk + lk][16 * qj + lr], x, 0, 0, 0);
        const int a = a_first + al, j = a_first + jl;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 16 * qi + lk + 4 * v, c = 16 * qj + lr;
            const double val = -x[v];
            sX[al][r][c] = val;
            V[(size_t)(NB * a + r) * ldw + NB * j + c] = val;
            W[(size_t)(NB * j + c) * ldw + NB * a + r] = val;
            if (Wf) Wf[(size_t)(NB * j + c) * ldw + NB * a + r] = (float)val;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ inverse of L
// Doubling step s -> 2s:  inv([L11 0; L21 L22]) has the off-diagonal block X21 = -X22 L21 X11.  Two batched products
// per level (T = L21 X11, X21 = -X22 T), 32x32 output tile per workgroup on the fp64 MFMA.  The levels start at s = 128
// (k_inv_diag inverts the 128 x 128 diagonal chunks in one launch).  V = inv(L) row-major, W = V' (+ fp32 copy): after the
// last level B = inv(L) A is ONE GEMM, B = W' A, with no dependency between row blocks.
// mode 0: T[pair] = L21 * X11 ;  mode 1: X21 = -X22 * T
__global__ void __launch_bounds__(256)
k_triinv_level(const double *S, int ldS, int m, int m_pad, double *V, double *W, float *Wf, double *Tbuf, int ldw,
               int s, int mode)
{
    __shared__ double sA[NB][NB + 1];
    __shared__ double sB[NB][NB + 1];
    const int tiles = s / NB;
    const int pair = blockIdx.x / (tiles * tiles);
    const int t = blockIdx.x % (tiles * tiles);
    const int tr = t / tiles, tc = t % tiles;
    const int r0 = pair * 2 * s; // first row of the pair's 2s x 2s diagonal block
    if (r0 + s + tr * NB >= m_pad) return; // no such rows in the second half
    const int tid = threadIdx.x;
    // 16x16 block (bi, bj) of the 32x32 output tile per wavefront, products on the fp64 MFMA (as in k_chol_step)
    typedef double acc4 __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wv = tid >> 6;
    const int bi = wv >> 1, bj = wv & 1, lr = lane & 15, lk = lane >> 4;
    acc4 acc = {0, 0, 0, 0};
    // k-chunks that can be non-zero: X11 and X22 are lower triangular.
    //   mode 0: X11[kk + r][tc*32 + c] = 0 when kk + 31 < tc*32      -> start at chunk tc
    //   mode 1: X22[tr*32 + r][kk + c] = 0 when kk > tr*32 + 31      -> stop after chunk tr
    const int kk_lo = mode == 0 ? tc * NB : 0;
    const int kk_hi = mode == 0 ? s : (tr + 1) * NB;
    // this thread's 4 + 4 elements of a chunk, fetched one chunk ahead of the MFMAs
    double pa[4], pb[4];
    auto fetch = [&](int kk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256;
            const int r = i / NB, c = i % NB;
            if (mode == 0) {
                const int gr = r0 + s + tr * NB + r, gc = r0 + kk + c; // L21 (rows m..m_pad of L are zero)
                pa[q] = (gr < m) ? S[(size_t)gr * ldS + gc] : 0.0;
                pb[q] = V[(size_t)(r0 + kk + r) * ldw + r0 + tc * NB + c]; // X11[kk + r][tc*32 + c]
            } else {
                pa[q] = V[(size_t)(r0 + s + tr * NB + r) * ldw + r0 + s + kk + c]; // X22[tr*32 + r][kk + c]
                pb[q] = Tbuf[(size_t)(r0 + s + kk + r) * ldw + tc * NB + c];       // T[kk + r][tc*32 + c]
            }
        }
    };
    fetch(kk_lo);
    for (int kk = kk_lo; kk < kk_hi; kk += NB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256;
            sA[i / NB][i % NB] = pa[q];
            sB[i / NB][i % NB] = pb[q];
        }
        __syncthreads();
        if (kk + NB < kk_hi) fetch(kk + NB);
#pragma unroll
        for (int k4 = 0; k4 < NB; k4 += 4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[16 * bi + lr][k4 + lk], sB[k4 + lk][16 * bj + lr], acc, 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
        const int gi = r0 + s + tr * NB + r, gj = tc * NB + c;
        if (mode == 0) {
            Tbuf[(size_t)gi * ldw + gj] = acc[q];
        } else {
            V[(size_t)gi * ldw + r0 + gj] = -acc[q];
            W[(size_t)(r0 + gj) * ldw + gi] = -acc[q];
            if (Wf) Wf[(size_t)(r0 + gj) * ldw + gi] = (float)(-acc[q]);
        }
    }
}

// ------------------------------------------------------------------------------------------------- dx = B' z
// y = inv(L)' z = W z (W upper triangular, fp64): with it dx = (H P)' y = G' inv(S) nu, the gain applied without going
// through B.  fp32 configuration only: B = inv(L) G comes out of an fp32 MFMA GEMM (accumulation error ~ sqrt(m) eps per
// element), G is the fp64-accumulated H P rounded once -- measured at N = 1000: the inverse-depth components were
// 2e-7 ... 1e-6 off (up to 8e-5 of a small rho) through B' z.  Inverse + GEMM path only (W exists there); the sweep path
// keeps B' z, inside the tolerance block-wise.  One wavefront per row.
__global__ void __launch_bounds__(256) k_yvec(const double *W, int ldw, int m, const double *z, double *y)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= m) return;
    const double *w = W + (size_t)i * ldw;
    double s = 0.0;
    for (int k = i + lane; k < m; k += 64) s += w[k] * z[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) y[i] = s;
}

template <typename T, bool USE_G, typename TG = T> // T: type of B and of the gathered rows G; TG: storage type of P
__global__ void __launch_bounds__(256)
k_dx_partial(const T *B, int ld, int m, int n, const double *z, double *part, int ldpart, double *sq_part, double *cam_part,
             const double *Bc, const TG *P, RowMap rm, double *dsave, double *csave, int avg, const T *G, const double *y, int *bexp,
             const double *st)
{
    __shared__ double sc[64][13]; // fp64 camera columns of a chunk of rows of B
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int ks = blockIdx.y;
    // the quaternion as it is before this update, for k_apply_normalize (whose workgroups each need it while one of them rewrites it)
    if (blockIdx.x == 0 && ks == 0 && threadIdx.x < 4) part[(size_t)DX_SPLIT * ldpart + threadIdx.x] = st[ST_X + 3 + threadIdx.x];
    if (ks == 0 && csave && j < n) {
        // phase 0 of the fp64 fix (k_fix_normalize): keep the diagonal and the camera rows of P as they are before the downdate
        // (avg: the first downdate after an arbitrary upload works on 0.5 (P(a,j) + P(j,a)))
        const bool mine = owns_row(rm, j);
        const TG *prow = P + (size_t)local_row(rm, j) * ld;
        if (mine) dsave[j] = (double)prow[j];
#pragma unroll
        for (int a = 0; a < 13; ++a)
            csave[(size_t)a * ldpart + j] = avg ? (double)((TG)0.5 * P[(size_t)a * ld + j] + (TG)0.5 * prow[a]) : (double)P[(size_t)a * ld + j];
    }
    const int per = (m + DX_SPLIT - 1) / DX_SPLIT;
    const int kb = ks * per, ke = min(m, kb + per);
    double s = 0.0, q = 0.0;
    int hi = 0; // exact downdate: largest biased exponent of this column's entries of B (the column scale of its digit planes)
    double c[13];
#pragma unroll
    for (int a = 0; a < 13; ++a) c[a] = 0.0;
    for (int k0 = kb; k0 < ke; k0 += 64) {
        const int cnt = min(64, ke - k0);
        if (Bc) { // fp64 camera columns of B (fp32 covariance only)
            __syncthreads();
            for (int i = threadIdx.x; i < cnt * 13; i += 256) sc[i / 13][i % 13] = Bc[(size_t)(k0 + i / 13) * 16 + i % 13];
            __syncthreads();
        }
        if (j < n) {
            // the whole chunk's rows are requested before the first is used: a split is one round trip to memory, not
            // one per few rows (the loop was latency-bound: 30 us for 23 MB)
            T bv[64], gv[USE_G ? 64 : 1];
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const int k = min(u, cnt - 1);
                bv[u] = B[(size_t)(k0 + k) * ld + j];
                if (USE_G) gv[u] = j >= 13 ? G[(size_t)(k0 + k) * ld + j] : (T)0;
            }
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                if (u < cnt) {
                    const double b = (Bc && j < 13) ? sc[u][j] : (double)bv[u];
                    // feature columns with the explicit inverse at hand: dx_j = sum_k (H P)_kj y_k (see k_yvec); camera columns: Bc' z (fp64)
                    if (USE_G && j >= 13) s += (double)gv[u] * y[k0 + u];
                    else s += b * z[k0 + u];
                    q += b * b; // (B'B)_jj in fp64, same pass over B: see k_fix_normalize
                    if (sizeof(T) == 8 && bexp) hi = max(hi, __double2hiint(b) & 0x7fffffff);
                    if (cam_part) {
#pragma unroll
                        for (int a = 0; a < 13; ++a) c[a] += sc[u][a] * b; // (B'B)_aj, camera rows
                    }
                }
            }
        }
    }
    if (j >= n) return;
    if (bexp && (hi >> 20)) atomicMax(&bexp[j], hi >> 20);
    part[(size_t)ks * ldpart + j] = s;
    if (sq_part) sq_part[(size_t)ks * ldpart + j] = q;
    if (cam_part) {
#pragma unroll
        for (int a = 0; a < 13; ++a) cam_part[((size_t)ks * 13 + a) * ldpart + j] = c[a];
    }
}

// fp32 covariance only.  The variances are where the downdate subtracts the most (an inverse-depth variance goes from
// 1 to 0.07 in one update) and an fp32 MFMA accumulation over ~1000 terms leaves an absolute error of ~3e-6 there, 30x
// the error of any off-diagonal element (measured at N = 1000).  The diagonal of B'B is therefore accumulated in fp64
// (k_dx_partial, no extra pass over B) and P_jj = P_jj(old) - (B'B)_jj is rounded to fp32 once.  The 13 camera rows
// and columns get the same treatment: the camera state -- above all the angular velocity, which is observed only
// through these cross-covariances -- inherits their error (measured: w block 2e-4 -> see DESIGN.md section 6).
// The old values are saved by k_dx_partial (its k-split 0) before the downdate; k_fix_normalize writes the new ones after it.
// quaternion normalisation and its Jacobian (Update.cpp:45-62, 303-312), one thread
__device__ inline void quat_norm_dev(double *st)
{
    double *q = st + ST_X + 3;
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    const double nrm = sqrt(r * r + x * x + y * y + z * z);
    const double a = 1.0 / (nrm * nrm * nrm);
    double *J = st + ST_JN;
    const double M[16] = {x * x + y * y + z * z, -r * x, -r * y, -r * z,
                          -x * r, r * r + y * y + z * z, -x * y, -x * z,
                          -y * r, -y * x, r * r + x * x + z * z, -y * z,
                          -z * r, -z * x, -z * y, r * r + x * x + y * y};
    for (int i = 0; i < 16; ++i) J[i] = M[i] * a;
    q[0] = r / nrm; q[1] = x / nrm; q[2] = y / nrm; q[3] = z / nrm;
    quat_to_rot(q, st + ST_R);
}

// stateUpdate (Update.cpp:147-204): x += dx with the DELTA dead-band on every component; R(q) recomputed from
// the un-normalised q (:168).
__global__ void __launch_bounds__(256)
k_state_apply(double *st, double *feat_pos, const int *feat_type, const int *feat_covpos, int N, const double *part,
              int ldpart, int normalise, const int *counts)
{
    if (filter_frozen(counts)) return; // the update's sweep failed: x stays as it was (engine.h)
    const int t = blockIdx.x * 256 + threadIdx.x;
    // this thread's feature parameter: the k-splits are requested together, before the camera part below
    const bool live = t < N * 6 && (t % 6) < feat_dim(feat_type[t / 6]);
    const int j = live ? feat_covpos[t / 6] + t % 6 : 0;
    double pv[DX_SPLIT];
#pragma unroll
    for (int ks = 0; ks < DX_SPLIT; ++ks) pv[ks] = part[(size_t)ks * ldpart + j];
    if (blockIdx.x == 0) { // camera part: one thread per component sums its splits, thread 0 applies and normalises
        __shared__ double cam[13];
        if (threadIdx.x < 13) {
            double s = 0.0;
#pragma unroll
            for (int ks = 0; ks < DX_SPLIT; ++ks) s += part[(size_t)ks * ldpart + threadIdx.x];
            cam[threadIdx.x] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double *x = st + ST_X;
            for (int i = 0; i < 13; ++i)
                if (fabs(cam[i]) > EKF_DELTA) x[i] += cam[i];
            // the covariance downdate that sits between the state update and the normalisation in the reference
            // (Update.cpp:299-312) does not read the state, so the normalisation is done here (not for updateOnlyState)
            if (normalise) quat_norm_dev(st);
            else quat_to_rot(x + 3, st + ST_R);
        }
    }
    if (!live) return;
    double s = 0.0;
#pragma unroll
    for (int ks = 0; ks < DX_SPLIT; ++ks) s += pv[ks];
    if (fabs(s) > EKF_DELTA) feat_pos[6 * (t / 6) + t % 6] += s;
}

// ------------------------------------------------------------------------------ quaternion normalisation tail
// normalizeQuaternionJacobian from the un-normalised q (Update.cpp:45-60, 305), then q /= |q| (:308-313).

// normalizeCovariance (Update.cpp:64-85): P <- D P D', D = diag(I3, J, I).  Five disjoint blocks; block 0 owns
// the 7x7 corner pieces, every other thread owns column j of the row strip 3..6 and row j of the column strip.
template <typename T>
__global__ void __launch_bounds__(256) k_normalize_cov(T *P, int ld, int n, const double *st, RowMap rm, const int *counts)
{
    __shared__ double J[16];
    __shared__ double C[7][7];
    const int tid = threadIdx.x;
    if (filter_frozen(counts)) return;
    if (tid < 16) J[tid] = st[ST_JN + tid];
    if (blockIdx.x == 0) {
        if (tid < 49) C[tid / 7][tid % 7] = (double)P[(size_t)(tid / 7) * ld + tid % 7];
        __syncthreads();
        if (tid < 12) { // P[0:3,3:7] J'
            const int i = tid / 4, a = tid % 4;
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += C[i][3 + k] * J[a * 4 + k];
            P[(size_t)i * ld + 3 + a] = (T)s;
        } else if (tid < 24) { // J P[3:7,0:3]
            const int t = tid - 12, a = t / 3, j = t % 3;
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += J[a * 4 + k] * C[3 + k][j];
            P[(size_t)(3 + a) * ld + j] = (T)s;
        } else if (tid < 40) { // J P[3:7,3:7] J' : upper triangle, mirrored (keeps P bitwise symmetric)
            const int t = tid - 24, a = t / 4, b = t % 4;
            if (a <= b) {
                double s = 0.0;
                for (int l = 0; l < 4; ++l) {
                    double u = 0.0;
                    for (int k = 0; k < 4; ++k) u += J[a * 4 + k] * C[3 + k][3 + l];
                    s += u * J[b * 4 + l];
                }
                P[(size_t)(3 + a) * ld + 3 + b] = (T)s;
                P[(size_t)(3 + b) * ld + 3 + a] = (T)s;
            }
        }
        return;
    }
    __syncthreads();
    const int j = 7 + (blockIdx.x - 1) * 256 + tid;
    if (j >= n) return;
    const bool mine = owns_row(rm, j); // sharded storage: the column strip exists only for owned rows
    T *prow = P + (size_t)local_row(rm, j) * ld;
    double col[4], row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        col[k] = (double)P[(size_t)(3 + k) * ld + j];
        row[k] = mine ? (double)prow[3 + k] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double s = 0.0, t = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s += J[a * 4 + k] * col[k];
            t += row[k] * J[a * 4 + k];
        }
        P[(size_t)(3 + a) * ld + j] = (T)s;
        if (mine) prow[3 + a] = (T)t;
    }
}

// stateUpdate and normalizeCovariance in ONE launch behind the downdate (exact and fp64 configurations, covariance updates):
// k_state_apply + k_normalize_cov, same arithmetic in the same order.  The downdate reads neither the state nor J, so the state
// update may follow it; what every workgroup needs for its strip of P is J, i.e. the un-normalised updated quaternion: each
// recomputes it from the quaternion as it was before the update (saved beside the k-splits of dx by the dx kernel -- workgroup 0
// rewrites the live one meanwhile) and its four components of dx.  Workgroup 0 owns the camera state and the 7 x 7 corner,
// workgroup b >= 1 the columns 7 + 256 (b - 1) ... of the strips; thread t of the grid owns feature parameter t.
template <typename T>
__global__ void __launch_bounds__(256)
k_apply_normalize(T *P, int ld, int n, double *st, RowMap rm, double *feat_pos, const int *feat_type, const int *feat_covpos, int N,
                  const double *part, int ldpart, const int *counts)
{
    __shared__ double J[16];
    __shared__ double C[7][7];
    __shared__ double cam[13];
    const int tid = threadIdx.x;
    if (filter_frozen(counts)) return; // the update's sweep failed: x and P stay as they were (engine.h)
    const int t = blockIdx.x * 256 + tid;
    const bool live = t < N * 6 && (t % 6) < feat_dim(feat_type[t / 6]);
    const int jf = live ? feat_covpos[t / 6] + t % 6 : 0;
    double pv[DX_SPLIT];
#pragma unroll
    for (int ks = 0; ks < DX_SPLIT; ++ks) pv[ks] = part[(size_t)ks * ldpart + jf];
    if (tid < 13 && (blockIdx.x == 0 || (tid >= 3 && tid < 7))) {
        double s = 0.0;
#pragma unroll
        for (int ks = 0; ks < DX_SPLIT; ++ks) s += part[(size_t)ks * ldpart + tid];
        cam[tid] = s;
    }
    if (blockIdx.x == 0 && tid < 49) C[tid / 7][tid % 7] = (double)P[(size_t)(tid / 7) * ld + tid % 7];
    __syncthreads();
    if (tid == 0) {
        if (blockIdx.x == 0) {
            double *x = st + ST_X;
            for (int i = 0; i < 13; ++i)
                if (fabs(cam[i]) > EKF_DELTA) x[i] += cam[i];
            quat_norm_dev(st);
            for (int i = 0; i < 16; ++i) J[i] = st[ST_JN + i];
        } else {
            const double *q0 = part + (size_t)DX_SPLIT * ldpart;
            double q[4];
            for (int i = 0; i < 4; ++i) {
                q[i] = q0[i];
                if (fabs(cam[3 + i]) > EKF_DELTA) q[i] += cam[3 + i];
            }
            const double r = q[0], x = q[1], y = q[2], z = q[3];
            const double nrm = sqrt(r * r + x * x + y * y + z * z);
            const double a = 1.0 / (nrm * nrm * nrm);
            const double M[16] = {x * x + y * y + z * z, -r * x, -r * y, -r * z,
                                  -x * r, r * r + y * y + z * z, -x * y, -x * z,
                                  -y * r, -y * x, r * r + x * x + z * z, -y * z,
                                  -z * r, -z * x, -z * y, r * r + x * x + y * y};
            for (int i = 0; i < 16; ++i) J[i] = M[i] * a;
        }
    }
    if (live) {
        double s = 0.0;
#pragma unroll
        for (int ks = 0; ks < DX_SPLIT; ++ks) s += pv[ks];
        if (fabs(s) > EKF_DELTA) feat_pos[6 * (t / 6) + t % 6] += s;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < 12) { // P[0:3,3:7] J'
            const int i = tid / 4, a = tid % 4;
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += C[i][3 + k] * J[a * 4 + k];
            P[(size_t)i * ld + 3 + a] = (T)s;
        } else if (tid < 24) { // J P[3:7,0:3]
            const int u = tid - 12, a = u / 3, j = u % 3;
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += J[a * 4 + k] * C[3 + k][j];
            P[(size_t)(3 + a) * ld + j] = (T)s;
        } else if (tid < 40) { // J P[3:7,3:7] J' : upper triangle, mirrored (keeps P bitwise symmetric)
            const int u = tid - 24, a = u / 4, b = u % 4;
            if (a <= b) {
                double s = 0.0;
                for (int l = 0; l < 4; ++l) {
                    double v = 0.0;
                    for (int k = 0; k < 4; ++k) v += J[a * 4 + k] * C[3 + k][3 + l];
                    s += v * J[b * 4 + l];
                }
                P[(size_t)(3 + a) * ld + 3 + b] = (T)s;
                P[(size_t)(3 + b) * ld + 3 + a] = (T)s;
            }
        }
        return;
    }
    const int j = 7 + (blockIdx.x - 1) * 256 + tid;
    if (j >= n) return;
    const bool mine = owns_row(rm, j); // sharded storage: the column strip exists only for owned rows
    T *prow = P + (size_t)local_row(rm, j) * ld;
    double col[4], row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        col[k] = (double)P[(size_t)(3 + k) * ld + j];
        row[k] = mine ? (double)prow[3 + k] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double s = 0.0, u = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s += J[a * 4 + k] * col[k];
            u += row[k] * J[a * 4 + k];
        }
        P[(size_t)(3 + a) * ld + j] = (T)s;
        if (mine) prow[3 + a] = (T)u;
    }
}

// The fp64 fix and k_normalize_cov in one launch (fp32 covariance, covariance updates): both own the same elements -- the
// thread of column j writes the fp64-accumulated camera rows of its column (and, for owned rows, their mirror) and then
// applies the quaternion-normalisation Jacobian to rows / columns 3..6 of it, from the values it has just rounded to T,
// so nothing is written twice or re-read.  The 13 x 13 camera block needs all of its own values: block 0 assembles it in
// LDS, transforms it there (upper triangle, mirrored: P stays bitwise symmetric) and writes it once.
template <typename T>
__global__ void __launch_bounds__(256)
k_fix_normalize(T *P, int ld, int n, RowMap rm, const double *dsave, const double *sq_part, const double *csave,
                const double *cam_part, int ldpart, const double *st)
{
    __shared__ double J[16];
    __shared__ double C[13][13], Cn[13][13];
    const int tid = threadIdx.x;
    if (tid < 16) J[tid] = st[ST_JN + tid];
    const int j = blockIdx.x * 256 + tid;
    const bool live = j < n;
    const bool mine = live && owns_row(rm, j);
    T *prow = P + (size_t)local_row(rm, live ? j : 0) * ld;
    double v[13];
    if (live) {
#pragma unroll
        for (int a = 0; a < 13; ++a) {
            double q = 0.0;
#pragma unroll
            for (int ks = 0; ks < DX_SPLIT; ++ks) q += cam_part[((size_t)ks * 13 + a) * ldpart + j];
            v[a] = (double)(T)(csave[(size_t)a * ldpart + j] - q); // rounded to T once
        }
        if (mine && j >= 13) {
            double q = 0.0;
#pragma unroll
            for (int ks = 0; ks < DX_SPLIT; ++ks) q += sq_part[(size_t)ks * ldpart + j];
            prow[j] = (T)(dsave[j] - q);
        }
    }
    if (blockIdx.x == 0 && tid < 13) {
#pragma unroll
        for (int a = 0; a < 13; ++a) C[a][tid] = v[a];
    }
    __syncthreads();
    if (live && j >= 13) {
        // column j of the camera rows (and row j of the camera columns): rows 3..6 <- J rows 3..6
        double w[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += J[a * 4 + k] * v[3 + k];
            w[a] = s;
        }
#pragma unroll
        for (int a = 0; a < 13; ++a) {
            const T out = (T)((a >= 3 && a < 7) ? w[a - 3] : v[a]);
            P[(size_t)a * ld + j] = out;
            if (mine) prow[a] = out;
        }
    }
    if (blockIdx.x != 0) return; // uniform: every thread of block 0 reaches the barrier below
    // the 13 x 13 block: Cn = D C D', D = diag(I3, J, I6), on the upper triangle
    for (int i = tid; i < 169; i += 256) {
        const int a = i / 13, b = i % 13;
        if (a > b) continue;
        const bool ra = a >= 3 && a < 7, rb = b >= 3 && b < 7;
        double s;
        if (!ra && !rb) s = C[a][b];
        else if (ra && !rb) { // J C[3:7, b]
            s = 0.0;
            for (int k = 0; k < 4; ++k) s += J[(a - 3) * 4 + k] * C[3 + k][b];
        } else if (!ra && rb) { // C[a, 3:7] J'
            s = 0.0;
            for (int k = 0; k < 4; ++k) s += C[a][3 + k] * J[(b - 3) * 4 + k];
        } else { // J C[3:7,3:7] J'
            s = 0.0;
            for (int l = 0; l < 4; ++l) {
                double u = 0.0;
                for (int k = 0; k < 4; ++k) u += J[(a - 3) * 4 + k] * C[3 + k][3 + l];
                s += u * J[(b - 3) * 4 + l];
            }
        }
        Cn[a][b] = s;
        Cn[b][a] = s;
    }
    __syncthreads();
    for (int i = tid; i < 169; i += 256) P[(size_t)(i / 13) * ld + i % 13] = (T)Cn[i / 13][i % 13];
}

// P-update launcher lives in kernels_pupdate.hip
void launch_p_update(EkfEngine *e, int m_pad, int m);

// ---------------------------------------------------------------------------------------------- persistent sweep launcher
// The registry of persistent sweeps of one device (engine.h: SweepRegistry).  The table below only FINDS the registry of a device;
// the registry itself belongs to the engines attached to it (shared_ptr) and goes with the last of them.
std::shared_ptr<SweepRegistry> sweep_registry_attach(int device)
{
    static std::mutex table_mu;
    static std::map<int, std::weak_ptr<SweepRegistry>> table;
    std::shared_ptr<SweepRegistry> reg;
    {
        std::lock_guard<std::mutex> lk(table_mu);
        reg = table[device].lock();
        if (!reg) {
            reg = std::make_shared<SweepRegistry>();
            reg->device = device;
            table[device] = reg;
        }
    }
    bool others;
    {
        std::lock_guard<std::mutex> lk(reg->mu);
        others = reg->members > 0;
        ++reg->members;
    }
    // a member that launched while it was alone recorded nothing: drain the device once, AFTER the count went up under the lock --
    // every persistent launch is made under the same lock, so it either saw the new count (and records from now on) or was enqueued
    // before this point (and is drained here)
    if (others) (void)hipDeviceSynchronize();
    return reg;
}

void sweep_registry_detach(const std::shared_ptr<SweepRegistry> &reg, const void *engine)
{
    if (!reg) return;
    std::lock_guard<std::mutex> lk(reg->mu);
    --reg->members;
    if (reg->owner == engine) reg->owner = nullptr; // (the event stays: it marks work of a stream that is drained before it is destroyed)
}

#ifdef EKF_SWEEP_TRACE // debug builds only (scripts/persist_trace.py): the stamps of the LAST persistent sweep launched while enabled
constexpr int PS_TRACE_WORDS = 4096;
static unsigned long long *g_ps_trace = nullptr;
static int g_ps_trace_on = 0, g_ps_trace_m = 0;
extern "C" int ekf_debug_persist_trace(int enable, unsigned long long *out, int *m)
{
    g_ps_trace_on = enable;
    if (out && g_ps_trace) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(out, g_ps_trace, sizeof(unsigned long long) * PS_TRACE_WORDS, hipMemcpyDeviceToHost);
        if (m) *m = g_ps_trace_m;
    }
    return 0;
}
#endif

// resident workgroups of k_chol_persist<PL> on the engine's device (two per CU at most), -1 if the query fails
template <bool PL>
static int persist_capacity(EkfEngine *e)
{
    int &cap = e->ps_cap[PL ? 1 : 0];
    if (cap == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chol_persist<PL>, 256, 0) != hipSuccess || per_cu < 1) cap = -1;
        else cap = std::min(per_cu, 2) * e->n_cus;
    }
    return cap;
}

// grid layout of the persistent sweep of an update of m rows with n_bcols 32-column blocks of B; false: it does not fit the device
// (decided BEFORE the gather / assembly launch, which skips the first block's factorisation for a persistent sweep)
struct PsLayout {
    int nbk, n_b, n_t, grid, layout_cus;
};
template <bool PL>
static bool persist_layout(EkfEngine *e, int m, int n_bcols, PsLayout &L)
{
    const int cap = persist_capacity<PL>(e);
    const int nbk = (m + NB - 1) / NB;
    if (cap < 64 || nbk > B_SWEEP_MAX / NB || !e->d.sweep_ctl) return false;
    const int ntiles = (nbk + 1) * (nbk + 2) / 2 - 1;
    // Block order: [chain][B workers][tile workers]; block n_cus -- which in-order dispatch would put on the chain workgroup's CU --
    // is an empty spacer.  One B worker per 32-column block (at most 3/5 of the slots), the rest tile workers: with one or two
    // tiles each they react to a published inverse within ~3 us, which the chain needs in its first panels (the band of two tiles
    // it applies itself covers the rest).  Measured at N = 1000 (profiles/r05_persist_layout.txt): 134 tile workers beside B workers
    // with CUs of their own 10.4 us per panel, 200 / 322 tile workers sharing the B workers' CUs 9.4 / 9.0.
    const int C = e->n_cus;
    const int n_b = std::max(1, std::min(n_bcols, (cap - 2) * 3 / 5));
    int n_t = std::min(ntiles, cap - 2 - n_b);
#ifdef EKF_SWEEP_TRACE // tuning runs of the debug build only
    if (const char *ev = std::getenv("EKF_PS_NT")) n_t = std::max(1, std::min(n_t, atoi(ev)));
#endif
    n_t = std::max(n_t, (ntiles + 63) / 64);
    if (1 + n_b + n_t + 1 > cap) return false;
    const bool spacer = 1 + n_b + n_t > C;
    L.nbk = nbk;
    L.n_b = n_b;
    L.n_t = n_t;
    L.grid = 1 + n_b + n_t + (spacer ? 1 : 0);
    L.layout_cus = spacer ? C : 0;
    return true;
}

// one launch for the whole sweep of an update of m rows (layout from persist_layout); false: not launched
template <bool PL>
static bool launch_persistent_sweep(EkfEngine *e, int m, const double *G, double *Bout, int n_bcols, const BPlanes &bp, const PsLayout &L)
{
    if (e->ps_epoch >= (1u << 22) || e->ps_epoch == 0) { // (re)start the epochs of the flags well before they can wrap
        if (hipMemsetAsync(e->d.sweep_ctl, 0, sweep_ctl_bytes(), e->stream) != hipSuccess) return false;
        e->ps_epoch = 0;
    }
    ++e->ps_epoch;
    PsArgs a{};
    a.S = e->d.S; a.LL = e->d.LL; a.ldS = e->ldS; a.m = m; a.nbk = L.nbk;
    a.V = e->d.Dinv; a.ldw = e->ldW; a.nu = e->d.nu; a.zvec = e->d.zvec; a.counts = e->d.counts;
    a.G = G; a.Bout = Bout; a.ld = e->ldP; a.bp = bp;
    a.ctl = (SweepCtl *)e->d.sweep_ctl; a.eb = e->ps_epoch * PS_EPOCH_STEP;
    a.n_b = L.n_b; a.n_bcols = n_bcols; a.n_t = L.n_t; a.n_cus = L.layout_cus;
    a.fault = (e->ps_fault > 0 && --e->ps_fault == 0) ? 1 : 0; // include/ekf_test_hooks.h: one sweep only
    a.trace = nullptr;
#ifdef EKF_SWEEP_TRACE
    if (g_ps_trace_on) {
        if (!g_ps_trace && hipMalloc(&g_ps_trace, sizeof(unsigned long long) * PS_TRACE_WORDS) != hipSuccess) g_ps_trace = nullptr;
        if (g_ps_trace) {
            (void)hipMemsetAsync(g_ps_trace, 0, sizeof(unsigned long long) * PS_TRACE_WORDS, e->stream);
            a.trace = g_ps_trace;
            g_ps_trace_m = m;
        }
    }
#endif
    SweepRegistry *reg = e->ps_reg.get();
    if (!reg) {
        k_chol_persist<PL><<<L.grid, 256, 0, e->stream>>>(a);
        return true;
    }
    // every persistent launch goes through the registry's lock; with other engines on the device it waits for the last persistent
    // sweep of any OTHER engine and leaves its own end behind as the event the next one waits for
    std::lock_guard<std::mutex> lk(reg->mu);
    const bool shared = reg->members > 1;
    if (shared && !reg->last && hipEventCreateWithFlags(&reg->last, hipEventDisableTiming) != hipSuccess) reg->last = nullptr;
    if (shared && reg->last && reg->owner && reg->owner != e) (void)hipStreamWaitEvent(e->stream, reg->last, 0);
    k_chol_persist<PL><<<L.grid, 256, 0, e->stream>>>(a);
    if (shared && reg->last) {
        (void)hipEventRecord(reg->last, e->stream);
        reg->owner = e;
    }
    return true;
}

// T: storage type of P; TB: type of H P, of its gathered rows G, of B = inv(L) G and of the arithmetic that forms it
// (EXACT: EKF_PRECISION_F32_EXACT, T = float, and EKF_PRECISION_F64_EXACT, T = double -- TB = double, the downdate by kernels_pexact.hip)
template <typename T, typename TB, bool EXACT = false>
static void update_impl(EkfEngine *e, int M, bool update_cov)
{
    hipStream_t s = e->stream;
    const int m = 2 * M, n = e->n, ld = e->ldP, ldS = e->ldS, ldw = e->ldW;
    const int m_pad = round_up(m, NB);
    const int n_pad = round_up(n, LD_ALIGN);
    // the kernels behind the sweep that write x or P leave them alone when the sweep failed (filter_frozen, engine.h); the fast fp32
    // configuration keeps its round-3 tail (its sweep is never the persistent launch: nothing to retry)
    const int *frz = (sizeof(T) == 4 && !EXACT) ? nullptr : e->d.counts;
    TB *G = (TB *)e->d.G; // gathered rows of H P (or the H P table itself, see planes_b)
    TB *A = (TB *)e->d.A; // B = inv(L) G
    // B = inv(L) G: up to B_SWEEP_MAX rows, row block k is formed inside the launch of panel k (forward substitution
    // beside the look-ahead factorisation: no explicit inverse, no GEMM launch); above it the per-launch row block becomes
    // longer than the factorisation it hides behind, and the explicit inverse + one big-tile GEMM is the better use of
    // the MFMA pipe.  e->b_path (ekf_set_update_path): 0 by size, 1 always in the sweep, 2 always by GEMM.
    const bool b_in_sweep = e->b_path == 1 || (e->b_path == 0 && m_pad <= B_SWEEP_MAX);
    if (sizeof(TB) == 4 && b_in_sweep && m_pad > B_SWEEP_MAX) {
        // fp32 forward substitution over more than 64 panels changes the filter's decisions (measured at N = 5000, DESIGN.md
        // section 6): the size switch to inverse + GEMM is an accuracy boundary -- EKF_UPDATE_PATH_SWEEP is refused there
        e->err = "EKF_UPDATE_PATH_SWEEP with more than 2048 measurement rows in the fp32 configuration";
        e->hook_rc = EKF_ERR_INVALID_ARG;
        return;
    }
    // exact configuration on one GPU, B in the sweep: its rows are formed from int8 digit planes (chol_bplanes.h), in single-panel
    // launches (the planes of L cover B_SWEEP_MAX rows; a sharded rank does not know the diagonal of the rows it does not own)
    const bool sharded = e->shard_world > 1;
    const bool shard_cols = EXACT && sharded && update_cov && e->exchange_hook && e->d.Bstage && m_pad + NB <= e->bstage_rows &&
                            (int)e->shard_row_begin.size() == e->shard_world + 1; // every rank forms its own columns of B
    const bool planes_b = EXACT && b_in_sweep && m_pad <= B_SWEEP_MAX && e->d.Lq != nullptr && update_cov && (!sharded || shard_cols);
    const bool apriori = planes_b || shard_cols || (EXACT && !b_in_sweep && update_cov && e->d.Wq != nullptr); // column scales of B from diag(P)
    // sharded step with the rows of B from digit planes: G by symmetry and S by columns (k_g_cols) instead of the exchange of the rows of G
    // (round 5: also above 2048 rows, where B = inv(L) G is the int8 GEMM over the rank's own column tiles: no rows of G travel at any size)
    bool sym_g = false;
    const bool gemm_own = EXACT && !b_in_sweep && update_cov && e->d.Wq != nullptr && shard_cols;
    if constexpr (EXACT)
        sym_g = shard_cols && (planes_b || gemm_own) && e->after_gather != nullptr && (int)e->shard_rb.size() == e->shard_world + 1 &&
                e->d.W != nullptr && e->d.Tbuf != nullptr;
    BPlanes bp{};
    // sharded: this rank forms the column blocks [cb0, cb1) of B -- the blocks whose first column lies in its share of the state
    // rows (rank 0: from column 0) -- and receives the others' digit planes afterwards (the B role divided by the ranks)
    int cb0 = 0, cb1 = n_pad / NB;
    std::vector<int32_t> col_rb; // column boundaries of the ranks' shares of the planes
    if (planes_b) {
        bp.Bq = e->d.Bq; bp.b_stride = (size_t)e->bq_rows * ld; bp.ldq = ld; bp.bexp = e->d.Bexp;
        bp.Lq = e->d.Lq; bp.nbk = e->lq_nbk; bp.l_stride = (size_t)e->lq_nbk * e->lq_nbk * 1024; bp.lexp = e->d.Lexp;
        bp.grow = sharded ? nullptr : e->d.Grow;
        bp.counts = e->d.counts;
        bp.n_live = n;
        bp.no_fp64 = 1; // dx = B'z from the planes: the fp64 rows of B (8 bytes per element written, then read by k_dx_partial) never exist
    }
    if (planes_b || shard_cols) {
        if (sharded) {
            const int W = e->shard_world;
            col_rb.assign(W + 1, 0);
            // (G by symmetry: a rank can only form the columns of the rows it holds, so the shares end exactly at the ownership
            // boundaries and the 32-column block that straddles one is formed by both neighbours, each valid in its own columns)
            for (int r = 1; r < W; ++r) col_rb[r] = std::min(n_pad, sym_g ? e->shard_row_begin[r] : round_up(e->shard_row_begin[r], NB));
            col_rb[W] = n_pad;
            e->last_col_rb = col_rb;
            if (planes_b) {
                cb0 = col_rb[e->shard_rank] / NB;
                cb1 = (col_rb[e->shard_rank + 1] + NB - 1) / NB;
                bp.bcol0 = cb0;
                bp.c_live0 = col_rb[e->shard_rank]; // (the saturation check of the cut: this rank's own columns)
                bp.n_live = std::min(n, (int)col_rb[e->shard_rank + 1]);
            }
            // the diagonal of P behind the a-priori column scales: every rank's own entries, then all of them
            launch_diag_extract(e, e->d.Pdiag);
            std::vector<int32_t> drb(e->shard_row_begin.begin(), e->shard_row_begin.end());
            e->hook_rc = e->exchange_hook(e, EKF_XCHG_PDIAG, e->d.Pdiag, sizeof(float), drb, "the diagonal of P");
            if (e->hook_rc) return;
        }
    }
    // ONE persistent launch for the whole sweep (chol_persist.h) where it applies: B inside the sweep, at most B_SWEEP_MAX rows, one
    // GPU, fp64 arithmetic of B (the fp64 and the exact configuration); otherwise the launch-per-panel sweep below
    const int sweep_mode = e->force_launches > 0 ? EKF_SWEEP_LAUNCHES : e->sweep_mode; // (the retry of a timed-out persistent sweep)
    const int lmode = (sweep_mode == EKF_SWEEP_PERSISTENT || sweep_mode == EKF_SWEEP_LAUNCHES) ? EKF_SWEEP_AUTO : sweep_mode;
    bool persist = (sweep_mode == EKF_SWEEP_AUTO || sweep_mode == EKF_SWEEP_PERSISTENT) && b_in_sweep && m_pad <= B_SWEEP_MAX &&
                   !sharded && sizeof(TB) == 8 && e->d.sweep_ctl != nullptr;
    // by size: one B worker per block of 32 columns is what the persistent sweep is laid out for; on wider maps (from 8192 state columns
    // on, where the launch-per-panel sweep switches to two panels per launch) its B workers take several blocks each and it loses:
    // N = 2000, 20.4 against 14.0 us per panel (profiles/r05_bench_n2000_f32x_persistent.json)
    if (persist && sweep_mode == EKF_SWEEP_AUTO && n_pad >= 8192) persist = false;
    if (persist && sweep_mode == EKF_SWEEP_AUTO && e->ps_backoff > 0) { // recent time-outs (a shared device): engine.h
        --e->ps_backoff;
        persist = false;
    }
    // its grid must fit the device's resident workgroups -- decided HERE, before the gather / assembly launch leaves the first
    // block's factorisation to the chain workgroup (a device partition with few CUs: the launch-per-panel sweep instead)
    PsLayout ps_layout{};
    if constexpr (sizeof(TB) == 8) {
        if (persist) persist = planes_b ? persist_layout<true>(e, m, cb1 - cb0, ps_layout) : persist_layout<false>(e, m, n_pad / NB, ps_layout);
    } else persist = false;
    if (persist && ++e->ps_ok_streak >= 256) e->ps_backoff_len = 0;
    e->last_update_M = M; // what a retry needs (engine.cpp: recover_failed_update)
    e->last_update_cov = update_cov;
    e->last_update_sym = e->p_exact_sym;
    e->last_update_persist = persist;
    double *V = e->d.Dinv, *W = b_in_sweep ? nullptr : e->d.W; // W = inv(L)' (and L row-major in LL): the GEMM path's
    float *Wf = sizeof(TB) == 4 && !b_in_sweep ? e->d.Wf : nullptr;
    bool merged_ga = false;
    {
        dim3 grid((n_pad / (int)(16 / sizeof(TB)) + 255) / 256, m_pad);
        if ((planes_b && !sharded) || sym_g) grid.x = 1; // no copy: one workgroup per row (bookkeeping, row map, its share of the column scales)
#define GATHER_ARGS e->d.matches, M, m_pad, (const TB *)e->d.HP, G, ld, n_pad, e->d.pred_uv, e->d.Hs, e->d.Hf, e->d.feat_type, e->d.feat_covpos, e->d.nu, \
                    e->d.mHs, e->d.mHf, e->d.mpos, e->d.mdim, e->d.HPc, e->d.Gc, EXACT ? e->d.Bexp : nullptr
        merged_ga = !sharded && !sym_g && !e->after_gather;
        if (merged_ga) {
            // gather and the assembly of S in one launch (the assembly reads the prediction tables, not the gather's output)
            const int mb = (M + 15) / 16;
            int *gr = planes_b ? e->d.Grow : nullptr;
#define GA_TAIL n, gr, e->cfg.cam.pixelErrorX, e->d.S, ldS, V, W, Wf, ldw, e->d.counts, planes_b ? e->d.Lexp : nullptr, persist ? 0 : 1, e->px_scale_shift
            if (apriori) k_gather_assemble<TB, T><<<mb * mb + (int)(grid.x * grid.y), 256, 0, s>>>(mb, (int)grid.x, GATHER_ARGS, (const T *)e->d.P, ld, GA_TAIL);
            else k_gather_assemble<TB, float><<<mb * mb + (int)(grid.x * grid.y), 256, 0, s>>>(mb, (int)grid.x, GATHER_ARGS, (const float *)nullptr, 0, GA_TAIL);
#undef GA_TAIL
        } else if (apriori && !sharded) // a-priori column scales from the diagonal of the covariance itself
            k_gather<TB, T><<<grid, 256, 0, s>>>(GATHER_ARGS, (const T *)e->d.P, ld, n, planes_b || sym_g ? e->d.Grow : nullptr, e->px_scale_shift);
        else
            k_gather<TB, float><<<grid, 256, 0, s>>>(GATHER_ARGS, apriori ? e->d.Pdiag : nullptr, 0, n,
                                                    (planes_b && !sharded) || sym_g ? e->d.Grow : nullptr, e->px_scale_shift);
#undef GATHER_ARGS
        if (planes_b && !sharded) G = (TB *)e->d.HP; // the consumers read H P through the row map (sym_g: G is formed by k_g_cols below)
    }
    // sharded step: every rank gathered the rows of the matches it owns; the others arrive here (engine.cpp)
    e->hook_rc = (e->after_gather && !sym_g) ? e->after_gather(e, M) : 0;
    if (e->hook_rc) return;
    {
        dim3 grid((M + 15) / 16, (M + 15) / 16);
        if (sym_g) {
            // G[:, own columns] (and the camera block) from the own rows of P; S by the block columns of the own matches, one
            // all-gather of those columns (transposed image in W, which the sweep path does not use), the rest of S's duties after it
            // (rows of B in the sweep: the rank's 32-column blocks; inverse + GEMM: its 128-column tiles, rounded outwards like the GEMM's)
            const int c_lo = planes_b ? cb0 * NB : col_rb[e->shard_rank] / 128 * 128;
            const int c_hi = planes_b ? cb1 * NB : std::min(n_pad, round_up(col_rb[e->shard_rank + 1], 128));
            // the transposed image of the own block columns of S for the exchange: the scratch of the triangular inverse
            // (always the inverse's scratch: round 5 used W below 2048 rows, which left W's other triangle dirty for a LATER update that
            // takes the inverse + GEMM route -- W = inv(L)' must be zero there; found by the saturation check of the cut, round 6)
            double *St = e->d.Tbuf;
            const int b_lo = e->shard_rb[e->shard_rank] / 2, b_hi = e->shard_rb[e->shard_rank + 1] / 2;
            if (c_lo > 0) k_g_cols<T><<<dim3(1, (m_pad + 63) / 64), 256, 0, s>>>((const T *)e->d.P, ld, e->rm, n, M, e->d.mHs, e->d.mHf, e->d.mpos,
                                                                          e->d.mdim, (double *)G, ld, 0, NB, m_pad);
            k_g_cols<T><<<dim3((c_hi - c_lo + 63) / 64, (m_pad + 63) / 64), 256, 0, s>>>((const T *)e->d.P, ld, e->rm, n, M, e->d.mHs, e->d.mHf,
                                                                                  e->d.mpos, e->d.mdim, (double *)G, ld, c_lo, c_hi, m_pad);
            k_assemble_S<TB><<<grid, 256, 0, s>>>(G, ld, M, e->d.mHs, e->d.mHf, e->d.mpos, e->d.mdim, e->cfg.cam.pixelErrorX, e->d.S, ldS, V,
                                                 nullptr, nullptr, ldw, e->d.counts, e->d.Lexp, nullptr, b_lo, b_hi, St, m_pad, 0);
            // (rows of the image are m_pad doubles: only the live columns of S travel, not the capacity-strided ldW)
            e->hook_rc = e->exchange_hook(e, EKF_XCHG_SCOLS, St, (size_t)m_pad * sizeof(double), e->shard_rb, "the columns of S");
            if (e->hook_rc) return;
            k_s_unpack<<<dim3((m + 255) / 256, m), 256, 0, s>>>(St, m_pad, e->d.S, ldS, m, 2 * b_lo, 2 * b_hi, e->d.Lexp);
            k_assemble_S<TB><<<dim3(1, 1), 256, 0, s>>>(G, ld, M, e->d.mHs, e->d.mHf, e->d.mpos, e->d.mdim, e->cfg.cam.pixelErrorX, e->d.S, ldS, V,
                                                       W, nullptr, ldw, e->d.counts, nullptr, nullptr, 0, 0, nullptr, 0, 1);
        } else if (!merged_ga)
        k_assemble_S<TB><<<grid, 256, 0, s>>>(G, ld, M, e->d.mHs, e->d.mHf, e->d.mpos, e->d.mdim,
                                             e->cfg.cam.pixelErrorX, e->d.S, ldS, V, W, Wf, ldw, e->d.counts,
                                             planes_b ? e->d.Lexp : nullptr, planes_b && !sharded ? e->d.Grow : nullptr, 0, M, nullptr, 0,
                                             persist ? 0 : 1); // (the persistent sweep's chain workgroup factorises the first block itself)
    }
    const int n_bblocks = b_in_sweep ? cb1 - cb0 : 0; // row block k of B = inv(L) G rides in the launch of panel k (sharded planes: own column blocks)
    hipEvent_t sw0 = nullptr, sw1 = nullptr;
    if (e->timing) { // the sweep's launches of this update, bracketed on the engine's stream (ekf_timing_sweep)
        (void)hipEventCreate(&sw0);
        (void)hipEventCreate(&sw1);
        (void)hipEventRecord(sw0, s);
    }
    // One panel per launch (k_chol_step) while the launch is as long as its look-ahead workgroup; two panels per launch
    // (chol_pair.h) once the rows of B are the longest role -- a pair launch reads the finished rows of B once for both panels
    // (the left-looking sum re-reads ALL of them every launch: at n = 12013 that traffic, not the factorisation, sets the pace).
    // Measured: N = 1000 9.1 us per panel in single launches against 9.55 in pairs; N = 2000 19.2 against 16.7.
    // A launch of the pair kernel with kbB = 0 eliminates one panel and prepares the first pair (it is also how a sweep in
    // EKF_SWEEP_PAIRS mode starts); once the 64 x 64 inverse exists the sweep stays in pairs.
    // (panels before this one) x n_pad from which a launch is in pairs.  Maps whose rows of B are more 32-column blocks than
    // the chip has CUs (n_pad > 8192; they get the 64-column B role of chol_pair.h) run in pairs from the first launch --
    // N = 2000: 14.0 us per panel from the start, 14.2 from 120 k, 14.5 from 195 k (single launches: 19.2); smaller maps
    // stay in single launches up to 195 k, i.e. N = 1000 (<= 190 k) always: 9.1 against 9.55 us per panel in pairs.
    const long long PAIR_FROM = n_bblocks > e->n_cus ? 0 : 195000;
    constexpr int PAIR_ROWS = 3072; // rows of the trailing matrix from which a sweep without the B role starts in pairs
    bool have_pair = false;
    int n_sweep_launches = 0;
    if (persist) {
        bool launched = false;
        if constexpr (sizeof(TB) == 8) {
            if (planes_b) launched = launch_persistent_sweep<true>(e, m, (const double *)G, nullptr, cb1 - cb0, bp, ps_layout);
            else launched = launch_persistent_sweep<false>(e, m, (const double *)G, (double *)A, n_pad / NB, BPlanes{}, ps_layout);
        }
        if (!launched) { // (only a failed memset of the flag block is left: the layout was checked before the gather)
            e->err = "persistent sweep could not be launched";
            e->hook_rc = EKF_ERR_HIP;
            return;
        }
        n_sweep_launches = 1;
    }
    for (int k0 = 0; k0 < m && !persist;) {
        ++n_sweep_launches;
        // (without the rows of B -- inverse + GEMM path -- a big sweep is bound by streaming the fp64 trailing matrix once per
        // launch: 2 x 8 m^2 bytes; pairs stream it once per two panels: N = 5000, m = 3000-4500: 16.6 -> 15.4 us per panel)
        const bool want_pairs = lmode == EKF_SWEEP_PAIRS ||
                                (lmode == EKF_SWEEP_AUTO &&
                                 (b_in_sweep ? (long long)(k0 / NB) * n_pad >= PAIR_FROM : m - k0 >= PAIR_ROWS));
        // rows of B from digit planes (chol_bplanes.h b_pair_rows_planes): two panels per launch halve the re-reads of the finished
        // planes and the rounds of workgroups of a wide map -- N = 2000: 15.4 -> 13.3 us per panel; N = 1000: 9.9 either way, so
        // AUTO takes pairs from 8192 state columns on (profiles/r04_sweep_pairs_planes.txt)
        const bool want_pairs_pl = lmode == EKF_SWEEP_PAIRS || (lmode == EKF_SWEEP_AUTO && n_pad >= 8192);
        const bool pair_launch = planes_b ? (have_pair || want_pairs_pl) : (have_pair || want_pairs);
        const int kbA = min(NB, m - k0);
        const int kbB = have_pair ? max(0, min(NB, m - k0 - NB)) : 0;
        const int k2 = k0 + kbA + (pair_launch ? kbB : 0); // first row of the trailing matrix (= m: nothing below)
        const int nrb = (m - k2 + NB - 1) / NB; // row blocks below
        const int nsr = (nrb + 1) / 2;          // 2 x 2 groups of tiles per side
        const int n_stiles = nrb == 0 ? 0 : 1 + (nrb >= 2 ? nsr * (nsr + 1) / 2 : 0); // look-ahead tile + groups
        const int n_rhs_blocks = max(1, (m - k2 + 63) / 64); // right-hand-side blocks, 64 rows each
        // pair launches of a large fp32 map: 64 columns of B per workgroup (chol_pair.h, b_pair_rows_wide)
        const bool b_wide = pair_launch && kbB > 0 && sizeof(TB) == 4 && n_bblocks > e->n_cus;
        const int n_bw = b_wide ? n_pad / (2 * NB) : n_bblocks;
        const int n_wgs = n_stiles + n_rhs_blocks + n_bw;
        const int spacer = n_wgs > e->n_cus ? e->n_cus : 0; // see k_chol_step: empty blocks where the look-ahead workgroup's CU comes round again
        const int n_spacers = spacer ? (n_wgs - 1) / (spacer - 1) : 0; // blocks spacer, 2 spacer, ... among n_wgs + n_spacers
        unsigned long long *tr = nullptr;
        int tr_abl = 0;
#ifdef EKF_SWEEP_TRACE
        if (g_trace && g_trace_n < TRACE_MAX) {
            tr = g_trace + (size_t)TRACE_SLOTS * g_trace_n++;
            static unsigned long long tags[TRACE_MAX];
            tags[g_trace_n - 1] = ((unsigned long long)k0 << 32) | (unsigned long long)m | (pair_launch ? 1ull << 31 : 0ull);
            (void)hipMemcpyAsync(tr + 5, &tags[g_trace_n - 1], sizeof(unsigned long long), hipMemcpyHostToDevice, s);
            tr_abl = g_trace_abl;
        }
#endif
        if (pair_launch) {
#define PAIR_ARGS e->d.S, e->d.LL, sizeof(TB) == 4 ? e->d.LLf : nullptr, ldS, m, m_pad, k0, kbA, kbB, k2, e->d.nu, n_stiles, V, W, Wf, ldw,      \
                  e->d.counts, sizeof(TB) == 4 ? e->d.Gc : nullptr, e->d.zvec, e->d.Bc, G, A, ld, n_bw, n_rhs_blocks,                          \
                  n_wgs > e->n_cus ? 1 : 0, spacer, tr, b_wide ? 1 : 0
            bool launched = false;
            if constexpr (EXACT) {
                if (planes_b) {
                    k_chol_pair<TB, TB, true><<<n_wgs + n_spacers, 256, 0, s>>>(PAIR_ARGS, bp);
                    launched = true;
                }
            }
            if (!launched) k_chol_pair<TB, TB><<<n_wgs + n_spacers, 256, 0, s>>>(PAIR_ARGS);
#undef PAIR_ARGS
            k0 += have_pair ? 2 * NB : NB;
            have_pair = true;
        } else {
            k_chol_step<TB, TB><<<n_wgs + n_spacers, 256, 0, s>>>(e->d.S, e->d.LL, sizeof(TB) == 4 ? e->d.LLf : nullptr, ldS, m, m_pad, k0, kbA, e->d.nu, n_stiles,
                                                             V, W, Wf, ldw, e->d.counts, sizeof(TB) == 4 ? e->d.Gc : nullptr,
                                                             e->d.zvec, e->d.Bc, G, A, ld, n_bblocks, n_rhs_blocks,
                                                             n_wgs > e->n_cus ? 1 : 0, spacer, tr, tr_abl, bp);
            k0 += NB;
        }
    }
    if (e->timing) {
        (void)hipEventRecord(sw1, s);
        e->sw_events.emplace_back(sw0, sw1);
        if (EXACT && update_cov) { // -> launch_p_update_exact: the bracket "sweep end .. downdate start"
            if (e->px_mid) (void)hipEventDestroy(e->px_mid); // (an update that failed between the two left it behind)
            e->px_mid = nullptr;
            if (hipEventCreate(&e->px_mid) == hipSuccess) (void)hipEventRecord(e->px_mid, s);
        }
        e->sw_m.push_back(b_in_sweep ? m : -m); // negative: the rows of B were not formed in these launches
        e->sw_launch.push_back(n_sweep_launches);
    }
    // inv(L): the 128 x 128 diagonal chunks in one launch, then by doubling 128 -> 256 -> ... until one block covers all rows
    const bool need_inverse = !b_in_sweep;
    if (need_inverse) {
        const int nbk = m_pad / NB, n_chunks = (nbk + INV_CH - 1) / INV_CH;
        if (nbk > 1) k_inv_diag<<<dim3(INV_CH - 1, n_chunks), 256, 0, s>>>(e->d.LL, ldS, V, W, Wf, ldw, nbk);
    }
    for (int sz = INV_CH * NB; need_inverse && sz < m_pad; sz *= 2) {
        const int npairs = (m_pad - sz + 2 * sz - 1) / (2 * sz); // pairs whose second half has rows
        {   // 32x32 output tiles on the fp64 MFMA at every level: these products are small (m^3/3 flop in total) and
            // need many workgroups with short k-loops rather than big tiles (64x64 tiles left 3/4 of the CUs idle)
            const int tiles = (sz / NB) * (sz / NB);
            k_triinv_level<<<npairs * tiles, 256, 0, s>>>(e->d.LL, ldS, m, m_pad, V, W, Wf, e->d.Tbuf, ldw, sz, 0);
            k_triinv_level<<<npairs * tiles, 256, 0, s>>>(e->d.LL, ldS, m, m_pad, V, W, Wf, e->d.Tbuf, ldw, sz, 1);
        }
    }
    // exact configuration above B_SWEEP_MAX rows: the GEMM on the int8 MFMA, straight into the digit planes of B (kernels_pexact.hip)
    bool gemm_planes = EXACT && !b_in_sweep && update_cov && e->d.Wq != nullptr && (!sharded || shard_cols);
#ifdef EKF_SWEEP_TRACE // accuracy experiments of the debug build: B = inv(L) G above 2048 rows by the fp64 GEMM, cut into planes afterwards
    if (std::getenv("EKF_DBG_FP64_GEMM") && !sharded) gemm_planes = false;
#endif
    // covariance updates of the exact and the fp64 configuration: the state update waits behind the downdate and shares a launch
    // with the normalisation of the covariance (k_apply_normalize); the fast fp32 configuration keeps its own tail (k_fix_normalize)
    const bool merged_tail = update_cov && !(sizeof(T) == 4 && !EXACT);
    if (gemm_planes) {
        const int c_lo = shard_cols ? col_rb[e->shard_rank] : 0, c_hi = shard_cols ? col_rb[e->shard_rank + 1] : n_pad;
        launch_b_gemm_planes(e, m, c_lo, c_hi);
    } else if (!b_in_sweep) {   // B = inv(L) G = W' G : one GEMM, k <= row (W upper triangular)
        const int TM = sizeof(TB) == 4 ? 128 : 64;
        XtyArgs g{};
        g.X = sizeof(TB) == 4 ? (const void *)Wf : (const void *)W; g.ldx = ldw;
        g.Y = G; g.ldy = ld;
        g.C = A; g.ldc = ld;
        g.M = m_pad; g.N = n_pad; g.K = m_pad;
        g.row0_first = 0; g.row0_stride = 0; g.m_lim = m_pad;
        g.tri = 2; g.tiles_i = (m_pad + TM - 1) / TM; g.tiles_j = (n_pad + TM - 1) / TM; g.alpha = 1.0;
        if (shard_cols) { // own columns only (tile-rounded outwards: a few columns of the neighbours are formed twice)
            g.tj0 = col_rb[e->shard_rank] / TM;
            g.tiles_j = (col_rb[e->shard_rank + 1] + TM - 1) / TM - g.tj0;
        }
        g.n_split = g.tiles_i / 2; // k-depth of row tile i is ~(i+1) TM: halve the units of the longer half
        launch_xty(e, g, 1, sizeof(TB) == 4, s);
    }
    if (shard_cols) {
        // every rank's column blocks of the digit planes to every rank: pack [column][plane][k group], row-block exchange, unpack;
        // dx = B'z then comes from the planes (the fp64 rows of B exist only for the own columns)
        const int m_k = round_up(m, 32), m16 = m_k / 16;
        if (!planes_b && !gemm_planes) launch_slice_columns(e, m, col_rb[e->shard_rank], col_rb[e->shard_rank + 1]); // B came out of the GEMM in fp64
        launch_planes_move(e, true, m_k, col_rb[e->shard_rank], col_rb[e->shard_rank + 1], 0, 0);
        e->hook_rc = e->exchange_hook(e, EKF_XCHG_BPLANES, e->d.Bstage, (size_t)PX_S * m16 * 16, col_rb, "the digit planes of B");
        if (e->hook_rc) return;
        launch_planes_move(e, false, m_k, 0, n_pad, col_rb[e->shard_rank], col_rb[e->shard_rank + 1]);
        launch_dx_planes(e, m_k);
        const int nt = max(e->N * 6, 1);
        if (!merged_tail)
            k_state_apply<<<(nt + 255) / 256, 256, 0, s>>>(e->d.state, e->d.feat_pos, e->d.feat_type, e->d.feat_covpos,
                                                           e->N, e->d.dx_part, ld, update_cov ? 1 : 0, frz);
    } else if (gemm_planes || planes_b) { // B exists as digit planes only
        launch_dx_planes(e, round_up(m, 32));
        const int nt = max(e->N * 6, 1);
        if (!merged_tail)
            k_state_apply<<<(nt + 255) / 256, 256, 0, s>>>(e->d.state, e->d.feat_pos, e->d.feat_type, e->d.feat_covpos,
                                                           e->N, e->d.dx_part, ld, update_cov ? 1 : 0, frz);
    } else {
        dim3 grid((n + 255) / 256, DX_SPLIT);
        const bool fix = update_cov && sizeof(T) == 4 && !EXACT; // (the exact downdate needs no fp64 repair of the diagonal / camera rows)
        const int avg = e->p_exact_sym ? 0 : 1; // an AVG downdate only exists on an unsharded engine: every row is local
        const double *Bc = nullptr;
        const TB *Gy = nullptr;
        if (sizeof(TB) == 4) { // (with B in fp64 its camera columns and B'z need no fp64 side channel)
            Bc = e->d.Bc; // inv(L) Gc, produced by the right-hand-side blocks of k_chol_step
            // feature columns of dx: with the explicit inverse at hand, (H P)' y with y = inv(L)' z in fp64 (k_yvec);
            // on the sweep path B' z (fp32 B): the same to 1e-6 of the largest component of a block, see DESIGN.md section 6
            Gy = need_inverse ? G : nullptr;
            if (Gy) k_yvec<<<(m + 3) / 4, 256, 0, s>>>(W, ldw, m, e->d.zvec, e->d.yvec);
        }
#define DX_LAUNCH(USEG) k_dx_partial<TB, USEG, T><<<grid, 256, 0, s>>>(A, ld, m, n, e->d.zvec, e->d.dx_part, ld, \
                                             fix ? e->d.sq_part : nullptr, fix ? e->d.cam_part : nullptr, Bc, (const T *)e->d.P, \
                                             e->rm, e->d.diag_save, fix ? e->d.cam_save : nullptr, avg, Gy, e->d.yvec, \
                                             (EXACT && update_cov && !apriori) ? e->d.Bexp : nullptr, e->d.state);
        if (Gy) { DX_LAUNCH(true) } else { DX_LAUNCH(false) }
#undef DX_LAUNCH

        const int nt = max(e->N * 6, 1);
        if (!merged_tail)
            k_state_apply<<<(nt + 255) / 256, 256, 0, s>>>(e->d.state, e->d.feat_pos, e->d.feat_type, e->d.feat_covpos,
                                                           e->N, e->d.dx_part, ld, update_cov ? 1 : 0, frz);
    }
    if (!update_cov) return;
    const bool fix_diag = sizeof(T) == 4 && !EXACT;
    if (EXACT) launch_p_update_exact(e, m, false, true, planes_b || shard_cols || gemm_planes); // (column scales: k_gather + k_dx_partial, or a-priori; planes: the sweep's / the GEMM's / the ranks')
    else launch_p_update(e, m_pad, m);
    if (fix_diag) {
        k_fix_normalize<T><<<(n + 255) / 256, 256, 0, s>>>((T *)e->d.P, ld, n, e->rm, e->d.diag_save, e->d.sq_part, e->d.cam_save,
                                                           e->d.cam_part, ld, e->d.state);
        return;
    }
    const int nb = 1 + (n > 7 ? (n - 7 + 255) / 256 : 0);
    if (merged_tail) {
        const int nt = max(e->N * 6, 1);
        k_apply_normalize<T><<<max(nb, (nt + 255) / 256), 256, 0, s>>>((T *)e->d.P, ld, n, e->d.state, e->rm, e->d.feat_pos, e->d.feat_type,
                                                                       e->d.feat_covpos, e->N, e->d.dx_part, ld, frz);
        return;
    }
    k_normalize_cov<T><<<nb, 256, 0, s>>>((T *)e->d.P, ld, n, e->d.state, e->rm, frz);
}

void launch_update(EkfEngine *e, int M, bool update_cov)
{
    if (M <= 0) return;
    if (e->exact && e->f32) update_impl<float, double, true>(e, M, update_cov);
    else if (e->exact) update_impl<double, double, true>(e, M, update_cov);
    else if (e->f32) update_impl<float, float>(e, M, update_cov);
    else update_impl<double, double>(e, M, update_cov);
}

} // namespace ekf
