// kernels_pexact.hip -- the covariance downdate  P <- P - B' B  with EXACT accumulation (EKF_PRECISION_F32_EXACT).
//
// Replaces covarianceUpdate + symmetrisation (EKF/Update.cpp:214-218, 303) for the "fp32 storage, accurate update"
// configuration.  The reference does every operation in double (Core/Base.h:67, typedef cv::Mat_<double> Matd); the fp32
// configuration (kernels_pupdate.hip) rounds B = inv(L) G to fp32 and accumulates B'B in a k-sequential fp32 chain, which
// costs ~1e-8 absolute on cross-feature entries of ~1e-3 (profiles/r03_parity_attribution.txt).  Here P stays fp32 in HBM,
// B is formed in fp64 and every entry of B'B is an EXACT integer sum:
//
//   * k_col_exp + k_slice_B: column j of B (all m rows) gets one power-of-two scale 2^e_j > max_k |B_kj|; every element is
//     the 38-bit integer X = rint(B 2^(38 - e_j)), cut into PX_S = 5 balanced base-256 digits d_0 (most significant) ..
//     d_4 in [-128, 127]: X = sum_s d_s 256^(4 - s).  Digit planes are stored [k / 16][column][k % 16] bytes, so the 16
//     bytes a lane feeds to v_mfma_i32_32x32x32_i8 are contiguous and a wavefront's piece of a slab is 1 KB of consecutive
//     bytes in HBM and in LDS (global_load_lds_dwordx4, no staging registers).
//   * k_p_update_i8: X_i . X_j = sum_L 256^(8 - L) sum_{s + t = L} d_s(i) . d_t(j).  The levels L = 0 .. 4 (15 digit
//     products per 32 rows of k, each ONE v_mfma_i32_32x32x32_i8 per 32 x 32 block: 32 cycles where the fp32 MFMA needs
//     1024 for the same 32 rows) are accumulated in int32 -- exact while 5 * 2^14 * m < 2^31, i.e. m <= PX_MAX_ROWS = 26208 rows
//     (|d d'| <= 2^14, at most 5 products per level and row of k; ekf_engine_create refuses larger maps in this configuration) --
//     and combined once in fp64; the dropped levels are below 2^-35 of max_i max_j per term.  P_new = fl32(P_old - v): ONE
//     rounding per entry and update, i.e. what fp32 storage of an fp64 update costs.
//   * integer sums are exactly symmetric (level L of (i, j) and of (j, i) are the same multiset of products), so diagonal
//     tiles are written in place and off-diagonal tiles are mirrored: P stays bitwise symmetric.
//
// Work per launch: 15/2 n^2 m int8 MACs on the upper triangle (the ALGORITHMIC count stays n^2 m flop); traffic 2 n^2 w of P
// + the digit planes (5 bytes per element of B, re-read through L2 once per tile row / column).
#include "engine.h"
#include "digit_planes.h"

#include <algorithm>
#include <cstdio>
#include <string>

namespace ekf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------ column scales
// bexp[j] = max over k of the biased exponent field of B[k][j] (0 for an all-zero column): |B_kj| < 2^(bexp - 1022).
// Camera columns (j < 13) come from the fp64 side table Bc when given (see k_chol_step: they ride through the sweep as
// right-hand sides).  Grid (n_pad / 256, PX_KSPLIT); bexp zeroed before the launch.
constexpr int PX_KSPLIT = 8;

__global__ void __launch_bounds__(256)
k_col_exp(const double *B, int ld, int m, int n_pad, const double *Bc, int *bexp)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_pad) return;
    const int per = (m + PX_KSPLIT - 1) / PX_KSPLIT;
    const int k0 = blockIdx.y * per, k1 = min(m, k0 + per);
    int hi = 0;
    const bool cam = Bc && j < 13;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = cam ? Bc[(size_t)(k + u) * 16 + j] : B[(size_t)(k + u) * ld + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) hi = max(hi, __double2hiint(v[u]) & 0x7fffffff);
    }
    for (; k < k1; ++k) {
        const double v = cam ? Bc[(size_t)k * 16 + j] : B[(size_t)k * ld + j];
        hi = max(hi, __double2hiint(v) & 0x7fffffff);
    }
    if (hi >> 20) atomicMax(&bexp[j], hi >> 20);
}

// ------------------------------------------------------------------------------------------------ zero pieces of plane 0
// Digit plane 0 holds the top byte of the 38-bit integers.  A column's scale is set by its largest entry -- a feature's own measurement
// rows, ~300 x its typical entry -- so for everything else the top byte is 0: on the bench's N = 1000 frames 99.8 % of plane 0 is zero
// and only 13 % (low-innovation update) / 33 % (high-innovation update) of its 32-row x 32-column pieces hold anything at all.  Nine
// of the downdate's fifteen digit products have a plane-0 operand; a product whose operand piece is all zeros adds exact zeros, so the
// downdate skips it (k_p_update_i8p: 30 -> ~14-18 MFMAs per step and wavefront) -- same integer sums, bit for bit.  This table says
// which pieces are not all zero: one byte per (32-column block, 16-row group), written by the last pass over the planes before the
// downdate (k_dx_planes; k_slice_B where the planes are cut from an fp64 B) -- every byte a launch of the downdate reads was written
// by that pass of the same update.  Called by all (active) lanes of a wavefront whose lanes 0..31 / 32..63 hold 32 consecutive
// columns each, for one 16-row group kb.
__device__ __forceinline__ void px_flag_plane0(uint8_t *bz, int bz_stride, int col, int kb, bool nonzero)
{
    const unsigned long long bal = __ballot(nonzero);
    const int lane = threadIdx.x & 63;
    const unsigned half = (unsigned)(bal >> (lane & 32));
    const unsigned long long act = __ballot(true);
    // the first active lane of each half writes (the last block of columns may be ragged)
    const unsigned acth = (unsigned)(act >> (lane & 32));
    if (acth != 0u && (lane & 31) == (int)__builtin_ctz(acth)) bz[(size_t)(col >> 5) * bz_stride + kb] = half != 0u ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ digit planes
// One thread per (16-row group kb, column): reads B[16 kb .. 16 kb + 15][column] (the wavefront reads 512 contiguous bytes
// per row), writes 16 bytes per plane at Bq[s][kb][column][0..15] (the wavefront writes 1 KB contiguous per plane).
// Workgroup = 64 columns x 4 row groups; grid (n_pad / 64, m_k / 64).  Rows >= m are zero.
__global__ void __launch_bounds__(256)
k_slice_B(const double *B, int ld, int m, int m_k, const double *Bc, const int *bexp, int8_t *Bq, int ldq, size_t plane_stride, int c_lo,
          int c_hi, uint8_t *bz, int bz_stride)
{
    const int col = (c_lo / 64) * 64 + blockIdx.x * 64 + (threadIdx.x & 63);
    const int kb = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (kb * 16 >= m_k) return; // (uniform in the wavefront)
    if (col < c_lo || col >= c_hi) return; // (row-sharded engines cut their own columns only)
    const bool cam = Bc && col < 13;
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = kb * 16 + i;
        const int kc = min(k, m - 1);
        const double x = cam ? Bc[(size_t)kc * 16 + col] : B[(size_t)kc * ld + col];
        v[i] = k < m ? x : 0.0;
    }
    // X = rint(B 2^(8 PX_S - 2 - e)), e = bexp - 1022: |X| <= 2^(8 PX_S - 2), so the top digit stays within [-65, 65]
    const int sh = 8 * PX_S - 2 - (bexp[col] - 1022);
    unsigned w[PX_S][4];
#pragma unroll
    for (int s = 0; s < PX_S; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[s][q] = 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned long long dw = px_digit_word(v[i], sh); // balanced digits without a carry chain, see digit_planes.h
#pragma unroll
        for (int s = 0; s < PX_S; ++s) w[s][i >> 2] |= px_digit_byte(dw, s) << (8 * (i & 3));
    }
#pragma unroll
    for (int s = 0; s < PX_S; ++s) {
        uint4 o = make_uint4(w[s][0], w[s][1], w[s][2], w[s][3]);
        *(uint4 *)(Bq + (size_t)s * plane_stride + ((size_t)kb * ldq + col) * 16) = o;
    }
    if (bz) px_flag_plane0(bz, bz_stride, col, kb, (w[0][0] | w[0][1] | w[0][2] | w[0][3]) != 0u);
}

// ------------------------------------------------------------------------------------------------ sharded filter
// Row-sharded engines in the exact configuration: rank r forms the rows of B only for ITS column blocks (the
// columns of the state rows it owns: the m^2 n work of B = inv(L) G divided by the number of ranks), from digit planes whose
// column scales need the diagonal of P of those columns; the downdate of its rows then needs every column's planes.
//   k_diag_extract   the diagonal of the owned rows into a table of n floats (completed by an exchange of 4 n bytes)
//   k_planes_pack    the planes of the columns [c_lo, c_hi) into the exchange layout [column][plane][k / 16][16]: a rank's share
//                    is one contiguous run of "rows" of PX_S m / 16 x 16 bytes, which is what the engine's row-block exchange moves
//   k_planes_unpack  the other ranks' columns back into [plane][k / 16][column][16]
//   k_dx_planes      dx = B'z from the planes, every column (the fp64 rows of B exist only for the own columns)
template <typename TP>
__global__ void __launch_bounds__(256) k_diag_extract(const TP *P, int ld, RowMap rm, int n, float *diag)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n && owns_row(rm, j)) diag[j] = (float)P[(size_t)local_row(rm, j) * ld + j]; // (only its exponent is used, rounded up)
}

// P <- 0.5 (P + P') in place, upper triangle mirrored (fp64-stored exact configuration: the first downdate after an arbitrary
// upload; the fp32-stored one has the AVG variant of its downdate kernel for that).  One thread per (i, j), j > i.
__global__ void __launch_bounds__(256) k_symmetrize_P(double *P, int ld, int n)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= n || j <= i) return;
    const double v = 0.5 * P[(size_t)i * ld + j] + 0.5 * P[(size_t)j * ld + i];
    P[(size_t)i * ld + j] = v;
    P[(size_t)j * ld + i] = v;
}

template <bool PACK>
__global__ void __launch_bounds__(256)
k_planes_move(int8_t *Bq, size_t b_stride, int ldq, int m16, int c_lo, int c_hi, int skip_lo, int skip_hi, int8_t *stage)
{
    // one 16-byte group per thread; consecutive threads take consecutive columns of one (plane, k group)
    const int ncol = c_hi - c_lo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)ncol * PX_S * m16) return;
    const int col = c_lo + (int)(idx % ncol);
    const int sk = (int)(idx / ncol), s = sk / m16, kb = sk % m16;
    if (!PACK && col >= skip_lo && col < skip_hi) return; // (unpack: the own columns are already in place)
    uint4 *pl = (uint4 *)(Bq + (size_t)s * b_stride + ((size_t)kb * ldq + col) * 16);
    uint4 *st = (uint4 *)(stage + (((size_t)col * PX_S + s) * m16 + kb) * 16);
    if (PACK) *st = *pl;
    else *pl = *st;
}

__global__ void __launch_bounds__(256)
k_dx_planes(const int8_t *Bq, size_t b_stride, int ldq, int m16, int n, const int *bexp, const double *z, double *part, int ldpart,
            const double *st, uint8_t *bz, int bz_stride)
{
    const int j = blockIdx.x * 256 + threadIdx.x, ks = blockIdx.y;
    // the quaternion as it is before this update, for k_apply_normalize (whose workgroups each need it while one of them rewrites it)
    if (blockIdx.x == 0 && ks == 0 && threadIdx.x < 4) part[(size_t)DX_SPLIT * ldpart + threadIdx.x] = st[ST_X + 3 + threadIdx.x];
    if (j >= n) return;
    const int per = (m16 + DX_SPLIT - 1) / DX_SPLIT;
    const int kb0 = ks * per, kb1 = min(m16, kb0 + per);
    double sum = 0.0;
    // (the next group's five 16-byte loads travel while this one is summed: a thread has only 2-4 groups, one round trip each otherwise)
    uint4 nx[PX_S];
    if (kb0 < kb1) {
#pragma unroll
        for (int s = 0; s < PX_S; ++s) nx[s] = *(const uint4 *)(Bq + (size_t)s * b_stride + ((size_t)kb0 * ldq + j) * 16);
    }
    for (int kb = kb0; kb < kb1; ++kb) {
        uint4 d[PX_S];
#pragma unroll
        for (int s = 0; s < PX_S; ++s) d[s] = nx[s];
        // (this pass sees every byte of the planes right before the downdate: it leaves the table of non-zero pieces of plane 0)
        if (bz) px_flag_plane0(bz, bz_stride, j, kb, (d[0].x | d[0].y | d[0].z | d[0].w) != 0u);
        if (kb + 1 < kb1) {
#pragma unroll
            for (int s = 0; s < PX_S; ++s) nx[s] = *(const uint4 *)(Bq + (size_t)s * b_stride + ((size_t)(kb + 1) * ldq + j) * 16);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            long long X = 0;
#pragma unroll
            for (int s = 0; s < PX_S; ++s) {
                const unsigned w = i < 4 ? d[s].x : (i < 8 ? d[s].y : (i < 12 ? d[s].z : d[s].w));
                X = X * 256 + (long long)(signed char)((w >> (8 * (i & 3))) & 255u);
            }
            sum += (double)X * z[16 * kb + i];
        }
    }
    part[(size_t)ks * ldpart + j] = ldexp(sum, (bexp[j] - 1022) - (8 * PX_S - 2));
}

void launch_slice_columns(EkfEngine *e, int m, int c_lo, int c_hi)
{
    const int m_k = round_up(m, 32);
    if (c_hi <= c_lo) return;
    const int nb = (c_hi - (c_lo / 64) * 64 + 63) / 64;
    k_slice_B<<<dim3(nb, (m_k + 63) / 64), 256, 0, e->stream>>>((const double *)e->d.A, e->ldP, m, m_k, nullptr, e->d.Bexp, e->d.Bq, e->ldP,
                                                              (size_t)e->bq_rows * e->ldP, c_lo, c_hi, nullptr, 0);
}

void launch_diag_extract(EkfEngine *e, float *diag)
{
    if (e->f32) k_diag_extract<float><<<(e->n + 255) / 256, 256, 0, e->stream>>>((const float *)e->d.P, e->ldP, e->rm, e->n, diag);
    else k_diag_extract<double><<<(e->n + 255) / 256, 256, 0, e->stream>>>((const double *)e->d.P, e->ldP, e->rm, e->n, diag);
}

void launch_planes_move(EkfEngine *e, bool pack, int m_k, int c_lo, int c_hi, int skip_lo, int skip_hi)
{
    const int m16 = m_k / 16;
    const long long total = (long long)(c_hi - c_lo) * PX_S * m16;
    if (total <= 0) return;
    const size_t b_stride = (size_t)e->bq_rows * e->ldP;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (pack) k_planes_move<true><<<grid, 256, 0, e->stream>>>(e->d.Bq, b_stride, e->ldP, m16, c_lo, c_hi, 0, 0, e->d.Bstage);
    else k_planes_move<false><<<grid, 256, 0, e->stream>>>(e->d.Bq, b_stride, e->ldP, m16, c_lo, c_hi, skip_lo, skip_hi, e->d.Bstage);
}

void launch_dx_planes(EkfEngine *e, int m_k)
{
    const size_t b_stride = (size_t)e->bq_rows * e->ldP;
    k_dx_planes<<<dim3((e->n + 255) / 256, DX_SPLIT), 256, 0, e->stream>>>(e->d.Bq, b_stride, e->ldP, m_k / 16, e->n, e->d.Bexp, e->d.zvec,
                                                                        e->d.dx_part, e->ldP, e->d.state, e->d.Bz, e->bz_stride);
}

// ------------------------------------------------------------------------------------------------ the downdate
// Workgroup = 512 threads = 8 wavefronts (2 x 4), tile 128 x 128 of the upper triangle; wavefront (wr, wc) owns rows
// 64 wr .. + 63 (two 32 x 32 MFMA blocks) and columns 32 wc .. + 31, five int32 accumulators per block (160 registers).
// Half units (64 of the tile's 128 rows, see build_units): one block per wavefront.
// Slab of one step = 32 rows of k: per plane [I side: 2 k-groups x 128 columns x 16 B][J side: the same] = 8 pieces of 1 KB,
// one per wavefront (global_load_lds_dwordx4), 40 KB per step, two buffers.
template <bool FULL>
__device__ __forceinline__ void px_step(const unsigned char *sb, int offA, int offB, v16i (&acc)[2][PX_S])
{
    // all five J digits, the I digits one plane ahead of their products (b: 20 registers, a: 8 + 8): with every operand of
    // the step requested up front the kernel needs 60 operand registers beside its 160 accumulators and spills in the loop
    v4i b[PX_S];
#pragma unroll
    for (int t = 0; t < PX_S; ++t) b[t] = *(const v4i *)(sb + t * 8192 + 4096 + offB);
    v4i a0 = *(const v4i *)(sb + offA), a1 = a0;
    if (FULL) a1 = *(const v4i *)(sb + offA + 32 * 16);
#pragma unroll
    for (int s = 0; s < PX_S; ++s) {
        v4i na0 = a0, na1 = a1;
        if (s + 1 < PX_S) {
            na0 = *(const v4i *)(sb + (s + 1) * 8192 + offA);
            if (FULL) na1 = *(const v4i *)(sb + (s + 1) * 8192 + offA + 32 * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < PX_S - s; ++t) {
            acc[0][s + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b[t], acc[0][s + t], 0, 0, 0);
            if (FULL) acc[1][s + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b[t], acc[1][s + t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0;
        a1 = na1;
    }
}

template <bool AVG>
__global__ void __launch_bounds__(512, 2)
k_p_update_i8(float *P, int ldp, int n, const int8_t *Bq, int ldq, size_t plane_stride, int m_k, const int *bexp, int per_xcd,
              const int4 *units, const int *counts)
{
    constexpr int TM = 128, MB = 32;
    if (filter_frozen(counts)) return; // the update's sweep failed: P stays as it was (engine.h)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][PX_S * 8192];
    const int4 unit = units[(size_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)]; // XCD-aware work order, see k_p_update
    if (unit.x < 0) return;
    const int ti = unit.x, tj = unit.y;
    const bool full = unit.z < 0;
    const bool diag = ti == tj;
    const int I0 = ti * TM, J0 = tj * TM;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;
    const int rbase = full ? wr * 2 * MB : unit.z * 2 * MB + wr * MB; // first tile row of the wavefront (64 or 32 rows)
    const int kg = lane >> 5, idx = lane & 31;

    // this wavefront's piece of every plane: wavefronts 0..3 the I side (k-group, column half), 4..7 the J side
    const int pside = wv >> 2, pkg = (wv >> 1) & 1, phalf = wv & 1;
    const bool piece_live = full || pside == 1 || phalf == unit.z; // a half unit reads 64 of the 128 I columns
    const int8_t *gsrc = Bq + ((size_t)pkg * ldq + (pside ? J0 : I0) + 64 * phalf + lane) * 16;
    const size_t step_stride = (size_t)2 * ldq * 16;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const int poff = __builtin_amdgcn_readfirstlane(wv * 1024);
#define PX_ISSUE(t, buf)                                                                                                      \
    if (piece_live) {                                                                                                         \
        _Pragma("unroll") for (int s = 0; s < PX_S; ++s)                                                                      \
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + (size_t)s * plane_stride + (size_t)(t) * step_stride),           \
                                             (lptr_t)(&smem[buf][s * 8192 + poff]), 16, 0, 0);                                \
    }

    v16i acc[2][PX_S];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int L = 0; L < PX_S; ++L)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][L][r] = 0;

    const int offA = (kg * TM + rbase + idx) * 16, offB = (kg * TM + wc * MB + idx) * 16;
    const int nk = m_k / 32;
    // (one loop per unit kind: with the whole / half test inside the loop the compiler copies the accumulators every step)
#define PX_LOOP(FULL_)                                                                                                        \
    PX_ISSUE(0, 0)                                                                                                            \
    for (int t = 0; t < nk; ++t) {                                                                                            \
        const int buf = t & 1;                                                                                                \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wavefront's pieces of step t have landed */                  \
        __syncthreads(); /* everyone's have; everyone has left the other buffer (step t - 1) */                               \
        if (t + 1 < nk) { PX_ISSUE(t + 1, buf ^ 1) }                                                                          \
        px_step<FULL_>(smem[buf], offA, offB, acc);                                                                           \
    }
    if (full) { PX_LOOP(true) } else { PX_LOOP(false) }
#undef PX_LOOP
#undef PX_ISSUE
    __syncthreads(); // the slabs become the transpose scratch

    // epilogue: v = 2^(e_i + e_j - 12) sum_L acc_L 256^-L (see the header), P <- fl32(P - v); off-diagonal tiles also write
    // the mirror image through a per-wavefront LDS transpose so that the mirrored stores are row-contiguous
    float *sT = reinterpret_cast<float *>(&smem[0][0]) + wv * MB * (MB + 1);
    const int gj = J0 + wc * MB + idx;
    const int ej = gj < n ? bexp[gj] - 1022 : 0;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        if (x == 1 && !full) continue;
        const int bi = I0 + rbase + x * MB;
        float pv[16];
        int ei[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gi = bi + (r & 3) + 8 * (r >> 2) + 4 * kg;
            const bool ok = gi < n && gj < n;
            ei[r] = gi < n ? bexp[gi] - 1022 : 0;
            if (AVG) pv[r] = ok ? 0.5f * P[(size_t)gi * ldp + gj] + 0.5f * P[(size_t)gj * ldp + gi] : 0.f;
            else pv[r] = ok ? P[(size_t)gi * ldp + gj] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int li = (r & 3) + 8 * (r >> 2) + 4 * kg;
            const int gi = bi + li;
            double tsum = (double)acc[x][PX_S - 1][r];
#pragma unroll
            for (int L = PX_S - 2; L >= 0; --L) tsum = fma(tsum, 1.0 / 256.0, (double)acc[x][L][r]);
            const double v = ldexp(tsum, ei[r] + ej - 12);
            const float out = (float)((double)pv[r] - v);
            // AVG on a diagonal tile: element (i, j) reads P(j, i), which belongs to another wavefront of this tile -- only the
            // i <= j elements work there and write both places (first update after an arbitrary upload only)
            const bool st = gi < n && gj < n && (!(AVG && diag) || gi <= gj);
            if (st) P[(size_t)gi * ldp + gj] = out;
            if (AVG && diag && st && gi != gj) P[(size_t)gj * ldp + gi] = out;
            if (!diag) sT[li * (MB + 1) + idx] = out;
        }
        if (!diag) {
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < MB / 2; ++it) {
                const int c = it * 2 + kg; // column of the block = row of the mirror
                const int mi = bi + idx, mj = J0 + wc * MB + c;
                const float v = sT[idx * (MB + 1) + c];
                if (mi < n && mj < n) P[(size_t)mj * ldp + mi] = v;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

#if PX_S_VALUE == 5 // (the persistent kernel is written out for five digit planes; other counts run k_p_update_i8 -- accuracy experiments)
// The step of the persistent kernel: the same products as px_step, operands fetched by hand-issued ds_read_b128.  The
// compiler does not see these reads, so (a) it cannot put "s_waitcnt vmcnt(0)" in front of them -- it does that to any LDS read
// that MAY alias an LDS-DMA in flight, and with a ring addressed by (step mod 3) every read may: the two slabs that are
// supposed to travel during the step were drained before its first product -- and (b) their waits are written here: LDS
// operations complete in order, so "at most k newer reads outstanding" is exact.  The waits name the registers they
// release ("+v"), which keeps every product behind its wait.
#ifndef PX_ZERO_C
#define PX_ZERO_C 1 // k_p_update_i8p: start a unit's accumulators with C = 0 in the first step's products instead of zeroing 160 registers
#endif
#ifndef PX_SKIP_ZERO
#define PX_SKIP_ZERO 1 // k_p_update_i8p: leave out the products whose plane-0 operand piece is all zeros (px_flag_plane0, px_step_ring_z)
#endif
#ifndef PX_STORE_SLACK
#define PX_STORE_SLACK 0 // k_p_update_i8p: 1 = no vmcnt wait in the first two steps behind an epilogue (its stores get two steps to drain);
                         // measured: no difference (profiles/r06_pu_i8_dual.txt) -- off
#endif
#ifndef PX_PRIO
#define PX_PRIO 0   // k_p_update_i8p: raised wave priority around the products of a step (measured)
#endif
#ifndef PX_ABL
#define PX_ABL 0 // timing ablations of scripts/micro/pu_i8_bench.hip only (WRONG results): 1 no epilogue, 2 no slab loads in the loop,
                 // 4 no barrier, 8 no operand reads
#endif
#define PX_DS_READ(dst, addr, off)                                                                                            \
    do {                                                                                                                      \
        if (PX_ABL & 8) asm volatile("" : "=v"(dst));                                                                         \
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));                                 \
    } while (0)
// PSTR: bytes per plane of a slab in LDS; BOFF: offset of the J side inside a plane's block; FIRST: the unit's first step -- every
// accumulator's first product takes the constant 0 as its C operand instead of a register that would have to be zeroed first (160
// moves per unit and wavefront)
template <bool FULL, int PSTR = 8192, int BOFF = 4096, bool FIRST = false>
__device__ __forceinline__ void px_step_ring(unsigned ldsA, unsigned ldsB, v16i (&acc)[2][PX_S])
{
    v4i b0, b1, b2, b3, b4, a[2][2];
    PX_DS_READ(b0, ldsB, BOFF);
    PX_DS_READ(b1, ldsB, BOFF + PSTR);
    PX_DS_READ(b2, ldsB, BOFF + 2 * PSTR);
    PX_DS_READ(b3, ldsB, BOFF + 3 * PSTR);
    PX_DS_READ(b4, ldsB, BOFF + 4 * PSTR);
    PX_DS_READ(a[0][0], ldsA, 0);
    if (FULL) PX_DS_READ(a[0][1], ldsA, 512);
#define PX_GROUP(s_, cur, nxt)                                                                                                \
    if (s_ + 1 < PX_S) {                                                                                                      \
        PX_DS_READ(a[nxt][0], ldsA, (s_ + 1) * PSTR);                                                                         \
        if (FULL) PX_DS_READ(a[nxt][1], ldsA, (s_ + 1) * PSTR + 512);                                                         \
    }                                                                                                                         \
    if (s_ == 0) {                                                                                                            \
        if (FULL) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(a[0][0]), "+v"(a[0][1])); \
        else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(a[0][0]));          \
    } else if (s_ + 1 < PX_S) {                                                                                               \
        if (FULL) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[cur][0]), "+v"(a[cur][1]));                                    \
        else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a[cur][0]));                                                          \
    } else {                                                                                                                  \
        if (FULL) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur][0]), "+v"(a[cur][1]));                                    \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur][0]));                                                          \
    }                                                                                                                         \
    {                                                                                                                         \
        const v4i bb[PX_S] = {b0, b1, b2, b3, b4};                                                                            \
        _Pragma("unroll") for (int t = 0; t < PX_S - s_; ++t) {                                                               \
            if (FIRST && s_ == 0) { /* level t's first product of the unit */                                                 \
                const v16i zc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                                             \
                acc[0][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[cur][0], bb[t], zc, 0, 0, 0);                             \
                if (FULL) acc[1][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[cur][1], bb[t], zc, 0, 0, 0);                   \
            } else {                                                                                                          \
                acc[0][s_ + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[cur][0], bb[t], acc[0][s_ + t], 0, 0, 0);            \
                if (FULL) acc[1][s_ + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[cur][1], bb[t], acc[1][s_ + t], 0, 0, 0);  \
            }                                                                                                                 \
        }                                                                                                                     \
    }
    PX_GROUP(0, 0, 1)
    PX_GROUP(1, 1, 0)
    PX_GROUP(2, 0, 1)
    PX_GROUP(3, 1, 0)
    PX_GROUP(4, 0, 1)
#undef PX_GROUP
}

// The same step with the all-zero pieces of digit plane 0 left out (px_flag_plane0): za0 / za1 / zb say that the plane-0 piece of the
// wavefront's first / second block of A rows / of its B columns is all zeros at this step -- then the five products a_0 b_t of that
// block (the four a_s b_0, s >= 1) add exact zeros and are neither fetched nor multiplied: 12 MFMAs instead of 30 where all three
// are zero (most steps: the non-zero pieces follow the matched features' own rows along a band).  Same sums, bit for bit.
// LDS reads complete in order, so "all but the last k reads" does not care how many optional reads came before them.
template <bool FULL>
__device__ __forceinline__ void px_step_ring_z(unsigned ldsA, unsigned ldsB, v16i (&acc)[2][PX_S], bool za0, bool za1, bool zb)
{
    const v4i zero4 = {0, 0, 0, 0};
    v4i b0 = zero4, b1, b2, b3, b4, a[2][2];
    a[0][0] = zero4;
    a[0][1] = zero4;
    PX_DS_READ(b1, ldsB, 4096 + 8192);
    PX_DS_READ(b2, ldsB, 4096 + 2 * 8192);
    PX_DS_READ(b3, ldsB, 4096 + 3 * 8192);
    PX_DS_READ(b4, ldsB, 4096 + 4 * 8192);
    if (!zb) PX_DS_READ(b0, ldsB, 4096);
    if (!za0) PX_DS_READ(a[0][0], ldsA, 0);
    if (FULL && !za1) PX_DS_READ(a[0][1], ldsA, 512);
    PX_DS_READ(a[1][0], ldsA, 8192);
    if (FULL) PX_DS_READ(a[1][1], ldsA, 8192 + 512);
    if (FULL) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(a[0][0]), "+v"(a[0][1]));
    else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(a[0][0]));
    {
        const v4i bb[PX_S] = {b0, b1, b2, b3, b4};
        if (!za0) {
#pragma unroll
            for (int t = 0; t < PX_S; ++t)
                if (!(t == 0 && zb)) acc[0][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][0], bb[t], acc[0][t], 0, 0, 0);
        }
        if (FULL && !za1) {
#pragma unroll
            for (int t = 0; t < PX_S; ++t)
                if (!(t == 0 && zb)) acc[1][t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0][1], bb[t], acc[1][t], 0, 0, 0);
        }
    }
#define PXZ_GROUP(s_, cur, nxt)                                                                                               \
    if (s_ + 1 < PX_S) {                                                                                                      \
        PX_DS_READ(a[nxt][0], ldsA, (s_ + 1) * 8192);                                                                         \
        if (FULL) PX_DS_READ(a[nxt][1], ldsA, (s_ + 1) * 8192 + 512);                                                         \
        if (FULL) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[cur][0]), "+v"(a[cur][1]));                                    \
        else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a[cur][0]));                                                          \
    } else {                                                                                                                  \
        if (FULL) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur][0]), "+v"(a[cur][1]));                                    \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[cur][0]));                                                          \
    }                                                                                                                         \
    {                                                                                                                         \
        const v4i bb[PX_S] = {b0, b1, b2, b3, b4};                                                                            \
        _Pragma("unroll") for (int t = 0; t < PX_S - s_; ++t) {                                                               \
            if (t == 0 && zb) continue;                                                                                       \
            acc[0][s_ + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[cur][0], bb[t], acc[0][s_ + t], 0, 0, 0);                \
            if (FULL) acc[1][s_ + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[cur][1], bb[t], acc[1][s_ + t], 0, 0, 0);      \
        }                                                                                                                     \
    }
    PXZ_GROUP(1, 1, 0)
    PXZ_GROUP(2, 0, 1)
    PXZ_GROUP(3, 1, 0)
    PXZ_GROUP(4, 0, 1)
#undef PXZ_GROUP
}

// ---------------------------------------------------------------------------------- the downdate, persistent form
// One workgroup per CU walks its share of the unit list (units slot, slot + slots, ... of its XCD's list: what the hardware
// dispatcher does with one-unit workgroups when every unit takes the same time).  What this buys over k_p_update_i8:
//   * the slab pipeline never drains: a ring of three 40 KB buffers, two steps of global_load_lds in flight (counted
//     vmcnt, raw s_barrier -- __syncthreads() would drain the LDS-DMA), and the ring runs on ACROSS units: the first slabs
//     of the next unit travel during the last steps and the epilogue of this one;
//   * the epilogue's 64 KB read-modify-write of P hides behind MFMA work instead of ending every launch round in a burst:
//     the old values are requested when the unit starts (32 registers), the new ones leave as fire-and-forget stores.
constexpr int PX_RING = 3;

// Zero-piece masks of a unit (px_flag_plane0's table): lane t of a wavefront looks at step 64 c + t's two 16-row groups of one block
// of 32 columns; a ballot makes the chunk's 64-bit mask (bit t set = something there).  Chunk 0 stays in scalar registers, the
// others are parked in the wavefront's 24 PX_ZCH bytes of LDS and fetched back every 64 steps -- through asm: an LDS access the
// compiler can see makes it drain the LDS-DMA ring first (see px_step_ring).
constexpr int PX_ZCH = 8; // chunks of 64 steps whose masks a wavefront keeps: units of up to 512 steps (16384 rows)
__device__ __forceinline__ unsigned px_piece_flags(const uint8_t *bz, int bz_stride, int col, int step, bool on, bool straddle)
{
    const unsigned short *pz = reinterpret_cast<const unsigned short *>(bz + (size_t)(col >> 5) * bz_stride);
    unsigned f = on ? pz[step] : 0u;
    // (a block of 32 columns is one column block of the table -- or, on a row-sharded engine whose owned rows do not start on a
    // multiple of 32, straddles two: then both count)
    if (straddle && (col & 31)) f |= on ? pz[(bz_stride >> 1) + step] : 0u;
    return f;
}
__device__ __forceinline__ void px_zmask_park(unsigned addr, unsigned long long a0, unsigned long long a1, unsigned long long b)
{ // (every lane writes the same three words)
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:%4\n\tds_write_b64 %0, %3 offset:%5"
                 :: "v"(addr), "v"(a0), "v"(a1), "v"(b), "n"(8 * PX_ZCH), "n"(16 * PX_ZCH) : "memory");
}
__device__ __forceinline__ void px_zmask_fetch(unsigned addr, unsigned long long &a0, unsigned long long &a1, unsigned long long &b)
{ // (this wavefront wrote them; LDS is in order per wavefront)
    v2u x, y, z;
    asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:%4\n\tds_read_b64 %2, %3 offset:%5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x), "=&v"(y), "=&v"(z) : "v"(addr), "n"(8 * PX_ZCH), "n"(16 * PX_ZCH) : "memory");
    auto u64 = [](v2u v) {
        return (unsigned long long)__builtin_amdgcn_readfirstlane(v[0]) | ((unsigned long long)__builtin_amdgcn_readfirstlane(v[1]) << 32);
    };
    a0 = u64(x);
    a1 = u64(y);
    b = u64(z);
}


// RECT (row-sharded storage; see k_p_update): the rank owns row tiles, not a triangle -- tile row 0 is the replicated
// camera block (13 live rows), tile row t >= 1 holds the owned global rows rm.r0 + (t - 1) TM ..., stored from local row
// rm.base on; every (row tile, column tile) pair is computed and written in place, nothing is mirrored.  The integer sums of
// (i, j) and (j, i) are the same number, so P[i][j] here is bit for bit P[j][i] on the rank that owns row j.
// TP: storage type of P.  float: one rounding to fp32 per entry and update (EKF_PRECISION_F32_EXACT).  double
// (EKF_PRECISION_F64_EXACT): the same exact sums subtracted from an fp64-stored P -- the arithmetic of the update is unchanged
// (38-bit digits of B under a-priori column scales), the storage no longer rounds; the epilogue stages half a block at a time
// (the staging area keeps its size) and requests a block's old values only when the block before has left its accumulators.
template <bool RECT, typename TP = float>
__global__ void __launch_bounds__(512, 2)
k_p_update_i8p(TP *__restrict__ P, int ldp, int n, const int8_t *__restrict__ Bq, int ldq, size_t plane_stride, int m_k,
               const int *__restrict__ bexp, int per_xcd, const int4 *__restrict__ units, int slots, RowMap rm, const int *__restrict__ counts,
               const uint8_t *__restrict__ bz, int bz_stride)
{
    constexpr int TM = 128, MB = 32, SLAB = PX_S * 8192;
    constexpr int ST = MB + 4; // row stride of the epilogue's staging image: 16-byte aligned rows
    if (filter_frozen(counts)) return; // the update's sweep failed: P stays as it was (engine.h; one scalar load before anything is in flight)
    __shared__ __attribute__((aligned(16))) unsigned char ring[PX_RING * SLAB];
    __shared__ __attribute__((aligned(16))) float sTall[8 * MB * ST];
    __shared__ int sExp[2][2 * TM];
    __shared__ int sMeta[2];
    __shared__ unsigned long long sZ[8 * 3 * PX_ZCH]; // zero-piece masks of the later chunks of a long unit, per wavefront
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;
    const int kg = lane >> 5, idx = lane & 31;
    const int nk = m_k / 32;
    // this workgroup's units: slot, slot + slots, ... of its XCD's list (see k_p_update).  A descriptor is fetched by a SCALAR
    // load (uniform address, lgkm counter): a vector load would sit on the same counter as the LDS-DMA ring and drain it.
    const int4 *ul = units + (size_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    auto unit_at = [&](int k) -> int4 {
        v4i u;
        const int4 *p = ul + (size_t)k * slots;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(u) : "s"(p) : "memory");
        return make_int4(u[0], u[1], u[2], u[3]);
    };
    int n_units = 0;
    for (int u = blockIdx.x >> 3; u < per_xcd; u += slots) {
        if (unit_at(n_units).x < 0) break;
        ++n_units;
    }
    if (n_units == 0) return;
    if (tid == 0) {
        sMeta[0] = ldp;
        sMeta[1] = 0;
    }
    __syncthreads();
    const int total = n_units * nk; // steps of the whole pipeline

    // this wavefront's piece of every plane and step: wavefronts 0..3 the I side (k-group, column half), 4..7 the J side
    // (half units fetch all 128 I columns too: every wavefront then has PX_S loads per step, which keeps the counted wait uniform)
    const int pside = wv >> 2, pkg = (wv >> 1) & 1, phalf = wv & 1;
    const size_t step_stride = (size_t)2 * ldq * 16;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const int poff = wv * 1024;
    int iu = 0, it = 0, ig = 0;      // issue cursor: unit, step, global step
    const int8_t *gsrc;
    // first global row (= column of B) of row tile t
    auto row0 = [&](int t) { return RECT ? (t == 0 ? 0 : rm.r0 + (t - 1) * TM) : t * TM; };
    {
        const int4 u0 = unit_at(0);
        gsrc = Bq + ((size_t)pkg * ldq + (pside ? u0.y * TM : row0(u0.x)) + 64 * phalf + lane) * 16;
    }
#define PXP_ISSUE()                                                                                                           \
    if (ig < total) {                                                                                                         \
        const int rb = (ig % PX_RING) * SLAB + poff;                                                                          \
        _Pragma("unroll") for (int s = 0; s < PX_S; ++s)                                                                      \
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + (size_t)s * plane_stride + (size_t)it * step_stride),            \
                                             (lptr_t)(&ring[rb + s * 8192]), 16, 0, 0);                                       \
        ++ig;                                                                                                                 \
        if (++it == nk) {                                                                                                     \
            it = 0;                                                                                                           \
            ++iu;                                                                                                             \
            if (iu < n_units) {                                                                                               \
                const int4 un = unit_at(iu);                                                                                    \
                gsrc = Bq + ((size_t)pkg * ldq + (pside ? un.y * TM : row0(un.x)) + 64 * phalf + lane) * 16;                  \
            }                                                                                                                 \
        }                                                                                                                     \
    }
    PXP_ISSUE()
    PXP_ISSUE()
    int g = 0; // global step being multiplied
    const unsigned ring_lds = (unsigned)(size_t)(lptr_t)&ring[0]; // LDS byte address of the ring
    float *sT = sTall + wv * MB * ST;
    const bool late = wv >= 4; // see the k-loop
    for (int ui = 0; ui < n_units; ++ui) {
        const int4 unit = unit_at(ui);
        const int ti = __builtin_amdgcn_readfirstlane(unit.x), tj = __builtin_amdgcn_readfirstlane(unit.y);
        const int uz = __builtin_amdgcn_readfirstlane(unit.z);
        const bool full = uz < 0;
        const bool diag = RECT || ti == tj;
        const int I0 = row0(ti), J0 = tj * TM;
        const int p_off = RECT ? (ti == 0 ? 0 : rm.base - rm.r0) : 0; // local minus global row
        const int ilim = RECT ? (ti == 0 ? 13 : rm.r1) : n;
        const int rbase = full ? wr * 2 * MB : uz * 2 * MB + wr * MB; // first tile row of the wavefront (64 or 32 rows)
        const int offA = (kg * TM + rbase + idx) * 16, offB = (kg * TM + wc * MB + idx) * 16;
        if (tid < 2 * TM) { // the tile's row and column scales, for the epilogue
            const int c = (tid < TM ? I0 : J0 - TM) + tid;
            sExp[ui & 1][tid] = c < n ? bexp[c] - 1022 : 0;
        }
        // which steps of this unit find an all-zero piece of digit plane 0 in the wavefront's two blocks of A rows and in its block of
        // B columns (px_piece_flags above).  Units of more than 64 PX_ZCH steps (updates above 16384 rows) multiply everything.
        unsigned long long nzA0 = ~0ull, nzA1 = ~0ull, nzB = ~0ull;
        const int n_ch = (nk + 63) >> 6;
        const bool skipz = PX_SKIP_ZERO && bz != nullptr && n_ch <= PX_ZCH;
        const unsigned sz_lds = (unsigned)(size_t)(lptr_t)&sZ[wv * 3 * PX_ZCH];
        int n_zero = 0;
        if (skipz) {
            const int colA = I0 + rbase;
            for (int c = n_ch - 1; c >= 0; --c) { // (downwards: chunk 0's masks are the ones left in the registers)
                const int step = 64 * c + lane;
                const bool on = step < nk;
                const unsigned fa0 = px_piece_flags(bz, bz_stride, colA, step, on, RECT), fa1 = px_piece_flags(bz, bz_stride, colA + MB, step, on, RECT);
                const unsigned fb = px_piece_flags(bz, bz_stride, J0 + wc * MB, step, on, RECT);
                nzA0 = __ballot(fa0 != 0u);
                nzA1 = __ballot(fa1 != 0u);
                nzB = __ballot(fb != 0u);
                const int cnt = nk - 64 * c < 64 ? nk - 64 * c : 64;
                n_zero += (cnt - __popcll(nzA0)) + (full ? cnt - __popcll(nzA1) : 0) + (cnt - __popcll(nzB));
                if (n_ch > 1) px_zmask_park(sz_lds + 8u * c, nzA0, nzA1, nzB);
            }
        }
        auto z_fetch = [&](int c) { px_zmask_fetch(sz_lds + 8u * c, nzA0, nzA1, nzB); };
        // a unit takes the skipping form of the step when at least a quarter of its plane-0 pieces are zero: on a fresh map 85-90 % of
        // them are, once the filter has converged the columns of B are flat and 20-40 % are (profiles/r06_plane0_pieces.txt) -- and the
        // skipping step costs ~4 % where it has nothing to skip (its optional reads and branches)
        const bool sparse_unit = __builtin_amdgcn_readfirstlane(skipz && 4 * n_zero >= (full ? 3 : 2) * nk ? 1 : 0) != 0;
        v16i acc[2][PX_S];
#if !PX_ZERO_C || PX_SKIP_ZERO // (PX_ZERO_C: the unit's first step starts the accumulators with the constant 0 as C; not with skipped products)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int L = 0; L < PX_S; ++L)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][L][r] = 0;
#endif
#define PXP_STEP(FULL_, FIRST_, T_, SPARSE_)                                                                                         \
    {                                                                                                                         \
        /* step g has landed (this wavefront's pieces): everything but the PX_S loads of step g + 1 is complete.           */ \
        /* The first two steps of a unit behind an epilogue need no wait: their slabs were requested before that epilogue  */ \
        /* and the epilogue waited for its old values of P, which are YOUNGER requests (the counter retires in order) --   */ \
        /* and a wait here would be a wait for the epilogue's STORES (vmcnt counts them too): the first step of every unit */ \
        /* stood until ~27 of its predecessor's 32 sixteen-byte stores had been acknowledged.                              */ \
        if (!(PX_STORE_SLACK && ui > 0 && (T_) < 2 && !(PX_ABL & 1))) {                                                       \
            if (g + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PX_S) : "memory");                                    \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
        }                                                                                                                     \
        if (!(PX_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* everyone's have; everyone has left buffer g - 1 */  \
        /* wavefronts w and w + 4 share a SIMD: one of them requests its pieces of step g + 2 before its products, the   */    \
        /* other after them, so that one multiplies while the other issues (an LDS-DMA costs 60-180 issue cycles)      */    \
        if (!late && !(PX_ABL & 2)) { PXP_ISSUE() }                                                                           \
        if (PX_PRIO) __builtin_amdgcn_s_setprio(2);                                                                           \
        if (PX_SKIP_ZERO && (SPARSE_)) {                                                                                      \
            if ((T_) > 0 && ((T_) & 63) == 0) z_fetch((T_) >> 6);                                                             \
            const bool za0_ = !((nzA0 >> ((T_) & 63)) & 1ull), za1_ = !((nzA1 >> ((T_) & 63)) & 1ull), zb_ = !((nzB >> ((T_) & 63)) & 1ull);       \
            px_step_ring_z<FULL_>(ring_lds + (g % PX_RING) * SLAB + offA, ring_lds + (g % PX_RING) * SLAB + offB, acc, za0_, za1_, zb_); \
        } else                                                                                                                \
        px_step_ring<FULL_, 8192, 4096, FIRST_>(ring_lds + (g % PX_RING) * SLAB + offA, ring_lds + (g % PX_RING) * SLAB + offB, acc); \
        if (PX_PRIO) __builtin_amdgcn_s_setprio(0);                                                                           \
        if (late && !(PX_ABL & 2)) { PXP_ISSUE() }                                                                            \
        ++g;                                                                                                                  \
    }
#define PXP_LOOP(FULL_, SPARSE_)                                                                                              \
    PXP_STEP(FULL_, (PX_ZERO_C != 0 && !PX_SKIP_ZERO), 0, SPARSE_)                                                            \
    for (int t = 1; t < nk; ++t) PXP_STEP(FULL_, false, t, SPARSE_)
        if (sparse_unit) {
            if (full) { PXP_LOOP(true, true) } else { PXP_LOOP(false, true) }
        } else {
            if (full) { PXP_LOOP(true, false) } else { PXP_LOOP(false, false) }
        }
#undef PXP_LOOP
#undef PXP_STEP
        if (PX_ABL & 1) {
            int h = 0;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int L = 0; L < PX_S; ++L)
#pragma unroll
                    for (int r = 0; r < 16; ++r) h ^= acc[x][L][r];
            if (h == 0x12345677) P[0] = (TP)0;
            continue;
        }
        // epilogue: v = 2^(e_i + e_j - 12) sum_L acc_L 256^-L, P <- fl32(P - v); off-diagonal tiles also write the mirror image
        // (the epilogue's addresses are formed from a copy of ldp the compiler cannot see through -- read back from LDS --:
        // otherwise it computes all 64 store addresses BEFORE the k-loop and spills them, 190 registers, around it)
        if constexpr (sizeof(TP) == 4) {
        float *Pe = reinterpret_cast<float *>(P);
        const int lde = __builtin_amdgcn_readfirstlane(sMeta[0]); // = ldp, read back from LDS after the loop
        float *sTe = sT + __builtin_amdgcn_readfirstlane(sMeta[1]); // + 0
        const int *se = sExp[ui & 1];
        const int ej = se[TM + wc * MB + idx];
        float *pe = Pe + (size_t)(I0 + p_off + rbase) * lde + J0 + wc * MB;
        const int le = 4 * kg * lde + idx;
        // P is addressed as (wavefront-uniform origin) + (lane offset) + (uniform row step).  Rows / columns up to the tile
        // grid's edge exist (the engine allocates P to a multiple of 128 rows), so the loads need no guards; the stores are
        // guarded.  The old values are requested here, not before the k-loop: 16 or 32 more live registers beside the 160
        // accumulators spill inside the loop.  The stores are fire-and-forget: they drain under the next unit's k-loop.
        float pv0[16], pv1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pv0[r] = (pe + ((r & 3) + 8 * (r >> 2)) * lde)[le];
        if (full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pv1[r] = (pe + (MB + (r & 3) + 8 * (r >> 2)) * lde)[le];
        }
        float *pm = Pe + (size_t)(J0 + wc * MB) * lde + I0 + rbase; // mirror image of block (0, .) (never used when RECT)
        // Both images of a block leave through the wavefront's staging area as 16-byte stores (a store instruction costs the
        // same issue time whatever its width: 64 four-byte stores per lane and unit were most of a unit's fixed cost):
        //   direct image  [row][column], rows of ST floats: lane (q8, q4) reads columns 4 q4 .. + 3 of rows 8 it + q8;
        //   mirror image  [column][row]: the same read pattern gives rows 4 q4 .. + 3 of the mirror's row 8 it + q8.
        const int q8 = lane >> 3, q4 = lane & 7;
        float *sTd = sTe + 4 * kg * ST + idx;  // + (row of the register) x ST: an immediate offset
        float *sTt = sTe + idx * ST + 4 * kg;  // + (row of the register)
        const float4 *sTq = reinterpret_cast<const float4 *>(sTe + q8 * ST + 4 * q4);
        const int lq = q8 * lde + 4 * q4;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (x == 1 && !full) continue;
            const int bi = I0 + rbase + x * MB;
            float out[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int li = (r & 3) + 8 * (r >> 2) + 4 * kg;
                double tsum = (double)acc[x][PX_S - 1][r];
#pragma unroll
                for (int L = PX_S - 2; L >= 0; --L) tsum = fma(tsum, 1.0 / 256.0, (double)acc[x][L][r]);
                const double v = ldexp(tsum, se[rbase + x * MB + li] + ej - 12);
                out[r] = (float)((double)(x == 0 ? pv0[r] : pv1[r]) - v);
                sTd[((r & 3) + 8 * (r >> 2)) * ST] = out[r];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const float4 v = sTq[i4 * 8 * ST / 4];
                const int gi = bi + 8 * i4 + q8, gj0 = J0 + wc * MB + 4 * q4;
                float *dst = (pe + (x * MB + 8 * i4) * lde) + lq;
                if (gi < ilim) {
                    if (gj0 + 3 < n) *reinterpret_cast<float4 *>(dst) = v;
                    else { // the ragged last column tile (n is not a multiple of 4): the padding stays untouched
                        if (gj0 < n) dst[0] = v.x;
                        if (gj0 + 1 < n) dst[1] = v.y;
                        if (gj0 + 2 < n) dst[2] = v.z;
                    }
                }
            }
            if (!diag) { // rows of an off-diagonal tile are all < n (its row range ends before its column range starts)
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) sTt[(r & 3) + 8 * (r >> 2)] = out[r];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const float4 v = sTq[i4 * 8 * ST / 4];
                    const int mj = J0 + wc * MB + 8 * i4 + q8; // row of the mirror = column of the block
                    if (mj < n) *reinterpret_cast<float4 *>((pm + x * MB + 8 * i4 * lde) + lq) = v;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
        } else {
            // ---- fp64-stored P
            double *Pe = reinterpret_cast<double *>(P);
            const int lde = __builtin_amdgcn_readfirstlane(sMeta[0]); // = ldp, read back from LDS after the loop
            double *sTe = reinterpret_cast<double *>(sT) + __builtin_amdgcn_readfirstlane(sMeta[1]); // + 0
            const int *se = sExp[ui & 1];
            const int ej = se[TM + wc * MB + idx];
            double *pe = Pe + (size_t)(I0 + p_off + rbase) * lde + J0 + wc * MB;
            const int le = 4 * kg * lde + idx;
            double *pm = Pe + (size_t)(J0 + wc * MB) * lde + I0 + rbase; // mirror image of block (0, .) (never used when RECT)
            // staging per HALF block (16 rows): direct image [row][column] with rows of ST2 doubles, read back as 16 bytes = two
            // columns per lane (lane = (row q, column pair c2), four rows per pass); mirror image [column][row] with rows of ST2T
            constexpr int ST2 = MB + 2, ST2T = 16 + 2;
            static_assert(16 * ST2 * 8 <= MB * ST * 4 && MB * ST2T * 8 <= MB * ST * 4, "the staging area holds half a block in fp64");
            const int qd = lane >> 4, c2 = lane & 15;   // direct read-back: row 4 i4 + qd, columns 2 c2, 2 c2 + 1
            const int qm = lane >> 3, c2m = lane & 7;   // mirror read-back: mirror row 8 i4 + qm, columns 2 c2m, 2 c2m + 1
            // half a block (eight registers = sixteen rows) at a time: its old values were requested while the half before left
            double pvn[8];
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) pvn[r8] = (pe + ((r8 & 3) + 8 * (r8 >> 2)) * lde)[le];
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                if (x == 1 && !full) continue;
                const int bi = I0 + rbase + x * MB;
#pragma unroll
                for (int h = 0; h < 2; ++h) { // registers 8 h .. 8 h + 7 = rows 16 h .. 16 h + 15 of the block
                    double out[8];
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = 8 * h + r8;
                        const int li = (r & 3) + 8 * (r >> 2) + 4 * kg;
                        double tsum = (double)acc[x][PX_S - 1][r];
#pragma unroll
                        for (int L = PX_S - 2; L >= 0; --L) tsum = fma(tsum, 1.0 / 256.0, (double)acc[x][L][r]);
                        out[r8] = pvn[r8] - ldexp(tsum, se[rbase + x * MB + li] + ej - 12);
                    }
                    if (h == 0 || (x == 0 && full)) { // the next half's old values
                        const int xn = h == 0 ? x : 1, hn = h ^ 1;
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) pvn[r8] = (pe + (xn * MB + 16 * hn + (r8 & 3) + 8 * (r8 >> 2)) * lde)[le];
                    }
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) sTe[((r8 & 3) + 8 * (r8 >> 2) + 4 * kg) * ST2 + idx] = out[r8];
                    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const double2 v = *reinterpret_cast<const double2 *>(sTe + (4 * i4 + qd) * ST2 + 2 * c2);
                        const int gi = bi + 16 * h + 4 * i4 + qd, gj0 = J0 + wc * MB + 2 * c2;
                        double *dst = pe + (size_t)(x * MB + 16 * h + 4 * i4 + qd) * lde + 2 * c2;
                        if (gi < ilim) {
                            if (gj0 + 1 < n) *reinterpret_cast<double2 *>(dst) = v;
                            else if (gj0 < n) dst[0] = v.x; // (n is odd: the padding stays untouched)
                        }
                    }
                    if (!diag) { // rows of an off-diagonal tile are all < n (its row range ends before its column range starts)
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) sTe[idx * ST2T + (r8 & 3) + 8 * (r8 >> 2) + 4 * kg] = out[r8];
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) {
                            const double2 v = *reinterpret_cast<const double2 *>(sTe + (8 * i4 + qm) * ST2T + 2 * c2m);
                            const int mj = J0 + wc * MB + 8 * i4 + qm; // row of the mirror = column of the block
                            if (mj < n) *reinterpret_cast<double2 *>(pm + (size_t)(8 * i4 + qm) * lde + x * MB + 16 * h + 2 * c2m) = v;
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
#undef PXP_ISSUE
}


// ---------------------------------------------------------------- the downdate, two independent workgroups per CU (round 6)
// (scripts/micro/pu_i8_bench.hip, variant 4, only: a MEASURED ALTERNATIVE that did not win -- bit-identical results on the whole matrix,
// 117-121 us against 108-115 at m = 298, 274 against 269-283 at m = 1014, 477-492 against 467-486 at m = 2000, N = 1000;
// profiles/r06_pu_i8_dual.txt.  Kept out of the library.)
#ifdef PX_BENCH
// What k_p_update_i8p cannot hide (profiles/r05_pu_i8_bench.txt): its eight wavefronts share ONE ring of slabs and one barrier, so
// all of them leave the k-loop together -- nobody multiplies during a unit's epilogue (6.9 us) and turnover (2.4 us), which at ten
// k-steps per unit (the m < 512 class) is 40 % of the unit.  Here a CU holds TWO independent workgroups of four wavefronts (one per
// SIMD, so a SIMD still runs two wavefronts of 160 accumulators each): each works on its own HALF tile (64 rows x 128 columns:
// wavefront wc = 64 x 32 = two MFMA blocks, as before) with its own ring and its own barrier.  The two drift apart by themselves,
// and whenever one is in its epilogue, waits for a slab or fetches a unit, the other one owns the MFMA pipe.
//   * slab of one step = 32 rows of k: per plane [I side: 2 k-groups x 64 columns x 16 B][J side: 2 k-groups x 128 columns x 16 B]
//     = 6 pieces of 1 KB, 30 KB per step (the J side is fetched per half tile: 1.5 x the L2 -> LDS traffic of the shared ring);
//     a wavefront issues eight LDS-DMA pieces per step (30 pieces dealt to four wavefronts, two of them fetch one piece twice:
//     a uniform count keeps the wait a constant); ring of TWO slabs (80 KB of LDS per workgroup), one step in flight;
//   * units are handed out dynamically: one counter per XCD list, a workgroup fetches the unit after next while it starts the
//     current one (the ring runs on across units), so a CU's two workgroups need no static balance.  The counters of the NEXT
//     launch are zeroed by this one (two sets, alternating);
//   * the accumulators are not zeroed: the first step's products take the constant 0 as C (px_step_ring<.., FIRST>).
// Same digits, same integer sums, same fp64 combination, same single rounding: bit-identical to k_p_update_i8p.
template <bool RECT, typename TP = float>
__global__ void __launch_bounds__(256, 2)
k_p_update_i8d(TP *__restrict__ P, int ldp, int n, const int8_t *__restrict__ Bq, int ldq, size_t plane_stride, int m_k,
               const int *__restrict__ bexp, int per_xcd, const int4 *__restrict__ units, RowMap rm, const int *__restrict__ counts,
               unsigned *__restrict__ ctr, int parity)
{
    constexpr int TM = 128, TH = 64, MB = 32, PSTR = 6144, SLAB = PX_S * PSTR;
    constexpr int ST = MB + 4; // row stride of the epilogue's staging image: 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) unsigned char ring[2 * SLAB];
    __shared__ __attribute__((aligned(16))) float sTall[4 * MB * ST];
    __shared__ int sExp[2][TH + TM];
    __shared__ int4 sUnit[3];
    __shared__ int sMeta[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // = wc: the wavefront's 32 columns of the tile
    const int kg = lane >> 5, idx = lane & 31;
    const int nk = m_k / 32;
    const int xcd = blockIdx.x & 7;
    // the other set of counters, for the next launch (also when this launch does nothing: the sets alternate per LAUNCH)
    if (blockIdx.x < 8 && tid == 0) ctr[(parity ^ 1) * 8 + blockIdx.x] = 0u;
    if (filter_frozen(counts)) return; // the update's sweep failed: P stays as it was (engine.h)
    const int4 *ul = units + (size_t)xcd * per_xcd;
    unsigned *my_ctr = ctr + parity * 8 + xcd;
    // next unit of this XCD's list (wave-uniform; called by wavefront 0 only).  The descriptor is fetched by a SCALAR load.
    auto grab = [&]() -> int4 {
        unsigned k = 0u;
        if (lane == 0) k = atomicAdd(my_ctr, 1u);
        k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
        if (k >= (unsigned)per_xcd) return make_int4(-1, -1, -1, 0);
        v4i u;
        const int4 *p = ul + k;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(u) : "s"(p) : "memory");
        return make_int4(u[0], u[1], u[2], u[3]);
    };
    if (wv == 0) {
        const int4 u0 = grab();
        const int4 u1 = u0.x < 0 ? u0 : grab();
        if (lane == 0) {
            sUnit[0] = u0;
            sUnit[1] = u1;
            sMeta[0] = ldp;
            sMeta[1] = 0;
        }
    }
    __syncthreads();
    auto unit_at = [&](int ui) -> int4 {
        const int4 u = sUnit[ui % 3];
        return make_int4(__builtin_amdgcn_readfirstlane(u.x), __builtin_amdgcn_readfirstlane(u.y), __builtin_amdgcn_readfirstlane(u.z), 0);
    };
    if (unit_at(0).x < 0) return;
    // first global row (= column of B) of row tile t
    auto row0 = [&](int t) { return RECT ? (t == 0 ? 0 : rm.r0 + (t - 1) * TM) : t * TM; };

    // ---- the slab pipeline.  A plane of a slab has six places of 1 KB: 0, 1 = I side k-group 0, 1; 2 .. 5 = J side k-group (p - 2) / 2,
    // column half (p - 2) % 2.  The wavefronts of parity par = wv & 1 own the places par, par + 2, par + 4 (a = 0, 1, 2) of all five
    // planes: fifteen pieces r = 3 s + a for two wavefronts -- hi = wv >> 1 = 0 takes r = 0 .. 7, hi = 1 takes r = 14 .. 7 (piece 7
    // is fetched twice: the same eight instructions for everybody, a constant wait).  Piece j of a wavefront: plane s_j = j / 3, place
    // a_j = j % 3, mirrored (4 - s_j, 2 - a_j) for hi = 1 -- every address is (a scalar that moves by a constant with j) + (one of
    // three lane offsets), which is what the LDS-DMA's scalar-base form wants.
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const size_t step_stride = (size_t)2 * ldq * 16;
    const int par = wv & 1, hi = wv >> 1;
    const long long plane_step = hi ? -(long long)plane_stride : (long long)plane_stride; // from piece plane s_j to s_j + 1
    const int8_t *plane_base = Bq + (hi ? 4 * plane_stride : (size_t)0);
    const int lds_plane0 = hi ? 4 * PSTR : 0, lds_plane_step = hi ? -PSTR : PSTR;
    const int lds_place0 = (par + (hi ? 4 : 0)) * 1024, lds_place_step = hi ? -2048 : 2048; // place a_j -> par + 2 a_j (mirrored: 2 - a_j)
    int iu = 0, it = 0, ig = 0; // issue cursor: unit, step, global step
    bool issue_live = true;
    unsigned voff[3]; // lane offsets of the three places of the unit being issued (in piece order a_j = 0, 1, 2)
    auto point = [&](const int4 &u) {
        const int colI = row0(u.x) + TH * u.z, colJ = u.y * TM + 64 * par;
        const unsigned vI = (unsigned)((par * ldq + colI + lane) * 16);  // place par: I side, k-group par
        const unsigned vJ0 = (unsigned)((colJ + lane) * 16);             // place par + 2: J side, k-group 0, column half par
        const unsigned vJ1 = (unsigned)((ldq + colJ + lane) * 16);       // place par + 4: J side, k-group 1
        voff[0] = hi ? vJ1 : vI;
        voff[1] = vJ0;
        voff[2] = hi ? vI : vJ1;
    };
    point(unit_at(0));
#define PXD_ISSUE()                                                                                                           \
    if (issue_live) {                                                                                                         \
        const bool last_ = it + 1 == nk;                                                                                      \
        int4 un_ = make_int4(0, 0, 0, 0);                                                                                     \
        if (last_) un_ = unit_at(iu + 1); /* (an LDS read: before the pieces are in flight, or the compiler drains them) */  \
        const int8_t *sb_ = plane_base + (size_t)it * step_stride;                                                            \
        const int rb_ = (ig & 1) * SLAB + lds_plane0 + lds_place0;                                                            \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                       \
            unsigned vo_ = voff[j % 3];                                                                                       \
            asm volatile("" : "+v"(vo_)); /* a fresh 32-bit offset: scalar base + lane offset stays visible as such */        \
            __builtin_amdgcn_global_load_lds((gptr_t)(sb_ + (long long)(j / 3) * plane_step + vo_),                           \
                                             (lptr_t)(&ring[rb_ + (j / 3) * lds_plane_step + (j % 3) * lds_place_step]), 16, 0, 0); \
        }                                                                                                                     \
        ++ig;                                                                                                                 \
        if (last_) {                                                                                                          \
            it = 0;                                                                                                           \
            ++iu;                                                                                                             \
            if (un_.x < 0) issue_live = false;                                                                                \
            else point(un_);                                                                                                  \
        } else ++it;                                                                                                          \
    }
    PXD_ISSUE()
    int g = 0; // global step being multiplied
    const unsigned ring_lds = (unsigned)(size_t)(lptr_t)&ring[0];
    float *sT = sTall + wv * MB * ST;
    // operand addresses inside a slab: A = the half tile's rows (two blocks of 32: + 512), B = this wavefront's 32 columns
    const unsigned offA = (unsigned)(kg * 1024 + idx * 16), offB = (unsigned)(kg * 2048 + (wv * MB + idx) * 16);
    for (int ui = 0;; ++ui) {
        const int4 unit = unit_at(ui);
        if (unit.x < 0) break;
        const int ti = unit.x, tj = unit.y, uz = unit.z;
        const bool diag = RECT || ti == tj;
        const int I0 = row0(ti) + TH * uz, J0 = tj * TM; // first row / column of the half tile
        const int p_off = RECT ? (ti == 0 ? 0 : rm.base - rm.r0) : 0; // local minus global row
        const int ilim = RECT ? (ti == 0 ? 13 : rm.r1) : n;
        if (tid < TH + TM) { // the half tile's row and column scales, for the epilogue
            const int c = (tid < TH ? I0 : J0 - TH) + tid;
            sExp[ui & 1][tid] = c < n ? bexp[c] - 1022 : 0;
        }
        v16i acc[2][PX_S];
        // (the first step of a unit on its own: its products start the accumulators, and wavefront 0 fetches the unit after next --
        // its descriptor is read two barriers from here at the earliest)
#define PXD_STEP(FIRST_)                                                                                                      \
    {                                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wavefront's pieces of step g have landed */                  \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* everyone's have; everyone has left the other buffer */ \
        if (FIRST_ && wv == 0) {                                                                                              \
            const int4 un = grab();                                                                                           \
            if (lane == 0) sUnit[(ui + 2) % 3] = un;                                                                          \
        }                                                                                                                     \
        PXD_ISSUE()                                                                                                           \
        const unsigned sb = ring_lds + (unsigned)(g & 1) * SLAB;                                                              \
        px_step_ring<true, PSTR, 2048, FIRST_>(sb + offA, sb + offB, acc);                                                    \
        ++g;                                                                                                                  \
    }
        PXD_STEP(true)
        for (int t = 1; t < nk; ++t) PXD_STEP(false)
#undef PXD_STEP
        // epilogue: v = 2^(e_i + e_j - 12) sum_L acc_L 256^-L, P <- fl32(P - v); off-diagonal tiles also write the mirror image
        // (as in k_p_update_i8p: addresses from a copy of ldp read back from LDS, old values requested here, both images leave
        // through the wavefront's staging area as 16-byte stores)
        if constexpr (sizeof(TP) == 4) {
            float *Pe = reinterpret_cast<float *>(P);
            const int lde = __builtin_amdgcn_readfirstlane(sMeta[0]);
            float *sTe = sT + __builtin_amdgcn_readfirstlane(sMeta[1]);
            const int *se = sExp[ui & 1];
            const int ej = se[TH + wv * MB + idx];
            float *pe = Pe + (size_t)(I0 + p_off) * lde + J0 + wv * MB;
            const int le = 4 * kg * lde + idx;
            float pv0[16], pv1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv0[r] = (pe + ((r & 3) + 8 * (r >> 2)) * lde)[le];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv1[r] = (pe + (MB + (r & 3) + 8 * (r >> 2)) * lde)[le];
            float *pm = Pe + (size_t)(J0 + wv * MB) * lde + I0; // mirror image of block (0, .) (never used when RECT)
            const int q8 = lane >> 3, q4 = lane & 7;
            float *sTd = sTe + 4 * kg * ST + idx;
            float *sTt = sTe + idx * ST + 4 * kg;
            const float4 *sTq = reinterpret_cast<const float4 *>(sTe + q8 * ST + 4 * q4);
            const int lq = q8 * lde + 4 * q4;
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int bi = I0 + x * MB;
                float out[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int li = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    double tsum = (double)acc[x][PX_S - 1][r];
#pragma unroll
                    for (int L = PX_S - 2; L >= 0; --L) tsum = fma(tsum, 1.0 / 256.0, (double)acc[x][L][r]);
                    const double v = ldexp(tsum, se[x * MB + li] + ej - 12);
                    out[r] = (float)((double)(x == 0 ? pv0[r] : pv1[r]) - v);
                    sTd[((r & 3) + 8 * (r >> 2)) * ST] = out[r];
                }
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const float4 v = sTq[i4 * 8 * ST / 4];
                    const int gi = bi + 8 * i4 + q8, gj0 = J0 + wv * MB + 4 * q4;
                    float *dst = (pe + (x * MB + 8 * i4) * lde) + lq;
                    if (gi < ilim) {
                        if (gj0 + 3 < n) *reinterpret_cast<float4 *>(dst) = v;
                        else { // the ragged last column tile (n is not a multiple of 4): the padding stays untouched
                            if (gj0 < n) dst[0] = v.x;
                            if (gj0 + 1 < n) dst[1] = v.y;
                            if (gj0 + 2 < n) dst[2] = v.z;
                        }
                    }
                }
                if (!diag) { // rows of an off-diagonal tile are all < n (its row range ends before its column range starts)
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 16; ++r) sTt[(r & 3) + 8 * (r >> 2)] = out[r];
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const float4 v = sTq[i4 * 8 * ST / 4];
                        const int mj = J0 + wv * MB + 8 * i4 + q8; // row of the mirror = column of the block
                        if (mj < n) *reinterpret_cast<float4 *>((pm + x * MB + 8 * i4 * lde) + lq) = v;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#undef PXD_ISSUE
}
#endif // PX_BENCH (k_p_update_i8d)

// ------------------------------------------------------------------------ the downdate, four wavefronts with 64 x 64 each
// (scripts/micro/pu_i8_bench.hip, variant 3, only: a MEASURED ALTERNATIVE that lost -- bit-identical results, 330 us against 259 us
// at m = 1014, N = 1000; profiles/r04_pu_i8_bench.txt.  Kept out of the library.)
// The idea: k_p_update_i8p reads 15 operands of 1 KB for 30 products per wavefront and step, eight wavefronts 120 KB per step --
// 940 LDS cycles at 128 B/clk beside 960 MFMA cycles per wavefront, plus the 40 KB the LDS-DMA writes.  Here the workgroup is
// FOUR wavefronts (one per SIMD, up to 512 registers each), wavefront (wr, wc) owns 64 x 64 = 2 x 2 MFMA blocks with five int32
// accumulators each (320 registers): 20 operand reads for 60 products, 80 KB per step.  Same units, slabs, ring, scalar
// descriptor loads and epilogue scheme as k_p_update_i8p; a wavefront fetches two 1 KB pieces per plane and step, one plane
// ahead of every product group.  What the ablations say (same switches as PX_ABL): the product stream alone is SLOWER from one
// wavefront per SIMD than from two (190 against 173 us), the LDS-DMA instructions cost 44 us where the second wavefront of
// the SIMD hides 16 of them (28), and the epilogue -- nobody multiplies meanwhile -- 68 against 30 us.  The operand reads were
// never the bound: 17-20 us in either kernel.
#ifdef PX_BENCH
#define PXQ_READ(dst, addr, off)                                                                                              \
    do {                                                                                                                      \
        if (PX_ABL & 8) asm volatile("" : "=v"(dst));                                                                         \
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));                                 \
    } while (0)
template <bool FULL, class ISSUE>
__device__ __forceinline__ void px_step_q(unsigned ldsA, unsigned ldsB, v16i (&acc)[2][2][PX_S], ISSUE &&issue)
{
    v4i b00, b01, b10, b11, b20, b21, b30, b31, b40, b41, a[2][2];
    PXQ_READ(b00, ldsB, 4096);
    PXQ_READ(b01, ldsB, 4096 + 512);
    PXQ_READ(a[0][0], ldsA, 0);
    if (FULL) PXQ_READ(a[0][1], ldsA, 512);
    PXQ_READ(b10, ldsB, 4096 + 8192);
    PXQ_READ(b11, ldsB, 4096 + 8192 + 512);
    PXQ_READ(b20, ldsB, 4096 + 2 * 8192);
    PXQ_READ(b21, ldsB, 4096 + 2 * 8192 + 512);
    PXQ_READ(b30, ldsB, 4096 + 3 * 8192);
    PXQ_READ(b31, ldsB, 4096 + 3 * 8192 + 512);
    PXQ_READ(b40, ldsB, 4096 + 4 * 8192);
    PXQ_READ(b41, ldsB, 4096 + 4 * 8192 + 512);
    PXQ_READ(a[1][0], ldsA, 8192);
    if (FULL) PXQ_READ(a[1][1], ldsA, 8192 + 512);
    // LDS operations complete in order: "at most k newer reads outstanding".  14 reads issued, the products of digit t of the
    // first group need the first 4 + 2 t of them (half units: one read less in front and one less behind: the same counts)
#define PXQ_WAIT(k, ...) asm volatile("s_waitcnt lgkmcnt(" #k ")" : __VA_ARGS__)
    // The products are written out as instructions: with more than 256 registers per lane the compiler puts EVERY MFMA result into
    // the accumulator file (256 registers) and, the 320 not fitting, copies accumulators to and fro in every step.  Here fifteen of
    // the twenty accumulators are pinned to the accumulator file ("a") and the five of block (1, 1) to the vector file ("v"; the
    // epilogue takes that block first, which frees its 80 registers for the rest).  An accumulator is written again four products later at the earliest, so no MFMA-after-MFMA hazard
    // arises; the caller separates the last product from the epilogue's reads (s_nop).
#define PXQ_MFMA(x_, c_, L_, av, bv)                                                                                          \
    if constexpr ((x_) == 1 && (c_) == 1)                                                                                     \
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[x_][c_][L_]) : "v"(av), "v"(bv));                     \
    else                                                                                                                      \
        asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[x_][c_][L_]) : "v"(av), "v"(bv));
#define PXQ_PROD(s_, t_, cur, bt0, bt1)                                                                                       \
    if constexpr ((s_) + (t_) < PX_S) {                                                                                       \
        PXQ_MFMA(0, 0, (s_) + (t_), a[cur][0], bt0)                                                                           \
        PXQ_MFMA(0, 1, (s_) + (t_), a[cur][0], bt1)                                                                           \
        if constexpr (FULL) {                                                                                                 \
            PXQ_MFMA(1, 0, (s_) + (t_), a[cur][1], bt0)                                                                       \
            PXQ_MFMA(1, 1, (s_) + (t_), a[cur][1], bt1)                                                                       \
        }                                                                                                                     \
    }
    issue(0);
    if constexpr (FULL) {
        PXQ_WAIT(10, "+v"(b00), "+v"(b01), "+v"(a[0][0]), "+v"(a[0][1]));
    } else {
        PXQ_WAIT(10, "+v"(b00), "+v"(b01), "+v"(a[0][0]));
    }
    PXQ_PROD(0, 0, 0, b00, b01)
    PXQ_WAIT(8, "+v"(b10), "+v"(b11));
    PXQ_PROD(0, 1, 0, b10, b11)
    PXQ_WAIT(6, "+v"(b20), "+v"(b21));
    PXQ_PROD(0, 2, 0, b20, b21)
    PXQ_WAIT(4, "+v"(b30), "+v"(b31));
    PXQ_PROD(0, 3, 0, b30, b31)
    PXQ_WAIT(2, "+v"(b40), "+v"(b41));
    PXQ_PROD(0, 4, 0, b40, b41)
    // groups 1 .. 4: the I digits of group s + 1 are requested before the products of group s
#define PXQ_GROUP(s_, cur, nxt)                                                                                               \
    issue(s_);                                                                                                                \
    if constexpr ((s_) + 1 < PX_S) {                                                                                          \
        PXQ_READ(a[nxt][0], ldsA, ((s_) + 1) * 8192);                                                                         \
        if constexpr (FULL) {                                                                                                 \
            PXQ_READ(a[nxt][1], ldsA, ((s_) + 1) * 8192 + 512);                                                               \
            PXQ_WAIT(2, "+v"(a[cur][0]), "+v"(a[cur][1]));                                                                    \
        } else {                                                                                                              \
            PXQ_WAIT(1, "+v"(a[cur][0]));                                                                                     \
        }                                                                                                                     \
    } else {                                                                                                                  \
        if constexpr (FULL) { PXQ_WAIT(0, "+v"(a[cur][0]), "+v"(a[cur][1])); } else { PXQ_WAIT(0, "+v"(a[cur][0])); }         \
    }                                                                                                                         \
    PXQ_PROD(s_, 0, cur, b00, b01)                                                                                            \
    PXQ_PROD(s_, 1, cur, b10, b11)                                                                                            \
    PXQ_PROD(s_, 2, cur, b20, b21)                                                                                            \
    PXQ_PROD(s_, 3, cur, b30, b31)
    PXQ_GROUP(1, 1, 0)
    PXQ_GROUP(2, 0, 1)
    PXQ_GROUP(3, 1, 0)
    PXQ_GROUP(4, 0, 1)
#undef PXQ_GROUP
#undef PXQ_PROD
#undef PXQ_MFMA
#undef PXQ_WAIT
}

template <bool RECT>
__global__ void __launch_bounds__(256, 1)
k_p_update_i8q(float *__restrict__ P, int ldp, int n, const int8_t *__restrict__ Bq, int ldq, size_t plane_stride, int m_k,
               const int *__restrict__ bexp, int per_xcd, const int4 *__restrict__ units, int slots, RowMap rm, const int * /*counts*/)
{
    constexpr int TM = 128, MB = 32, SLAB = PX_S * 8192;
    constexpr int ST = MB + 4; // row stride of the epilogue's staging image: 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) unsigned char ring[PX_RING * SLAB];
    __shared__ __attribute__((aligned(16))) float sTall[4 * MB * ST];
    __shared__ int sExp[2][2 * TM];
    __shared__ int sMeta[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 1, wc = wv & 1;
    const int kg = lane >> 5, idx = lane & 31;
    const int nk = m_k / 32;
    const int4 *ul = units + (size_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    auto unit_at = [&](int k) -> int4 { // scalar load: a vector load would sit on the LDS-DMA's counter
        v4i u;
        const int4 *p = ul + (size_t)k * slots;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(u) : "s"(p) : "memory");
        return make_int4(u[0], u[1], u[2], u[3]);
    };
    int n_units = 0;
    for (int u = blockIdx.x >> 3; u < per_xcd; u += slots) {
        if (unit_at(n_units).x < 0) break;
        ++n_units;
    }
    if (n_units == 0) return;
    if (tid == 0) {
        sMeta[0] = ldp;
        sMeta[1] = 0;
    }
    __syncthreads();
    const int total = n_units * nk; // steps of the whole pipeline

    // this wavefront's two pieces of every plane and step: wavefronts 0, 1 the I side (k-group 0, 1), 2, 3 the J side; piece h =
    // columns 64 h .. + 63 of the side
    const int pside = wv >> 1, pkg = wv & 1;
    const size_t step_stride = (size_t)2 * ldq * 16;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const int poff = wv * 2048;
    int iu = 0, it = 0, ig = 0; // issue cursor: unit, step, global step
    const int8_t *gsrc;
    auto row0 = [&](int t) { return RECT ? (t == 0 ? 0 : rm.r0 + (t - 1) * TM) : t * TM; };
    {
        const int4 u0 = unit_at(0);
        gsrc = Bq + ((size_t)pkg * ldq + (pside ? u0.y * TM : row0(u0.x)) + lane) * 16;
    }
    auto issue_plane = [&](int s) { // the two pieces of plane s of step ig; after the last plane the cursor moves on
        if (ig < total) {
            const int rb = (ig % PX_RING) * SLAB + poff + s * 8192;
            const int8_t *src = gsrc + (size_t)s * plane_stride + (size_t)it * step_stride;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&ring[rb]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + 1024), (lptr_t)(&ring[rb + 1024]), 16, 0, 0);
            if (s == PX_S - 1) {
                ++ig;
                if (++it == nk) {
                    it = 0;
                    ++iu;
                    if (iu < n_units) {
                        const int4 un = unit_at(iu);
                        gsrc = Bq + ((size_t)pkg * ldq + (pside ? un.y * TM : row0(un.x)) + lane) * 16;
                    }
                }
            }
        }
    };
#pragma unroll
    for (int s = 0; s < PX_S; ++s) issue_plane(s);
#pragma unroll
    for (int s = 0; s < PX_S; ++s) issue_plane(s);
    int g = 0; // global step being multiplied
    const unsigned ring_lds = (unsigned)(size_t)(lptr_t)&ring[0];
    float *sT = sTall + wv * MB * ST;
    for (int ui = 0; ui < n_units; ++ui) {
        const int4 unit = unit_at(ui);
        const int ti = __builtin_amdgcn_readfirstlane(unit.x), tj = __builtin_amdgcn_readfirstlane(unit.y);
        const int uz = __builtin_amdgcn_readfirstlane(unit.z);
        const bool full = uz < 0;
        const bool diag = RECT || ti == tj;
        const int I0 = row0(ti), J0 = tj * TM;
        const int p_off = RECT ? (ti == 0 ? 0 : rm.base - rm.r0) : 0; // local minus global row
        const int ilim = RECT ? (ti == 0 ? 13 : rm.r1) : n;
        const int rbase = full ? wr * 2 * MB : uz * 2 * MB + wr * MB; // first tile row of the wavefront (64 or 32 rows)
        const int offA = (kg * TM + rbase + idx) * 16, offB = (kg * TM + wc * 2 * MB + idx) * 16;
        { // the tile's row and column scales, for the epilogue
            const int c = (tid < TM ? I0 : J0 - TM) + tid;
            sExp[ui & 1][tid] = c < n ? bexp[c] - 1022 : 0;
        }
        v16i acc[2][2][PX_S];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int L = 0; L < PX_S; ++L)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[x][c][L][r] = 0;
#define PXQ_LOOP(FULL_)                                                                                                       \
    for (int t = 0; t < nk; ++t, ++g) {                                                                                       \
        /* step g has landed (this wavefront's pieces): everything but the 2 PX_S loads of step g + 1 is complete */          \
        if (g + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PX_S) : "memory");                                    \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
        if (!(PX_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* everyone's have; everyone has left buffer g - 1 */ \
        if (PX_ABL & 2) px_step_q<FULL_>(ring_lds + (g % PX_RING) * SLAB + offA, ring_lds + (g % PX_RING) * SLAB + offB, acc, [](int) {}); \
        else px_step_q<FULL_>(ring_lds + (g % PX_RING) * SLAB + offA, ring_lds + (g % PX_RING) * SLAB + offB, acc, issue_plane); \
    }
        if (full) { PXQ_LOOP(true) } else { PXQ_LOOP(false) }
#undef PXQ_LOOP
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // the last products' results, before the epilogue reads them
        if (PX_ABL & 1) {
            int h = 0;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int L = 0; L < PX_S; ++L)
#pragma unroll
                        for (int r = 0; r < 16; ++r) h ^= acc[x][c][L][r];
            if (h == 0x12345677) P[0] = 0.0f;
            continue;
        }
        // epilogue: as k_p_update_i8p, four blocks per wavefront
        float *Pe = P;
        const int lde = __builtin_amdgcn_readfirstlane(sMeta[0]); // = ldp, read back from LDS after the loop
        float *sTe = sT + __builtin_amdgcn_readfirstlane(sMeta[1]); // + 0
        const int *se = sExp[ui & 1];
        float *pe = Pe + (size_t)(I0 + p_off + rbase) * lde + J0 + wc * 2 * MB;
        const int le = 4 * kg * lde + idx;
        float pv[2][2][16];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (x == 1 && !full) continue;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[x][c][r] = (pe + (x * MB + (r & 3) + 8 * (r >> 2)) * lde + c * MB)[le];
        }
        const int q8 = lane >> 3, q4 = lane & 7;
        float *sTd = sTe + 4 * kg * ST + idx;  // + (row of the register) x ST: an immediate offset
        float *sTt = sTe + idx * ST + 4 * kg;  // + (row of the register)
        const float4 *sTq = reinterpret_cast<const float4 *>(sTe + q8 * ST + 4 * q4);
        const int lq = q8 * lde + 4 * q4;
#pragma unroll
        for (int xi = 0; xi < 2; ++xi) {
            const int x = 1 - xi; // block (1, 1) first: its accumulators sit in the vector file
            if (x == 1 && !full) continue;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int c = 1 - ci;
                const int bi = I0 + rbase + x * MB, bj = J0 + (wc * 2 + c) * MB;
                const int ej = se[TM + (wc * 2 + c) * MB + idx];
                float *pm = Pe + (size_t)bj * lde + I0 + rbase + x * MB; // the block's mirror image (never used when RECT)
                float out[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int li = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    double tsum = (double)acc[x][c][PX_S - 1][r];
#pragma unroll
                    for (int L = PX_S - 2; L >= 0; --L) tsum = fma(tsum, 1.0 / 256.0, (double)acc[x][c][L][r]);
                    const double v = ldexp(tsum, se[rbase + x * MB + li] + ej - 12);
                    out[r] = (float)((double)pv[x][c][r] - v);
                    sTd[((r & 3) + 8 * (r >> 2)) * ST] = out[r];
                }
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const float4 v = sTq[i4 * 8 * ST / 4];
                    const int gi = bi + 8 * i4 + q8, gj0 = bj + 4 * q4;
                    float *dst = (pe + (x * MB + 8 * i4) * lde + c * MB) + lq;
                    if (gi < ilim) {
                        if (gj0 + 3 < n) *reinterpret_cast<float4 *>(dst) = v;
                        else { // the ragged last column tile (n is not a multiple of 4): the padding stays untouched
                            if (gj0 < n) dst[0] = v.x;
                            if (gj0 + 1 < n) dst[1] = v.y;
                            if (gj0 + 2 < n) dst[2] = v.z;
                        }
                    }
                }
                if (!diag) { // rows of an off-diagonal tile are all < n (its row range ends before its column range starts)
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int r = 0; r < 16; ++r) sTt[(r & 3) + 8 * (r >> 2)] = out[r];
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        const float4 v = sTq[i4 * 8 * ST / 4];
                        const int mj = bj + 8 * i4 + q8; // row of the mirror = column of the block
                        if (mj < n) *reinterpret_cast<float4 *>((pm + 8 * i4 * lde) + lq) = v;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}
#endif // PX_BENCH (k_p_update_i8q)


// ------------------------------------------------------------------------------- B = inv(L) G as digit planes (m > 2048 rows)
// Above B_SWEEP_MAX rows the sweep only factorises S; inv(L) is formed explicitly (k_inv_diag, k_triinv_level) and B = inv(L) G is
// one GEMM.  In fp64 (k_xty<double>) that GEMM was the largest item of an N = 5000 update after the downdate (m^2 n flop at ~45
// TFLOP/s: 13 ms at m = 8900).  Here it runs on the int8 MFMA with the machinery of the downdate: W = inv(L)' (k-major: row k =
// all rows i of inv(L)) and G (k-major) are cut into digit planes with their columns' true scales (k_col_exp / k_slice_B: both are
// complete before the GEMM), the products of level < PX_S are accumulated exactly, and the epilogue cuts B -- with the a-priori
// column scale sqrt(P_jj), see chol_bplanes.h -- straight into the digit planes the downdate reads; B never exists in fp64
// (dx = B'z comes from the planes, k_dx_planes).  Tile = 128 rows of B x 128 columns, same wavefront layout, ring and issue
// scheme as k_p_update_i8p; the k-range of a row tile ends at its last row (inv(L) is lower triangular).
__global__ void __launch_bounds__(512, 2)
k_b_gemm_i8p(const int8_t *__restrict__ Wq, int ldw, size_t w_stride, const int *__restrict__ wexp, const int8_t *__restrict__ Gq, int ldq,
             size_t g_stride, const int *__restrict__ gexp, int8_t *__restrict__ Bq, size_t b_stride, const int *__restrict__ bexp, int m,
             int tiles_j, int tj0, int n_units_total, int *counts, int c_live0, int n_live, const uint8_t *__restrict__ wz,
             const uint8_t *__restrict__ gz, int bz_stride)
{
    constexpr int TM = 128, MB = 32, SLAB = PX_S * 8192;
    __shared__ __attribute__((aligned(16))) unsigned char ring[PX_RING * SLAB];
    __shared__ __attribute__((aligned(16))) unsigned char sBy[8][PX_S][MB][16]; // per wavefront: digit bytes [plane][column][16 rows]
    __shared__ int sExp[2][2 * TM];
    __shared__ unsigned long long sZ[8 * 3 * PX_ZCH]; // zero-piece masks (k_p_update_i8p)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;
    const int kg = lane >> 5, idx = lane & 31;
    // units: u = blockIdx.x, + gridDim.x, ...; unit u -> row tile (from the BOTTOM: the longest k-ranges first) and column tile
    const int G_ = gridDim.x;
    auto tile_of = [&](int u, int &ti, int &tj) {
        ti = (m + TM - 1) / TM - 1 - u / tiles_j;
        tj = tj0 + u % tiles_j;
    };
    auto steps_of = [&](int ti) { return min((ti + 1) * TM, ((m + 31) / 32 * 32)) / 32; }; // k < first row past the tile (multiples of 32)
    int n_units = 0, total = 0;
    for (int u = blockIdx.x; u < n_units_total; u += G_) {
        int ti, tj;
        tile_of(u, ti, tj);
        total += steps_of(ti);
        ++n_units;
    }
    if (n_units == 0) return;
    const int pside = wv >> 2, pkg = (wv >> 1) & 1, phalf = wv & 1;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const int poff = wv * 1024;
    int iu = 0, it = 0, ig = 0, ink;
    const int8_t *gsrc;
    size_t sstride, pstride;
    auto point = [&](int ui) {
        int ti, tj;
        tile_of(blockIdx.x + ui * G_, ti, tj);
        ink = steps_of(ti);
        if (pside) { gsrc = Gq + ((size_t)pkg * ldq + tj * TM + 64 * phalf + lane) * 16; sstride = (size_t)2 * ldq * 16; pstride = g_stride; }
        else { gsrc = Wq + ((size_t)pkg * ldw + ti * TM + 64 * phalf + lane) * 16; sstride = (size_t)2 * ldw * 16; pstride = w_stride; }
    };
    point(0);
#define BG_ISSUE()                                                                                                            \
    if (ig < total) {                                                                                                         \
        const int rb = (ig % PX_RING) * SLAB + poff;                                                                          \
        _Pragma("unroll") for (int s = 0; s < PX_S; ++s)                                                                      \
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + (size_t)s * pstride + (size_t)it * sstride),                     \
                                             (lptr_t)(&ring[rb + s * 8192]), 16, 0, 0);                                       \
        ++ig;                                                                                                                 \
        if (++it == ink) {                                                                                                    \
            it = 0;                                                                                                           \
            if (++iu < n_units) point(iu);                                                                                    \
        }                                                                                                                     \
    }
    BG_ISSUE()
    BG_ISSUE()
    int g = 0;
    const unsigned ring_lds = (unsigned)(size_t)(lptr_t)&ring[0];
    const bool late = wv >= 4;
    for (int ui = 0; ui < n_units; ++ui) {
        int ti, tj;
        tile_of(blockIdx.x + ui * G_, ti, tj);
        const int nk = steps_of(ti);
        const int I0 = ti * TM, J0 = tj * TM;
        const int rbase = wr * 2 * MB;
        const int offA = (kg * TM + rbase + idx) * 16, offB = (kg * TM + wc * MB + idx) * 16;
        if (tid < 2 * TM) { // scales of the tile's rows of W' (= rows of inv(L)) and of its columns of G, for the epilogue
            const int c = tid < TM ? I0 + tid : J0 + tid - TM;
            sExp[ui & 1][tid] = (tid < TM ? (c < m ? wexp[c] : 1022) : gexp[c]) - 1022;
        }
        v16i acc[2][PX_S];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int L = 0; L < PX_S; ++L)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][L][r] = 0;
        // zero pieces of plane 0 of W' (the wavefront's two blocks of rows of inv(L)) and of G (its block of columns), as in the
        // downdate: inv(L) and G = H P have their large entries at the matched features' own rows and columns on a fresh map
        unsigned long long nzA0 = ~0ull, nzA1 = ~0ull, nzB = ~0ull;
        const int n_ch = (nk + 63) >> 6;
        const bool skipz = PX_SKIP_ZERO && wz != nullptr && n_ch <= PX_ZCH;
        const unsigned sz_lds = (unsigned)(size_t)(lptr_t)&sZ[wv * 3 * PX_ZCH];
        int n_zero = 0;
        if (skipz) {
            for (int c = n_ch - 1; c >= 0; --c) {
                const int step = 64 * c + lane;
                const bool on = step < nk;
                const unsigned fa0 = px_piece_flags(wz, bz_stride, I0 + rbase, step, on, false);
                const unsigned fa1 = px_piece_flags(wz, bz_stride, I0 + rbase + MB, step, on, false);
                const unsigned fb = px_piece_flags(gz, bz_stride, J0 + wc * MB, step, on, false);
                nzA0 = __ballot(fa0 != 0u);
                nzA1 = __ballot(fa1 != 0u);
                nzB = __ballot(fb != 0u);
                const int cnt = nk - 64 * c < 64 ? nk - 64 * c : 64;
                n_zero += 3 * cnt - __popcll(nzA0) - __popcll(nzA1) - __popcll(nzB);
                if (n_ch > 1) px_zmask_park(sz_lds + 8u * c, nzA0, nzA1, nzB);
            }
        }
        const bool sparse_unit = __builtin_amdgcn_readfirstlane(skipz && 4 * n_zero >= 3 * nk ? 1 : 0) != 0;
#define BG_STEP(SPARSE_)                                                                                                      \
    {                                                                                                                         \
        if (g + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PX_S) : "memory");                                        \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                                       \
        if (!late) { BG_ISSUE() }                                                                                             \
        if (SPARSE_) {                                                                                                        \
            if (t > 0 && (t & 63) == 0) px_zmask_fetch(sz_lds + 8u * (t >> 6), nzA0, nzA1, nzB);                              \
            const bool za0_ = !((nzA0 >> (t & 63)) & 1ull), za1_ = !((nzA1 >> (t & 63)) & 1ull), zb_ = !((nzB >> (t & 63)) & 1ull); \
            px_step_ring_z<true>(ring_lds + (g % PX_RING) * SLAB + offA, ring_lds + (g % PX_RING) * SLAB + offB, acc, za0_, za1_, zb_); \
        } else                                                                                                                \
            px_step_ring<true>(ring_lds + (g % PX_RING) * SLAB + offA, ring_lds + (g % PX_RING) * SLAB + offB, acc);          \
        if (late) { BG_ISSUE() }                                                                                              \
    }
        if (sparse_unit) {
            for (int t = 0; t < nk; ++t, ++g) BG_STEP(true)
        } else {
            for (int t = 0; t < nk; ++t, ++g) BG_STEP(false)
        }
#undef BG_STEP
        // epilogue: B_ij = 2^(e_i + e_j - 12) sum_L acc_L 256^-L, cut with the a-priori scale of column j into PX_S digit bytes, through
        // the wavefront's byte image [plane][column][row] into the planes' 16-byte groups (16 consecutive rows of one column)
        const int *se = sExp[ui & 1];
        const int gj = J0 + wc * MB + idx;
        const int ej = se[TM + wc * MB + idx];
        const int shb = 8 * PX_S - 2 - (bexp[gj] - 1022);
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int h = 0; h < 2; ++h) { // rows 16 h .. 16 h + 15 of the block: the accumulator registers 8 h .. 8 h + 7
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 8 * h + rr;
                    const int li = (r & 3) + 8 * (r >> 2) + 4 * kg; // row of the block, 16 h <= li < 16 h + 16
                    double tsum = (double)acc[x][PX_S - 1][r];
#pragma unroll
                    for (int L = PX_S - 2; L >= 0; --L) tsum = fma(tsum, 1.0 / 256.0, (double)acc[x][L][r]);
                    const int gi = I0 + rbase + x * MB + li;
                    const double v = gi < m ? ldexp(tsum, se[rbase + x * MB + li] + ej - 12) : 0.0;
                    const unsigned long long dw = px_digit_word_checked(v, shb, (gj >= c_live0 && gj < n_live) ? counts : nullptr);
#pragma unroll
                    for (int s = 0; s < PX_S; ++s) sBy[wv][s][idx][li - 16 * h] = (unsigned char)px_digit_byte(dw, s);
                }
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
                __builtin_amdgcn_wave_barrier();
                const size_t kb = (size_t)(I0 + rbase + x * MB) / 16 + h;
                if (lane < 32 && (int)(kb * 16) < (m + 31) / 32 * 32) { // lane = column: its sixteen rows, one 16-byte group per plane
#pragma unroll
                    for (int s = 0; s < PX_S; ++s) {
                        const uint4 q = *(const uint4 *)&sBy[wv][s][lane][0];
                        *(uint4 *)(Bq + (size_t)s * b_stride + (kb * ldq + J0 + wc * MB + lane) * 16) = q;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
#undef BG_ISSUE
}
#endif

// B = inv(L) G on the int8 MFMA, straight into the digit planes of B (columns [c_lo, c_hi), multiples of 128 rounded outwards).
// W = inv(L)' in e->d.W (k-major, ldW), G in e->d.G (k-major fp64); e->d.Bexp holds the a-priori column scales of B.
void launch_b_gemm_planes(EkfEngine *e, int m, int c_lo, int c_hi)
{
    hipStream_t s = e->stream;
    const int m_k = round_up(m, 32), ld = e->ldP, ldw = e->ldW;
    const int n_pad = round_up(e->n, LD_ALIGN);
    const size_t g_stride = (size_t)e->bq_rows * ld, w_stride = (size_t)e->bq_rows * ldw;
    const int mw_pad = round_up(m, 128); // columns of W' the row tiles read
    // digit planes of W' (columns = rows of inv(L)) and of G, each with its columns' true scales
    (void)hipMemsetAsync(e->d.Wexp, 0, sizeof(int) * (size_t)ldw, s);
    k_col_exp<<<dim3((mw_pad + 255) / 256, PX_KSPLIT), 256, 0, s>>>(e->d.W, ldw, m, mw_pad, nullptr, e->d.Wexp);
    k_slice_B<<<dim3(mw_pad / 64, (m_k + 63) / 64), 256, 0, s>>>(e->d.W, ldw, m, m_k, nullptr, e->d.Wexp, e->d.Wq, ldw, w_stride, 0, mw_pad, e->d.Wz,
                                                                e->bz_stride);
    const int tj0 = c_lo / 128, tiles_j = (std::min(c_hi, n_pad) + 127) / 128 - tj0;
    const int g_lo = tj0 * 128, g_hi = (tj0 + tiles_j) * 128;
    (void)hipMemsetAsync(e->d.Gexp, 0, sizeof(int) * (size_t)ld, s);
    k_col_exp<<<dim3((n_pad + 255) / 256, PX_KSPLIT), 256, 0, s>>>((const double *)e->d.G, ld, m, n_pad, nullptr, e->d.Gexp);
    k_slice_B<<<dim3((g_hi - g_lo) / 64, (m_k + 63) / 64), 256, 0, s>>>((const double *)e->d.G, ld, m, m_k, nullptr, e->d.Gexp, e->d.Gq, ld, g_stride,
                                                                      g_lo, g_hi, e->d.Gz, e->bz_stride);
    const int tiles_i = (m + 127) / 128;
    const int n_units = tiles_i * tiles_j;
#if PX_S_VALUE == 5
    k_b_gemm_i8p<<<std::min(e->n_cus, n_units), 512, 0, s>>>(e->d.Wq, ldw, w_stride, e->d.Wexp, e->d.Gq, ld, g_stride, e->d.Gexp, e->d.Bq,
                                                           (size_t)e->bq_rows * ld, e->d.Bexp, m, tiles_j, tj0, n_units, e->d.counts, c_lo, std::min(e->n, c_hi),
                                                           e->px_dense ? nullptr : e->d.Wz, e->d.Gz, e->bz_stride);
#else // (builds with another digit count are accuracy experiments on the rows-of-B-in-the-sweep path only)
    (void)n_units;
    e->hook_rc = EKF_ERR_INVALID_ARG;
#endif
}

// ------------------------------------------------------------------------------------------------ launcher
void build_units(EkfEngine *e, int nt, int nrt, bool rect, int order); // kernels_pupdate.hip

#ifdef PX_BENCH // scripts/micro/pu_i8_bench.hip only: 1 = one workgroup per unit (k_p_update_i8), 4 = two workgroups per CU (k_p_update_i8d)
int g_px_variant = 0;
#else
constexpr int g_px_variant = 0;
#endif

// B (fp64, k-major, ld = e->ldP) sits in e->d.A; camera columns from e->d.Bc when use_bc
// exps_ready: e->d.Bexp already holds the column scales of columns 0 .. n - 1 (the engine collects them in the pass that forms
// dx = B'z); otherwise k_col_exp computes them here (scripts/micro/pu_i8_bench.hip)
// planes_ready: the digit planes of rows 0 .. m - 1 are in e->d.Bq already (the sweep formed B from them, chol_bplanes.h)
void launch_p_update_exact(EkfEngine *e, int m, bool use_bc, bool exps_ready, bool planes_ready)
{
    hipStream_t s = e->stream;
    const int n = e->n, ld = e->ldP;
    const int n_pad = round_up(n, LD_ALIGN);
    const int m_k = round_up(m, 32);
    const double *B = (const double *)e->d.A;
    const double *Bc = use_bc ? e->d.Bc : nullptr;
    const size_t plane_stride = (size_t)e->bq_rows * ld; // bytes per digit plane
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (e->timing) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        if (e->px_mid) { // the update recorded the end of its sweep: the bracket covers everything up to the downdate
            e2 = e->px_mid;
            e->px_mid = nullptr;
        } else {
            (void)hipEventCreate(&e2);
            (void)hipEventRecord(e2, s);
        }
    }
    if (!exps_ready) {
        (void)hipMemsetAsync(e->d.Bexp, 0, sizeof(int) * (size_t)ld, s);
        k_col_exp<<<dim3((n_pad + 255) / 256, PX_KSPLIT), 256, 0, s>>>(B, ld, m, n_pad, Bc, e->d.Bexp);
    }
    if (!planes_ready) k_slice_B<<<dim3(n_pad / 64, (m_k + 63) / 64), 256, 0, s>>>(B, ld, m, m_k, Bc, e->d.Bexp, e->d.Bq, ld, plane_stride, 0, n_pad, e->d.Bz, e->bz_stride);
    const int nt = (n + 127) / 128;
    const bool rect = e->shard_world > 1;
    const int owned = e->rm.r1 - e->rm.r0;
    const int nrt = 1 + (owned + 127) / 128; // camera tile + owned row tiles
    const int slots_saved = e->pu_slots;
    e->pu_slots = e->n_cus; // one 512-thread workgroup per CU is resident (160 accumulator registers per lane)
    build_units(e, nt, nrt, rect, 0);
    e->pu_slots = slots_saved;
    const int grid = e->pu_per_xcd * 8;
    const int4 *tm = (const int4 *)e->d.pu_tilemap;
    if (grid == 0 || !tm) { // build_units failed (e->hook_rc is set): nothing was launched, the timing events go back
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (e2) (void)hipEventDestroy(e2);
        return;
    }
    // two independent workgroups per CU on half tiles, units handed out dynamically (k_p_update_i8d): fp32-stored P
#ifdef PX_BENCH
    const bool dual = PX_S == 5 && e->f32 && (e->p_exact_sym || rect) && g_px_variant == 4;
#else
    constexpr bool dual = false;
#endif
    int grid_d = 0, per_d = 0;
    const int4 *tm_d = nullptr;
    if (dual) {
        if (!e->d.pu_ctr) {
            if (hipMalloc((void **)&e->d.pu_ctr, 16 * sizeof(unsigned)) != hipSuccess || hipMemset(e->d.pu_ctr, 0, 16 * sizeof(unsigned)) != hipSuccess) {
                e->d.pu_ctr = nullptr;
                e->err = "exact downdate: unit counters";
                e->hook_rc = EKF_ERR_HIP;
            }
        }
        const int keep = e->pu_slots;
        e->pu_slots = 1 << 20; // a unit list of half tiles only (build_units: one half-round)
        build_units(e, nt, nrt, rect, 0);
        e->pu_slots = keep;
        per_d = e->pu_per_xcd;
        tm_d = (const int4 *)e->d.pu_tilemap;
        grid_d = 2 * e->n_cus;
        if (!e->d.pu_ctr || !tm_d || per_d == 0) {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (e2) (void)hipEventDestroy(e2);
            return;
        }
    }
    if (e->timing) (void)hipEventRecord(e0, s);
    if (dual) {
#if PX_S_VALUE == 5 && defined(PX_BENCH)
        if (rect) k_p_update_i8d<true><<<grid_d, 256, 0, s>>>((float *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, per_d, tm_d, e->rm, e->d.counts, e->d.pu_ctr, e->pu_parity);
        else k_p_update_i8d<false><<<grid_d, 256, 0, s>>>((float *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, per_d, tm_d, e->rm, e->d.counts, e->d.pu_ctr, e->pu_parity);
        e->pu_parity ^= 1;
#endif
    } else
#if PX_S_VALUE == 5
    if (!e->f32) { // fp64-stored P (EKF_PRECISION_F64_EXACT): the persistent kernel with the fp64 epilogue; an arbitrary upload is
                   // symmetrised first (0.5 (P + P') - B'B = 0.5 ((P - B'B) + (P - B'B)'): B'B is symmetric)
        if (!e->p_exact_sym && !rect) k_symmetrize_P<<<dim3((n + 255) / 256, n), 256, 0, s>>>((double *)e->d.P, ld, n);
        if (rect) k_p_update_i8p<true, double><<<e->n_cus, 512, 0, s>>>((double *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, e->pu_per_xcd, tm, e->n_cus / 8, e->rm, e->d.counts, e->px_dense ? nullptr : e->d.Bz, e->bz_stride);
        else k_p_update_i8p<false, double><<<e->n_cus, 512, 0, s>>>((double *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, e->pu_per_xcd, tm, e->n_cus / 8, e->rm, e->d.counts, e->px_dense ? nullptr : e->d.Bz, e->bz_stride);
    } else
#else
    if (!e->f32) { // (other digit counts: fp32-stored P only)
        e->hook_rc = EKF_ERR_INVALID_ARG;
        return;
    }
#endif
    if (!e->p_exact_sym && !rect) k_p_update_i8<true><<<grid, 512, 0, s>>>((float *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, e->pu_per_xcd, tm, e->d.counts);
    else if (g_px_variant == 1 || PX_S != 5) k_p_update_i8<false><<<grid, 512, 0, s>>>((float *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, e->pu_per_xcd, tm, e->d.counts);
#if PX_S_VALUE == 5
#ifdef PX_BENCH
    else if (g_px_variant == 3 && rect) k_p_update_i8q<true><<<e->n_cus, 256, 0, s>>>((float *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, e->pu_per_xcd, tm, e->n_cus / 8, e->rm, e->d.counts);
    else if (g_px_variant == 3) k_p_update_i8q<false><<<e->n_cus, 256, 0, s>>>((float *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, e->pu_per_xcd, tm, e->n_cus / 8, e->rm, e->d.counts);
#endif
    else if (rect) k_p_update_i8p<true><<<e->n_cus, 512, 0, s>>>((float *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, e->pu_per_xcd, tm, e->n_cus / 8, e->rm, e->d.counts, e->px_dense ? nullptr : e->d.Bz, e->bz_stride);
    else k_p_update_i8p<false><<<e->n_cus, 512, 0, s>>>((float *)e->d.P, ld, n, e->d.Bq, ld, plane_stride, m_k, e->d.Bexp, e->pu_per_xcd, tm, e->n_cus / 8, e->rm, e->d.counts, e->px_dense ? nullptr : e->d.Bz, e->bz_stride);
#endif
    bool launched = true;
    {   // a launch that the runtime refuses (resources) would leave P silently un-downdated
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            launched = false;
            e->err = std::string("exact downdate launch: ") + hipGetErrorString(le);
            std::fprintf(stderr, "ekf: %s\n", e->err.c_str());
            e->hook_rc = EKF_ERR_HIP;
        }
    }
    if (e->timing) {
        (void)hipEventRecord(e1, s);
        e->pu_events.emplace_back(e0, e1);
        e->pu_work.push_back(rect ? (double)(owned + 13) * (double)n * (double)m * 2.0 : (double)n * (double)n * (double)m);
        e->pu_m.push_back(m);
        e->px_events.emplace_back(e2, e0); // sweep end (or this call's start) -> downdate start
    }
    if (launched) e->p_exact_sym = true; // (a refused launch leaves P as uploaded: the next downdate still symmetrises it)
}

} // namespace ekf
