// kernels_gemm.hip -- C = alpha X' Y on the MFMA pipe, both operands k-major ("row k = all columns"), optional
// triangular k-limits and a transposed second output.  Used for
//   * B = inv(L) (H P): X = W = inv(L)' (upper triangular), Y = the gathered rows of H P        [replaces the
//     blocked forward substitution: with the factor inverted once, K = P H' inv(S) of EKF/Update.cpp:105 becomes
//     one GEMM with no sequential dependency between row blocks]
//   * the upper levels (block size >= 256) of the recursive inverse of L:  X21 = -X22 L21 X11
// Same tiling as the covariance downdate (kernels_pupdate.hip): 256 threads = 2 x 2 wavefronts x (2 x 2 MFMA
// blocks), tile TM = 4 MB (fp32 128, fp64 64), k-slab 16 (fp32) / 32 (fp64), register-staged double buffering through LDS.
#include "engine.h"
#include "mma_tile.h"

namespace ekf {

// BK: k-slab depth.  The fp64 instances are small, latency-bound GEMMs (a 64x64 tile's MFMAs of one 16-deep slab take
// 0.2 us, a global load round trip over 1 us), so they run with 32-deep slabs: half as many round trips.
template <typename T, int BK>
__global__ void __launch_bounds__(256, BK * sizeof(T) >= 128 ? 2 : 3) k_xty(XtyArgs a)
{
    using M = Mma<T>;
    constexpr int MB = M::MB, TM = 4 * MB, VEC = M::VEC;
    constexpr int LOADS = BK * TM / (256 * VEC);
    static_assert(LOADS == 2 || LOADS == 4, "two or four 16-byte pieces per thread and slab");
    __shared__ __attribute__((aligned(16))) T smem[4 * BK * TM];
    T(*sI)[BK][TM] = reinterpret_cast<T(*)[BK][TM]>(smem);
    T(*sJ)[BK][TM] = reinterpret_cast<T(*)[BK][TM]>(smem + 2 * BK * TM);

    // Work units.  Row tiles are walked from the bottom (tri == 2: the bottom rows have the longest k-range); the
    // a.n_split bottom row tiles are cut into two HALF units (64 of the 128 / 32 of the 64 tile rows each) so that the
    // longest unit, which bounds the launch when every unit is resident at once, is half as long.
    const int per = (a.tiles_i + a.n_split) * a.tiles_j;
    const int b = blockIdx.x / per;
    int t = blockIdx.x % per, ti, tj, half = -1;
    if (t < 2 * a.n_split * a.tiles_j) {
        ti = a.tiles_i - 1 - t / (2 * a.tiles_j);
        tj = (t % (2 * a.tiles_j)) >> 1;
        half = t & 1;
    } else {
        t -= 2 * a.n_split * a.tiles_j;
        ti = a.tri == 2 ? a.tiles_i - 1 - a.n_split - t / a.tiles_j : a.n_split + t / a.tiles_j;
        tj = t % a.tiles_j;
    }
    const int I0 = ti * TM, J0 = (tj + a.tj0) * TM;
    // rows of this batch element that exist (the last pair of a level may be cut by m_pad)
    const int Mb = min(a.M, a.m_lim - (a.row0_first + b * a.row0_stride));
    if (I0 >= Mb) return;
    const T *X = (const T *)a.X + (size_t)b * a.xb;
    const T *Y = (const T *)a.Y + (size_t)b * a.yb;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const bool full = half < 0;
    const int rbase = full ? wr * 2 * MB : half * 2 * MB + wr * MB; // as in the downdate's half units
    const int klane = lane / MB, idx = lane % MB;
    typename M::acc_t c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) c00[r] = c01[r] = c10[r] = c11[r] = (T)0;

    // k-range: tri 1: Y[k][j] = 0 for k < j ; tri 2: X[k][i] = 0 for k > i
    const int k_lo = a.tri == 1 ? J0 : 0;
    const int k_hi = a.tri == 2 ? min(a.K, I0 + TM) : a.K;
    const int nk = (k_hi - k_lo) / BK; // K, the tile edges and BK are multiples of 32 (16 for fp32): whole slabs
    using V = typename M::vec_t;
    const size_t xslab = (size_t)BK * a.ldx, yslab = (size_t)BK * a.ldy;
#define XT_PIECE(q)                                                                                              \
    const int lk##q = ((tid + q * 256) * VEC) / TM, lc##q = ((tid + q * 256) * VEC) % TM;                         \
    const T *gI##q = X + (size_t)(k_lo + lk##q) * a.ldx + I0 + lc##q;                                            \
    const T *gJ##q = Y + (size_t)(k_lo + lk##q) * a.ldy + J0 + lc##q;                                            \
    V rI##q = V(), rJ##q = V();                                                                                  \
    if (q < LOADS) { rI##q = *(const V *)gI##q; rJ##q = *(const V *)gJ##q; }
    XT_PIECE(0)
    XT_PIECE(1)
    XT_PIECE(2)
    XT_PIECE(3)
#undef XT_PIECE
#define XT_STORE(q, bf) *(V *)(&sI[bf][lk##q][lc##q]) = rI##q; *(V *)(&sJ[bf][lk##q][lc##q]) = rJ##q;
#define XT_LOAD(q, kt) rI##q = *(const V *)(gI##q + (kt) * xslab); rJ##q = *(const V *)(gJ##q + (kt) * yslab);
    XT_STORE(0, 0)
    XT_STORE(1, 0)
    if (LOADS == 4) {
        XT_STORE(2, 0)
        XT_STORE(3, 0)
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            XT_LOAD(0, (size_t)(kt + 1))
            XT_LOAD(1, (size_t)(kt + 1))
            if (LOADS == 4) {
                XT_LOAD(2, (size_t)(kt + 1))
                XT_LOAD(3, (size_t)(kt + 1))
            }
        }
        if (full) pu_slab<T, true, TM, BK>(sI[buf], sJ[buf], klane, rbase + idx, wc * 2 * MB + idx, c00, c01, c10, c11);
        else pu_slab<T, false, TM, BK>(sI[buf], sJ[buf], klane, rbase + idx, wc * 2 * MB + idx, c00, c01, c10, c11);
        if (more) {
            XT_STORE(0, buf ^ 1)
            XT_STORE(1, buf ^ 1)
            if (LOADS == 4) {
                XT_STORE(2, buf ^ 1)
                XT_STORE(3, buf ^ 1)
            }
        }
        __syncthreads();
    }
#undef XT_STORE
#undef XT_LOAD

    T *C = a.C ? (T *)a.C + (size_t)b * a.cb : nullptr;
    double *Ct = a.Ct ? (double *)a.Ct + (size_t)b * a.ctb : nullptr;
    float *Ctf = a.Ctf ? a.Ctf + (size_t)b * a.ctb : nullptr;
    const T alpha = (T)a.alpha;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            if (x == 1 && !full) continue;
            const int bi = I0 + rbase + x * MB, bj = J0 + wc * 2 * MB + y * MB;
            const typename M::acc_t &cc = x == 0 ? (y == 0 ? c00 : c01) : (y == 0 ? c10 : c11);
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) {
                const int gi = bi + M::row(r, lane), gj = bj + M::col(lane);
                if (gi < Mb && gj < a.N) {
                    T v = alpha * cc[r];
                    if (C) C[(size_t)gi * a.ldc + gj] = v;
                    if (Ct) Ct[(size_t)gj * a.ldct + gi] = (double)v;
                    if (Ctf) Ctf[(size_t)gj * a.ldct + gi] = (float)v;
                }
            }
        }
}

void launch_xty(EkfEngine *e, const XtyArgs &a, int batch, bool f32, hipStream_t stream)
{
    (void)e;
    const int grid = batch * (a.tiles_i + a.n_split) * a.tiles_j;
    if (grid <= 0) return;
    if (f32) k_xty<float, 16><<<grid, 256, 0, stream>>>(a);
    else k_xty<double, 32><<<grid, 256, 0, stream>>>(a);
}

} // namespace ekf
