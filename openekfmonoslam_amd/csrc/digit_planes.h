// digit_planes.h -- cutting a scaled fp64 value into PX_S balanced base-256 digits (EKF_PRECISION_F32_EXACT; kernels_pexact.hip,
// chol_bplanes.h).  X = rint(v 2^sh), |X| <= 2^(8 PX_S - 2), is wanted as X = sum_s d_s 256^(PX_S - 1 - s) with every d_s in
// [-128, 127] (int8 operands of v_mfma_i32_32x32x32_i8).  No carry chain is needed: with K = 0x80 repeated PX_S times,
// the BYTES of X + K are b_s = d_s + 128 for exactly that digit set (sum_s (b_s - 128) 256^(..) = X + K - K), and b - 128
// reinterpreted as int8 is b ^ 0x80.  So the digits are the bytes of (X + K) ^ K; the top byte stays in [0x40, 0xC0].
#pragma once
#include <hip/hip_runtime.h>

namespace ekf {

__device__ __forceinline__ unsigned long long px_digit_word(double v, int sh)
{
    constexpr unsigned long long K = PX_S == 5 ? 0x8080808080ull : (PX_S == 6 ? 0x808080808080ull : 0x80808080ull);
    const long long X = __double2ll_rn(ldexp(v, sh));
    return ((unsigned long long)X + K) ^ K; // byte (PX_S - 1 - s) = digit s as int8
}

// ... for the rows of B under an A-PRIORI column scale (|B_kj| <= sqrt(P_jj), chol_bplanes.h): the bound holds for a positive
// semi-definite P; a covariance uploaded with |P_ij| > sqrt(P_ii P_jj) (or a NaN) breaks it, and digits that wrap would downdate P with
// garbage in silence.  An integer that does not fit the PX_S digits raises the sticky code EKF_ERR_NON_FINITE instead -- the kernels
// behind the sweep then leave the filter untouched (engine.h: filter_frozen) and the update reports it.
__device__ __forceinline__ unsigned long long px_digit_word_checked(double v, int sh, int *counts)
{
    constexpr unsigned long long K = PX_S == 5 ? 0x8080808080ull : (PX_S == 6 ? 0x808080808080ull : 0x80808080ull);
    const long long X = __double2ll_rn(ldexp(v, sh));
    const long long lim = 1ll << (8 * PX_S - 1);
    if (counts && !(X > -lim && X < lim)) atomicMax(&counts[CNT_ERR], (int)EKF_ERR_NON_FINITE);
    return ((unsigned long long)X + K) ^ K;
}

// digit s of the word
__device__ __forceinline__ unsigned px_digit_byte(unsigned long long w, int s) { return (unsigned)(w >> (8 * (PX_S - 1 - s))) & 255u; }

} // namespace ekf
