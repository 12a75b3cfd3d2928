// mma_tile.h -- MFMA tile helpers shared by the GEMM-shaped kernels (covariance downdate, B = inv(L) A, the upper
// levels of the triangular inverse): C/D layouts of the two MFMA shapes used and the k-slab inner loop.
#pragma once
#include <hip/hip_runtime.h>

namespace ekf {

template <typename T>
struct Mma;

template <>
struct Mma<float> {
    static constexpr int MB = 32, NACC = 16;
    typedef float acc_t __attribute__((ext_vector_type(16)));
    typedef float4 vec_t; // 16-byte global/LDS vector
    static constexpr int VEC = 4;
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // C/D layout of the 32x32 f32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    __device__ static __forceinline__ int row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
    __device__ static __forceinline__ int col(int lane) { return lane & 31; }
};

template <>
struct Mma<double> {
    static constexpr int MB = 16, NACC = 4;
    typedef double acc_t __attribute__((ext_vector_type(4)));
    typedef double2 vec_t;
    static constexpr int VEC = 2;
    __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // C/D layout of the 16x16 f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 reg
    __device__ static __forceinline__ int row(int reg, int lane) { return (lane >> 4) + 4 * reg; }
    __device__ static __forceinline__ int col(int lane) { return lane & 15; }
};

#ifndef PU_BK_VALUE
#define PU_BK_VALUE 16
#endif
constexpr int PU_BK = PU_BK_VALUE;

// MFMAs of one k-slab held in LDS.  Operands of k-step kk+KI are fetched BEFORE the MFMAs of k-step kk are issued:
// a wavefront issues in order, so without this its matrix pipe idles for one LDS round trip per k-step.
template <typename T, bool FULL, int TMc, int BKc = PU_BK>
__device__ __forceinline__ void pu_slab(const T (*sIb)[TMc], const T (*sJb)[TMc], int klane, int ra, int cb,
                                        typename Mma<T>::acc_t &c00, typename Mma<T>::acc_t &c01,
                                        typename Mma<T>::acc_t &c10, typename Mma<T>::acc_t &c11)
{
    using M = Mma<T>;
    constexpr int MB = M::MB, KI = 64 / MB;
    T a0 = sIb[klane][ra], a1 = FULL ? sIb[klane][ra + MB] : (T)0;
    T b0 = sJb[klane][cb], b1 = sJb[klane][cb + MB];
#pragma unroll
    for (int kk = 0; kk < BKc; kk += KI) {
        T na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
        if (kk + KI < BKc) {
            na0 = sIb[kk + KI + klane][ra];
            if (FULL) na1 = sIb[kk + KI + klane][ra + MB];
            nb0 = sJb[kk + KI + klane][cb];
            nb1 = sJb[kk + KI + klane][cb + MB];
        }
        __builtin_amdgcn_sched_barrier(0); // keep the fetches above the MFMAs they hide behind
        c00 = M::mma(a0, b0, c00);
        c01 = M::mma(a0, b1, c01);
        if (FULL) {
            c10 = M::mma(a1, b0, c10);
            c11 = M::mma(a1, b1, c11);
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
}

} // namespace ekf
