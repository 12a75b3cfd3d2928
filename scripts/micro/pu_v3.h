// Experimental downdate kernels for scripts/micro/pu_bench.hip (included after kernels_pupdate.hip).
//
// k_pu3: the same tiles and units as k_p_update (fp32), but NO workgroup barrier in the k-loop: every wavefront stages its
// OWN operand slabs (64 rows of the row operand, 64 columns of the column operand, BK deep) by LDS-DMA into a private ring of
// NST stages and waits for them with counted vmcnt.  The price is that a slab is fetched by the two wavefronts that share it
// (L1 / L2 traffic x2); what it buys is that the four wavefronts of a tile never wait for each other.
#pragma once

namespace ekf {

typedef const __attribute__((address_space(1))) void *pu3_gptr_t;
typedef __attribute__((address_space(3))) void *pu3_lptr_t;

template <int N>
__device__ __forceinline__ void pu3_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ABL (timing ablations, wrong results): 1 no epilogue, 2 no operand loads after the prologue, 4 no LDS reads in the loop
template <int BK, int NST, int MINW, int ABL = 0>
__global__ void __launch_bounds__(256, MINW)
k_pu3(float *P, int ldp, int n, const float *B, int ldb, int m_k, int per_xcd, const int4 *units)
{
    using M = Mma<float>;
    constexpr int MB = 32, KI = 2;
    constexpr int SLAB = BK * 64;          // floats of one operand slab of one wavefront
    constexpr int STAGE = 2 * SLAB;        // row operand + column operand
    constexpr int NPS = 2 * (BK / 4);      // LDS-DMA instructions per stage (each moves 4 k-rows x 64 floats = 1 KB)
    static_assert(BK % 4 == 0 && NST >= 2 && NST <= 4, "geometry");
    static_assert(NST * STAGE >= MB * (MB + 1), "the epilogue's transpose scratch lives in the wavefront's ring");
    __shared__ __attribute__((aligned(16))) float smem[4 * NST * STAGE];

    const int4 unit = units[(size_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)];
    if (unit.x < 0) return;
    const int ti = unit.x, tj = unit.y;
    const bool full = unit.z < 0;
    const bool diag = ti == tj;
    const int I0 = ti * 128, J0 = tj * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 1, wc = wv & 1;
    const int klane = lane >> 5, idx = lane & 31;
    // rows of this wavefront: full unit 64 (two MFMA blocks), half unit 32 of the 64-row slab both wavefront rows load
    const int arow0 = full ? wr * 64 : unit.z * 64;   // first column of B (= row of the tile) of the staged row slab
    const int rofs = full ? 0 : wr * 32;              // this wavefront's rows inside that slab
    const int rbase = arow0 + rofs;                   // first tile row of the wavefront
    float *ring = smem + wv * NST * STAGE;

    typename M::acc_t c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) c00[r] = c01[r] = c10[r] = c11[r] = 0.f;

    const int nk = m_k / BK;
    // lane l of a DMA instruction: k-row l / 16, floats 4 (l % 16) .. +3 of the 64-float slab row
    const float *gA = B + (size_t)(lane >> 4) * ldb + I0 + arow0 + 4 * (lane & 15);
    const float *gB = B + (size_t)(lane >> 4) * ldb + J0 + wc * 64 + 4 * (lane & 15);
    const size_t step4 = (size_t)4 * ldb;
    auto issue = [&](int kt) {
        float *dst = ring + (kt % NST) * STAGE;
        const float *a = gA + (size_t)kt * BK * ldb, *b = gB + (size_t)kt * BK * ldb;
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) {
            __builtin_amdgcn_global_load_lds((pu3_gptr_t)(a + q * step4), (pu3_lptr_t)(dst + q * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((pu3_gptr_t)(b + q * step4), (pu3_lptr_t)(dst + SLAB + q * 256), 16, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s);
    for (int kt = 0; kt < nk; ++kt) {
        if (!(ABL & 2) && kt + NST - 1 < nk) issue(kt + NST - 1); // its buffer was read in iteration kt - 1 (reads waited for)
        const int ahead = min(NST - 1, nk - 1 - kt);             // stages younger than kt that may still be in flight
        if (ABL & 2) { if (kt == 0) pu3_wait<0>(); }
        else if (ahead >= 3) pu3_wait<3 * NPS>();
        else if (ahead == 2) pu3_wait<2 * NPS>();
        else if (ahead == 1) pu3_wait<NPS>();
        else pu3_wait<0>();
        const float *sA = ring + (kt % NST) * STAGE, *sB = sA + SLAB;
        if (ABL & 4) {
            float a0 = sA[klane * 64 + rofs + idx], b0 = sB[klane * 64 + idx];
#pragma unroll
            for (int kk = 0; kk < BK; kk += KI) {
                c00 = M::mma(a0, b0, c00);
                c01 = M::mma(a0, b0, c01);
                if (full) {
                    c10 = M::mma(a0, b0, c10);
                    c11 = M::mma(a0, b0, c11);
                }
            }
            continue;
        }
        float a0 = sA[klane * 64 + rofs + idx], a1 = full ? sA[klane * 64 + rofs + 32 + idx] : 0.f;
        float b0 = sB[klane * 64 + idx], b1 = sB[klane * 64 + 32 + idx];
#pragma unroll
        for (int kk = 0; kk < BK; kk += KI) {
            float na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
            if (kk + KI < BK) {
                na0 = sA[(kk + KI + klane) * 64 + rofs + idx];
                if (full) na1 = sA[(kk + KI + klane) * 64 + rofs + 32 + idx];
                nb0 = sB[(kk + KI + klane) * 64 + idx];
                nb1 = sB[(kk + KI + klane) * 64 + 32 + idx];
            }
            __builtin_amdgcn_sched_barrier(0);
            c00 = M::mma(a0, b0, c00);
            c01 = M::mma(a0, b1, c01);
            if (full) {
                c10 = M::mma(a1, b0, c10);
                c11 = M::mma(a1, b1, c11);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    }
    if (ABL & 1) {
        if (c00[0] + c01[1] + c10[2] + c11[3] == 12345.f) P[0] = 0.f;
        return;
    }
    // epilogue: as k_p_update (tile written in place, mirror image through a per-wavefront LDS transpose)
    float *sT = ring;
    typename M::acc_t pv[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            if (x == 1 && !full) continue;
            if (y == 0) {
#pragma unroll
                for (int yy = 0; yy < 2; ++yy) {
                    const int pbi = I0 + rbase + x * MB, pbj = J0 + wc * 64 + yy * MB;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gi = pbi + M::row(r, lane), gj = pbj + M::col(lane);
                        pv[yy][r] = (gi < n && gj < n) ? P[(size_t)gi * ldp + gj] : 0.f;
                    }
                }
            }
            const int bi = I0 + rbase + x * MB, bj = J0 + wc * 64 + y * MB;
            const typename M::acc_t &cc = x == 0 ? (y == 0 ? c00 : c01) : (y == 0 ? c10 : c11);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int li = M::row(r, lane), lj = M::col(lane);
                const int gi = bi + li, gj = bj + lj;
                float v = 0.f;
                if (gi < n && gj < n) {
                    v = pv[y][r] - cc[r];
                    P[(size_t)gi * ldp + gj] = v;
                }
                if (!diag) sT[li * (MB + 1) + lj] = v;
            }
            if (!diag) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < MB / KI; ++it) {
                    const int c = it * KI + klane;
                    const int gi = bi + idx, gj = bj + c;
                    const float v = sT[idx * (MB + 1) + c];
                    if (gi < n && gj < n) P[(size_t)gj * ldp + gi] = v;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
}

template <int BK, int NST, int MINW, int ABL = 0>
static void launch_pu3(EkfEngine *e, int m_pad, int m)
{
    const int n = e->n, nt = (n + 127) / 128;
    const int m_k = round_up(m, BK);
    build_units(e, nt, 0, false, m_pad >= 512 ? 1 : 0);
    k_pu3<BK, NST, MINW, ABL><<<e->pu_per_xcd * 8, 256, 0, e->stream>>>((float *)e->d.P, e->ldP, n, (const float *)e->d.A, e->ldP, m_k,
                                                                        e->pu_per_xcd, (const int4 *)e->d.pu_tilemap);
}

} // namespace ekf

namespace ekf {
// k_p_update (fp32, register-staged slabs, workgroup barrier per slab) with timing ablations (wrong results):
// ABL 1 no epilogue, 2 no global loads in the loop, 4 no LDS reads (operands constant), 8 no barrier / LDS stores
template <int ABL>
__global__ void __launch_bounds__(256, 3)
k_pu_abl(float *P, int ldp, int n, const float *B, int ldb, int m_pad, int per_xcd, const int4 *units)
{
    using T = float;
    using M = Mma<T>;
    constexpr int MB = 32, TM = 128, KI = 2, VEC = 4;
    // 48 KB instead of the 32 KB used: pins every ablation at three workgroups per CU (what 150 VGPRs give the real kernel)
    __shared__ __attribute__((aligned(16))) T smem[6 * PU_BK * TM];
    T(*sI)[PU_BK][TM] = reinterpret_cast<T(*)[PU_BK][TM]>(smem);
    T(*sJ)[PU_BK][TM] = reinterpret_cast<T(*)[PU_BK][TM]>(smem + 2 * PU_BK * TM);
    const int4 unit = units[(size_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)];
    if (unit.x < 0) return;
    const int ti = unit.x, tj = unit.y;
    const bool full = unit.z < 0;
    const bool diag = ti == tj;
    const int I0 = ti * TM, J0 = tj * TM;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int rbase = full ? wr * 2 * MB : unit.z * 2 * MB + wr * MB;
    const int klane = lane / MB, idx = lane % MB;
    typename M::acc_t c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) c00[r] = c01[r] = c10[r] = c11[r] = 0.f;
    using V = float4;
    const int nk = m_pad / PU_BK;
    const size_t slab = (size_t)PU_BK * ldb;
#define PUA_PIECE(q) const int lk##q = ((tid + q * 256) * VEC) / TM, lc##q = ((tid + q * 256) * VEC) % TM; \
                    const T *gI##q = B + (size_t)lk##q * ldb + I0 + lc##q; const T *gJ##q = B + (size_t)lk##q * ldb + J0 + lc##q; \
                    V rI##q = *(const V *)gI##q, rJ##q = *(const V *)gJ##q;
    PUA_PIECE(0)
    PUA_PIECE(1)
#undef PUA_PIECE
#define PUA_STORE(q, b) *(V *)(&sI[b][lk##q][lc##q]) = rI##q; *(V *)(&sJ[b][lk##q][lc##q]) = rJ##q;
#define PUA_LOAD(q, off) rI##q = *(const V *)(gI##q + off); rJ##q = *(const V *)(gJ##q + off);
    PUA_STORE(0, 0)
    PUA_STORE(1, 0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        if (more && !(ABL & 2)) {
            const size_t off = (size_t)(kt + 1) * slab;
            PUA_LOAD(0, off)
            PUA_LOAD(1, off)
        }
        if (ABL & 16) { // operands by 16-byte LDS reads (two per operand block and slab instead of eight 4-byte ones; garbage values)
            const float4 *q = reinterpret_cast<const float4 *>(&sI[buf][0][0]) + lane * 5;
            const float4 *qj = reinterpret_cast<const float4 *>(&sJ[buf][0][0]) + lane * 5;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 a0 = q[h * 320], a1 = q[h * 320 + 1], b0 = qj[h * 320], b1 = qj[h * 320 + 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    c00 = M::mma(a0[e], b0[e], c00);
                    c01 = M::mma(a0[e], b1[e], c01);
                    if (full) {
                        c10 = M::mma(a1[e], b0[e], c10);
                        c11 = M::mma(a1[e], b1[e], c11);
                    }
                }
            }
        } else if (ABL & 64) { // 8-byte operand pairs (the shipped fp32 kernel's reads), interleaved with the MFMAs or (ABL & 128) all up-front
            const float *pa = &sI[buf][klane][(rbase + 2 * idx) & 127], *pb = &sJ[buf][klane][wc * 64 + 2 * idx];
            if (ABL & 128) {
                float2 a[8], b[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { a[q] = *(const float2 *)(pa + 2 * q * TM); b[q] = *(const float2 *)(pb + 2 * q * TM); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    c00 = M::mma(a[q].x, b[q].x, c00); c01 = M::mma(a[q].x, b[q].y, c01);
                    if (full) { c10 = M::mma(a[q].y, b[q].x, c10); c11 = M::mma(a[q].y, b[q].y, c11); }
                }
            } else {
                float2 a = *(const float2 *)pa, b = *(const float2 *)pb;
#pragma unroll
                for (int kk = 0; kk < PU_BK; kk += 2) {
                    float2 na = a, nb = b;
                    if (kk + 2 < PU_BK) { na = *(const float2 *)(pa + (kk + 2) * TM); nb = *(const float2 *)(pb + (kk + 2) * TM); }
                    __builtin_amdgcn_sched_barrier(0);
                    c00 = M::mma(a.x, b.x, c00); c01 = M::mma(a.x, b.y, c01);
                    if (full) { c10 = M::mma(a.y, b.x, c10); c11 = M::mma(a.y, b.y, c11); }
                    __builtin_amdgcn_sched_barrier(0);
                    a = na; b = nb;
                }
            }
        } else if (ABL & 256) { // 16-byte reads, two per EIGHT MFMAs, interleaved (garbage values): the K2 layout's instruction stream
            const float4 *q = reinterpret_cast<const float4 *>(&sI[buf][0][0]) + klane * 64 + idx + wr * 128;
            const float4 *qj = reinterpret_cast<const float4 *>(&sJ[buf][0][0]) + klane * 64 + idx + wc * 128;
            float4 a = q[0], b = qj[0];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                float4 na = a, nb = b;
                if (h + 1 < 4) { na = q[(h + 1) * 256 % 448]; nb = qj[(h + 1) * 256 % 448]; }
                __builtin_amdgcn_sched_barrier(0);
                c00 = M::mma(a.x, b.x, c00); c01 = M::mma(a.x, b.z, c01);
                if (full) { c10 = M::mma(a.z, b.x, c10); c11 = M::mma(a.z, b.z, c11); }
                c00 = M::mma(a.y, b.y, c00); c01 = M::mma(a.y, b.w, c01);
                if (full) { c10 = M::mma(a.w, b.y, c10); c11 = M::mma(a.w, b.w, c11); }
                __builtin_amdgcn_sched_barrier(0);
                a = na; b = nb;
            }
        } else if (ABL & 32) { // 16x16x4 MFMA shape: 16 accumulators of 4 registers, same LDS read count (garbage values)
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 *cc = reinterpret_cast<f4 *>(&c00);
            f4 *cd = reinterpret_cast<f4 *>(&c01);
            f4 *ce = reinterpret_cast<f4 *>(&c10);
            f4 *cf = reinterpret_cast<f4 *>(&c11);
            const int lq = lane >> 4, lm = lane & 15;
#pragma unroll
            for (int kk = 0; kk < PU_BK; kk += 4) {
                T a[4], b[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a[r] = sI[buf][kk + lq][(wr * 64 + 16 * r + lm + 16 * (lq & 1)) & 127];
                    b[r] = sJ[buf][kk + lq][(wc * 64 + 16 * r + lm + 16 * (lq & 1)) & 127];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[r], cc[r], 0, 0, 0);
                    cd[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[r], cd[r], 0, 0, 0);
                    ce[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[r], ce[r], 0, 0, 0);
                    cf[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[r], cf[r], 0, 0, 0);
                }
            }
        } else if (ABL & 4) {
            T a0 = sI[buf][klane][rbase + idx], b0 = sJ[buf][klane][wc * 2 * MB + idx];
#pragma unroll
            for (int kk = 0; kk < PU_BK; kk += KI) {
                c00 = M::mma(a0, b0, c00);
                c01 = M::mma(a0, b0, c01);
                if (full) {
                    c10 = M::mma(a0, b0, c10);
                    c11 = M::mma(a0, b0, c11);
                }
            }
        } else {
            if (full) pu_slab<T, true, TM>(sI[buf], sJ[buf], klane, rbase + idx, wc * 2 * MB + idx, c00, c01, c10, c11);
            else pu_slab<T, false, TM>(sI[buf], sJ[buf], klane, rbase + idx, wc * 2 * MB + idx, c00, c01, c10, c11);
        }
        if (!(ABL & 8)) {
            if (more) {
                PUA_STORE(0, buf ^ 1)
                PUA_STORE(1, buf ^ 1)
            }
            __syncthreads();
        }
    }
#undef PUA_STORE
#undef PUA_LOAD
    if (ABL & 1) {
        if (c00[0] + c01[1] + c10[2] + c11[3] == 12345.f) P[0] = 0.f;
        return;
    }
    T *sT = smem + wv * MB * (MB + 1);
    typename M::acc_t pv[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            if (x == 1 && !full) continue;
            if (y == 0) {
#pragma unroll
                for (int yy = 0; yy < 2; ++yy) {
                    const int pbi = I0 + rbase + x * MB, pbj = J0 + wc * 2 * MB + yy * MB;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gi = pbi + M::row(r, lane), gj = pbj + M::col(lane);
                        pv[yy][r] = (gi < n && gj < n) ? P[(size_t)gi * ldp + gj] : (T)0;
                    }
                }
            }
            const int bi = I0 + rbase + x * MB, bj = J0 + wc * 2 * MB + y * MB;
            const typename M::acc_t &cc = x == 0 ? (y == 0 ? c00 : c01) : (y == 0 ? c10 : c11);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int li = M::row(r, lane), lj = M::col(lane);
                const int gi = bi + li, gj = bj + lj;
                T v = (T)0;
                if (gi < n && gj < n) {
                    v = pv[y][r] - cc[r];
                    P[(size_t)gi * ldp + gj] = v;
                }
                if (!diag) sT[li * (MB + 1) + lj] = v;
            }
            if (!diag) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < MB / KI; ++it) {
                    const int c = it * KI + klane;
                    const int gi = bi + idx, gj = bj + c;
                    const T v = sT[idx * (MB + 1) + c];
                    if (gi < n && gj < n) P[(size_t)gj * ldp + gi] = v;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
}

template <int ABL>
static void launch_pu_abl(EkfEngine *e, int m_pad, int m)
{
    const int n = e->n, nt = (n + 127) / 128;
    build_units(e, nt, 0, false, m_pad >= 512 ? 1 : 0);
    k_pu_abl<ABL><<<e->pu_per_xcd * 8, 256, 0, e->stream>>>((float *)e->d.P, e->ldP, n, (const float *)e->d.A, e->ldP, round_up(m, 16),
                                                            e->pu_per_xcd, (const int4 *)e->d.pu_tilemap);
}
} // namespace ekf
