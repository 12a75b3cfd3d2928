// chol_persist.h -- the blocked Cholesky sweep of S = H P H' + R (replaces S.inv(), EKF/Update.cpp:108) and the rows of
// B = inv(L) (H P) as ONE persistent launch.  Included by kernels_update.hip inside namespace ekf, after chol_bplanes.h.
//
// The launch-per-panel sweep (k_chol_step) pays, per 32-row panel, a kernel boundary (~1.7 us), the cold loads of every role
// (~2 us) and the longest role of the launch; at N = 1000 that is 9.8 us x 42 dependent launches = 43 % of a frame.  Here the
// critical chain lives in ONE resident workgroup and everything else follows it through flags:
//
//   chain workgroup (ticket 0)   for every panel k:  factorise A_kk and invert its factor (block_chol_inv32_w2: two free-running
//                                wavefronts; the other two meanwhile fetch the next two tiles), publish inv(L_kk), apply panel k to
//                                tile (k+1, k+1) itself:  L(k+1,k) = S(k+1,k) inv(L_kk)',  A(k+1,k+1) = S(k+1,k+1) - L L'
//   tile workers                 fixed owners of the tiles (i, j), i >= j, of the trailing matrix -- plus one extra block ROW that
//                                carries nu' (the right-hand side: its "panel blocks" are z' = (inv(L) nu)', so z needs no role of
//                                its own).  As soon as inv(L_kk) and the panel blocks S(i,k), S(j,k) are published an owner applies
//                                panel k to its tile (recomputing L(i,k), L(j,k): two more 32^3 products, one hand-off less on the
//                                path to the chain).  The owner of a panel block (i, k) stores the official L(i,k): digit planes
//                                (exact configuration) or the mirrored fp64 copy (fp64 configuration), and raises lrdy(i, k).
//   B workers                    fixed owners of 32-column blocks of B:  B_k = inv(L_kk) (G_k - sum_{j<k} L(k,j) B_j), the sum
//                                from int8 digit planes (chol_bplanes.h) or on the fp64 MFMA; the blocks j < k - 1 are summed
//                                while the chain still factorises panel k.
//
// Hand-offs follow the guide's recipe R1: payloads stored write-through (sc1), the storing wavefronts drain (s_waitcnt vmcnt(0)),
// one relaxed agent-scope flag; consumers poll the flag from one lane (s_sleep between polls) and read the payload with sc1 loads.
// Flags are monotone and carry an epoch (no clearing between sweeps).  Every wait is on work of an earlier level (or, for the tile
// workers, no wait at all: they pick the most urgent RUNNABLE task of their tiles), so the launch needs all its workgroups resident
// (the host sizes the grid by the occupancy query) but no particular placement; the block order only tries to give the chain
// workgroup a CU of its own (an empty spacer block where in-order dispatch would double up).  EVERY spin is bounded by wall time: a wait that exceeds PS_TIMEOUT_TICKS sets the
// sticky error EKF_ERR_TIMEOUT, wakes every other waiter and ends the launch -- never a hung stream.
#pragma once

struct SweepCtl {
    unsigned reserved;  // (roles come from blockIdx: no tickets)
    unsigned inv_ready; // epoch base + panels whose inv(L_kk) is published
    unsigned err;       // != 0: the launch is failing, everybody leaves
    unsigned pad[29];
    unsigned flags[1];  // done[PS_NBC][PS_NBC], then lrdy[PS_NBC][PS_NBC]
};
constexpr int PS_NBC = B_SWEEP_MAX / NB + 2;        // block rows of the flag tables: 64 panels + the nu row + 1
constexpr unsigned PS_EPOCH_STEP = 128;             // flag values of one sweep stay below this
constexpr long long PS_TIMEOUT_TICKS = 500000ll;    // 5 ms of the 100 MHz constant clock (a sweep of 2048 rows takes ~1 ms; round 5: 30 ms)
constexpr int PS_BATCH = 4;                         // panels a tile worker applies to one tile per task in the L form
constexpr int PS_CACHE = 4;                         // tiles a tile worker keeps in LDS between their panels (the rest live in S)
inline size_t sweep_ctl_bytes() { return sizeof(SweepCtl) + sizeof(unsigned) * 2 * PS_NBC * PS_NBC; }

struct PsArgs {
    double *S;
    double *LL;
    int ldS, m, nbk;
    double *V;
    int ldw;
    double *nu, *zvec;
    int *counts;
    const double *G; // gathered rows of H P (or the H P table itself, bp.grow)
    double *Bout;    // fp64 rows of B (fp64 configuration)
    int ld;
    BPlanes bp;
    SweepCtl *ctl;
    unsigned eb;          // epoch base of this sweep's flag values
    int n_b, n_bcols, n_t;
    int n_cus;  // > 0: block n_cus is an empty spacer (in-order dispatch would put it on the chain workgroup's CU)
    int fault;  // test aid (ekf_debug_stall_next_sweep): the chain workgroup leaves at once, as if it had never become resident
    unsigned long long *trace; // debug builds (-DEKF_SWEEP_TRACE): per-panel time stamps of the roles, see PS_TRACE
};

// debug aid (scripts/persist_trace.py): 10 ns stamps per panel k -- chain [k][0..4], first B worker 1024 + [k][0..4], last B worker
// 2048 + ..., tile workers 3072 + [k]: 0 latest end of a tile of column k + 1, 1 latest end of a panel block L(i,k), 2 latest end of any tile
#ifdef EKF_SWEEP_TRACE
#define PS_TRACE(base, k, slot) if (a.trace && threadIdx.x == 0 && (base) != 2048) a.trace[(base) + 8 * (k) + (slot)] = wall_clock64();
#define PS_TRACE_MAX(base, k, slot) if (a.trace && threadIdx.x == 0) atomicMax(&a.trace[(base) + 8 * (k) + (slot)], (unsigned long long)wall_clock64());
#else
#define PS_TRACE(base, k, slot)
#define PS_TRACE_MAX(base, k, slot)
#endif

__device__ __forceinline__ double ps_ld(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void ps_st(double *p, double x)
{
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ps_flag(const unsigned *f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool ps_reached(unsigned v, unsigned target) { return (int)(v - target) >= 0; }

// the error word carries the epoch too: a failure of an earlier sweep does not end this one
__device__ __forceinline__ bool ps_failing(const SweepCtl *ctl, unsigned eb)
{
    const unsigned v = ps_flag(&ctl->err);
    return v != 0u && v - eb < PS_EPOCH_STEP;
}
__device__ __forceinline__ void ps_fail(SweepCtl *ctl, int *counts, unsigned eb, unsigned code)
{
    __hip_atomic_store(&ctl->pad[0], code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // which wait (debugging aid)
    __hip_atomic_store(&ctl->err, eb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicMax(&counts[CNT_ERR], (int)EKF_ERR_TIMEOUT);
}

// one lane polls `flag` until it reaches `target`; false when the launch is failing (somebody's error, or this wait's time-out)
// (LAZY: a wait nobody on the critical path depends on backs off to ~0.5 us between polls: hundreds of workgroups polling one word
// every hundred cycles delay the very store they are waiting for)
template <bool LAZY = false>
__device__ __forceinline__ bool ps_spin(const unsigned *flag, unsigned target, SweepCtl *ctl, int *counts, unsigned eb, unsigned code)
{
    if (ps_reached(ps_flag(flag), target)) return true;
    const long long t0 = wall_clock64();
    unsigned n = 0;
    for (;;) {
        if (LAZY && n >= 4) __builtin_amdgcn_s_sleep(16);
        else __builtin_amdgcn_s_sleep(2);
        if (ps_reached(ps_flag(flag), target)) return true;
        if ((++n & 31u) == 0) {
            if (ps_failing(ctl, eb)) return false;
            if (wall_clock64() - t0 > PS_TIMEOUT_TICKS) {
                ps_fail(ctl, counts, eb, code);
                return false;
            }
        }
    }
}

// wavefront-wide: flag >= target (lane 0 polls, no barrier: the wavefronts of a workgroup wait independently); uniform in the wavefront
template <bool LAZY = false>
__device__ __forceinline__ bool ps_wwait(const unsigned *flag, unsigned target, SweepCtl *ctl, int *counts, unsigned eb, unsigned code)
{
    int o = 1;
    if ((threadIdx.x & 63) == 0) o = ps_spin<LAZY>(flag, target, ctl, counts, eb, code) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(o) != 0;
}
// wavefront-wide: flags[first + step t] >= target for t = 0 .. count - 1 (count <= 64: lane t polls its own word)
__device__ __forceinline__ bool ps_wwait_strided(const unsigned *flags, int first, int step, int count, unsigned target, SweepCtl *ctl, int *counts,
                                                 unsigned eb, unsigned code)
{
    const int lane = threadIdx.x & 63;
    const long long t0 = wall_clock64();
    unsigned n = 0;
    for (;;) {
        const bool ready = lane >= count || ps_reached(ps_flag(flags + first + step * lane), target);
        if (__builtin_amdgcn_ballot_w64(!ready) == 0ull) return true;
        if (n < 4) __builtin_amdgcn_s_sleep(2);
        else __builtin_amdgcn_s_sleep(16);
        if ((++n & 31u) == 0) {
            if (ps_failing(ctl, eb)) return false;
            if (wall_clock64() - t0 > PS_TIMEOUT_TICKS) {
                if (lane == 0) ps_fail(ctl, counts, eb, code);
                return false;
            }
        }
    }
}

// workgroup-wide: flag >= target (thread 0 polls); uniform result
__device__ __forceinline__ bool ps_wait(const unsigned *flag, unsigned target, SweepCtl *ctl, int *counts, unsigned eb, unsigned code)
{
    __shared__ int ok_s;
    if (threadIdx.x == 0) ok_s = ps_spin(flag, target, ctl, counts, eb, code) ? 1 : 0;
    __syncthreads();
    const int r = ok_s;
    __syncthreads();
    return r != 0;
}

// workgroup-wide: flags[0 .. count) >= target, polled by the lanes of wavefront 0 (count <= 64 per pass); uniform result
__device__ __forceinline__ bool ps_wait_all(const unsigned *flags, int count, unsigned target, SweepCtl *ctl, int *counts, unsigned eb, unsigned code)
{
    __shared__ int okv_s;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int ok = 1;
        const long long t0 = wall_clock64();
        unsigned n = 0;
        for (;;) {
            bool ready = true;
            for (int b = 0; b < count; b += 64) {
                const int j = b + lane;
                if (j < count) ready = ready && ps_reached(ps_flag(flags + j), target);
            }
            if (__builtin_amdgcn_ballot_w64(!ready) == 0ull) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++n & 31u) == 0) {
                if (ps_failing(ctl, eb)) { ok = 0; break; }
                if (wall_clock64() - t0 > PS_TIMEOUT_TICKS) {
                    if (lane == 0) ps_fail(ctl, counts, eb, code);
                    ok = 0;
                    break;
                }
            }
        }
        if (lane == 0) okv_s = ok;
    }
    __syncthreads();
    const int r = okv_s;
    __syncthreads();
    return r != 0;
}

// intra-workgroup hand-off through an LDS word (the chain workgroup's two helper wavefronts): wait until *word >= value, or until
// the workgroup's failure word is raised / the wall-time bound runs out (then false, with the failure word raised for the partner)
__device__ __forceinline__ bool ps_lds_wait(int *word, int value, int *fail_s)
{
    if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= value) return true;
    const long long t0 = wall_clock64();
    unsigned n = 0;
    for (;;) {
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= value) return true;
        if ((++n & 15u) == 0) {
            if (__hip_atomic_load(fail_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return false;
            if (wall_clock64() - t0 > PS_TIMEOUT_TICKS) {
                __hip_atomic_store(fail_s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return false;
            }
        }
    }
}

// every storing wavefront drains its stores, then ONE lane raises the flag
__device__ __forceinline__ void ps_publish(unsigned *flag, unsigned value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A B' of two 32 x 32 blocks in LDS, this wavefront's 16 x 16 quadrant (w >> 1, w & 1) in accumulator layout: element q of the
// lane = row 16 bi + (lane >> 4) + 4 q, column 16 bj + (lane & 15)
__device__ __forceinline__ acc4_t ps_prod_abt(const double (*A)[NB + 1], const double (*Bm)[NB + 1])
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    return quad_prod<true>(acc4_t{0, 0, 0, 0}, A, Bm, w >> 1, w & 1, lane & 15, lane >> 4);
}

// tile t of the column-major enumeration: columns j = 0 .. nbk - 1, rows i = j .. nbk (row nbk is the nu row)
__device__ __forceinline__ void ps_tile(int t, int nbk, int &i, int &j)
{
    int jj = 0, rem = t;
    while (rem >= nbk - jj + 1) {
        rem -= nbk - jj + 1;
        ++jj;
    }
    j = jj;
    i = jj + rem;
}

// digit planes of one 32 x 32 block of L (LDS) with write-through stores: thread (k half, 8-byte half, row) of the first 128
// threads cuts eight values and stores eight bytes per plane (store_l_planes of chol_bplanes.h, one block, sc1)
__device__ __forceinline__ void ps_store_l_planes(const BPlanes &bp, int m, int i0, const double (*sL)[NB + 1], int k0)
{
    const int tid = threadIdx.x;
    if (tid >= 128) return;
    const int r = tid & 31, kg = (tid >> 5) & 1, hf = (tid >> 6) & 1;
    const int er = i0 + r < m ? bp.lexp[i0 + r] - 1022 : 0;
    const int sh = 8 * PX_S - 2 - er;
    unsigned w[PX_S][2];
#pragma unroll
    for (int s = 0; s < PX_S; ++s) w[s][0] = w[s][1] = 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int col = 16 * kg + 8 * hf + c;
        const double v = (i0 + r < m) ? sL[r][col] : 0.0;
        const unsigned long long dw = px_digit_word(v, sh);
#pragma unroll
        for (int s = 0; s < PX_S; ++s) w[s][c >> 2] |= px_digit_byte(dw, s) << (8 * (c & 3));
    }
    const size_t off = ((size_t)(i0 / NB) * bp.nbk + k0 / NB) * 1024 + kg * 512 + r * 16 + hf * 8;
#pragma unroll
    for (int s = 0; s < PX_S; ++s)
        __hip_atomic_store((unsigned long long *)(bp.Lq + (size_t)s * bp.l_stride + off), ((unsigned long long)w[s][1] << 32) | w[s][0],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// What a B worker that owns ONE block of columns (CHAINED: its next call is the next panel of the same columns) knows about the next
// panel when it gets there: the tail of a panel looks once at every flag the next panel waits for (wavefront-uniform answers; a
// worker that trails the chain finds all of them up).  Only these three bits travel between the calls: operands requested ahead
// would have to stay in registers across the loop's back edge, and the compiler parks them in scratch (measured: 13.6 us per panel).
struct PsBNext {
    bool sums, last, inv; // all L(k+1, j), j < k, published; L(k+1, k) published; inv(L_{k+1,k+1}) published
};

// Row block k of B from digit planes, persistent form of b_rows_planes (chol_bplanes.h): the planes of L come from other
// workgroups of this launch (sc1 buffer loads), the planes of the finished rows of B are this workgroup's own (plain loads; the
// barriers of a panel's tail drain its stores long before anyone reads them).  Every wait is per WAVEFRONT (no barrier around a
// poll): a wavefront whose wait fails raises fail_s and the workgroup leaves together at the next barrier.
// pool: [0..3] the wavefronts' partial sums (then [0] the finished block), [4] the right-hand side, [5] CHAINED: the digit planes of
// the block just finished, for the wavefront that owns it in the next panel (no trip through memory, no barrier for visibility).
// Fixed cost per panel before this form (profiles/r05_persist_layout.txt): ~7 of 9 us -- flag polls and first operands one round trip
// after the other, the last block's planes read back from memory behind a barrier, exponents loaded in the conversion.
__device__ __forceinline__ bool ps_b_rows_planes(const bool CHAINED, const PsArgs &a, int k, int bcol, double (*pool)[NB][NB + 1],
                                                 double (*sLi)[NB + 1], int *fail_s, PsBNext &nx, const int ec)
{
    const BPlanes &bp = a.bp;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kg = lane >> 5, idx = lane & 31;
    const int k0 = k * NB, c0 = bcol * NB, kp = k;
    const int m = a.m, ld = a.ld;
    unsigned *lrdy = a.ctl->flags + PS_NBC * PS_NBC;
    const int tb = 1024;
    (void)tb;
    int8_t *sBq = (int8_t *)pool[5]; // [plane][k half][column][16]
    // this thread's four elements of G_k and the scale of its row of L (cold, needed at the end)
    const int r4 = tid >> 3, cg = (tid & 7) * 4;
    double g4[4];
    {
        const int gr = bp.grow ? bp.grow[k0 + r4] : k0 + r4;
#pragma unroll
        for (int e = 0; e < 4; ++e) g4[e] = gr >= 0 ? a.G[(size_t)gr * ld + c0 + cg + e] : 0.0;
    }
    const int er4 = k0 + r4 < m ? bp.lexp[k0 + r4] - 1022 : 0;
    bp_v16i acc[PX_S];
#pragma unroll
    for (int L = 0; L < PX_S; ++L)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[L][r] = 0;
    const __amdgpu_buffer_rsrc_t lq = __builtin_amdgcn_make_buffer_rsrc((void *)bp.Lq, 0, (int)((size_t)PX_S * bp.l_stride), 0x00020000);
    const unsigned pl_off = (unsigned)(((size_t)kp * bp.nbk * 2 + kg) * 512 + idx * 16);      // + j * 1024 + s * l_stride
    const int8_t *pb = bp.Bq + ((size_t)kg * bp.ldq + c0 + idx) * 16;                            // + 2 j * ldq * 16 + s * b_stride
    const size_t bstep = (size_t)2 * bp.ldq * 16;
    // inv(L_kk): a B worker that trails the chain finds it published already -- its four elements travel beside the sums
    double gv[4];
    bool have_inv = CHAINED && nx.inv;
    if (!have_inv) {
        have_inv = ps_reached(ps_flag(&a.ctl->inv_ready), a.eb + k + 1);
        have_inv = __builtin_amdgcn_readfirstlane(have_inv ? 1 : 0) != 0;
    }
    if (have_inv) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256;
            gv[q] = ps_ld(a.V + (size_t)(k0 + i / NB) * a.ldw + k0 + i % NB);
        }
    }
#ifndef PS_STAGES
#define PS_STAGES 2 // operand blocks in flight (a third stage spills 49 registers at two workgroups per CU: 12.0 against 9.5 us per panel)
#endif
    bp_v4i la[2][PX_S], lb[2][PX_S];
#define PSB_LOAD(S_, J_)                                                                                          \
    _Pragma("unroll") for (int s = 0; s < PX_S; ++s) {                                                            \
        la[S_][s] = __builtin_amdgcn_raw_buffer_load_b128(lq, pl_off + (unsigned)(J_) * 1024u + (unsigned)(s * bp.l_stride), 0, 16); \
        lb[S_][s] = *(const bp_v4i *)(pb + (size_t)(J_) * bstep + (size_t)s * bp.b_stride);                       \
    }
#define PSB_MMA(S_)                                                                                               \
    _Pragma("unroll") for (int s = 0; s < PX_S; ++s)                                                              \
        _Pragma("unroll") for (int t = 0; t < PX_S - s; ++t)                                                      \
            acc[s + t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(la[S_][s], lb[S_][t], acc[s + t], 0, 0, 0);
    const bool mine = kp > 0 && ((kp - 1) & 3) == wv;
    bool ok = true;
    // the blocks j < kp - 1 (their L(k, j) were published at least one panel ago), this wavefront's j = wv, wv + 4, ...
    {
        const int jA = kp - 1;
        const int cnt = jA > wv ? (jA - wv + 3) / 4 : 0;
        if (cnt > 0 && !(CHAINED && nx.sums))
            ok = ps_wwait_strided(lrdy + (size_t)kp * PS_NBC, wv, 4, cnt, a.eb + 1, a.ctl, a.counts, a.eb, 0x5000u + k);
        if (ok) {
            if (cnt > 0) { PSB_LOAD(0, wv) }
            for (int i = 0; i < cnt; i += 2) {
                if (i + 1 < cnt) { PSB_LOAD(1, wv + 4 * (i + 1)) }
                PSB_MMA(0)
                if (i + 1 < cnt) {
                    if (i + 2 < cnt) { PSB_LOAD(0, wv + 4 * (i + 2)) }
                    PSB_MMA(1)
                }
            }
        }
    }
    if (bcol == a.bp.bcol0) { PS_TRACE(tb, k, 1) }
    // the last block, j = kp - 1, by the wavefront whose turn it is.  CHAINED: its planes of B are in LDS since the previous tail (whose
    // last barrier everyone has passed).  Otherwise (several blocks of columns per worker) they are this workgroup's own stores of an
    // earlier call: every wavefront's stores are older than the loads it has just waited for, so one barrier makes them visible.
    // (One copy of the products for both cases: the address is a flat one.)
    if (kp > 0) {
        if (!CHAINED) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (ok && mine) {
            if (!(CHAINED && nx.last)) ok = ps_wwait(lrdy + (size_t)kp * PS_NBC + kp - 1, a.eb + 1, a.ctl, a.counts, a.eb, 0x5100u + k);
            if (ok) {
                const int8_t *own = CHAINED ? (const int8_t *)(sBq + ((size_t)kg * NB + idx) * 16) : pb + (size_t)(kp - 1) * bstep;
                const size_t own_step = CHAINED ? (size_t)2 * NB * 16 : bp.b_stride;
#pragma unroll
                for (int s = 0; s < PX_S; ++s) {
                    la[0][s] = __builtin_amdgcn_raw_buffer_load_b128(lq, pl_off + (unsigned)(kp - 1) * 1024u + (unsigned)(s * bp.l_stride), 0, 16);
                    lb[0][s] = *(const bp_v4i *)(own + (size_t)s * own_step);
                }
                PSB_MMA(0)
            }
        }
    }
#undef PSB_LOAD
#undef PSB_MMA
    if (bcol == a.bp.bcol0) { PS_TRACE(tb, k, 2) }
    // inv(L_kk), unless it travelled with the sums: every wavefront waits for itself, every thread fetches its four elements
    if (ok && !have_inv) {
        ok = ps_wwait(&a.ctl->inv_ready, a.eb + k + 1, a.ctl, a.counts, a.eb, 0x5200u + k);
        if (ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + q * 256;
                gv[q] = ps_ld(a.V + (size_t)(k0 + i / NB) * a.ldw + k0 + i % NB);
            }
        }
    }
    if (bcol == a.bp.bcol0) { PS_TRACE(tb, k, 3) }
    if (ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) sLi[(tid + q * 256) / NB][(tid + q * 256) % NB] = gv[q];
    } else if (lane == 0) *fail_s = 1;
    // one look at everything the NEXT panel waits for (lane j: L(k+1, j) published?  all lanes: how many inverses are out); the
    // answer is read behind the barrier below
    const int K = k + 1;
    const bool look = CHAINED && K < a.nbk;
    unsigned f_l = 0, f_inv = 0;
    if (look) {
        if (lane < K) f_l = ps_flag(lrdy + (size_t)K * PS_NBC + lane);
        f_inv = ps_flag(&a.ctl->inv_ready);
    }
    // levels -> fp64 (no loads here: the column scale came with the call, the row scales of L are applied to the sum below); the four
    // wavefronts' partial sums meet in LDS (pool[0..3]), one barrier
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
        double tsum = (double)acc[PX_S - 1][r];
#pragma unroll
        for (int L = PX_S - 2; L >= 0; --L) tsum = fma(tsum, 1.0 / 256.0, (double)acc[L][r]);
        pool[wv][row][idx] = ldexp(tsum, ec - 12);
    }
    __syncthreads();
    if (bcol == a.bp.bcol0) { PS_TRACE(tb, k, 5) }
    if (*fail_s) return false;
    nx.sums = nx.last = nx.inv = false;
    if (look) {
        const bool up = lane >= K || ps_reached(f_l, a.eb + 1);
        const unsigned long long down = __builtin_amdgcn_ballot_w64(!up);
        nx.sums = K < 2 || (down & ((1ull << (K - 1)) - 1ull)) == 0ull;
        nx.last = ((down >> (K - 1)) & 1ull) == 0ull;
        nx.inv = __builtin_amdgcn_readfirstlane(ps_reached(f_inv, a.eb + K + 1) ? 1 : 0) != 0;
    }
    double(*sR)[NB + 1] = pool[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        sR[r4][cg + e] = g4[e] - ldexp((pool[0][r4][cg + e] + pool[1][r4][cg + e]) + (pool[2][r4][cg + e] + pool[3][r4][cg + e]), er4);
    __syncthreads();
    double(*sO)[NB + 1] = pool[0];
    {   // B_k = Linv_k R on the fp64 MFMA, one 16 x 16 quadrant per wavefront
        const int bi = wv >> 1, bj = wv & 1, lr = lane & 15, lk = lane >> 4;
        const acc4_t o = quad_prod<false>(acc4_t{0, 0, 0, 0}, sLi, sR, bi, bj, lr, lk);
#pragma unroll
        for (int q = 0; q < 4; ++q) sO[16 * bi + lk + 4 * q][16 * bj + lr] = o[q];
    }
    __syncthreads();
    if (bcol == a.bp.bcol0) { PS_TRACE(tb, k, 6) }
    {   // the block's digit planes: thread (k half, 4-row quarter, column) cuts four rows of its column: four bytes per plane
        // (column = tid & 31 = this thread's lane & 31: the column scale that came with the call is this column's)
        const int col = tid & 31, qr = (tid >> 5) & 3, kh = tid >> 7;
        const int sh = 8 * PX_S - 2 - ec;
        unsigned w[PX_S];
#pragma unroll
        for (int s = 0; s < PX_S; ++s) w[s] = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * kh + 4 * qr + i;
            const unsigned long long dw = px_digit_word_checked(k0 + row < m ? sO[row][col] : 0.0, sh, (c0 + col >= bp.c_live0 && c0 + col < bp.n_live) ? a.counts : nullptr);
#pragma unroll
            for (int s = 0; s < PX_S; ++s) w[s] |= px_digit_byte(dw, s) << (8 * i);
        }
        int8_t *dst = bp.Bq + ((size_t)(2 * kp + kh) * bp.ldq + c0 + col) * 16 + 4 * qr;
#pragma unroll
        for (int s = 0; s < PX_S; ++s) {
            *(unsigned *)(dst + (size_t)s * bp.b_stride) = w[s];
            if (CHAINED) *(unsigned *)(sBq + ((size_t)(s * 2 + kh) * NB + col) * 16 + 4 * qr) = w[s];
        }
    }
    if (bcol == a.bp.bcol0) { PS_TRACE(tb, k, 7) }
    // (frees the LDS blocks; the stores of the planes drain beside the next panel's sums: that panel's barriers come before anyone reads them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    return true;
}

// Row block k of B on the fp64 MFMA (fp64 configuration), persistent form of the B role of k_chol_step: L' comes mirrored from LL
// (other workgroups' write-through stores, sc1 loads), the finished rows of B are this workgroup's own.  Waits as above.
__device__ __forceinline__ bool ps_b_rows_f64(const PsArgs &a, int k, int bcol, double (*pool)[NB][NB + 1], double (*sLi)[NB + 1], int *fail_s)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lm = lane & 15, lq = lane >> 4;
    const int k0 = k * NB, c0 = bcol * NB, kp = k, ld = a.ld, ldS = a.ldS;
    double(*red)[NB][NB + 1] = pool;
    double(*sR)[NB + 1] = pool[2];
    unsigned *lrdy = a.ctl->flags + PS_NBC * PS_NBC;
    const int r4 = tid >> 3, cg = (tid & 7) * 4;
    double g4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g4[e] = a.G[(size_t)(k0 + r4) * ld + c0 + cg + e];
    acc4_t acc[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) acc[bi][bj] = acc4_t{0, 0, 0, 0};
    // one finished block j: 32 rows of k in eight steps of four; a[row 16 bi + lm][k = lq] = L(k0 + 16 bi + lm, 32 j + 4 st + lq)
    auto block = [&](int j) {
        double la[8][2], lb[8][2];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const size_t kr = (size_t)j * NB + st * 4 + lq;
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                la[st][bb] = ps_ld(a.LL + kr * ldS + k0 + 16 * bb + lm);
                lb[st][bb] = a.Bout[kr * ld + c0 + 16 * bb + lm];
            }
        }
#pragma unroll
        for (int st = 0; st < 8; ++st)
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj) acc[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[st][bi], lb[st][bj], acc[bi][bj], 0, 0, 0);
    };
    bool ok = true;
    {
        const int jA = kp - 1;
        const int cnt = jA > wv ? (jA - wv + 3) / 4 : 0;
        if (cnt > 0) ok = ps_wwait_strided(lrdy + (size_t)kp * PS_NBC, wv, 4, cnt, a.eb + 1, a.ctl, a.counts, a.eb, 0x6000u + k);
        if (ok)
            for (int j = wv; j < jA; j += 4) block(j);
    }
    if (ok && kp > 0 && ((kp - 1) & 3) == wv) {
        ok = ps_wwait(lrdy + (size_t)kp * PS_NBC + kp - 1, a.eb + 1, a.ctl, a.counts, a.eb, 0x6100u + k);
        if (ok) block(kp - 1);
    }
    if (ok) ok = ps_wwait(&a.ctl->inv_ready, a.eb + k + 1, a.ctl, a.counts, a.eb, 0x6200u + k);
    if (ok) {
        double gv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + q * 256;
            gv[q] = ps_ld(a.V + (size_t)(k0 + i / NB) * a.ldw + k0 + i % NB);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sLi[(tid + q * 256) / NB][(tid + q * 256) % NB] = gv[q];
    } else if (lane == 0) *fail_s = 1;
    if (wv >= 2) {
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[wv - 2][16 * bi + lq + 4 * q][16 * bj + lm] = acc[bi][bj][q];
    }
    __syncthreads();
    if (*fail_s) return false;
    if (wv < 2) {
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[wv][16 * bi + lq + 4 * q][16 * bj + lm] += acc[bi][bj][q];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) sR[r4][cg + e] = g4[e] - (red[0][r4][cg + e] + red[1][r4][cg + e]);
    __syncthreads();
    {
        const int bi = wv >> 1, bj = wv & 1, lr = lane & 15, lk = lane >> 4;
        const acc4_t o = quad_prod<false>(acc4_t{0, 0, 0, 0}, sLi, sR, bi, bj, lr, lk);
#pragma unroll
        for (int q = 0; q < 4; ++q) a.Bout[(size_t)(k0 + 16 * bi + lk + 4 * q) * ld + c0 + 16 * bj + lr] = o[q];
    }
    __syncthreads();
    return true;
}

template <bool PL> // PL: the rows of B from int8 digit planes (EKF_PRECISION_F32_EXACT); otherwise fp64 rows of B
__global__ void __launch_bounds__(256, 2) k_chol_persist(PsArgs a)
{
    // LDS, overlaid by role: [0] inv(L_kk) (every role), then  chain: five 32 x 32 blocks + the scratch of block_chol_inv32_w2;
    // B worker: five blocks;  tile worker: four blocks + PS_CACHE cached tiles (76 KB: two workgroups fit a CU)
    typedef double blk_t[NB][NB + 1];
    __shared__ __attribute__((aligned(32))) blk_t lds_blocks[1 + 4 + PS_CACHE];
    static_assert(sizeof(W2Scratch) <= sizeof(blk_t) * (PS_CACHE - 1), "the factorisation scratch overlays the tile cache");
    blk_t *pool = lds_blocks + 1;
    __shared__ int hfail_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int bi = w >> 1, bj = w & 1, lr = lane & 15, lk = lane >> 4;
    SweepCtl *ctl = a.ctl;
    unsigned *done = ctl->flags;
    unsigned *lrdy = ctl->flags + PS_NBC * PS_NBC;
    const int m = a.m, nbk = a.nbk, ldS = a.ldS;
    // role of this block: 0 chain, 1 .. n_b B workers, then tile workers; block n_cus (if any) is the chain's empty neighbour
    int role = (int)blockIdx.x;
    if (a.n_cus > 0) {
        if ((int)blockIdx.x == a.n_cus) return;
        if ((int)blockIdx.x > a.n_cus) role = (int)blockIdx.x - 1;
    }
    if (tid == 0) hfail_s = 0;
    __syncthreads();
    const int ticket = role;
    double(*sX)[NB + 1] = lds_blocks[0];
    if (ticket == 0) {
        if (a.fault) return; // injected stall: every other role now waits for an inverse that never comes -> watchdog
        // ---------------------------------------------------------------------------------------------------- critical chain
        // Panel f: wavefronts 0, 1 factorise A_ff (block_chol_inv32_w2); wavefronts 2, 3 meanwhile prepare row r = f + 1 -- the
        // two tiles the chain needs right after this factorisation:
        //     Y1 = S(r,f)   with the panels <= f - 1 applied,     Y2 = S(r,r)   with the panels <= f - 1 applied,
        // taking the tiles from their owners with the panels <= f - 2 applied and applying panel kk = f - 1 THEMSELVES (inv(L_kk) and
        // L1 = L(f,kk) are still in LDS):  L2 = S(r,kk) inv(L_kk)',  Y1 = U1 - L2 L1',  Y2 = U2 - L2 L2'.  The owners' results the
        // chain depends on are therefore one panel older than the panel it eliminates: a whole cycle of slack for the hand-off
        // (flag, 24 KB through L2) that otherwise sits on the critical path.  Y1 is final: it goes back to S for the tasks of
        // level f (done(r,f) = f, raised here).  Then, all four wavefronts:  L1 = Y1 inv(L_ff)',  A(r,r) = Y2 - L1 L1'.
        double(*sA)[NB + 1] = pool[0];
        double(*Y0)[NB + 1] = pool[1];
        double(*Y1)[NB + 1] = pool[2];
        double(*Y2)[NB + 1] = pool[3];
        double(*sLi)[NB + 1] = pool[4];
        W2Scratch *ws = reinterpret_cast<W2Scratch *>(&pool[5]);
        __shared__ int hs_l2, hs_wb;
        for (int i = tid; i < NB * NB; i += 256) {
            const int r = i / NB, c = i % NB;
            sA[r][c] = (r < m && c <= r) ? a.S[(size_t)r * ldS + c] : ((r == c) ? 1.0 : 0.0);
        }
        if (tid == 0) { hs_l2 = 0; hs_wb = 0; }
        __syncthreads();
        for (int k = 0; k < nbk; ++k) {
            const int k0 = k * NB, k1 = k0 + NB;
            const bool more = k + 1 < nbk;
            PS_TRACE(0, k, 0)
            auto helpers = [&](int h) {
                if (!more) return;
                const int lm = lane & 15, lq = lane >> 4;
                const int r0 = k1; // first row of block row r = k + 1
                // rows of this wavefront: 16 h .. 16 h + 15 of the 32; accumulator layout of its quadrants (h, 0), (h, 1)
                if (k == 0) { // nothing to apply yet: the tiles as k_assemble_S wrote them
#pragma unroll
                    for (int bjj = 0; bjj < 2; ++bjj)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int row = 16 * h + lq + 4 * q, col = 16 * bjj + lm;
                            const bool rl = r0 + row < m;
                            Y1[row][col] = rl ? a.S[(size_t)(r0 + row) * ldS + k0 + col] : 0.0;
                            Y2[row][col] = (rl && col <= row) ? a.S[(size_t)(r0 + row) * ldS + r0 + col] : 0.0;
                        }
                    return;
                }
                const int kk = k - 1, kk0 = kk * NB;
                bool ok = true;
                if (kk > 0) // the three tiles (k+1, k-1 .. k+1) with the panels <= k - 2 applied: adjacent flags, one look for all
                    ok = ps_wwait_strided(done + (size_t)(k + 1) * PS_NBC, kk, 1, 3, a.eb + kk, ctl, a.counts, a.eb, 0x1000u + k);
                if (!ok) {
                    if (lane == 0) hfail_s = 1;
                    return;
                }
#ifdef EKF_SWEEP_TRACE
                if (a.trace && lane == 0 && h == 0) a.trace[8 * k + 4] = wall_clock64(); // the three tiles are published
#endif
                // U0 = S(r,kk) straight into the MFMA A-operand layout (row 16 h + lm, k = 4 st + lq), U1, U2 in accumulator layout
                double ua[8], u1[2][4], u2[2][4];
                {
                    const bool rl = r0 + 16 * h + lm < m;
                    const double *src = a.S + (size_t)(r0 + 16 * h + lm) * ldS + kk0 + lq;
#pragma unroll
                    for (int st = 0; st < 8; ++st) ua[st] = rl ? ps_ld(src + 4 * st) : 0.0;
                }
#pragma unroll
                for (int bjj = 0; bjj < 2; ++bjj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 16 * h + lq + 4 * q, col = 16 * bjj + lm;
                        const bool rl = r0 + row < m;
                        u1[bjj][q] = rl ? ps_ld(a.S + (size_t)(r0 + row) * ldS + k0 + col) : 0.0;
                        u2[bjj][q] = (rl && col <= row) ? ps_ld(a.S + (size_t)(r0 + row) * ldS + r0 + col) : 0.0;
                    }
                // L2 = U0 inv(L_kk)' : quadrants (h, 0), (h, 1), this wavefront's rows of Y0
#pragma unroll
                for (int bjj = 0; bjj < 2; ++bjj) {
                    acc4_t c = {0, 0, 0, 0};
#pragma unroll
                    for (int st = 0; st < 8; ++st) c = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[st], sX[16 * bjj + lm][4 * st + lq], c, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) Y0[16 * h + lq + 4 * q][16 * bjj + lm] = c[q];
                }
                asm volatile("" ::: "memory"); // (this wavefront's LDS stores and loads complete in order)
                if (h == 0 && lane == 0) __hip_atomic_store(&hs_l2, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (wavefront 2's rows of L2, for wavefront 3)
                // Y1 = U1 - L2 L1' : final; to LDS and back to S
#pragma unroll
                for (int bjj = 0; bjj < 2; ++bjj) {
                    const acc4_t t = quad_prod<true>(acc4_t{0, 0, 0, 0}, Y0, sLi, h, bjj, lm, lq);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 16 * h + lq + 4 * q, col = 16 * bjj + lm;
                        const double nv = u1[bjj][q] - t[q];
                        Y1[row][col] = nv;
                        if (r0 + row < m) ps_st(a.S + (size_t)(r0 + row) * ldS + k0 + col, nv);
                    }
                }
                // Y2 = U2 - L2 L2' : the lower quadrants -- (0, 0) here, (1, 0) and (1, 1) there, (1, 0) needs the other rows of L2
                if (h == 1) {
                    // (bounded like every other spin: wavefront 2 may have left through the failure path -- its own time-out or
                    // somebody's error -- after this wavefront saw the same three flags arrive)
                    if (!ps_lds_wait(&hs_l2, k, &hfail_s)) return;
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int bjj = 0; bjj < 2; ++bjj) {
                    if (bjj > h) continue;
                    const acc4_t t = quad_prod<true>(acc4_t{0, 0, 0, 0}, Y0, Y0, h, bjj, lm, lq);
#pragma unroll
                    for (int q = 0; q < 4; ++q) Y2[16 * h + lq + 4 * q][16 * bjj + lm] = u2[bjj][q] - t[q];
                }
                // the write-back of Y1 has left both wavefronts: done(r, k) = k
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (h == 1) {
                    if (lane == 0) __hip_atomic_store(&hs_wb, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    if (!ps_lds_wait(&hs_wb, k, &hfail_s)) return;
                    if (lane == 0) __hip_atomic_store(&done[(size_t)(k + 1) * PS_NBC + k], a.eb + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            if (!block_chol_inv32_w2<2>(sA, sX, ws, helpers) && tid == 0) atomicMax(&a.counts[CNT_ERR], (int)EKF_ERR_NOT_POSITIVE_DEFINITE);
            PS_TRACE(0, k, 1)
#ifdef EKF_SWEEP_TRACE
            if (a.trace && tid == 0) { // (written before the join's barrier)
                a.trace[8 * k + 5] = g_w2_clock[0];
                a.trace[8 * k + 6] = g_w2_clock[1];
                a.trace[8 * k + 7] = g_w2_clock[2] > g_w2_clock[3] ? g_w2_clock[2] : g_w2_clock[3];
            }
#endif
            // inv(L_kk) to V, write-through; the flag goes out behind the first product (the stores drain beside it).  (Measured and
            // dropped, profiles/r05_persist_layout.txt: the stores by two wavefronts only, drains and both flags at the panel's end.)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = tid + 256 * q;
                ps_st(a.V + (size_t)(k0 + (e >> 5)) * a.ldw + k0 + (e & 31), sX[e >> 5][e & 31]);
            }
            if (!more) {
                ps_publish(&ctl->inv_ready, a.eb + k + 1);
                break;
            }
            if (hfail_s) return;
            {   // L1 = L(k+1,k) = Y1 inv(L_kk)'
                const acc4_t l = ps_prod_abt(Y1, sX);
#pragma unroll
                for (int q = 0; q < 4; ++q) sLi[16 * bi + lk + 4 * q][16 * bj + lr] = l[q];
            }
            ps_publish(&ctl->inv_ready, a.eb + k + 1); // (its barrier also orders sLi)
            PS_TRACE(0, k, 2)
            {   // A(k+1,k+1) = Y2 - L1 L1', identity-padded beyond the live rows
                const acc4_t t = ps_prod_abt(sLi, sLi);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
                    const bool lv = (k1 + r < m) && (c <= r);
                    sA[r][c] = lv ? Y2[r][c] - t[q] : ((r == c) ? 1.0 : 0.0);
                }
            }
            __syncthreads();
            PS_TRACE(0, k, 3)
        }
        return;
    }
    if (ticket > a.n_b) {
        // ------------------------------------------------------------------------------------------------------ tile workers
        // A worker owns the tiles me, me + n_t, ... of the column-major enumeration.  Its tasks, per tile (i, j):
        //   * apply the panels 0 .. upd - 1 (upd = j; the chain workgroup takes the last panel of the tile left of the diagonal and
        //     the last two of a diagonal tile itself).  Panel p can be applied in two forms:
        //       L form     S(i,j) -= L(i,p) L(j,p)'  from the official fp64 blocks of L (flags lrdy): a streaming product straight
        //                  from L2 into the MFMA operand layout, no LDS, no barrier, any number of consecutive panels per task;
        //       S form     from the panel blocks S(i,p), S(j,p) and inv(L_pp) (two more 32^3 products): available one hand-off
        //                  EARLIER than the L form -- what the last panel of a tile needs, which the next column waits for;
        //   * (off the diagonal) the panel-block step at level j: L(i,j) = S(i,j) inv(L_jj)' is final -- stored as fp64 (row-major
        //     for the L form, mirrored for the fp64 rows of B) and as digit planes (exact configuration), flag lrdy(i, j); the extra
        //     block row carries nu' and yields z'.
        // Across its tiles a worker takes panel-block steps and last panels first (they unblock others), then the runnable task with
        // the least slack (key 5 j + 7 p: deadline ~ j cycles of the chain, remaining work ~ j - p tasks), and never blocks on one
        // task while another could run.
        const int me = ticket - 1 - a.n_b;
        if (me >= a.n_t) return;
        double(*sP)[NB + 1] = pool[0];
        double(*sQ)[NB + 1] = pool[1];
        double(*sLi)[NB + 1] = pool[2];
        double(*sLj)[NB + 1] = pool[3];
        __shared__ int s_ti[64], s_tj[64], s_prog[64], s_upd[64], s_choice, s_kind, s_cnt;
        const int ntiles = (nbk + 1) * (nbk + 2) / 2 - 1; // columns j = 0 .. nbk - 1, rows j .. nbk
        const int ntl = me < ntiles ? min(64, (ntiles - me + a.n_t - 1) / a.n_t) : 0;
        if (tid < 64) {
            int i = 0, j = 0, upd = 0;
            if (tid < ntl) {
                ps_tile(me + tid * a.n_t, nbk, i, j);
                const bool sub = i == j + 1 && i < nbk;
                upd = (i == j) ? max(j - 2, 0) : (sub ? max(j - 1, 0) : j);
            }
            // s_prog: panels applied (0 .. upd), upd + 1 once the panel-block step is done too (diagonal tiles have none)
            s_ti[tid] = i; s_tj[tid] = j; s_prog[tid] = 0; s_upd[tid] = upd;
        }
        __syncthreads();
        int inv_k = -1; // the panel whose inverse sits in sX
        __shared__ int s_nchoice, s_nkind, s_ncnt; // the task picked while the current one's operands travel (-3: none)
        if (tid == 0) s_nchoice = -3;
        // the scan (wavefront 0): lane t looks at its tile's next task; `exclude`: the tile of the task in flight; a non-blocking
        // scan returns -3 when nothing else can run now
        auto scan = [&](int exclude, bool blocking, int &choice, int &kind, int &cnt) {
            const long long t0 = wall_clock64();
            unsigned n = 0;
            choice = -2; kind = 0; cnt = 1;
            for (;;) {
                const unsigned inv = ps_flag(&ctl->inv_ready);
                int key = 1 << 30, lfc = 0;
                bool pending = false;
                if (lane < ntl && lane != exclude) {
                    const int p = s_prog[lane], i = s_ti[lane], j = s_tj[lane], upd = s_upd[lane];
                    const bool sub = i == j + 1 && i < nbk;
                    int kd = -1;
                    if (p < upd) {
                        pending = true;
                        // L form first (cheap; as many consecutive panels as are published, PS_BATCH at most: their operands
                        // travel together), S form if that is what is there
                        bool run = true;
#pragma unroll
                        for (int q = 0; q < PS_BATCH; ++q) {
                            bool lf = run && p + q < upd && ps_reached(ps_flag(&lrdy[(size_t)i * PS_NBC + min(p + q, PS_NBC - 1)]), a.eb + 1);
                            if (lf && j != i) lf = ps_reached(ps_flag(&lrdy[(size_t)j * PS_NBC + min(p + q, PS_NBC - 1)]), a.eb + 1);
                            run = lf;
                            lfc += lf ? 1 : 0;
                        }
                        if (lfc > 0) kd = 2;
                        else if (ps_reached(inv, a.eb + p + 1)) {
                            bool r = true;
                            if (p > 0) {
                                r = ps_reached(ps_flag(&done[(size_t)i * PS_NBC + p]), a.eb + p);
                                if (r && j != i) r = ps_reached(ps_flag(&done[(size_t)j * PS_NBC + p]), a.eb + p);
                            }
                            if (r) kd = 0;
                        }
                    } else if (i != j && p == upd) {
                        pending = true;
                        bool r = ps_reached(inv, a.eb + j + 1);
                        // the block left of the diagonal gets its last panel from the chain workgroup, which hands it back
                        if (r && sub && j > 0) r = ps_reached(ps_flag(&done[(size_t)i * PS_NBC + j]), a.eb + j);
                        if (r) kd = 1;
                    }
                    // what unblocks others first: the panel-block steps (every L-form update of that panel waits for them) and a
                    // tile's last panel (the next column waits for it); then the least slack
                    if (kd >= 0) key = ((kd == 1 || p == upd - 1) ? 0 : (1 << 20)) | ((5 * j + 7 * min(p, j)) << 8) | (kd << 6) | lane;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) key = min(key, __shfl_xor(key, o, 64));
                if (key < (1 << 30)) {
                    choice = key & 63;
                    kind = (key >> 6) & 3;
                    cnt = max(1, __shfl(lfc, choice, 64));
                    return;
                }
                if (!blocking) { choice = -3; return; }
                if (__builtin_amdgcn_ballot_w64(pending) == 0ull) { choice = -1; return; } // every tile of this worker is finished
                if (n < 8) __builtin_amdgcn_s_sleep(2);
                else __builtin_amdgcn_s_sleep(8);
                if ((++n & 31u) == 0) {
                    if (ps_failing(ctl, a.eb)) return;
                    if (wall_clock64() - t0 > PS_TIMEOUT_TICKS) {
                        if (lane == 0) ps_fail(ctl, a.counts, a.eb, 0x3000u);
                        return;
                    }
                }
            }
        };
        __syncthreads();
        for (;;) {
            if (w == 0) {
                int choice = s_nchoice, kind = s_nkind, cnt = s_ncnt;
                if (choice == -3) scan(-1, true, choice, kind, cnt);
                if (lane == 0) { s_choice = choice; s_kind = kind; s_cnt = cnt; s_nchoice = -3; }
            }
            __syncthreads();
            const int ch = s_choice;
            if (ch < 0) return; // -1 finished, -2 failing
            const int kind = s_kind, nb = s_cnt;
#ifdef EKF_SWEEP_TRACE
            int tlog = -1;
            if (a.trace && me == a.n_t / 3 && tid == 0) { // the task log of one worker: [scan done, end, kind | panel | count | tile]
                tlog = (int)a.trace[2048];
                if (tlog < 250) { a.trace[2048] = tlog + 1; a.trace[2052 + 4 * tlog] = wall_clock64(); }
            }
#endif
            const int i = s_ti[ch], j = s_tj[ch], upd = s_upd[ch];
            const int k = kind == 1 ? j : s_prog[ch]; // the panel of this task (the first one of a batch)
            const int k0 = k * NB;
            const bool is_nu = i == nbk; // the right-hand-side row: one live row (nu' / z')
            const bool sub = i == j + 1 && i < nbk;
            const int i0 = i * NB, j0 = j * NB;
            const bool cached = ch < PS_CACHE;
            double(*sC)[NB + 1] = pool[4 + (cached ? ch : 0)];
            // (panel 0 finds the tile as k_assemble_S / k_gather wrote it; the block left of the diagonal comes back from the chain)
            const bool in_lds = cached && s_prog[ch] > 0 && !(sub && kind == 1);
            int done_to = 0; // panels applied after this task (update tasks)
            if (kind == 2) {
                // ---- L form: nb consecutive panels, streaming, per wavefront (its quadrant (bi, bj) of the tile)
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
                    const bool live = is_nu ? (r == 0 && j0 + c < m) : (i0 + r < m && j0 + c < m && (i != j || c <= r));
                    if (!live) v[q] = 0.0;
                    else if (in_lds) v[q] = sC[r][c];
                    else v[q] = is_nu ? ps_ld(a.nu + j0 + c) : ps_ld(a.S + (size_t)(i0 + r) * ldS + j0 + c);
                }
                // operands in MFMA layout: a[row 16 bi + lr][k = 4 st + lk] = L(i0 + 16 bi + lr, 32 q + 4 st + lk), b likewise from row block j
                const bool arow = is_nu ? (bi == 0 && lr == 0) : (i0 + 16 * bi + lr < m);
                const bool brow = j0 + 16 * bj + lr < m;
                const double *pa_ = is_nu ? a.zvec + lk : a.LL + (size_t)(i0 + 16 * bi + lr) * ldS + lk;
                const double *pb_ = a.LL + (size_t)(j0 + 16 * bj + lr) * ldS + lk;
                // all nb <= PS_BATCH panels are requested at once: ONE round trip to L2 / the memory-side cache, then 8 MFMAs per panel
                acc4_t acc = {0, 0, 0, 0};
                double la[PS_BATCH][8], lb[PS_BATCH][8];
#pragma unroll
                for (int q = 0; q < PS_BATCH; ++q) {
                    if (q < nb) {
#pragma unroll
                        for (int st = 0; st < 8; ++st) {
                            la[q][st] = arow ? ps_ld(pa_ + (size_t)(k + q) * NB + 4 * st) : 0.0;
                            lb[q][st] = brow ? ps_ld(pb_ + (size_t)(k + q) * NB + 4 * st) : 0.0;
                        }
                    }
                }
                if (w == 0) { // while they travel: the next task (non-blocking; this tile is busy)
                    int c2, k2, n2;
                    scan(ch, false, c2, k2, n2);
                    if (lane == 0) { s_nchoice = c2; s_nkind = k2; s_ncnt = n2; }
                }
#pragma unroll
                for (int q = 0; q < PS_BATCH; ++q) {
                    if (q < nb) {
#pragma unroll
                        for (int st = 0; st < 8; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[q][st], lb[q][st], acc, 0, 0, 0);
                    }
                }
                done_to = k + nb;
                const bool last = done_to == upd;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
                    const double nv = v[q] - acc[q];
                    if (cached) sC[r][c] = nv;
                    if (!cached || last) {
                        if (is_nu) {
                            if (r == 0 && j0 + c < m) ps_st(a.nu + j0 + c, nv);
                        } else if (i0 + r < m && j0 + c < m && (i != j || c <= r)) ps_st(a.S + (size_t)(i0 + r) * ldS + j0 + c, nv);
                    }
                }
                if (!cached || last) ps_publish(&done[(size_t)i * PS_NBC + j], a.eb + done_to);
                else __syncthreads();
                PS_TRACE_MAX(3072, done_to - 1, 2)
            } else {
                // ---- S form of one panel, or the panel-block step: operands through LDS
                double pa[4], pb[4], v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + 256 * q, r = e >> 5, c = e & 31;
                    if (kind == 1 && in_lds) pa[q] = sC[r][c]; // its own tile has become a block of panel k
                    else if (is_nu) pa[q] = (r == 0 && k0 + c < m) ? ps_ld(a.nu + k0 + c) : 0.0;
                    else pa[q] = (i0 + r < m) ? ps_ld(a.S + (size_t)(i0 + r) * ldS + k0 + c) : 0.0;
                    pb[q] = (kind == 0 && j != i && j0 + r < m) ? ps_ld(a.S + (size_t)(j0 + r) * ldS + k0 + c) : 0.0;
                }
                if (kind == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
                        const bool live = is_nu ? (r == 0 && j0 + c < m) : (i0 + r < m && j0 + c < m && (i != j || c <= r));
                        if (!live) v[q] = 0.0;
                        else if (in_lds) v[q] = sC[r][c];
                        else v[q] = is_nu ? ps_ld(a.nu + j0 + c) : ps_ld(a.S + (size_t)(i0 + r) * ldS + j0 + c);
                    }
                }
                if (inv_k != k) {
                    double gv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int e = tid + 256 * q;
                        gv[q] = ps_ld(a.V + (size_t)(k0 + (e >> 5)) * a.ldw + k0 + (e & 31));
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) sX[(tid + 256 * q) >> 5][(tid + 256 * q) & 31] = gv[q];
                    inv_k = k;
                }
                if (w == 0) { // while the operands travel: the next task (non-blocking; this tile is busy)
                    int c2, k2, n2;
                    scan(ch, false, c2, k2, n2);
                    if (lane == 0) { s_nchoice = c2; s_nkind = k2; s_ncnt = n2; }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + 256 * q;
                    sP[e >> 5][e & 31] = pa[q];
                    sQ[e >> 5][e & 31] = pb[q];
                }
                __syncthreads();
                const acc4_t li = ps_prod_abt(sP, sX);
                if (kind == 1) {
                    // a block of panel k: L(i,k) is final
                    if (is_nu) {
                        if (bi == 0 && lk == 0) {
                            const int c = 16 * bj + lr;
                            if (k0 + c < m) ps_st(a.zvec + k0 + c, li[0]);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
                            sLi[r][c] = li[q];
                            if (i0 + r < m) ps_st(a.LL + (size_t)(i0 + r) * ldS + k0 + c, li[q]); // row-major: the L form of the tile updates
                        }
                        __syncthreads();
                        if (PL) ps_store_l_planes(a.bp, m, i0, sLi, k0);
                        else {
                            for (int e = tid; e < NB * NB; e += 256) { // mirrored: LL[k0 + c][i0 + r] = L(i0 + r, k0 + c)
                                const int c = e / NB, r = e % NB;
                                ps_st(a.LL + (size_t)(k0 + c) * ldS + i0 + r, (i0 + r < m) ? sLi[r][c] : 0.0);
                            }
                        }
                    }
                    ps_publish(&lrdy[(size_t)i * PS_NBC + k], a.eb + 1);
                    PS_TRACE_MAX(3072, k, 1)
                    done_to = upd + 1;
                } else {
                    const acc4_t lj = (i == j) ? li : ps_prod_abt(sQ, sX);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sLi[16 * bi + lk + 4 * q][16 * bj + lr] = li[q];
                        sLj[16 * bi + lk + 4 * q][16 * bj + lr] = lj[q];
                    }
                    __syncthreads();
                    const acc4_t u = ps_prod_abt(sLi, sLj);
                    done_to = k + 1;
                    const bool last = done_to == upd;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = 16 * bi + lk + 4 * q, c = 16 * bj + lr;
                        const double nv = v[q] - u[q];
                        if (cached) sC[r][c] = nv;
                        if (!cached || last) {
                            if (is_nu) {
                                if (r == 0 && j0 + c < m) ps_st(a.nu + j0 + c, nv);
                            } else if (i0 + r < m && j0 + c < m && (i != j || c <= r)) ps_st(a.S + (size_t)(i0 + r) * ldS + j0 + c, nv);
                        }
                    }
                    if (!cached || last) ps_publish(&done[(size_t)i * PS_NBC + j], a.eb + done_to);
                    else __syncthreads();
                    if (j == k + 1) { PS_TRACE_MAX(3072, k, 0) }
                    PS_TRACE_MAX(3072, k, 2)
                }
            }
#ifdef EKF_SWEEP_TRACE
            if (tlog >= 0 && tlog < 250) {
                a.trace[2052 + 4 * tlog + 1] = wall_clock64();
                a.trace[2052 + 4 * tlog + 2] = (unsigned long long)kind | ((unsigned long long)k << 8) | ((unsigned long long)nb << 16) | ((unsigned long long)i << 24) | ((unsigned long long)j << 32);
            }
#endif
            if (tid == 0) s_prog[ch] = done_to;
            // (the next scan is wavefront 0's, in program order behind this store; the LDS blocks are free: every product above
            // was followed by a barrier)
        }
    }
    // ---------------------------------------------------------------------------------------------------------- rows of B
    {
        const int me = ticket - 1;
        // one block of columns per worker: the tail of a panel learns what the next one would wait for (one inlined body for both
        // cases: a second copy of the sum's loop costs the kernel ~40 spilled registers)
        const bool chained = PL && a.n_b == a.n_bcols;
        PsBNext nx;
        nx.sums = nx.last = nx.inv = false;
        for (int k = 0; k < nbk; ++k) {
            for (int cb = me; cb < a.n_bcols; cb += a.n_b) {
                bool ok;
                const int tb = me == 0 ? 1024 : 2048;
                if (me == 0 || me == a.n_b - 1) { PS_TRACE(tb, k, 0) }
                if (PL) {
                    const int bcol = cb + a.bp.bcol0;
                    ok = ps_b_rows_planes(chained, a, k, bcol, pool, sX, &hfail_s, nx, a.bp.bexp[bcol * NB + (tid & 31)] - 1022);
                } else ok = ps_b_rows_f64(a, k, cb, pool, sX, &hfail_s);
                if (!ok) return;
                if (me == 0 || me == a.n_b - 1) { PS_TRACE(tb, k, 4) }
            }
        }
    }
}
