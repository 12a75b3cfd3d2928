// engine.cpp -- C ABI of the MI355X EKF engine (see include/ekf_engine.h).  Host orchestration only: every
// number is produced by the HIP kernels in kernels_*.hip; there is no CPU compute path.
#include "engine.h"
#include "../../include/ekf_test_hooks.h"

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>

#include <dlfcn.h>
#include <rccl/rccl.h> // types only: the library is loaded at run time (rccl_api)

using namespace ekf;

namespace ekf {
void launch_partition(EkfEngine *e, const EkfMatch *src, int M, const uint8_t *flags, EkfMatch *dst1, EkfMatch *dst0,
                      int *cnt1, bool map_update = false, const uint8_t *d_kdesc = nullptr, int *d_idx0 = nullptr,
                      int publish_seq = 0, bool rescue = false);
void launch_outlier_idx(EkfEngine *e, const EkfMatch *src, int M, int *idx);
} // namespace ekf

#define HIPCHK(call)                                                                                          \
    do {                                                                                                      \
        hipError_t _st = (call);                                                                              \
        if (_st != hipSuccess) {                                                                              \
            e->err = std::string(#call) + ": " + hipGetErrorString(_st);                                      \
            return EKF_ERR_HIP;                                                                               \
        }                                                                                                     \
    } while (0)

// RCCL entry points used by the sharded engine, resolved from librccl.so.1 on first use (the same library
// torch.distributed's "nccl" backend has loaded, when the host is PyTorch)
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

static RcclApi &rccl_api()
{
    static RcclApi api;
    static std::once_flag once; // engines may be created from several threads: the table is filled exactly once
    std::call_once(once, [] {
        // a copy that is already mapped (torch.distributed's "nccl" backend) first: two RCCL copies in one process would
        // each keep their own device state
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // EKF_RCCL_LIBRARY: this library and no other (a site's own RCCL build; the test suite's in-process stand-in,
        // tests/cpp/mock_rccl.cpp, which lets a one-GPU box drive the in-stream exchange with several ranks)
        if (const char *own = std::getenv("EKF_RCCL_LIBRARY")) {
            if (*own) {
                api.lib = dlopen(own, RTLD_NOW | RTLD_LOCAL);
                if (!api.lib) return;
            }
        }
        for (const char *name : names)
            if (api.lib || (api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD))) break;
        for (int i = 0; !api.lib && i < 3; ++i) api.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) return;
#define RCCL_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, sym))
        RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
        RCCL_SYM(CommInitRank, "ncclCommInitRank");
        RCCL_SYM(CommDestroy, "ncclCommDestroy");
        RCCL_SYM(GroupStart, "ncclGroupStart");
        RCCL_SYM(GroupEnd, "ncclGroupEnd");
        RCCL_SYM(Send, "ncclSend");
        RCCL_SYM(Recv, "ncclRecv");
        RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv;
    });
    return api;
}

template <typename T>
static hipError_t dalloc(T **p, size_t count, bool zero = true)
{
    if (count == 0) count = 1;
    hipError_t st = hipMalloc((void **)p, count * sizeof(T));
    if (st == hipSuccess && zero) st = hipMemset(*p, 0, count * sizeof(T));
    return st;
}

static int exchange_rows(EkfEngine *e, int what, void *base, size_t row_bytes, const std::vector<int32_t> &rb, const char *name);

// element size of H P and of its gathered rows (fp64 beside the exact downdate, else the covariance's)
static inline size_t hp_elem_bytes(const EkfEngine *e) { return e->exact ? 8 : (e->f32 ? 4 : 8); }

extern "C" {

int ekf_abi_version(void) { return 1; }

int ekf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ekf_shard_rows(int n_features, int world, int rank, int *row_begin, int *row_end)
{
    if (world <= 0 || rank < 0 || rank >= world || n_features < 0 || !row_begin || !row_end) return EKF_ERR_INVALID_ARG;
    const int per = n_features / world, extra = n_features % world;
    const int f0 = rank * per + (rank < extra ? rank : extra);
    const int f1 = f0 + per + (rank < extra ? 1 : 0);
    *row_begin = (rank == 0) ? 0 : 13 + 6 * f0;
    *row_end = 13 + 6 * f1;
    return EKF_OK;
}

const char *ekf_last_error(const EkfEngine *e) { return e ? e->err.c_str() : "null engine"; }

void ekf_engine_destroy(EkfEngine *e)
{
    if (!e) return;
    sweep_registry_detach(e->ps_reg, e);
    e->ps_reg.reset();
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    DeviceArrays &d = e->d;
    void *ptrs[] = {d.state,     d.feat_pos,  d.feat_type, d.feat_covpos, d.feat_desc, d.feat_times_predicted, d.feat_times_matched, d.P, d.P2, d.mm_scratch, d.mm_index,        d.pred_vis, d.pred_vis_full, d.step_preds,
                    d.pred_uv,   d.pred_vis2, d.pred_uv2,  d.pred_S,      d.Hs,        d.Hf,       d.HP,
                    d.work_idx,  d.work_flag, d.plist,     d.plist_sub,   d.counts, d.shard_feat,    d.kps,      d.kdesc,
                    d.mt_valid,  d.mt_kp,     d.mt_dist,   d.matches,     d.msel,      d.mout,     d.match_of_feat,
                    d.hyp_count, d.hyp_flags, d.best_flags, d.A,          d.S,         d.nu,       d.Dinv,     d.W, d.Wf, d.G, d.LL, d.LLf, d.Tbuf, d.gates, d.cell_resp, d.cell_xy,
                    d.mHs,       d.mHf,       d.mpos,      d.mdim,        d.dx_part,   d.mask,     d.preds_out, d.sq_part, d.diag_save, d.cam_part, d.cam_save, d.HPc, d.Gc, d.Bc, d.zvec, d.yvec,
                    e->frames.kps, e->frames.desc, d.mt_xy, d.tmpl, e->img.px[0], e->img.px[1], e->img.px[2], e->img.px2[0], e->img.px2[1], e->img.px2[2], e->img.raw, e->img.seq, d.sweep_ctl, d.pu_ctr, d.Bq, d.Bexp, d.Bz, d.Lq, d.Lexp, d.Grow, d.Pdiag, d.Bstage, d.Wq, d.Gq, d.Wexp, d.Gexp, d.Wz, d.Gz};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &kv : e->pu_tables)
        if (kv.second.first) (void)hipFree(kv.second.first);
    for (auto &ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &pr : e->px_events) (void)hipEventDestroy(pr.first);
    if (e->px_mid) (void)hipEventDestroy(e->px_mid);
    for (auto &pr : e->pu_events) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (auto &pr : e->sw_events) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (e->comm && rccl_api().ok) (void)rccl_api().CommDestroy((ncclComm_t)e->comm);
    if (e->h_mirror) (void)hipHostFree(e->h_mirror);
    if (e->stream2) { (void)hipStreamSynchronize(e->stream2); (void)hipStreamDestroy(e->stream2); }
    if (e->ev_main) (void)hipEventDestroy(e->ev_main);
    if (e->ev_prefetch) (void)hipEventDestroy(e->ev_prefetch);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

static int create_impl(const EkfEngineConfig *cfg, int rank, int world, EkfEngine **out)
{
    if (!cfg || !out || cfg->max_features <= 0 || world < 1 || rank < 0 || rank >= world) return EKF_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return EKF_ERR_NO_DEVICE;
    EkfEngine *e = new (std::nothrow) EkfEngine();
    if (!e) return EKF_ERR_CAPACITY;
    e->cfg = *cfg;
    if (e->cfg.ransac_batch <= 0) e->cfg.ransac_batch = 32;
    if (cfg->device >= 0) e->device = cfg->device;
    else if (hipGetDevice(&e->device) != hipSuccess) e->device = 0;
    const EkfCamera &c = cfg->cam;
    e->cam = CamD{c.fx, c.fy, c.k1, c.k2, c.cx, c.cy, c.dx, c.dy, c.pixelErrorX, c.pixelErrorY, c.angularVisionX,
                  c.angularVisionY, c.pixelsX, c.pixelsY};
    const EkfParams &p = cfg->par;
    e->par = ParD{p.linearAccelSD, p.angularAccelSD, p.matchingCompCoefSecondBestVSFirst,
                  p.ransacThresholdPredictDistance, p.ransacAllInliersProbability, p.ransacChi2Threshold};
    if (cfg->precision < EKF_PRECISION_F64 || cfg->precision > EKF_PRECISION_AUTO) {
        delete e;
        return EKF_ERR_INVALID_ARG;
    }
    // EKF_PRECISION_AUTO: the fastest configuration that holds EVERY feature parameter within 1e-5 of the fp64 reference on the
    // maps it was measured on: the exact int8 update on an fp32-stored covariance up to
    // EKF_AUTO_F32_MAX_FEATURES features; on an fp64-stored one up to EKF_AUTO_EXACT_MAX_FEATURES (fp32 storage alone leaves 1e-5
    // on fresh maps of >= 1400 features); all-fp64 above (at N = 5000 the 38-bit digits of the exact update themselves leave
    // 2.5e-5 ... 3.4e-5 on one far feature in the third frame of a fresh map)
    if (cfg->precision == EKF_PRECISION_AUTO)
        e->cfg.precision = cfg->max_features <= EKF_AUTO_F32_MAX_FEATURES ? EKF_PRECISION_F32_EXACT
                           : (cfg->max_features <= EKF_AUTO_EXACT_MAX_FEATURES ? EKF_PRECISION_F64_EXACT : EKF_PRECISION_F64);
    e->exact = e->cfg.precision == EKF_PRECISION_F32_EXACT || e->cfg.precision == EKF_PRECISION_F64_EXACT;
    e->f32 = e->cfg.precision == EKF_PRECISION_F32 || e->cfg.precision == EKF_PRECISION_F32_EXACT;
    if ((cfg->flags & 0xff) == 1) { // EKF_DESCRIPTOR_F32_L2(cols)
        const int cols = (cfg->flags >> 8) & 0xffff;
        if (cols < 1 || cols > 1024) { delete e; return EKF_ERR_INVALID_ARG; }
        e->desc_f32 = true;
        e->desc_bytes = 4 * cols;
    } else if ((cfg->flags & 0xff) != 0) { delete e; return EKF_ERR_INVALID_ARG; }
    e->shard_rank = rank;
    e->shard_world = world;
    e->cap = cfg->max_features;
    e->ncap = 13 + 6 * e->cap;
    e->mcap = round_up(2 * e->cap, NB);
    // the exact downdate accumulates digit products in int32: five products of |d d'| <= 2^14 per level and row of k, so the sums
    // are exact only while 5 * 2^14 * m < 2^31 (kernels_pexact.hip); refuse maps whose updates could exceed that
    if (e->exact && e->mcap > PX_MAX_ROWS) {
        delete e;
        return EKF_ERR_INVALID_ARG;
    }
    e->kcap = cfg->max_keypoints > 0 ? cfg->max_keypoints : 4 * e->cap;
    e->ldP = round_up(e->ncap, LD_ALIGN);
    e->ldS = round_up(e->mcap, 128) + 128;
    auto fail = [&](hipError_t st, const char *what) {
        e->err = std::string(what) + ": " + hipGetErrorString(st);
        std::fprintf(stderr, "ekf_engine_create: %s\n", e->err.c_str());
        ekf_engine_destroy(e);
        return EKF_ERR_HIP;
    };
    hipError_t st;
    if ((st = hipSetDevice(e->device)) != hipSuccess) return fail(st, "hipSetDevice");
    (void)hipDeviceGetAttribute(&e->n_cus, hipDeviceAttributeMultiprocessorCount, e->device);
    if (e->n_cus <= 0) e->n_cus = 256;
#ifdef EKF_SWEEP_TRACE // debug builds only (scripts/build_trace_variant.sh), for scripts/contention_probe.py: "first,count" of the CU mask of the main stream
    if (const char *cm = std::getenv("EKF_PROBE_CU_MASK")) {
        int first = 0, count = 0;
        if (std::sscanf(cm, "%d,%d", &first, &count) == 2 && count > 0) {
            std::vector<uint32_t> mask((e->n_cus + 31) / 32, 0u);
            for (int c = first; c < first + count && c < e->n_cus; ++c) mask[c / 32] |= 1u << (c % 32);
            if ((st = hipExtStreamCreateWithCUMask(&e->stream, (uint32_t)mask.size(), mask.data())) != hipSuccess) return fail(st, "hipExtStreamCreateWithCUMask");
        }
    }
#endif
    if (!e->stream && (st = hipStreamCreate(&e->stream)) != hipSuccess) return fail(st, "hipStreamCreate");
    if ((st = hipStreamCreate(&e->stream2)) != hipSuccess) return fail(st, "hipStreamCreate");
    if ((st = hipEventCreateWithFlags(&e->ev_main, hipEventDisableTiming)) != hipSuccess) return fail(st, "hipEventCreate");
    if ((st = hipEventCreateWithFlags(&e->ev_prefetch, hipEventDisableTiming)) != hipSuccess) return fail(st, "hipEventCreate");
    DeviceArrays &d = e->d;
    const size_t w = e->f32 ? 4 : 8;
    const size_t cap = e->cap, mcap = e->mcap;
#define ALLOC(ptr, count)                                                        \
    if ((st = dalloc(&(ptr), (count))) != hipSuccess) return fail(st, "hipMalloc " #ptr)
    ALLOC(d.state, ST_COUNT);
    ALLOC(d.feat_pos, 6 * cap);
    ALLOC(d.feat_type, cap);
    ALLOC(d.feat_covpos, cap);
    ALLOC(d.feat_desc, (size_t)e->desc_bytes * cap);
    ALLOC(d.feat_times_predicted, cap);
    ALLOC(d.feat_times_matched, cap);
    {
        uint8_t *raw = nullptr;
        // rows of P kept here: all of them, or (sharded) camera block + the largest share of the features
        e->p_rows_cap = world == 1 ? round_up(e->ncap, LD_ALIGN)
                                   : SHARD_BASE + round_up(6 * ((e->cap + world - 1) / world), LD_ALIGN) + LD_ALIGN;
        if ((st = dalloc(&raw, (size_t)e->p_rows_cap * e->ldP * w)) != hipSuccess) return fail(st, "hipMalloc P");
        d.P = raw;
        const size_t wb = e->exact ? 8 : w; // element size of H P, its gathered rows and B = inv(L) G
        if ((st = dalloc(&raw, (size_t)mcap * e->ldP * wb)) != hipSuccess) return fail(st, "hipMalloc HP");
        d.HP = raw;
        // + one row: the row operand of a sharded downdate tile may read up to 127 columns past n
        if ((st = dalloc(&raw, (size_t)(mcap + 1) * e->ldP * wb)) != hipSuccess) return fail(st, "hipMalloc A");
        d.A = raw;
        if (e->exact) { // digit planes of B and their column scales (kernels_pexact.hip)
            e->bq_rows = round_up((int)mcap, 64) + 64;
            if ((st = dalloc(&d.Bq, (size_t)PX_S * e->bq_rows * e->ldP)) != hipSuccess) return fail(st, "hipMalloc Bq");
            if ((st = dalloc(&d.Bexp, (size_t)e->ldP)) != hipSuccess) return fail(st, "hipMalloc Bexp");
            e->px_scale_shift = 0; // engine.h: measured, and left at 0 -- one bit already costs the 1e-5 component-wise gate in one of five N = 1000 scenes
            if (const char *ev = std::getenv("EKF_PX_SCALE_SHIFT")) e->px_scale_shift = std::max(0, std::min(6, std::atoi(ev)));
            e->bz_stride = e->bq_rows / 16;
            if ((st = dalloc(&d.Bz, (size_t)(e->ldP / 32 + 8) * e->bz_stride)) != hipSuccess) return fail(st, "hipMalloc Bz");
            {   // rows of B from digit planes (chol_bplanes.h): planes of L for the sweeps that form B
                e->lq_nbk = (std::min((int)mcap, B_SWEEP_MAX) + NB - 1) / NB + 2;
                if ((st = dalloc(&d.Lq, (size_t)PX_S * e->lq_nbk * e->lq_nbk * 1024)) != hipSuccess) return fail(st, "hipMalloc Lq");
                if ((st = dalloc(&d.Lexp, (size_t)mcap + 256)) != hipSuccess) return fail(st, "hipMalloc Lexp");
                if ((st = dalloc(&d.Grow, (size_t)mcap + 256)) != hipSuccess) return fail(st, "hipMalloc Grow");
            }
            if ((int)mcap > B_SWEEP_MAX) { // B = inv(L) G on the int8 MFMA for updates above B_SWEEP_MAX rows: planes of inv(L)' and of G
                if ((st = dalloc(&d.Wq, (size_t)PX_S * e->bq_rows * (round_up((int)mcap, 128) + 128))) != hipSuccess) return fail(st, "hipMalloc Wq");
                if ((st = dalloc(&d.Gq, (size_t)PX_S * e->bq_rows * e->ldP)) != hipSuccess) return fail(st, "hipMalloc Gq");
                if ((st = dalloc(&d.Wexp, (size_t)round_up((int)mcap, 128) + 128)) != hipSuccess) return fail(st, "hipMalloc Wexp");
                if ((st = dalloc(&d.Gexp, (size_t)e->ldP)) != hipSuccess) return fail(st, "hipMalloc Gexp");
                // their tables of non-zero pieces of plane 0 (k_slice_B writes them, k_b_gemm_i8p reads them)
                if ((st = dalloc(&d.Wz, (size_t)((round_up((int)mcap, 128) + 128) / 32 + 8) * e->bz_stride)) != hipSuccess) return fail(st, "hipMalloc Wz");
                if ((st = dalloc(&d.Gz, (size_t)(e->ldP / 32 + 8) * e->bz_stride)) != hipSuccess) return fail(st, "hipMalloc Gz");
            }
            if (world > 1) { // sharded: the diagonal table and the exchange image of the planes (rows of B up to B_SWEEP_MAX)
                if ((st = dalloc(&d.Pdiag, (size_t)e->ldP)) != hipSuccess) return fail(st, "hipMalloc Pdiag");
                e->bstage_rows = e->bq_rows; // any update's rows of B
                if ((st = dalloc(&d.Bstage, (size_t)PX_S * e->bstage_rows * e->ldP)) != hipSuccess) return fail(st, "hipMalloc Bstage");
            }
        }
        if ((st = dalloc(&raw, (size_t)(mcap + 1) * e->ldP * wb)) != hipSuccess) return fail(st, "hipMalloc G");
        d.G = raw;
    }
    ALLOC(d.mm_scratch, (size_t)60 * cap + 4 * (size_t)e->ldP + 64);
    ALLOC(d.mm_index, (size_t)e->ncap + 8);
    ALLOC(d.pred_vis, cap);
    ALLOC(d.pred_vis_full, cap);
    ALLOC(d.step_preds, cap);
    ALLOC(d.pred_uv, 2 * cap);
    ALLOC(d.pred_vis2, cap);
    ALLOC(d.pred_uv2, 2 * cap);
    ALLOC(d.pred_S, 4 * cap);
    ALLOC(d.Hs, 14 * cap);
    ALLOC(d.Hf, 12 * cap);
    ALLOC(d.work_idx, cap);
    ALLOC(d.work_flag, cap);
    ALLOC(d.plist, cap);
    ALLOC(d.plist_sub, cap);
    ALLOC(d.counts, CNT_COUNT);
    ALLOC(d.shard_feat, MAX_SHARD_WORLD + 1);
    ALLOC(d.kps, (size_t)e->kcap);
    ALLOC(d.kdesc, (size_t)e->kcap * e->desc_bytes);
    ALLOC(d.mt_valid, cap);
    ALLOC(d.mt_kp, cap);
    ALLOC(d.mt_dist, cap);
    ALLOC(d.mt_xy, cap);
    ALLOC(d.tmpl, (size_t)3 * 121 * cap);
    ALLOC(d.gates, (size_t)8 * cap);
    ALLOC(d.matches, cap);
    ALLOC(d.msel, cap);
    ALLOC(d.mout, cap);
    ALLOC(d.match_of_feat, cap);
    // (a frame's first batch is ransac_batch hypotheses; the batches behind it -- a fresh map needs hundreds of hypotheses -- are
    // RANSAC_WIDE_FACTOR times wider: one workgroup per hypothesis, and 32 workgroups leave seven eighths of the chip idle)
    ALLOC(d.hyp_count, (size_t)e->cfg.ransac_batch * RANSAC_WIDE_FACTOR);
    ALLOC(d.hyp_flags, (size_t)e->cfg.ransac_batch * RANSAC_WIDE_FACTOR * mcap);
    ALLOC(d.best_flags, mcap);
    // tiles of the GEMM-shaped kernels read whole 128-wide slabs: leading dimensions and row counts carry slack
    e->ldW = round_up((int)mcap, 128) + 128;
    const size_t mw = (size_t)round_up((int)mcap, 128) + 128;
    ALLOC(d.S, mw * e->ldS);
    ALLOC(d.LL, mw * e->ldS);
    if (e->f32 && !e->exact) ALLOC(d.LLf, mw * e->ldS);
    ALLOC(d.nu, mcap);
    ALLOC(d.Dinv, mw * e->ldW);
    ALLOC(d.W, mw * e->ldW);
    ALLOC(d.Tbuf, mw * e->ldW);
    {   // flags of the persistent sweep (chol_persist.h): done / lrdy tables of (B_SWEEP_MAX / 32 + 2)^2 words each, zeroed here once
        uint8_t *raw = nullptr;
        if ((st = dalloc(&raw, (size_t)256 + sizeof(unsigned) * 2 * (B_SWEEP_MAX / NB + 2) * (B_SWEEP_MAX / NB + 2))) != hipSuccess) return fail(st, "hipMalloc sweep_ctl");
        d.sweep_ctl = raw;
    }
    if (e->f32 && !e->exact) ALLOC(d.Wf, mw * e->ldW);
    ALLOC(d.mHs, 14 * cap);
    ALLOC(d.mHf, 12 * cap);
    ALLOC(d.mpos, cap);
    ALLOC(d.mdim, cap);
    ALLOC(d.dx_part, (size_t)DX_SPLIT * e->ldP + 8); // + the quaternion before the update (k_apply_normalize)
    ALLOC(d.sq_part, (size_t)DX_SPLIT * e->ldP);
    ALLOC(d.diag_save, (size_t)e->ldP);
    ALLOC(d.cam_part, (size_t)DX_SPLIT * 13 * e->ldP);
    ALLOC(d.cam_save, (size_t)13 * e->ldP);
    ALLOC(d.HPc, (size_t)mcap * 16);
    ALLOC(d.Gc, mw * 16);
    ALLOC(d.Bc, mw * 16);
    ALLOC(d.zvec, mw);
    ALLOC(d.yvec, mw);
    ALLOC(d.mask, mcap);
    ALLOC(d.preds_out, cap);
#undef ALLOC
    for (auto &ev : e->ev)
        if ((st = hipEventCreate(&ev)) != hipSuccess) return fail(st, "hipEventCreate");
    e->h_counts.assign(CNT_COUNT, 0);
    {   // host mirror of the counter block (optional: without it read_counts falls back to memcpy + synchronise)
        void *hp = nullptr;
        if (hipHostMalloc(&hp, 64 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            std::memset(hp, 0, 64 * sizeof(int));
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                e->h_mirror = (int *)hp;
                e->d_mirror = (int *)dp;
            } else {
                (void)hipHostFree(hp);
            }
        }
    }
    e->shard_feat_begin.assign(world + 1, 0);
    e->exchange_hook = exchange_rows;
    e->ps_reg = sweep_registry_attach(e->device); // persistent sweeps of the engines on one device are ordered (engine.h)
    *out = e;
    return EKF_OK;
}

int ekf_engine_create(const EkfEngineConfig *cfg, EkfEngine **out) { return create_impl(cfg, 0, 1, out); }

int ekf_engine_create_sharded(const EkfEngineConfig *cfg, int rank, int world, EkfEngine **out)
{
    if (world > MAX_SHARD_WORLD) return EKF_ERR_INVALID_ARG;
    return create_impl(cfg, rank, world, out);
}

int ekf_set_exchange(EkfEngine *e, EkfExchangeFn fn, void *user)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    e->xchg = fn;
    e->xchg_user = user;
    return EKF_OK;
}

int ekf_comm_unique_id(uint8_t id[EKF_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == EKF_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return EKF_ERR_INVALID_ARG;
    RcclApi &api = rccl_api();
    if (!api.ok) return EKF_ERR_COMM;
    ncclUniqueId u;
    if (api.GetUniqueId(&u) != ncclSuccess) return EKF_ERR_COMM;
    std::memcpy(id, &u, EKF_COMM_ID_BYTES);
    return EKF_OK;
}

int ekf_comm_init(EkfEngine *e, const uint8_t id[EKF_COMM_ID_BYTES])
{
    if (!e || !id) return EKF_ERR_INVALID_ARG;
    RcclApi &api = rccl_api();
    if (!api.ok) {
        e->err = "librccl.so.1 could not be loaded";
        return EKF_ERR_COMM;
    }
    HIPCHK(hipSetDevice(e->device));
    if (e->comm) {
        (void)api.CommDestroy((ncclComm_t)e->comm);
        e->comm = nullptr;
    }
    ncclUniqueId u;
    std::memcpy(&u, id, EKF_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    const ncclResult_t r = api.CommInitRank(&c, e->shard_world, u, e->shard_rank);
    if (r != ncclSuccess) {
        e->err = std::string("ncclCommInitRank: ") + (api.GetErrorString ? api.GetErrorString(r) : "error");
        return EKF_ERR_COMM;
    }
    e->comm = c;
    return EKF_OK;
}

// Completes a replicated per-feature table: rank r owns rows [rb[r], rb[r+1]) of row_bytes each and has just written
// them (on the engine's stream).  In-engine transport: every pair of ranks exchanges its blocks directly, enqueued on the
// same stream; otherwise the host's callback (which needs the stream idle).
static int exchange_rows(EkfEngine *e, int what, void *base, size_t row_bytes, const std::vector<int32_t> &rb, const char *name)
{
    const int world = e->shard_world, me = e->shard_rank;
    if (what == EKF_XCHG_BPLANES) e->xchg_bytes_planes += (long long)((size_t)(rb[world] - rb[0] - (rb[me + 1] - rb[me])) * row_bytes);
    if (e->comm && !e->xchg) { // a callback installed after ekf_comm_init takes over (ranks that fell back by consensus)
        RcclApi &api = rccl_api();
        ncclComm_t c = (ncclComm_t)e->comm;
        uint8_t *b = (uint8_t *)base;
        const size_t my_lo = (size_t)rb[me] * row_bytes, my_n = (size_t)(rb[me + 1] - rb[me]) * row_bytes;
        ncclResult_t r = api.GroupStart();
        for (int p = 0; p < world && r == ncclSuccess; ++p) {
            if (p == me) continue;
            const size_t lo = (size_t)rb[p] * row_bytes, cnt = (size_t)(rb[p + 1] - rb[p]) * row_bytes;
            if (my_n > 0) r = api.Send(b + my_lo, my_n, ncclChar, p, c, e->stream);
            if (r == ncclSuccess && cnt > 0) r = api.Recv(b + lo, cnt, ncclChar, p, c, e->stream);
        }
        const ncclResult_t r2 = api.GroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess) {
            e->err = std::string("RCCL exchange of ") + name + " failed";
            return EKF_ERR_COMM;
        }
        return EKF_OK;
    }
    if (!e->xchg) {
        e->err = "sharded engine without a transport (ekf_comm_init or ekf_set_exchange)";
        return EKF_ERR_COMM;
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->xchg(e->xchg_user, what, base, row_bytes, rb.data(), world, me)) {
        e->err = std::string("exchange of ") + name + " failed";
        return EKF_ERR_COMM;
    }
    return EKF_OK;
}

int ekf_shard_info(const EkfEngine *e, int *rank, int *world, int *row_begin, int *row_end)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    if (rank) *rank = e->shard_rank;
    if (world) *world = e->shard_world;
    if (row_begin) *row_begin = e->shard_rank == 0 ? 0 : e->rm.r0;
    if (row_end) *row_end = e->rm.r1;
    return EKF_OK;
}

int ekf_shard_counters(EkfEngine *e, int64_t *plane_bytes_received, int32_t *own_columns_begin, int32_t *own_columns_end)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    if (plane_bytes_received) *plane_bytes_received = e->xchg_bytes_planes;
    const int n_pad = round_up(e->n, LD_ALIGN), W = e->shard_world, r = e->shard_rank;
    // the shares the last update actually used (they end at the ownership boundaries on the by-symmetry route, at multiples of
    // 32 on the inverse + GEMM route); before the first update: the rounded convention
    auto col = [&](int k) {
        if ((int)e->last_col_rb.size() == W + 1) return (int)e->last_col_rb[k];
        return k <= 0 ? 0 : (k >= W ? n_pad : std::min(n_pad, round_up(e->shard_row_begin.size() > (size_t)k ? e->shard_row_begin[k] : e->n, NB)));
    };
    if (own_columns_begin) *own_columns_begin = col(r);
    if (own_columns_end) *own_columns_end = col(r + 1);
    return EKF_OK;
}

int ekf_device_copy(EkfEngine *e, void *dst, const void *src, size_t bytes)
{
    if (!e || (bytes > 0 && (!dst || !src))) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    if (bytes > 0) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    return EKF_OK;
}

// feature partition of a sharded engine (same arithmetic as ekf_shard_rows) and the row map that follows from it
static void refresh_row_map(EkfEngine *e)
{
    const int W = e->shard_world, N = e->N;
    if (W == 1) {
        e->rm = RowMap{13, e->n, 13};
        e->shard_feat_begin.assign(2, 0);
        e->shard_feat_begin[1] = N;
        e->shard_row_begin = {0, e->n};
        return;
    }
    const int per = N / W, extra = N % W;
    for (int r = 0; r <= W; ++r) e->shard_feat_begin[r] = r * per + (r < extra ? r : extra);
    const int f0 = e->shard_feat_begin[e->shard_rank], f1 = e->shard_feat_begin[e->shard_rank + 1];
    auto row_of = [&](int f) { return f < N ? e->h_covpos[f] : e->n; };
    e->rm = RowMap{row_of(f0), row_of(f1), SHARD_BASE};
    e->shard_row_begin.assign(W + 1, 0);
    for (int r = 1; r <= W; ++r) e->shard_row_begin[r] = row_of(e->shard_feat_begin[r]);
    (void)hipMemcpyAsync(e->d.shard_feat, e->shard_feat_begin.data(), (size_t)(W + 1) * sizeof(int), hipMemcpyHostToDevice, e->stream);
    (void)hipStreamSynchronize(e->stream); // the host vector may be reassigned before the copy would otherwise run
}

#define NOT_WHEN_SHARDED(e)                                                \
    if ((e)->shard_world > 1) {                                            \
        (e)->err = "map management is not available on a sharded engine";  \
        return EKF_ERR_INVALID_ARG;                                        \
    }

int ekf_get_map_features(EkfEngine *e, uint8_t *desc32, uint32_t *times_predicted, uint32_t *times_matched)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t N = (size_t)e->N;
    if (N == 0) return EKF_OK;
    if (desc32) HIPCHK(hipMemcpy(desc32, e->d.feat_desc, N * e->desc_bytes, hipMemcpyDeviceToHost));
    if (times_predicted) HIPCHK(hipMemcpy(times_predicted, e->d.feat_times_predicted, N * 4, hipMemcpyDeviceToHost));
    if (times_matched) HIPCHK(hipMemcpy(times_matched, e->d.feat_times_matched, N * 4, hipMemcpyDeviceToHost));
    return EKF_OK;
}

int ekf_state_dim(const EkfEngine *e) { return e ? e->n : 0; }
int ekf_descriptor_bytes(const EkfEngine *e) { return e ? e->desc_bytes : 0; }
int ekf_get_precision(const EkfEngine *e) { return e ? e->cfg.precision : -1; }
int ekf_num_features(const EkfEngine *e) { return e ? e->N : 0; }

static int finish_update(EkfEngine *e);

// With ekf_set_async_errors a step does not read the error flag of its last update back; the flag is sticky on the
// device and is reported (and cleared) by the next step, by ekf_synchronize or by ekf_get_state, whichever comes first.
static int take_pending_error(EkfEngine *e) { return e->async_errors ? finish_update(e) : EKF_OK; }

int ekf_synchronize(EkfEngine *e)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    return take_pending_error(e);
}

// ------------------------------------------------------------------------------------------- state transfer
int ekf_set_state(EkfEngine *e, const double x13[13], int n_features, const double *feature_pos,
                  const int32_t *feature_type, const uint8_t *desc32, const double *P)
{
    if (!e || !x13 || n_features < 0 || (n_features > 0 && !feature_pos)) return EKF_ERR_INVALID_ARG;
    if (n_features > e->cap) return EKF_ERR_CAPACITY;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    std::vector<int> type(n_features), covpos(n_features);
    int pos = 13;
    for (int i = 0; i < n_features; ++i) {
        const int t = feature_type ? feature_type[i] : EKF_FEATURE_INVERSE_DEPTH;
        if (t != EKF_FEATURE_DEPTH && t != EKF_FEATURE_INVERSE_DEPTH) return EKF_ERR_INVALID_ARG;
        type[i] = t;
        covpos[i] = pos;
        pos += (t == EKF_FEATURE_INVERSE_DEPTH) ? 6 : 3;
    }
    const int n = pos;
    double st[ST_R + 9];
    std::memcpy(st, x13, 13 * sizeof(double));
    {
        const double r = x13[3], x = x13[4], y = x13[5], z = x13[6];
        double *M = st + ST_R;
        M[0] = r * r + x * x - y * y - z * z; M[1] = 2 * (x * y - r * z); M[2] = 2 * (z * x + r * y);
        M[3] = 2 * (x * y + r * z); M[4] = r * r - x * x + y * y - z * z; M[5] = 2 * (y * z - r * x);
        M[6] = 2 * (z * x - r * y); M[7] = 2 * (y * z + r * x); M[8] = r * r - x * x - y * y + z * z;
    }
    HIPCHK(hipMemcpy(e->d.state, st, sizeof(st), hipMemcpyHostToDevice));
    if (n_features > 0) {
        HIPCHK(hipMemcpy(e->d.feat_pos, feature_pos, (size_t)6 * n_features * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d.feat_type, type.data(), (size_t)n_features * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d.feat_covpos, covpos.data(), (size_t)n_features * sizeof(int), hipMemcpyHostToDevice));
        if (desc32)
            HIPCHK(hipMemcpy(e->d.feat_desc, desc32, (size_t)n_features * e->desc_bytes, hipMemcpyHostToDevice));
        else
            HIPCHK(hipMemset(e->d.feat_desc, 0, (size_t)n_features * e->desc_bytes));
    }
    e->N = n_features;
    e->n = n;
    e->n_pred = 0;
    e->n_gates = 0;
    e->h_type = type;
    e->h_covpos = covpos;
    refresh_row_map(e);
    if (e->shard_world > 1 && SHARD_BASE + (e->rm.r1 - e->rm.r0) > e->p_rows_cap) return EKF_ERR_CAPACITY;
    if (P) {
        const size_t w = e->f32 ? 4 : 8;
        HIPCHK(hipMemset(e->d.P, 0, (size_t)e->p_rows_cap * e->ldP * w));
        const bool sharded = e->shard_world > 1;
        // row blocks to upload: (first global row, count, first local row)
        const int blocks[2][3] = {{0, sharded ? 13 : n, 0}, {e->rm.r0, sharded ? e->rm.r1 - e->rm.r0 : 0, e->rm.base}};
        std::vector<float> tmp;
        std::vector<double> sym;
        for (const auto &b : blocks) {
            const int g0 = b[0], cnt = b[1], l0 = b[2];
            if (cnt <= 0) continue;
            const double *src = P + (size_t)g0 * n;
            if (sharded) { // no cross-rank averaging pass exists: symmetrise on the way in, in T arithmetic
                sym.resize((size_t)cnt * n);
                for (int i = 0; i < cnt; ++i)
                    for (int j = 0; j < n; ++j) {
                        const double a = P[(size_t)(g0 + i) * n + j], c = P[(size_t)j * n + g0 + i];
                        sym[(size_t)i * n + j] = e->f32 ? (double)(0.5f * (float)a + 0.5f * (float)c) : 0.5 * a + 0.5 * c;
                    }
                src = sym.data();
            }
            uint8_t *dst = (uint8_t *)e->d.P + (size_t)l0 * e->ldP * w;
            if (e->f32) {
                tmp.resize((size_t)cnt * n);
                for (size_t i = 0; i < (size_t)cnt * n; ++i) tmp[i] = (float)src[i];
                HIPCHK(hipMemcpy2D(dst, (size_t)e->ldP * 4, tmp.data(), (size_t)n * 4, (size_t)n * 4, cnt, hipMemcpyHostToDevice));
            } else {
                HIPCHK(hipMemcpy2D(dst, (size_t)e->ldP * 8, src, (size_t)n * 8, (size_t)n * 8, cnt, hipMemcpyHostToDevice));
            }
        }
        e->p_exact_sym = sharded;
    }
    HIPCHK(hipMemset(e->d.pred_vis, 0, (size_t)e->cap * sizeof(int)));
    HIPCHK(hipMemset(e->d.pred_vis_full, 0, (size_t)e->cap * sizeof(int)));
    e->n_step_preds = 0;
    HIPCHK(hipMemset(e->d.feat_times_predicted, 0, (size_t)e->cap * sizeof(unsigned)));
    HIPCHK(hipMemset(e->d.feat_times_matched, 0, (size_t)e->cap * sizeof(unsigned)));
    return EKF_OK;
}

int ekf_get_state(EkfEngine *e, double x13[13], double *feature_pos, double *P)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const int pending = take_pending_error(e); // the state is still copied out: the caller sees what the failed update left
    if (x13) HIPCHK(hipMemcpy(x13, e->d.state, 13 * sizeof(double), hipMemcpyDeviceToHost));
    if (feature_pos && e->N > 0)
        HIPCHK(hipMemcpy(feature_pos, e->d.feat_pos, (size_t)6 * e->N * sizeof(double), hipMemcpyDeviceToHost));
    if (P) {
        const int n = e->n;
        const size_t w = e->f32 ? 4 : 8;
        const bool sharded = e->shard_world > 1;
        const int blocks[2][3] = {{0, sharded ? 13 : n, 0}, {e->rm.r0, sharded ? e->rm.r1 - e->rm.r0 : 0, e->rm.base}};
        std::vector<float> tmp;
        for (const auto &b : blocks) { // a sharded engine fills the rows it holds and leaves the others untouched
            const int g0 = b[0], cnt = b[1], l0 = b[2];
            if (cnt <= 0) continue;
            const uint8_t *src = (const uint8_t *)e->d.P + (size_t)l0 * e->ldP * w;
            double *dst = P + (size_t)g0 * n;
            if (e->f32) {
                tmp.resize((size_t)cnt * n);
                HIPCHK(hipMemcpy2D(tmp.data(), (size_t)n * 4, src, (size_t)e->ldP * 4, (size_t)n * 4, cnt, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < (size_t)cnt * n; ++i) dst[i] = (double)tmp[i];
            } else {
                HIPCHK(hipMemcpy2D(dst, (size_t)n * 8, src, (size_t)e->ldP * 8, (size_t)n * 8, cnt, hipMemcpyDeviceToHost));
            }
        }
    }
    return pending;
}


// ------------------------------------------------------------------------------------------ map management
static int check_async(EkfEngine *e);
static int dims_of(int type) { return type == EKF_FEATURE_INVERSE_DEPTH ? 6 : 3; }

int ekf_reset(EkfEngine *e)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    // initState + initCovariance, EKF/CommonFunctions.cpp:39-80
    double x[13] = {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, EKF_EPSILON, EKF_EPSILON, EKF_EPSILON};
    double P[169];
    std::memset(P, 0, sizeof(P));
    for (int i = 0; i < 7; ++i) P[i * 13 + i] = EKF_EPSILON;
    for (int i = 0; i < 3; ++i) {
        P[(i + 7) * 13 + i + 7] = e->cfg.par.initLinearAccelSD * e->cfg.par.initLinearAccelSD;
        P[(i + 10) * 13 + i + 10] = e->cfg.par.initAngularAccelSD * e->cfg.par.initAngularAccelSD;
    }
    return ekf_set_state(e, x, 0, nullptr, nullptr, nullptr, P);
}

int ekf_get_camera_covariance(EkfEngine *e, double P13[169])
{
    if (!e || !P13) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->f32) {
        float tmp[169];
        HIPCHK(hipMemcpy2D(tmp, 13 * 4, e->d.P, (size_t)e->ldP * 4, 13 * 4, 13, hipMemcpyDeviceToHost));
        for (int i = 0; i < 169; ++i) P13[i] = (double)tmp[i];
    } else {
        HIPCHK(hipMemcpy2D(P13, 13 * 8, e->d.P, (size_t)e->ldP * 8, 13 * 8, 13, hipMemcpyDeviceToHost));
    }
    return EKF_OK;
}

int ekf_get_unseen_features(EkfEngine *e, int32_t *feat_idx, int *count)
{
    if (!e || !count) return EKF_ERR_INVALID_ARG;
    *count = 0;
    if (e->N == 0) return EKF_OK;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    std::vector<int> vis(e->N);
    HIPCHK(hipMemcpy(vis.data(), e->d.pred_vis_full, (size_t)e->N * sizeof(int), hipMemcpyDeviceToHost));
    int k = 0;
    for (int i = 0; i < e->N; ++i)
        if (!vis[i]) {
            if (feat_idx) feat_idx[k] = i;
            ++k;
        }
    *count = k;
    return EKF_OK;
}

int ekf_keep_step_predictions(EkfEngine *e, int on)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    e->keep_step_preds = on != 0;
    if (!on) e->n_step_preds = 0;
    return EKF_OK;
}

int ekf_get_step_predictions(EkfEngine *e, EkfPrediction *preds, int *n_preds)
{
    if (!e || !n_preds) return EKF_ERR_INVALID_ARG;
    *n_preds = e->n_step_preds;
    if (!preds || e->n_step_preds == 0) return EKF_OK;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(preds, e->d.step_preds, (size_t)e->n_step_preds * sizeof(EkfPrediction), hipMemcpyDeviceToHost));
    return EKF_OK;
}

int ekf_get_feature_layout(EkfEngine *e, int32_t *type, int32_t *covpos)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    for (int i = 0; i < e->N; ++i) {
        if (type) type[i] = e->h_type[i];
        if (covpos) covpos[i] = e->h_covpos[i];
    }
    return EKF_OK;
}

int ekf_add_features(EkfEngine *e, const double *uv, const uint8_t *desc32, int count)
{
    if (!e || count < 0 || (count > 0 && !uv)) return EKF_ERR_INVALID_ARG;
    if (count == 0) return EKF_OK;
    NOT_WHEN_SHARDED(e)
    if (e->N + count > e->cap) return EKF_ERR_CAPACITY;
    HIPCHK(hipSetDevice(e->device));
    double *d_uv = e->d.mm_scratch, *d_Jpo = d_uv + 2 * (size_t)e->cap, *d_Jhr = d_Jpo + 42 * (size_t)count;
    HIPCHK(hipMemcpyAsync(d_uv, uv, (size_t)2 * count * sizeof(double), hipMemcpyHostToDevice, e->stream));
    uint8_t *dd = e->d.feat_desc + (size_t)e->N * e->desc_bytes;
    if (desc32) HIPCHK(hipMemcpyAsync(dd, desc32, (size_t)count * e->desc_bytes, hipMemcpyHostToDevice, e->stream));
    else HIPCHK(hipMemsetAsync(dd, 0, (size_t)count * e->desc_bytes, e->stream));
    HIPCHK(hipMemsetAsync(e->d.feat_times_predicted + e->N, 0, (size_t)count * 4, e->stream));
    HIPCHK(hipMemsetAsync(e->d.feat_times_matched + e->N, 0, (size_t)count * 4, e->stream));
    launch_add_features(e, d_uv, count, d_Jpo, d_Jhr);
    for (int j = 0; j < count; ++j) {
        e->h_type.push_back(EKF_FEATURE_INVERSE_DEPTH);
        e->h_covpos.push_back(e->n + 6 * j);
    }
    e->N += count;
    e->n += 6 * count;
    e->n_pred = 0;
    refresh_row_map(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    return check_async(e);
}

// drop the listed rows (sorted row flags) from P and the listed features from the SoA arrays
static int compact_map(EkfEngine *e, const std::vector<uint8_t> &drop_feature, const std::vector<uint8_t> &drop_row)
{
    NOT_WHEN_SHARDED(e)
    const int N = e->N, n = e->n;
    std::vector<int> new2old;
    for (int i = 0; i < n; ++i)
        if (!drop_row[i]) new2old.push_back(i);
    const int n_new = (int)new2old.size();
    if (!e->d.P2) {
        const size_t bytes = (size_t)round_up(e->ncap, LD_ALIGN) * e->ldP * (e->f32 ? 4 : 8);
        HIPCHK(hipMalloc(&e->d.P2, bytes));
        HIPCHK(hipMemset(e->d.P2, 0, bytes));
    }
    HIPCHK(hipMemcpyAsync(e->d.mm_index, new2old.data(), (size_t)n_new * sizeof(int), hipMemcpyHostToDevice, e->stream));
    launch_compact_P(e, n_new, e->d.mm_index);
    // SoA arrays: small, compacted through the host
    std::vector<double> pos(6 * (size_t)N);
    std::vector<uint8_t> desc((size_t)N * e->desc_bytes);
    std::vector<unsigned> tp(N), tm(N);
    std::vector<uint8_t> tmpl((size_t)N * 363);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(tmpl.data(), e->d.tmpl, tmpl.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pos.data(), e->d.feat_pos, pos.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(desc.data(), e->d.feat_desc, desc.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tp.data(), e->d.feat_times_predicted, (size_t)N * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tm.data(), e->d.feat_times_matched, (size_t)N * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(e->h_type.data(), e->d.feat_type, (size_t)N * 4, hipMemcpyDeviceToHost)); // conversions change types
    int w = 0, p = 13;
    for (int i = 0; i < N; ++i) {
        if (drop_feature[i]) continue;
        std::memmove(&pos[6 * (size_t)w], &pos[6 * (size_t)i], 6 * sizeof(double));
        std::memmove(&desc[(size_t)w * e->desc_bytes], &desc[(size_t)i * e->desc_bytes], e->desc_bytes);
        std::memmove(&tmpl[(size_t)w * 363], &tmpl[(size_t)i * 363], 363);
        tp[w] = tp[i];
        tm[w] = tm[i];
        e->h_type[w] = e->h_type[i];
        e->h_covpos[w] = p;
        p += dims_of(e->h_type[w]);
        ++w;
    }
    e->h_type.resize(w);
    e->h_covpos.resize(w);
    if (p != n_new) {
        e->err = "map compaction: layout mismatch";
        return EKF_ERR_INVALID_ARG;
    }
    if (w > 0) {
        HIPCHK(hipMemcpy(e->d.feat_pos, pos.data(), (size_t)6 * w * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d.feat_desc, desc.data(), (size_t)w * e->desc_bytes, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d.tmpl, tmpl.data(), (size_t)w * 363, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d.feat_times_predicted, tp.data(), (size_t)w * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d.feat_times_matched, tm.data(), (size_t)w * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d.feat_type, e->h_type.data(), (size_t)w * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d.feat_covpos, e->h_covpos.data(), (size_t)w * 4, hipMemcpyHostToDevice));
    }
    e->N = w;
    e->n = n_new;
    refresh_row_map(e);
    e->n_pred = 0;
    return check_async(e);
}

int ekf_remove_features(EkfEngine *e, const int32_t *feat_idx, int count)
{
    if (!e || count < 0 || (count > 0 && !feat_idx)) return EKF_ERR_INVALID_ARG;
    if (count == 0) return EKF_OK;
    HIPCHK(hipSetDevice(e->device));
    std::vector<uint8_t> df(e->N, 0), dr(e->n, 0);
    for (int i = 0; i < count; ++i) {
        const int f = feat_idx[i];
        if (f < 0 || f >= e->N || (i > 0 && feat_idx[i] <= feat_idx[i - 1])) return EKF_ERR_INVALID_ARG;
        df[f] = 1;
        for (int k = 0; k < dims_of(e->h_type[f]); ++k) dr[e->h_covpos[f] + k] = 1;
    }
    return compact_map(e, df, dr);
}

int ekf_remove_bad_features(EkfEngine *e, int *n_removed)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const int N = e->N;
    std::vector<unsigned> tp(N), tm(N);
    if (N > 0) {
        HIPCHK(hipMemcpy(tp.data(), e->d.feat_times_predicted, (size_t)N * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(tm.data(), e->d.feat_times_matched, (size_t)N * 4, hipMemcpyDeviceToHost));
    }
    std::vector<int32_t> idx;
    for (int i = 0; i < N; ++i) { // EKF/MapManagement.cpp:291-302 (float ratio, 0/0 = NaN keeps the feature)
        const float pct = static_cast<float>(tm[i]) / static_cast<float>(tp[i]);
        if (pct < e->cfg.par.goodFeatureMatchingPercent) idx.push_back(i);
    }
    if (n_removed) *n_removed = (int)idx.size();
    return ekf_remove_features(e, idx.data(), (int)idx.size());
}

int ekf_convert_inverse_depth_to_depth(EkfEngine *e, int *converted_index)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    if (converted_index) *converted_index = -1;
    if (e->N == 0) return EKF_OK;
    NOT_WHEN_SHARDED(e)
    HIPCHK(hipSetDevice(e->device));
    double *d_li = e->d.mm_scratch;
    launch_linearity(e, d_li);
    std::vector<double> li(e->N);
    HIPCHK(hipMemcpyAsync(li.data(), d_li, (size_t)e->N * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    int fi = -1;
    for (int i = 0; i < e->N; ++i) // first inverse-depth feature in map order below the threshold, one per call (:494-521)
        if (e->h_type[i] == EKF_FEATURE_INVERSE_DEPTH && li[i] < e->cfg.par.inverseDepthLinearityIndexThreshold) {
            fi = i;
            break;
        }
    if (fi < 0) return EKF_OK;
    const int pos = e->h_covpos[fi];
    double *d_J = e->d.mm_scratch + e->cap, *d_T3 = d_J + 32;
    launch_convert(e, fi, pos, d_J, d_T3);
    e->h_type[fi] = EKF_FEATURE_DEPTH;
    std::vector<uint8_t> df(e->N, 0), dr(e->n, 0);
    dr[pos + 3] = dr[pos + 4] = dr[pos + 5] = 1;
    if (converted_index) *converted_index = fi;
    return compact_map(e, df, dr);
}

// ----------------------------------------------------------------------------------------------- utilities
// The device counter block is mirrored into a page of host memory the GPU can write (hipHostMallocMapped, coherent):
// a one-wavefront kernel copies the 16 ints and then a sequence number, the host polls the sequence number.  The
// per-frame control flow needs ~6 of these round trips (how many predictions / matches / inliers / rescued decide the
// next launches); a memcpy + stream synchronise costs ~20 us each, the poll a few.
constexpr int POLL_SECONDS = 20; // bound of a host poll in wall time (not in iterations: their rate depends on the host); then the plain path

static int read_counts(EkfEngine *e)
{
    if (e->h_mirror) {
        const int seq = ++e->mirror_seq;
        launch_publish_counts(e, e->d_mirror, seq);
        volatile int *m = e->h_mirror;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(POLL_SECONDS);
        for (long spin = 0;; ++spin) {
            if (m[CNT_COUNT] == seq) {
                std::atomic_thread_fence(std::memory_order_acquire);
                for (int i = 0; i < CNT_COUNT; ++i) e->h_counts[i] = m[i];
                return EKF_OK;
            }
            if ((spin & 0xfffff) == 0xfffff) { // every ~1e6 polls: the stream drained without the write, or the time bound: fall back
                if (hipStreamQuery(e->stream) == hipSuccess && m[CNT_COUNT] != seq) break;
                if (std::chrono::steady_clock::now() > t_end) break;
            }
        }
    }
    HIPCHK(hipMemcpyAsync(e->h_counts.data(), e->d.counts, CNT_COUNT * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return EKF_OK;
}

// read_counts for a stage whose last kernel published the counter block itself (publish_counts_block, sequence number
// taken with next_publish_seq before that launch): poll only
static int next_publish_seq(EkfEngine *e) { return e->h_mirror ? ++e->mirror_seq : 0; }

static int wait_counts(EkfEngine *e, int seq)
{
    if (seq <= 0) return read_counts(e);
    volatile int *m = e->h_mirror;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(POLL_SECONDS);
    for (long spin = 0;; ++spin) {
        if (m[CNT_COUNT] == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            for (int i = 0; i < CNT_COUNT; ++i) e->h_counts[i] = m[i];
            return EKF_OK;
        }
        if ((spin & 0xfffff) == 0xfffff &&
            ((hipStreamQuery(e->stream) == hipSuccess && m[CNT_COUNT] != seq) || std::chrono::steady_clock::now() > t_end)) break;
#if defined(__x86_64__)
        __builtin_ia32_pause(); // the caller's thread polls one cache line; leave the core's other thread its issue slots
#endif
    }
    return read_counts(e); // the stream drained without the write (or the poll gave up): the plain path
}

static int check_async(EkfEngine *e)
{
    HIPCHK(hipGetLastError());
    return EKF_OK;
}

// ---------------------------------------------------------------------------------------------------- stages
int ekf_predict(EkfEngine *e)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    launch_predict(e);
    return check_async(e);
}

// Sharded filter: the FULL exchange of the H.P table (2N x ldP: 192 MB at N = 2000, 1.2 GB at N = 5000 in fp32) and of its
// fp64 camera columns -- what rounds 1-2 did after every prediction.  EKF::step no longer needs it (see `lean` below); the
// stateless stage calls of the ABI (ekf_ransac / ekf_update / ... on a caller-ordered match list) still do.
static int complete_hp_table(EkfEngine *e)
{
    if (e->shard_world <= 1 || e->hp_complete) return EKF_OK;
    int rc;
    std::vector<int32_t> rb(e->shard_world + 1);
    for (int r = 0; r <= e->shard_world; ++r) rb[r] = 2 * e->shard_feat_begin[r];
    if ((rc = exchange_rows(e, EKF_XCHG_HP, e->d.HP, (size_t)e->ldP * hp_elem_bytes(e), rb, "the H.P row blocks"))) return rc;
    if (e->f32 && !e->exact && (rc = exchange_rows(e, EKF_XCHG_HPC, e->d.HPc, 16 * sizeof(double), rb, "the fp64 camera columns of H.P"))) return rc;
    e->hp_complete = true;
    return EKF_OK;
}

// device-side core of predictCameraMeasurements; leaves the count in h_counts.
// lean (sharded EKF::step): only the 2x2 innovation blocks S_i travel after a prediction (32 bytes per feature; every rank
// gates and matches all predictions).  The H.P rows stay with their owners until somebody consumes them: the rows of a
// RANSAC batch's hypotheses (ransac_dev) and the rows of the matches an update selects (lean_gather_exchange) -- 2M of
// the 2N rows, and for the outlier re-prediction nothing but the S_i (missing 2a / EKF.cpp:473).
static int predict_measurements_dev(EkfEngine *e, const int *d_idx, int count, int *n_out, bool count_predicted = false,
                                    bool lean = false)
{
    launch_predict_features(e, d_idx, d_idx ? count : e->N, false);
    const bool slots = lean && !d_idx && e->shard_world > 1; // sharded step: where each rank's run of the predicted list starts (matching by owner)
    if (slots) launch_shard_bounds_idx(e, e->d.plist, e->d.counts + CNT_NPRED);
    int rc = read_counts(e);
    if (rc) return rc;
    const int np = e->h_counts[d_idx ? CNT_NPRED_SUB : CNT_NPRED];
    if (slots) {
        e->slot_rb.assign(e->shard_world + 1, 0);
        for (int r = 0; r <= e->shard_world; ++r) e->slot_rb[r] = e->h_counts[CNT_SHARD0 + r];
    } else if (!d_idx) e->slot_rb.clear();
    launch_hp_rows(e, d_idx ? e->d.plist_sub : e->d.plist, np, count_predicted);
    if (!d_idx) e->n_pred = np;
    *n_out = np;
    if (e->shard_world > 1 && np > 0) {
        // every rank wrote the H.P rows and S_i of the features it owns
        e->hp_complete = false;
        if (!lean && (rc = complete_hp_table(e))) return rc;
        std::vector<int32_t> rb(e->shard_world + 1);
        for (int r = 0; r <= e->shard_world; ++r) rb[r] = e->shard_feat_begin[r];
        if ((rc = exchange_rows(e, EKF_XCHG_PRED_S, e->d.pred_S, 4 * sizeof(double), rb, "the innovation covariance blocks"))) return rc;
    }
    if (!d_idx && e->keep_step_preds) { // EKF.cpp:294-305 draws the step's predictions as they were BEFORE the updates
        launch_pack_predictions(e, e->d.plist, np, e->d.step_preds);
        e->n_step_preds = np;
    }
    if (!d_idx && e->img.valid) { // image mode: keep the gates of this prediction for detectNewImageFeatures' mask
        launch_gate_snapshot(e, np);
        e->n_gates = np;
    }
    return check_async(e);
}

static int download_predictions(EkfEngine *e, const int *d_list, int np, EkfPrediction *preds, double *Hs, double *Hf)
{
    if (np <= 0) return EKF_OK;
    const int N = e->N;
    std::vector<int> list(np);
    std::vector<double> uv(2 * (size_t)N), S(4 * (size_t)N), hs(14 * (size_t)N), hf(12 * (size_t)N);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(list.data(), d_list, np * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(uv.data(), e->d.pred_uv, uv.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(S.data(), e->d.pred_S, S.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hs.data(), e->d.Hs, hs.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hf.data(), e->d.Hf, hf.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < np; ++k) {
        const int fi = list[k];
        if (preds) {
            preds[k].featureIndex = fi;
            preds[k]._pad = 0;
            preds[k].imagePos[0] = uv[2 * fi];
            preds[k].imagePos[1] = uv[2 * fi + 1];
            for (int i = 0; i < 4; ++i) preds[k].covarianceMatrix[i] = S[4 * fi + i];
        }
        if (Hs)
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 13; ++c) Hs[(size_t)26 * k + r * 13 + c] = c < 7 ? hs[14 * fi + r * 7 + c] : 0.0;
        if (Hf)
            for (int i = 0; i < 12; ++i) Hf[(size_t)12 * k + i] = hf[12 * fi + i];
    }
    return EKF_OK;
}

int ekf_predict_measurements(EkfEngine *e, const int32_t *feat_idx, int count, EkfPrediction *preds, int *n_preds,
                             double *Hs, double *Hf)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    const bool sub = feat_idx && count > 0;
    if (sub) {
        if (count > e->cap) return EKF_ERR_CAPACITY;
        for (int i = 0; i < count; ++i)
            if (feat_idx[i] < 0 || feat_idx[i] >= e->N) return EKF_ERR_INVALID_ARG;
        HIPCHK(hipMemcpyAsync(e->d.work_idx, feat_idx, (size_t)count * sizeof(int), hipMemcpyHostToDevice, e->stream));
    }
    int np = 0;
    int rc = predict_measurements_dev(e, sub ? e->d.work_idx : nullptr, count, &np);
    if (rc) return rc;
    if (n_preds) *n_preds = np;
    if (preds || Hs || Hf) return download_predictions(e, sub ? e->d.plist_sub : e->d.plist, np, preds, Hs, Hf);
    return EKF_OK;
}

int ekf_predict_measurement_state(EkfEngine *e, EkfPrediction *preds, int *n_preds)
{
    if (!e || !preds || !n_preds) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    launch_state_only_predict(e, e->d.preds_out);
    int rc = read_counts(e);
    if (rc) return rc;
    const int np = e->h_counts[CNT_NPRED_SUB];
    if (np > 0) HIPCHK(hipMemcpy(preds, e->d.preds_out, (size_t)np * sizeof(EkfPrediction), hipMemcpyDeviceToHost));
    *n_preds = np;
    return EKF_OK;
}

static int upload_keypoints(EkfEngine *e, const EkfKeypoint *kps, const uint8_t *desc32, int n_kp)
{
    if (n_kp > e->kcap) return EKF_ERR_CAPACITY;
    if (n_kp > 0) {
        HIPCHK(hipMemcpyAsync(e->d.kps, kps, (size_t)n_kp * sizeof(EkfKeypoint), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->d.kdesc, desc32, (size_t)n_kp * e->desc_bytes, hipMemcpyHostToDevice, e->stream));
    }
    e->n_kp = n_kp;
    return EKF_OK;
}

static int match_dev(EkfEngine *e, const EkfKeypoint *d_kps, const uint8_t *d_desc, int n_kp, int *n_matches)
{
    // kernels read e->d.kps / kdesc; staged frames alias them in
    EkfKeypoint *save_k = e->d.kps;
    uint8_t *save_d = e->d.kdesc;
    e->d.kps = const_cast<EkfKeypoint *>(d_kps);
    e->d.kdesc = const_cast<uint8_t *>(d_desc);
    launch_match(e, e->n_pred, n_kp);
    e->d.kps = save_k;
    e->d.kdesc = save_d;
    int rc = read_counts(e);
    if (rc) return rc;
    *n_matches = e->h_counts[CNT_NMATCH];
    return check_async(e);
}

// Sharded EKF::step ("Matching: features/ellipses independent => shard by feature; image replicated; all-gather match
// lists"): every rank gates and matches the predictions of the features it OWNS -- one run of slots of the feature-ordered
// predicted list (slot_rb, predict_measurements_dev) -- with either matcher, the per-slot tables (valid flag, keypoint index or
// matched pixel, distance) are all-gathered, and every rank compacts the same tables into the same match list
// (Matching.cpp:217-262 divided by the ranks; the list is identical to the unsharded one, which the tests assert).
static int match_sharded_dev(EkfEngine *e, bool use_ncc, const EkfKeypoint *d_kps, const uint8_t *d_desc, int n_kp, int *n_matches)
{
    const int W = e->shard_world, me = e->shard_rank, np = e->n_pred;
    if ((int)e->slot_rb.size() != W + 1 || e->slot_rb[0] != 0 || e->slot_rb[W] != np) {
        e->err = "sharded matching: the predicted list is not in feature order";
        return EKF_ERR_INVALID_ARG;
    }
    if (use_ncc && !e->img.valid) {
        e->err = "NCC matcher: no image uploaded";
        return EKF_ERR_INVALID_ARG;
    }
    EkfKeypoint *save_k = e->d.kps;
    uint8_t *save_d = e->d.kdesc;
    if (!use_ncc) { // kernels read e->d.kps / kdesc; staged frames alias them in
        e->d.kps = const_cast<EkfKeypoint *>(d_kps);
        e->d.kdesc = const_cast<uint8_t *>(d_desc);
    }
    int rc = EKF_OK;
    if (np > 0) {
        if (use_ncc) launch_match_ncc_slots(e, e->slot_rb[me], e->slot_rb[me + 1]);
        else launch_match_slots(e, n_kp, e->slot_rb[me], e->slot_rb[me + 1]);
        rc = exchange_rows(e, EKF_XCHG_MATCH_VALID, e->d.mt_valid, sizeof(int), e->slot_rb, "the match flags");
        if (!rc) rc = use_ncc ? exchange_rows(e, EKF_XCHG_MATCH_XY, e->d.mt_xy, sizeof(EkfKeypoint), e->slot_rb, "the matched pixels")
                              : exchange_rows(e, EKF_XCHG_MATCH_KP, e->d.mt_kp, sizeof(int), e->slot_rb, "the matched keypoint indices");
        if (!rc) rc = exchange_rows(e, EKF_XCHG_MATCH_DIST, e->d.mt_dist, sizeof(float), e->slot_rb, "the match distances");
    }
    if (!rc) {
        if (use_ncc) {
            if (np > 0) launch_match_compact_slots(e, np, e->d.mt_xy);
            else HIPCHK(hipMemsetAsync(e->d.counts + CNT_NMATCH, 0, sizeof(int), e->stream));
        } else launch_match_compact(e, np);
    }
    e->d.kps = save_k;
    e->d.kdesc = save_d;
    if (rc) return rc;
    if ((rc = read_counts(e))) return rc;
    *n_matches = e->h_counts[CNT_NMATCH];
    return check_async(e);
}

int ekf_match(EkfEngine *e, const EkfKeypoint *kps, const uint8_t *desc32, int n_kp, EkfMatch *matches,
              int *n_matches)
{
    if (!e || n_kp < 0 || (n_kp > 0 && (!kps || !desc32))) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    int rc = upload_keypoints(e, kps, desc32, n_kp);
    if (rc) return rc;
    int M = 0;
    rc = match_dev(e, e->d.kps, e->d.kdesc, n_kp, &M);
    if (rc) return rc;
    if (n_matches) *n_matches = M;
    if (matches && M > 0) HIPCHK(hipMemcpy(matches, e->d.matches, (size_t)M * sizeof(EkfMatch), hipMemcpyDeviceToHost));
    return EKF_OK;
}

// RANSAC over e->d.matches[0..M); on return best_flags holds the inlier mask, h_counts the loop state.
// lean (sharded EKF::step, feature-ordered match list): the hypotheses are divided by feature ownership, see below.
static int ransac_dev(EkfEngine *e, int M, bool lean = false)
{
    launch_ransac_init(e, M);
    launch_match_index(e, M);
    const int batch = e->cfg.ransac_batch;
    const int W = e->shard_world, me = e->shard_rank;
    // Sharded EKF::step ("RANSAC: hypotheses are independent => shard hypotheses across GPUs, all-gather the support
    // counts"): hypothesis h uses the H.P rows of ITS feature as gain columns, and those rows are with the feature's owner -- so every
    // rank evaluates the hypotheses of the features it owns (one run [mb[me], mb[me + 1]) of the feature-ordered match list: no row of
    // H.P travels) and the ranks all-gather a batch's support counts (4 bytes per hypothesis) and inlier masks (M bytes per
    // hypothesis) before the sequential bookkeeping, which every rank then replays identically (1PointRansac.cpp:125-180).
    const bool by_owner = lean && W > 1 && !e->hp_complete;
    std::vector<int32_t> mb, rb;
    if (by_owner) {
        launch_shard_bounds(e, e->d.matches, M);
        int rc = read_counts(e);
        if (rc) return rc;
        mb.assign(W + 1, 0);
        for (int r = 0; r <= W; ++r) mb[r] = e->h_counts[CNT_SHARD0 + r];
        if (mb[0] != 0 || mb[W] != M) {
            e->err = "sharded RANSAC: the match list is not in feature order";
            return EKF_ERR_INVALID_ARG;
        }
        rb.assign(W + 1, 0);
    }
    for (int h0 = 0, nb = batch; h0 < M; h0 += nb, nb = batch * RANSAC_WIDE_FACTOR) {
        const int batch = nb; // (this batch's width: the first one narrow -- a converged filter ends inside it --, the others wide)
        if (by_owner) {
            const int h1 = std::min(M, h0 + batch);
            launch_ransac_hyp(e, M, h0, batch, nullptr, mb[me], mb[me + 1]);
            for (int r = 0; r <= W; ++r) rb[r] = std::min(std::max(mb[r], h0), h1) - h0; // rows of the batch's tables by owner
            int rc;
            if ((rc = exchange_rows(e, EKF_XCHG_HYP_COUNT, e->d.hyp_count, sizeof(int), rb, "the support counts of a RANSAC batch"))) return rc;
            if ((rc = exchange_rows(e, EKF_XCHG_HYP_FLAGS, e->d.hyp_flags, (size_t)e->mcap, rb, "the inlier masks of a RANSAC batch"))) return rc;
            launch_ransac_select(e, M, h0, batch, nullptr, 0);
        } else {
            launch_ransac_batch(e, M, h0, batch);
        }
        int rc = read_counts(e);
        if (rc) return rc;
        if (e->h_counts[CNT_RS_DONE]) break;
    }
    return check_async(e);
}

// The device code keys predictions, Jacobians and H.P rows by featureIndex: one match per feature (what the matcher
// produces, Matching.cpp:217-262).  A list with a repeated featureIndex is rejected instead of being half-processed.
static int validate_matches(const EkfEngine *e, const EkfMatch *m, int M)
{
    if (M < 0 || M > e->cap || (M > 0 && !m)) return EKF_ERR_INVALID_ARG;
    std::vector<uint8_t> seen((size_t)e->N, 0);
    for (int i = 0; i < M; ++i) {
        const int f = m[i].featureIndex;
        if (f < 0 || f >= e->N || seen[f]) return EKF_ERR_INVALID_ARG;
        seen[f] = 1;
    }
    return EKF_OK;
}

int ekf_ransac(EkfEngine *e, const EkfMatch *matches, int M, uint8_t *inlier_mask, int *n_hypotheses)
{
    if (!e || !inlier_mask) return EKF_ERR_INVALID_ARG;
    int rc = validate_matches(e, matches, M);
    if (rc) return rc;
    if (n_hypotheses) *n_hypotheses = 0;
    if (M == 0) return EKF_OK;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->d.matches, matches, (size_t)M * sizeof(EkfMatch), hipMemcpyHostToDevice, e->stream));
    if ((rc = complete_hp_table(e))) return rc; // a sharded step leaves the H.P table with the owners' rows only
    rc = ransac_dev(e, M);
    if (rc) return rc;
    HIPCHK(hipMemcpy(inlier_mask, e->d.best_flags, (size_t)M, hipMemcpyDeviceToHost));
    if (n_hypotheses) *n_hypotheses = e->h_counts[CNT_RS_NEXT];
    return EKF_OK;
}

// Sharded EKF::step: k_gather has copied the H.P rows of the selected matches out of this rank's table, of which only the
// rows of OWNED features are valid.  The list is in feature order, so rank r's rows are rows [shard_rb[r], shard_rb[r+1])
// of the gathered matrix: the usual row-block exchange completes it (and the fp64 camera columns beside it).
static int lean_gather_exchange(EkfEngine *e, int M)
{
    (void)M;
    int rc;
    if ((rc = exchange_rows(e, EKF_XCHG_HP, e->d.G, (size_t)e->ldP * hp_elem_bytes(e), e->shard_rb, "the gathered H.P rows"))) return rc;
    if (e->f32 && !e->exact && (rc = exchange_rows(e, EKF_XCHG_HPC, e->d.Gc, 16 * sizeof(double), e->shard_rb, "their fp64 camera columns"))) return rc;
    return EKF_OK;
}

// update with the matches in e->d.msel[0..M); lean: see lean_gather_exchange
static int update_dev(EkfEngine *e, int M, bool update_cov, bool lean = false)
{
    if (M <= 0) return EKF_OK; // Update.cpp:292
    lean = lean && e->shard_world > 1 && !e->hp_complete;
    if (lean) { // where each rank's run of the (feature-ordered) selected matches starts
        launch_shard_bounds(e, e->d.msel, M);
        int rc = read_counts(e);
        if (rc) return rc;
        e->shard_rb.assign(e->shard_world + 1, 0);
        for (int r = 0; r <= e->shard_world; ++r) e->shard_rb[r] = 2 * e->h_counts[CNT_SHARD0 + r];
        if (e->shard_rb[0] != 0 || e->shard_rb[e->shard_world] != 2 * M) {
            e->err = "sharded update: the match list is not in feature order";
            return EKF_ERR_INVALID_ARG;
        }
    }
    EkfMatch *save = e->d.matches;
    e->d.matches = e->d.msel;
    e->after_gather = lean ? lean_gather_exchange : nullptr;
    launch_update(e, M, update_cov);
    e->after_gather = nullptr;
    e->d.matches = save;
    if (e->hook_rc) return e->hook_rc;
    return check_async(e);
}

// e->h_counts[CNT_ERR] != 0 was read back: the last update enqueued failed and, with everything enqueued behind it, left the filter
// as it was (filter_frozen, engine.h).  Clears the flag.  A persistent sweep that timed out (another process's kernels on the
// device, a partition that could not hold the grid: chol_persist.h) is not the caller's problem: the SAME update -- its match list
// is still in d.msel -- runs again on the launch-per-panel sweep, which has no cross-workgroup waits, from the untouched P
// (S, nu and the gathered rows are formed again); *status stays EKF_OK when that succeeds and the engine counts the retry.
// Anything else (S not positive definite) goes to *status: that update is skipped, as the reference skips it (cv::invert
// returns zeros, K = 0: EKF/Update.cpp:101-108).  Returns a hard error (HIP, exchange) or EKF_OK.
static int recover_failed_update(EkfEngine *e, int *status)
{
    const int code = e->h_counts[CNT_ERR];
    if (!code) return EKF_OK;
    HIPCHK(hipMemsetAsync(e->d.counts + CNT_ERR, 0, sizeof(int), e->stream));
    e->h_counts[CNT_ERR] = 0;
    e->p_exact_sym = e->last_update_sym; // (the frozen downdate did not symmetrise an uploaded P)
    if (code == EKF_ERR_TIMEOUT && e->last_update_persist && e->last_update_M > 0) {
        e->ps_backoff_len = std::min(std::max(64, 2 * e->ps_backoff_len), 4096); // see engine.h
        e->ps_backoff = e->ps_backoff_len;
        e->ps_ok_streak = 0;
        ++e->force_launches;
        int rc = update_dev(e, e->last_update_M, e->last_update_cov, false);
        --e->force_launches;
        ++e->sweep_retries;
        if (rc) return rc;
        if ((rc = read_counts(e))) return rc;
        if (!e->h_counts[CNT_ERR]) return EKF_OK;
        const int code2 = e->h_counts[CNT_ERR]; // the retry's own outcome (it cannot time out)
        HIPCHK(hipMemsetAsync(e->d.counts + CNT_ERR, 0, sizeof(int), e->stream));
        e->h_counts[CNT_ERR] = 0;
        e->p_exact_sym = e->last_update_sym;
        *status = code2;
    } else {
        *status = code;
    }
    e->err = *status == EKF_ERR_TIMEOUT ? "the persistent Cholesky sweep timed out"
             : (*status == EKF_ERR_NON_FINITE ? "a row of B = inv(L) H P does not fit its a-priori column scale (covariance not positive semi-definite, or not finite)"
                                              : "S = H P H' + R is not positive definite");
    return EKF_OK;
}

static int finish_update(EkfEngine *e)
{
    int rc = read_counts(e);
    if (rc) return rc;
    int status = EKF_OK;
    if ((rc = recover_failed_update(e, &status))) return rc;
    return status;
}

int ekf_update(EkfEngine *e, const EkfMatch *matches, int M)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    int rc = validate_matches(e, matches, M);
    if (rc) return rc;
    if (M == 0) return EKF_OK;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->d.msel, matches, (size_t)M * sizeof(EkfMatch), hipMemcpyHostToDevice, e->stream));
    if ((rc = complete_hp_table(e))) return rc;
    rc = update_dev(e, M, true);
    if (rc) return rc;
    return finish_update(e);
}

int ekf_update_only_state(EkfEngine *e, const EkfMatch *matches, int M)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    int rc = validate_matches(e, matches, M);
    if (rc) return rc;
    if (M == 0) return EKF_OK;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->d.msel, matches, (size_t)M * sizeof(EkfMatch), hipMemcpyHostToDevice, e->stream));
    if ((rc = complete_hp_table(e))) return rc;
    rc = update_dev(e, M, false);
    if (rc) return rc;
    return finish_update(e);
}

int ekf_rescue(EkfEngine *e, const EkfMatch *outliers, int M, uint8_t *rescued_mask)
{
    if (!e || !rescued_mask) return EKF_ERR_INVALID_ARG;
    int rc = validate_matches(e, outliers, M);
    if (rc) return rc;
    if (M == 0) return EKF_OK;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->d.mout, outliers, (size_t)M * sizeof(EkfMatch), hipMemcpyHostToDevice, e->stream));
    EkfMatch *save = e->d.matches;
    e->d.matches = e->d.mout;
    launch_rescue(e, M);
    e->d.matches = save;
    HIPCHK(hipMemcpyAsync(rescued_mask, e->d.mask, (size_t)M, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return check_async(e);
}

// ------------------------------------------------------------------------------------------------------ step
static void harvest_pu_events(EkfEngine *e)
{
    for (auto &pr : e->px_events) { // exact downdate: the column-scale + digit-plane kernels (their end event is the downdate's start event, destroyed below)
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) e->slice_ms += ms;
        (void)hipEventDestroy(pr.first);
    }
    e->px_events.clear();
    for (size_t i = 0; i < e->pu_events.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->pu_events[i].first, e->pu_events[i].second) == hipSuccess) {
            e->times.p_update_kernel_ms += ms;
            e->times.p_update_launches += 1;
            e->times.p_update_flops += e->pu_work[i];
            e->times.p_update_bytes += 2.0 * (double)e->n * (double)e->n * (e->f32 ? 4.0 : 8.0);
            e->pu_log.emplace_back(e->pu_m[i], ms);
        }
        (void)hipEventDestroy(e->pu_events[i].first);
        (void)hipEventDestroy(e->pu_events[i].second);
    }
    e->pu_events.clear();
    e->pu_work.clear();
    e->pu_m.clear();
    for (size_t i = 0; i < e->sw_events.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->sw_events[i].first, e->sw_events[i].second) == hipSuccess) {
            const double m = std::abs(e->sw_m[i]);
            e->sweep_ms += ms;
            e->sweep_panels += ((long long)m + NB - 1) / NB;
            e->sweep_launches += i < e->sw_launch.size() ? e->sw_launch[i] : 0;
            e->sweep_updates += 1;
            e->sweep_flops_f64 += m * m * m / 3.0;                        // Cholesky of S
            if (e->sw_m[i] > 0) e->sweep_flops_b += m * m * (double)e->n;  // B = inv(L) (H P): forward substitution, m^2 n
        }
        (void)hipEventDestroy(e->sw_events[i].first);
        (void)hipEventDestroy(e->sw_events[i].second);
    }
    e->sw_events.clear();
    e->sw_launch.clear();
    e->sw_m.clear();
}

struct StageTimer {
    EkfEngine *e;
    std::vector<hipEvent_t> evs;
    explicit StageTimer(EkfEngine *eng) : e(eng) {}
    ~StageTimer() // early returns of step_dev skip finish(): the events recorded so far are released here
    {
        for (auto ev : evs) (void)hipEventDestroy(ev);
    }
    void mark()
    {
        if (!e->timing) return;
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return;
        (void)hipEventRecord(ev, e->stream);
        evs.push_back(ev);
    }
    void finish()
    {
        if (!e->timing || evs.size() < 7) return;
        (void)hipStreamSynchronize(e->stream);
        double *dst[6] = {&e->times.prediction_ms, &e->times.matching_ms, &e->times.ransac_ms,
                          &e->times.update_li_ms,  &e->times.rescue_ms,   &e->times.update_hi_ms};
        for (int i = 0; i < 6; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, evs[i], evs[i + 1]) == hipSuccess) *dst[i] += ms;
        }
        for (auto ev : evs) (void)hipEventDestroy(ev);
        evs.clear();
        e->times.steps += 1;
        harvest_pu_events(e);
    }
};

// EKF::step (EKF.cpp:242-556) with keypoints already on the device
static int match_ncc_dev(EkfEngine *e, int *n_matches);

static int step_dev_fast(EkfEngine *e, const EkfKeypoint *d_kps, const uint8_t *d_desc, int n_kp, EkfStepInfo *info);

static int step_dev(EkfEngine *e, const EkfKeypoint *d_kps, const uint8_t *d_desc, int n_kp, EkfStepInfo *info,
                    bool use_ncc = false)
{
    if (!use_ncc && !e->img.valid && !e->keep_step_preds && e->shard_world == 1) return step_dev_fast(e, d_kps, d_desc, n_kp, info);
    EkfStepInfo li;
    std::memset(&li, 0, sizeof(li));
    int status = EKF_OK, rc;
    const int retries0 = e->sweep_retries;
    StageTimer tm(e);
    tm.mark();
    // 1-2. prediction (:273-284)
    launch_predict(e);
    int np = 0;
    const bool lean = e->shard_world > 1; // sharded: rows of H.P travel when consumed, not after every prediction
    if ((rc = predict_measurements_dev(e, nullptr, e->N, &np, true, lean))) return rc; // + timesPredicted++ (EKF.cpp:572)
    li.n_predicted = np;
    tm.mark();
    // 4. matching (:337)
    int M = 0;
    if (lean) rc = match_sharded_dev(e, use_ncc, d_kps, d_desc, n_kp, &M); // every rank matches the predictions of its own features
    else rc = use_ncc ? match_ncc_dev(e, &M) : match_dev(e, d_kps, d_desc, n_kp, &M);
    if (rc) return rc;
    li.n_matches = M;
    tm.mark();
    // 6. 1-point RANSAC (:402); predictions/Jacobians are looked up by featureIndex (:368-392)
    int ni = 0, no = 0;
    if (M > 0) {
        if ((rc = ransac_dev(e, M, lean))) return rc;
        li.n_hypotheses = e->h_counts[CNT_RS_NEXT];
        ni = e->h_counts[CNT_RS_BEST];
        no = M - ni;
        // + updateMapFeatures for the low-innovation inliers (MapManagement.cpp:88-113), same launch
        launch_partition(e, e->d.matches, M, e->d.best_flags, e->d.msel, e->d.mout, nullptr, true, d_desc, e->d.work_idx);
    }
    li.n_inliers = ni;
    li.n_outliers = no;
    tm.mark();
    // 7. low-innovation update (:430)
    if ((rc = update_dev(e, ni, true, lean))) return rc;
    tm.mark();
    // 8-9. re-predict the outliers with the updated state / covariance, rescue (:473-506)
    int nr = 0;
    if (no > 0) {
        int nop = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            if ((rc = predict_measurements_dev(e, e->d.work_idx, no, &nop, false, lean))) return rc;
            if (!e->h_counts[CNT_ERR]) break;
            // the first update failed (seen at this stage's read-back; everything behind it was frozen): run it again or skip it
            // (recover_failed_update), then this stage again
            if ((rc = recover_failed_update(e, &status))) return rc;
        }
        if (nop > 0) {
            EkfMatch *save = e->d.matches;
            e->d.matches = e->d.mout;
            launch_rescue(e, no);
            e->d.matches = save;
            launch_partition(e, e->d.mout, no, e->d.mask, e->d.msel, nullptr, e->d.counts + CNT_NRESC, true, d_desc); // rescued matches join the inliers (EKF.cpp:552-556)
            if ((rc = read_counts(e))) return rc;
            nr = e->h_counts[CNT_NRESC];
        }
    }
    li.n_rescued = nr;
    tm.mark();
    // 10. high-innovation update (:529-532)
    if ((rc = update_dev(e, nr, true, lean))) return rc;
    tm.mark();
    if ((rc = read_counts(e))) return rc;
    if ((rc = recover_failed_update(e, &status))) return rc;
    tm.finish();
    li.status = status;
    li.n_sweep_retries = e->sweep_retries - retries0;
    if (info) *info = li;
    return status;
}

// EKF::step for the common case (descriptor matcher, whole filter on this GPU, nothing asked to be kept for the host)
// with HALF the host round trips of step_dev: how many features were predicted, how many matches there are and how many
// outliers were re-predicted only decide launch sizes, so the launches use upper bounds (N, N, the outlier count) and
// the kernels read the counts on the device.  What the host still needs to know -- the RANSAC result (loop state and
// inlier count: the size of the first update), the rescued count (the size of the second), the error flag -- comes from
// three read-backs instead of six (two with ekf_set_async_errors); each one used to idle the GPU for ~10-15 us.
static int step_dev_fast(EkfEngine *e, const EkfKeypoint *d_kps, const uint8_t *d_desc, int n_kp, EkfStepInfo *info)
{
    EkfStepInfo li;
    std::memset(&li, 0, sizeof(li));
    int status = EKF_OK, rc;
    const int retries0 = e->sweep_retries;
    const int N = e->N;
    int *cnt = e->d.counts;
    // A failed update is seen at the next read-back; until then everything enqueued behind it has left the filter alone
    // (filter_frozen, engine.h), so whatever ran meanwhile is simply run again after recover_failed_update.
    bool restarted = false;
restart:
    StageTimer tm(e);
    tm.mark();
    // 1-2. prediction (:273-284), timesPredicted++ (EKF.cpp:572)
    {   // covariance strips and pixel predictions in one launch; with more than 256 features the compaction of the predicted list
        // rides in the launch of the H P rows
        const bool deferred = launch_predict_with_features(e, N);
        launch_hp_rows(e, e->d.plist, N, true, cnt + CNT_NPRED, deferred);
    }
    tm.mark();
    // 4. matching (:337)
    {
        EkfKeypoint *save_k = e->d.kps;
        uint8_t *save_d = e->d.kdesc;
        e->d.kps = const_cast<EkfKeypoint *>(d_kps);
        e->d.kdesc = const_cast<uint8_t *>(d_desc);
        launch_match(e, N, n_kp, cnt + CNT_NPRED, true); // + the RANSAC loop state and the feature -> match index (same launch)
        e->d.kps = save_k;
        e->d.kdesc = save_d;
    }
    tm.mark();
    // 6. 1-point RANSAC (:402): the first batch is launched before anything is known on the host
    const int batch = e->cfg.ransac_batch;
    const int seq_r = next_publish_seq(e);
    launch_ransac_batch(e, N, 0, batch, cnt + CNT_NMATCH, seq_r);
    if ((rc = wait_counts(e, seq_r))) return rc;
    if (e->h_counts[CNT_ERR] && !restarted) {
        // the previous step's last update failed and its final read-back was skipped (ekf_set_async_errors): this step's prediction
        // did not touch the filter -- recover (that update again, or skipped and reported here), then this step from its start
        if ((rc = recover_failed_update(e, &status))) return rc;
        restarted = true;
        goto restart;
    }
    const int np = e->h_counts[CNT_NPRED], M = e->h_counts[CNT_NMATCH];
    e->n_pred = np;
    li.n_predicted = np;
    li.n_matches = M;
    for (int h0 = batch, nb = batch * RANSAC_WIDE_FACTOR; !e->h_counts[CNT_RS_DONE] && h0 < M; h0 += nb) {
        launch_ransac_batch(e, M, h0, nb); // (wide batches behind the first: see ransac_dev)
        if ((rc = read_counts(e))) return rc;
    }
    int ni = 0, no = 0;
    if (M > 0) {
        li.n_hypotheses = e->h_counts[CNT_RS_NEXT];
        ni = e->h_counts[CNT_RS_BEST];
        no = M - ni;
        // + updateMapFeatures for the low-innovation inliers (MapManagement.cpp:88-113), same launch
        launch_partition(e, e->d.matches, M, e->d.best_flags, e->d.msel, e->d.mout, nullptr, true, d_desc, e->d.work_idx);
    }
    li.n_inliers = ni;
    li.n_outliers = no;
    tm.mark();
    // 7. low-innovation update (:430)
    if ((rc = update_dev(e, ni, true))) return rc;
    tm.mark();
    // 8-9. re-predict the outliers with the updated state / covariance, rescue (:473-506).  An outlier that is not
    // re-predicted has pred_vis = 0 and is not rescued; with none re-predicted nothing is (the reference's case is
    // undefined behaviour, see step_dev).
    int nr = 0;
    if (no > 0) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            launch_predict_features(e, e->d.work_idx, no, false);
            launch_hp_rows(e, e->d.plist_sub, no, false, cnt + CNT_NPRED_SUB);
            const int seq_p = next_publish_seq(e);
            // rescueOutliers (EKF.cpp:84-97) and the partition it feeds in one launch: rescued matches join the inliers (EKF.cpp:552-556)
            launch_partition(e, e->d.mout, no, e->d.mask, e->d.msel, nullptr, cnt + CNT_NRESC, true, d_desc, nullptr, seq_p, true);
            if ((rc = wait_counts(e, seq_p))) return rc;
            if (!e->h_counts[CNT_ERR]) break;
            // the first update failed: the partition above did nothing (the inliers are still in d.msel); that update again or
            // skipped, then this stage again
            if ((rc = recover_failed_update(e, &status))) return rc;
        }
        nr = e->h_counts[CNT_NRESC];
    }
    li.n_rescued = nr;
    tm.mark();
    // 10. high-innovation update (:529-532)
    if ((rc = update_dev(e, nr, true))) return rc;
    tm.mark();
    if (!e->async_errors) {
        if ((rc = read_counts(e))) return rc;
        if ((rc = recover_failed_update(e, &status))) return rc;
    }
    tm.finish();
    li.status = status;
    li.n_sweep_retries = e->sweep_retries - retries0;
    if (info) *info = li;
    return status;
}

int ekf_set_async_errors(EkfEngine *e, int on)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    e->async_errors = on != 0;
    return EKF_OK;
}

int ekf_set_update_path(EkfEngine *e, int path)
{
    if (!e || path < EKF_UPDATE_PATH_AUTO || path > EKF_UPDATE_PATH_GEMM) return EKF_ERR_INVALID_ARG;
    e->b_path = path;
    return EKF_OK;
}

int ekf_set_sweep_mode(EkfEngine *e, int mode)
{
    if (!e || mode < EKF_SWEEP_PAIRS || mode > EKF_SWEEP_LAUNCHES) return EKF_ERR_INVALID_ARG;
    e->sweep_mode = mode;
    return EKF_OK;
}

int ekf_step(EkfEngine *e, const EkfKeypoint *kps, const uint8_t *desc32, int n_kp, EkfStepInfo *info)
{
    if (!e || n_kp < 0 || (n_kp > 0 && (!kps || !desc32))) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    int rc = upload_keypoints(e, kps, desc32, n_kp);
    if (rc) return rc;
    return step_dev(e, e->d.kps, e->d.kdesc, n_kp, info);
}

int ekf_frames_upload(EkfEngine *e, int n_frames, const int32_t *kp_counts, const EkfKeypoint *kps_concat,
                      const uint8_t *desc_concat)
{
    if (!e || n_frames < 0 || (n_frames > 0 && (!kp_counts || !kps_concat || !desc_concat))) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->frames.kps) (void)hipFree(e->frames.kps);
    if (e->frames.desc) (void)hipFree(e->frames.desc);
    e->frames = Frames();
    size_t total = 0;
    for (int i = 0; i < n_frames; ++i) {
        if (kp_counts[i] < 0) return EKF_ERR_INVALID_ARG;
        e->frames.offset.push_back((int)total);
        e->frames.count.push_back(kp_counts[i]);
        total += (size_t)kp_counts[i];
    }
    e->frames.n = n_frames;
    if (total == 0) return EKF_OK;
    HIPCHK(hipMalloc((void **)&e->frames.kps, total * sizeof(EkfKeypoint)));
    HIPCHK(hipMalloc((void **)&e->frames.desc, total * e->desc_bytes));
    HIPCHK(hipMemcpy(e->frames.kps, kps_concat, total * sizeof(EkfKeypoint), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->frames.desc, desc_concat, total * e->desc_bytes, hipMemcpyHostToDevice));
    return EKF_OK;
}

int ekf_step_frame(EkfEngine *e, int frame, EkfStepInfo *info)
{
    if (!e || frame < 0 || frame >= e->frames.n) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    const size_t off = (size_t)e->frames.offset[frame];
    return step_dev(e, e->frames.kps + off, e->frames.desc + off * e->desc_bytes, e->frames.count[frame], info);
}

// ------------------------------------------------------------------------------------- NCC matcher (mode B)
static int ensure_pyramid(EkfEngine *e, int w, int h)
{
    if (e->img.w[0] == w && e->img.h[0] == h && e->img.px[0]) return EKF_OK;
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->stream2) HIPCHK(hipStreamSynchronize(e->stream2));
    e->img.prefetched = -1;
    int lw = w, lh = h;
    for (int l = 0; l < 3; ++l) {
        if (e->img.px[l]) (void)hipFree(e->img.px[l]);
        if (e->img.px2[l]) (void)hipFree(e->img.px2[l]);
        e->img.px[l] = e->img.px2[l] = nullptr;
        e->img.w[l] = lw;
        e->img.h[l] = lh;
        HIPCHK(hipMalloc((void **)&e->img.px[l], (size_t)std::max(lw, 1) * std::max(lh, 1)));
        HIPCHK(hipMalloc((void **)&e->img.px2[l], (size_t)std::max(lw, 1) * std::max(lh, 1)));
        lw /= 2;
        lh /= 2;
    }
    return EKF_OK;
}

static int valid_image_args(const uint8_t *image, int w, int h, int stride, int channels)
{
    return image && w >= 4 && h >= 4 && (channels == 1 || channels == 3 || channels == 4) && stride >= w * channels;
}

int ekf_image_upload(EkfEngine *e, const uint8_t *image, int width, int height, int stride, int channels)
{
    if (!e || !valid_image_args(image, width, height, stride, channels)) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    int rc = ensure_pyramid(e, width, height);
    if (rc) return rc;
    const size_t bytes = (size_t)stride * height;
    if (e->img.raw_cap < bytes) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->img.raw) (void)hipFree(e->img.raw);
        e->img.raw = nullptr;
        HIPCHK(hipMalloc((void **)&e->img.raw, bytes));
        e->img.raw_cap = bytes;
    }
    HIPCHK(hipMemcpyAsync(e->img.raw, image, bytes, hipMemcpyHostToDevice, e->stream));
    launch_ncc_pyramid(e, e->img.raw, stride, channels);
    e->img.valid = true;
    return check_async(e);
}

int ekf_get_image_level(EkfEngine *e, int level, uint8_t *out, int *width, int *height)
{
    if (!e || level < 0 || level > 2 || !e->img.valid) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (width) *width = e->img.w[level];
    if (height) *height = e->img.h[level];
    if (out) HIPCHK(hipMemcpy(out, e->img.px[level], (size_t)e->img.w[level] * e->img.h[level], hipMemcpyDeviceToHost));
    return EKF_OK;
}

int ekf_capture_templates(EkfEngine *e, const int32_t *feat_idx, const double *uv, int count)
{
    if (!e || count < 0 || (count > 0 && (!feat_idx || !uv)) || !e->img.valid) return EKF_ERR_INVALID_ARG;
    if (count > e->cap) return EKF_ERR_CAPACITY;
    for (int i = 0; i < count; ++i)
        if (feat_idx[i] < 0 || feat_idx[i] >= e->N) return EKF_ERR_INVALID_ARG;
    if (count == 0) return EKF_OK;
    HIPCHK(hipSetDevice(e->device));
    // work_idx / pred_uv2 are free between stages
    HIPCHK(hipMemcpyAsync(e->d.work_idx, feat_idx, (size_t)count * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d.pred_uv2, uv, (size_t)count * 2 * sizeof(double), hipMemcpyHostToDevice, e->stream));
    launch_ncc_capture(e, e->d.work_idx, e->d.pred_uv2, count);
    HIPCHK(hipStreamSynchronize(e->stream)); // the host arrays may be reused by the caller
    return check_async(e);
}

// detectNewImageFeatures (EKF/DetectNewImageFeatures.cpp:337-367) on the current image: device = masked corner
// candidates (kernels_detect.hip), host = the zone heuristic of searchFeaturesByZone (:171-330), with the two
// nondeterministic choices of the reference made deterministic: zones of equal population keep their id order (qsort
// is unstable) and the pick inside a zone is its strongest remaining candidate (the reference draws rand()).
int ekf_detect_new_features(EkfEngine *e, int max_new, int divide_times, double mask_ellipse_size, double min_response,
                            double *uv_out, int *count)
{
    if (!e || !count || max_new < 0 || divide_times < 0 || divide_times > 6 || (max_new > 0 && !uv_out)) return EKF_ERR_INVALID_ARG;
    *count = 0;
    if (!e->img.valid) {
        e->err = "new-feature detection: no image uploaded";
        return EKF_ERR_INVALID_ARG;
    }
    if (max_new == 0) return EKF_OK;
    HIPCHK(hipSetDevice(e->device));
    const int w = e->img.w[0], h = e->img.h[0];
    const int cells_x = w / 16, cells_y = h / 16, ncell = cells_x * cells_y;
    if (ncell <= 0) return EKF_OK;
    if (e->cells_cap < ncell) {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->d.cell_resp) (void)hipFree(e->d.cell_resp);
        if (e->d.cell_xy) (void)hipFree(e->d.cell_xy);
        e->d.cell_resp = nullptr;
        e->d.cell_xy = nullptr;
        HIPCHK(hipMalloc((void **)&e->d.cell_resp, (size_t)ncell * sizeof(long long)));
        HIPCHK(hipMalloc((void **)&e->d.cell_xy, (size_t)ncell * 2 * sizeof(int)));
        e->cells_cap = ncell;
    }
    launch_detect_cells(e, e->n_gates, cells_x, cells_y, e->d.cell_resp, e->d.cell_xy);
    std::vector<long long> resp(ncell);
    std::vector<int> xy(2 * (size_t)ncell);
    std::vector<double> gates(8 * (size_t)std::max(e->n_gates, 1));
    HIPCHK(hipMemcpyAsync(resp.data(), e->d.cell_resp, resp.size() * sizeof(long long), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(xy.data(), e->d.cell_xy, xy.size() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    if (e->n_gates > 0)
        HIPCHK(hipMemcpyAsync(gates.data(), e->d.gates, (size_t)e->n_gates * 8 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    int rc = check_async(e);
    if (rc) return rc;

    struct Cand { int x, y; long long r; };
    std::vector<Cand> cands;
    const long long thr = min_response >= 9.2e18 ? 0x7fffffffffffffffLL : (min_response <= 0 ? 0 : (long long)min_response);
    for (int c = 0; c < ncell; ++c)
        if (resp[c] >= 0 && resp[c] >= thr) cands.push_back(Cand{xy[2 * c], xy[2 * c + 1], resp[c]});
    if ((int)cands.size() <= max_new) { // :357-370: fewer than asked for -> all of them
        for (size_t i = 0; i < cands.size(); ++i) {
            uv_out[2 * i] = cands[i].x;
            uv_out[2 * i + 1] = cands[i].y;
        }
        *count = (int)cands.size();
        return EKF_OK;
    }
    const int zones_row = 1 << divide_times;
    const int zw = std::max(w / zones_row, 1), zh = std::max(h / zones_row, 1);
    const int nzone = zones_row * zones_row;
    auto zone_of = [&](double x, double y) { // getPointZone :87-92
        const int id = ((int)y / zh) * (w / zw) + (int)x / zw;
        return std::min(std::max(id, 0), nzone - 1);
    };
    struct Zone { int id, count; std::vector<int> cand; };
    std::vector<Zone> zones(nzone);
    for (int z = 0; z < nzone; ++z) zones[z] = Zone{z, 0, {}};
    for (size_t i = 0; i < cands.size(); ++i) zones[zone_of(cands[i].x, cands[i].y)].cand.push_back((int)i);
    for (int k = 0; k < e->n_gates; ++k) zones[zone_of(gates[8 * (size_t)k + 5], gates[8 * (size_t)k + 6])].count++;
    std::vector<int> order(nzone);
    for (int z = 0; z < nzone; ++z) order[z] = z;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return zones[a].count < zones[b].count; });
    // mask of the features added in this call: the "ellipse" of diag(size, size), a disc of integer radius
    const int radius = (int)std::nearbyint((float)(2.0 * std::sqrt(mask_ellipse_size * EKF_CHISQ_95_2)));
    std::vector<std::pair<int, int>> added;
    int left = max_new, n_out = 0;
    size_t head = 0; // order[head..] is the list; zones without candidates are popped from the front
    while (head < order.size() && left > 0) {
        Zone &z = zones[order[head]];
        if (z.cand.empty()) {
            ++head;
            continue;
        }
        size_t best = 0;
        for (size_t i = 1; i < z.cand.size(); ++i) { // strongest, ties -> first in cell order
            const Cand &a = cands[z.cand[i]], &bb = cands[z.cand[best]];
            if (a.r > bb.r || (a.r == bb.r && z.cand[i] < z.cand[best])) best = i;
        }
        const Cand c = cands[z.cand[best]];
        bool free_px = true;
        for (const auto &a : added) {
            const double dx = (double)(float)c.x - (double)(float)a.first, dy = (double)(float)c.y - (double)(float)a.second;
            if (2.0 * std::sqrt(dx * dx + dy * dy) <= 2.0 * radius) { free_px = false; break; }
        }
        if (free_px) {
            uv_out[2 * n_out] = c.x;
            uv_out[2 * n_out + 1] = c.y;
            ++n_out;
            z.count++;
            for (size_t p = head; p + 1 < order.size(); ++p) { // :272-296: keep the list ordered by population
                if (zones[order[p]].count >= zones[order[p + 1]].count) std::swap(order[p], order[p + 1]);
                else break;
            }
            added.emplace_back(c.x, c.y);
            --left;
        }
        // the candidate is consumed either way (:314-318); z may have moved, so erase through the zone object
        Zone &zz = zones[zone_of(c.x, c.y)];
        for (size_t i = 0; i < zz.cand.size(); ++i)
            if (cands[zz.cand[i]].x == c.x && cands[zz.cand[i]].y == c.y) {
                zz.cand[i] = zz.cand.back();
                zz.cand.pop_back();
                break;
            }
    }
    *count = n_out;
    return EKF_OK;
}

static int match_ncc_dev(EkfEngine *e, int *n_matches)
{
    if (!e->img.valid) {
        e->err = "NCC matcher: no image uploaded";
        return EKF_ERR_INVALID_ARG;
    }
    launch_match_ncc(e, e->n_pred);
    int rc = read_counts(e);
    if (rc) return rc;
    *n_matches = e->h_counts[CNT_NMATCH];
    return check_async(e);
}

int ekf_match_ncc(EkfEngine *e, EkfMatch *matches, int *n_matches)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    int M = 0;
    int rc = match_ncc_dev(e, &M);
    if (rc) return rc;
    if (n_matches) *n_matches = M;
    if (matches && M > 0) HIPCHK(hipMemcpy(matches, e->d.matches, (size_t)M * sizeof(EkfMatch), hipMemcpyDeviceToHost));
    return EKF_OK;
}

int ekf_step_image(EkfEngine *e, const uint8_t *image, int width, int height, int stride, int channels, EkfStepInfo *info)
{
    int rc = ekf_image_upload(e, image, width, height, stride, channels);
    if (rc) return rc;
    return step_dev(e, nullptr, nullptr, 0, info, true);
}

int ekf_images_upload(EkfEngine *e, int n_frames, const uint8_t *images, int width, int height, int stride, int channels)
{
    if (!e || n_frames < 0 || (n_frames > 0 && !valid_image_args(images, width, height, stride, channels))) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->stream2) HIPCHK(hipStreamSynchronize(e->stream2));
    e->img.prefetched = -1;
    if (e->img.seq) (void)hipFree(e->img.seq);
    e->img.seq = nullptr;
    e->img.seq_n = 0;
    if (n_frames == 0) return EKF_OK;
    const size_t bytes = (size_t)stride * height * n_frames;
    HIPCHK(hipMalloc((void **)&e->img.seq, bytes));
    HIPCHK(hipMemcpy(e->img.seq, images, bytes, hipMemcpyHostToDevice));
    e->img.seq_n = n_frames;
    e->img.seq_w = width;
    e->img.seq_h = height;
    e->img.seq_stride = stride;
    e->img.seq_channels = channels;
    return EKF_OK;
}

static int staged_pyramid(EkfEngine *e, int frame)
{
    if (frame < 0 || frame >= e->img.seq_n) return EKF_ERR_INVALID_ARG;
    int rc = ensure_pyramid(e, e->img.seq_w, e->img.seq_h);
    if (rc) return rc;
    const size_t frame_bytes = (size_t)e->img.seq_stride * e->img.seq_h;
    if (frame == e->img.prefetched) { // reduced on stream2 while the previous frame was being filtered
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_prefetch, 0));
        for (int l = 0; l < 3; ++l) std::swap(e->img.px[l], e->img.px2[l]);
    } else {
        launch_ncc_pyramid(e, e->img.seq + (size_t)frame * frame_bytes, e->img.seq_stride, e->img.seq_channels);
    }
    e->img.prefetched = -1;
    e->img.valid = true;
    // image-only work of the NEXT staged frame goes to the second stream, behind everything already queued on the main
    // stream (the last readers of the buffer it overwrites), and runs concurrently with this frame's filter step
    if (frame + 1 < e->img.seq_n && e->stream2) {
        HIPCHK(hipEventRecord(e->ev_main, e->stream));
        HIPCHK(hipStreamWaitEvent(e->stream2, e->ev_main, 0));
        launch_ncc_pyramid_on(e, e->stream2, e->img.px2, e->img.seq + (size_t)(frame + 1) * frame_bytes, e->img.seq_stride,
                              e->img.seq_channels);
        HIPCHK(hipEventRecord(e->ev_prefetch, e->stream2));
        e->img.prefetched = frame + 1;
    }
    return EKF_OK;
}

int ekf_select_staged_image(EkfEngine *e, int frame)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    int rc = staged_pyramid(e, frame);
    return rc ? rc : check_async(e);
}

int ekf_step_staged_image(EkfEngine *e, int frame, EkfStepInfo *info)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    int rc = staged_pyramid(e, frame);
    if (rc) return rc;
    return step_dev(e, nullptr, nullptr, 0, info, true);
}

// -------------------------------------------------------------------------------------------------- timing
int ekf_timing_enable(EkfEngine *e, int on)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    e->timing = on != 0;
    return EKF_OK;
}

int ekf_timing_reset(EkfEngine *e)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(e->stream);
    harvest_pu_events(e);
    e->pu_log.clear();
    std::memset(&e->times, 0, sizeof(e->times));
    e->sweep_ms = e->sweep_flops_f64 = e->sweep_flops_b = 0.0;
    e->sweep_panels = e->sweep_updates = e->sweep_launches = 0;
    e->slice_ms = 0.0;
    return EKF_OK;
}

int ekf_timing_get(EkfEngine *e, EkfStageTimes *out)
{
    if (!e || !out) return EKF_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(e->stream);
    harvest_pu_events(e);
    *out = e->times;
    return EKF_OK;
}

int ekf_timing_sweep(EkfEngine *e, double *kernel_ms, int64_t *panels, int64_t *updates, double *flops_fp64, double *flops_b)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(e->stream);
    harvest_pu_events(e);
    if (kernel_ms) *kernel_ms = e->sweep_ms;
    if (panels) *panels = e->sweep_panels;
    if (updates) *updates = e->sweep_updates;
    if (flops_fp64) *flops_fp64 = e->sweep_flops_f64;
    if (flops_b) *flops_b = e->sweep_flops_b;
    return EKF_OK;
}

int ekf_get_sweep_retries(const EkfEngine *e) { return e ? e->sweep_retries : 0; }

int ekf_debug_stall_next_sweep(EkfEngine *e) // include/ekf_test_hooks.h
{
    if (!e) return EKF_ERR_INVALID_ARG;
    e->ps_fault = 1;
    return EKF_OK;
}

int ekf_debug_dense_products(EkfEngine *e, int on)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    e->px_dense = on != 0;
    return EKF_OK;
}

int ekf_debug_plane0_pieces(EkfEngine *e, int *nonzero, int *total)
{
    if (!e || !nonzero || !total || !e->d.Bz) return EKF_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const int cbs = (e->n + 31) / 32, groups = (2 * e->last_update_M + 15) / 16;
    std::vector<uint8_t> row((size_t)e->bz_stride);
    int nz = 0;
    for (int c = 0; c < cbs; ++c) {
        HIPCHK(hipMemcpy(row.data(), e->d.Bz + (size_t)c * e->bz_stride, (size_t)groups, hipMemcpyDeviceToHost));
        for (int k = 0; k < groups; ++k) nz += row[k] ? 1 : 0;
    }
    *nonzero = nz;
    *total = cbs * groups;
    return EKF_OK;
}

int ekf_debug_stall_sweep_after(EkfEngine *e, int skip)
{
    if (!e || skip < 0) return EKF_ERR_INVALID_ARG;
    e->ps_fault = skip + 1; // counted down by the persistent launches; the one that reaches zero runs without its chain workgroup
    return EKF_OK;
}

int ekf_round_covariance_to_f32(EkfEngine *e)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    if (e->f32) return EKF_ERR_INVALID_ARG; // an fp64 engine only: the fp32 configurations store fp32 already
    HIPCHK(hipSetDevice(e->device));
    launch_round_P_f32(e);
    return EKF_OK;
}

int ekf_timing_sweep_launches(EkfEngine *e, int64_t *launches, double *slice_ms)
{
    if (!e) return EKF_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(e->stream);
    harvest_pu_events(e);
    if (launches) *launches = e->sweep_launches;
    if (slice_ms) *slice_ms = e->slice_ms;
    return EKF_OK;
}

int ekf_timing_p_update_launches(EkfEngine *e, int capacity, int32_t *m_rows, float *ms, int *count)
{
    if (!e || !count) return EKF_ERR_INVALID_ARG;
    (void)hipStreamSynchronize(e->stream);
    harvest_pu_events(e);
    *count = (int)e->pu_log.size();
    for (int i = 0; i < *count && i < capacity; ++i) {
        if (m_rows) m_rows[i] = e->pu_log[i].first;
        if (ms) ms[i] = e->pu_log[i].second;
    }
    return EKF_OK;
}

} // extern "C"
