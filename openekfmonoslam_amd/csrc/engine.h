// engine.h -- internal layout of the device-resident filter and the kernel launchers (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ekf_engine.h"
#include "device_math.h"

namespace ekf {

constexpr int NB = 32;          // Cholesky panel width
constexpr int LD_ALIGN = 128;   // leading dimensions are multiples of this many elements
constexpr int B_SWEEP_MAX = 2048; // rows of S up to which B = inv(L) G is formed inside the sweep's launches
constexpr int DX_SPLIT = 16;    // k-splits of the dx = B' z reduction
constexpr int RANSAC_WIDE_FACTOR = 8; // hypotheses per launch behind a frame's first batch, in units of cfg.ransac_batch (engine.cpp)
#ifndef PX_S_VALUE
#define PX_S_VALUE 5
#endif
constexpr int PX_S = PX_S_VALUE; // EKF_PRECISION_F32_EXACT: balanced base-256 digits per element of B (kernels_pexact.hip)
// rows of B up to which the int32 level sums of the exact downdate cannot wrap: PX_S products of |d d'| <= 2^14 per level and row
constexpr int PX_MAX_ROWS = ((1 << 17) / PX_S) / 32 * 32; // 26208 at PX_S = 5

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

// Row storage of P.  A rank keeps the 13 camera rows (replicated on every rank) at local rows 0..12 and the
// global rows [r0, r1) it owns from local row `base` on; every row holds ALL n columns.  Unsharded engine:
// r0 = base = 13, r1 = n, i.e. the identity.  Sharded: base = SHARD_BASE so that the owned block
// starts on a tile boundary of the downdate kernel.
constexpr int SHARD_BASE = 128;
struct RowMap {
    int r0, r1, base;
};
__host__ __device__ inline int local_row(const RowMap &m, int i) { return i < 13 ? i : m.base + (i - m.r0); }
__host__ __device__ inline bool owns_row(const RowMap &m, int i) { return i < 13 || (i >= m.r0 && i < m.r1); }

// integer slots of the device counter block
enum {
    CNT_NPRED = 0,    // predictions of the last full prediction
    CNT_NPRED_SUB,    // predictions of the last subset prediction
    CNT_NMATCH,       // matches produced by k_match
    CNT_ERR,          // sticky error flag (Cholesky breakdown, ...)
    CNT_RS_BEST,      // RANSAC: best support size
    CNT_RS_BESTH,     // RANSAC: hypothesis index of the best
    CNT_RS_NHYP,      // RANSAC: adaptive hypothesis bound
    CNT_RS_NEXT,      // RANSAC: next hypothesis to examine
    CNT_RS_DONE,      // RANSAC: loop finished
    CNT_NRESC,        // rescued count
    CNT_AUX0,         // (free)
    CNT_AUX1,
    CNT_SHARD0,       // sharded filter: CNT_SHARD0 + r = first entry of a feature-sorted match list that rank r owns
                      // (r = 0 .. world, at most 16 ranks: slots 12 .. 28; k_shard_bounds)
    CNT_COUNT = 32
};
constexpr int MAX_SHARD_WORLD = 16;

// doubles in the device state block
enum { ST_X = 0, ST_R = 13, ST_F = 32, ST_GQG = 32 + 169, ST_JN = 32 + 338, ST_COUNT = 32 + 338 + 16 };

struct DeviceArrays {
    // map + state
    double *state = nullptr;    // ST_COUNT doubles: x13, R, F, GQG, Jnorm
    double *feat_pos = nullptr; // 6 per feature
    int *feat_type = nullptr;
    int *feat_covpos = nullptr;
    uint8_t *feat_desc = nullptr;
    unsigned *feat_times_predicted = nullptr; // MapFeature::timesPredicted / timesMatched (EKF/MapFeature.h:72-73)
    unsigned *feat_times_matched = nullptr;
    void *P = nullptr; // T [ncap x ldP]
    void *P2 = nullptr; // second buffer of the same size, allocated on the first removal (compaction target)
    double *mm_scratch = nullptr; // map management scratch: Jpo/Jhr of a batch, 3 x ldP conversion rows, N linearity values
    int *mm_index = nullptr;      // new2old row map (ncap ints)
    // prediction tables, keyed by feature index
    int *pred_vis = nullptr;      // visibility per feature, latest prediction (full or subset)
    int *pred_vis_full = nullptr; // visibility per feature, last FULL prediction (the step's unseenFeatures)
    EkfPrediction *step_preds = nullptr; // snapshot of the last full prediction (ekf_keep_step_predictions)
    double *pred_uv = nullptr; // 2 per feature
    int *pred_vis2 = nullptr;    // scratch copies for state-only predictions
    double *pred_uv2 = nullptr;
    double *pred_S = nullptr;  // 4 per feature
    double *Hs = nullptr;      // 2x7 per feature
    double *Hf = nullptr;      // 2x6 per feature
    void *HP = nullptr;        // T [2*cap x ldP]: rows 2f, 2f+1 = H_f P   (fp64 in EKF_PRECISION_F32_EXACT)
    // work lists
    int *work_idx = nullptr;   // input feature indices of a subset prediction
    int *work_flag = nullptr;  // per work item: predicted?
    int *plist = nullptr;      // compacted feature indices, full prediction
    int *plist_sub = nullptr;  // compacted feature indices, subset prediction
    int *counts = nullptr;     // CNT_COUNT ints
    int *shard_feat = nullptr; // sharded filter: first feature of every rank, world + 1 ints
    // keypoints of the current frame
    EkfKeypoint *kps = nullptr;
    uint8_t *kdesc = nullptr;
    // matching
    int *mt_valid = nullptr;  // per prediction slot
    int *mt_kp = nullptr;
    float *mt_dist = nullptr;
    EkfKeypoint *mt_xy = nullptr; // NCC matcher: matched pixel per prediction slot
    uint8_t *tmpl = nullptr;      // NCC matcher: 3 levels x 121 bytes per feature
    double *gates = nullptr;      // new-feature detector: gate + centre + radius (8 doubles) per prediction of the last full prediction
    long long *cell_resp = nullptr; // detector: best response per 16x16 cell
    int *cell_xy = nullptr;         // detector: its pixel
    EkfMatch *matches = nullptr; // compacted matches (prediction order) / uploaded matches
    EkfMatch *msel = nullptr;    // matches selected for an update (inliers / rescued), update order
    EkfMatch *mout = nullptr;    // outlier matches
    EkfPrediction *preds_out = nullptr; // staging for prediction downloads
    int *match_of_feat = nullptr;
    // RANSAC
    int *hyp_count = nullptr;
    uint8_t *hyp_flags = nullptr;  // batch x mcap
    uint8_t *best_flags = nullptr; // mcap
    // update
    void *G = nullptr;   // T [(mcap + 1) x ldP] : rows of H P gathered for the selected matches
    void *A = nullptr;   // T [(mcap + 1) x ldP] : B = inv(L) G  (k-major operand of the downdate)
    double *S = nullptr;  // (mcap + slack) x ldS: lower triangle of S, updated in place by the sweep
    double *LL = nullptr; // same shape: L below the 32x32 diagonal blocks and L' mirrored above them
    float *LLf = nullptr; // fp32 covariance: the mirrored part (L') again in fp32, the A operand of the rows of B
    double *Tbuf = nullptr; // [mw x ldW] scratch of the doubling steps (T = L21 X11)
    double *nu = nullptr;
    double *Dinv = nullptr; // V = inv(L), row-major [mw x ldW], built up from 32x32 diagonal blocks by doubling
    double *W = nullptr;    // W = inv(L)' (upper triangular), row-major [mw x ldW]: the k-major operand of B = W' G
    float *Wf = nullptr;    // fp32 copy of W (fp32 configuration)
    double *mHs = nullptr;  // per selected match
    double *mHf = nullptr;
    int *mpos = nullptr;
    int *mdim = nullptr;
    double *dx_part = nullptr; // DX_SPLIT x ldP (+ 4: the quaternion as it was before the update)
    double *sq_part = nullptr; // DX_SPLIT x ldP: partial sums of squares of the columns of B (fp64 diagonal of B'B)
    double *diag_save = nullptr; // ldP: diagonal of P before a downdate
    double *cam_part = nullptr;  // DX_SPLIT x 13 x ldP: partial sums of the camera rows of B'B (fp64)
    double *cam_save = nullptr;  // 13 x ldP: camera rows of P before a downdate
    double *HPc = nullptr;       // [2 cap x 16]: fp64 camera columns of the H P rows
    double *Gc = nullptr;        // [(mcap + slack) x 16]: those of the gathered rows (working right-hand sides of the sweep)
    double *Bc = nullptr;        // [(mcap + slack) x 16]: inv(L) Gc, the fp64 camera columns of B
    double *zvec = nullptr;      // [mcap + slack]: z = inv(L) nu (nu itself is the sweep's working vector)
    double *yvec = nullptr;      // [mcap + slack]: y = inv(L)' z = inv(S) nu (fp32 configuration: dx = (H P)' y)
    uint8_t *mask = nullptr;   // generic byte mask output (rescue)
    void *pu_tilemap = nullptr; // int2 (ti, tj) per upper-triangle tile, XCD-friendly order
    void *sweep_ctl = nullptr;  // flags of the persistent Cholesky sweep (chol_persist.h), zeroed once
    unsigned *pu_ctr = nullptr; // exact downdate, two workgroups per CU (k_p_update_i8d): unit counters, two sets of eight (one per XCD list)
    int8_t *Bq = nullptr;       // EKF_PRECISION_F32_EXACT: PX_S digit planes of B, each [bq_rows / 16][ldP][16] bytes (kernels_pexact.hip)
    int *Bexp = nullptr;        // its column scales (biased exponents), ldP ints
    uint8_t *Bz = nullptr;      // ... and which 16-row x 32-column pieces of digit plane 0 hold anything but zeros: [column / 32][bz_stride] bytes
                                // (written with the planes' last reader before the downdate, k_dx_planes / k_slice_B; read by k_p_update_i8p)
    uint8_t *Wz = nullptr, *Gz = nullptr; // the same tables for the planes of inv(L)' and of G (B = inv(L) G above B_SWEEP_MAX rows)
    int8_t *Lq = nullptr;       // digit planes of L for the rows of B formed from planes (chol_bplanes.h): PX_S x lq_nbk^2 KB
    int *Lexp = nullptr;        // row scales of L (biased exponents)
    int *Grow = nullptr;        // row of the H P table behind every gathered row (k_gather without the copy)
    int8_t *Wq = nullptr, *Gq = nullptr; // digit planes of inv(L)' and of G for B = inv(L) G on the int8 MFMA (updates above B_SWEEP_MAX rows)
    int *Wexp = nullptr, *Gexp = nullptr; // their column scales
    float *Pdiag = nullptr;     // sharded exact configuration: diagonal of P, n floats, completed by an exchange
    int8_t *Bstage = nullptr;   // ... and the digit planes of B in the exchange layout [column][plane][k / 16][16]
};

// current frame of the NCC matcher: gray pyramid (level 0 = full resolution) + the raw upload staging buffer
struct Image {
    uint8_t *px[3] = {nullptr, nullptr, nullptr};
    uint8_t *px2[3] = {nullptr, nullptr, nullptr}; // second pyramid: the NEXT staged frame is reduced into it on stream2
    int prefetched = -1;                           // staged frame whose pyramid sits (or is being built) in px2
    int w[3] = {0, 0, 0}, h[3] = {0, 0, 0};
    uint8_t *raw = nullptr;
    size_t raw_cap = 0;
    bool valid = false;
    // staged sequence (ekf_images_upload)
    uint8_t *seq = nullptr;
    int seq_n = 0, seq_w = 0, seq_h = 0, seq_stride = 0, seq_channels = 0;
};

// Persistent Cholesky sweeps (chol_persist.h) need ALL their workgroups resident at once; two of them in flight on one device, each
// only partly resident, would wait for each other's workgroups until the watchdog ends both.  Engines of one process that share a
// device therefore share one registry (held by shared_ptr: it lives exactly as long as its engines): with more than one member every
// persistent launch is ordered behind the last persistent launch of any OTHER member by an event.  (Other processes on the device
// cannot be ordered; the watchdog and the automatic retry on the launch-per-panel sweep cover them.)
struct SweepRegistry {
    std::mutex mu;
    int device = 0;
    int members = 0;                  // engines attached
    hipEvent_t last = nullptr;        // end of the most recent persistent sweep recorded here
    const void *owner = nullptr;      // the engine that recorded it
    ~SweepRegistry()
    {
        if (last) (void)hipEventDestroy(last);
    }
};
std::shared_ptr<SweepRegistry> sweep_registry_attach(int device); // kernels_update.hip
void sweep_registry_detach(const std::shared_ptr<SweepRegistry> &reg, const void *engine);

struct Frames {
    int n = 0;
    std::vector<int> offset, count;
    EkfKeypoint *kps = nullptr;
    uint8_t *desc = nullptr;
};

} // namespace ekf

struct EkfEngine {
    EkfEngineConfig cfg;
    ekf::CamD cam;
    ekf::ParD par;
    int device = 0;
    int cap = 0, ncap = 0, mcap = 0, kcap = 0;
    int ldP = 0, ldS = 0, ldW = 0;
    int N = 0, n = 0;
    bool f32 = false;   // P stored in fp32 (and, unless `exact`, H P, its gathered rows and B)
    bool exact = false; // EKF_PRECISION_F32_EXACT: P in fp32; H P, G and B in fp64; downdate with exact accumulation (kernels_pexact.hip)
    int desc_bytes = EKF_DESC_BYTES; // bytes per descriptor row
    bool desc_f32 = false;           // CV_32F descriptors / L2 distance (Matching.cpp:60-73) instead of CV_8U / Hamming
    // row sharding: world == 1 means the whole matrix lives here
    int shard_rank = 0, shard_world = 1;
    int p_rows_cap = 0;                  // rows allocated for P
    ekf::RowMap rm{13, 13, 13};          // refreshed by set_state / map management
    std::vector<int> shard_feat_begin;   // [world + 1] first feature of each rank
    EkfExchangeFn xchg = nullptr;        // all-gather of per-feature row blocks between the ranks
    void *comm = nullptr;                // ncclComm_t of the in-engine transport (ekf_comm_init), or null
    bool hp_complete = true;             // sharded: the H.P table holds EVERY rank's rows since the last prediction
    int (*after_gather)(EkfEngine *, int) = nullptr; // sharded step: completes the gathered rows right after k_gather
    // the row-block exchange of engine.cpp (RCCL send / recv or the host callback), callable from the update's launcher
    int (*exchange_hook)(EkfEngine *, int what, void *base, size_t row_bytes, const std::vector<int32_t> &rb, const char *name) = nullptr;
    std::vector<int32_t> shard_row_begin; // [world + 1] first state row of each rank's share (0 for rank 0: it also holds the camera's), n at the end
    long long xchg_bytes_planes = 0;      // bytes of digit planes this rank received (tests assert them against the model)
    int hook_rc = 0;                     // its status (launch_update returns nothing)
    std::vector<int32_t> shard_rb;       // row boundaries of the gathered rows by owner (from CNT_SHARD0..)
    std::vector<int32_t> slot_rb;        // sharded step: prediction slots of the last full prediction by owner (from CNT_SHARD0..)
    std::vector<int32_t> last_col_rb;    // column shares of the planes of B used by the last sharded update (ekf_shard_counters)
    void *xchg_user = nullptr;
    int n_cus = 256;           // compute units of the device (launch-shape decisions)
    int b_path = 0;            // ekf_set_update_path: 0 by size (B_SWEEP_MAX), 1 B in the sweep, 2 inverse + GEMM
    int sweep_mode = 2;        // ekf_set_sweep_mode: EKF_SWEEP_* (2 AUTO: one persistent launch per update where it applies, chol_persist.h)
    unsigned ps_epoch = 0;                // persistent sweep: epoch of its flags
    int ps_fault = 0;                     // include/ekf_test_hooks.h: countdown; the persistent sweep that takes it to zero runs without its chain workgroup
    std::shared_ptr<ekf::SweepRegistry> ps_reg; // the device's registry of persistent sweeps (shared with the other engines on it)
    // automatic retry of an update whose persistent sweep timed out (EKF_ERR_TIMEOUT): the kernels behind a failed sweep leave the
    // filter untouched (filter_frozen), so the update is run again from the same P on the launch-per-panel sweep
    int force_launches = 0;               // > 0: updates use the launch-per-panel sweep whatever sweep_mode says (the retry itself)
    int sweep_retries = 0;                // updates re-run this way since the engine was created (ekf_get_sweep_retries)
    // back-off (EKF_SWEEP_AUTO only): a device that another process shares makes persistent sweeps time out again and again, 5 ms each;
    // after a time-out the next ps_backoff updates take the launch-per-panel sweep, twice as many after every further time-out (64 ..
    // 4096), and 256 persistent sweeps in a row without one forget the history
    int ps_backoff = 0, ps_backoff_len = 0, ps_ok_streak = 0;
    int last_update_M = 0;                // matches of the last update enqueued (its list is still in d.msel)
    bool last_update_cov = true;          // ... a covariance update (false: updateOnlyState)
    bool last_update_sym = false;         // p_exact_sym as it was before that update
    bool last_update_persist = false;     // that update's sweep was the persistent launch
    int ps_cap[2] = {0, 0};    // resident workgroups of k_chol_persist<false / true> on this device (0: not asked yet, -1: unusable)
    bool async_errors = false; // ekf_set_async_errors: no read-back at the end of a step
    bool p_exact_sym = false; // P known to be bitwise symmetric (engine-maintained invariant)
    int n_pred = 0;           // predictions of the last full prediction
    int n_gates = 0;          // gates snapshotted for the new-feature detector
    bool keep_step_preds = false; // snapshot every full prediction for ekf_get_step_predictions
    int n_step_preds = 0;
    int cells_cap = 0;        // detector cell buffers allocated for this many cells
    int n_kp = 0;
    long long pu_tilemap_nt = -1;
    std::map<long long, std::pair<void *, int>> pu_tables; // built work lists of the downdate: key -> (device list, units per XCD)
    int pu_per_xcd = 0;
    int bq_rows = 0;          // rows of B a digit plane holds (multiple of 64)
    int bz_stride = 0;        // 16-row groups per column block of d.Bz (= bq_rows / 16)
    bool px_dense = false;    // include/ekf_test_hooks.h: the downdate and the int8 GEMM multiply every digit product (the tables are not consulted)
    int px_scale_shift = 0;   // bits of head-room added to the a-priori column scales of B (exponent of sqrt(P_jj)); EKF_PX_SCALE_SHIFT in the
                              // environment sets it (a measurement knob).  With 1 / 2 bits digit plane 0 is 82 / 95-97 % zero pieces on converged
                              // maps too and the downdate skips their products: 1187 -> 1266 updates/s at N = 1000 -- but the 38-bit integers are
                              // then 37- / 36-bit ones, and the far features' inverse depths need every bit: one bit puts one of the five
                              // N = 1000 scenes at 1.12e-5 component-wise, two bits N = 2000 at 1.9e-5 (profiles/r06_scale_shift.txt).  Left at 0.
    int lq_nbk = 0;           // 32-row blocks per side of the digit planes of L
    int bstage_rows = 0;      // rows of B the exchange image of the planes holds (sharded exact configuration)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> px_events; // exact configuration: end of the sweep -> start of the downdate (inverse, GEMM, digit planes, dx, state)
    hipEvent_t px_mid = nullptr;                               // recorded after the sweep's last launch (timing only)
    int pu_slots = 0;         // resident workgroups of the downdate kernel on this device (0: not asked yet, -1: unknown)
    int pu_parity = 0;        // k_p_update_i8d: the counter set the next launch uses (it zeroes the other one)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;             // image-only work of the next frame, overlapped with the update
    hipEvent_t ev_main = nullptr, ev_prefetch = nullptr;
    ekf::DeviceArrays d;
    ekf::Frames frames;
    ekf::Image img;
    std::string err;
    // timing
    bool timing = false;
    EkfStageTimes times{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pu_events; // P-update kernel brackets not yet harvested
    std::vector<double> pu_work;                               // n^2 m of each bracket
    std::vector<int> pu_m;                                     // m of each bracket
    std::vector<std::pair<int, float>> pu_log;                 // harvested (m, ms) per launch
    std::vector<std::pair<hipEvent_t, hipEvent_t>> sw_events;  // brackets of the Cholesky sweep's launches of one update
    std::vector<int> sw_m;                                     // m of each bracket
    std::vector<int> sw_launch;                                // launches of each bracket (pair launches cover two panels)
    double sweep_ms = 0.0, sweep_flops_f64 = 0.0, sweep_flops_b = 0.0; // harvested totals (ekf_timing_sweep)
    long long sweep_panels = 0, sweep_updates = 0, sweep_launches = 0;
    double slice_ms = 0.0;                                     // exact configuration: sweep end -> downdate start (harvested)
    // host scratch
    std::vector<int> h_counts;
    int *h_mirror = nullptr, *d_mirror = nullptr; // GPU-writable host page: counters + sequence number (read_counts)
    int mirror_seq = 0;
    std::vector<int> h_type, h_covpos; // host mirror of the map layout (type, covariance position per feature)
};

namespace ekf {

// C[i][j] = alpha sum_k X[k][i] Y[k][j] (kernels_gemm.hip); batch element b adds b * (xb, yb, cb, ctb) elements
struct XtyArgs {
    const void *X; int ldx; long long xb;
    const void *Y; int ldy; long long yb;
    void *C; int ldc; long long cb;       // may be null
    void *Ct; int ldct; long long ctb;    // transposed copy in fp64, may be null
    float *Ctf;                           // transposed copy in fp32 (same layout as Ct), may be null
    int M, N, K;                          // output rows / columns / k-depth per batch element
    int row0_first, row0_stride, m_lim;   // rows of element b that exist: min(M, m_lim - (row0_first + b row0_stride))
    int tri;                              // 0: all k; 1: Y[k][j] = 0 for k < j; 2: X[k][i] = 0 for k > i
    int tiles_i, tiles_j;                 // row tiles / column tiles
    int tj0;                              // first column tile (row-sharded engines form their own columns of B only)
    int n_split;                          // bottom row tiles cut into two half units (tri == 2, batch 1 only)
    double alpha;
};
void launch_xty(EkfEngine *e, const XtyArgs &a, int batch, bool f32, hipStream_t stream);

// A failed factorisation of S (EKF_ERR_NOT_POSITIVE_DEFINITE) or a timed-out persistent sweep (EKF_ERR_TIMEOUT) leaves its code in
// counts[CNT_ERR] until the host has read it.  While it is set the filter is FROZEN: the kernels that write the state, the covariance
// or the map's bookkeeping return at once (the downdate and the state update of the failed update itself, and whatever of the
// following stages was enqueued before the host looked), so the host finds x, P and the map exactly as they were before the failed
// update and can run it again (time-out) or go on without it (S not positive definite: the reference's cv::invert returns zeros
// there, i.e. K = 0 and an unchanged filter, EKF/Update.cpp:101-108).  counts == nullptr: no guard (stage calls that synchronise).
#ifdef __HIPCC__
__device__ __forceinline__ bool filter_frozen(const int *counts) { return counts != nullptr && counts[CNT_ERR] != 0; }
#endif

// Exclusive prefix sum of one int per thread over a 1024-thread workgroup (and the total): inside a wavefront by shuffles,
// across the 16 wavefronts through LDS -- one barrier, where a Hillis-Steele scan in LDS needs twenty.  Device code only.
#ifdef __HIPCC__
__device__ __forceinline__ int block_exclusive_scan_1024(int c, int *wtot /* __shared__ int[16] */, int *total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wtot[wv] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = wtot[w];
        base += w < wv ? t : 0;
        tot += t;
    }
    *total = tot;
    return base + x - c;
}
#endif

// ---- launchers (kernels_*.hip) ; T selected by e->f32 ----------------------------------------------------
void launch_predict(EkfEngine *e);
// full (idx == nullptr) or subset prediction; fills tables, compacted list and CNT_NPRED / CNT_NPRED_SUB
bool launch_predict_features(EkfEngine *e, const int *d_idx, int count, bool state_only, bool defer_compact = false);
bool launch_predict_with_features(EkfEngine *e, int count); // step path: prepare, then covariance strips + all features in one launch
// d_count != nullptr: n_list is an upper bound, the list's length is read on the device
void launch_hp_rows(EkfEngine *e, const int *d_list, int n_list, bool count_predicted = false, const int *d_count = nullptr, bool from_flags = false);
// d_npred != nullptr: n_pred is an upper bound, the number of predictions is read on the device
void launch_match(EkfEngine *e, int n_pred, int n_kp, const int *d_npred = nullptr, bool with_ransac_init = false);
// d_M != nullptr (RANSAC launchers): M is an upper bound, the number of matches is read on the device
void launch_match_index(EkfEngine *e, int M, const int *d_M = nullptr);
// sharded filter: per-rank boundaries of a feature-sorted match list -> counts[CNT_SHARD0 ..]
void launch_shard_bounds(EkfEngine *e, const EkfMatch *list, int count);
void launch_shard_bounds_idx(EkfEngine *e, const int *list, const int *d_count); // ... of a feature-index list whose length is on the device
// sharded filter: matching and RANSAC hypotheses divided by feature ownership (kernels_match.hip, kernels_ncc.hip, kernels_ransac.hip)
void launch_match_slots(EkfEngine *e, int n_kp, int s_lo, int s_hi);
void launch_match_ncc_slots(EkfEngine *e, int s_lo, int s_hi);
void launch_match_compact(EkfEngine *e, int n_pred);
void launch_match_compact_slots(EkfEngine *e, int n_pred, const EkfKeypoint *d_slot_xy);
void launch_ransac_hyp(EkfEngine *e, int M, int h0, int batch, const int *d_M, int h_lo, int h_hi);
void launch_ransac_select(EkfEngine *e, int M, int h0, int batch, const int *d_M, int publish_seq);
// publish_seq > 0: the batch's bookkeeping kernel also publishes the counter block (see publish_counts_block)
void launch_ransac_batch(EkfEngine *e, int M, int h0, int batch, const int *d_M = nullptr, int publish_seq = 0);
void launch_ransac_init(EkfEngine *e, int M);
void launch_update(EkfEngine *e, int M, bool update_cov);
void launch_p_update_exact(EkfEngine *e, int m, bool use_bc, bool exps_ready = false, bool planes_ready = false); // kernels_pexact.hip
void launch_round_P_f32(EkfEngine *e);                        // kernels_map.hip
void launch_diag_extract(EkfEngine *e, float *diag);          // kernels_pexact.hip: sharded exact configuration
void launch_planes_move(EkfEngine *e, bool pack, int m_k, int c_lo, int c_hi, int skip_lo, int skip_hi);
void launch_dx_planes(EkfEngine *e, int m_k);
void launch_slice_columns(EkfEngine *e, int m, int c_lo, int c_hi);
void launch_b_gemm_planes(EkfEngine *e, int m, int c_lo, int c_hi);
void launch_rescue(EkfEngine *e, int M);
void launch_state_only_predict(EkfEngine *e, EkfPrediction *d_out); // predictMeasurementState on current state
void launch_add_features(EkfEngine *e, const double *d_uv, int count, double *d_Jpo, double *d_Jhr);
void launch_compact_P(EkfEngine *e, int n_new, const int *d_new2old);
void launch_linearity(EkfEngine *e, double *d_out);
void launch_convert(EkfEngine *e, int fi, int pos, double *d_J, double *d_T3);
void launch_ncc_pyramid(EkfEngine *e, const uint8_t *d_raw, int stride, int channels);
void launch_ncc_pyramid_on(EkfEngine *e, hipStream_t stream, uint8_t *const px[3], const uint8_t *d_raw, int stride, int channels);
void launch_ncc_capture(EkfEngine *e, const int *d_idx, const double *d_uv, int count);
void launch_match_ncc(EkfEngine *e, int n_pred);
void launch_gate_snapshot(EkfEngine *e, int n_pred);
void launch_detect_cells(EkfEngine *e, int n_gates, int cells_x, int cells_y, long long *d_resp, int *d_xy);
void launch_publish_counts(EkfEngine *e, int *d_mirror, int seq);
// The same publication from inside a kernel that ends a stage (saves the separate launch): called by EVERY thread of a
// block after the block's last write to `counts`; lanes 0..15 copy the counters to the GPU-writable host page, lane 0
// then releases the sequence number the host polls.
__device__ __forceinline__ void publish_counts_block(const int *counts, int *mirror, int seq)
{
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 64) { // first wavefront
        if (t < CNT_COUNT) __hip_atomic_store(&mirror[t], counts[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __builtin_amdgcn_wave_barrier();
        if (t == 0) {
            __threadfence_system();
            __hip_atomic_store(&mirror[CNT_COUNT], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
void launch_pack_predictions(EkfEngine *e, const int *d_list, int n, EkfPrediction *d_out);

} // namespace ekf
