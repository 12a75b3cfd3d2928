// kernels_ransac.hip -- A6/A8, 1-point RANSAC (EKF/1PointRansac.cpp:101-234) and A9 outlier rescue
// (EKF/EKF.cpp:68-119).
//
// The reference evaluates hypotheses one at a time and re-reads all of P for each (dense P H_i', m = 2:
// Update.cpp:105 via updateOnlyState).  Here the H_i P row pairs cached by k_hp_rows ARE those gain columns
// (P symmetric), so a hypothesis costs no pass over P at all: one workgroup per hypothesis forms
// dx = (H_i P)' inv(S_i) nu_i, applies it with the reference's dead-band to a private copy of the camera state and
// to every feature, re-projects all N features and counts the matches inside the pixel threshold.  A batch of
// hypotheses runs in one launch; a single small block then replays the reference's sequential bookkeeping
// (strict improvement, adaptive hypothesis bound) in hypothesis order, so the winner equals the sequential loop's.
#include "engine.h"

namespace ekf {

constexpr int HYP_THREADS = 512;  // threads of a hypothesis' workgroup

template <typename T>
__global__ void __launch_bounds__(HYP_THREADS)
k_ransac_hyp(const double *st, CamD cam, double thr, const double *feat_pos, const int *feat_type,
             const int *feat_covpos, int N, const T *HP, int ld, const double *uv_tab, const double *S_tab,
             const EkfMatch *matches, int M, const int *match_of_feat, int h0, int *hyp_count, uint8_t *hyp_flags,
             int mcap, const int *d_M, int h_lo, int h_hi)
{
    __shared__ double sx[13], sRt[9], sRinv[9], sw[2];
    __shared__ int s_cnt[HYP_THREADS / 64];
    if (d_M) M = *d_M; // the number of matches is only known on the device (step path without read-backs)
    const int h = h0 + blockIdx.x;
    if (h >= M) return;
    // sharded filter: this rank evaluates the hypotheses [h_lo, h_hi) -- those of the features it owns, whose H P rows (the gain
    // columns) it holds; the others' support counts and masks arrive by an all-gather (engine.cpp: ransac_dev)
    if (h < h_lo || h >= h_hi) return;
    const int tid = threadIdx.x;
    const int fi = matches[h].featureIndex;
    const T *g0 = HP + (size_t)(2 * fi) * ld;
    const T *g1 = g0 + ld;
    if (tid == 0) {
        // S = H (P H') + R with R = I * pixelErrorX (Update.cpp:95-107); S_tab holds H P H' + I
        double S[4] = {S_tab[4 * fi] - 1.0 + cam.pixelErrorX, S_tab[4 * fi + 1], S_tab[4 * fi + 2],
                       S_tab[4 * fi + 3] - 1.0 + cam.pixelErrorX};
        double Si[4];
        inv2(S, Si);
        const double ax = matches[h].imagePos[0] - uv_tab[2 * fi];
        const double ay = matches[h].imagePos[1] - uv_tab[2 * fi + 1];
        const double n0 = fabs(ax) > EKF_DELTA ? ax : 0.0; // Update.cpp:133-134
        const double n1 = fabs(ay) > EKF_DELTA ? ay : 0.0;
        const double w0 = Si[0] * n0 + Si[1] * n1;
        const double w1 = Si[2] * n0 + Si[3] * n1;
        sw[0] = w0;
        sw[1] = w1;
        double x[13];
        for (int i = 0; i < 13; ++i) {
            const double dx = (double)g0[i] * w0 + (double)g1[i] * w1;
            x[i] = st[ST_X + i] + (fabs(dx) > EKF_DELTA ? dx : 0.0); // Update.cpp:150-186
            sx[i] = x[i];
        }
        double R[9];
        quat_to_rot(x + 3, R); // setOrientation without normalising, Update.cpp:168
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) sRt[j * 3 + i] = R[i * 3 + j];
        inv3(R, sRinv);
    }
    __syncthreads();
    const double w0 = sw[0], w1 = sw[1];
    uint8_t *flags = hyp_flags + (size_t)blockIdx.x * mcap;
    const int nthr = blockDim.x; // 256 for small maps, HYP_THREADS otherwise
    for (int k = tid; k < M; k += nthr) flags[k] = 0; // this hypothesis' support mask (only matched features are written below)
    __syncthreads();
    int cnt = 0;
    for (int f = tid; f < N; f += nthr) { // the loop body is a chain of dependent loads: few iterations per thread (1024 threads would spill)
        const int mi = match_of_feat[f];
        if (mi >= M) continue; // features without a match cannot add support (1PointRansac.cpp:60-82)
        const int type = feat_type[f], pos = feat_covpos[f], d = feat_dim(type);
        double fp[6];
        for (int a = 0; a < 6; ++a) fp[a] = feat_pos[6 * f + a];
        for (int a = 0; a < d; ++a) {
            const double dx = (double)g0[pos + a] * w0 + (double)g1[pos + a] * w1;
            if (fabs(dx) > EKF_DELTA) fp[a] += dx; // Update.cpp:190-204
        }
        double uv[2];
        int ok = 0;
        if (predict_pixel(cam, sx, sRt, sRinv, fp, type, uv)) {
            const double xd = matches[mi].imagePos[0] - uv[0];
            const double yd = matches[mi].imagePos[1] - uv[1];
            ok = sqrt(xd * xd + yd * yd) < thr ? 1 : 0;
        }
        flags[mi] = (uint8_t)ok;
        cnt += ok;
    }
    cnt = wave_sum_i(cnt);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < nthr / 64; ++w) tot += s_cnt[w];
        hyp_count[blockIdx.x] = tot;
    }
}

// Sequential bookkeeping of the hypothesis loop (1PointRansac.cpp:125,164-178) over one batch.  The decisions depend only on the
// batch's support counts and the loop state, and of the inlier masks only the LAST improving hypothesis' survives: every thread
// replays the loop in registers (no barrier, no trip to memory per hypothesis -- as one thread with the state in global memory and
// two barriers per hypothesis this kernel took 7 us), then the workgroup copies that one mask and thread 0 writes the state back.
__global__ void __launch_bounds__(256)
k_ransac_select(int *counts, const int *hyp_count, const uint8_t *hyp_flags, uint8_t *best_flags, int M, int h0,
                int batch, int mcap, double prob, const int *d_M, int *mirror, int publish_seq)
{
    if (d_M) M = *d_M;
    const int tid = threadIdx.x;
    int best = counts[CNT_RS_BEST], besth = counts[CNT_RS_BESTH], nhyp = counts[CNT_RS_NHYP], next = counts[CNT_RS_NEXT];
    __syncthreads(); // (everyone has read the state thread 0 rewrites below)
    int last = -1;
    bool stop = false, done = false;
    for (int b = 0; b < batch; ++b) {
        const int i = h0 + b;
        if (!((unsigned)i < (unsigned)nhyp && i < M)) {
            stop = true;
            done = true;
            break;
        }
        const int ns = hyp_count[b];
        if (ns > best) {
            best = ns;
            besth = i;
            const double e = 1.0 - (double)ns / (double)M;
            nhyp = (int)(log(1.0 - prob) / log(1.0 - (1.0 - e)));
            last = b;
        }
        next = i + 1;
    }
    if (!stop) {
        const int i = h0 + batch;
        if (!((unsigned)i < (unsigned)nhyp && i < M)) done = true;
    }
    if (last >= 0)
        for (int k = tid; k < M; k += 256) best_flags[k] = hyp_flags[(size_t)last * mcap + k];
    if (tid == 0) {
        counts[CNT_RS_BEST] = best;
        counts[CNT_RS_BESTH] = besth;
        counts[CNT_RS_NHYP] = nhyp;
        counts[CNT_RS_NEXT] = next;
        if (done) counts[CNT_RS_DONE] = 1;
    }
    if (publish_seq > 0) publish_counts_block(counts, mirror, publish_seq);
}

// loop state of the hypothesis loop, "no match" in match_of_feat (N entries), zeroed best mask (M entries, M <= N)
__global__ void __launch_bounds__(256) k_ransac_init(int *counts, int *match_of_feat, uint8_t *best_flags, int N, int M)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        counts[CNT_RS_BEST] = 0;
        counts[CNT_RS_BESTH] = -1;
        counts[CNT_RS_NHYP] = 1000; // numberOfHipotesis, 1PointRansac.cpp:116
        counts[CNT_RS_NEXT] = 0;
        counts[CNT_RS_DONE] = 0;
    }
    if (i < N) match_of_feat[i] = 0x7fffffff;
    if (i < M) best_flags[i] = 0;
}

void launch_ransac_init(EkfEngine *e, int M)
{
    const int n = max(max(e->N, M), 1);
    k_ransac_init<<<(n + 255) / 256, 256, 0, e->stream>>>(e->d.counts, e->d.match_of_feat, e->d.best_flags, e->N, M);
}

// the hypotheses [h0, h0 + batch) restricted to [h_lo, h_hi) (whole batch: 0, INT_MAX)
void launch_ransac_hyp(EkfEngine *e, int M, int h0, int batch, const int *d_M, int h_lo, int h_hi)
{
    const int nb = batch; // hyp_flags rows are zeroed by their workgroups, hyp_count[b] is only read for launched hypotheses
    const double thr = e->cfg.par.ransacThresholdPredictDistance;
    if (e->f32 && !e->exact) // (EKF_PRECISION_F32_EXACT keeps H P in fp64)
        k_ransac_hyp<float><<<nb, e->N > 256 ? HYP_THREADS : 256, 0, e->stream>>>(e->d.state, e->cam, thr, e->d.feat_pos, e->d.feat_type,
                                                       e->d.feat_covpos, e->N, (const float *)e->d.HP, e->ldP,
                                                       e->d.pred_uv, e->d.pred_S, e->d.matches, M,
                                                       e->d.match_of_feat, h0, e->d.hyp_count, e->d.hyp_flags,
                                                       e->mcap, d_M, h_lo, h_hi);
    else
        k_ransac_hyp<double><<<nb, e->N > 256 ? HYP_THREADS : 256, 0, e->stream>>>(e->d.state, e->cam, thr, e->d.feat_pos, e->d.feat_type,
                                                        e->d.feat_covpos, e->N, (const double *)e->d.HP, e->ldP,
                                                        e->d.pred_uv, e->d.pred_S, e->d.matches, M,
                                                        e->d.match_of_feat, h0, e->d.hyp_count, e->d.hyp_flags,
                                                        e->mcap, d_M, h_lo, h_hi);
}

void launch_ransac_select(EkfEngine *e, int M, int h0, int batch, const int *d_M, int publish_seq)
{
    k_ransac_select<<<1, 256, 0, e->stream>>>(e->d.counts, e->d.hyp_count, e->d.hyp_flags, e->d.best_flags, M, h0,
                                              batch, e->mcap, e->cfg.par.ransacAllInliersProbability, d_M, e->d_mirror,
                                              e->d_mirror ? publish_seq : 0);
}

void launch_ransac_batch(EkfEngine *e, int M, int h0, int batch, const int *d_M, int publish_seq)
{
    launch_ransac_hyp(e, M, h0, batch, d_M, 0, 0x7fffffff);
    launch_ransac_select(e, M, h0, batch, d_M, publish_seq);
}

// ------------------------------------------------------------------------------------------------------ A9
// nu' inv(S_i) nu < chi2 (EKF.cpp:84-97) against the predictions of the last subset prediction.
__global__ void __launch_bounds__(256)
k_rescue(const EkfMatch *m, int M, const int *vis, const double *uv_tab, const double *S_tab, double chi2,
         uint8_t *mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int fi = m[i].featureIndex;
    const double d0 = m[i].imagePos[0] - uv_tab[2 * fi];
    const double d1 = m[i].imagePos[1] - uv_tab[2 * fi + 1];
    double Si[4];
    inv2(S_tab + 4 * fi, Si);
    const double t0 = d0 * Si[0] + d1 * Si[2];
    const double t1 = d0 * Si[1] + d1 * Si[3];
    const double v = t0 * d0 + t1 * d1;
    mask[i] = (vis[fi] && v < chi2) ? 1 : 0;
}

void launch_rescue(EkfEngine *e, int M)
{
    if (M <= 0) return;
    k_rescue<<<(M + 255) / 256, 256, 0, e->stream>>>(e->d.matches, M, e->d.pred_vis, e->d.pred_uv, e->d.pred_S,
                                                     e->cfg.par.ransacChi2Threshold, e->d.mask);
}

} // namespace ekf
