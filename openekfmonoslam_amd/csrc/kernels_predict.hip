// kernels_predict.hip -- prediction stages on gfx950:
//   A1/A2  stateAndCovariancePrediction   EKF/StateAndCovariancePrediction.cpp:43-65,154-253
//   A3     predictMeasurementState         EKF/MeasurementPrediction.cpp:203-265
//   A4     predictMeasurementCovariance    EKF/MeasurementPrediction.cpp:595-700
// All of it is HBM/latency-bound strip work over the (13+6N)-wide covariance: coalesced row reads, fp64 math,
// no GEMM shapes.
#include "engine.h"

namespace ekf {

__device__ void predict_prepare_serial(double *st, double *sG);

// ------------------------------------------------------------------------------------------------ A1 + A2
// F (13x13), G Q G' (13x13) from the PRE-prediction state, then the state prediction itself (covariance is predicted before
// the state, :251-252; dt = 1, :246).  One workgroup: thread 0 forms F, G and the predicted state (a chain of a few hundred dependent
// fp64 operations); the 169 entries of G Q G' -- a thousand multiply-adds, half of the 9 us the kernel took as one thread -- are dealt
// to the workgroup behind one barrier.
// counts != nullptr (EKF::step's launch): a step enqueued behind a failed update leaves the filter as it is (engine.h: filter_frozen)
__global__ void __launch_bounds__(256) k_predict_prepare(double *st, ParD par, const int *counts)
{
    __shared__ double sG[13 * 6];
    if (filter_frozen(counts)) return;
    const double dt = 1.0;
    const double ln = par.linearAccelSD * par.linearAccelSD * dt * dt;
    const double an = par.angularAccelSD * par.angularAccelSD * dt * dt;
    if (blockIdx.x != 0) return;
    if (threadIdx.x == 0) predict_prepare_serial(st, sG);
    __syncthreads();
    if (threadIdx.x < 169) {
        const int i = threadIdx.x / 13, j = threadIdx.x % 13;
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += (sG[i * 6 + k] * (k < 3 ? ln : an)) * sG[j * 6 + k];
        st[ST_GQG + i * 13 + j] = s;
    }
}

__device__ void predict_prepare_serial(double *st, double *sG)
{
    const double dt = 1.0;
    double *x = st + ST_X;
    double *F = st + ST_F, *GQG = st + ST_GQG;
    for (int i = 0; i < 169; ++i) F[i] = 0.0;
    for (int i = 0; i < 13; ++i) F[i * 13 + i] = 1.0;
    for (int i = 0; i < 3; ++i) F[i * 13 + i + 7] = dt;
    // d q_new / d q = right-multiplication matrix of quat(w dt)  (:71-92)
    double w[3] = {x[10] * dt, x[11] * dt, x[12] * dt};
    double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    double qr[4];
    if (nw < EKF_EPSILON) {
        qr[0] = 1; qr[1] = 0; qr[2] = 0; qr[3] = 0;
    } else {
        const double s = sin(nw / 2);
        qr[0] = cos(nw / 2); qr[1] = s * w[0] / nw; qr[2] = s * w[1] / nw; qr[3] = s * w[2] / nw;
    }
    {
        const double qw = qr[0], qx = qr[1], qy = qr[2], qz = qr[3];
        const double Fq[16] = {qw, -qx, -qy, -qz, qx, qw, qz, -qy, qy, -qz, qw, qx, qz, qy, -qx, qw};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) F[(3 + i) * 13 + 3 + j] = Fq[i * 4 + j];
    }
    double G[13 * 6];
    for (int i = 0; i < 78; ++i) G[i] = 0.0;
    if (fabs(x[10]) < EKF_EPSILON && fabs(x[11]) < EKF_EPSILON && fabs(x[12]) < EKF_EPSILON) {
        for (int i = 0; i < 3; ++i) F[(i + 10) * 13 + i + 10] = 0.0; // :176-184, G quaternion block stays 0
    } else {
        const double om = sqrt(x[10] * x[10] + x[11] * x[11] + x[12] * x[12]);
        const double *q = x + 3;
        const double Qm[16] = {q[0], -q[1], -q[2], -q[3], q[1], q[0], -q[3], q[2],
                               q[2], q[3], q[0], -q[1], q[3], -q[2], q[1], q[0]};
        const double sh = sin(om * dt / 2.0), ch = cos(om * dt / 2.0);
        double D[12];
        for (int a = 0; a < 3; ++a) {
            const double wa = x[10 + a];
            D[a] = (-dt / 2.0) * (wa / om) * sh; // :100-103
            for (int b = 0; b < 3; ++b) {
                const double wb = x[10 + b];
                double v;
                if (a == b) // :107-111
                    v = (dt / 2.0) * wa * wa / (om * om) * ch + (1.0 / om) * (1.0 - wa * wa / (om * om)) * sh;
                else // :115-119
                    v = (wa * wb / (om * om)) * ((dt / 2.0) * ch - (1.0 / om) * sh);
                D[3 + a * 3 + b] = v;
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += Qm[i * 4 + k] * D[k * 3 + j];
                F[(3 + i) * 13 + 10 + j] = s;
                G[(3 + i) * 6 + 3 + j] = s;
            }
    }
    for (int i = 0; i < 3; ++i) {
        G[(i + 7) * 6 + i] = 1.0;
        G[(i + 10) * 6 + i + 3] = 1.0;
        G[i * 6 + i] = 1.0 * dt;
    }
    for (int i = 0; i < 78; ++i) sG[i] = G[i];
    (void)GQG;
    // predictState :43-65
    for (int i = 0; i < 3; ++i) x[i] += x[7 + i] * dt;
    {
        const double q1w = x[3], q1x = x[4], q1y = x[5], q1z = x[6];
        const double q2w = qr[0], q2x = qr[1], q2y = qr[2], q2z = qr[3];
        x[3] = q1w * q2w - q1x * q2x - q1y * q2y - q1z * q2z;
        x[4] = q1w * q2x + q1x * q2w + q1y * q2z - q1z * q2y;
        x[5] = q1w * q2y - q1x * q2z + q1y * q2w + q1z * q2x;
        x[6] = q1w * q2z + q1x * q2y - q1y * q2x + q1z * q2w;
    }
    quat_to_rot(x + 3, st + ST_R);
}

// P[0:13,0:13] = F P F' + G Q G'; P[0:13,13:] = F P[0:13,13:]; P[13:,0:13] = P[13:,0:13] F'  (:226-239).
// Block 0 owns the corner; every other thread owns one column j of the row strip and row j of the column strip
// (13 coalesced loads + 13 contiguous loads).  The three regions are disjoint, so one launch updates in place.
// Sharded storage (RowMap): the camera rows are replicated, so every rank computes the whole row strip; the column
// strip only exists for the rows a rank owns.  Both strips are the same arithmetic on bitwise-equal operands, which
// keeps P[a][j] on one rank identical to P[j][a] on the owner of row j.
// (the body takes its workgroup index as an argument: k_predict_cov_features runs it beside the pixel predictions in ONE launch)
template <typename T>
__device__ __forceinline__ void predict_cov_body(const int bx, T *P, int ld, int n, const double *st, RowMap rm)
{
    __shared__ double sF[169];
    __shared__ double sC[169];
    __shared__ double sFP[169];
    const int tid = threadIdx.x;
    for (int i = tid; i < 169; i += 256) sF[i] = st[ST_F + i];
    if (bx == 0) {
        for (int i = tid; i < 169; i += 256) sC[i] = (double)P[(size_t)(i / 13) * ld + (i % 13)];
        __syncthreads();
        if (tid < 169) {
            const int i = tid / 13, j = tid % 13;
            double s = 0.0;
            for (int k = 0; k < 13; ++k) s += sF[i * 13 + k] * sC[k * 13 + j];
            sFP[tid] = s;
        }
        __syncthreads();
        if (tid < 169) { // upper triangle, mirrored: F C F' + G Q G' is symmetric; this keeps P bitwise symmetric
            const int i = tid / 13, j = tid % 13;
            if (i <= j) {
                double s = 0.0;
                for (int k = 0; k < 13; ++k) s += sFP[i * 13 + k] * sF[j * 13 + k];
                const T v = (T)(s + st[ST_GQG + tid]);
                P[(size_t)i * ld + j] = v;
                P[(size_t)j * ld + i] = v;
            }
        }
        return;
    }
    __syncthreads();
    const int j = 13 + (bx - 1) * 256 + tid;
    if (j >= n) return;
    const bool mine = owns_row(rm, j);
    T *prow = P + (size_t)local_row(rm, j) * ld;
    double col[13], row[13];
#pragma unroll
    for (int a = 0; a < 13; ++a) col[a] = (double)P[(size_t)a * ld + j];
#pragma unroll
    for (int a = 0; a < 13; ++a) row[a] = mine ? (double)prow[a] : 0.0;
#pragma unroll
    for (int a = 0; a < 13; ++a) {
        double s = 0.0, t = 0.0;
#pragma unroll
        for (int b = 0; b < 13; ++b) {
            s += sF[a * 13 + b] * col[b];
            t += row[b] * sF[a * 13 + b];
        }
        P[(size_t)a * ld + j] = (T)s;
        if (mine) prow[a] = (T)t;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_predict_cov(T *P, int ld, int n, const double *st, RowMap rm)
{
    predict_cov_body<T>((int)blockIdx.x, P, ld, n, st, rm);
}

void launch_predict(EkfEngine *e)
{
    k_predict_prepare<<<1, 256, 0, e->stream>>>(e->d.state, e->par, nullptr);
    const int nb = 1 + (e->n > 13 ? (e->n - 13 + 255) / 256 : 0);
    if (e->f32)
        k_predict_cov<float><<<nb, 256, 0, e->stream>>>((float *)e->d.P, e->ldP, e->n, e->d.state, e->rm);
    else
        k_predict_cov<double><<<nb, 256, 0, e->stream>>>((double *)e->d.P, e->ldP, e->n, e->d.state, e->rm);
}

// ------------------------------------------------------------------------------------------------------ A3
// One thread per work item: pixel prediction, visibility, and (when predicted) the Jacobian blocks.
// list != null (ONE workgroup of BLOCK threads covers all work items): the ordered compaction of the predicted items -- k_compact's
// job -- in the same launch: list[k] = feature index of the k-th predicted item, *out_count = how many.
template <int BLOCK>
__device__ __forceinline__ void
predict_features_body(const int bx, const double *st, const CamD &cam, const double *feat_pos, const int *feat_type, const int *idx,
                      int count, int *flag, int *vis, double *uv_tab, double *Hs_tab, double *Hf_tab, int *vis_full, int *list, int *out_count)
{
    __shared__ int wtot[16];
    const int w = bx * BLOCK + threadIdx.x;
    if (list && threadIdx.x < 16) wtot[threadIdx.x] = 0;
    bool ok = false;
    int fi = 0;
    if (w < count) {
        fi = idx ? idx[w] : w;
        const double *x = st + ST_X;
        const double *R = st + ST_R;
        double Rt[9], Rinv[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Rt[j * 3 + i] = R[i * 3 + j];
        inv3(R, Rinv);
        double fp[6];
        for (int i = 0; i < 6; ++i) fp[i] = feat_pos[6 * fi + i];
        const int type = feat_type[fi];
        double uv[2];
        ok = predict_pixel(cam, x, Rt, Rinv, fp, type, uv);
        flag[w] = ok ? 1 : 0;
        vis[fi] = ok ? 1 : 0;
        // the list of unseen features the map management consumes is the one of the step's FULL prediction (EKF.cpp:277-284);
        // the outlier re-prediction after the first update must not disturb it (its own "unseen" list is discarded, :472-476)
        if (vis_full) vis_full[fi] = ok ? 1 : 0;
        if (ok) {
            uv_tab[2 * fi] = uv[0];
            uv_tab[2 * fi + 1] = uv[1];
            if (Hs_tab) {
                double Hs[14], Hf[12];
                measurement_jacobians(cam, x, Rinv, fp, type, uv, Hs, Hf);
                for (int i = 0; i < 14; ++i) Hs_tab[14 * fi + i] = Hs[i];
                for (int i = 0; i < 12; ++i) Hf_tab[12 * fi + i] = Hf[i];
            }
        }
    }
    if (!list) return; // (uniform)
    __syncthreads();
    int total;
    const int pos = block_exclusive_scan_1024(ok ? 1 : 0, wtot, &total); // (wavefronts that do not exist left zeros)
    if (ok) list[pos] = fi;
    if (threadIdx.x == 0) *out_count = total;
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_predict_features(const double *st, CamD cam, const double *feat_pos, const int *feat_type, const int *idx,
                   int count, int *flag, int *vis, double *uv_tab, double *Hs_tab, double *Hf_tab, int *vis_full, int *list, int *out_count)
{
    predict_features_body<BLOCK>((int)blockIdx.x, st, cam, feat_pos, feat_type, idx, count, flag, vis, uv_tab, Hs_tab, Hf_tab, vis_full, list, out_count);
}

// EKF::step's prediction: the covariance strips (predictCovariance) and the pixel predictions + Jacobians of every feature
// (predictMeasurementState) both need only what k_predict_prepare left in the state block and touch disjoint data: ONE launch,
// workgroups [0, nb_cov) the strips, the rest the features.
template <typename T>
__global__ void __launch_bounds__(256)
k_predict_cov_features(int nb_cov, T *P, int ld, int n, RowMap rm, const double *st, CamD cam, const double *feat_pos, const int *feat_type,
                       int count, int *flag, int *vis, double *uv_tab, double *Hs_tab, double *Hf_tab, int *vis_full, int *list, int *out_count,
                       const int *counts)
{
    if ((int)blockIdx.x < nb_cov) {
        if (filter_frozen(counts)) return; // (the pixel predictions only fill tables: they may run)
        predict_cov_body<T>((int)blockIdx.x, P, ld, n, st, rm);
    } else predict_features_body<256>((int)blockIdx.x - nb_cov, st, cam, feat_pos, feat_type, nullptr, count, flag, vis, uv_tab, Hs_tab, Hf_tab,
                                    vis_full, list, out_count);
}

// launch_predict + launch_predict_features(e, nullptr, count, false, true) of the step path in two launches instead of three (or
// four); returns whether the compaction of the predicted list is left to launch_hp_rows(..., from_flags = true)
bool launch_predict_with_features(EkfEngine *e, int count)
{
    if (count <= 0) {
        launch_predict(e);
        return launch_predict_features(e, nullptr, count, false, true);
    }
    k_predict_prepare<<<1, 256, 0, e->stream>>>(e->d.state, e->par, e->d.counts);
    const int nb_cov = 1 + (e->n > 13 ? (e->n - 13 + 255) / 256 : 0);
    const int nb_f = (count + 255) / 256;
    const bool one = count <= 256; // one workgroup of features: it compacts its own list
    int *list = one ? e->d.plist : nullptr, *cnt = one ? e->d.counts + CNT_NPRED : nullptr;
#define PCF_ARGS e->ldP, e->n, e->rm, e->d.state, e->cam, e->d.feat_pos, e->d.feat_type, count, e->d.work_flag, e->d.pred_vis, e->d.pred_uv, e->d.Hs, \
                 e->d.Hf, e->d.pred_vis_full, list, cnt, e->d.counts
    if (e->f32) k_predict_cov_features<float><<<nb_cov + nb_f, 256, 0, e->stream>>>(nb_cov, (float *)e->d.P, PCF_ARGS);
    else k_predict_cov_features<double><<<nb_cov + nb_f, 256, 0, e->stream>>>(nb_cov, (double *)e->d.P, PCF_ARGS);
#undef PCF_ARGS
    return !one;
}

// Ordered compaction of flag[0..count) by one 1024-thread block: list[k] = feature index of the k-th predicted
// work item (input order, as the reference's push_back order), *out_count = how many.
__global__ void __launch_bounds__(1024) k_compact(const int *flag, const int *idx, int count, int *list, int *out_count)
{
    __shared__ int wtot[16];
    const int tid = threadIdx.x;
    const int per = (count + 1023) / 1024;
    const int b = tid * per, e = min(count, b + per);
    int c = 0;
    for (int i = b; i < e; ++i) c += flag[i] ? 1 : 0;
    int total;
    int pos = block_exclusive_scan_1024(c, wtot, &total);
    for (int i = b; i < e; ++i)
        if (flag[i]) list[pos++] = idx ? idx[i] : i;
    if (tid == 1023) *out_count = total;
}

// state_only: pixel predictions into the scratch tables (vis2/uv2, list plist_sub, counter CNT_NPRED_SUB) so the
// tables the following stages consume stay untouched.
// defer_compact (the step's full prediction, more than 256 features): the compaction is left to the launch of k_hp_rows that follows
// (launch_hp_rows(..., from_flags = true)); returns whether it was
bool launch_predict_features(EkfEngine *e, const int *d_idx, int count, bool state_only, bool defer_compact)
{
    const bool sub = state_only || d_idx != nullptr;
    if (count <= 0) {
        (void)hipMemsetAsync(e->d.counts + (sub ? CNT_NPRED_SUB : CNT_NPRED), 0, sizeof(int), e->stream);
        return false;
    }
    int *list = sub ? e->d.plist_sub : e->d.plist, *cnt = e->d.counts + (sub ? CNT_NPRED_SUB : CNT_NPRED);
#define PF_ARGS(L_, C_) e->d.state, e->cam, e->d.feat_pos, e->d.feat_type, d_idx, count, e->d.work_flag, state_only ? e->d.pred_vis2 : e->d.pred_vis, \
                        state_only ? e->d.pred_uv2 : e->d.pred_uv, state_only ? nullptr : e->d.Hs, state_only ? nullptr : e->d.Hf,                      \
                        sub ? nullptr : e->d.pred_vis_full, L_, C_
    // up to 256 work items (small maps; the outliers re-predicted after the first update): one workgroup, the compaction in the same
    // launch -- one launch fewer per prediction (N = 200: 4051-4091 -> 4131-4150 updates/s).  (One 1024-thread workgroup for up to 1024
    // items was measured too: the prediction of an N = 1000 map got 4 us slower on one CU than it gained.)
    if (count <= 256) {
        k_predict_features<256><<<1, 256, 0, e->stream>>>(PF_ARGS(list, cnt));
        return false;
    }
    const int nb = (count + 255) / 256;
    k_predict_features<256><<<nb, 256, 0, e->stream>>>(PF_ARGS(nullptr, nullptr));
#undef PF_ARGS
    if (defer_compact && !sub && !d_idx) return true;
    k_compact<<<1, 1024, 0, e->stream>>>(e->d.work_flag, d_idx, count, sub ? e->d.plist_sub : e->d.plist,
                                         e->d.counts + (sub ? CNT_NPRED_SUB : CNT_NPRED));
    return false;
}

// ------------------------------------------------------------------------------------------------------ A4
// One workgroup per predicted feature: the 2 x n row pair  H_f P = Hf P[pos:pos+d,:] + Hs P[0:7,:]  (:644, columns
// 7..12 of Hs are structurally zero), streamed with coalesced reads of 7+d rows of P, and the 2x2 innovation
// covariance S_f = (H_f P) H_f' + I (:651-653) from the fp64 values of the 13 columns H_f touches.
// TO: storage type of the row pairs (= T, except EKF_PRECISION_F32_EXACT: P in fp32, H P in fp64 -- an H P rounded to fp32
// independently of P makes S = (H P) H' + R and B = inv(L) (H P) inconsistent with the P they downdate, and that
// inconsistency, unlike the rounding of P itself, is amplified by the conditioning of S: measured at N = 2000, 1.2e-5
// component-wise with an fp32 H P against 2.5e-7 for fp32 storage of P alone)
template <typename T, typename TO>
__global__ void __launch_bounds__(256)
k_hp_rows(const T *P, int ld, int n, const int *list, const int *feat_type, const int *feat_covpos,
          const double *Hs_tab, const double *Hf_tab, TO *HP, double *S_tab, RowMap rm, double *HPc, unsigned *times_predicted,
          const int *d_count, const int *flag, int n_items, int *list_out, int *count_out, const int *counts)
{
    const int tid = threadIdx.x;
    if (flag) {
        // the step's FULL prediction: workgroup x = feature x, predicted or not (k_predict_features' flags) -- nobody needs the compacted
        // list here, so its compaction (k_compact's job, needed by the matcher) rides in this launch as one more workgroup
        if ((int)blockIdx.x == n_items) {
            if (blockIdx.y != 0) return;
            __shared__ int wtot[16];
            if (tid < 16) wtot[tid] = 0;
            __syncthreads();
            const int per = (n_items + 255) / 256;
            const int b = tid * per, e = min(n_items, b + per);
            int c = 0;
            for (int i = b; i < e; ++i) c += flag[i] ? 1 : 0;
            int total;
            int pos = block_exclusive_scan_1024(c, wtot, &total); // (the wavefronts that do not exist left zeros)
            for (int i = b; i < e; ++i)
                if (flag[i]) list_out[pos++] = i;
            if (tid == 0) *count_out = total;
            return;
        }
        if (!flag[blockIdx.x]) return;
    } else if (d_count && (int)blockIdx.x >= *d_count) return; // the grid is an upper bound when the length of the list is only known on the device (step path: no read-back)
    __shared__ double sH[26];     // Hs (2x7) then Hf (2x6)
    __shared__ double sHP[2][13]; // fp64 H P at columns 0..6 and pos..pos+d-1
    const int fi = flag ? (int)blockIdx.x : list[blockIdx.x];
    const int d = feat_dim(feat_type[fi]);
    const int pos = feat_covpos[fi];
    // updateMapFeatures' timesPredicted++ (MapManagement.cpp:81-86) rides along when a step asks for it (every rank counts
    // every feature: the map is replicated)
    // blockIdx.y: the chunk of 256 x 16 bytes of the row pair this workgroup produces (one round of loads per workgroup:
    // a feature's 7 + d rows are 13 x n x w bytes, and one workgroup walking all of them was six dependent rounds -- the
    // outlier re-prediction, with fewer features than CUs, ran at a third of the bandwidth of the full pass)
    const bool first = blockIdx.y == 0;
    if (times_predicted && tid == 0 && first && !filter_frozen(counts)) times_predicted[fi]++;
    if (!owns_row(rm, pos)) return; // sharded: the owner of the feature's rows computes them, the exchange delivers them
    const T *Pf = P + (size_t)local_row(rm, pos) * ld;
    if (tid < 14) sH[tid] = Hs_tab[14 * fi + tid];
    else if (tid < 26) sH[tid] = Hf_tab[12 * fi + tid - 14];
    // chunk 0 also forms the 2 x 2 S_f: the fp64 H P at the 7 camera columns and at the feature's own d columns, one
    // thread per column (same operations in the same order as the row pair's elements below)
    double pcol[13];
    int jcol = -1;
    if (first && tid >= 64 && tid < 64 + 7 + d) {
        jcol = tid - 64 < 7 ? tid - 64 : pos + (tid - 64 - 7);
#pragma unroll
        for (int a = 0; a < 6; ++a) pcol[a] = a < d ? (double)Pf[(size_t)a * ld + jcol] : 0.0;
#pragma unroll
        for (int a = 0; a < 7; ++a) pcol[6 + a] = (double)P[(size_t)a * ld + jcol];
    }
    __syncthreads();
    if (jcol >= 0) {
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
            if (a < d) {
                a0 += sH[14 + a] * pcol[a];
                a1 += sH[20 + a] * pcol[a];
            }
#pragma unroll
        for (int a = 0; a < 7; ++a) {
            b0 += sH[a] * pcol[6 + a];
            b1 += sH[7 + a] * pcol[6 + a];
        }
        sHP[0][tid - 64] = a0 + b0;
        sHP[1][tid - 64] = a1 + b1;
    }
    TO *o0 = HP + (size_t)(2 * fi) * ld;
    TO *o1 = o0 + ld;
    // 16 bytes per lane and row: a wavefront reads 1 KB (fp32) / 1 KB (fp64, two columns) of CONTIGUOUS row per load
    // instead of 256 B, which the HBM-bound pass over P needs (rows are ld elements apart; ld is a multiple of 128, so
    // reading up to the padded row end is in bounds; columns >= n are never stored)
    constexpr int VW = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    const int jb = ((int)blockIdx.y * 256 + tid) * VW;
    if (jb < n) {
        vec_t pf[6], pc[7];
#pragma unroll
        for (int a = 0; a < 6; ++a)
            if (a < d) pf[a] = *(const vec_t *)(Pf + (size_t)a * ld + jb);
#pragma unroll
        for (int a = 0; a < 7; ++a) pc[a] = *(const vec_t *)(P + (size_t)a * ld + jb);
        typedef TO ovec_t __attribute__((ext_vector_type(VW)));
        ovec_t r0, r1;
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            const int j = jb + v;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
                if (a < d) {
                    const double p = (double)pf[a][v];
                    a0 += sH[14 + a] * p;
                    a1 += sH[20 + a] * p;
                }
            double b0 = 0.0, b1 = 0.0;
#pragma unroll
            for (int a = 0; a < 7; ++a) {
                const double p = (double)pc[a][v];
                b0 += sH[a] * p;
                b1 += sH[7 + a] * p;
            }
            a0 += b0;
            a1 += b1;
            r0[v] = (TO)a0;
            r1[v] = (TO)a1;
            if (j < 13) { // fp64 copy of the camera columns: the camera part of B = inv(L) (H P) is solved in fp64
                HPc[(size_t)(2 * fi) * 16 + j] = a0;
                HPc[(size_t)(2 * fi + 1) * 16 + j] = a1;
            }
        }
        if (jb + VW <= n) {
            *(ovec_t *)(o0 + jb) = r0;
            *(ovec_t *)(o1 + jb) = r1;
        } else {
#pragma unroll
            for (int v = 0; v < VW; ++v)
                if (jb + v < n) { o0[jb + v] = r0[v]; o1[jb + v] = r1[v]; }
        }
    }
    if (!first) return;
    __syncthreads();
    if (tid < 4) {
        const int r = tid >> 1, c = tid & 1;
        double s1 = 0.0, s2 = 0.0;
        for (int a = 0; a < 7; ++a) s1 += sHP[r][a] * sH[c * 7 + a];
        for (int a = 0; a < d; ++a) s2 += sHP[r][7 + a] * sH[14 + c * 6 + a];
        S_tab[4 * fi + tid] = s1 + s2 + (r == c ? 1.0 : 0.0);
    }
}

// from_flags: the step's full prediction when launch_predict_features left the compaction to this launch (deferred_compact): the
// workgroups go by k_predict_features' flags, one more workgroup writes the compacted list and its length
void launch_hp_rows(EkfEngine *e, const int *d_list, int n_list, bool count_predicted, const int *d_count, bool from_flags)
{
    unsigned *tp = count_predicted ? e->d.feat_times_predicted : nullptr;
    if (n_list <= 0) return;
    const int chunks_f = (e->n + 256 * 4 - 1) / (256 * 4), chunks_d = (e->n + 256 * 2 - 1) / (256 * 2);
    const int *flag = from_flags ? e->d.work_flag : nullptr;
    int *lo = from_flags ? e->d.plist : nullptr, *co = from_flags ? e->d.counts + CNT_NPRED : nullptr;
    const int gx = n_list + (from_flags ? 1 : 0);
    if (e->exact && e->f32)
        k_hp_rows<float, double><<<dim3(gx, chunks_f), 256, 0, e->stream>>>((const float *)e->d.P, e->ldP, e->n, d_list, e->d.feat_type,
                                                        e->d.feat_covpos, e->d.Hs, e->d.Hf, (double *)e->d.HP,
                                                        e->d.pred_S, e->rm, e->d.HPc, tp, d_count, flag, n_list, lo, co, e->d.counts);
    else if (e->f32)
        k_hp_rows<float, float><<<dim3(gx, chunks_f), 256, 0, e->stream>>>((const float *)e->d.P, e->ldP, e->n, d_list, e->d.feat_type,
                                                        e->d.feat_covpos, e->d.Hs, e->d.Hf, (float *)e->d.HP,
                                                        e->d.pred_S, e->rm, e->d.HPc, tp, d_count, flag, n_list, lo, co, e->d.counts);
    else
        k_hp_rows<double, double><<<dim3(gx, chunks_d), 256, 0, e->stream>>>((const double *)e->d.P, e->ldP, e->n, d_list,
                                                         e->d.feat_type, e->d.feat_covpos, e->d.Hs, e->d.Hf,
                                                         (double *)e->d.HP, e->d.pred_S, e->rm, e->d.HPc, tp, d_count, flag, n_list, lo, co, e->d.counts);
}

// predictMeasurementState on the current state into an EkfPrediction array (device), all features.
__global__ void __launch_bounds__(256)
k_pack_predictions(const int *list, const int *count, const double *uv_tab, const double *S_tab, EkfPrediction *out,
                   int with_S)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= *count) return;
    const int fi = list[k];
    EkfPrediction p;
    p.featureIndex = fi;
    p._pad = 0;
    p.imagePos[0] = uv_tab[2 * fi];
    p.imagePos[1] = uv_tab[2 * fi + 1];
    for (int i = 0; i < 4; ++i) p.covarianceMatrix[i] = with_S ? S_tab[4 * fi + i] : 0.0;
    out[k] = p;
}

__global__ void __launch_bounds__(256)
k_pack_predictions_n(const int *list, int n, const double *uv_tab, const double *S_tab, EkfPrediction *out)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int fi = list[k];
    EkfPrediction p;
    p.featureIndex = fi;
    p._pad = 0;
    p.imagePos[0] = uv_tab[2 * fi];
    p.imagePos[1] = uv_tab[2 * fi + 1];
    for (int i = 0; i < 4; ++i) p.covarianceMatrix[i] = S_tab[4 * fi + i];
    out[k] = p;
}

void launch_pack_predictions(EkfEngine *e, const int *d_list, int n, EkfPrediction *d_out)
{
    if (n > 0) k_pack_predictions_n<<<(n + 255) / 256, 256, 0, e->stream>>>(d_list, n, e->d.pred_uv, e->d.pred_S, d_out);
}

void launch_state_only_predict(EkfEngine *e, EkfPrediction *d_out)
{
    launch_predict_features(e, nullptr, e->N, true);
    if (e->N > 0)
        k_pack_predictions<<<(e->N + 255) / 256, 256, 0, e->stream>>>(e->d.plist_sub, e->d.counts + CNT_NPRED_SUB,
                                                                      e->d.pred_uv2, e->d.pred_S, d_out, 0);
}

} // namespace ekf
