// kernels_match.hip -- A5, the stage of matchPredictedFeatures downstream of the detector
// (EKF/Matching.cpp:217-262): per prediction, gate the frame's keypoints with the uncertainty ellipse
// (Core/EKFMath.cpp:271-351), descriptor distance -- both branches of computeDistance: Hamming on 32-byte CV_8U
// descriptors (Matching.cpp:76-92) or L2 on CV_32F descriptors (:60-73) -- and the reference's 2-element "best" list
// (Matching.cpp:116-144, 169-175).  Integer/byte work, a few KB per block, all L2-resident:
// one workgroup per prediction, keypoints strided over the lanes, candidates replayed in keypoint order so the
// order-dependent list logic gives the reference's answer.
#include "engine.h"
#include "gate.h"

namespace ekf {

__global__ void __launch_bounds__(256)
k_match(const int *plist, const double *uv_tab, const double *S_tab, const uint8_t *feat_desc,
        const EkfKeypoint *kps, const uint8_t *kdesc, int n_kp, double coef, int *mt_valid, int *mt_kp,
        float *mt_dist, int desc_bytes, int desc_f32, const int *d_npred, int slot0)
{
    // slot0: first prediction slot of this launch (a rank of a sharded filter matches the predictions of ITS features only)
    if (d_npred && slot0 + (int)blockIdx.x >= *d_npred) return; // grid = upper bound, count on the device
    __shared__ Gate g;
    __shared__ uint32_t qd[1024]; // the map feature's descriptor: 8 words (CV_8U) or up to 1024 floats (CV_32F)
    constexpr int PASS = 8; // keypoints per thread and pass: their loads are in flight together, one barrier pair per pass
    __shared__ int c_idx[256 * PASS];
    __shared__ double c_dist[256 * PASS]; // computeDistance returns double (an integer value for Hamming)
    __shared__ int wave_cnt[PASS][4];
    // list state (thread 0); DMatch::distance is a float (Matching.cpp:133)
    __shared__ int s_list_n, s_front;
    __shared__ float s_dfront, s_dback;
    __shared__ double s_min;

    const int k = slot0 + (int)blockIdx.x, tid = threadIdx.x;
    const int fi = plist[k];
    if (tid == 0) {
        float axes[2];
        double angle;
        ellipse_from_cov(S_tab + 4 * fi, axes, &angle);
        const int aw = (int)rintf(axes[0]), ah = (int)rintf(axes[1]); // cv::Size(Size2f): round half to even
        gate_from_ellipse((float)uv_tab[2 * fi], (float)uv_tab[2 * fi + 1], aw, ah, angle, &g);
        s_list_n = 0; s_front = -1; s_dfront = 0.f; s_dback = 0.f; s_min = -1.0;
    }
    const int words = desc_bytes / 4;
    for (int w = tid; w < words; w += 256) qd[w] = ((const uint32_t *)(feat_desc + (size_t)fi * desc_bytes))[w];
    const int lane = tid & 63, wv = tid >> 6;
    for (int base = 0; base < n_kp; base += 256 * PASS) {
        // this pass's keypoints, requested before the gate is needed
        float kx[PASS], ky[PASS];
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            const int j = min(base + i * 256 + tid, n_kp - 1);
            kx[i] = kps[j].x;
            ky[i] = kps[j].y;
        }
        __syncthreads(); // the gate and the descriptor (first pass); the candidate buffers are free (later passes)
        bool inside[PASS];
        double dist[PASS];
        int rank[PASS];
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
            const int j = base + i * 256 + tid;
            const double px = (double)kx[i], py = (double)ky[i];
            const double a1x = px - g.f1x, a1y = py - g.f1y, a2x = px - g.f2x, a2y = py - g.f2y;
            const double ns = sqrt(a1x * a1x + a1y * a1y) + sqrt(a2x * a2x + a2y * a2y);
            inside[i] = j < n_kp && ns <= g.two_major;
            dist[i] = 0.0;
            if (inside[i]) {
                const uint32_t *cd = (const uint32_t *)(kdesc + (size_t)j * desc_bytes);
                if (desc_f32) { // Matching.cpp:60-73: float difference, float square, double sum in column order, sqrt
                    double acc = 0.0;
                    for (int w = 0; w < words; ++w) {
                        const float subs = __uint_as_float(qd[w]) - __uint_as_float(cd[w]);
                        acc += (double)(subs * subs);
                    }
                    dist[i] = sqrt(acc);
                } else {
                    int d = 0;
                    for (int w = 0; w < words; ++w) d += __popc(cd[w] ^ qd[w]);
                    dist[i] = (double)d;
                }
            }
            // ordered compaction of the candidates: keypoint order = (i, wavefront, lane)
            const unsigned long long bal = __ballot(inside[i]);
            rank[i] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_cnt[i][wv] = __popcll(bal);
        }
        __syncthreads();
        int nc = 0;
#pragma unroll
        for (int i = 0; i < PASS; ++i) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int c = wave_cnt[i][w];
                if (w == wv && inside[i]) {
                    c_idx[nc + rank[i]] = base + i * 256 + tid;
                    c_dist[nc + rank[i]] = dist[i];
                }
                nc += c;
            }
        }
        __syncthreads();
        if (tid == 0) {
            // findBestNMatches, nBest = 2: push_front when (dist < min) or fewer than two entries
            for (int c = 0; c < nc; ++c) {
                const double dc = c_dist[c];
                if (dc < s_min || s_list_n < 2) { // minDistance starts at -1: the first two candidates always enter (:130)
                    s_min = s_min < 0 ? dc : fmin(s_min, dc);
                    s_dback = s_dfront;
                    s_front = c_idx[c];
                    s_dfront = (float)dc;
                    if (s_list_n < 2) ++s_list_n;
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        // matchICDescriptors: accept a lone candidate, or front <= back * coef (distances are floats in DMatch)
        const bool ok = s_list_n == 1 || (s_list_n >= 2 && (double)s_dfront <= (double)s_dback * coef);
        mt_valid[k] = ok ? 1 : 0;
        mt_kp[k] = ok ? s_front : -1;
        mt_dist[k] = s_dfront;
    }
}

// compacted match list in prediction order (matches.push_back order, Matching.cpp:247-262)
__global__ void __launch_bounds__(1024)
k_match_compact(const int *plist, int n_pred, const int *mt_valid, const int *mt_kp, const float *mt_dist,
                const EkfKeypoint *kps, int by_slot, EkfMatch *out, int *out_count, const int *d_npred,
                int *rs_counts, int *match_of_feat, uint8_t *best_flags, int N)
{
    if (d_npred) n_pred = *d_npred;
    __shared__ int wtot[16];
    const int tid = threadIdx.x;
    if (rs_counts) {
        // what k_ransac_init + k_match_index do for an arbitrary match list (1PointRansac.cpp:58-82, :116), folded in for the
        // matcher's own list: a feature appears in it at most once, so "the first match of feature f" is the match itself
        if (tid == 0) {
            rs_counts[CNT_RS_BEST] = 0;
            rs_counts[CNT_RS_BESTH] = -1;
            rs_counts[CNT_RS_NHYP] = 1000;
            rs_counts[CNT_RS_NEXT] = 0;
            rs_counts[CNT_RS_DONE] = 0;
        }
        for (int i = tid; i < N; i += 1024) {
            match_of_feat[i] = 0x7fffffff;
            best_flags[i] = 0;
        }
        __syncthreads();
    }
    const int per = (n_pred + 1023) / 1024;
    const int b = tid * per, e = min(n_pred, b + per);
    int c = 0;
    for (int i = b; i < e; ++i) c += mt_valid[i];
    int total;
    int pos = block_exclusive_scan_1024(c, wtot, &total);
    for (int i = b; i < e; ++i)
        if (mt_valid[i]) {
            EkfMatch m;
            m.featureIndex = plist[i];
            // by_slot (NCC matcher): the matched pixel sits in kps[prediction slot]; there is no keypoint index
            const int kp = by_slot ? i : mt_kp[i];
            m.keypointIndex = by_slot ? -1 : kp;
            m.imagePos[0] = (double)kps[kp].x;
            m.imagePos[1] = (double)kps[kp].y;
            m.distance = mt_dist[i];
            m._pad = 0.f;
            if (rs_counts) match_of_feat[m.featureIndex] = pos;
            out[pos++] = m;
        }
    if (tid == 1023) *out_count = total;
}

void launch_match(EkfEngine *e, int n_pred, int n_kp, const int *d_npred, bool with_ransac_init)
{
    if (n_pred <= 0) {
        (void)hipMemsetAsync(e->d.counts + CNT_NMATCH, 0, sizeof(int), e->stream);
        if (with_ransac_init) launch_ransac_init(e, e->N);
        return;
    }
    k_match<<<n_pred, 256, 0, e->stream>>>(e->d.plist, e->d.pred_uv, e->d.pred_S, e->d.feat_desc, e->d.kps,
                                           e->d.kdesc, n_kp, e->cfg.par.matchingCompCoefSecondBestVSFirst,
                                           e->d.mt_valid, e->d.mt_kp, e->d.mt_dist, e->desc_bytes, e->desc_f32 ? 1 : 0, d_npred, 0);
    k_match_compact<<<1, 1024, 0, e->stream>>>(e->d.plist, n_pred, e->d.mt_valid, e->d.mt_kp, e->d.mt_dist,
                                               e->d.kps, 0, e->d.matches, e->d.counts + CNT_NMATCH, d_npred,
                                               with_ransac_init ? e->d.counts : nullptr, e->d.match_of_feat, e->d.best_flags, e->N);
}

// Sharded filter ("Matching: features/ellipses independent => shard by feature ... all-gather match lists"): a rank
// gates and matches the prediction slots [s_lo, s_hi) -- the predictions of the features it owns: the predicted list is in feature
// order, so they are ONE run of slots -- into the per-slot tables; the ranks all-gather the tables (engine.cpp) and every rank
// compacts the same complete tables into the same match list.
void launch_match_slots(EkfEngine *e, int n_kp, int s_lo, int s_hi)
{
    if (s_hi <= s_lo) return;
    k_match<<<s_hi - s_lo, 256, 0, e->stream>>>(e->d.plist, e->d.pred_uv, e->d.pred_S, e->d.feat_desc, e->d.kps, e->d.kdesc, n_kp,
                                                e->cfg.par.matchingCompCoefSecondBestVSFirst, e->d.mt_valid, e->d.mt_kp, e->d.mt_dist,
                                                e->desc_bytes, e->desc_f32 ? 1 : 0, nullptr, s_lo);
}

void launch_match_compact(EkfEngine *e, int n_pred)
{
    if (n_pred <= 0) {
        (void)hipMemsetAsync(e->d.counts + CNT_NMATCH, 0, sizeof(int), e->stream);
        return;
    }
    k_match_compact<<<1, 1024, 0, e->stream>>>(e->d.plist, n_pred, e->d.mt_valid, e->d.mt_kp, e->d.mt_dist, e->d.kps, 0, e->d.matches,
                                               e->d.counts + CNT_NMATCH, nullptr, nullptr, nullptr, nullptr, 0);
}

void launch_match_compact_slots(EkfEngine *e, int n_pred, const EkfKeypoint *d_slot_xy)
{
    k_match_compact<<<1, 1024, 0, e->stream>>>(e->d.plist, n_pred, e->d.mt_valid, e->d.mt_kp, e->d.mt_dist, d_slot_xy,
                                               1, e->d.matches, e->d.counts + CNT_NMATCH, nullptr, nullptr, nullptr, nullptr, 0);
}

// match_of_feat[f] = smallest match index whose featureIndex is f, or -1 (the linear searches of
// 1PointRansac.cpp:58-82 stop at the first hit)
__global__ void __launch_bounds__(256) k_match_index(const EkfMatch *m, int M, int *match_of_feat, int N, const int *d_M)
{
    if (d_M) M = *d_M;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const int f = m[i].featureIndex;
    if (f >= 0 && f < N) atomicMin(&match_of_feat[f], i);
}

__global__ void __launch_bounds__(256) k_fill_int(int *p, int n, int v)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

void launch_match_index(EkfEngine *e, int M, const int *d_M)
{
    if (e->N <= 0) return;
    // match_of_feat was set to "none" by launch_ransac_init, which precedes this launch
    if (M > 0) k_match_index<<<(M + 255) / 256, 256, 0, e->stream>>>(e->d.matches, M, e->d.match_of_feat, e->N, d_M);
}

// Stable partition of src[0..M) by flags: dst1 receives the flagged matches, dst0 (optional) the others, both in
// the original order (1PointRansac.cpp:213-227, EKF.cpp:110-117).
__global__ void __launch_bounds__(1024)
k_partition(const EkfMatch *src, int M, const uint8_t *flags, EkfMatch *dst1, EkfMatch *dst0, int *cnt1, const uint8_t *kdesc,
            uint8_t *feat_desc, unsigned *times_matched, int desc_bytes, int *idx0, const int *counts, int *mirror,
            int publish_seq, uint8_t *rescue_mask, const int *vis, const double *uv_tab, const double *S_tab, double chi2)
{
    __shared__ int wtot[16];
    const int tid = threadIdx.x;
    if (filter_frozen(counts)) { // behind a failed update (engine.h): the lists and the map's bookkeeping stay; the host still gets its block
        if (publish_seq > 0) publish_counts_block(counts, mirror, publish_seq);
        return;
    }
    const int per = (M + 1023) / 1024;
    const int b = tid * per, e = min(M, b + per);
    if (rescue_mask) {
        // rescueOutliers (EKF.cpp:84-97) in the same launch: the flags of this partition are nu' inv(S_i) nu < chi2 against
        // the re-predicted measurements -- k_rescue's expression, thread-private entries of the mask
        for (int i = b; i < e; ++i) {
            const int fi = src[i].featureIndex;
            const double d0 = src[i].imagePos[0] - uv_tab[2 * fi];
            const double d1 = src[i].imagePos[1] - uv_tab[2 * fi + 1];
            double Si[4];
            inv2(S_tab + 4 * fi, Si);
            const double t0 = d0 * Si[0] + d1 * Si[2];
            const double t1 = d0 * Si[1] + d1 * Si[3];
            const double v = t0 * d0 + t1 * d1;
            rescue_mask[i] = (vis[fi] && v < chi2) ? 1 : 0;
        }
        flags = rescue_mask;
    }
    int c = 0;
    for (int i = b; i < e; ++i) c += flags[i] ? 1 : 0;
    int total;
    int p1 = block_exclusive_scan_1024(c, wtot, &total);
    int p0 = b - p1;
    for (int i = b; i < e; ++i) {
        if (flags[i]) dst1[p1++] = src[i];
        else if (dst0) {
            if (idx0) idx0[p0] = src[i].featureIndex; // the work list of the outlier re-prediction (EKF.cpp:473)
            dst0[p0++] = src[i];
        }
    }
    if (tid == 1023 && cnt1) *cnt1 = total;
    if (publish_seq > 0) publish_counts_block(counts, mirror, publish_seq); // the count above is part of the block
    if (!times_matched) return;
    // updateMapFeatures for the selected matches (MapManagement.cpp:88-113), fused: timesMatched++ and the map descriptor
    // replaced by the matched keypoint's (matcher mode B has no keypoint: keypointIndex < 0)
    __syncthreads();
    const int count = total;
    for (int i = tid; i < count; i += 1024) {
        const int fi = dst1[i].featureIndex, kp = dst1[i].keypointIndex;
        atomicAdd(&times_matched[fi], 1u); // one match per feature on every engine path; atomic all the same
        if (kp >= 0 && kdesc) {
            const uint32_t *sd = (const uint32_t *)(kdesc + (size_t)kp * desc_bytes);
            uint32_t *dd = (uint32_t *)(feat_desc + (size_t)fi * desc_bytes);
            for (int w = 0; w < desc_bytes / 4; ++w) dd[w] = sd[w];
        }
    }
}

void launch_partition(EkfEngine *e, const EkfMatch *src, int M, const uint8_t *flags, EkfMatch *dst1, EkfMatch *dst0,
                      int *cnt1, bool map_update, const uint8_t *d_kdesc, int *d_idx0, int publish_seq, bool rescue)
{
    if (M <= 0) {
        if (cnt1) (void)hipMemsetAsync(cnt1, 0, sizeof(int), e->stream);
        return;
    }
    k_partition<<<1, 1024, 0, e->stream>>>(src, M, flags, dst1, dst0, cnt1, d_kdesc, e->d.feat_desc,
                                           map_update ? e->d.feat_times_matched : nullptr, e->desc_bytes, d_idx0, e->d.counts,
                                           e->d_mirror, e->d_mirror ? publish_seq : 0, rescue ? e->d.mask : nullptr, e->d.pred_vis,
                                           e->d.pred_uv, e->d.pred_S, e->cfg.par.ransacChi2Threshold);
}





// Sharded filter.  The match lists of a step are in feature order (predictions are compacted in feature
// order, partitions are stable) and features are dealt to the ranks in contiguous index ranges, so the entries a rank owns
// form ONE run of the list: thread r finds where rank r's run starts (lower bound of its first feature).
__global__ void k_shard_bounds(const EkfMatch *list, int count, const int *feat_begin, int world, int *counts)
{
    const int r = threadIdx.x;
    const int f = r <= world ? feat_begin[r] : 0;
    int lo = 0, hi = r <= world ? count : 0;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (list[mid].featureIndex < f) lo = mid + 1;
        else hi = mid;
    }
    if (r <= world) counts[CNT_SHARD0 + r] = lo;
    // the boundaries mean "rank r's rows are ONE run" only for a list in strictly increasing feature order: verify it (one pass,
    // all 64 threads) and poison the last boundary otherwise -- the host then refuses the update instead of exchanging rows
    // that were never written
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    for (int i = threadIdx.x; i + 1 < count; i += blockDim.x)
        if (list[i].featureIndex >= list[i + 1].featureIndex) bad = 1;
    __syncthreads();
    if (bad && r == world) counts[CNT_SHARD0 + r] = -1;
}

// ... the same for a list of feature indices whose length is on the device (the predicted list of a step: plist, counts[CNT_NPRED])
__global__ void k_shard_bounds_idx(const int *list, const int *d_count, const int *feat_begin, int world, int *counts)
{
    const int r = threadIdx.x, count = *d_count;
    const int f = r <= world ? feat_begin[r] : 0;
    int lo = 0, hi = r <= world ? count : 0;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (list[mid] < f) lo = mid + 1;
        else hi = mid;
    }
    if (r <= world) counts[CNT_SHARD0 + r] = lo;
}

void launch_shard_bounds_idx(EkfEngine *e, const int *list, const int *d_count)
{
    k_shard_bounds_idx<<<1, 64, 0, e->stream>>>(list, d_count, e->d.shard_feat, e->shard_world, e->d.counts);
}

void launch_shard_bounds(EkfEngine *e, const EkfMatch *list, int count)
{
    k_shard_bounds<<<1, 64, 0, e->stream>>>(list, count, e->d.shard_feat, e->shard_world, e->d.counts);
}

__global__ void __launch_bounds__(256) k_outlier_idx(const EkfMatch *src, int M, int *idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) idx[i] = src[i].featureIndex;
}

void launch_outlier_idx(EkfEngine *e, const EkfMatch *src, int M, int *idx)
{
    if (M > 0) k_outlier_idx<<<(M + 255) / 256, 256, 0, e->stream>>>(src, M, idx);
}

// counter block -> GPU-writable host page, then the sequence number the host polls (engine.cpp read_counts)
__global__ void k_publish_counts(const int *counts, int *mirror, int seq)
{
    const int t = threadIdx.x;
    if (t < CNT_COUNT) __hip_atomic_store(&mirror[t], counts[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (t == 0) {
        __threadfence_system();
        __hip_atomic_store(&mirror[CNT_COUNT], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

void launch_publish_counts(EkfEngine *e, int *d_mirror, int seq)
{
    k_publish_counts<<<1, 64, 0, e->stream>>>(e->d.counts, d_mirror, seq);
}

} // namespace ekf
