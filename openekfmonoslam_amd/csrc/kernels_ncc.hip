// kernels_ncc.hip -- matcher mode B: the image-taking form of matchPredictedFeatures (EKF/Matching.h:66 takes the
// cv::Mat frame) done without a detector: an 11x11 template per map feature, zero-mean NCC evaluated at every pixel
// of the predicted uncertainty ellipse (the gate of Matching.cpp:217-241) on the coarsest level of a 3-level 2x
// pyramid, then refined through the 4x4 children at the two finer levels.  All image arithmetic is integer, the
// score is num^2/den in fp64 (one multiply, one divide of identically rounded operands), so the result is
// bit-identical to the CPU definition the tests check against.
//
// Byte work: a frame is ~1.6 MB of pyramid, a prediction touches a <= 43x43 window of the coarse level.  One
// workgroup per prediction stages the window and the template in LDS; nothing here is GEMM-shaped.
#include "engine.h"
#include "gate.h"

namespace ekf {

constexpr int NCC_R = 5, NCC_T = 11, NCC_TT = 121, NCC_MAXRAD = 16;
constexpr int NCC_WIN = 2 * NCC_MAXRAD + 1 + 2 * NCC_R; // 43

// ---- pyramid -----------------------------------------------------------------------------------------------
// gray = (77 R + 150 G + 29 B + 128) >> 8; 3 channels = B G R (cv::imread order, Img/FileSequenceImageGenerator),
// 4 channels = R G B A (android jni/EKFNative.cpp:163)
__global__ void __launch_bounds__(256)
k_ncc_gray(const uint8_t *raw, int w, int h, int stride, int channels, uint8_t *out)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t *p = raw + (size_t)y * stride + (size_t)x * channels;
    int g;
    if (channels == 1) g = p[0];
    else if (channels == 3) g = (77 * p[2] + 150 * p[1] + 29 * p[0] + 128) >> 8;
    else g = (77 * p[0] + 150 * p[1] + 29 * p[2] + 128) >> 8;
    out[(size_t)y * w + x] = (uint8_t)g;
}

__global__ void __launch_bounds__(256) k_ncc_down(const uint8_t *src, int sw, uint8_t *dst, int dw, int dh)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const uint8_t *r0 = src + (size_t)(2 * y) * sw + 2 * x, *r1 = r0 + sw;
    dst[(size_t)y * dw + x] = (uint8_t)((r0[0] + r0[1] + r1[0] + r1[1] + 2) >> 2);
}

struct Pyr {
    const uint8_t *px[3];
    int w[3], h[3];
};

__device__ inline int pyr_at(const Pyr &p, int l, int x, int y)
{
    x = min(max(x, 0), p.w[l] - 1);
    y = min(max(y, 0), p.h[l] - 1);
    return p.px[l][(size_t)y * p.w[l] + x];
}

__device__ inline int to_level(double u, int l) { return (int)floor((u + 0.5) / (double)(1 << l)); }

// templates of the listed features from the current pyramid: block = (item, level), 121 active lanes
__global__ void __launch_bounds__(128)
k_ncc_capture(Pyr pyr, const int *feat_idx, const double *uv, uint8_t *tmpl)
{
    const int i = blockIdx.x, l = blockIdx.y, t = threadIdx.x;
    if (t >= NCC_TT) return;
    const int cx = to_level(uv[2 * i], l), cy = to_level(uv[2 * i + 1], l);
    const int dy = t / NCC_T - NCC_R, dx = t % NCC_T - NCC_R;
    tmpl[((size_t)feat_idx[i] * 3 + l) * NCC_TT + t] = (uint8_t)pyr_at(pyr, l, cx + dx, cy + dy);
}

// ---- matching ----------------------------------------------------------------------------------------------
// zncc^2 of a candidate whose 11x11 window starts at sw[oy][ox] (LDS window of row pitch `pitch`)
__device__ inline double ncc_key(const uint8_t *win, int pitch, int ox, int oy, const uint8_t *tp, int st, int stt)
{
    int s = 0, ss = 0, sx = 0;
    for (int dy = 0; dy < NCC_T; ++dy) {
        const uint8_t *wr = win + (oy + dy) * pitch + ox;
        const uint8_t *tr = tp + dy * NCC_T;
#pragma unroll
        for (int dx = 0; dx < NCC_T; ++dx) {
            const int wv = wr[dx], tv = tr[dx];
            s += wv;
            ss += wv * wv;
            sx += wv * tv;
        }
    }
    const long long n = NCC_TT;
    const long long num = n * sx - (long long)s * st;
    const long long den = (n * ss - (long long)s * s) * (n * stt - (long long)st * st);
    if (num <= 0 || den <= 0) return -1.0;
    const double dn = (double)num;
    return dn * dn / (double)den;
}

// block argmax of (key, candidate index): larger key wins, equal keys -> smaller index (raster order, the
// CPU loop's strict '>')
__device__ inline void block_argmax(double &key, int &idx, double *s_key, int *s_idx)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int o = 32; o > 0; o >>= 1) {
        const double k2 = __shfl_down(key, o);
        const int i2 = __shfl_down(idx, o);
        if (k2 > key || (k2 == key && i2 < idx)) { key = k2; idx = i2; }
    }
    if (lane == 0) { s_key[wv] = key; s_idx[wv] = idx; }
    __syncthreads();
    key = s_key[0]; idx = s_idx[0];
    for (int w = 1; w < 4; ++w)
        if (s_key[w] > key || (s_key[w] == key && s_idx[w] < idx)) { key = s_key[w]; idx = s_idx[w]; }
    __syncthreads();
}

__global__ void __launch_bounds__(256)
k_ncc_match(Pyr pyr, const int *plist, const double *uv_tab, const double *S_tab, const uint8_t *tmpl,
            int *mt_valid, EkfKeypoint *mt_xy, float *mt_dist, int slot0)
{
    __shared__ Gate g;
    __shared__ int s_geom[4]; // c2x, c2y, rad
    __shared__ uint8_t s_win[NCC_WIN * NCC_WIN + 3];
    __shared__ uint8_t s_t[NCC_TT + 3];
    __shared__ int s_tsum[2];
    __shared__ double s_key[4];
    __shared__ int s_idx[4];

    const int k = slot0 + (int)blockIdx.x, tid = threadIdx.x; // slot0: see launch_match_ncc_slots
    const int fi = plist[k];
    const double pu = uv_tab[2 * fi], pv = uv_tab[2 * fi + 1];
    if (tid == 0) {
        float axes[2];
        double angle;
        ellipse_from_cov(S_tab + 4 * fi, axes, &angle);
        const int aw = (int)rintf(axes[0]), ah = (int)rintf(axes[1]);
        gate_from_ellipse((float)pu, (float)pv, aw, ah, angle, &g);
        const int major = aw > ah ? aw : ah;
        s_geom[0] = to_level(pu, 2);
        s_geom[1] = to_level(pv, 2);
        s_geom[2] = min((major >> 2) + 1, NCC_MAXRAD);
    }
    __syncthreads();

    int bx = s_geom[0], by = s_geom[1];
    double bkey = -3.0;
    for (int l = 2; l >= 0; --l) {
        // candidate window at this level: coarse = the gated square around the prediction, finer = 4x4 children
        int x0, y0, cw;
        if (l == 2) { x0 = s_geom[0] - s_geom[2]; y0 = s_geom[1] - s_geom[2]; cw = 2 * s_geom[2] + 1; }
        else { x0 = 2 * bx - 1; y0 = 2 * by - 1; cw = 4; }
        const int pitch = cw + 2 * NCC_R;
        for (int i = tid; i < pitch * pitch; i += 256)
            s_win[i] = (uint8_t)pyr_at(pyr, l, x0 - NCC_R + i % pitch, y0 - NCC_R + i / pitch);
        if (tid < NCC_TT) s_t[tid] = tmpl[((size_t)fi * 3 + l) * NCC_TT + tid];
        __syncthreads();
        if (tid == 0) {
            int st = 0, stt = 0;
            for (int i = 0; i < NCC_TT; ++i) { st += s_t[i]; stt += s_t[i] * s_t[i]; }
            s_tsum[0] = st; s_tsum[1] = stt;
        }
        __syncthreads();
        const int st = s_tsum[0], stt = s_tsum[1];
        double key = -3.0;
        int idx = 0x7fffffff;
        for (int c = tid; c < cw * cw; c += 256) {
            const int ox = c % cw, oy = c / cw, x = x0 + ox, y = y0 + oy;
            if (x < 0 || y < 0 || x >= pyr.w[l] || y >= pyr.h[l]) continue;
            if (l == 2 && !(x == s_geom[0] && y == s_geom[1])) {
                const float fx = (float)((x + 0.5) * 4 - 0.5), fy = (float)((y + 0.5) * 4 - 0.5);
                if (!gate_contains(g, (double)fx, (double)fy)) continue;
            }
            const double kk = ncc_key(s_win, pitch, ox, oy, s_t, st, stt);
            if (kk > key) { key = kk; idx = c; } // c ascending per thread: first maximum kept
        }
        block_argmax(key, idx, s_key, s_idx);
        if (idx != 0x7fffffff) { bx = x0 + idx % cw; by = y0 + idx / cw; }
        bkey = key; // -3 when this level had no candidate (position carried over, as the CPU loop does)
    }
    if (tid == 0) {
        const bool ok = bkey >= 0.64 && gate_contains(g, (double)(float)bx, (double)(float)by);
        mt_valid[k] = ok ? 1 : 0;
        EkfKeypoint p;
        p.x = (float)bx;
        p.y = (float)by;
        mt_xy[k] = p;
        mt_dist[k] = ok ? (float)(1.0 - sqrt(bkey)) : 0.f;
    }
}

static Pyr pyr_of(const EkfEngine *e)
{
    Pyr p;
    for (int l = 0; l < 3; ++l) { p.px[l] = e->img.px[l]; p.w[l] = e->img.w[l]; p.h[l] = e->img.h[l]; }
    return p;
}

void launch_ncc_pyramid_on(EkfEngine *e, hipStream_t stream, uint8_t *const px[3], const uint8_t *d_raw, int stride, int channels)
{
    const int w = e->img.w[0], h = e->img.h[0];
    k_ncc_gray<<<dim3((w + 255) / 256, h), 256, 0, stream>>>(d_raw, w, h, stride, channels, px[0]);
    for (int l = 1; l < 3; ++l)
        if (e->img.w[l] > 0 && e->img.h[l] > 0)
            k_ncc_down<<<dim3((e->img.w[l] + 255) / 256, e->img.h[l]), 256, 0, stream>>>(px[l - 1], e->img.w[l - 1], px[l],
                                                                                         e->img.w[l], e->img.h[l]);
}

void launch_ncc_pyramid(EkfEngine *e, const uint8_t *d_raw, int stride, int channels)
{
    launch_ncc_pyramid_on(e, e->stream, e->img.px, d_raw, stride, channels);
}

void launch_ncc_capture(EkfEngine *e, const int *d_idx, const double *d_uv, int count)
{
    if (count > 0) k_ncc_capture<<<dim3(count, 3), 128, 0, e->stream>>>(pyr_of(e), d_idx, d_uv, e->d.tmpl);
}

void launch_match_compact_slots(EkfEngine *e, int n_pred, const EkfKeypoint *d_slot_xy);

void launch_match_ncc(EkfEngine *e, int n_pred)
{
    if (n_pred <= 0) {
        (void)hipMemsetAsync(e->d.counts + CNT_NMATCH, 0, sizeof(int), e->stream);
        return;
    }
    k_ncc_match<<<n_pred, 256, 0, e->stream>>>(pyr_of(e), e->d.plist, e->d.pred_uv, e->d.pred_S, e->d.tmpl, e->d.mt_valid,
                                               e->d.mt_xy, e->d.mt_dist, 0);
    launch_match_compact_slots(e, n_pred, e->d.mt_xy);
}

// sharded filter: the NCC search of the prediction slots [s_lo, s_hi) only (see launch_match_slots, kernels_match.hip); the per-slot
// tables are completed by an all-gather and compacted by launch_match_compact_slots on every rank
void launch_match_ncc_slots(EkfEngine *e, int s_lo, int s_hi)
{
    if (s_hi <= s_lo) return;
    k_ncc_match<<<s_hi - s_lo, 256, 0, e->stream>>>(pyr_of(e), e->d.plist, e->d.pred_uv, e->d.pred_S, e->d.tmpl, e->d.mt_valid, e->d.mt_xy,
                                                    e->d.mt_dist, s_lo);
}

} // namespace ekf
