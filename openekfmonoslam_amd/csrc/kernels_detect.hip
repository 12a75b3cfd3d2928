// kernels_detect.hip -- candidate pixels for NEW map features, the detector half of detectNewImageFeatures
// (EKF/DetectNewImageFeatures.cpp:337-367) for matcher mode B.  The reference masks the image with the uncertainty
// ellipses of the current predictions (buildImageMask :102-127), runs an OpenCV detector (STAR, third-party, absent
// here) and spreads the picks over image zones (:171-330, done on the host in engine.cpp).  This build's detector is
// its own: a Harris-type corner measure in INTEGER arithmetic on the gray level-0 image,
//     R = 16 (Sxx Syy - Sxy^2) - (Sxx + Syy)^2,   S.. = 5x5 sums of products of 3x3 Sobel gradients,
// reduced to the best unmasked pixel of every 16x16 cell (ties: raster order).  Integer => bit-identical to the CPU
// definition in oracle/ekf_oracle.c.  HBM traffic: one read of the frame; everything else lives in LDS.
#include "engine.h"
#include "gate.h"

namespace ekf {

constexpr int DC = 16;      // cell edge
constexpr int DBORDER = 16; // no candidates closer than this to the frame edge

// gate (foci form) + bounding radius of every prediction of the last full prediction, in prediction order
__global__ void __launch_bounds__(256)
k_gate_snapshot(const int *plist, int n_pred, const double *uv_tab, const double *S_tab, double *gates)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_pred) return;
    const int fi = plist[k];
    float axes[2];
    double angle;
    ellipse_from_cov(S_tab + 4 * fi, axes, &angle);
    const int aw = (int)rintf(axes[0]), ah = (int)rintf(axes[1]);
    Gate g;
    const float cx = (float)uv_tab[2 * fi], cy = (float)uv_tab[2 * fi + 1];
    gate_from_ellipse(cx, cy, aw, ah, angle, &g);
    double *o = gates + 8 * (size_t)k;
    o[0] = g.f1x; o[1] = g.f1y; o[2] = g.f2x; o[3] = g.f2y; o[4] = g.two_major;
    o[5] = (double)cx; o[6] = (double)cy; o[7] = (double)(aw > ah ? aw : ah);
}

__global__ void __launch_bounds__(256)
k_detect_cells(const uint8_t *img, int w, int h, const double *gates, int n_gates, int cells_x, long long *cell_resp,
               int *cell_xy)
{
    __shared__ int sg[DC + 6][DC + 6];      // gray, halo 3
    __shared__ short sIx[DC + 4][DC + 4];   // Sobel, halo 2
    __shared__ short sIy[DC + 4][DC + 4];
    __shared__ double sGate[64][8];
    __shared__ long long s_best[4];
    __shared__ int s_idx[4];
    const int tid = threadIdx.x;
    const int cx0 = (blockIdx.x % cells_x) * DC, cy0 = (blockIdx.x / cells_x) * DC;
    for (int i = tid; i < (DC + 6) * (DC + 6); i += 256) {
        const int ly = i / (DC + 6), lx = i % (DC + 6);
        const int x = min(max(cx0 + lx - 3, 0), w - 1), y = min(max(cy0 + ly - 3, 0), h - 1);
        sg[ly][lx] = img[(size_t)y * w + x];
    }
    __syncthreads();
    for (int i = tid; i < (DC + 4) * (DC + 4); i += 256) {
        const int ly = i / (DC + 4), lx = i % (DC + 4); // centre at sg[ly + 1][lx + 1]
        const int a = sg[ly][lx], b = sg[ly][lx + 1], c = sg[ly][lx + 2];
        const int d = sg[ly + 1][lx], f = sg[ly + 1][lx + 2];
        const int g = sg[ly + 2][lx], hh = sg[ly + 2][lx + 1], k = sg[ly + 2][lx + 2];
        sIx[ly][lx] = (short)((c + 2 * f + k) - (a + 2 * d + g));
        sIy[ly][lx] = (short)((g + 2 * hh + k) - (a + 2 * b + c));
    }
    __syncthreads();
    const int lx = tid % DC, ly = tid / DC;
    const int x = cx0 + lx, y = cy0 + ly;
    long long resp = -1;
    bool ok = x >= DBORDER && y >= DBORDER && x < w - DBORDER && y < h - DBORDER;
    if (ok) {
        long long sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const int ix = sIx[ly + dy][lx + dx], iy = sIy[ly + dy][lx + dx];
                sxx += ix * ix;
                syy += iy * iy;
                sxy += ix * iy;
            }
        const long long tr = sxx + syy;
        resp = 16 * (sxx * syy - sxy * sxy) - tr * tr;
    }
    // mask: inside any prediction's gate ellipse (buildImageMask)
    const double px = (double)(float)x, py = (double)(float)y;
    for (int base = 0; base < n_gates; base += 64) {
        const int cnt = min(64, n_gates - base);
        __syncthreads();
        for (int i = tid; i < cnt * 8; i += 256) sGate[i / 8][i % 8] = gates[(size_t)(base + i / 8) * 8 + i % 8];
        __syncthreads();
        if (ok)
            for (int gk = 0; gk < cnt; ++gk) {
                const double *g = sGate[gk];
                if (fabs(px - g[5]) > g[7] + 1.0 || fabs(py - g[6]) > g[7] + 1.0) continue;
                const double a1x = px - g[0], a1y = py - g[1], a2x = px - g[2], a2y = py - g[3];
                if (sqrt(a1x * a1x + a1y * a1y) + sqrt(a2x * a2x + a2y * a2y) <= g[4]) { ok = false; break; }
            }
    }
    if (!ok) resp = -1;
    // block argmax, ties -> smallest pixel index in the cell (raster order)
    int idx = tid;
    const int lane = tid & 63, wv = tid >> 6;
    for (int o = 32; o > 0; o >>= 1) {
        const long long r2 = __shfl_down(resp, o);
        const int i2 = __shfl_down(idx, o);
        if (r2 > resp || (r2 == resp && i2 < idx)) { resp = r2; idx = i2; }
    }
    if (lane == 0) { s_best[wv] = resp; s_idx[wv] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < 4; ++q)
            if (s_best[q] > resp || (s_best[q] == resp && s_idx[q] < idx)) { resp = s_best[q]; idx = s_idx[q]; }
        cell_resp[blockIdx.x] = resp;
        cell_xy[2 * blockIdx.x] = cx0 + idx % DC;
        cell_xy[2 * blockIdx.x + 1] = cy0 + idx / DC;
    }
}

void launch_gate_snapshot(EkfEngine *e, int n_pred)
{
    if (n_pred > 0)
        k_gate_snapshot<<<(n_pred + 255) / 256, 256, 0, e->stream>>>(e->d.plist, n_pred, e->d.pred_uv, e->d.pred_S, e->d.gates);
}

void launch_detect_cells(EkfEngine *e, int n_gates, int cells_x, int cells_y, long long *d_resp, int *d_xy)
{
    k_detect_cells<<<cells_x * cells_y, 256, 0, e->stream>>>(e->img.px[0], e->img.w[0], e->img.h[0], e->d.gates, n_gates,
                                                             cells_x, d_resp, d_xy);
}

} // namespace ekf
