// lk_oracle.cpp — TEST INFRASTRUCTURE: CPU restatement of the pyramidal Lucas-Kanade tracker the reference calls as
//   cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(11,11), 3,
//                            TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW)
// at src/frontend.cpp:150-153 (TrackLastFrame) and :358-361 (FindFeaturesInRight).  OpenCV 3.4.x is not under /root/reference:
// this follows its published algorithm (modules/video/src/lkpyramid.cpp: buildOpticalFlowPyramid -> pyrDown 5x5 [1 4 6 4 1]/16
// REFLECT_101; calcSharrDeriv 3x3 Scharr into shorts; LKTrackerInvoker with W_BITS = 14 fixed-point bilinear weights).
// PARITY UNPINNED against OpenCV (no golden vectors exist; its SSE/NEON/scalar builds already differ from each other in the
// float accumulation order).  This restatement DEFINES the window sums (A11, A12, A22, b1, b2, err) as exact integer sums
// converted to float once; every other float expression is evaluated left to right without contraction.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "oracle.h"

namespace {

inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline int cv_round(float v) { return (int)lrintf(v); }
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

struct Img { const uint8_t* p; int w, h, step; };
inline int px(const Img& I, int x, int y) { return I.p[(size_t)reflect101(y, I.h) * I.step + reflect101(x, I.w)]; }   // REFLECT_101 border

// Scharr derivative at an in-image pixel (calcSharrDeriv: reflect-101 neighbours); zero outside the image (BORDER_CONSTANT)
inline void scharr(const Img& I, int x, int y, int& dx, int& dy) {
    if (x < 0 || x >= I.w || y < 0 || y >= I.h) { dx = dy = 0; return; }
    int t0[3], t1[3];
    for (int k = 0; k < 3; k++) {
        const int xx = x + k - 1;
        const int a = px(I, xx, y - 1), b = px(I, xx, y), c = px(I, xx, y + 1);
        t0[k] = (a + c) * 3 + b * 10;
        t1[k] = c - a;
    }
    dx = t0[2] - t0[0];
    dy = (t1[2] + t1[0]) * 3 + t1[1] * 10;
}

}  // namespace

extern "C" {

/* cv::pyrDown, 8UC1: dst (w+1)/2 x (h+1)/2 */
int orc_pyr_down(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep) {
    if (!src || !dst || w < 1 || h < 1) return -1;
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    Img I{src, w, h, sstep};
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int s = 0;
            for (int j = 0; j < 5; j++) {
                int r = 0;
                for (int i = 0; i < 5; i++) r += k[i] * px(I, 2 * x + i - 2, 2 * y + j - 2);
                s += k[j] * r;
            }
            dst[(size_t)y * dstep + x] = (uint8_t)((s + 128) >> 8);
        }
    return 0;
}

/* calcOpticalFlowPyrLK with OPTFLOW_USE_INITIAL_FLOW; next_pts in: initial guess, out: tracked position.
 * status[i] = 1 tracked; err[i] (nullable) = mean absolute patch difference / 32 at level 0.  Returns levels used - 1. */
int orc_lk_track(const uint8_t* prev, const uint8_t* next, int rows, int cols, int pstep, int nstep,
                 const float* prev_pts, float* next_pts, int n, int win, int max_level, int max_iters, float eps,
                 float min_eig_threshold, uint8_t* status, float* err) {
    if (!prev || !next || rows < 1 || cols < 1 || n < 0 || win < 3 || win > 31 || max_level < 0 || !status) return -1;
    // pyramids (buildOpticalFlowPyramid: stop when the NEXT level would not be larger than the window)
    std::vector<std::vector<uint8_t>> bufP, bufN;
    std::vector<Img> P, N;
    P.push_back({prev, cols, rows, pstep}); N.push_back({next, cols, rows, nstep});
    int levels = 0;
    {
        int w = cols, h = rows;
        for (int l = 0; l <= max_level; l++) {
            if (l > 0) {
                const int dw = (P[l - 1].w + 1) / 2, dh = (P[l - 1].h + 1) / 2;
                bufP.emplace_back((size_t)dw * dh); bufN.emplace_back((size_t)dw * dh);
                orc_pyr_down(P[l - 1].p, P[l - 1].w, P[l - 1].h, P[l - 1].step, bufP.back().data(), dw);
                orc_pyr_down(N[l - 1].p, N[l - 1].w, N[l - 1].h, N[l - 1].step, bufN.back().data(), dw);
                P.push_back({bufP.back().data(), dw, dh, dw}); N.push_back({bufN.back().data(), dw, dh, dw});
            }
            levels = l;
            w = (w + 1) / 2; h = (h + 1) / 2;
            if (w <= win || h <= win) break;
        }
    }
    const float eps2 = eps * eps;
    const float halfWin = (float)(win - 1) * 0.5f;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    for (int i = 0; i < n; i++) { status[i] = 1; if (err) err[i] = 0.f; }
    std::vector<short> Ipatch((size_t)win * win), dIx((size_t)win * win), dIy((size_t)win * win);
    for (int level = levels; level >= 0; level--) {
        const Img& I = P[level]; const Img& J = N[level];
        const float sc = (float)(1. / (1 << level));
        for (int pi = 0; pi < n; pi++) {
            float prx = prev_pts[2 * pi] * sc, pry = prev_pts[2 * pi + 1] * sc;
            float nx, ny;
            if (level == levels) { nx = next_pts[2 * pi] * sc; ny = next_pts[2 * pi + 1] * sc; }      // OPTFLOW_USE_INITIAL_FLOW
            else { nx = next_pts[2 * pi] * 2.f; ny = next_pts[2 * pi + 1] * 2.f; }
            next_pts[2 * pi] = nx; next_pts[2 * pi + 1] = ny;
            prx -= halfWin; pry -= halfWin;
            const int ipx = cv_floor(prx), ipy = cv_floor(pry);
            if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
                if (level == 0) { status[pi] = 0; if (err) err[pi] = 0.f; }
                continue;
            }
            float a = prx - ipx, b = pry - ipy;
            int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    const int X = ipx + x, Y = ipy + y;
                    const int ival = descale(px(I, X, Y) * iw00 + px(I, X + 1, Y) * iw01 + px(I, X, Y + 1) * iw10 + px(I, X + 1, Y + 1) * iw11, W_BITS - 5);
                    int d00x, d00y, d01x, d01y, d10x, d10y, d11x, d11y;
                    scharr(I, X, Y, d00x, d00y); scharr(I, X + 1, Y, d01x, d01y); scharr(I, X, Y + 1, d10x, d10y); scharr(I, X + 1, Y + 1, d11x, d11y);
                    const int ixval = descale(d00x * iw00 + d01x * iw01 + d10x * iw10 + d11x * iw11, W_BITS);
                    const int iyval = descale(d00y * iw00 + d01y * iw01 + d10y * iw10 + d11y * iw11, W_BITS);
                    Ipatch[y * win + x] = (short)ival; dIx[y * win + x] = (short)ixval; dIy[y * win + x] = (short)iyval;
                    sA11 += (int64_t)ixval * ixval; sA12 += (int64_t)ixval * iyval; sA22 += (int64_t)iyval * iyval;
                }
            const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
            if (minEig < min_eig_threshold || D < 1.19209290e-07f) {
                if (level == 0) status[pi] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= halfWin; ny -= halfWin;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_iters; j++) {
                const int inx = cv_floor(nx), iny = cv_floor(ny);
                if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                    if (level == 0) status[pi] = 0;
                    break;
                }
                a = nx - inx; b = ny - iny;
                iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t sb1 = 0, sb2 = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        const int X = inx + x, Y = iny + y;
                        const int diff = descale(px(J, X, Y) * iw00 + px(J, X + 1, Y) * iw01 + px(J, X, Y + 1) * iw10 + px(J, X + 1, Y + 1) * iw11, W_BITS - 5) -
                                         Ipatch[y * win + x];
                        sb1 += (int64_t)diff * dIx[y * win + x]; sb2 += (int64_t)diff * dIy[y * win + x];
                    }
                const float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
                const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
                nx += dx; ny += dy;
                next_pts[2 * pi] = nx + halfWin; next_pts[2 * pi + 1] = ny + halfWin;
                if (dx * dx + dy * dy <= eps2) break;
                if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                    next_pts[2 * pi] -= dx * 0.5f; next_pts[2 * pi + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[pi] && err && level == 0) {          // patch error at the final position; also the last bounds test
                const float fx = next_pts[2 * pi] - halfWin, fy = next_pts[2 * pi + 1] - halfWin;
                const int inx = cv_floor(fx), iny = cv_floor(fy);
                if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) { status[pi] = 0; continue; }
                const float aa = fx - inx, bb = fy - iny;
                iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
                iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
                iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t se = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        const int X = inx + x, Y = iny + y;
                        const int diff = descale(px(J, X, Y) * iw00 + px(J, X + 1, Y) * iw01 + px(J, X, Y + 1) * iw10 + px(J, X + 1, Y + 1) * iw11, W_BITS - 5) -
                                         Ipatch[y * win + x];
                        se += diff < 0 ? -diff : diff;
                    }
                err[pi] = (float)se * (1.f / (32 * win * win));
            }
        }
    }
    return levels;
}

}  // extern "C"
