// lk.hip — pyramidal Lucas-Kanade tracker on gfx950, the operator the reference calls as
//   cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(11,11), 3,
//                            TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW)
// in Frontend::TrackLastFrame (src/frontend.cpp:150-153) and Frontend::FindFeaturesInRight (:358-361)  [SURVEY.md §8(f) rank 1].
//
// Arithmetic (OpenCV lkpyramid.cpp, restated by the test oracle): pyrDown 5x5 [1 4 6 4 1]/16 REFLECT_101, 3x3 Scharr
// derivatives as shorts (zero outside the image), W_BITS = 14 fixed-point bilinear patch extraction, 2x2 normal equations.
// The window sums are exact integer sums converted to float once (the restatement's definition), so the tracker is bit-exact
// against the oracle; every float expression uses explicit non-fused IEEE operations.
//
// One wave per point, all pyramid levels inside the wave (coarse to fine), no global scratch: the (win+3)^2 patch of I, its
// Scharr derivatives and the (win+1)^2 window of J live in LDS; the template patch (I, Ix, Iy at the window pixels) stays in
// registers (2 window pixels per lane).  Sums over the window are DPP wave reductions of 32-bit halves (exact).
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace myslam_hip {

constexpr int LK_MAXL = 6;          // pyramid levels supported (the reference uses maxLevel = 3)
constexpr int LK_MAXWIN = 15;       // window sizes up to 15x15 fit 4 window pixels per lane (11x11 -> 2 per lane)
constexpr int LK_WPL = 4;

struct LkGeom {
    int levels;                     // top level index actually used
    int w[LK_MAXL + 1], h[LK_MAXL + 1];
    size_t off[LK_MAXL + 1];        // byte offset of level l >= 1 inside one image's pyramid block
    size_t bytes;                   // pyramid block bytes per image (levels >= 1)
};

__device__ __forceinline__ int lk_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}

// cv::pyrDown 8UC1: one thread per destination pixel
__global__ __launch_bounds__(256) void k_pyr_down(const uint8_t* __restrict__ src, int sw, int sh, int sstep, size_t sstride,
                                                  uint8_t* __restrict__ dst, int dw, int dh, size_t dstride) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const uint8_t* S = src + (size_t)b * sstride;
    int xs[5];
#pragma unroll
    for (int i = 0; i < 5; i++) xs[i] = lk_reflect101(2 * x + i - 2, sw);
    int s = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint8_t* row = S + (size_t)lk_reflect101(2 * y + j - 2, sh) * sstep;
        const int r = row[xs[0]] + 4 * row[xs[1]] + 6 * row[xs[2]] + 4 * row[xs[3]] + row[xs[4]];
        s += (j == 0 || j == 4) ? r : (j == 2 ? 6 * r : 4 * r);
    }
    dst[(size_t)b * dstride + (size_t)y * dw + x] = (uint8_t)((s + 128) >> 8);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int lk_dpp_add(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int lk_wave_sum(int v) {                 // total in lane 63, broadcast to the wave
    v = lk_dpp_add<0xB1, 0xf>(v); v = lk_dpp_add<0x4E, 0xf>(v); v = lk_dpp_add<0x141, 0xf>(v); v = lk_dpp_add<0x140, 0xf>(v);
    v = lk_dpp_add<0x142, 0xa>(v); v = lk_dpp_add<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
// exact wave sum of per-lane int32 partials whose total may exceed 32 bits: sum the high part and the low byte separately
__device__ __forceinline__ long long lk_wave_sum64(int v) {
    const int hi = lk_wave_sum(v >> 8), lo = lk_wave_sum(v & 0xff);
    return (long long)hi * 256 + lo;
}

struct LkArgs {
    const uint8_t* prev; const uint8_t* next; int rows, cols, pstep, nstep; size_t pstride, nstride;
    const uint8_t* pyrP; const uint8_t* pyrN;        // levels >= 1 of every image
    LkGeom g;
    const float* prev_pts; float* next_pts; const int32_t* counts; int n_fixed, cap;
    int win, max_iters; float eps2, min_eig;
    uint8_t* status; float* err;
};

__global__ __launch_bounds__(256) void k_lk_track(LkArgs a) {
    constexpr int PI_MAX = LK_MAXWIN + 3, PJ_MAX = LK_MAXWIN + 1;
    __shared__ uint8_t s_I[4][PI_MAX * PI_MAX];
    __shared__ short s_dx[4][PJ_MAX * PJ_MAX], s_dy[4][PJ_MAX * PJ_MAX];
    __shared__ uint8_t s_J[4][PJ_MAX * PJ_MAX];
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pi = blockIdx.x * 4 + wave;
    const int n = a.counts ? a.counts[b] : a.n_fixed;
    if (pi >= n) return;                                            // wave-uniform
    const int win = a.win, ww = win * win, pI = win + 3, pJ = win + 1;
    const float halfWin = (float)(win - 1) * 0.5f;
    const float* pp = a.prev_pts + ((size_t)b * a.cap + pi) * 2;
    float* np_ = a.next_pts + ((size_t)b * a.cap + pi) * 2;
    const float p0x = pp[0], p0y = pp[1];
    float outx = np_[0], outy = np_[1];                             // nextPts[ptidx], carried across levels
    int status = 1; float errv = 0.f;
    uint8_t* sI = s_I[wave]; short* sdx = s_dx[wave]; short* sdy = s_dy[wave]; uint8_t* sJ = s_J[wave];
    constexpr int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);

    for (int level = a.g.levels; level >= 0; level--) {
        const int lw = a.g.w[level], lh = a.g.h[level];
        const uint8_t* I; const uint8_t* J; int istep, jstep;
        if (level == 0) { I = a.prev + (size_t)b * a.pstride; J = a.next + (size_t)b * a.nstride; istep = a.pstep; jstep = a.nstep; }
        else { I = a.pyrP + (size_t)b * a.g.bytes + a.g.off[level]; J = a.pyrN + (size_t)b * a.g.bytes + a.g.off[level]; istep = jstep = lw; }
        const float sc = (float)(1. / (1 << level));
        float prx = __fmul_rn(p0x, sc), pry = __fmul_rn(p0y, sc);
        float nx, ny;
        if (level == a.g.levels) { nx = __fmul_rn(outx, sc); ny = __fmul_rn(outy, sc); }          // OPTFLOW_USE_INITIAL_FLOW
        else { nx = __fmul_rn(outx, 2.f); ny = __fmul_rn(outy, 2.f); }
        outx = nx; outy = ny;
        prx = __fsub_rn(prx, halfWin); pry = __fsub_rn(pry, halfWin);
        const int ipx = (int)floorf(prx), ipy = (int)floorf(pry);
        if (ipx < -win || ipx >= lw || ipy < -win || ipy >= lh) {
            if (level == 0) { status = 0; errv = 0.f; }
            continue;
        }
        // ---- stage the (win+3)^2 patch of I (origin ipx-1, ipy-1), REFLECT_101 ----
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < pI * pI; i += 64) {
            const int r = i / pI, c = i - r * pI;
            sI[i] = I[(size_t)lk_reflect101(ipy - 1 + r, lh) * istep + lk_reflect101(ipx - 1 + c, lw)];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- Scharr derivatives at the (win+1)^2 positions the bilinear taps touch; zero outside the image ----
        for (int i = lane; i < pJ * pJ; i += 64) {
            const int r = i / pJ, c = i - r * pJ;
            const int X = ipx + c, Y = ipy + r;
            int dx = 0, dy = 0;
            if (X >= 0 && X < lw && Y >= 0 && Y < lh) {
                const uint8_t* q = sI + r * pI + c;                 // top-left of the 3x3 neighbourhood
                const int a0 = q[0], a1 = q[1], a2 = q[2], b0 = q[pI], b2 = q[pI + 2], c0 = q[2 * pI], c1 = q[2 * pI + 1], c2 = q[2 * pI + 2];
                const int b1 = q[pI + 1];
                const int t00 = (a0 + c0) * 3 + b0 * 10, t02 = (a2 + c2) * 3 + b2 * 10;
                const int t10 = c0 - a0, t11 = c1 - a1, t12 = c2 - a2;
                (void)b1;
                dx = t02 - t00;
                dy = (t12 + t10) * 3 + t11 * 10;
            }
            sdx[i] = (short)dx; sdy[i] = (short)dy;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- template patch: I, Ix, Iy at the window pixels (registers), normal matrix ----
        float fa = __fsub_rn(prx, (float)ipx), fb = __fsub_rn(pry, (float)ipy);
        int iw00 = __float2int_rn(__fmul_rn(__fmul_rn(__fsub_rn(1.f, fa), __fsub_rn(1.f, fb)), (float)(1 << W_BITS)));
        int iw01 = __float2int_rn(__fmul_rn(__fmul_rn(fa, __fsub_rn(1.f, fb)), (float)(1 << W_BITS)));
        int iw10 = __float2int_rn(__fmul_rn(__fmul_rn(__fsub_rn(1.f, fa), fb), (float)(1 << W_BITS)));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int Ip[LK_WPL], Ix[LK_WPL], Iy[LK_WPL];
        int s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
        for (int k = 0; k < LK_WPL; k++) {
            const int i = lane + 64 * k;
            Ip[k] = Ix[k] = Iy[k] = 0;
            if (i < ww) {
                const int y = i / win, x = i - y * win;
                const uint8_t* q = sI + (y + 1) * pI + x + 1;
                Ip[k] = (q[0] * iw00 + q[1] * iw01 + q[pI] * iw10 + q[pI + 1] * iw11 + (1 << (W_BITS - 6))) >> (W_BITS - 5);
                const short* dxp = sdx + y * pJ + x; const short* dyp = sdy + y * pJ + x;
                Ix[k] = (dxp[0] * iw00 + dxp[1] * iw01 + dxp[pJ] * iw10 + dxp[pJ + 1] * iw11 + (1 << (W_BITS - 1))) >> W_BITS;
                Iy[k] = (dyp[0] * iw00 + dyp[1] * iw01 + dyp[pJ] * iw10 + dyp[pJ + 1] * iw11 + (1 << (W_BITS - 1))) >> W_BITS;
                s11 += Ix[k] * Ix[k]; s12 += Ix[k] * Iy[k]; s22 += Iy[k] * Iy[k];
            }
        }
        const float A11 = __fmul_rn((float)lk_wave_sum64(s11), FLT_SCALE), A12 = __fmul_rn((float)lk_wave_sum64(s12), FLT_SCALE);
        const float A22 = __fmul_rn((float)lk_wave_sum64(s22), FLT_SCALE);
        float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
        const float dA = __fsub_rn(A11, A22);
        const float minEig = __fdiv_rn(__fsub_rn(__fadd_rn(A22, A11), __fsqrt_rn(__fadd_rn(__fmul_rn(dA, dA), __fmul_rn(__fmul_rn(4.f, A12), A12)))),
                                       (float)(2 * win * win));
        if (minEig < a.min_eig || D < 1.19209290e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = __fdiv_rn(1.f, D);
        nx = __fsub_rn(nx, halfWin); ny = __fsub_rn(ny, halfWin);
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < a.max_iters; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= lw || iny < -win || iny >= lh) {
                if (level == 0) status = 0;
                break;
            }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < pJ * pJ; i += 64) {
                const int r = i / pJ, c = i - r * pJ;
                sJ[i] = J[(size_t)lk_reflect101(iny + r, lh) * jstep + lk_reflect101(inx + c, lw)];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            fa = __fsub_rn(nx, (float)inx); fb = __fsub_rn(ny, (float)iny);
            iw00 = __float2int_rn(__fmul_rn(__fmul_rn(__fsub_rn(1.f, fa), __fsub_rn(1.f, fb)), (float)(1 << W_BITS)));
            iw01 = __float2int_rn(__fmul_rn(__fmul_rn(fa, __fsub_rn(1.f, fb)), (float)(1 << W_BITS)));
            iw10 = __float2int_rn(__fmul_rn(__fmul_rn(__fsub_rn(1.f, fa), fb), (float)(1 << W_BITS)));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int sb1 = 0, sb2 = 0;
#pragma unroll
            for (int k = 0; k < LK_WPL; k++) {
                const int i = lane + 64 * k;
                if (i < ww) {
                    const int y = i / win, x = i - y * win;
                    const uint8_t* q = sJ + y * pJ + x;
                    const int diff = ((q[0] * iw00 + q[1] * iw01 + q[pJ] * iw10 + q[pJ + 1] * iw11 + (1 << (W_BITS - 6))) >> (W_BITS - 5)) - Ip[k];
                    sb1 += diff * Ix[k]; sb2 += diff * Iy[k];
                }
            }
            const float b1 = __fmul_rn((float)lk_wave_sum64(sb1), FLT_SCALE), b2 = __fmul_rn((float)lk_wave_sum64(sb2), FLT_SCALE);
            const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, b2), __fmul_rn(A22, b1)), D);
            const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, b1), __fmul_rn(A11, b2)), D);
            nx = __fadd_rn(nx, dx); ny = __fadd_rn(ny, dy);
            outx = __fadd_rn(nx, halfWin); outy = __fadd_rn(ny, halfWin);
            if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) <= a.eps2) break;
            if (j > 0 && fabsf(__fadd_rn(dx, pdx)) < 0.01f && fabsf(__fadd_rn(dy, pdy)) < 0.01f) {
                outx = __fsub_rn(outx, __fmul_rn(dx, 0.5f)); outy = __fsub_rn(outy, __fmul_rn(dy, 0.5f));
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {          // patch error at the final position; also the last bounds test (err is always requested)
            const float fx = __fsub_rn(outx, halfWin), fy = __fsub_rn(outy, halfWin);
            const int inx = (int)floorf(fx), iny = (int)floorf(fy);
            if (inx < -win || inx >= lw || iny < -win || iny >= lh) { status = 0; continue; }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < pJ * pJ; i += 64) {
                const int r = i / pJ, c = i - r * pJ;
                sJ[i] = J[(size_t)lk_reflect101(iny + r, lh) * jstep + lk_reflect101(inx + c, lw)];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            fa = __fsub_rn(fx, (float)inx); fb = __fsub_rn(fy, (float)iny);
            iw00 = __float2int_rn(__fmul_rn(__fmul_rn(__fsub_rn(1.f, fa), __fsub_rn(1.f, fb)), (float)(1 << W_BITS)));
            iw01 = __float2int_rn(__fmul_rn(__fmul_rn(fa, __fsub_rn(1.f, fb)), (float)(1 << W_BITS)));
            iw10 = __float2int_rn(__fmul_rn(__fmul_rn(__fsub_rn(1.f, fa), fb), (float)(1 << W_BITS)));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int se = 0;
#pragma unroll
            for (int k = 0; k < LK_WPL; k++) {
                const int i = lane + 64 * k;
                if (i < ww) {
                    const int y = i / win, x = i - y * win;
                    const uint8_t* q = sJ + y * pJ + x;
                    const int diff = ((q[0] * iw00 + q[1] * iw01 + q[pJ] * iw10 + q[pJ + 1] * iw11 + (1 << (W_BITS - 6))) >> (W_BITS - 5)) - Ip[k];
                    se += diff < 0 ? -diff : diff;
                }
            }
            errv = __fmul_rn((float)lk_wave_sum64(se), 1.f / (float)(32 * win * win));
        }
    }
    if (lane == 0) {
        np_[0] = outx; np_[1] = outy;
        a.status[(size_t)b * a.cap + pi] = (uint8_t)status;
        if (a.err) a.err[(size_t)b * a.cap + pi] = errv;
    }
}

}  // namespace myslam_hip

using namespace myslam_hip;

struct myslam_lk {
    hipStream_t stream = nullptr;
    int win = 11, max_level = 3, max_iters = 30; float eps = 0.01f, min_eig = 1e-4f;
    int rows = 0, cols = 0, batchCap = 0;
    LkGeom g{};
    uint8_t *d_pyrP = nullptr, *d_pyrN = nullptr;
    // host-entry staging
    uint8_t* d_img = nullptr; size_t imgBytes = 0; float* d_pts = nullptr; uint8_t* d_st = nullptr; int ptsCap = 0;
    // myslam_lk_track_cached / myslam_lk_prefetch: two (image, pyramid) slots that remember WHICH image they hold (a caller's token): the
    // `next` image of one tracked frame is the `prev` image of the following one (Frontend::TrackLastFrame, frontend.cpp:150-153), and the
    // frame after that can be uploaded while the current one is still being optimised
    struct Slot { uint8_t* img = nullptr; size_t imgBytes = 0; uint8_t* pyr = nullptr; size_t pyrBytes = 0; uint64_t tok = 0; int rows = 0, cols = 0, step = 0; };
    Slot slot[2]; int lastNext = 0;
    // the cached call's points travel as ONE pinned upload [prev_pts | next_pts] and ONE pinned download [next_pts | err | status]
    uint8_t* h_pin = nullptr; uint8_t* d_stage = nullptr; int stageCap = 0;
};

static int lk_plan(myslam_lk* h, int rows, int cols) {
    LkGeom g{};
    g.w[0] = cols; g.h[0] = rows; g.off[0] = 0; g.bytes = 0; g.levels = 0;
    int w = cols, hh = rows;
    for (int l = 0; l <= h->max_level; l++) {        // buildOpticalFlowPyramid: stop when the NEXT level would not exceed the window
        if (l > 0) {
            g.w[l] = (g.w[l - 1] + 1) / 2; g.h[l] = (g.h[l - 1] + 1) / 2;
            g.off[l] = g.bytes; g.bytes += ((size_t)g.w[l] * g.h[l] + 255) & ~(size_t)255;
        }
        g.levels = l;
        w = (w + 1) / 2; hh = (hh + 1) / 2;
        if (w <= h->win || hh <= h->win) break;
    }
    h->g = g; h->rows = rows; h->cols = cols; h->batchCap = 0;
    return MYSLAM_OK;
}

static int lk_ensure(myslam_lk* h, int batch, int rows, int cols) {
    if (rows != h->rows || cols != h->cols) { MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream)); lk_plan(h, rows, cols); }
    if (batch > h->batchCap) {
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->d_pyrP) (void)hipFree(h->d_pyrP);
        if (h->d_pyrN) (void)hipFree(h->d_pyrN);
        h->d_pyrP = h->d_pyrN = nullptr;
        const size_t nb = std::max<size_t>(256, (size_t)batch * h->g.bytes);
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_pyrP, nb));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_pyrN, nb));
        h->batchCap = batch;
    }
    return MYSLAM_OK;
}

// levels 1 .. of `batch` images (level 0 at src0, row pitch step0, image stride stride0) into pyr
static void lk_pyramid(myslam_lk* h, const uint8_t* src0, int step0, size_t stride0, uint8_t* pyr, int batch) {
    const LkGeom& g = h->g;
    for (int l = 1; l <= g.levels; l++) {
        const uint8_t* src = (l == 1) ? src0 : pyr + g.off[l - 1];
        const int sstep = (l == 1) ? step0 : g.w[l - 1];
        const size_t sstride = (l == 1) ? stride0 : g.bytes;
        hipLaunchKernelGGL(k_pyr_down, dim3((g.w[l] + 63) / 64, (g.h[l] + 3) / 4, batch), dim3(256), 0, h->stream, src, g.w[l - 1], g.h[l - 1], sstep, sstride,
                           pyr + g.off[l], g.w[l], g.h[l], g.bytes);
    }
}

static int lk_track_launch(myslam_lk* h, const uint8_t* d_prev, const uint8_t* d_next, const uint8_t* pyrP, const uint8_t* pyrN, int batch, int rows, int cols,
                           int pstep, int nstep, size_t pstride, size_t nstride, const float* d_prev_pts, float* d_next_pts, const int32_t* d_counts,
                           int n_fixed, int cap, uint8_t* d_status, float* d_err) {
    LkArgs a{d_prev, d_next, rows, cols, pstep, nstep, pstride, nstride, pyrP, pyrN, h->g, d_prev_pts, d_next_pts, d_counts, n_fixed, cap,
             h->win, h->max_iters, h->eps * h->eps, h->min_eig, d_status, d_err};
    hipLaunchKernelGGL(k_lk_track, dim3((cap + 3) / 4, batch), dim3(256), 0, h->stream, a);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

static int lk_run(myslam_lk* h, const uint8_t* d_prev, const uint8_t* d_next, int batch, int rows, int cols, int pstep, int nstep,
                  size_t pstride, size_t nstride, const float* d_prev_pts, float* d_next_pts, const int32_t* d_counts, int n_fixed, int cap,
                  uint8_t* d_status, float* d_err) {
    int rc = lk_ensure(h, batch, rows, cols);
    if (rc) return rc;
    lk_pyramid(h, d_prev, pstep, pstride, h->d_pyrP, batch);
    lk_pyramid(h, d_next, nstep, nstride, h->d_pyrN, batch);
    return lk_track_launch(h, d_prev, d_next, h->d_pyrP, h->d_pyrN, batch, rows, cols, pstep, nstep, pstride, nstride, d_prev_pts, d_next_pts, d_counts, n_fixed,
                           cap, d_status, d_err);
}

// ---- cached form: slots that remember their image ----
static int lk_plan_cached(myslam_lk* h, int rows, int cols) {
    if (rows == h->rows && cols == h->cols) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    lk_plan(h, rows, cols);
    h->slot[0].tok = h->slot[1].tok = 0;                     // another geometry: nothing cached is usable
    return MYSLAM_OK;
}

// image -> slot i (one contiguous copy with the caller's row pitch, then the pyramid levels), asynchronous on the handle's stream
static int lk_fill_slot(myslam_lk* h, int i, const uint8_t* img, uint64_t tok, int rows, int cols, int step) {
    myslam_lk::Slot& S = h->slot[i];
    const size_t ib = (size_t)rows * step, pb = std::max<size_t>(256, h->g.bytes);
    if (ib > S.imgBytes || pb > S.pyrBytes) {
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (S.img) (void)hipFree(S.img);
        if (S.pyr) (void)hipFree(S.pyr);
        S = myslam_lk::Slot();
        MYSLAM_HIP_CHECK(hipMalloc((void**)&S.img, ib)); S.imgBytes = ib;
        MYSLAM_HIP_CHECK(hipMalloc((void**)&S.pyr, pb)); S.pyrBytes = pb;
    }
    S.tok = 0;                                               // not valid until everything below is enqueued
    MYSLAM_HIP_CHECK(hipMemcpyAsync(S.img, img, ib - (size_t)(step - cols), hipMemcpyHostToDevice, h->stream));     // the last row ends at its last pixel
    lk_pyramid(h, S.img, step, ib, S.pyr, 1);
    MYSLAM_HIP_CHECK(hipGetLastError());
    S.tok = tok; S.rows = rows; S.cols = cols; S.step = step;
    return MYSLAM_OK;
}

static int lk_find_slot(const myslam_lk* h, uint64_t tok, int rows, int cols, int step) {
    if (!tok) return -1;
    for (int i = 0; i < 2; i++) if (h->slot[i].tok == tok && h->slot[i].rows == rows && h->slot[i].cols == cols && h->slot[i].step == step) return i;
    return -1;
}

extern "C" {

int myslam_lk_create(myslam_lk** out, int win, int max_level, int max_iters, float eps, float min_eig_threshold) {
    if (!out || win < 3 || win > LK_MAXWIN || (win & 1) == 0 || max_level < 0 || max_level > LK_MAXL || max_iters < 1 || !(eps >= 0)) return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    myslam_lk* h = new myslam_lk();
    h->win = win; h->max_level = max_level; h->max_iters = max_iters; h->eps = eps; h->min_eig = min_eig_threshold;
    *out = h;
    return MYSLAM_OK;
}

int myslam_lk_destroy(myslam_lk* h) {
    if (!h) return MYSLAM_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->d_pyrP, h->d_pyrN, h->d_img, h->d_pts, h->d_st, h->slot[0].img, h->slot[0].pyr, h->slot[1].img, h->slot[1].pyr, h->d_stage};
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete h;
    return MYSLAM_OK;
}

int myslam_lk_set_stream(myslam_lk* h, void* s) {
    if (!h) return MYSLAM_ERR_INVALID;
    h->stream = (hipStream_t)s;
    return MYSLAM_OK;
}

int myslam_lk_track_batch(myslam_lk* h, const uint8_t* d_prev, const uint8_t* d_next, int batch, int rows, int cols, int step, size_t stride,
                          const float* d_prev_pts, float* d_next_pts, const int32_t* d_counts, int cap, uint8_t* d_status, float* d_err) {
    if (!h || !d_prev || !d_next || batch < 1 || rows < 1 || cols < 1 || step < cols || !d_prev_pts || !d_next_pts || !d_counts || cap < 1 || !d_status)
        return MYSLAM_ERR_INVALID;
    return lk_run(h, d_prev, d_next, batch, rows, cols, step, step, stride, stride, d_prev_pts, d_next_pts, d_counts, 0, cap, d_status, d_err);
}

int myslam_lk_track(myslam_lk* h, const uint8_t* prev, const uint8_t* next, int rows, int cols, int prev_step, int next_step,
                    const float* prev_pts, float* next_pts, int n, uint8_t* status, float* err) {
    if (!h || n < 0 || (n > 0 && (!prev_pts || !next_pts || !status))) return MYSLAM_ERR_INVALID;
    if (n == 0) return MYSLAM_OK;
    if (!prev || !next || rows < 1 || cols < 1 || prev_step < cols || next_step < cols) return MYSLAM_ERR_INVALID;
    // both images travel as ONE contiguous copy each with their own row pitch (a 2-D copy from pageable memory goes row by row: 5 ms
    // for a 1241 x 376 pair against 0.1 ms), the kernels read them with that pitch
    const size_t pb = ((size_t)rows * prev_step + 255) & ~(size_t)255, nb = (size_t)rows * next_step;
    if (pb + nb > h->imgBytes) {
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->d_img) (void)hipFree(h->d_img);
        h->d_img = nullptr; h->imgBytes = 0;
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_img, pb + nb)); h->imgBytes = pb + nb;
    }
    if (n > h->ptsCap) {
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->d_pts) (void)hipFree(h->d_pts);
        if (h->d_st) (void)hipFree(h->d_st);
        h->d_pts = nullptr; h->d_st = nullptr; h->ptsCap = 0;
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_pts, sizeof(float) * 5 * (size_t)n)); MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_st, (size_t)n));
        h->ptsCap = n;
    }
    hipStream_t s = h->stream;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_img, prev, (size_t)rows * prev_step - (size_t)(prev_step - cols), hipMemcpyHostToDevice, s));      // the last row ends at its last pixel
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_img + pb, next, (size_t)rows * next_step - (size_t)(next_step - cols), hipMemcpyHostToDevice, s));
    float* d_pp = h->d_pts; float* d_np = d_pp + 2 * (size_t)h->ptsCap; float* d_err = d_np + 2 * (size_t)h->ptsCap;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(d_pp, prev_pts, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(d_np, next_pts, sizeof(float) * 2 * n, hipMemcpyHostToDevice, s));
    int rc = lk_run(h, h->d_img, h->d_img + pb, 1, rows, cols, prev_step, next_step, pb, nb, d_pp, d_np, nullptr, n, n, h->d_st, d_err);
    if (rc) return rc;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(next_pts, d_np, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, s));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(status, h->d_st, n, hipMemcpyDeviceToHost, s));
    if (err) MYSLAM_HIP_CHECK(hipMemcpyAsync(err, d_err, sizeof(float) * n, hipMemcpyDeviceToHost, s));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return MYSLAM_OK;
}

int myslam_lk_prefetch(myslam_lk* h, const uint8_t* img, uint64_t token, int rows, int cols, int step) {
    if (!h || !img || !token || rows < 1 || cols < 1 || step < cols) return MYSLAM_ERR_INVALID;
    int rc = lk_plan_cached(h, rows, cols);
    if (rc) return rc;
    if (lk_find_slot(h, token, rows, cols, step) >= 0) return MYSLAM_OK;                  // already there
    return lk_fill_slot(h, 1 - h->lastNext, img, token, rows, cols, step);                // never the slot the next call tracks FROM
}

int myslam_lk_track_cached(myslam_lk* h, const uint8_t* prev, uint64_t prev_token, const uint8_t* next, uint64_t next_token, int rows, int cols,
                           int prev_step, int next_step, const float* prev_pts, float* next_pts, int n, uint8_t* status, float* err) {
    if (!h || n < 0 || (n > 0 && (!prev_pts || !next_pts || !status))) return MYSLAM_ERR_INVALID;
    if (n == 0) return MYSLAM_OK;
    if (!prev || !next || rows < 1 || cols < 1 || prev_step < cols || next_step < cols) return MYSLAM_ERR_INVALID;
    int rc = lk_plan_cached(h, rows, cols);
    if (rc) return rc;
    int sp = lk_find_slot(h, prev_token, rows, cols, prev_step), sn = lk_find_slot(h, next_token, rows, cols, next_step);
    if (sp >= 0 && sp == sn) sn = -1;                                                     // one token for both images: the second is uploaded
    if (sp < 0) { sp = sn < 0 ? 0 : 1 - sn; if ((rc = lk_fill_slot(h, sp, prev, prev_token, rows, cols, prev_step))) return rc; }
    if (sn < 0) { sn = 1 - sp; if ((rc = lk_fill_slot(h, sn, next, next_token, rows, cols, next_step))) return rc; }
    if (n > h->stageCap) {
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->d_stage) (void)hipFree(h->d_stage);
        if (h->h_pin) (void)hipHostFree(h->h_pin);
        h->d_stage = nullptr; h->h_pin = nullptr; h->stageCap = 0;
        const int cap = std::max(n, 512);
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_stage, (size_t)21 * cap + 16)); MYSLAM_HIP_CHECK(hipHostMalloc((void**)&h->h_pin, (size_t)21 * cap + 16));
        h->stageCap = cap;
    }
    hipStream_t s = h->stream;
    // device block: [prev_pts 8n][next_pts 8n][err 4n][status n]; upload = the first 16n bytes, download = the last 13n
    float* d_pp = reinterpret_cast<float*>(h->d_stage); float* d_np = d_pp + 2 * (size_t)n; float* d_err = d_np + 2 * (size_t)n;
    uint8_t* d_st = reinterpret_cast<uint8_t*>(d_err + n);
    memcpy(h->h_pin, prev_pts, sizeof(float) * 2 * n); memcpy(h->h_pin + sizeof(float) * 2 * n, next_pts, sizeof(float) * 2 * n);
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stage, h->h_pin, sizeof(float) * 4 * n, hipMemcpyHostToDevice, s));
    const myslam_lk::Slot &P = h->slot[sp], &N = h->slot[sn];
    if ((rc = lk_track_launch(h, P.img, N.img, P.pyr, N.pyr, 1, rows, cols, prev_step, next_step, P.imgBytes, N.imgBytes, d_pp, d_np, nullptr, n, n, d_st, d_err)))
        return rc;
    h->lastNext = sn;
    uint8_t* hp = h->h_pin + sizeof(float) * 2 * n;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(hp, d_np, (size_t)13 * n, hipMemcpyDeviceToHost, s));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(next_pts, hp, sizeof(float) * 2 * n);
    if (err) memcpy(err, hp + sizeof(float) * 2 * n, sizeof(float) * n);
    memcpy(status, hp + sizeof(float) * 3 * n, n);
    return MYSLAM_OK;
}

}  // extern "C"
