// calc.hip — DeepLCD / CALC descriptor network on gfx950 (replaces class DeepLCD, reference
// include/myslam/deeplcd.h:21-48 and src/deeplcd.cpp:10-91; architecture SURVEY.md Appendix A.6).
//
//   u8 image --blur 7x7 (sigma<=0 table, optionally in place: reference quirk)--> resize 160x120 --> /255
//   conv1 64@5x5 s2 p4 + ReLU -> maxpool 3x3 s2 (ceil) -> LRN(5,1e-4,.75)        [VALU, lane = channel]
//   conv2 128@4x4 s1 p2 + ReLU                                                   [bf16 matrix cores, f32 accuracy]
//   maxpool -> LRN -> conv3 4@3x3 + ReLU -> flatten (Caffe NCHW order) -> L2 normalise
//
// The layer list is DATA (myslam_calc_layer records: from deploy.prototxt through host/myslam_caffe.hpp, from the CALCW2 model file, or
// the SURVEY A.6 default).  A list with the geometry above runs on the fused kernels, which take the LRN constants and the ReLU flags
// as arguments; any other list of Convolution / ReLU / max-Pooling / LRN layers runs on the generic layer-by-layer kernels at the
// end of this file (so a differing pad or LRN window in the real prototxt needs no new kernel); everything else is refused with
// MYSLAM_ERR_UNSUPPORTED.
// Activations are NHWC (channel-last) in HBM so that a wave's 64 lanes map to 64 channels (coalesced
// 256-byte rows) and conv2's im2col rows are contiguous 64-byte channel runs.  All arithmetic is f32-accurate
// (the reference runs Caffe in f32).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "orb_plan.h"
#include "../host/myslam_caffe.hpp"

namespace myslam_hip {

void launch_blur(const BlurArgs& a, int batch, hipStream_t s);
void gauss_q8(int kind, int q[7]);

constexpr int IN_H = 120, IN_W = 160;
// the network input lives in a zero-padded plane (conv1 pad 4 + what the clipped pool windows still touch): no bounds tests
constexpr int IN_PAD = 4, IN_PH = IN_H + IN_PAD + 5, IN_PW = 176, IN_PLANE = IN_PH * IN_PW;
constexpr int C1 = 64, H1 = 62, W1 = 82, HP1 = 31, WP1 = 41;
constexpr int C2 = 128, H2 = 32, W2 = 42, HP2 = 16, WP2 = 21;
constexpr int C3 = 4, H3 = 14, W3 = 19;
constexpr int K2 = 64 * 16;                 // conv2 reduction length
constexpr int M2 = H2 * W2;                 // conv2 output pixels per image (1344)
constexpr size_t NWEIGHTS = 64 * 25 + 64 + 128 * 64 * 16 + 128 + 4 * 128 * 9 + 4;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- input: bilinear 8U resize (same fixed-point arithmetic as the pyramid) + u8 -> f32 * (1/255) ----
__global__ __launch_bounds__(256) void k_lcd_input(const uint8_t* __restrict__ src, int sw, int sh, int spitch, size_t sstride,
                                                   const int32_t* __restrict__ xofs, const int16_t* __restrict__ xa,
                                                   const int32_t* __restrict__ yofs, const int16_t* __restrict__ yb,
                                                   float* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= IN_H * IN_W) return;
    const int dy = i / IN_W, dx = i - dy * IN_W;
    const int sy = yofs[dy];
    const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
    const uint8_t* S0 = src + (size_t)b * sstride + (size_t)sy0 * spitch;
    const uint8_t* S1 = src + (size_t)b * sstride + (size_t)sy1 * spitch;
    const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1);
    const int a0 = xa[2 * dx], a1 = xa[2 * dx + 1], b0 = yb[2 * dy], b1 = yb[2 * dy + 1];
    const int r0 = S0[sx] * a0 + S0[sx1] * a1, r1 = S1[sx] * a0 + S1[sx1] * a1;
    int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    out[(size_t)b * IN_PLANE + (dy + IN_PAD) * IN_PW + dx + IN_PAD] = (float)v * (float)(1.0 / 255.0);      // deeplcd.cpp:64 convertTo(CV_32F, 1/255.)
}

// ---- the same input WITHOUT materialising the blurred frame (blur_in_place = 0): the 160 x 120 resize reads four blurred pixels per
// output, 77 k of the 467 k the full Gaussian would produce.  One thread per output pixel evaluates exactly those four 7 x 7
// responses from an 8 x 8 source window — the blur is exact integer arithmetic ((sum_r q_r sum_c q_c p + 32768) >> 16, no
// intermediate rounding), so the results equal the two-pass kernel's bit for bit.  REFLECT_101 rows by index; a window that leaves
// the image sideways takes the byte-wise path.
__device__ __forceinline__ int lcd_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}
struct LcdTaps { int q[7]; };
__global__ __launch_bounds__(256) void k_lcd_input_fused(const uint8_t* __restrict__ src, int sw, int sh, int spitch, size_t sstride,
                                                         const int32_t* __restrict__ xofs, const int16_t* __restrict__ xa,
                                                         const int32_t* __restrict__ yofs, const int16_t* __restrict__ yb, LcdTaps tp,
                                                         float* __restrict__ out) {
    MYSLAM_SIDE_PRIO();
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= IN_H * IN_W) return;
    const int dy = i / IN_W, dx = i - dy * IN_W;
    const int sy = yofs[dy];
    const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
    const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1);
    const uint8_t* img = src + (size_t)b * sstride;
    const uint32_t qa = (uint32_t)tp.q[0] | ((uint32_t)tp.q[1] << 8) | ((uint32_t)tp.q[2] << 16) | ((uint32_t)tp.q[3] << 24);
    const uint32_t qb = (uint32_t)tp.q[4] | ((uint32_t)tp.q[5] << 8) | ((uint32_t)tp.q[6] << 16);
    // horizontal 7-tap sums at columns sx and sx1 for the rows sy0-3 .. sy0+4 (sy1 = sy0 + 1 except at the clamped bottom row)
    uint32_t h0[8], h1[8];
    if (sx >= 3 && sx + 4 < sw && sy0 >= 3 && sy0 + 4 < sh) {      // the whole 8 x 8 window lies inside the image: straight-line code
        const uint8_t* row = img + (size_t)(sy0 - 3) * spitch + (sx - 3);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t lo, hi;
            __builtin_memcpy(&lo, row + (size_t)j * spitch, 4); __builtin_memcpy(&hi, row + (size_t)j * spitch + 4, 4);     // bytes sx-3..sx, sx+1..sx+4
            h0[j] = __builtin_amdgcn_udot4(lo, qa, __builtin_amdgcn_udot4(hi, qb, 0u, false), false);
            const uint32_t lo1 = __builtin_amdgcn_alignbyte(hi, lo, 1u), hi1 = hi >> 8;             // the same window one pixel to the right
            h1[j] = __builtin_amdgcn_udot4(lo1, qa, __builtin_amdgcn_udot4(hi1, qb, 0u, false), false);
        }
    } else {                                                       // border outputs: byte-wise with REFLECT_101 indices
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            const uint8_t* row = img + (size_t)lcd_reflect101(sy0 - 3 + j, sh) * spitch;
            uint32_t a = 0, c = 0;
#pragma unroll 1
            for (int k = 0; k < 7; k++) {
                a += (uint32_t)tp.q[k] * row[lcd_reflect101(sx - 3 + k, sw)];
                c += (uint32_t)tp.q[k] * row[lcd_reflect101(sx1 - 3 + k, sw)];
            }
            // (dynamic index into a register array: kept out of the fast path)
#pragma unroll
            for (int jj = 0; jj < 8; jj++) if (jj == j) { h0[jj] = a; h1[jj] = c; }
        }
    }
    // vertical taps: the blurred row sy0 uses window rows 0..6, the blurred row sy1 rows (sy1 - sy0) .. (sy1 - sy0) + 6
    const int o = sy1 - sy0;                                  // 0 or 1
    uint32_t b00 = 32768u, b01 = 32768u, b10 = 32768u, b11 = 32768u;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        b00 += (uint32_t)tp.q[k] * h0[k]; b01 += (uint32_t)tp.q[k] * h1[k];
        const uint32_t u0 = o ? h0[k + 1] : h0[k], u1 = o ? h1[k + 1] : h1[k];
        b10 += (uint32_t)tp.q[k] * u0; b11 += (uint32_t)tp.q[k] * u1;
    }
    const int p00 = (int)(b00 >> 16), p01 = (int)(b01 >> 16), p10 = (int)(b10 >> 16), p11 = (int)(b11 >> 16);
    const int a0 = xa[2 * dx], a1 = xa[2 * dx + 1], w0 = yb[2 * dy], w1 = yb[2 * dy + 1];
    const int r0 = p00 * a0 + p01 * a1, r1 = p10 * a0 + p11 * a1;
    int v = (((w0 * (r0 >> 4)) >> 16) + ((w1 * (r1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    out[(size_t)b * IN_PLANE + (dy + IN_PAD) * IN_PW + dx + IN_PAD] = (float)v * (float)(1.0 / 255.0);
}

// LRN denominator scale^-0.75 = rsqrt(scale) * sqrt(rsqrt(scale)) on the hardware rsq / sqrt units (scale >= 1; within 3 ulp of
// powf, two orders of magnitude inside the descriptor tolerance) instead of the ~150-instruction powf expansion
__device__ __forceinline__ float lrn_pow_m075(float scale) {
    const float r = __builtin_amdgcn_rsqf(scale);
    return r * __builtin_amdgcn_sqrtf(r);
}
// LRN constants of one layer (lrn_param of the prototxt): y = x * (k + alpha / n * sum x^2)^-beta; aon = alpha / n evaluated in f32 on
// the host exactly as Caffe does; fast = beta is 0.75 (the hardware rsq / sqrt form above), otherwise powf
struct LrnP { float aon, beta, k; int fast; };
__device__ __forceinline__ float lrn_factor(float ss, const LrnP& p) {
    const float scale = p.k + p.aon * ss;
    return p.fast ? lrn_pow_m075(scale) : powf(scale, -p.beta);
}

// ---- max-pool 3x3 s2 (Caffe ceil mode, clipped windows) + LRN(5) across the 128 channels of the conv2 map, 2 x 2 pooled pixels per wave ----
// Lane l owns channels (2l, 2l+1): one 8-byte load per input pixel and lane (a coalesced 512-byte row per wave), all 25 loads of the
// 5 x 5 input block issued before the first use (coordinates clamped instead of clipped: a duplicate does not change a maximum),
// every input pixel read 1.56 instead of 2.25 times, the LRN neighbours over lane shuffles instead of LDS; LRN summation order
// channels c-2 .. c+2.
__global__ __launch_bounds__(256) void k_pool_lrn128_2x2(const float* __restrict__ in, int H, int W, int OH, int OW, LrnP lp, float* __restrict__ out) {
    MYSLAM_SIDE_PRIO();
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int TW = (OW + 1) >> 1, TH = (OH + 1) >> 1;
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (tile >= TW * TH) return;
    const int ty = tile / TW, tx = tile - ty * TW;
    const int oy = 2 * ty, ox = 2 * tx, y0 = 2 * oy, x0 = 2 * ox;
    const float2* I = reinterpret_cast<const float2*>(in + (size_t)b * H * W * 128) + lane;
    float2 v[5][5];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
        for (int c = 0; c < 5; c++) v[r][c] = I[((size_t)min(y0 + r, H - 1) * W + min(x0 + c, W - 1)) * 64];
    auto pmax = [&](int r0, int c0) {
        float2 m = v[r0][c0];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) { m.x = fmaxf(m.x, v[r0 + r][c0 + c].x); m.y = fmaxf(m.y, v[r0 + r][c0 + c].y); }
        return m;
    };
    auto lrn_store = [&](float2 m, int py, int px) {
        float2 pv, nx;
        pv.x = __shfl_up(m.x, 1, 64); pv.y = __shfl_up(m.y, 1, 64); nx.x = __shfl_down(m.x, 1, 64); nx.y = __shfl_down(m.y, 1, 64);
        if (lane == 0) pv = make_float2(0.f, 0.f);
        if (lane == 63) nx = make_float2(0.f, 0.f);
        float sa = 0.f, sb = 0.f;       // channel 2l: c-2 .. c+2 = pv.x pv.y m.x m.y nx.x; channel 2l+1: pv.y m.x m.y nx.x nx.y
        sa += pv.x * pv.x; sa += pv.y * pv.y; sa += m.x * m.x; sa += m.y * m.y; sa += nx.x * nx.x;
        sb += pv.y * pv.y; sb += m.x * m.x; sb += m.y * m.y; sb += nx.x * nx.x; sb += nx.y * nx.y;
        float2 o;
        o.x = m.x * lrn_factor(sa, lp);
        o.y = m.y * lrn_factor(sb, lp);
        reinterpret_cast<float2*>(out + ((size_t)b * OH * OW + (size_t)py * OW + px) * 128)[lane] = o;
    };
    const bool row1 = oy + 1 < OH, col1 = ox + 1 < OW;           // wave-uniform
    lrn_store(pmax(0, 0), oy, ox);
    if (col1) lrn_store(pmax(0, 2), oy, ox + 1);
    if (row1) lrn_store(pmax(2, 0), oy + 1, ox);
    if (row1 && col1) lrn_store(pmax(2, 2), oy + 1, ox + 1);
}

// ---- conv1 + ReLU + max-pool + LRN fused, 2 x 2 POOLED pixels per wave, lane = channel ----
// The input window of a pool window is wave-uniform: it is fetched with scalar loads and fed to v_fmac as SGPR operands, so the conv1
// activation map (1.3 MB per image) never exists in HBM.  Neighbouring 3x3 / stride-2 pool windows share a row and a column of conv1
// outputs: one wave per pooled pixel would compute every conv1 output 2.25 times.  Here a wave owns a 2 x 2 block of pooled pixels = 5 x 5 conv1 outputs (1.56 per pooled pixel
// instead of 2.25: -31 % FMAs), walks the conv rows top to bottom (5 input rows x 13 columns of wave-uniform scalars per conv
// row) and folds each output into the maxima of the pooled pixels it belongs to.  Per output: tap order (ky, kx), bias, ReLU,
// Caffe's clipped ceil-mode windows, LRN over channels c-2 .. c+2 (zero padded) across lane shuffles.
constexpr int HT1 = (HP1 + 1) / 2, WT1 = (WP1 + 1) / 2;           // 2 x 2 tiles of the pooled map
__global__ __launch_bounds__(256) void k_conv1_pool_lrn2(const float* __restrict__ in, const float* __restrict__ w1t /*[25][64]*/,
                                                         const float* __restrict__ b1, int relu, LrnP lp, float* __restrict__ out /*[HP1*WP1][64]*/) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (tile >= HT1 * WT1) return;
    float w[25];
#pragma unroll
    for (int k = 0; k < 25; k++) w[k] = w1t[k * 64 + lane];
    const float bias = b1[lane];
    const int ty = tile / WT1, tx = tile - ty * WT1;
    const int oy = 2 * ty, ox = 2 * tx;                            // first pooled pixel of the tile
    const bool row1 = oy + 1 < HP1, col1 = ox + 1 < WP1;           // wave-uniform: the tile's second pooled row / column exists
    const float* I = in + (size_t)b * IN_PLANE + (4 * oy) * IN_PW + 4 * ox;      // window origin of pooled pixel (oy, ox)
    float m00 = -INFINITY, m01 = -INFINITY, m10 = -INFINITY, m11 = -INFINITY;
    // the 5 x 13 input window of a conv row lives in scalar registers; consecutive conv rows share three of its five rows, so it is
    // kept rolling: two new rows per step (loading all five afresh overflows the scalar file and spills through v_writelane)
    float win[5][13];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 13; c++) win[r + 2][c] = I[r * IN_PW + c];
#pragma unroll
    for (int cy = 0; cy < 5; cy++) {                                // conv row 2 oy + cy
        if (cy >= 3 && !row1) break;                                // rows 3, 4 only feed the second pooled row (and would read past the padded plane)
#pragma unroll
        for (int c = 0; c < 13; c++) { win[0][c] = win[2][c]; win[1][c] = win[3][c]; win[2][c] = win[4][c]; }
#pragma unroll
        for (int r = 3; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 13; c++) win[r][c] = I[(2 * cy + r) * IN_PW + c];
#pragma unroll
        for (int cx = 0; cx < 5; cx++) {                            // conv column 2 ox + cx
            // taps (0, 1) and (2, 3) of a kernel row as packed f32 multiply-adds (v_pk_fma_f32: two per lane and instruction; the window
            // pair is a wave-uniform register pair), tap 4 alone: 15 instead of 25 instructions per output
            typedef float c1_f2 __attribute__((ext_vector_type(2)));
            c1_f2 acc2 = {0.f, 0.f};
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 5; ky++) {
#pragma unroll
                for (int kp = 0; kp < 2; kp++) {
                    const c1_f2 wv = {w[ky * 5 + 2 * kp], w[ky * 5 + 2 * kp + 1]}, xv = {win[ky][2 * cx + 2 * kp], win[ky][2 * cx + 2 * kp + 1]};
                    acc2 = __builtin_elementwise_fma(wv, xv, acc2);
                }
                acc += w[ky * 5 + 4] * win[ky][2 * cx + 4];
            }
            acc += acc2.x + acc2.y;
            const bool inside = 2 * oy + cy < H1 && 2 * ox + cx < W1;      // Caffe ceil-mode pooling: clipped windows
            const float pre = acc + bias;
            const float v = inside ? (relu ? fmaxf(pre, 0.f) : pre) : -INFINITY;
            if (cy <= 2 && cx <= 2) m00 = fmaxf(m00, v);
            if (cy <= 2 && cx >= 2) m01 = fmaxf(m01, v);
            if (cy >= 2 && cx <= 2) m10 = fmaxf(m10, v);
            if (cy >= 2 && cx >= 2) m11 = fmaxf(m11, v);
        }
    }
    auto lrn_store = [&](float m, int py, int px) {              // LRN(5) across the 64 channels (zero padded)
        const float um1 = __shfl_up(m, 1, 64), um2 = __shfl_up(m, 2, 64), dp1 = __shfl_down(m, 1, 64), dp2 = __shfl_down(m, 2, 64);
        const float v0 = lane >= 2 ? um2 : 0.f, v1 = lane >= 1 ? um1 : 0.f, v3 = lane <= 62 ? dp1 : 0.f, v4 = lane <= 61 ? dp2 : 0.f;
        float ss = 0.f;
        ss += v0 * v0; ss += v1 * v1; ss += m * m; ss += v3 * v3; ss += v4 * v4;
        out[((size_t)b * HP1 * WP1 + py * WP1 + px) * 64 + lane] = m * lrn_factor(ss, lp);
    };
    lrn_store(m00, oy, ox);
    if (col1) lrn_store(m01, oy, ox + 1);
    if (row1) lrn_store(m10, oy + 1, ox);
    if (row1 && col1) lrn_store(m11, oy + 1, ox + 1);
}

// f32 wave sum on the DPP network; the total lands in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f32(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63_f32(float v) {
    v = dpp_add_f32<0xB1, 0xf>(v); v = dpp_add_f32<0x4E, 0xf>(v); v = dpp_add_f32<0x141, 0xf>(v); v = dpp_add_f32<0x140, 0xf>(v);
    v = dpp_add_f32<0x142, 0xa>(v); v = dpp_add_f32<0x143, 0xc>(v);
    return v;
}

constexpr int CV_BN = 128;

// ---- conv2 + ReLU as an implicit GEMM on the bf16 matrix cores with fp32 accuracy ----
// C[M = batch*1344][N = 128] = A[M][K = 1024] * Wt[K][N];  k = (ky*4+kx)*64 + ic  (channel runs contiguous in NHWC).
// An fp32-input MFMA (v_mfma_f32_32x32x2_f32) runs at the f32 VECTOR rate — in practice it competes with the VALU-bound ORB kernels of
// the other stream instead of running under them (measured: 1.48 against 0.93 ms per 512 frames, no gain from co-residency,
// DESIGN.md section 4).  The bf16 matrix pipe is 16x faster and separate.  Every f32 operand is split exactly into three bf16 pieces a = h + m + l (8 + 8 + 8 significand bits,
// round-to-nearest conversions, exact residuals); of the nine partial products the six largest are kept:
//     a b  ~  hh + hm + mh + hl + lh + mm        (dropped: ml, lm, ll <= 2^-23 |a b|, the rounding level of an f32 product)
// accumulated in f32 by v_mfma_f32_32x32x16_bf16.  Weights are split once on the host ([stage][piece][n][16 k], so a stage's slab
// is one contiguous 12 KB block); activations are split while they are staged into LDS (11 VALU per pair of elements).
// Block tile 128 x 128, 4 waves x (2 x 2) tiles, BK = 16 = one MFMA K: 24 MFMAs of 32 cycles per wave and stage against
// 32 x 64 cycles for the f32 form.
typedef __bf16 cv_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void cv_split3(float a0, float a1, uint32_t& h, uint32_t& m, uint32_t& l) {    // two elements per dword, a0 low
    const __bf16 h0 = (__bf16)a0, h1 = (__bf16)a1;
    const float r0 = a0 - (float)h0, r1 = a1 - (float)h1;
    const __bf16 m0 = (__bf16)r0, m1 = (__bf16)r1;
    const float q0 = r0 - (float)m0, q1 = r1 - (float)m1;
    const __bf16 l0 = (__bf16)q0, l1 = (__bf16)q1;
    h = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
    m = (uint32_t)__builtin_bit_cast(unsigned short, m0) | ((uint32_t)__builtin_bit_cast(unsigned short, m1) << 16);
    l = (uint32_t)__builtin_bit_cast(unsigned short, l0) | ((uint32_t)__builtin_bit_cast(unsigned short, l1) << 16);
}

__global__ __launch_bounds__(256) void k_conv2_bf16x6(const float* __restrict__ in /*[B][31*41][64]*/,
                                                      const uint4* __restrict__ wt3 /*[64 stages][3][128 n][2 k-halves] x 8 bf16*/,
                                                      const float* __restrict__ b2, float* __restrict__ out /*[B*1344][128]*/, int Mtotal, int relu) {
    constexpr int BM = 128, BN = 128;
    // [piece][k half][row]: 8 bf16 per uint4.  K-half major, the second half shifted by 128 bytes: the 16 lanes a ds_read_b128 / ds_write_b128
    // serves per cycle then touch 64 distinct banks (row-major [row][k half] put lanes i and i + 8 on the same banks); the waves' LDS wait
    // fell by a third (PMC SQ_WAIT_INST_LDS 19.9 M -> 13.2 M per 64 frames), the kernel by 1-2 %
    constexpr int KH = BM + 8;
    __shared__ uint4 s_a[2][3][2 * KH];
    __shared__ uint4 s_b[2][3][2 * KH];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = blockIdx.x * BM;
    const int am = t >> 1, akh = t & 1;                // staging role: row m, k half (8 consecutive k)
    const int gm = m0 + am;
    const bool mvalid = gm < Mtotal;
    const int img = mvalid ? gm / M2 : 0;
    const int pix = mvalid ? gm - img * M2 : 0;
    const int oy = pix / W2, ox = pix - oy * W2;
    const float* inb = in + (size_t)img * HP1 * WP1 * 64;

    // the prefetched slab of the next stage lives in registers across the MFMA block: unconditional loads from a clamped address
    // plus a select at store time (a branch here sends the values through scratch and waits for the loads on the spot)
    float4 ra0 = make_float4(0, 0, 0, 0), ra1 = ra0;
    uint4 rb0 = make_uint4(0, 0, 0, 0), rb1 = rb0, rb2 = rb0;
    bool rav = false;
#define CV2_LOAD_STAGE(S)                                                                                                     \
    {                                                                                                                         \
        const int tap_ = (S) >> 2, ic0_ = ((S) & 3) * 16;                                                                     \
        const int iy_ = oy + (tap_ >> 2) - 2, ix_ = ox + (tap_ & 3) - 2;                                                      \
        rav = mvalid && iy_ >= 0 && iy_ < HP1 && ix_ >= 0 && ix_ < WP1;                                                       \
        const float4* src_ = reinterpret_cast<const float4*>(inb + (rav ? ((size_t)iy_ * WP1 + ix_) * 64 : 0) + ic0_ + akh * 8); \
        ra0 = src_[0]; ra1 = src_[1];                                                                                         \
        const uint4* wsrc_ = wt3 + (size_t)(S) * (3 * BN * 2) + t;                                                            \
        rb0 = wsrc_[0]; rb1 = wsrc_[BN * 2]; rb2 = wsrc_[2 * BN * 2];                                                         \
    }
    auto store_stage = [&](int buf) {
        const float z = rav ? 1.f : 0.f;
        uint4 h, m, l;
        cv_split3(ra0.x * z, ra0.y * z, h.x, m.x, l.x); cv_split3(ra0.z * z, ra0.w * z, h.y, m.y, l.y);
        cv_split3(ra1.x * z, ra1.y * z, h.z, m.z, l.z); cv_split3(ra1.z * z, ra1.w * z, h.w, m.w, l.w);
        const int si = akh * KH + am;                                     // (row, k half) of this thread's slab piece
        s_a[buf][0][si] = h; s_a[buf][1][si] = m; s_a[buf][2][si] = l;
        s_b[buf][0][si] = rb0; s_b[buf][1][si] = rb1; s_b[buf][2][si] = rb2;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    CV2_LOAD_STAGE(0)
    store_stage(0);
    __syncthreads();
    constexpr int NSTAGE = K2 / 16;
    const int lr = lane & 31, lk = lane >> 5;
    for (int s = 0; s < NSTAGE; s++) {
        const int buf = s & 1;
        if (s + 1 < NSTAGE) CV2_LOAD_STAGE(s + 1)
        cv_bf16x8 A[2][3], Bm[2][3];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int p = 0; p < 3; p++) A[i][p] = __builtin_bit_cast(cv_bf16x8, s_a[buf][p][lk * KH + wm + 32 * i + lr]);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) Bm[j][p] = __builtin_bit_cast(cv_bf16x8, s_b[buf][p][lk * KH + wn + 32 * j + lr]);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][2], Bm[j][0], c, 0, 0, 0);       // l h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], Bm[j][2], c, 0, 0, 0);       // h l
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], Bm[j][1], c, 0, 0, 0);       // m m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], Bm[j][0], c, 0, 0, 0);       // m h
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], Bm[j][1], c, 0, 0, 0);       // h m
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], Bm[j][0], c, 0, 0, 0);       // h h
                acc[i][j] = c;
            }
        if (s + 1 < NSTAGE) store_stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int n = wn + j * 32 + lr;
        const float bias = b2[n];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float v = acc[i][j][r] + bias;
                if (m < Mtotal) out[(size_t)m * CV_BN + n] = relu ? fmaxf(v, 0.f) : v;
            }
    }
}

// ---- the same implicit GEMM on the f16 matrix cores, three products instead of six (round 3) ----
// An f32 operand is split into two f16 pieces with 22 significant bits between them:  a = h + m' 2^-11,  h = f16(a),  m' = f16((a - h) 2^11)
// (a - h is exact in f32, the power-of-two scaling keeps the residual out of f16's subnormal range).  Then
//     2^11 a b  ~  h_a (2^11 h_b)  +  h_a m'_b  +  m'_a h_b            (dropped: m_a m_b and the two rounding terms, <= 2^-21 |a b| together)
// so that ONE accumulator takes all three products: the weight side carries a third plane Hs = 2^11 h_b (exact in f16 while |w| < 2^5: checked
// when the model is loaded, as is the bound 65504 on the activations — models outside keep the bf16 x 6 kernel).  Half the matrix
// instructions of the bf16 form (12 per wave and K step) and 5 instead of 6 operand planes through LDS; measured against an f64 reference
// the error is of the same order (tools/conv2_error.py).  Same tiling, staging roles and LDS layout as k_conv2_bf16x6.
typedef _Float16 cv_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void cv_split2_f16(float a0, float a1, uint32_t& h, uint32_t& m) {    // two elements per dword, a0 low
    const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
    const _Float16 m0 = (_Float16)((a0 - (float)h0) * 2048.f), m1 = (_Float16)((a1 - (float)h1) * 2048.f);
    h = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
    m = (uint32_t)__builtin_bit_cast(unsigned short, m0) | ((uint32_t)__builtin_bit_cast(unsigned short, m1) << 16);
}

// Round 4: the kernel no longer stages anything per K step.
//   * A block owns THREE OUTPUT ROWS of one image (126 pixels = one 128-row M tile, 11 blocks per image).  Its input is a 6 x 45-pixel patch of
//     the conv1 map, which arrives already split (conv1's epilogue writes a = h + m' 2^-11 as f16 pairs).  The patch of ONE channel group (16
//     channels: 64 bytes per pixel, [h k-half 0][h k-half 1][m' 0][m' 1], padded to 80 bytes so that the 16-byte operand reads of 16 lanes spread
//     over all banks) is 21.6 KB of LDS; the 16 taps of that group are im2col'ed by ADDRESS: a lane's A operand of tap (ky, kx) is the same
//     LDS read at a constant offset.  Four patches per block (the next one is fetched into registers under the current one's 192 MFMAs).
//   * A wave owns 32 of the 128 output channels for all 128 rows: its B operands (three f16 planes per K step) come straight from the
//     L2-resident weight slab in operand layout (one coalesced 1 KB read per plane), two K steps ahead; nothing of B passes through LDS.
//   * 8 block barriers in total (the staging form had 128), 21.6 KB of LDS (was 43.5: it now fits beside six FAST blocks), every activation
//     fetched once per block and channel group (was once per tap: 16 x), no split arithmetic.
// K order: channel group outer, tap inner (the staging form ran tap-major): the f32 accumulation order differs in the last bits, the
// error against f64 is unchanged (tools/conv2_error.py).
constexpr int C2R = 3, C2PW = WP1 + 4, C2PH = C2R + 3, C2PIX = C2PH * C2PW, C2PS = 5;        // patch 6 x 45 pixels, 5 uint4 (80 bytes) per pixel
constexpr int C2NB = (H2 + C2R - 1) / C2R;                                                   // blocks per image
static_assert(C2R * W2 <= 128, "three output rows are one 128-row M tile");

__global__ __launch_bounds__(256) void k_conv2_f16x3(const float* __restrict__ in /*[B][31*41] x 256 bytes: per group of 16 channels 16 f16 h, then 16 f16 m'*/,
                                                     const uint4* __restrict__ wt3 /*[64 stages = tap * 4 + group][3: Hs, m', h][128 n][2 k-halves] x 8 f16*/,
                                                     const float* __restrict__ b2, float* __restrict__ out /*[B*1344][128]*/, int batch, int relu) {
    MYSLAM_SIDE_PRIO();
    __shared__ uint4 s_patch[C2PIX * C2PS];
    // (a limited grid — blocks walking several work items, as k_describe2 does — was measured for this kernel too, round 4: 1 .. 4 blocks per CU
    // gave 7.03-7.11 ms per step against 7.03-7.08 unlimited: nothing, not built in)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lr = lane & 31, lk = lane >> 5;
    const int img = blockIdx.y, row0 = C2R * blockIdx.x;
    const int nvalid = min(C2R, H2 - row0) * W2;                  // output pixels of this block
    const uint4* inq = reinterpret_cast<const uint4*>(in + (size_t)img * HP1 * WP1 * 64);        // 16 uint4 per pixel, 4 per channel group
    // patch staging roles: item = (patch pixel, 16-byte piece); 1080 items, up to 5 per thread
    constexpr int NIT = (C2PIX * 4 + 255) / 256;
    int psrc[NIT], pdst[NIT];                                      // source uint4 index inside the image (-1: zero padding), LDS index (-1: no item)
#pragma unroll
    for (int u = 0; u < NIT; u++) {
        const int idx = t + 256 * u, pix = idx >> 2, piece = idx & 3;
        const int prow = pix / C2PW, pcol = pix - prow * C2PW;
        const int iy = row0 - 2 + prow, ix = pcol - 2;
        const bool item = idx < C2PIX * 4, inside = item && iy >= 0 && iy < HP1 && ix >= 0 && ix < WP1;
        psrc[u] = inside ? (iy * WP1 + ix) * 16 + piece : -1;
        pdst[u] = item ? pix * C2PS + piece : -1;
    }
    uint4 pr[NIT];
#define C2_PLOAD(CG)                                                                                          \
    _Pragma("unroll") for (int u = 0; u < NIT; u++) {                                                           \
        const uint4 v_ = inq[max(psrc[u], 0) + 4 * (CG)];                                                       \
        const uint32_t k_ = psrc[u] >= 0 ? 0xffffffffu : 0u;                                                    \
        pr[u] = make_uint4(v_.x & k_, v_.y & k_, v_.z & k_, v_.w & k_);                                         \
    }
#define C2_PSTORE()                                                                                           \
    _Pragma("unroll") for (int u = 0; u < NIT; u++) if (pdst[u] >= 0) s_patch[pdst[u]] = pr[u];
    // A operand bases of the wave's four 32-row tiles: row m -> output pixel (m / 42, m % 42) -> patch pixel of tap (0, 0)
    int abase[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = min(32 * i + lr, C2R * W2 - 1);             // rows past the block's pixels repeat the last one (never stored)
        const int orow = m / W2, ocol = m - orow * W2;
        abase[i] = (orow * C2PW + ocol) * C2PS + lk;
    }
    const int wbase = (32 * wave + lr) * 2 + lk;                   // this lane's uint4 inside a (stage, plane) block of 256
    uint4 Bq[4][3];
#define C2_BLOAD(SET, SPRIME)                                                                                 \
    {                                                                                                         \
        const int sp_ = min((SPRIME), 63), S_ = (sp_ & 15) * 4 + (sp_ >> 4);      /* stage index of the weight slab: tap * 4 + group */ \
        const uint4* w_ = wt3 + (size_t)S_ * 768 + wbase;                                                     \
        Bq[SET][0] = w_[0]; Bq[SET][1] = w_[256]; Bq[SET][2] = w_[512];                                       \
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

    C2_PLOAD(0)
    C2_BLOAD(0, 0)
    C2_BLOAD(1, 1)
    C2_PSTORE()
    __syncthreads();
    for (int cg = 0; cg < 4; cg++) {
        C2_PLOAD(min(cg + 1, 3))                                   // next group's patch: in flight under this group's 16 taps (branch-free)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 16; tap++) {
            C2_BLOAD((tap + 2) & 3, cg * 16 + tap + 2)             // two K steps ahead (past the end: the last one again)
            __builtin_amdgcn_sched_barrier(0);                     // ... and issued HERE: left alone, the scheduler sinks every load to its first use (vmcnt(0) in front of each MFMA group)
            const int toff = ((tap >> 2) * C2PW + (tap & 3)) * C2PS;
            cv_f16x8 Ah[4], Am[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Ah[i] = __builtin_bit_cast(cv_f16x8, s_patch[abase[i] + toff]);
                Am[i] = __builtin_bit_cast(cv_f16x8, s_patch[abase[i] + toff + 2]);
            }
            __builtin_amdgcn_sched_barrier(0);                     // all eight operand reads are issued before the first MFMA waits for one
            const cv_f16x8 B0 = __builtin_bit_cast(cv_f16x8, Bq[tap & 3][0]), B1 = __builtin_bit_cast(cv_f16x8, Bq[tap & 3][1]),
                           B2 = __builtin_bit_cast(cv_f16x8, Bq[tap & 3][2]);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Am[i], B2, acc[i], 0, 0, 0);        // m'_a h_b
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i], B1, acc[i], 0, 0, 0);        // h_a m'_b
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i], B0, acc[i], 0, 0, 0);        // h_a 2^11 h_b
        }
        __syncthreads();                                           // every wave has read the patch
        if (cg < 3) C2_PSTORE()
        __syncthreads();
    }
#undef C2_PLOAD
#undef C2_PSTORE
#undef C2_BLOAD
    const int n = 32 * wave + lr;
    const float bias = b2[n];
    float* ob = out + ((size_t)img * M2 + (size_t)row0 * W2) * CV_BN + n;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float v = acc[i][r] * (1.0f / 2048.f) + bias;
            if (m < nvalid) ob[(size_t)m * CV_BN] = relu ? fmaxf(v, 0.f) : v;
        }
}

#undef CV2_LOAD_STAGE

// ---- conv1 + ReLU + max-pool + LRN on the f16 matrix cores (round 3) ----
// The VALU form above spends 234 wave instructions per pooled pixel (scalar-window bookkeeping, 15 multiply-adds per conv output, every conv output
// computed 1.56 times) — and VALU issue is what the whole pipeline is short of.  Here a block owns 2 x 4 pooled pixels = 5 x 9 conv outputs (18 KB of LDS and 70 registers: it fits into what six FAST blocks leave of a CU —
// the 2 x 8 tile was 20 % faster alone and 2.5 % slower under the pipeline):
//   1. its 13 x 21 input patch goes to LDS as two f16 planes (a = h + m' 2^-11, as k_conv2_f16x3 splits its operands);
//   2. the conv outputs are a [64 x 32] x [32 x 64] product (K = 5 rows x 6 slots, the sixth and two more with zero weights; stride-2 im2col = 4 dword
//      LDS reads per lane, plane and K step at constant offsets from the lane's window origin), three partial products into one accumulator;
//   3. bias and ReLU go into an LDS conv map [64][64]; 4. lane = channel: 3 x 3 maximum over the map (Caffe's clipped ceil-mode windows), LRN over
//      lane shuffles, one 256-byte row per pooled pixel out.
// ~80 instead of 234 wave instructions per pooled pixel, no conv output computed twice inside a block.  Same range conditions as k_conv2_f16x3.
constexpr int C1T_PH = 2, C1T_PW = 4, C1T_CH = 2 * C1T_PH + 1, C1T_CW = 2 * C1T_PW + 1;       // pooled tile, conv outputs it needs
constexpr int C1T_IH = 2 * (C1T_CH - 1) + 5, C1T_IW = 2 * (C1T_CW - 1) + 5, C1T_IP = 24;       // input patch 13 x 21, 24 halfs per LDS row
constexpr int C1T_NCP = C1T_CH * C1T_CW, C1T_NM = (C1T_NCP + 31) / 32, C1T_CP = 66;             // 45 conv pixels, 2 M tiles, conv map pitch (floats)
constexpr int C1T_TX = (WP1 + C1T_PW - 1) / C1T_PW, C1T_TY = (HP1 + C1T_PH - 1) / C1T_PH;
static_assert(C1T_NM <= 4, "one M tile per wave");

__global__ __launch_bounds__(256) void k_conv1_f16x3_pool_lrn(const float* __restrict__ in, const uint4* __restrict__ w1h /*[2 n tiles][2 k steps][3: Hs, m', h][64 lanes]*/,
                                                              const float* __restrict__ b1, int relu, LrnP lp, float* __restrict__ out /*[HP1*WP1][64]*/) {
    MYSLAM_SIDE_PRIO();
    __shared__ __attribute__((aligned(8))) _Float16 s_h[C1T_IH * C1T_IP + 4], s_m[C1T_IH * C1T_IP + 4];
    static_assert(C1T_IP % 2 == 0, "dword reads of pixel pairs");
    __shared__ float s_conv[32 * C1T_NM * C1T_CP];
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ty = blockIdx.x / C1T_TX, tx = blockIdx.x - ty * C1T_TX;
    const int py0 = ty * C1T_PH, px0 = tx * C1T_PW;
    // 1. input patch: padded-plane rows 4 py0 .., columns 4 px0 .. (rows / columns past the plane only feed conv outputs outside the map: zeros)
    const float* I = in + (size_t)b * IN_PLANE;
    static_assert(C1T_IP % 4 == 0 && C1T_IH * (C1T_IP / 4) <= 256 && (4 * C1T_PW) % 4 == 0, "one float4 of the patch per thread");
    if (t < C1T_IH * (C1T_IP / 4)) {                                   // whole LDS rows (the zero-weight K slots still need finite operands), 4 pixels per thread
        const int r = t / (C1T_IP / 4), c4 = 4 * (t - r * (C1T_IP / 4));
        const int iy = 4 * py0 + r, ix = 4 * px0 + c4;                  // 16-byte aligned: 4 px0 and the plane pitch are multiples of 4
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy < IN_PH && ix + 4 <= IN_PW) a = *reinterpret_cast<const float4*>(I + iy * IN_PW + ix);
        if (c4 + 3 >= C1T_IW) { if (c4 + 1 >= C1T_IW) a.y = 0.f; if (c4 + 2 >= C1T_IW) a.z = 0.f; a.w = 0.f; if (c4 >= C1T_IW) a.x = 0.f; }
        const _Float16 h0 = (_Float16)a.x, h1 = (_Float16)a.y, h2 = (_Float16)a.z, h3 = (_Float16)a.w;
        typedef _Float16 c1_h4 __attribute__((ext_vector_type(4)));
        const c1_h4 hv = {h0, h1, h2, h3};
        const c1_h4 mv = {(_Float16)((a.x - (float)h0) * 2048.f), (_Float16)((a.y - (float)h1) * 2048.f), (_Float16)((a.z - (float)h2) * 2048.f),
                          (_Float16)((a.w - (float)h3) * 2048.f)};
        *reinterpret_cast<c1_h4*>(&s_h[r * C1T_IP + c4]) = hv;
        *reinterpret_cast<c1_h4*>(&s_m[r * C1T_IP + c4]) = mv;
    }
    __syncthreads();
    // 2. conv outputs of M tile `wave`.  K slot k = 6 ky + kx' with kx' = 0..5 (kx' = 5 and slots 30, 31 carry zero weights): a lane's 8 slots are
    //    4 aligned dwords of the patch (two horizontally adjacent pixels each) at constant offsets from its window origin — 4 ds_read_b32 per
    //    plane and K step instead of 8 two-byte reads + packing
    if (wave < C1T_NM) {
        const int mrow = lane & 31, kh = lane >> 5;
        const int cp = min(32 * wave + mrow, C1T_NCP - 1);             // rows past the conv pixels of the tile repeat the last one (their results are never read)
        const int cy = cp / C1T_CW, cx = cp - cy * C1T_CW;
        const uint32_t* ph = reinterpret_cast<const uint32_t*>(s_h) + (2 * cy * C1T_IP + 2 * cx) / 2;      // window origin: an even half index
        const uint32_t* pm = reinterpret_cast<const uint32_t*>(s_m) + (2 * cy * C1T_IP + 2 * cx) / 2;
        cv_f16x8 Ah[2], Am[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            uint32_t dh[4], dm[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // dword d of the 16 (d = 8 ks + 4 kh + j): patch row d / 3, pixel pair d % 3; d = 15 is padding (zero weights): dword 14 again
                const int d0 = min(8 * ks + j, 14), d1 = min(8 * ks + 4 + j, 14);
                const int o0 = (d0 / 3) * (C1T_IP / 2) + d0 % 3, o1 = (d1 / 3) * (C1T_IP / 2) + d1 % 3;      // compile-time constants
                const int o = kh ? o1 : o0;
                dh[j] = ph[o]; dm[j] = pm[o];
            }
            Ah[ks] = __builtin_bit_cast(cv_f16x8, make_uint4(dh[0], dh[1], dh[2], dh[3]));
            Am[ks] = __builtin_bit_cast(cv_f16x8, make_uint4(dm[0], dm[1], dm[2], dm[3]));
        }
        const float lo = relu ? 0.f : -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const uint4* wp = w1h + ((nt * 2 + ks) * 3) * 64 + lane;               // L2-resident, 12 KB
                const cv_f16x8 B0 = __builtin_bit_cast(cv_f16x8, wp[0]), B1 = __builtin_bit_cast(cv_f16x8, wp[64]), B2 = __builtin_bit_cast(cv_f16x8, wp[128]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Am[ks], B2, acc, 0, 0, 0);          // m'_a h_w
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], B1, acc, 0, 0, 0);          // h_a m'_w
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[ks], B0, acc, 0, 0, 0);          // h_a 2^11 h_w
            }
            const int n = 32 * nt + (lane & 31);
            const float bias = b1[n];
            float* o = s_conv + (32 * wave + 4 * kh) * C1T_CP + n;                                // accumulator row r -> conv pixel 32 wave + (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
            for (int r = 0; r < 16; r++) o[((r & 3) + 8 * (r >> 2)) * C1T_CP] = fmaxf(acc[r] * (1.0f / 2048.f) + bias, lo);      // all rows of the M tile; validity is the pool's business
        }
    }
    __syncthreads();
    // 3. lane = channel: pooled maximum + LRN(5) across the 64 channels (zero padded), as k_conv1_pool_lrn2
    for (int pp = wave; pp < C1T_PH * C1T_PW; pp += 4) {
        const int qy = pp / C1T_PW, qx = pp - qy * C1T_PW;
        const int py = py0 + qy, px = px0 + qx;
        if (py >= HP1 || px >= WP1) continue;                           // wave-uniform
        float m = -INFINITY;
        const float* cm = s_conv + ((2 * qy) * C1T_CW + 2 * qx) * C1T_CP + lane;
        if (2 * py + 2 < H1 && 2 * px + 2 < W1) {                       // wave-uniform: the whole 3 x 3 window lies inside the conv map
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int dx = 0; dx < 3; dx++) m = fmaxf(m, cm[(dy * C1T_CW + dx) * C1T_CP]);
        } else {                                                        // Caffe ceil-mode pooling: clipped windows at the map's last row / column
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int dx = 0; dx < 3; dx++)
                    if (2 * py + dy < H1 && 2 * px + dx < W1) m = fmaxf(m, cm[(dy * C1T_CW + dx) * C1T_CP]);
        }
        const float um1 = __shfl_up(m, 1, 64), um2 = __shfl_up(m, 2, 64), dp1 = __shfl_down(m, 1, 64), dp2 = __shfl_down(m, 2, 64);
        const float v0 = lane >= 2 ? um2 : 0.f, v1 = lane >= 1 ? um1 : 0.f, v3 = lane <= 62 ? dp1 : 0.f, v4 = lane <= 61 ? dp2 : 0.f;
        float ss = 0.f;
        ss += v0 * v0; ss += v1 * v1; ss += m * m; ss += v3 * v3; ss += v4 * v4;
        // the pooled / normalised map leaves as two f16 planes, a = h + m' 2^-11 — what k_conv2_f16x3 multiplies with (it used to split the f32
        // map itself, every value 16 times)
        const float y = m * lrn_factor(ss, lp);
        const _Float16 yh = (_Float16)y, ym = (_Float16)((y - (float)yh) * 2048.f);
        unsigned short* oh = reinterpret_cast<unsigned short*>(out + ((size_t)b * HP1 * WP1 + py * WP1 + px) * 64);      // 256 bytes per pixel
        const int oi = (lane >> 4) * 32 + (lane & 15);              // group of 16 channels: its 16 h halves, then its 16 m' halves
        oh[oi] = __builtin_bit_cast(unsigned short, yh);
        oh[oi + 16] = __builtin_bit_cast(unsigned short, ym);
    }
}

// ---- conv3 + ReLU + flatten (NCHW order) + L2 normalise ----
// CV3_NB blocks of 256 threads per image: 16 waves share the 266 output pixels (wave g takes pixels g, g + 16, ...), each lane keeps its 18
// weight quadruples (k = lane + 64 j) in registers; k_l2norm_1064 follows.  Round 2 ran ONE 1024-thread block per image: 16 waves x 101 registers need a nearly empty CU, so under the pipeline the
// kernel waited for whole CUs to drain (1.7 ms resident for 0.05 ms of work) — and the LCD chain it sits in is as long as the step.  A
// 256-thread block moves in as soon as one FAST block retires (+2.7 % frames/s).
constexpr int CV3_T = 256, CV3_NB = 4, CV3_NW = CV3_NB * CV3_T / 64;
__global__ __launch_bounds__(CV3_T) void k_conv3_norm(const float* __restrict__ in /*[B][16*21][128]*/,
                                                      const float* __restrict__ w3t /*[1152][4]*/, const float* __restrict__ b3,
                                                      float* __restrict__ out /*[B][1064], not yet normalised*/, int relu) {
    MYSLAM_SIDE_PRIO();
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gw = blockIdx.x * (CV3_T / 64) + wave;
    const float* I = in + (size_t)b * HP2 * WP2 * 128;
    const float4* Wt = reinterpret_cast<const float4*>(w3t);
    float* O = out + (size_t)b * MYSLAM_LCD_DIM;
    float4 w[18];
#pragma unroll
    for (int j = 0; j < 18; j++) w[j] = Wt[lane + 64 * j];
    for (int p = gw; p < H3 * W3; p += CV3_NW) {
        const int oy = p / W3, ox = p - oy * W3;
        float v[18];
#pragma unroll
        for (int j = 0; j < 18; j++) {
            const int k = lane + 64 * j, tap = k >> 7, ic = k & 127;
            v[j] = I[((size_t)(oy + tap / 3) * WP2 + ox + tap % 3) * 128 + ic];
        }
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 18; j++) { acc.x += v[j] * w[j].x; acc.y += v[j] * w[j].y; acc.z += v[j] * w[j].z; acc.w += v[j] * w[j].w; }
        acc.x = wave_sum_lane63_f32(acc.x); acc.y = wave_sum_lane63_f32(acc.y);
        acc.z = wave_sum_lane63_f32(acc.z); acc.w = wave_sum_lane63_f32(acc.w);
        if (lane == 63) {
            float r[4] = {acc.x + b3[0], acc.y + b3[1], acc.z + b3[2], acc.w + b3[3]};
#pragma unroll
            for (int c = 0; c < 4; c++) O[c * H3 * W3 + p] = relu ? fmaxf(r[c], 0.f) : r[c];
        }
    }
}

// L2 normalisation of the 1064 values of an image (deeplcd.cpp:84-90), one wave per image, in place.  A kernel of its own: the blocks of an
// image run on different XCDs, and making their values visible to a "last block" inside k_conv3 takes an agent-scope release — an L2
// write-back per block on this chip (measured: 0.05 -> 0.51 ms per 512 images).
__global__ __launch_bounds__(256) void k_l2norm_1064(float* __restrict__ out, int batch) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= batch) return;
    float* O = out + (size_t)b * MYSLAM_LCD_DIM;
    constexpr int NV = (MYSLAM_LCD_DIM + 63) / 64;
    float vv[NV];
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const int i = lane + 64 * u;
        vv[u] = i < MYSLAM_LCD_DIM ? O[i] : 0.f;
        ss += vv[u] * vv[u];
    }
    ss = wave_sum_lane63_f32(ss);
    const float nrm = sqrtf(__shfl(ss, 63, 64));                              // deeplcd.cpp:88
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const int i = lane + 64 * u;
        if (i < MYSLAM_LCD_DIM) O[i] = vv[u] / nrm;
    }
}

// ------------------------------------------------------------------------------------------------
// Generic layer kernels (NHWC f32): any list of Convolution / ReLU / max-Pooling (Caffe ceil mode) / LRN (across channels) layers.
// Slower than the fused kernels (no fusion, no matrix cores) — they exist so that a layer list that differs from SURVEY A.6
// (pad, LRN window, an extra layer ...) still runs, and they serve the stage taps of the parity tests.
// ------------------------------------------------------------------------------------------------
// one wave per output pixel, lane -> output channels lane, lane + 64, ...; the input value of a tap is wave-uniform
__global__ __launch_bounds__(256) void k_conv_generic(const float* __restrict__ in, size_t in_stride, int in_pitch /*floats per row*/, int H, int W, int IC,
                                                      const float* __restrict__ wt /*[K*K*IC][OC]*/, const float* __restrict__ bias, int OC, int K, int S,
                                                      int P, int relu, float* __restrict__ out /*[OH*OW][OC]*/, int OH, int OW) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (p >= OH * OW) return;
    const int oy = p / OW, ox = p - oy * OW;
    const float* I = in + (size_t)b * in_stride;
    for (int oc0 = 0; oc0 < OC; oc0 += 64) {
        const int oc = oc0 + lane;
        float acc = 0.f;
        for (int ky = 0; ky < K; ky++) {
            const int iy = oy * S + ky - P;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < K; kx++) {
                const int ix = ox * S + kx - P;
                if (ix < 0 || ix >= W) continue;
                const float* src = I + (size_t)iy * in_pitch + (size_t)ix * IC;
                const float* wk = wt + (size_t)((ky * K + kx) * IC) * OC + oc;
                if (oc < OC)
                    for (int ic = 0; ic < IC; ic++) acc += wk[(size_t)ic * OC] * src[ic];
            }
        }
        if (oc < OC) {
            const float v = acc + bias[oc];
            out[((size_t)b * OH * OW + p) * OC + oc] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

__global__ __launch_bounds__(256) void k_relu_generic(float* __restrict__ x, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = fmaxf(x[i], 0.f);
}

// Caffe max pooling, ceil mode, windows clipped to the input (pad 0); thread per output element
__global__ __launch_bounds__(256) void k_pool_generic(const float* __restrict__ in, int H, int W, int C, int K, int S, float* __restrict__ out, int OH, int OW) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)OH * OW * C) return;
    const int c = (int)(i % C), p = (int)(i / C), oy = p / OW, ox = p - oy * OW;
    const int y0 = oy * S, x0 = ox * S, y1 = min(y0 + K, H), x1 = min(x0 + K, W);
    const float* I = in + (size_t)b * H * W * C;
    float m = -INFINITY;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) m = fmaxf(m, I[((size_t)y * W + x) * C + c]);
    out[(size_t)b * OH * OW * C + i] = m;
}

// Caffe LRN across channels: y = x * (k + alpha / n * sum_{|j - c| <= n/2} x_j^2)^-beta
__global__ __launch_bounds__(256) void k_lrn_generic(const float* __restrict__ in, int HW, int C, int n, LrnP lp, float* __restrict__ out) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)HW * C) return;
    const int c = (int)(i % C);
    const float* v = in + (size_t)b * HW * C + i - c;
    float ss = 0.f;
    for (int j = max(0, c - n / 2); j <= min(C - 1, c + n / 2); j++) ss += v[j] * v[j];
    out[(size_t)b * HW * C + i] = v[c] * lrn_factor(ss, lp);
}

// flatten in Caffe's NCHW order + L2 normalise (deeplcd.cpp:80-88); one block per image, N = C * HW = 1064
__global__ __launch_bounds__(256) void k_flatten_norm_generic(const float* __restrict__ in /*[HW][C]*/, int HW, int C, float* __restrict__ out) {
    __shared__ float s_red[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* I = in + (size_t)b * HW * C;
    float ss = 0.f;
    for (int i = threadIdx.x; i < HW * C; i += 256) ss += I[i] * I[i];
    ss = wave_sum_lane63_f32(ss);
    if (lane == 63) s_red[wave] = ss;
    __syncthreads();
    const float nrm = sqrtf(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    for (int i = threadIdx.x; i < HW * C; i += 256) { const int c = i % C, p = i / C; out[(size_t)b * HW * C + (size_t)c * HW + p] = I[i] / nrm; }
}

static void lcd_resize_tables(int ssize, int dsize, bool is_x, std::vector<int32_t>& ofs, std::vector<int16_t>& coef) {
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    ofs.resize(dsize); coef.resize(2 * dsize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (is_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = (int16_t)lrintf((1.f - f) * 2048.f);
        coef[2 * d + 1] = (int16_t)lrintf(f * 2048.f);
    }
}

// the SURVEY A.6 layer list (the defaults of the flat CALCW1 blob)
static const myslam_calc_layer kDefaultLayers[10] = {
    {MYSLAM_CALC_CONV, 64, 5, 2, 4, 0, 0.f, 0.f, 0.f},  {MYSLAM_CALC_RELU, 0, 0, 0, 0, 0, 0.f, 0.f, 0.f},
    {MYSLAM_CALC_POOL_MAX, 0, 3, 2, 0, 0, 0.f, 0.f, 0.f}, {MYSLAM_CALC_LRN, 0, 0, 0, 0, 5, 1e-4f, 0.75f, 1.f},
    {MYSLAM_CALC_CONV, 128, 4, 1, 2, 0, 0.f, 0.f, 0.f}, {MYSLAM_CALC_RELU, 0, 0, 0, 0, 0, 0.f, 0.f, 0.f},
    {MYSLAM_CALC_POOL_MAX, 0, 3, 2, 0, 0, 0.f, 0.f, 0.f}, {MYSLAM_CALC_LRN, 0, 0, 0, 0, 5, 1e-4f, 0.75f, 1.f},
    {MYSLAM_CALC_CONV, 4, 3, 1, 0, 0, 0.f, 0.f, 0.f},   {MYSLAM_CALC_RELU, 0, 0, 0, 0, 0, 0.f, 0.f, 0.f}};

// shape walk of a layer list from 1 x IN_H x IN_W: validates it, counts the weights, returns the output size
struct LayerShape { int C, H, W; };
static int walk_layers(const myslam_calc_layer* L, int n, std::vector<LayerShape>& shapes, size_t& nweights) {
    LayerShape s{1, IN_H, IN_W};
    nweights = 0; shapes.clear();
    if (!L || n < 1 || n > 64) return MYSLAM_ERR_INVALID;
    for (int i = 0; i < n; i++) {
        const myslam_calc_layer& l = L[i];
        switch (l.type) {
            case MYSLAM_CALC_CONV:
                if (l.num_output < 1 || l.num_output > 1024 || l.kernel < 1 || l.kernel > 15 || l.stride < 1 || l.pad < 0 || l.pad >= l.kernel) return MYSLAM_ERR_INVALID;
                if (s.H + 2 * l.pad < l.kernel || s.W + 2 * l.pad < l.kernel) return MYSLAM_ERR_INVALID;
                nweights += (size_t)l.num_output * s.C * l.kernel * l.kernel + (size_t)l.num_output;
                s = {l.num_output, (s.H + 2 * l.pad - l.kernel) / l.stride + 1, (s.W + 2 * l.pad - l.kernel) / l.stride + 1};
                break;
            case MYSLAM_CALC_RELU: break;
            case MYSLAM_CALC_POOL_MAX: {
                if (l.kernel < 1 || l.stride < 1 || l.kernel > s.H || l.kernel > s.W) return MYSLAM_ERR_INVALID;
                if (l.pad != 0) return MYSLAM_ERR_UNSUPPORTED;
                int oh = (int)ceilf((float)(s.H - l.kernel) / l.stride) + 1, ow = (int)ceilf((float)(s.W - l.kernel) / l.stride) + 1;      // Caffe ceil mode
                // Caffe clips the last window only when pad > 0 (pooling_layer.cpp); with pad = 0 a stride larger than the kernel can leave a
                // last window that starts outside the map (Caffe emits -FLT_MAX there): no CALC-like net does that — refused
                if ((oh - 1) * l.stride >= s.H || (ow - 1) * l.stride >= s.W) return MYSLAM_ERR_UNSUPPORTED;
                s = {s.C, oh, ow};
                break;
            }
            case MYSLAM_CALC_LRN:
                if (l.local_size < 1 || !(l.local_size & 1) || !(l.k > 0.f) || !(l.alpha >= 0.f)) return MYSLAM_ERR_INVALID;
                break;
            default: return MYSLAM_ERR_UNSUPPORTED;
        }
        shapes.push_back(s);
    }
    if ((size_t)s.C * s.H * s.W != MYSLAM_LCD_DIM) return MYSLAM_ERR_UNSUPPORTED;          // deeplcd.cpp:80 asserts 1064 outputs
    return MYSLAM_OK;
}

// does the list have the geometry the fused kernels are written for?  (optional ReLUs; a missing LRN = the identity LRN)
struct FusedPlan { bool ok; int relu[3]; LrnP lrn[2]; };
static FusedPlan match_fused(const myslam_calc_layer* L, int n) {
    FusedPlan f{false, {0, 0, 0}, {{0.f, 0.75f, 1.f, 1}, {0.f, 0.75f, 1.f, 1}}};
    const int geo[3][4] = {{C1, 5, 2, 4}, {C2, 4, 1, 2}, {C3, 3, 1, 0}};
    int i = 0;
    for (int blk = 0; blk < 3; blk++) {
        if (i >= n || L[i].type != MYSLAM_CALC_CONV || L[i].num_output != geo[blk][0] || L[i].kernel != geo[blk][1] || L[i].stride != geo[blk][2] ||
            L[i].pad != geo[blk][3])
            return f;
        i++;
        if (i < n && L[i].type == MYSLAM_CALC_RELU) { f.relu[blk] = 1; i++; }
        if (blk == 2) break;
        if (i >= n || L[i].type != MYSLAM_CALC_POOL_MAX || L[i].kernel != 3 || L[i].stride != 2 || L[i].pad != 0) return f;
        i++;
        if (i < n && L[i].type == MYSLAM_CALC_LRN) {
            if (L[i].local_size != 5) return f;
            f.lrn[blk] = {L[i].alpha / (float)L[i].local_size, L[i].beta, L[i].k, L[i].beta == 0.75f ? 1 : 0};
            i++;
        }
    }
    f.ok = (i == n);
    return f;
}

}  // namespace myslam_hip

using namespace myslam_hip;

struct myslam_lcd {
    hipStream_t stream = nullptr;
    // the model
    std::vector<myslam_calc_layer> layers; std::vector<LayerShape> shapes;
    FusedPlan fused{};                 // fused.ok: the layer list has the geometry of the fused kernels
    int forceGeneric = 0;              // myslam_lcd_set_option(GENERIC_KERNELS)
    int forceBf16 = 0;                 // myslam_lcd_set_option(CONV2_BF16X6): keep the six-product bf16 kernel
    // the two f16 matrix-core kernels go together: conv1 hands conv2 its input as two f16 planes (h, m'); the bf16 / f32 pair keeps an f32 map
    bool f16_family() const { return d_w1h && d_w2h && !forceBf16; }
    int skipMask = 0;                  // myslam_lcd_set_option(SKIP_KERNELS): TIMING ONLY — bit 0 input, 1 conv1, 2 conv2, 3 pool2, 4 conv3 + norm are not launched
    std::vector<float*> d_wt, d_b;     // per convolution: weights re-laid out as [K*K*IC][OC], bias
    uint4* d_w2s = nullptr;            // fused path: conv2 weights split into three bf16 pieces, [stage][piece][n][k half] x 8 bf16
    uint4* d_w1h = nullptr;            // conv1 weights as MFMA operands of k_conv1_f16x3_pool_lrn ([n tile][k step][2^11 h, m', h][lane]), same condition as d_w2h
    uint4* d_w2h = nullptr;            // the same as three f16 planes (2^11 h, m', h) when the model's ranges allow k_conv2_f16x3, else nullptr
    size_t actMax = 0;                 // largest activation (floats per image) of the generic path
    // resize tables for the current source size
    int rows = 0, cols = 0;
    int32_t *d_xofs = nullptr, *d_yofs = nullptr; int16_t *d_xa = nullptr, *d_yb = nullptr;
    // batch buffers
    int batchCap = 0; size_t blurBytes = 0; int blurPitch = 0;
    uint8_t* d_blur = nullptr;
    float *d_in = nullptr, *d_p1 = nullptr, *d_a2 = nullptr, *d_p2 = nullptr;      // fused path activations
    float *d_g0 = nullptr, *d_g1 = nullptr; int genericCap = 0;                    // generic path ping-pong (allocated on first use)
    // host-entry staging
    uint8_t* d_stageImg = nullptr; size_t stageBytes = 0; float* d_stageOut = nullptr; float* h_stageOut = nullptr;

    int ensure_tables(int r, int c);
    int ensure_batch(int batch, int r, int c);
    int ensure_generic(int batch);
    int forward_generic(int batch, float* d_out, int stop_after, float** tap);
    int describe(uint8_t* d_imgs, int batch, int r, int c, int step, size_t stride, int blur_in_place, float* d_out);
    void free_all();
};

template <typename T>
static int lcd_alloc(T*& p, size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    if (!n) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipMalloc((void**)&p, n * sizeof(T)));
    return MYSLAM_OK;
}

void myslam_lcd::free_all() {
    void* ptrs[] = {d_w2s, d_w2h, d_w1h, d_xofs, d_yofs, d_xa, d_yb, d_blur, d_in, d_p1, d_a2, d_p2, d_g0, d_g1, d_stageImg, d_stageOut};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (float* p : d_wt) if (p) (void)hipFree(p);
    for (float* p : d_b) if (p) (void)hipFree(p);
    if (h_stageOut) (void)hipHostFree(h_stageOut);
}

int myslam_lcd::ensure_tables(int r, int c) {
    if (r == rows && c == cols) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<int32_t> xo, yo; std::vector<int16_t> xa, yb;
    lcd_resize_tables(c, IN_W, true, xo, xa);
    lcd_resize_tables(r, IN_H, false, yo, yb);
    int rc;
    if ((rc = lcd_alloc(d_xofs, xo.size())) || (rc = lcd_alloc(d_xa, xa.size())) || (rc = lcd_alloc(d_yofs, yo.size())) ||
        (rc = lcd_alloc(d_yb, yb.size())))
        return rc;
    if ((rc = upload_table(d_xofs, xo.data(), xo.size() * 4))) return rc;
    if ((rc = upload_table(d_xa, xa.data(), xa.size() * 2))) return rc;
    if ((rc = upload_table(d_yofs, yo.data(), yo.size() * 4))) return rc;
    if ((rc = upload_table(d_yb, yb.data(), yb.size() * 2))) return rc;
    rows = r; cols = c; batchCap = 0;
    return MYSLAM_OK;
}

int myslam_lcd::ensure_batch(int batch, int r, int c) {
    int rc = ensure_tables(r, c);
    if (rc) return rc;
    if (batch <= batchCap) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(stream));
    blurPitch = (c + 63) / 64 * 64;
    blurBytes = ((size_t)blurPitch * r + 255) / 256 * 256;
    if ((rc = lcd_alloc(d_blur, blurBytes * batch))) return rc;
    if ((rc = lcd_alloc(d_in, (size_t)batch * IN_PLANE))) return rc;
    MYSLAM_HIP_CHECK(hipMemsetAsync(d_in, 0, (size_t)batch * IN_PLANE * sizeof(float), stream));     // the padding stays zero: writers touch the interior only
    if (fused.ok) {
        if ((rc = lcd_alloc(d_p1, (size_t)batch * HP1 * WP1 * C1))) return rc;
        if ((rc = lcd_alloc(d_a2, (size_t)batch * H2 * W2 * C2))) return rc;
        if ((rc = lcd_alloc(d_p2, (size_t)batch * HP2 * WP2 * C2))) return rc;
    }
    batchCap = batch; genericCap = 0;
    return MYSLAM_OK;
}

int myslam_lcd::ensure_generic(int batch) {
    if (batch <= genericCap) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(stream));
    int rc;
    if ((rc = lcd_alloc(d_g0, (size_t)batch * actMax)) || (rc = lcd_alloc(d_g1, (size_t)batch * actMax))) return rc;
    genericCap = batch;
    return MYSLAM_OK;
}

// layer by layer on the generic kernels.  stop_after >= 0: stop after that layer index and return its activation in *tap.
int myslam_lcd::forward_generic(int batch, float* d_out, int stop_after, float** tap) {
    int rc = ensure_generic(batch);
    if (rc) return rc;
    hipStream_t s = stream;
    const float* cur = d_in + IN_PAD * IN_PW + IN_PAD; size_t curStride = IN_PLANE; int curPitch = IN_PW;      // the interior of the padded input plane
    LayerShape sh{1, IN_H, IN_W};
    float* bufs[2] = {d_g0, d_g1}; int nb = 0, conv = 0;
    ScopedProf sp(P_CONV2, s);
    for (size_t i = 0; i < layers.size(); i++) {
        const myslam_calc_layer& l = layers[i];
        const LayerShape o = shapes[i];
        if (l.type == MYSLAM_CALC_RELU) {
            if (cur == d_in + IN_PAD * IN_PW + IN_PAD) return MYSLAM_ERR_UNSUPPORTED;                       // a ReLU on the raw input (never in a CALC net)
            const size_t n = (size_t)batch * o.C * o.H * o.W;
            hipLaunchKernelGGL(k_relu_generic, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, const_cast<float*>(cur), n);
        } else {
            float* dst = bufs[nb]; nb ^= 1;
            if (l.type == MYSLAM_CALC_CONV) {
                hipLaunchKernelGGL(k_conv_generic, dim3((o.H * o.W + 3) / 4, batch), dim3(256), 0, s, cur, curStride, curPitch, sh.H, sh.W, sh.C,
                                   d_wt[conv], d_b[conv], o.C, l.kernel, l.stride, l.pad, 0, dst, o.H, o.W);
                conv++;
            } else if (l.type == MYSLAM_CALC_POOL_MAX) {
                if (curPitch != sh.W * sh.C) return MYSLAM_ERR_UNSUPPORTED;
                const size_t n = (size_t)o.C * o.H * o.W;
                hipLaunchKernelGGL(k_pool_generic, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, s, cur, sh.H, sh.W, sh.C, l.kernel, l.stride, dst, o.H, o.W);
            } else {
                if (curPitch != sh.W * sh.C) return MYSLAM_ERR_UNSUPPORTED;
                const size_t n = (size_t)o.C * o.H * o.W;
                const LrnP lp{l.alpha / (float)l.local_size, l.beta, l.k, l.beta == 0.75f ? 1 : 0};
                hipLaunchKernelGGL(k_lrn_generic, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, s, cur, o.H * o.W, o.C, l.local_size, lp, dst);
            }
            cur = dst; curStride = (size_t)o.C * o.H * o.W; curPitch = o.W * o.C;
        }
        sh = o;
        if ((int)i == stop_after) { if (tap) *tap = const_cast<float*>(cur); MYSLAM_HIP_CHECK(hipGetLastError()); return MYSLAM_OK; }
    }
    hipLaunchKernelGGL(k_flatten_norm_generic, dim3(batch), dim3(256), 0, s, cur, sh.H * sh.W, sh.C, d_out);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

static int lcd_forward(myslam_lcd* h, int batch, float* d_out) {
    if (!h->fused.ok || h->forceGeneric) return h->forward_generic(batch, d_out, -1, nullptr);
    hipStream_t s = h->stream;
    const FusedPlan& f = h->fused;
    if (!(h->skipMask & 2)) {
        ScopedProf sp(P_CONV1, s);
        if (h->f16_family()) hipLaunchKernelGGL(k_conv1_f16x3_pool_lrn, dim3(C1T_TX * C1T_TY, batch), dim3(256), 0, s, h->d_in, h->d_w1h, h->d_b[0], f.relu[0], f.lrn[0], h->d_p1);
        else hipLaunchKernelGGL(k_conv1_pool_lrn2, dim3((HT1 * WT1 + 3) / 4, batch), dim3(256), 0, s, h->d_in, h->d_wt[0], h->d_b[0], f.relu[0], f.lrn[0], h->d_p1);
    }
    if (!(h->skipMask & 4)) {
        ScopedProf sp(P_CONV2, s);
        const int Mtotal = batch * M2;
        if (h->f16_family()) hipLaunchKernelGGL(k_conv2_f16x3, dim3(C2NB, batch), dim3(256), 0, s, h->d_p1, h->d_w2h, h->d_b[1], h->d_a2, batch, f.relu[1]);
        else hipLaunchKernelGGL(k_conv2_bf16x6, dim3((Mtotal + 127) / 128), dim3(256), 0, s, h->d_p1, h->d_w2s, h->d_b[1], h->d_a2, Mtotal, f.relu[1]);
    }
    if (!(h->skipMask & 8)) {
        ScopedProf sp(P_POOL2, s);
        hipLaunchKernelGGL(k_pool_lrn128_2x2, dim3((((HP2 + 1) / 2) * ((WP2 + 1) / 2) + 3) / 4, batch), dim3(256), 0, s, h->d_a2, H2, W2, HP2, WP2, f.lrn[1], h->d_p2);
    }
    if (!(h->skipMask & 16)) {
        ScopedProf sp(P_CONV3, s);
        hipLaunchKernelGGL(k_conv3_norm, dim3(CV3_NB, batch), dim3(CV3_T), 0, s, h->d_p2, h->d_wt[2], h->d_b[2], d_out, f.relu[2]);
        hipLaunchKernelGGL(k_l2norm_1064, dim3((batch + 3) / 4), dim3(256), 0, s, d_out, batch);
    }
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

int myslam_lcd::describe(uint8_t* d_imgs, int batch, int r, int c, int step, size_t stride, int blur_in_place, float* d_out) {
    if (!d_imgs || batch <= 0 || r <= 0 || c <= 0 || step < c || !d_out) return MYSLAM_ERR_INVALID;
    int rc = ensure_batch(batch, r, c);
    if (rc) return rc;
    if (!(skipMask & 1)) {
        ScopedProf sp(P_LCD_PRE, stream);
        if (!blur_in_place) {
            LcdTaps tp;
            gauss_q8(1, tp.q);
            hipLaunchKernelGGL(k_lcd_input_fused, dim3((IN_H * IN_W + 255) / 256, batch), dim3(256), 0, stream, d_imgs, c, r, step, stride,
                               d_xofs, d_xa, d_yofs, d_yb, tp, d_in);
        } else {
            BlurArgs a{};                                 // GaussianBlur(img, img, Size(7,7), 0)  deeplcd.cpp:46  (src0 / n0 = 0: no in-place images)
            a.src = d_imgs; a.dst = d_blur; a.w = c; a.h = r; a.spitch = step; a.dpitch = blurPitch; a.sstride = stride; a.dstride = blurBytes;
            gauss_q8(1, a.q);
            launch_blur(a, batch, stream);
            for (int b = 0; b < batch; b++)               // the reference mutates the caller's pixels
                MYSLAM_HIP_CHECK(hipMemcpy2DAsync(d_imgs + (size_t)b * stride, step, d_blur + (size_t)b * blurBytes, blurPitch, c, r,
                                                  hipMemcpyDeviceToDevice, stream));
            hipLaunchKernelGGL(k_lcd_input, dim3((IN_H * IN_W + 255) / 256, batch), dim3(256), 0, stream, d_blur, c, r, blurPitch, blurBytes,
                               d_xofs, d_xa, d_yofs, d_yb, d_in);       // cv::resize(.., Size(160,120))  deeplcd.cpp:48-50
        }
    }
    return lcd_forward(this, batch, d_out);
}

static int lcd_create(myslam_lcd** out, const myslam_calc_layer* layers, int nlayers, const float* weights, size_t nweights) {
    if (!out || !weights) return MYSLAM_ERR_INVALID;
    std::vector<LayerShape> shapes; size_t need = 0;
    int rc = walk_layers(layers, nlayers, shapes, need);
    if (rc) return rc;
    if (nweights != need) return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    myslam_lcd* h = new myslam_lcd();
    h->layers.assign(layers, layers + nlayers); h->shapes = shapes;
    h->fused = match_fused(layers, nlayers);
    for (auto& s : shapes) h->actMax = std::max(h->actMax, (size_t)s.C * s.H * s.W);
    auto fail = [&](int code) { h->free_all(); delete h; return code; };
    // convolution weights: [OC][IC][K][K] -> [K*K*IC][OC] (k = (ky*K + kx)*IC + ic: channel runs contiguous, one coalesced row per k)
    const float* w = weights; int ic = 1;
    std::vector<float> w2t;
    for (int i = 0, conv = 0; i < nlayers; i++) {
        if (layers[i].type != MYSLAM_CALC_CONV) continue;
        const int OC = layers[i].num_output, K = layers[i].kernel, KK = K * K;
        std::vector<float> wt((size_t)KK * ic * OC);
        for (int oc = 0; oc < OC; oc++)
            for (int c = 0; c < ic; c++)
                for (int t = 0; t < KK; t++) wt[((size_t)t * ic + c) * OC + oc] = w[((size_t)oc * ic + c) * KK + t];
        float *dw = nullptr, *db = nullptr;
        if (hipMalloc((void**)&dw, wt.size() * sizeof(float)) != hipSuccess) return fail(MYSLAM_ERR_HIP);
        h->d_wt.push_back(dw);
        if (hipMalloc((void**)&db, (size_t)OC * sizeof(float)) != hipSuccess) return fail(MYSLAM_ERR_HIP);
        h->d_b.push_back(db);
        if (upload_table(dw, wt.data(), wt.size() * sizeof(float)) != MYSLAM_OK ||
            upload_table(db, w + (size_t)OC * ic * KK, (size_t)OC * sizeof(float)) != MYSLAM_OK)
            return fail(MYSLAM_ERR_HIP);
        if (conv == 1) w2t.swap(wt);
        w += (size_t)OC * ic * KK + OC; ic = OC; conv++;
    }
    if (h->fused.ok) {
        // conv2 weights as three bf16 pieces (round to nearest even, exact residuals), stage-major so a stage's slab is contiguous
        auto to_bf16 = [](float x) -> uint16_t {
            uint32_t u; memcpy(&u, &x, 4);
            if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
            u += 0x7fffu + ((u >> 16) & 1u);
            return (uint16_t)(u >> 16);
        };
        auto from_bf16 = [](uint16_t b) -> float { const uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; };
        std::vector<uint16_t> w2s((size_t)64 * 3 * 128 * 16);
        for (int k = 0; k < K2; k++) {
            const int st = k >> 4, kk = k & 15;
            for (int oc = 0; oc < 128; oc++) {
                const float a = w2t[(size_t)k * 128 + oc];
                const uint16_t hb = to_bf16(a); const float r = a - from_bf16(hb);
                const uint16_t mb = to_bf16(r); const float q = r - from_bf16(mb);
                const uint16_t lb = to_bf16(q);
                const uint16_t pcs[3] = {hb, mb, lb};
                for (int p = 0; p < 3; p++) w2s[(((size_t)st * 3 + p) * 128 + oc) * 16 + kk] = pcs[p];
            }
        }
        if (hipMalloc((void**)&h->d_w2s, w2s.size() * 2) != hipSuccess ||
            upload_table(h->d_w2s, w2s.data(), w2s.size() * 2) != MYSLAM_OK)
            return fail(MYSLAM_ERR_HIP);
    }
    if (h->fused.ok) {
        // f16 planes for k_conv2_f16x3 — only when nothing can leave f16's range: |w2| < 2^5 (2^11 h must stay below 65504) and conv2's input,
        // the LRN'd / pooled conv1 map, bounded by sum |w1| + |b1| (inputs are pixels / 255) < 65504 with an LRN that cannot amplify (k >= 1)
        const float* w1 = weights; const int n1 = layers[0].num_output, k1 = layers[0].kernel * layers[0].kernel;     // conv1: 1 input channel
        double bound1 = 0, wmax2 = 0;
        for (int oc = 0; oc < n1; oc++) {
            double sacc = std::fabs((double)w1[(size_t)n1 * k1 + oc]);
            for (int t = 0; t < k1; t++) sacc += std::fabs((double)w1[(size_t)oc * k1 + t]);
            bound1 = std::max(bound1, sacc);
        }
        for (float v : w2t) wmax2 = std::max(wmax2, (double)std::fabs(v));
        const bool finite = std::isfinite(bound1) && std::isfinite(wmax2);
        if (finite && bound1 < 60000.0 && wmax2 < 31.0 && h->fused.lrn[0].k >= 1.0f && h->fused.lrn[0].beta >= 0.0f &&
            h->fused.lrn[0].aon >= 0.0f && std::isfinite(h->fused.lrn[0].aon) && std::isfinite(h->fused.lrn[0].beta)) {      // alpha < 0 would make (k + alpha/n ss)^-beta exceed 1 (walk_layers rejects it already)
            auto to_f16 = [](float x) -> uint16_t { const _Float16 v = (_Float16)x; uint16_t u; memcpy(&u, &v, 2); return u; };   // round to nearest even
            auto from_f16 = [](uint16_t b) -> float { _Float16 v; memcpy(&v, &b, 2); return (float)v; };
            std::vector<uint16_t> w2h((size_t)64 * 3 * 128 * 16);
            for (int k = 0; k < K2; k++) {
                const int st = k >> 4, kk = k & 15;
                for (int oc = 0; oc < 128; oc++) {
                    const float a = w2t[(size_t)k * 128 + oc];
                    const uint16_t hb = to_f16(a); const float hf = from_f16(hb);
                    const uint16_t pcs[3] = {to_f16(hf * 2048.f), to_f16((a - hf) * 2048.f), hb};
                    for (int p = 0; p < 3; p++) w2h[(((size_t)st * 3 + p) * 128 + oc) * 16 + kk] = pcs[p];
                }
            }
            if (hipMalloc((void**)&h->d_w2h, w2h.size() * 2) != hipSuccess ||
                upload_table(h->d_w2h, w2h.data(), w2h.size() * 2) != MYSLAM_OK)
                return fail(MYSLAM_ERR_HIP);
            double wmax1 = 0;
            for (int i = 0; i < n1 * k1; i++) wmax1 = std::max(wmax1, (double)std::fabs(w1[i]));
            if (wmax1 < 31.0 && k1 == 25 && n1 == 64) {
                // conv1 weights in the B-operand layout of v_mfma_f32_32x32x16_f16: lane (n = lane & 31, k half = lane >> 5) holds k = 16 ks + 8 kh + j
                std::vector<uint16_t> w1h((size_t)2 * 2 * 3 * 64 * 8);
                for (int nt = 0; nt < 2; nt++)
                    for (int ks = 0; ks < 2; ks++)
                        for (int ln = 0; ln < 64; ln++)
                            for (int j = 0; j < 8; j++) {
                                const int n = 32 * nt + (ln & 31), k = 16 * ks + 8 * (ln >> 5) + j;
                                const int ky = k / 6, kx = k % 6;                                   // K slot -> tap (kx = 5 and slots 30, 31: zero)
                                const float a = (ky < 5 && kx < 5) ? w1[(size_t)n * 25 + ky * 5 + kx] : 0.f;
                                const uint16_t hb = to_f16(a); const float hf = from_f16(hb);
                                const uint16_t pcs[3] = {to_f16(hf * 2048.f), to_f16((a - hf) * 2048.f), hb};
                                for (int pc = 0; pc < 3; pc++) w1h[((((size_t)(nt * 2 + ks) * 3 + pc) * 64 + ln) * 8) + j] = pcs[pc];
                            }
                if (hipMalloc((void**)&h->d_w1h, w1h.size() * 2) != hipSuccess ||
                    upload_table(h->d_w1h, w1h.data(), w1h.size() * 2) != MYSLAM_OK)
                    return fail(MYSLAM_ERR_HIP);
            }
        }
    }
    *out = h;
    return MYSLAM_OK;
}

extern "C" {

size_t myslam_lcd_nweights(void) { return NWEIGHTS; }

int myslam_lcd_default_layers(myslam_calc_layer* layers, int cap) {
    if (layers) { if (cap < 10) return MYSLAM_ERR_CAPACITY; memcpy(layers, kDefaultLayers, sizeof(kDefaultLayers)); }
    return 10;
}

int myslam_lcd_create(myslam_lcd** out, const float* weights, size_t nweights) {
    return lcd_create(out, kDefaultLayers, 10, weights, nweights);
}

int myslam_lcd_create_from_layers(myslam_lcd** out, const myslam_calc_layer* layers, int nlayers, const float* weights, size_t nweights) {
    return lcd_create(out, layers, nlayers, weights, nweights);
}

// host only: what myslam_lcd_create_from_caffe would load (parse + validate, no device needed)
int myslam_calc_parse_caffe(const char* prototxt_path, const char* caffemodel_path, myslam_calc_layer* layers, int cap, int* nlayers,
                            float* weights, size_t wcap, size_t* nweights) {
    myslam_caffe::Model m;
    int rc = myslam_caffe::load(prototxt_path, caffemodel_path, m);
    if (rc) return rc;
    if (m.in_c != 1 || m.in_h != IN_H || m.in_w != IN_W) return MYSLAM_ERR_UNSUPPORTED;      // the reference resizes every frame to 160 x 120 grey (deeplcd.cpp:50)
    std::vector<LayerShape> shapes; size_t need = 0;
    if ((rc = walk_layers(m.layers.data(), (int)m.layers.size(), shapes, need))) return rc;
    if (need != m.weights.size()) return MYSLAM_ERR_INVALID;
    if (nlayers) *nlayers = (int)m.layers.size();
    if (nweights) *nweights = m.weights.size();
    if (layers) { if (cap < (int)m.layers.size()) return MYSLAM_ERR_CAPACITY; memcpy(layers, m.layers.data(), m.layers.size() * sizeof(myslam_calc_layer)); }
    if (weights) { if (wcap < m.weights.size()) return MYSLAM_ERR_CAPACITY; memcpy(weights, m.weights.data(), m.weights.size() * sizeof(float)); }
    return MYSLAM_OK;
}

// DeepLCD::DeepLCD(prototxt, caffemodel)  deeplcd.cpp:10-31
int myslam_lcd_create_from_caffe(myslam_lcd** out, const char* prototxt_path, const char* caffemodel_path) {
    if (!out) return MYSLAM_ERR_INVALID;
    myslam_caffe::Model m;
    int rc = myslam_caffe::load(prototxt_path, caffemodel_path, m);
    if (rc) return rc;
    if (m.in_c != 1 || m.in_h != IN_H || m.in_w != IN_W) return MYSLAM_ERR_UNSUPPORTED;
    return lcd_create(out, m.layers.data(), (int)m.layers.size(), m.weights.data(), m.weights.size());
}

// own model files (little endian):
//   "CALCW1\0\0", u64 count, count f32                                                  — weights of the SURVEY A.6 layer list
//   "CALCW2\0\0", u32 nlayers, nlayers x myslam_calc_layer (36 bytes), u64 count, count f32 — layer list + weights
int myslam_lcd_create_from_file(myslam_lcd** out, const char* path) {
    if (!out || !path) return MYSLAM_ERR_INVALID;
    FILE* f = fopen(path, "rb");
    if (!f) return MYSLAM_ERR_INVALID;
    char magic[8]; uint64_t n = 0; uint32_t nl = 0;
    std::vector<myslam_calc_layer> L(kDefaultLayers, kDefaultLayers + 10);
    bool ok = fread(magic, 1, 8, f) == 8;
    if (ok && memcmp(magic, "CALCW2\0\0", 8) == 0) {
        ok = fread(&nl, 4, 1, f) == 1 && nl >= 1 && nl <= 64;
        if (ok) { L.resize(nl); ok = fread(L.data(), sizeof(myslam_calc_layer), nl, f) == nl; }
    } else if (!ok || memcmp(magic, "CALCW1\0\0", 8) != 0) {
        ok = false;
    }
    ok = ok && fread(&n, 8, 1, f) == 1 && n > 0 && n < (1u << 28);
    std::vector<float> w;
    if (ok) { w.resize(n); ok = fread(w.data(), sizeof(float), n, f) == n; }
    fclose(f);
    if (!ok) return MYSLAM_ERR_INVALID;
    return lcd_create(out, L.data(), (int)L.size(), w.data(), n);
}

int myslam_lcd_destroy(myslam_lcd* h) {
    if (!h) return MYSLAM_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    h->free_all();
    delete h;
    return MYSLAM_OK;
}

int myslam_lcd_set_stream(myslam_lcd* h, void* s) {
    if (!h) return MYSLAM_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    h->stream = (hipStream_t)s;
    return MYSLAM_OK;
}

int myslam_lcd_set_option(myslam_lcd* h, int option, int value) {
    if (!h) return MYSLAM_ERR_INVALID;
    if (option == MYSLAM_LCD_OPT_GENERIC_KERNELS) { h->forceGeneric = value != 0; return MYSLAM_OK; }
    if (option == MYSLAM_LCD_OPT_CONV2_BF16X6) { h->forceBf16 = value != 0; return MYSLAM_OK; }
    if (option == MYSLAM_LCD_OPT_SKIP_KERNELS) { h->skipMask = value & 31; return MYSLAM_OK; }
    return MYSLAM_ERR_INVALID;
}

int myslam_lcd_conv2_products(const myslam_lcd* h) {
    if (!h) return MYSLAM_ERR_INVALID;
    if (!h->fused.ok || h->forceGeneric) return 0;
    return h->f16_family() ? 3 : 6;
}
int myslam_lcd_uses_fused_kernels(const myslam_lcd* h) { return h ? (h->fused.ok && !h->forceGeneric ? 1 : 0) : MYSLAM_ERR_INVALID; }

float myslam_lcd_score(const float* d1, const float* d2) {      // deeplcd.cpp:35-39 (host: 1064 FMAs)
    float s = 0;
    for (int i = 0; i < MYSLAM_LCD_DIM; i++) s += d1[i] * d2[i];
    return s;
}

int myslam_lcd_describe_batch(myslam_lcd* h, uint8_t* d_imgs, int batch, int rows, int cols, int step, size_t img_stride,
                              int blur_in_place, float* d_descr) {
    if (!h) return MYSLAM_ERR_INVALID;
    return h->describe(d_imgs, batch, rows, cols, step, img_stride, blur_in_place, d_descr);
}

static int lcd_stage(myslam_lcd* h, size_t bytes) {
    if (bytes > h->stageBytes) { int rc = lcd_alloc(h->d_stageImg, bytes); if (rc) return rc; h->stageBytes = bytes; }
    if (!h->d_stageOut) { int rc = lcd_alloc(h->d_stageOut, (size_t)MYSLAM_LCD_DIM); if (rc) return rc; }
    if (!h->h_stageOut) MYSLAM_HIP_CHECK(hipHostMalloc((void**)&h->h_stageOut, sizeof(float) * MYSLAM_LCD_DIM));
    return MYSLAM_OK;
}

int myslam_lcd_calc_descr_original_img(myslam_lcd* h, uint8_t* img, int rows, int cols, int step, int blur_in_place, float* descr) {
    if (!h || !img || !descr || rows <= 0 || cols <= 0 || step < cols) return MYSLAM_ERR_INVALID;   // reference asserts !empty
    int rc = lcd_stage(h, (size_t)rows * step);
    if (rc) return rc;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageImg, img, (size_t)rows * step, hipMemcpyHostToDevice, h->stream));
    if ((rc = h->describe(h->d_stageImg, 1, rows, cols, step, (size_t)rows * step, blur_in_place, h->d_stageOut))) return rc;
    if (blur_in_place) MYSLAM_HIP_CHECK(hipMemcpyAsync(img, h->d_stageImg, (size_t)rows * step, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->h_stageOut, h->d_stageOut, sizeof(float) * MYSLAM_LCD_DIM, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    memcpy(descr, h->h_stageOut, sizeof(float) * MYSLAM_LCD_DIM);
    return MYSLAM_OK;
}

int myslam_lcd_calc_descr(myslam_lcd* h, const uint8_t* img, int step, float* descr) {
    if (!h || !img || !descr || step < IN_W) return MYSLAM_ERR_INVALID;
    int rc = lcd_stage(h, (size_t)IN_H * step);
    if (rc) return rc;
    if ((rc = h->ensure_batch(1, IN_H, IN_W))) return rc;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageImg, img, (size_t)IN_H * step, hipMemcpyHostToDevice, h->stream));
    // 160x120 -> 160x120 resize is the identity in this arithmetic (weights 2048/0); reuse the input kernel
    hipLaunchKernelGGL(k_lcd_input, dim3((IN_H * IN_W + 255) / 256, 1), dim3(256), 0, h->stream, h->d_stageImg, IN_W, IN_H, step,
                       (size_t)IN_H * step, h->d_xofs, h->d_xa, h->d_yofs, h->d_yb, h->d_in);
    if ((rc = lcd_forward(h, 1, h->d_stageOut))) return rc;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->h_stageOut, h->d_stageOut, sizeof(float) * MYSLAM_LCD_DIM, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    memcpy(descr, h->h_stageOut, sizeof(float) * MYSLAM_LCD_DIM);
    return MYSLAM_OK;
}

// stage taps of the SURVEY A.6 list (whatever kernels run it): 0 = conv1+relu [62*82][64], 1 = pool1+lrn [31*41][64],
// 2 = conv2+relu [32*42][128], 3 = pool2+lrn [16*21][128], 4 = descriptor [1064].  Taps 1-3 come from the fused kernels' buffers when the
// fused path is active; tap 0 (never materialised there) and every tap of a generic-path handle come from the generic layer kernels.
int myslam_lcd_debug_forward(myslam_lcd* h, const float* in, float* out_stage, int stage, size_t cap_floats) {
    if (!h || !in || !out_stage || stage < 0 || stage > 4) return MYSLAM_ERR_INVALID;
    int rc = h->ensure_batch(1, h->rows ? h->rows : IN_H, h->cols ? h->cols : IN_W);
    if (rc) return rc;
    if ((rc = lcd_stage(h, 16))) return rc;
    MYSLAM_HIP_CHECK(hipMemcpy2DAsync(h->d_in + IN_PAD * IN_PW + IN_PAD, IN_PW * sizeof(float), in, IN_W * sizeof(float), IN_W * sizeof(float), IN_H,
                                      hipMemcpyHostToDevice, h->stream));
    const bool fusedNow = h->fused.ok && !h->forceGeneric;
    const float* src = nullptr; size_t n = 0;
    if (stage == 4) {
        if ((rc = lcd_forward(h, 1, h->d_stageOut))) return rc;
        src = h->d_stageOut; n = MYSLAM_LCD_DIM;
    } else if (fusedNow && stage >= 1) {
        if ((rc = lcd_forward(h, 1, h->d_stageOut))) return rc;
        const float* bufs[4] = {nullptr, h->d_p1, h->d_a2, h->d_p2};
        const size_t sizes[4] = {0, (size_t)HP1 * WP1 * C1, (size_t)H2 * W2 * C2, (size_t)HP2 * WP2 * C2};
        src = bufs[stage]; n = sizes[stage];
    } else {
        // layer index whose output is the tap: the stage-th "block end" (a conv followed by its ReLU / a pool followed by its LRN)
        int idx = -1, seen = -1;
        for (size_t i = 0; i < h->layers.size(); i++) {
            const int t = h->layers[i].type;
            if (t == MYSLAM_CALC_CONV || t == MYSLAM_CALC_POOL_MAX) seen++;
            if (seen == stage) idx = (int)i;
            if (seen > stage) break;
        }
        if (idx < 0) return MYSLAM_ERR_INVALID;
        float* tap = nullptr;
        if ((rc = h->forward_generic(1, h->d_stageOut, idx, &tap))) return rc;
        src = tap; n = (size_t)h->shapes[idx].C * h->shapes[idx].H * h->shapes[idx].W;
    }
    if (cap_floats < n) return MYSLAM_ERR_CAPACITY;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(out_stage, src, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (fusedNow && stage == 1 && h->f16_family()) {               // the f16 kernels keep this map as two f16 planes: a = h + m' 2^-11
        std::vector<unsigned short> raw(2 * n);
        memcpy(raw.data(), out_stage, n * sizeof(float));
        for (size_t i = 0; i < n; i++) {                               // element i = (pixel, channel c): halves 32 (c / 16) + c % 16 (h) and + 16 (m') of the pixel's 128
            _Float16 hh, mm;
            const size_t o = (i >> 6) * 128 + ((i & 63) >> 4) * 32 + (i & 15);
            memcpy(&hh, &raw[o], 2); memcpy(&mm, &raw[o + 16], 2);
            out_stage[i] = (float)hh + (float)mm * (1.0f / 2048.f);
        }
    }
    return MYSLAM_OK;
}

}  // extern "C"
