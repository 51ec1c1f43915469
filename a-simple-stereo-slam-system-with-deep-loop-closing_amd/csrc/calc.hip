// calc.hip — DeepLCD / CALC descriptor network on gfx950 (replaces class DeepLCD, reference
// include/myslam/deeplcd.h:21-48 and src/deeplcd.cpp:10-91; architecture SURVEY.md Appendix A.6).
//
//   u8 image --blur 7x7 (sigma<=0 table, optionally in place: reference quirk)--> resize 160x120 --> /255
//   conv1 64@5x5 s2 p4 + ReLU -> maxpool 3x3 s2 (ceil) -> LRN(5,1e-4,.75)        [VALU, lane = channel]
//   conv2 128@4x4 s1 p2 + ReLU                                                   [fp32 MFMA implicit GEMM]
//   maxpool -> LRN -> conv3 4@3x3 + ReLU -> flatten (Caffe NCHW order) -> L2 normalise
//
// Activations are NHWC (channel-last) in HBM so that a wave's 64 lanes map to 64 channels (coalesced
// 256-byte rows) and conv2's im2col rows are contiguous 64-byte channel runs.  All arithmetic is f32
// (the reference runs Caffe in f32); conv2 uses v_mfma_f32_32x32x2_f32, which is an exact f32 FMA chain.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "orb_plan.h"

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

// ---- conv1 + ReLU: lane = output channel, one wave walks 8 output pixels ----
__global__ __launch_bounds__(256) void k_conv1(const float* __restrict__ in, const float* __restrict__ w1t /*[25][64]*/,
                                               const float* __restrict__ b1, float* __restrict__ out /*[H1*W1][64]*/) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w[25];
#pragma unroll
    for (int k = 0; k < 25; k++) w[k] = w1t[k * 64 + lane];
    const float bias = b1[lane];
    const float* I = in + (size_t)b * IN_PLANE;
    const int p0 = (blockIdx.x * 4 + wave) * 8;
    for (int p = p0; p < min(p0 + 8, H1 * W1); p++) {
        const int oy = p / W1, ox = p - oy * W1;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 5; ky++) {
            const int iy = oy * 2 + ky - 4;
#pragma unroll
            for (int kx = 0; kx < 5; kx++) {
                const int ix = ox * 2 + kx - 4;
                const float v = I[(iy + IN_PAD) * IN_PW + ix + IN_PAD];
                acc += w[ky * 5 + kx] * v;
            }
        }
        out[((size_t)b * H1 * W1 + p) * 64 + lane] = fmaxf(acc + bias, 0.f);
    }
}

// LRN denominator scale^-0.75 = rsqrt(scale) * sqrt(rsqrt(scale)) on the hardware rsq / sqrt units (scale >= 1; within 3 ulp of
// powf, two orders of magnitude inside the descriptor tolerance) instead of the ~150-instruction powf expansion
__device__ __forceinline__ float lrn_pow_m075(float scale) {
    const float r = __builtin_amdgcn_rsqf(scale);
    return r * __builtin_amdgcn_sqrtf(r);
}

// ---- max-pool 3x3 s2 (Caffe ceil mode, clipped windows) + LRN across channels; one wave per output pixel ----
template <int C>
__global__ __launch_bounds__(256) void k_pool_lrn(const float* __restrict__ in, int H, int W, int OH, int OW,
                                                  float* __restrict__ out) {
    constexpr int PER = C / 64;
    __shared__ float s_v[4][C + 4];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wave;
    if (p < OH * OW) {
        const int oy = p / OW, ox = p - oy * OW;
        const int y0 = oy * 2, x0 = ox * 2, y1 = min(y0 + 3, H), x1 = min(x0 + 3, W);
        const float* I = in + (size_t)b * H * W * C;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int c = lane + 64 * q;
            float m = -INFINITY;
            for (int y = y0; y < y1; y++)
                for (int x = x0; x < x1; x++) m = fmaxf(m, I[((size_t)y * W + x) * C + c]);
            s_v[wave][c + 2] = m;
        }
        if (lane < 2) { s_v[wave][lane] = 0.f; s_v[wave][C + 2 + lane] = 0.f; }
    }
    __syncthreads();
    if (p < OH * OW) {
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int c = lane + 64 * q;
            const float* v = &s_v[wave][c];          // v[0..4] = channels c-2..c+2 (zero padded)
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 5; j++) ss += v[j] * v[j];
            const float scale = 1.f + (1e-4f / 5.f) * ss;
            out[((size_t)b * OH * OW + p) * C + c] = v[2] * lrn_pow_m075(scale);
        }
    }
}

// ---- the same operator for the 128-channel conv2 map, 2 x 2 pooled pixels per wave ----
// Lane l owns channels (2l, 2l+1): one 8-byte load per input pixel and lane (a coalesced 512-byte row per wave), all 25 loads of the
// 5 x 5 input block issued before the first use (coordinates clamped instead of clipped: a duplicate does not change a maximum),
// every input pixel read 1.56 instead of 2.25 times, the LRN neighbours over lane shuffles instead of LDS.  Same maxima, same
// LRN summation order (channels c-2 .. c+2) as k_pool_lrn<128>: identical results.
__global__ __launch_bounds__(256) void k_pool_lrn128_2x2(const float* __restrict__ in, int H, int W, int OH, int OW, float* __restrict__ out) {
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
        o.x = m.x * lrn_pow_m075(1.f + (1e-4f / 5.f) * sa);
        o.y = m.y * lrn_pow_m075(1.f + (1e-4f / 5.f) * sb);
        reinterpret_cast<float2*>(out + ((size_t)b * OH * OW + (size_t)py * OW + px) * 128)[lane] = o;
    };
    const bool row1 = oy + 1 < OH, col1 = ox + 1 < OW;           // wave-uniform
    lrn_store(pmax(0, 0), oy, ox);
    if (col1) lrn_store(pmax(0, 2), oy, ox + 1);
    if (row1) lrn_store(pmax(2, 0), oy + 1, ox);
    if (row1 && col1) lrn_store(pmax(2, 2), oy + 1, ox + 1);
}

// ---- conv1 + ReLU + max-pool + LRN fused: one wave per POOLED pixel, lane = channel ----
// The 3x3 (clipped) pool window needs 9 conv1 outputs = a 9x9 input window, which is wave-uniform: it is fetched with
// scalar loads and fed to v_fmac as SGPR operands, so the conv1 activation map (1.3 MB per image) never exists in HBM.
// Same operation order per output as k_conv1 + k_pool_lrn (tap order ky,kx; LRN sum over c-2..c+2): the results agree with
// the unfused pair to the last bit or two (multiply-add contraction); the LRN neighbours come over lane shuffles instead of LDS.
__global__ __launch_bounds__(256) void k_conv1_pool_lrn(const float* __restrict__ in, const float* __restrict__ w1t /*[25][64]*/,
                                                        const float* __restrict__ b1, float* __restrict__ out /*[HP1*WP1][64]*/) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int p = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (p >= HP1 * WP1) return;
    float w[25];
#pragma unroll
    for (int k = 0; k < 25; k++) w[k] = w1t[k * 64 + lane];
    const float bias = b1[lane];
    const int oy = p / WP1, ox = p - oy * WP1;
    const float* I = in + (size_t)b * IN_PLANE + (4 * oy) * IN_PW + 4 * ox;      // window origin (input row 4 oy - 4, col 4 ox - 4)
    float win[9][9];
#pragma unroll
    for (int r = 0; r < 9; r++)
#pragma unroll
        for (int c = 0; c < 9; c++) win[r][c] = I[r * IN_PW + c];
    float m = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 5; ky++)
#pragma unroll
                for (int kx = 0; kx < 5; kx++) acc += w[ky * 5 + kx] * win[2 * dy + ky][2 * dx + kx];
            const bool inside = 2 * oy + dy < H1 && 2 * ox + dx < W1;      // Caffe ceil-mode pooling: clipped window
            m = fmaxf(m, inside ? fmaxf(acc + bias, 0.f) : -INFINITY);
        }
    // LRN(5, 1e-4, 0.75) across the 64 channels (zero padded)
    const float um1 = __shfl_up(m, 1, 64), um2 = __shfl_up(m, 2, 64), dp1 = __shfl_down(m, 1, 64), dp2 = __shfl_down(m, 2, 64);
    const float v0 = lane >= 2 ? um2 : 0.f, v1 = lane >= 1 ? um1 : 0.f, v3 = lane <= 62 ? dp1 : 0.f, v4 = lane <= 61 ? dp2 : 0.f;
    float ss = 0.f;
    ss += v0 * v0; ss += v1 * v1; ss += m * m; ss += v3 * v3; ss += v4 * v4;
    const float scale = 1.f + (1e-4f / 5.f) * ss;
    out[((size_t)b * HP1 * WP1 + p) * 64 + lane] = m * lrn_pow_m075(scale);
}

// ---- the same fused operator, 2 x 2 POOLED pixels per wave ----
// Neighbouring 3x3 / stride-2 pool windows share a row and a column of conv1 outputs: one wave per pooled pixel computes every
// conv1 output 2.25 times.  Here a wave owns a 2 x 2 block of pooled pixels = 5 x 5 conv1 outputs (1.56 per pooled pixel
// instead of 2.25: -31 % FMAs), walks the conv rows top to bottom (5 input rows x 13 columns of wave-uniform scalars per conv
// row) and folds each output into the maxima of the pooled pixels it belongs to.  Per output the tap order (ky, kx), bias,
// ReLU, clipping and the LRN are those of k_conv1_pool_lrn (results agree to the last bit or two: the compiler contracts the
// multiply-add chains of the two kernels differently).
constexpr int HT1 = (HP1 + 1) / 2, WT1 = (WP1 + 1) / 2;           // 2 x 2 tiles of the pooled map
__global__ __launch_bounds__(256) void k_conv1_pool_lrn2(const float* __restrict__ in, const float* __restrict__ w1t /*[25][64]*/,
                                                         const float* __restrict__ b1, float* __restrict__ out /*[HP1*WP1][64]*/) {
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
#pragma unroll
    for (int cy = 0; cy < 5; cy++) {                                // conv row 2 oy + cy
        if (cy >= 3 && !row1) break;                                // rows 3, 4 only feed the second pooled row (and would read past the padded plane)
        float win[5][13];
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 13; c++) win[r][c] = I[(2 * cy + r) * IN_PW + c];
#pragma unroll
        for (int cx = 0; cx < 5; cx++) {                            // conv column 2 ox + cx
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 5; ky++)
#pragma unroll
                for (int kx = 0; kx < 5; kx++) acc += w[ky * 5 + kx] * win[ky][2 * cx + kx];
            const bool inside = 2 * oy + cy < H1 && 2 * ox + cx < W1;      // Caffe ceil-mode pooling: clipped windows
            const float v = inside ? fmaxf(acc + bias, 0.f) : -INFINITY;
            if (cy <= 2 && cx <= 2) m00 = fmaxf(m00, v);
            if (cy <= 2 && cx >= 2) m01 = fmaxf(m01, v);
            if (cy >= 2 && cx <= 2) m10 = fmaxf(m10, v);
            if (cy >= 2 && cx >= 2) m11 = fmaxf(m11, v);
        }
    }
    auto lrn_store = [&](float m, int py, int px) {              // LRN(5, 1e-4, 0.75) across the 64 channels (zero padded)
        const float um1 = __shfl_up(m, 1, 64), um2 = __shfl_up(m, 2, 64), dp1 = __shfl_down(m, 1, 64), dp2 = __shfl_down(m, 2, 64);
        const float v0 = lane >= 2 ? um2 : 0.f, v1 = lane >= 1 ? um1 : 0.f, v3 = lane <= 62 ? dp1 : 0.f, v4 = lane <= 61 ? dp2 : 0.f;
        float ss = 0.f;
        ss += v0 * v0; ss += v1 * v1; ss += m * m; ss += v3 * v3; ss += v4 * v4;
        const float scale = 1.f + (1e-4f / 5.f) * ss;
        out[((size_t)b * HP1 * WP1 + py * WP1 + px) * 64 + lane] = m * lrn_pow_m075(scale);
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

// ---- conv2 + ReLU as an fp32 MFMA implicit GEMM ----
// C[M = batch*1344][N = 128] = A[M][K = 1024] * Wt[K][N];  k = (ky*4+kx)*64 + ic  (channel runs contiguous in NHWC)
// 4 waves as 2x2, each wave TI x TJ MFMA 32x32 tiles -> block tile (64 TI) x (64 TJ); BK k per stage, register-staged double
// buffering through LDS.
//   <2, 2, 16>  128 x 128 tile, 179 registers: the stand-alone configuration (68 % of the fp32-MFMA peak)
//   <1, 1, 16>   64 x 64 tile, <= 80 registers and 17 KB of LDS: small enough to sit on a CU NEXT TO six waves per SIMD of the
//               VALU-bound FAST kernel, so the matrix cores work while the vector ALUs are saturated
constexpr int CV_BN = 128;

template <int TI, int TJ, int BK>
__global__ __launch_bounds__(256) void k_conv2_mfma(const float* __restrict__ in /*[B][31*41][64]*/,
                                                    const float* __restrict__ wt /*[1024][128]*/, const float* __restrict__ b2,
                                                    float* __restrict__ out /*[B*1344][128]*/, int Mtotal) {
    constexpr int BM = 64 * TI, BN = 64 * TJ, PA = BM + 4, PB = BN + 4;
    constexpr int AK = BM * BK / 256;                  // consecutive k per thread of the A slab (2, 4 or 8)
    constexpr int BPT = BK * BN / 256;                 // consecutive n per thread of the B slab (4 or 8)
    static_assert(CV_BN % BN == 0 && (AK == 2 || AK == 4 || AK == 8) && (BPT == 4 || BPT == 8) && 64 % BK == 0, "unsupported conv2 tiling");
    __shared__ __attribute__((aligned(16))) float s_a[2][BK * PA];
    __shared__ __attribute__((aligned(16))) float s_b[2][BK * PB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * (32 * TI), wn = (wave & 1) * (32 * TJ);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // A staging role: thread -> (row m, AK consecutive k)
    constexpr int ATPR = BK / AK;                      // threads per A row
    const int am = t / ATPR, ak = (t % ATPR) * AK;
    const int gm = m0 + am;
    const bool mvalid = gm < Mtotal;
    const int img = mvalid ? gm / M2 : 0;
    const int pix = mvalid ? gm - img * M2 : 0;
    const int oy = pix / W2, ox = pix - oy * W2;
    const float* inb = in + (size_t)img * HP1 * WP1 * 64;
    // B staging role: thread -> (k row, BPT consecutive n)
    constexpr int BTPR = BN / BPT;
    const int bk = t / BTPR, bn = (t % BTPR) * BPT;

    float ra[AK], rb[BPT];
    auto load_stage = [&](int s) {
        constexpr int SPT = 64 / BK;                   // stages per filter tap
        const int tap = s / SPT, ic0 = (s % SPT) * BK;
        const int iy = oy + (tap >> 2) - 2, ix = ox + (tap & 3) - 2;
        if (mvalid && iy >= 0 && iy < HP1 && ix >= 0 && ix < WP1) {
            const float* src = inb + ((size_t)iy * WP1 + ix) * 64 + ic0 + ak;
            if constexpr (AK == 2) { const float2 v = *reinterpret_cast<const float2*>(src); ra[0] = v.x; ra[1] = v.y; }
            else {
#pragma unroll
                for (int q = 0; q < AK / 4; q++) { const float4 v = reinterpret_cast<const float4*>(src)[q]; ra[4 * q] = v.x; ra[4 * q + 1] = v.y; ra[4 * q + 2] = v.z; ra[4 * q + 3] = v.w; }
            }
        } else {
#pragma unroll
            for (int q = 0; q < AK; q++) ra[q] = 0.f;
        }
        const float4* wsrc = reinterpret_cast<const float4*>(wt + (size_t)(s * BK + bk) * CV_BN + n0 + bn);
#pragma unroll
        for (int q = 0; q < BPT / 4; q++) { const float4 v = wsrc[q]; rb[4 * q] = v.x; rb[4 * q + 1] = v.y; rb[4 * q + 2] = v.z; rb[4 * q + 3] = v.w; }
    };
    auto store_stage = [&](int buf) {
        float* a = s_a[buf];
#pragma unroll
        for (int q = 0; q < AK; q++) a[(ak + q) * PA + am] = ra[q];
        float4* bdst = reinterpret_cast<float4*>(&s_b[buf][bk * PB + bn]);
#pragma unroll
        for (int q = 0; q < BPT / 4; q++) bdst[q] = make_float4(rb[4 * q], rb[4 * q + 1], rb[4 * q + 2], rb[4 * q + 3]);
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    load_stage(0);
    store_stage(0);
    __syncthreads();
    constexpr int NSTAGE = K2 / BK;
    const int lr = lane & 31, lk = lane >> 5;
    for (int s = 0; s < NSTAGE; s++) {
        const int buf = s & 1;
        if (s + 1 < NSTAGE) load_stage(s + 1);
        const float* a = s_a[buf];
        const float* bb = s_b[buf];
#pragma unroll
        for (int kq = 0; kq < BK / 2; kq++) {
            const int k = 2 * kq + lk;
            float av[TI], bv[TJ];
#pragma unroll
            for (int i = 0; i < TI; i++) av[i] = a[k * PA + wm + 32 * i + lr];
#pragma unroll
            for (int j = 0; j < TJ; j++) bv[j] = bb[k * PB + wn + 32 * j + lr];
#pragma unroll
            for (int i = 0; i < TI; i++)
#pragma unroll
                for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < NSTAGE) store_stage(buf ^ 1);
        __syncthreads();
    }
    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TJ; j++) {
        const int n = n0 + wn + j * 32 + lr;
        const float bias = b2[n];
#pragma unroll
        for (int i = 0; i < TI; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m < Mtotal) out[(size_t)m * CV_BN + n] = fmaxf(acc[i][j][r] + bias, 0.f);
            }
    }
}

// ---- conv2 on the bf16 matrix cores with fp32 accuracy ----
// The fp32-input MFMA above runs at the f32 VECTOR rate — in practice it competes with the VALU-bound ORB kernels of the other
// stream instead of running under them (measured: no gain from co-residency).  The bf16 matrix pipe is 16x
// faster and separate.  Every f32 operand is split exactly into three bf16 pieces a = h + m + l (8 + 8 + 8 significand bits,
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
                                                      const float* __restrict__ b2, float* __restrict__ out /*[B*1344][128]*/, int Mtotal) {
    constexpr int BM = 128, BN = 128;
    __shared__ uint4 s_a[2][3][BM * 2];                // [piece][row m][k half]: 8 bf16 per uint4
    __shared__ uint4 s_b[2][3][BN * 2];
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
        s_a[buf][0][t] = h; s_a[buf][1][t] = m; s_a[buf][2][t] = l;       // index (m * 2 + k half) == t
        s_b[buf][0][t] = rb0; s_b[buf][1][t] = rb1; s_b[buf][2][t] = rb2;
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
            for (int p = 0; p < 3; p++) A[i][p] = __builtin_bit_cast(cv_bf16x8, s_a[buf][p][(wm + 32 * i + lr) * 2 + lk]);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) Bm[j][p] = __builtin_bit_cast(cv_bf16x8, s_b[buf][p][(wn + 32 * j + lr) * 2 + lk]);
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
                if (m < Mtotal) out[(size_t)m * CV_BN + n] = fmaxf(acc[i][j][r] + bias, 0.f);
            }
    }
}

#undef CV2_LOAD_STAGE

// ---- conv3 + ReLU + flatten (NCHW order) + L2 normalise; one 1024-thread block per image ----
// 16 waves share the 266 output pixels; each lane keeps its 18 weight quadruples (k = lane + 64 j) in registers.
constexpr int CV3_T = 1024, CV3_NW = CV3_T / 64;
__global__ __launch_bounds__(CV3_T) void k_conv3_norm(const float* __restrict__ in /*[B][16*21][128]*/,
                                                      const float* __restrict__ w3t /*[1152][4]*/, const float* __restrict__ b3,
                                                      float* __restrict__ out /*[B][1064]*/, int relu) {
    __shared__ float s_o[C3 * H3 * W3];
    __shared__ float s_red[CV3_NW];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* I = in + (size_t)b * HP2 * WP2 * 128;
    const float4* Wt = reinterpret_cast<const float4*>(w3t);
    float4 w[18];
#pragma unroll
    for (int j = 0; j < 18; j++) w[j] = Wt[lane + 64 * j];
    for (int p = wave; p < H3 * W3; p += CV3_NW) {
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
            for (int c = 0; c < 4; c++) s_o[c * H3 * W3 + p] = relu ? fmaxf(r[c], 0.f) : r[c];
        }
    }
    __syncthreads();
    float ss = 0.f;
    for (int i = threadIdx.x; i < MYSLAM_LCD_DIM; i += CV3_T) ss += s_o[i] * s_o[i];
    ss = wave_sum_lane63_f32(ss);
    if (lane == 63) s_red[wave] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < CV3_NW; i++) tot += s_red[i];
    const float nrm = sqrtf(tot);                                             // deeplcd.cpp:88
    for (int i = threadIdx.x; i < MYSLAM_LCD_DIM; i += CV3_T) out[(size_t)b * MYSLAM_LCD_DIM + i] = s_o[i] / nrm;
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

}  // namespace myslam_hip

using namespace myslam_hip;

struct myslam_lcd {
    hipStream_t stream = nullptr;
    float *d_w1t = nullptr, *d_b1 = nullptr, *d_w2t = nullptr, *d_b2 = nullptr, *d_w3t = nullptr, *d_b3 = nullptr;
    uint4* d_w2s = nullptr;            // conv2 weights split into three bf16 pieces, [stage][piece][n][k half] x 8 bf16
    int relu3 = 1;
    // resize tables for the current source size
    int rows = 0, cols = 0;
    int32_t *d_xofs = nullptr, *d_yofs = nullptr; int16_t *d_xa = nullptr, *d_yb = nullptr;
    // batch buffers
    int batchCap = 0; size_t blurBytes = 0; int blurPitch = 0;
    uint8_t* d_blur = nullptr;
    float *d_in = nullptr, *d_a1 = nullptr, *d_p1 = nullptr, *d_a2 = nullptr, *d_p2 = nullptr;
    // host-entry staging
    uint8_t* d_stageImg = nullptr; size_t stageBytes = 0; float* d_stageOut = nullptr;

    int ensure_tables(int r, int c);
    int ensure_batch(int batch, int r, int c);
    int forward(int batch);     // d_in -> d_out
    int describe(uint8_t* d_imgs, int batch, int r, int c, int step, size_t stride, int blur_in_place, float* d_out);
};

template <typename T>
static int lcd_alloc(T*& p, size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    if (!n) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipMalloc((void**)&p, n * sizeof(T)));
    return MYSLAM_OK;
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
    MYSLAM_HIP_CHECK(hipMemcpy(d_xofs, xo.data(), xo.size() * 4, hipMemcpyHostToDevice));
    MYSLAM_HIP_CHECK(hipMemcpy(d_xa, xa.data(), xa.size() * 2, hipMemcpyHostToDevice));
    MYSLAM_HIP_CHECK(hipMemcpy(d_yofs, yo.data(), yo.size() * 4, hipMemcpyHostToDevice));
    MYSLAM_HIP_CHECK(hipMemcpy(d_yb, yb.data(), yb.size() * 2, hipMemcpyHostToDevice));
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
    if ((rc = lcd_alloc(d_a1, (size_t)batch * H1 * W1 * C1))) return rc;
    if ((rc = lcd_alloc(d_p1, (size_t)batch * HP1 * WP1 * C1))) return rc;
    if ((rc = lcd_alloc(d_a2, (size_t)batch * H2 * W2 * C2))) return rc;
    if ((rc = lcd_alloc(d_p2, (size_t)batch * HP2 * WP2 * C2))) return rc;
    batchCap = batch;
    return MYSLAM_OK;
}

static int lcd_forward(myslam_lcd* h, int batch, float* d_out) {
    hipStream_t s = h->stream;
    {
        ScopedProf sp(P_CONV1, s);
        static const char* env = getenv("MYSLAM_CONV1_V");        // tuning aid: 1 = unfused conv1 -> pool/LRN pair
        if (env && atoi(env) == 1) {
            hipLaunchKernelGGL(k_conv1, dim3((H1 * W1 + 31) / 32, batch), dim3(256), 0, s, h->d_in, h->d_w1t, h->d_b1, h->d_a1);
            hipLaunchKernelGGL((k_pool_lrn<C1>), dim3((HP1 * WP1 + 3) / 4, batch), dim3(256), 0, s, h->d_a1, H1, W1, HP1, WP1, h->d_p1);
        } else if (env && atoi(env) == 2) {                       // one wave per pooled pixel
            hipLaunchKernelGGL(k_conv1_pool_lrn, dim3((HP1 * WP1 + 3) / 4, batch), dim3(256), 0, s, h->d_in, h->d_w1t, h->d_b1, h->d_p1);
        } else {
            hipLaunchKernelGGL(k_conv1_pool_lrn2, dim3((HT1 * WT1 + 3) / 4, batch), dim3(256), 0, s, h->d_in, h->d_w1t, h->d_b1, h->d_p1);
        }
    }
    {
        ScopedProf sp(P_CONV2, s);
        const int Mtotal = batch * M2;
        static const int lite = [] { const char* e = getenv("MYSLAM_CONV2_V"); return e ? atoi(e) : 3; }();     // 0/1/2: f32-input MFMA tilings
        if (lite == 3)
            hipLaunchKernelGGL(k_conv2_bf16x6, dim3((Mtotal + 127) / 128), dim3(256), 0, s, h->d_p1, h->d_w2s, h->d_b2, h->d_a2, Mtotal);
        else if (lite == 1)
            hipLaunchKernelGGL((k_conv2_mfma<1, 1, 16>), dim3((Mtotal + 63) / 64, 2), dim3(256), 0, s, h->d_p1, h->d_w2t, h->d_b2, h->d_a2, Mtotal);
        else if (lite == 2)
            hipLaunchKernelGGL((k_conv2_mfma<1, 2, 8>), dim3((Mtotal + 63) / 64), dim3(256), 0, s, h->d_p1, h->d_w2t, h->d_b2, h->d_a2, Mtotal);
        else
            hipLaunchKernelGGL((k_conv2_mfma<2, 2, 16>), dim3((Mtotal + 127) / 128), dim3(256), 0, s, h->d_p1, h->d_w2t, h->d_b2, h->d_a2, Mtotal);
    }
    {
        ScopedProf sp(P_CONV3, s);
        static const char* envp = getenv("MYSLAM_POOL2_V");         // tuning aid: 1 = one wave per pooled pixel, LRN through LDS
        if (envp && atoi(envp) == 1)
            hipLaunchKernelGGL((k_pool_lrn<C2>), dim3((HP2 * WP2 + 3) / 4, batch), dim3(256), 0, s, h->d_a2, H2, W2, HP2, WP2, h->d_p2);
        else
            hipLaunchKernelGGL(k_pool_lrn128_2x2, dim3((((HP2 + 1) / 2) * ((WP2 + 1) / 2) + 3) / 4, batch), dim3(256), 0, s, h->d_a2, H2, W2, HP2, WP2, h->d_p2);
        hipLaunchKernelGGL(k_conv3_norm, dim3(batch), dim3(CV3_T), 0, s, h->d_p2, h->d_w3t, h->d_b3, d_out, h->relu3);
    }
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

int myslam_lcd::describe(uint8_t* d_imgs, int batch, int r, int c, int step, size_t stride, int blur_in_place, float* d_out) {
    if (!d_imgs || batch <= 0 || r <= 0 || c <= 0 || step < c || !d_out) return MYSLAM_ERR_INVALID;
    int rc = ensure_batch(batch, r, c);
    if (rc) return rc;
    {
        ScopedProf sp(P_LCD_PRE, stream);
        static const int fused = [] { const char* e = getenv("MYSLAM_LCD_PRE_V"); return e ? atoi(e) != 1 : 1; }();     // tuning aid: 1 = two-pass blur + resize
        if (!blur_in_place && fused) {
            LcdTaps tp;
            gauss_q8(1, tp.q);
            hipLaunchKernelGGL(k_lcd_input_fused, dim3((IN_H * IN_W + 255) / 256, batch), dim3(256), 0, stream, d_imgs, c, r, step, stride,
                               d_xofs, d_xa, d_yofs, d_yb, tp, d_in);
        } else {
            BlurArgs a;                                   // GaussianBlur(img, img, Size(7,7), 0)  deeplcd.cpp:46
            a.src = d_imgs; a.dst = d_blur; a.w = c; a.h = r; a.spitch = step; a.dpitch = blurPitch; a.sstride = stride; a.dstride = blurBytes;
            gauss_q8(1, a.q);
            launch_blur(a, batch, stream);
            if (blur_in_place)                            // the reference mutates the caller's pixels
                for (int b = 0; b < batch; b++)
                    MYSLAM_HIP_CHECK(hipMemcpy2DAsync(d_imgs + (size_t)b * stride, step, d_blur + (size_t)b * blurBytes, blurPitch, c, r,
                                                      hipMemcpyDeviceToDevice, stream));
            hipLaunchKernelGGL(k_lcd_input, dim3((IN_H * IN_W + 255) / 256, batch), dim3(256), 0, stream, d_blur, c, r, blurPitch, blurBytes,
                               d_xofs, d_xa, d_yofs, d_yb, d_in);       // cv::resize(.., Size(160,120))  deeplcd.cpp:48-50
        }
    }
    return lcd_forward(this, batch, d_out);
}

extern "C" {

size_t myslam_lcd_nweights(void) { return NWEIGHTS; }

int myslam_lcd_create(myslam_lcd** out, const float* weights, size_t nweights) {
    if (!out || !weights || nweights != NWEIGHTS) return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    const float* w1 = weights;            const float* b1 = w1 + 64 * 25;
    const float* w2 = b1 + 64;            const float* b2 = w2 + 128 * 64 * 16;
    const float* w3 = b2 + 128;           const float* b3 = w3 + 4 * 128 * 9;
    std::vector<float> w1t(25 * 64), w2t((size_t)K2 * 128), w3t(1152 * 4);
    for (int oc = 0; oc < 64; oc++) for (int k = 0; k < 25; k++) w1t[k * 64 + oc] = w1[oc * 25 + k];
    for (int oc = 0; oc < 128; oc++)
        for (int ic = 0; ic < 64; ic++)
            for (int t = 0; t < 16; t++) w2t[((size_t)t * 64 + ic) * 128 + oc] = w2[((size_t)oc * 64 + ic) * 16 + t];
    for (int oc = 0; oc < 4; oc++)
        for (int ic = 0; ic < 128; ic++)
            for (int t = 0; t < 9; t++) w3t[((size_t)t * 128 + ic) * 4 + oc] = w3[((size_t)oc * 128 + ic) * 9 + t];
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
    myslam_lcd* h = new myslam_lcd();
    auto up = [&](float*& d, const float* src, size_t n) -> int {
        MYSLAM_HIP_CHECK(hipMalloc((void**)&d, n * sizeof(float)));
        MYSLAM_HIP_CHECK(hipMemcpy(d, src, n * sizeof(float), hipMemcpyHostToDevice));
        return MYSLAM_OK;
    };
    int rc;
    if ((rc = up(h->d_w1t, w1t.data(), w1t.size())) || (rc = up(h->d_b1, b1, 64)) || (rc = up(h->d_w2t, w2t.data(), w2t.size())) ||
        (rc = up(h->d_b2, b2, 128)) || (rc = up(h->d_w3t, w3t.data(), w3t.size())) || (rc = up(h->d_b3, b3, 4))) {
        delete h;
        return rc;
    }
    if (hipMalloc((void**)&h->d_w2s, w2s.size() * 2) != hipSuccess ||
        hipMemcpy(h->d_w2s, w2s.data(), w2s.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { delete h; return MYSLAM_ERR_HIP; }
    *out = h;
    return MYSLAM_OK;
}

// own flat model file: magic "CALCW1\0\0", uint64 count, then count f32 (little endian)
int myslam_lcd_create_from_file(myslam_lcd** out, const char* path) {
    if (!out || !path) return MYSLAM_ERR_INVALID;
    FILE* f = fopen(path, "rb");
    if (!f) return MYSLAM_ERR_INVALID;
    char magic[8]; uint64_t n = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "CALCW1\0\0", 8) != 0 || fread(&n, 8, 1, f) != 1 || n != NWEIGHTS) { fclose(f); return MYSLAM_ERR_INVALID; }
    std::vector<float> w(n);
    const size_t got = fread(w.data(), sizeof(float), n, f);
    fclose(f);
    if (got != n) return MYSLAM_ERR_INVALID;
    return myslam_lcd_create(out, w.data(), n);
}

int myslam_lcd_destroy(myslam_lcd* h) {
    if (!h) return MYSLAM_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    void* ptrs[] = {h->d_w2s, h->d_w1t, h->d_b1, h->d_w2t, h->d_b2, h->d_w3t, h->d_b3, h->d_xofs, h->d_yofs, h->d_xa, h->d_yb, h->d_blur,
                    h->d_in, h->d_a1, h->d_p1, h->d_a2, h->d_p2, h->d_stageImg, h->d_stageOut};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete h;
    return MYSLAM_OK;
}

int myslam_lcd_set_stream(myslam_lcd* h, void* s) {
    if (!h) return MYSLAM_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    h->stream = (hipStream_t)s;
    return MYSLAM_OK;
}

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
    return MYSLAM_OK;
}

int myslam_lcd_calc_descr_original_img(myslam_lcd* h, uint8_t* img, int rows, int cols, int step, int blur_in_place, float* descr) {
    if (!h || !img || !descr || rows <= 0 || cols <= 0 || step < cols) return MYSLAM_ERR_INVALID;   // reference asserts !empty
    int rc = lcd_stage(h, (size_t)rows * step);
    if (rc) return rc;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_stageImg, img, (size_t)rows * step, hipMemcpyHostToDevice, h->stream));
    if ((rc = h->describe(h->d_stageImg, 1, rows, cols, step, (size_t)rows * step, blur_in_place, h->d_stageOut))) return rc;
    if (blur_in_place) MYSLAM_HIP_CHECK(hipMemcpyAsync(img, h->d_stageImg, (size_t)rows * step, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(descr, h->d_stageOut, sizeof(float) * MYSLAM_LCD_DIM, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
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
    MYSLAM_HIP_CHECK(hipMemcpyAsync(descr, h->d_stageOut, sizeof(float) * MYSLAM_LCD_DIM, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    return MYSLAM_OK;
}

// stage taps: 0 = conv1+relu [62*82][64], 1 = pool1+lrn [31*41][64], 2 = conv2+relu [32*42][128],
//             3 = pool2+lrn [16*21][128], 4 = descriptor [1064]
int myslam_lcd_debug_forward(myslam_lcd* h, const float* in, float* out_stage, int stage, size_t cap_floats) {
    if (!h || !in || !out_stage || stage < 0 || stage > 4) return MYSLAM_ERR_INVALID;
    int rc = h->ensure_batch(1, h->rows ? h->rows : IN_H, h->cols ? h->cols : IN_W);
    if (rc) return rc;
    if ((rc = lcd_stage(h, 16))) return rc;
    MYSLAM_HIP_CHECK(hipMemcpy2DAsync(h->d_in + IN_PAD * IN_PW + IN_PAD, IN_PW * sizeof(float), in, IN_W * sizeof(float), IN_W * sizeof(float), IN_H,
                                      hipMemcpyHostToDevice, h->stream));
    if ((rc = lcd_forward(h, 1, h->d_stageOut))) return rc;
    if (stage == 0)      // the conv1 activation map is not materialised by the fused kernel: produce the tap on request
        hipLaunchKernelGGL(k_conv1, dim3((H1 * W1 + 31) / 32, 1), dim3(256), 0, h->stream, h->d_in, h->d_w1t, h->d_b1, h->d_a1);
    const float* src[5] = {h->d_a1, h->d_p1, h->d_a2, h->d_p2, h->d_stageOut};
    const size_t n[5] = {(size_t)H1 * W1 * C1, (size_t)HP1 * WP1 * C1, (size_t)H2 * W2 * C2, (size_t)HP2 * WP2 * C2, MYSLAM_LCD_DIM};
    if (cap_floats < n[stage]) return MYSLAM_ERR_CAPACITY;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(out_stage, src[stage], n[stage] * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    return MYSLAM_OK;
}

}  // extern "C"
