// orb_kernels.hip — HIP kernels of the ORB extractor for gfx950 (wave64, LDS-tiled).
//
// Reference behaviour being reproduced (all paths relative to the reference tree):
//   pyramid            src/ORBextractor.cpp:1229-1265   (cv::resize INTER_LINEAR cascade)
//   grid FAST          src/ORBextractor.cpp:814-883     (cv::FAST + 3x3 NMS per 30-px cell, 20 -> 7 fallback)
//   oct-tree           src/ORBextractor.cpp:526-810
//   orientation        src/ORBextractor.cpp:27-55
//   blur + rBRIEF      src/ORBextractor.cpp:58-98, :965-970
//
// Built with -ffp-contract=off: BRIEF sample coordinates and fastAtan2 are float expressions whose
// integer/float results must be bit-identical to a non-contracting CPU evaluation.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "orb_plan.h"

namespace myslam_hip {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ---- block trace (profiling builds only: tools/build_variants.sh orb_kernels.hip bt:-DMYSLAM_BLOCK_TRACE; tools/block_trace_report.py) --------------
// Every block (describe: every work item) that runs on shader engine 0 of XCD 0 leaves one 32-byte record {start, duration | kernel | CU | block id, phase marks, -} in a caller-provided
// buffer: which kernel's blocks are resident on which CU at what time, i.e. the measured form of "where do the idle issue slots sit".  The product
// build carries none of this (no symbol, no instruction).
#ifdef MYSLAM_BLOCK_TRACE
__device__ unsigned long long* g_bt_buf = nullptr;
__device__ unsigned int g_bt_cap = 0, g_bt_n = 0;
struct BlockTrace {
    unsigned long long t0 = 0, marks = 0, aux = 0; int kid; unsigned hw = 0; bool on = false;      // aux: a per-kernel note (k_octree: level | candidates << 8 | rounds << 32)
    // phase boundary i (0..3) of the block: time since its start in 10 ns units, 16 bits each (k_fast_strip: tile staged / scored / NMS done)
    __device__ __forceinline__ void mark(int i) {
        if (on) marks |= ((__builtin_amdgcn_s_memrealtime() - t0) & 0xffffull) << (16 * i);
    }
    __device__ __forceinline__ BlockTrace(int kid_) : kid(kid_) {
        if (threadIdx.x != 0 || g_bt_buf == nullptr) return;
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;            // HW_REG_XCC_ID
        if (xcc != 0) return;
        hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                   // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
        if (((hw >> 13) & 7) != 0) return;                                                // shader engine 0 only (8 CUs): all blocks of XCD 0 on one counter slowed the step by half
        on = true; t0 = __builtin_amdgcn_s_memrealtime();                                 // 100 MHz
    }
    __device__ __forceinline__ ~BlockTrace() {
        if (!on) return;
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
        const unsigned i = atomicAdd(&g_bt_n, 1u);
        if (i < g_bt_cap) {
            g_bt_buf[4 * i] = t0; g_bt_buf[4 * i + 2] = marks; g_bt_buf[4 * i + 3] = aux;
            g_bt_buf[4 * i + 1] = (dt & 0xffffffull) | ((unsigned long long)(kid & 0xf) << 24) | ((unsigned long long)((hw >> 8) & 0xff) << 32) |
                                  ((unsigned long long)(blockIdx.x & 0xffffff) << 40);
        }
    }
};
#define MYSLAM_BT(kid) BlockTrace bt_(kid)
#define MYSLAM_BT_MARK(i) bt_.mark(i)
#define MYSLAM_BT_AUX(v) bt_.aux = (v)
#define MYSLAM_BT_PARAM , BlockTrace* btp_
#define MYSLAM_BT_ARG , &bt_
#define MYSLAM_BT_MARKP(i) btp_->mark(i)
#else
#define MYSLAM_BT(kid) ((void)0)
#define MYSLAM_BT_MARK(i) ((void)0)
#define MYSLAM_BT_AUX(v) ((void)0)
#define MYSLAM_BT_PARAM
#define MYSLAM_BT_ARG
#define MYSLAM_BT_MARKP(i) ((void)0)
#endif

__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};

// circular patch rows: umax[v], ORBextractor.cpp:429-444 for HALF_PATCH_SIZE = 15 (a constant of the algorithm)
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// ------------------------------------------------------------------------------------------------
// K1: bilinear down-scale of one pyramid level from the previous one (fixed point, 11-bit weights)
// ------------------------------------------------------------------------------------------------
// source coordinate + 11-bit weights of destination index d (OpenCV resize.cpp: fx = (dx+0.5)*scale - 0.5)
__device__ __forceinline__ void resize_coord(int d, double scale, int ssize, bool is_x, int& s, int& c0, int& c1) {
    float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);
    int si = (int)floorf(f);
    f = __fsub_rn(f, (float)si);
    if (is_x) {
        if (si < 0) { f = 0.f; si = 0; }
        if (si >= ssize - 1) { f = 0.f; si = ssize - 1; }
    }
    s = si;
    c0 = __float2int_rn(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    c1 = __float2int_rn(__fmul_rn(f, 2048.f));
}

// 4 destination pixels per thread; the <= 16 source bytes they need come from 4 aligned dwords per source row
__global__ __launch_bounds__(256) void k_resize(ResizeArgs a) {
    const int b = blockIdx.z;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int dx4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy >= a.dh || dx4 >= a.dw) return;
    int sy, b0, b1;
    resize_coord(dy, a.scale_y, a.sh, false, sy, b0, b1);
    const int sy0 = min(max(sy, 0), a.sh - 1), sy1 = min(max(sy + 1, 0), a.sh - 1);
    const uint8_t* S0 = a.src + (size_t)b * a.sstride + (size_t)sy0 * a.spitch;
    const uint8_t* S1 = a.src + (size_t)b * a.sstride + (size_t)sy1 * a.spitch;
    int sx[4], a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) resize_coord(min(dx4 + k, a.dw - 1), a.scale_x, a.sw, true, sx[k], a0[k], a1[k]);
    const int xal = sx[0] & ~3;
    uint32_t w0[4], w1[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int xo = xal + 4 * j;
        const bool ok = xo + 4 <= a.spitch;
        w0[j] = ok ? *reinterpret_cast<const uint32_t*>(S0 + xo) : 0u;
        w1[j] = ok ? *reinterpret_cast<const uint32_t*>(S1 + xo) : 0u;
    }
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int o = sx[k] - xal;                       // 0 .. 12 for scale factors up to 2.6
        const int q = o >> 2;
        const uint32_t lo0 = (q == 0) ? w0[0] : (q == 1) ? w0[1] : (q == 2) ? w0[2] : w0[3];
        const uint32_t hi0 = (q == 0) ? w0[1] : (q == 1) ? w0[2] : (q == 2) ? w0[3] : 0u;
        const uint32_t lo1 = (q == 0) ? w1[0] : (q == 1) ? w1[1] : (q == 2) ? w1[2] : w1[3];
        const uint32_t hi1 = (q == 0) ? w1[1] : (q == 1) ? w1[2] : (q == 2) ? w1[3] : 0u;
        const uint32_t v0 = __builtin_amdgcn_alignbyte(hi0, lo0, (uint32_t)(o & 3));      // bytes sx, sx+1 of row 0
        const uint32_t v1 = __builtin_amdgcn_alignbyte(hi1, lo1, (uint32_t)(o & 3));
        // at the right border sx = sw-1 and the weight of sx+1 is 0 (OpenCV clamps fx there), so its value is irrelevant
        const int r0 = (int)(v0 & 0xff) * a0[k] + (int)((v0 >> 8) & 0xff) * a1[k];
        const int r1 = (int)(v1 & 0xff) * a0[k] + (int)((v1 >> 8) & 0xff) * a1[k];
        int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v = min(max(v, 0), 255);
        packed |= (uint32_t)v << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(a.dst + (size_t)b * a.dstride + (size_t)dy * a.dpitch + dx4) = packed;
}

// K1c: up to RC_MAX consecutive pyramid levels in ONE launch, for launches with little work (a live stream's frame, the one-frame drop-ins).  A
// recorded one-pair step is bound by the NUMBER of its dependent launches, not by the work in them (tools/node_count_probe.sh: ~4.6 us per
// graph node whatever it does, chip-wide), and seven of its 24 launches were pyramid levels.  Level l + 1 is written from level l as k_resize
// does; the threads of level l + 2 do not wait for it: each computes the four pixels of level l + 1 it needs itself (and those of level l + 3
// the sixteen below) — integer arithmetic, so the value is the one the other thread stores: 5 (21) interpolations per pixel instead of 1, on
// levels that shrink by 1.44 each, at a batch size where the chip is idle.  Same bits as k_resize / k_resize_strip (tests/test_gpu_fallbacks.py).
constexpr int RC_MAX = 4;
struct ResizeChain {
    const uint8_t* src; int sw, sh, spitch; size_t sstride;      // the level in memory the chain starts from
    int n;                                                      // levels produced (1 .. RC_MAX)
    uint8_t* dst[RC_MAX]; int dw[RC_MAX], dh[RC_MAX], dpitch[RC_MAX]; size_t dstride;
    double scale_x[RC_MAX], scale_y[RC_MAX];                    // of produced level j against level j - 1
    int blk0[RC_MAX + 1];                                       // first block of produced level j; blk0[n] = blocks per image
};

// pixel (x, y) of produced level J (J = 0: the level in memory), computed from the level in memory
template <int J>
__device__ __forceinline__ int rc_pix(const ResizeChain& c, const uint8_t* S, int x, int y) {
    if constexpr (J == 0) {
        return S[(size_t)y * c.spitch + x];
    } else {
        const int sw = J == 1 ? c.sw : c.dw[J - 2], sh = J == 1 ? c.sh : c.dh[J - 2];
        int sx, a0, a1, sy, b0, b1;
        resize_coord(x, c.scale_x[J - 1], sw, true, sx, a0, a1);
        resize_coord(y, c.scale_y[J - 1], sh, false, sy, b0, b1);
        const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
        const int sx1 = min(sx + 1, sw - 1);                    // at the right border the weight of sx + 1 is 0 (OpenCV clamps fx there)
        const int r0 = rc_pix<J - 1>(c, S, sx, sy0) * a0 + rc_pix<J - 1>(c, S, sx1, sy0) * a1;
        const int r1 = rc_pix<J - 1>(c, S, sx, sy1) * a0 + rc_pix<J - 1>(c, S, sx1, sy1) * a1;
        const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        return min(max(v, 0), 255);
    }
}

// pixels a thread produces: 4 (one dword store) down to the third level of a chain, 1 on the fourth (85 interpolations per pixel: spread wider)
__host__ __device__ constexpr int rc_px(int J) { return J >= 4 ? 1 : 4; }

template <int J>
__device__ __forceinline__ void rc_store(const ResizeChain& c, const uint8_t* S, uint8_t* D, int item) {
    const int dw = c.dw[J - 1], dh = c.dh[J - 1], gpr = (dw + 3) >> 2;
    if constexpr (rc_px(J) == 4) {
        const int dy = item / gpr, dx4 = (item - dy * gpr) * 4;
        if (dy >= dh) return;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) packed |= (uint32_t)rc_pix<J>(c, S, min(dx4 + k, dw - 1), dy) << (8 * k);
        *reinterpret_cast<uint32_t*>(D + (size_t)dy * c.dpitch[J - 1] + dx4) = packed;      // row pitches are multiples of 64: the padding takes the last group
    } else {
        const int dy = item / (4 * gpr), dx = item - dy * 4 * gpr;                           // the same bytes, padding of the last group included
        if (dy >= dh) return;
        D[(size_t)dy * c.dpitch[J - 1] + dx] = (uint8_t)rc_pix<J>(c, S, min(dx, dw - 1), dy);
    }
}

__device__ __forceinline__ void rc_block(const ResizeChain& c, int blk, int b) {
    const uint8_t* S = c.src + (size_t)b * c.sstride;
    if (blk < c.blk0[1]) rc_store<1>(c, S, c.dst[0] + (size_t)b * c.dstride, blk * 256 + (int)threadIdx.x);
    else if (blk < c.blk0[2]) rc_store<2>(c, S, c.dst[1] + (size_t)b * c.dstride, (blk - c.blk0[1]) * 256 + (int)threadIdx.x);
    else if (blk < c.blk0[3]) rc_store<3>(c, S, c.dst[2] + (size_t)b * c.dstride, (blk - c.blk0[2]) * 256 + (int)threadIdx.x);
    else rc_store<4>(c, S, c.dst[3] + (size_t)b * c.dstride, (blk - c.blk0[3]) * 256 + (int)threadIdx.x);
}

__global__ __launch_bounds__(256) void k_resize_chain(ResizeChain c) { rc_block(c, blockIdx.x, blockIdx.y); }

// K1b: the same arithmetic, register-only column strips.  Lane l owns destination columns 4l..4l+3 of a 256-column strip
// and walks RS_R destination rows: x coordinates / weights are computed once, every needed source row is sampled once
// (one unaligned 2-byte load per pixel = the two horizontal taps, v_perm + v_dot2_u32_u16 = the 11-bit interpolation) and
// reused by the destination rows that share it (the OpenCV row cache), all row decisions are wave-uniform.
constexpr int RS_R = 8;                      // destination rows per band (power of two)
constexpr int RS_NB = 2;                     // bands a wave walks with one set of column coordinates (RS_NB * RS_R <= 64)
constexpr int RS_MAXR = 12;                  // source rows a band may span: RS_R * scale_y + 2 (scale factors up to 1.25)

__global__ __launch_bounds__(256) void k_resize_strip(ResizeArgs a, int nstrips, int nbands) {
    MYSLAM_SIDE_PRIO();
    MYSLAM_BT(1);
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));   // scalar: row addressing goes to the SALU
    const int ngroups = (nbands + RS_NB - 1) / RS_NB;                                              // a wave walks RS_NB consecutive bands of its strip
    if (wid >= nstrips * ngroups) return;
    const int strip = wid % nstrips, group = wid / nstrips;
    const int b = blockIdx.z;
    const int dx4 = strip * 256 + 4 * lane;
    const bool has = dx4 < a.dw;
    const bool ext = b < a.n0;                                   // block-uniform: level 0 read in place
    const uint8_t* src = ext ? a.src0 + (size_t)b * a.sstride0 : a.src + (size_t)b * a.sstride;
    const int spitch = ext ? a.spitch0 : a.spitch;
    uint8_t* dst = a.dst + (size_t)b * a.dstride;
    int sx[4]; uint32_t aw[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int c0, c1;
        resize_coord(min(dx4 + k, a.dw - 1), a.scale_x, a.sw, true, sx[k], c0, c1);
        aw[k] = (uint32_t)c0 | ((uint32_t)c1 << 16);
    }
    // One unaligned 8-byte load per source row covers the byte pairs of all four destination columns (their source columns span at
    // most 3 scale_x + 2 <= 7 bytes; the launcher checks scale_x): the kernel is bound by the ISSUE of its gather loads, so four
    // 2-byte loads per row cost four times as much.  The pairs are picked with byte permutes whose selectors are lane constants.
    uint32_t selk[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t o = (uint32_t)(sx[k] - sx[0]); selk[k] = 0x0c000c00u | ((o + 1u) << 16) | o; }
    // the row coordinates of all RS_NB bands: lane k computes row dy0 + k once (RS_NB * RS_R <= 64), the loops read them back as scalars
    const int gy0 = group * RS_NB * RS_R;
    int ysy, yb0, yb1;
    resize_coord(min(gy0 + (lane & (RS_NB * RS_R - 1)), a.dh - 1), a.scale_y, a.sh, false, ysy, yb0, yb1);
    for (int bi = 0; bi < RS_NB; bi++) {
        const int dy0 = gy0 + bi * RS_R, dy1 = min(dy0 + RS_R, a.dh), l0 = bi * RS_R;
        if (dy0 >= a.dh) break;                                      // uniform
        // every source row the band needs, fetched in ONE batch of loads (the kernel is latency-bound otherwise) and interpolated
        // horizontally once (the OpenCV row cache, kept in registers)
        const int rfirst = min(max(__builtin_amdgcn_readlane(ysy, l0), 0), a.sh - 1);
        const int rlast = min(max(__builtin_amdgcn_readlane(ysy, l0 + dy1 - 1 - dy0) + 1, 0), a.sh - 1);
        const int nrows = rlast - rfirst + 1;                        // <= RS_MAXR (checked by the launcher)
        uint2 raw8[RS_MAXR];
        if (ext) {                                                   // block-uniform: a caller's rows may start at any byte
#pragma unroll
            for (int rr = 0; rr < RS_MAXR; rr++) {
                const uint8_t* row = src + (size_t)(rfirst + min(rr, nrows - 1)) * spitch;
                raw8[rr] = make_uint2(0u, 0u);
                // at the right border sx = sw-1 and the weight of sx+1 is 0 (OpenCV clamps fx there): the bytes past the row are never weighted
                if (has && rr < nrows) __builtin_memcpy(&raw8[rr], row + sx[0], 8);
            }
        } else {
            // pyramid planes: 12 bytes from the dword-aligned address below sx[0], shifted into place with two v_alignbyte — an 8-byte load
            // at BYTE alignment costs the texture addresser 32 cycles per wave instruction, the aligned 12-byte one 17.7 (tools/ta_probe.hip)
            uint32_t raw12[RS_MAXR][3];
            const uint32_t xal = (uint32_t)sx[0] & ~3u, sh = (uint32_t)sx[0] & 3u;
#pragma unroll
            for (int rr = 0; rr < RS_MAXR; rr++) {
                const uint8_t* row = src + (size_t)(rfirst + min(rr, nrows - 1)) * spitch;
                raw12[rr][0] = raw12[rr][1] = raw12[rr][2] = 0u;
                if (has && rr < nrows) __builtin_memcpy(raw12[rr], __builtin_assume_aligned(row + xal, 4), 12);
            }
#pragma unroll
            for (int rr = 0; rr < RS_MAXR; rr++)
                raw8[rr] = make_uint2(__builtin_amdgcn_alignbyte(raw12[rr][1], raw12[rr][0], sh), __builtin_amdgcn_alignbyte(raw12[rr][2], raw12[rr][1], sh));
        }
        // horizontal pass: raw[rr][k] <- 16 x the row-cache value (<= 32640) of source row rfirst + rr at this lane's 4 columns
        // (the low four bits are cleared instead of shifted out: the vertical step multiplies 24-bit operands and keeps bits 32..)
        // (a band of 8 destination rows at scale 1.2 spans 10 or 11 source rows, RS_MAXR = 12 is the bound for 1.25: the rows a band does not
        // have are skipped by a scalar branch — the vertical pass below never reads them — instead of being interpolated from zeros)
        uint32_t raw[RS_MAXR][4];
#pragma unroll
        for (int rr = 0; rr < RS_MAXR; rr++) {
            if (rr >= RS_R + 1 && rr >= nrows) {                          // wave-uniform; rows 0 .. RS_R always exist for scale >= 1
#pragma unroll
                for (int k = 0; k < 4; k++) raw[rr][k] = 0u;
                continue;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t pp = __builtin_amdgcn_perm(raw8[rr].y, raw8[rr].x, selk[k]);       // (p0, p1) as two u16
                raw[rr][k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pp), __builtin_bit_cast(u16x2, aw[k]), 0u, false) & ~15u;
            }
        }
        // vertical pass: walk the source rows statically (the row cache stays in registers, no run-time register indexing) and emit the
        // destination rows whose upper source row is the current one — at most one per source row for scale >= 1; the row coordinates
        // are wave-uniform scalars.  (b * t) >> 16 with b <= 2048 and t <= 32640 is the high word of (b << 12) * (16 t): both factors
        // fit 24 bits, so it is ONE full-rate v_mul_hi_u32_u24 (a 32-bit v_mul_lo_u32 runs at quarter rate).
        int dy = dy0;
        int sy = __builtin_amdgcn_readlane(ysy, l0);
        uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane(yb0, l0) << 12, b1 = (uint32_t)__builtin_amdgcn_readlane(yb1, l0) << 12;
#pragma unroll
        for (int rr = 0; rr < RS_MAXR; rr++) {
            while (dy < dy1 && min(max(sy, 0), a.sh - 1) - rfirst == rr) {          // uniform
                const bool same = min(max(sy + 1, 0), a.sh - 1) - rfirst == rr;     // clamped at an image border: both taps on this row
                uint32_t packed = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t tA = raw[rr][k], tB = same ? raw[rr][k] : raw[rr + 1 < RS_MAXR ? rr + 1 : rr][k];
                    uint32_t hA, hB;        // both factors < 2^24 by construction: the instruction itself, without the masks C++ needs to say so
                    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hA) : "s"(b0), "v"(tA));
                    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hB) : "s"(b1), "v"(tB));
                    const uint32_t v = (hA + hB + 2u) >> 2;                         // in [0, 255]: the weights of each axis sum to 2048
                    packed |= v << (8 * k);
                }
                if (has) *reinterpret_cast<uint32_t*>(dst + (size_t)dy * a.dpitch + dx4) = packed;
                dy++;
                if (dy < dy1) {
                    sy = __builtin_amdgcn_readlane(ysy, l0 + dy - dy0);
                    b0 = (uint32_t)__builtin_amdgcn_readlane(yb0, l0 + dy - dy0) << 12; b1 = (uint32_t)__builtin_amdgcn_readlane(yb1, l0 + dy - dy0) << 12;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2: 7x7 separable Gaussian, Q8 coefficients (<= 255 each, sum <= 257: the Q8.8 row sums fit 16 bits), REFLECT_101, u8 saturation.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}

// Four vertical sums (Q16.16 + rounding seed, each < 2^25) -> four u8 in one dword, SATURATED as OpenCV's ufixedpoint32 -> uint8_t
// conversion does: taps that sum to 257 (OpenCV 3.4.8's independently rounded sigma = 2 table) reach 257 on saturated image regions.
// v_perm picks the two high halves, v_sat_pk_u8_i16 clamps both to 0..255: 5 instructions per 4 pixels.
__device__ __forceinline__ uint32_t pack4_sat_hi16(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3) {
    uint32_t p01 = __builtin_amdgcn_perm(s1, s0, 0x07060302u), p23 = __builtin_amdgcn_perm(s3, s2, 0x07060302u), u01, u23;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(u01) : "v"(p01));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(u23) : "v"(p23));
    return (u01 & 0xffffu) | (u23 << 16);
}

// K2b: LDS-tiled form on the dot-product units (any alignment, any size: the in-place DeepLCD blur and tiny images):
// horizontal pass = 2 x v_dot4_u32_u8 per pixel on byte-aligned windows (v_alignbyte), two rows at a time so that the
// u16 results are stored as vertical pairs (h[2j][x], h[2j+1][x]); vertical pass = 4 x v_dot2_u32_u16 per pixel on
// those pairs with the rounding constant as accumulator seed.  128 x 64 outputs per block.
constexpr int B2_W = 128, B2_H = 32, B2_IH = B2_H + 6, B2_IP = B2_W + 8, B2_IDW = B2_IP / 4, B2_PR = B2_IH / 2;

__global__ __launch_bounds__(256) void k_blur7_dot(BlurArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[B2_IH * B2_IP];
    __shared__ __attribute__((aligned(16))) uint32_t s_hp[B2_PR * B2_W];
    const int t = threadIdx.x;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * B2_W, y0 = blockIdx.y * B2_H;
    const uint8_t* src = a.src + (size_t)b * a.sstride;
    const bool aligned = ((a.spitch & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 3) == 0);
    const int nout = min(B2_H, a.h - y0);                  // output rows of this tile
    const int nrp = (nout + 6 + 1) >> 1;                   // input row pairs needed
    // 1. stage the input tile (rows reflected; columns outside [0,w) are fixed up in step 2)
    for (int i = t; i < 2 * nrp * B2_IDW; i += 256) {
        const int r = i / B2_IDW, k = i - r * B2_IDW;
        const int gy = reflect101(y0 + r - 3, a.h);
        const int gx = x0 - 4 + 4 * k;
        const uint8_t* row = src + (size_t)gy * a.spitch;
        uint32_t v;
        if (aligned && gx >= 0 && gx + 4 <= a.spitch) {
            v = *reinterpret_cast<const uint32_t*>(row + gx);
        } else {
            v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) v |= (uint32_t)row[min(max(gx + j, 0), a.spitch - 1)] << (8 * j);
        }
        *reinterpret_cast<uint32_t*>(&s_in[r * B2_IP + 4 * k]) = v;
    }
    __syncthreads();
    // 2. BORDER_REFLECT_101 columns: x in [-3,-1] and [w, w+2]; their mirror images are inside this tile
    for (int i = t; i < 2 * nrp * 6; i += 256) {
        const int r = i / 6, j = i - r * 6;
        const int x = (j < 3) ? j - 3 : a.w + (j - 3);
        const int c = x - (x0 - 4);
        if (c >= 0 && c < B2_IP) {
            const int cr = reflect101(x, a.w) - (x0 - 4);
            if (cr >= 0 && cr < B2_IP) s_in[r * B2_IP + c] = s_in[r * B2_IP + cr];
        }
    }
    __syncthreads();
    // 3. horizontal pass, 4 x-positions x 2 rows per item
    const uint32_t qa = (uint32_t)a.q[0] | ((uint32_t)a.q[1] << 8) | ((uint32_t)a.q[2] << 16) | ((uint32_t)a.q[3] << 24);
    const uint32_t qb = (uint32_t)a.q[4] | ((uint32_t)a.q[5] << 8) | ((uint32_t)a.q[6] << 16);
    for (int i = t; i < nrp * (B2_W / 4); i += 256) {
        const int rp = i >> 5, qd = i & 31;
        uint32_t hh[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(&s_in[(2 * rp + rr) * B2_IP + 4 * qd]);
            const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
            const uint32_t A0 = __builtin_amdgcn_alignbyte(w1, w0, 1u), A1 = __builtin_amdgcn_alignbyte(w1, w0, 2u);
            const uint32_t A2 = __builtin_amdgcn_alignbyte(w1, w0, 3u), A3 = w1;
            const uint32_t B0 = __builtin_amdgcn_alignbyte(w2, w1, 1u), B1 = __builtin_amdgcn_alignbyte(w2, w1, 2u);
            const uint32_t B2 = __builtin_amdgcn_alignbyte(w2, w1, 3u), B3 = w2;
            hh[rr][0] = __builtin_amdgcn_udot4(A0, qa, __builtin_amdgcn_udot4(B0, qb, 0u, false), false);
            hh[rr][1] = __builtin_amdgcn_udot4(A1, qa, __builtin_amdgcn_udot4(B1, qb, 0u, false), false);
            hh[rr][2] = __builtin_amdgcn_udot4(A2, qa, __builtin_amdgcn_udot4(B2, qb, 0u, false), false);
            hh[rr][3] = __builtin_amdgcn_udot4(A3, qa, __builtin_amdgcn_udot4(B3, qb, 0u, false), false);
        }
        *reinterpret_cast<uint4*>(&s_hp[rp * B2_W + 4 * qd]) =
            make_uint4(hh[0][0] | (hh[1][0] << 16), hh[0][1] | (hh[1][1] << 16), hh[0][2] | (hh[1][2] << 16), hh[0][3] | (hh[1][3] << 16));
    }
    __syncthreads();
    // 4. vertical pass: output rows (2 yp, 2 yp + 1) x 4 x-positions per item
    const uint32_t t01 = (uint32_t)a.q[0] | ((uint32_t)a.q[1] << 16), t23 = (uint32_t)a.q[2] | ((uint32_t)a.q[3] << 16);
    const uint32_t t45 = (uint32_t)a.q[4] | ((uint32_t)a.q[5] << 16), t6_ = (uint32_t)a.q[6];
    const uint32_t t_0 = (uint32_t)a.q[0] << 16, t12 = (uint32_t)a.q[1] | ((uint32_t)a.q[2] << 16);
    const uint32_t t34 = (uint32_t)a.q[3] | ((uint32_t)a.q[4] << 16), t56 = (uint32_t)a.q[5] | ((uint32_t)a.q[6] << 16);
    uint8_t* dst = a.dst + (size_t)b * a.dstride;
    for (int i = t; i < ((nout + 1) >> 1) * (B2_W / 4); i += 256) {
        const int yp = i >> 5, qd = i & 31;
        const int oy = y0 + 2 * yp, ox = x0 + 4 * qd;
        if (ox >= a.w) continue;
        uint4 D[4];
#pragma unroll
        for (int j = 0; j < 4; j++) D[j] = *reinterpret_cast<const uint4*>(&s_hp[(yp + j) * B2_W + 4 * qd]);
        uint32_t S0[4], S1[4];
#pragma unroll
        for (int xi = 0; xi < 4; xi++) {
            const uint32_t d0 = xi == 0 ? D[0].x : xi == 1 ? D[0].y : xi == 2 ? D[0].z : D[0].w;
            const uint32_t d1 = xi == 0 ? D[1].x : xi == 1 ? D[1].y : xi == 2 ? D[1].z : D[1].w;
            const uint32_t d2 = xi == 0 ? D[2].x : xi == 1 ? D[2].y : xi == 2 ? D[2].z : D[2].w;
            const uint32_t d3 = xi == 0 ? D[3].x : xi == 1 ? D[3].y : xi == 2 ? D[3].z : D[3].w;
            uint32_t s0 = 32768u, s1 = 32768u;
#define DOT2(d, tp, acc) __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d), __builtin_bit_cast(u16x2, tp), acc, false)
            s0 = DOT2(d0, t01, s0); s0 = DOT2(d1, t23, s0); s0 = DOT2(d2, t45, s0); s0 = DOT2(d3, t6_, s0);
            s1 = DOT2(d0, t_0, s1); s1 = DOT2(d1, t12, s1); s1 = DOT2(d2, t34, s1); s1 = DOT2(d3, t56, s1);
#undef DOT2
            S0[xi] = s0; S1[xi] = s1;
        }
        const uint32_t lo = pack4_sat_hi16(S0[0], S0[1], S0[2], S0[3]), hi = pack4_sat_hi16(S1[0], S1[1], S1[2], S1[3]);
        // destination pitch is a multiple of 64: the 4-byte store never leaves the row; bytes past w are padding
        if (a.dtiled) {                                                  // oy is even: both rows lie in the same tile
            uint8_t* o = dst + tiled_off(ox, oy, a.dpitch);
            *reinterpret_cast<uint32_t*>(o) = lo;
            if (oy + 1 < a.h) *reinterpret_cast<uint32_t*>(o + 16) = hi;
            continue;
        }
        *reinterpret_cast<uint32_t*>(dst + (size_t)oy * a.dpitch + ox) = lo;
        if (oy + 1 < a.h) *reinterpret_cast<uint32_t*>(dst + (size_t)(oy + 1) * a.dpitch + ox) = hi;
    }
}

// K2c: register-only variant of K2b.  One wave filters a 256-column x B3_R-row band walking down the rows: lane l owns
// columns 4l..4l+3, loads ONE aligned dword per input row (a fully coalesced 256-byte row segment per wave), gets its
// neighbours' dwords over the DPP network (wave_shr/shl), keeps the last four vertical pairs of horizontal sums in
// registers and emits two output rows per two input rows.  No LDS, no barriers.  Needs w >= 8 and taps <= 255.
constexpr int B3_R = 32;                       // output rows per wave (64: fewer waves, 0.81 ms against 0.74 per 1024 images)

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t dpp_wave_shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }

// All strip-capable levels of a pyramid in ONE launch (one wave per 256-column x 32-row band of some level): a one-frame call spends
// more time between its launches than inside them (8 blur launches -> 1), and the small levels fill the gaps of the large ones.
struct BlurMulti { BlurArgs a[MAXL]; int wave0[MAXL + 1]; int nstrips[MAXL]; int n; };
constexpr int B3_TS = 144;                     // LDS bytes per staged tile (128 + 16: the dword writes of a wave then spread over all banks)
// one wave's band: gw = wave id over all levels (wave-uniform), b = image, tl = the wave's 16 * B3_TS bytes of LDS (tiled destinations: a wave
// parks 8 output rows of its 256 columns (16 tiles) there and writes them out as 2 KB of whole cache lines — two 16-byte stores per lane
// instead of eight dword stores that each touch 16 lines).  Called by k_blur7_strip and by the oct-tree launch of small batches (k_octree<512>).
__device__ __forceinline__ void blur7_strip_wave(const BlurMulti& M, int gw, int b, uint8_t* const tl) {
    const int lane = threadIdx.x & 63;
    if (gw >= M.wave0[M.n]) return;
    int lvl = 0;
    while (lvl + 1 < M.n && gw >= M.wave0[lvl + 1]) lvl++;
    const BlurArgs& a = M.a[lvl];
    const int wid = gw - M.wave0[lvl], nstrips = M.nstrips[lvl];                                  // wave id -> (band, strip) of its level
    const int strip = wid % nstrips, band = wid / nstrips;
    const int x0 = strip * 256 + 4 * lane, y0 = band * B3_R;
    const bool ext = b < a.n0;                                   // block-uniform: level 0 read in place (rows of any alignment; the dword
    const uint8_t* src = ext ? a.src0 + (size_t)b * a.sstride0 : a.src + (size_t)b * a.sstride;      // that holds column w-1 may reach into
    const int spitch = ext ? a.spitch0 : a.spitch;                                                   // the next row: never the last image)
    const int sread = ext ? ((a.w + 3) & ~3) : a.spitch;         // bytes of a row that may be read
    uint8_t* dst = a.dst + (size_t)b * a.dstride;
    const int nout = min(B3_R, a.h - y0);
    const int npair = (nout + 6 + 1) >> 1;                       // input row pairs to walk
    // column roles
    const bool has = x0 < a.w;                                   // owns at least one image column
    const bool lastq = has && x0 + 4 >= a.w;                     // owns column w-1
    const bool needR = lane == 63 && x0 + 4 < a.w;
    const int m = a.w - x0;                                      // valid bytes in the last dword (1..4) when lastq
    uint32_t sel1 = 0x07060504u, sel2 = 0;
    if (lastq) {                                                 // REFLECT_101 on the right: byte j >= m comes from relative offset 2m-2-j
        sel1 = 0; 
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int o1 = j < m ? j : 2 * m - 2 - j, o2 = 2 * m - 2 - (4 + j);
            sel1 |= (uint32_t)min(max(4 + o1, 0), 7) << (8 * j);
            sel2 |= (uint32_t)min(max(4 + o2, 0), 7) << (8 * j);
        }
    }
    // lane 63 whose right neighbour (in the next wave) is the image's last dword mirrors that dword's tail itself
    uint32_t selR = 0x07060504u;
    if (needR && x0 + 8 >= a.w) {
        const int mr = a.w - (x0 + 4);
        selR = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) selR |= (uint32_t)min(max(4 + (j < mr ? j : 2 * mr - 2 - j), 0), 7) << (8 * j);
    }
    const uint32_t qa = (uint32_t)a.q[0] | ((uint32_t)a.q[1] << 8) | ((uint32_t)a.q[2] << 16) | ((uint32_t)a.q[3] << 24);
    const uint32_t qb = (uint32_t)a.q[4] | ((uint32_t)a.q[5] << 8) | ((uint32_t)a.q[6] << 16);
    const uint32_t t01 = (uint32_t)a.q[0] | ((uint32_t)a.q[1] << 16), t23 = (uint32_t)a.q[2] | ((uint32_t)a.q[3] << 16);
    const uint32_t t45 = (uint32_t)a.q[4] | ((uint32_t)a.q[5] << 16), t6_ = (uint32_t)a.q[6];
    const uint32_t t_0 = (uint32_t)a.q[0] << 16, t12 = (uint32_t)a.q[1] | ((uint32_t)a.q[2] << 16);
    const uint32_t t34 = (uint32_t)a.q[3] | ((uint32_t)a.q[4] << 16), t56 = (uint32_t)a.q[5] | ((uint32_t)a.q[6] << 16);

    // the two dwords of one input row this lane loads: its own and — lanes 0 / 63 only — the neighbour strip's adjacent dword (the
    // other lanes get their neighbours over DPP and re-read their own dword).  Both loads are unconditional from clamped
    // addresses (rows are mirrored at most once: the launcher guarantees h >= 8; every row holds `spitch` readable bytes): no
    // branch sits between a load and its use, so the prefetch of the next row pair stays in flight behind a counted s_waitcnt.
    const int xo = min(x0, sread - 4);
    const int xe = lane == 0 ? max(x0 - 4, 0) : lane == 63 ? min(x0 + 4, sread - 4) : xo;
    auto load_row = [&](int j, uint32_t& c, uint32_t& e) {
        int y = y0 - 3 + j;
        y = y < 0 ? -y : y;
        y = y >= a.h ? 2 * a.h - 2 - y : y;
        const uint8_t* row = src + (size_t)y * spitch;
        __builtin_memcpy(&c, row + xo, 4);                       // (a caller's rows need not be dword-aligned)
        __builtin_memcpy(&e, row + xe, 4);
    };
    // horizontal pass of one row: 4 sums (<= 65280).  The neighbours' dwords arrive over DPP; lanes 0 / 63 have no DPP source and
    // keep the `old` operand = the dword loaded from the adjacent strip.  Waves that touch an image border (edge_tag = true)
    // additionally mirror the columns outside the image with byte permutes.
    auto hrow = [&](auto edge_tag, uint32_t c, uint32_t l, uint32_t r, uint32_t* h) {
        uint32_t w0, w1, w2;
        if constexpr (decltype(edge_tag)::value) {
            const uint32_t lfix = x0 > 0 ? l : __builtin_amdgcn_perm(c, c, 0x01020300u);            // x = -3..-1 mirror x = 3..1
            w0 = (uint32_t)__builtin_amdgcn_update_dpp((int)lfix, (int)c, 0x138, 0xf, 0xf, false);
            w1 = __builtin_amdgcn_perm(c, w0, sel1);                                                // identity unless this lane owns column w-1
            const uint32_t rfix = __builtin_amdgcn_perm(r, c, selR);
            w2 = (uint32_t)__builtin_amdgcn_update_dpp((int)rfix, (int)w1, 0x130, 0xf, 0xf, false);
            if (lastq) w2 = __builtin_amdgcn_perm(c, w0, sel2);
        } else {
            w0 = (uint32_t)__builtin_amdgcn_update_dpp((int)l, (int)c, 0x138, 0xf, 0xf, false);
            w1 = c;
            w2 = (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)c, 0x130, 0xf, 0xf, false);
        }
        const uint32_t A0 = __builtin_amdgcn_alignbyte(w1, w0, 1u), A1 = __builtin_amdgcn_alignbyte(w1, w0, 2u);
        const uint32_t A2 = __builtin_amdgcn_alignbyte(w1, w0, 3u), A3 = w1;
        const uint32_t B0 = __builtin_amdgcn_alignbyte(w2, w1, 1u), B1 = __builtin_amdgcn_alignbyte(w2, w1, 2u);
        const uint32_t B2 = __builtin_amdgcn_alignbyte(w2, w1, 3u), B3 = w2;
        h[0] = __builtin_amdgcn_udot4(A0, qa, __builtin_amdgcn_udot4(B0, qb, 0u, false), false);
        h[1] = __builtin_amdgcn_udot4(A1, qa, __builtin_amdgcn_udot4(B1, qb, 0u, false), false);
        h[2] = __builtin_amdgcn_udot4(A2, qa, __builtin_amdgcn_udot4(B2, qb, 0u, false), false);
        h[3] = __builtin_amdgcn_udot4(A3, qa, __builtin_amdgcn_udot4(B3, qb, 0u, false), false);
    };
    // the parked tile row -> memory: chunk c = lane + 64 j of the 128 (tile, row) chunks in memory order; tiles past the row pitch do not exist
    auto flush_tiles = [&](int ty) __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint8_t* o = dst + (size_t)ty * a.dpitch * 8 + (size_t)strip * 2048;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int c = lane + 64 * j;
            const uint4 v = *reinterpret_cast<const uint4*>(tl + (c >> 3) * B3_TS + ((c & 7) << 4));
            if (strip * 256 + 16 * (c >> 3) < a.dpitch) *reinterpret_cast<uint4*>(o + 16 * c) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto run = [&](auto edge_tag) {
        int dirty_ty = -1;                                     // tile row with parked rows that are not written out yet (wave-uniform)
        uint32_t D[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < 4; k++) D[u][k] = 0;
        uint32_t c0, e0, c1, e1;
        load_row(0, c0, e0); load_row(1, c1, e1);
        for (int base = 0; base < npair; base += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int pi = base + u;
                if (pi < npair) {
                    uint32_t nc0, ne0, nc1, ne1;
                    load_row(2 * pi + 2, nc0, ne0); load_row(2 * pi + 3, nc1, ne1);      // prefetch (rows past the band are mirrored / unused)
                    uint32_t he[4], ho[4];
                    hrow(edge_tag, c0, e0, e0, he); hrow(edge_tag, c1, e1, e1, ho);
#pragma unroll
                    for (int k = 0; k < 4; k++) D[u][k] = he[k] | (ho[k] << 16);
                    if (pi >= 3) {
                        const int oy = y0 + 2 * (pi - 3);
                        uint32_t S0[4], S1[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t d0 = D[(u + 1) & 3][k], d1 = D[(u + 2) & 3][k], d2 = D[(u + 3) & 3][k], d3 = D[u][k];
                            uint32_t s0 = 32768u, s1 = 32768u;
#define DOT2(d, tp, acc) __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, d), __builtin_bit_cast(u16x2, tp), acc, false)
                            s0 = DOT2(d0, t01, s0); s0 = DOT2(d1, t23, s0); s0 = DOT2(d2, t45, s0); s0 = DOT2(d3, t6_, s0);
                            s1 = DOT2(d0, t_0, s1); s1 = DOT2(d1, t12, s1); s1 = DOT2(d2, t34, s1); s1 = DOT2(d3, t56, s1);
#undef DOT2
                            S0[k] = s0; S1[k] = s1;
                        }
                        const uint32_t lo = pack4_sat_hi16(S0[0], S0[1], S0[2], S0[3]), hi = pack4_sat_hi16(S1[0], S1[1], S1[2], S1[3]);
                        if (a.dtiled) {                                  // wave-uniform.  Rows oy, oy + 1 of the tile row: oy & 7 = 2 ((u + 1) & 3)
                            uint8_t* w = tl + (lane >> 2) * B3_TS + ((lane & 3) << 2) + 32 * ((u + 1) & 3);
                            *reinterpret_cast<uint32_t*>(w) = lo;
                            *reinterpret_cast<uint32_t*>(w + 16) = hi;       // (row h of a plane whose height is odd: inside the padded tile, never read)
                            dirty_ty = oy >> 3;
                            if (u == 2) { flush_tiles(dirty_ty); dirty_ty = -1; }
                        } else if (has) {      // destination pitch is a multiple of 64: the dword never leaves the row; bytes past w are padding
                            *reinterpret_cast<uint32_t*>(dst + (size_t)oy * a.dpitch + x0) = lo;
                            if (oy + 1 < a.h) *reinterpret_cast<uint32_t*>(dst + (size_t)(oy + 1) * a.dpitch + x0) = hi;
                        }
                    }
                    c0 = nc0; e0 = ne0; c1 = nc1; e1 = ne1;
                }
            }
        }
        if (dirty_ty >= 0) flush_tiles(dirty_ty);              // the band's last tile row (fewer than 8 rows: the others are padding)
    };
    // interior strips: no lane sees an image border, the neighbour dwords of lanes 0 / 63 are plain loads
    const bool interior = strip > 0 && (strip + 1) * 256 + 4 <= a.w;
    if (interior) run(std::false_type{}); else run(std::true_type{});
}
__global__ __launch_bounds__(256) void k_blur7_strip(BlurMulti M) {
    MYSLAM_SIDE_PRIO();
    MYSLAM_BT(3);
    __shared__ __attribute__((aligned(16))) uint8_t s_tl[4][16 * B3_TS];
    blur7_strip_wave(M, __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6))), blockIdx.z, s_tl[threadIdx.x >> 6]);      // scalar wave id: row addressing goes to the SALU
}

// ------------------------------------------------------------------------------------------------
// K2d: the 7 x 7 Gaussian on the int8 matrix cores (MYSLAM_ORB_OPT_BLUR_MFMA).  The blur is exact integer arithmetic —
// out = sat8((sum_r q_r (sum_c q_c p) + 32768) >> 16) — i.e. two banded (Toeplitz) matrix products, Th over the columns and Tv over the
// rows, with REFLECT_101 folded into the bands.  Why: the step is bound by VALU issue while the matrix pipe idles;
// this form needs ~180 VALU instructions per 1024 pixels against ~420 in k_blur7_strip.  One wave walks a 32-column strip of one level
// downwards in 32 x 32 tiles:
//   1. H' = (P - 128) x Th^T + 128   A = the image tile as it lies in memory (lane = row, 16 consecutive pixels per lane and k block),
//                                    B = Th of this strip (table), K = the 64 columns around the strip: 2 MFMAs.  With the accumulator
//                                    seeded with 128, H' = H - 128 sum(q) + 128 is a signed 16-bit number for any tap table of sum <= 257.
//                                    The result leaves every lane with ONE column and 16 rows of H' — the shape of a B operand whose k
//                                    slots are rows —
//   2. V = Tv x H                    so the vertical pass needs no data movement: H' is split into its high and low byte planes, each is
//                                    multiplied by the Tv blocks of the tile above, this tile and the tile below (table): 6 MFMAs,
//                                    out = min(255, (256 S_hi + S_lo + vconst) >> 16).
//   3. transpose                     the result is again one column per lane; one more MFMA against a 0/1 selection matrix with the operand
//                                    roles swapped returns it with one ROW per lane: lane (row, half) holds 16 consecutive pixels = one
//                                    16-byte row of a 16 x 8 destination tile, so a wave's store is 8 whole 128-byte cache lines.
// All coefficient logic (bands, mirrored borders, partial tiles) lives in host-built operand tables; the kernel has no special cases.
// (Round 1 had this kernel with row-major output, where its 32-row x 32-byte stores bounded it; the tiled planes of round 3 remove that.)
typedef int bl_v4i __attribute__((ext_vector_type(4)));
typedef int bl_v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t bl_pack_byte(int r0, int r1, int r2, int r3, int which) {     // byte `which` (0..2) of four registers
    const uint32_t s2 = 0x0c0c0400u + 0x0101u * (uint32_t)which;          // (b.byte, a.byte) -> bytes 0, 1
    const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)r1, (uint32_t)r0, s2), t23 = __builtin_amdgcn_perm((uint32_t)r3, (uint32_t)r2, s2);
    return __builtin_amdgcn_perm(t23, t01, 0x05040100u);
}

struct BlurMfmaMulti { BlurArgs a[MAXL]; int wave0[MAXL + 1]; int tabHOff[MAXL], tabVOff[MAXL]; int n; };

__global__ __launch_bounds__(256) void k_blur7_mfma(BlurMfmaMulti M) {
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (gw >= M.wave0[M.n]) return;
    int lvl = 0;
    while (lvl + 1 < M.n && gw >= M.wave0[lvl + 1]) lvl++;
    const BlurArgs& a = M.a[lvl];
    const int wid = gw - M.wave0[lvl];                              // 32-column strip of the level
    const int ntile = (a.h + 31) >> 5;
    const int b = blockIdx.z;
    const int x0 = 32 * wid, ws = min(max(x0 - 16, 0), a.w - 64);   // the 64 source columns of the strip (inside the image for any pitch)
    const bool ext = b < a.n0;                                      // block-uniform: level 0 read in place (rows of any alignment)
    const uint8_t* src = ext ? a.src0 + (size_t)b * a.sstride0 : a.src + (size_t)b * a.sstride;
    const int spitch = ext ? a.spitch0 : a.spitch;
    uint8_t* dst = a.dst + (size_t)b * a.dstride;
    const int li = lane & 31, lh = lane >> 5;
    const uint4* tabH = a.tabH + M.tabHOff[lvl];
    const uint4* tabV = a.tabV + M.tabVOff[lvl];
    const bl_v4i TH0 = __builtin_bit_cast(bl_v4i, tabH[(size_t)(wid * 2 + 0) * 64 + lane]);
    const bl_v4i TH1 = __builtin_bit_cast(bl_v4i, tabH[(size_t)(wid * 2 + 1) * 64 + lane]);
    const bl_v4i ID = __builtin_bit_cast(bl_v4i, a.ident[lane]);
    const bl_v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bl_v16i seed16 = {128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};
    const bl_v4i zero4 = {0, 0, 0, 0};
    const int vconst = a.vconst;
    // pixels are requested ahead of their use (every stage of a tile depends on the previous one)
    auto load_px = [&](int ty, uint4& p0, uint4& p1) __attribute__((always_inline)) {
        const int y = min(32 * ty + li, a.h - 1);
        const uint8_t* row = src + (size_t)y * spitch + ws + 16 * lh;
        __builtin_memcpy(&p0, row, 16); __builtin_memcpy(&p1, row + 32, 16);
    };
    // H' tile as two int8 planes (hi = H' >> 8, lo = (H' & 255) - 128), k slot b of a lane <-> tile row (b&3) + 8(b>>2) + 4 lh
    auto htile = [&](uint4 p0, uint4 p1, bl_v4i& hi, bl_v4i& lo) __attribute__((always_inline)) {
        p0.x ^= 0x80808080u; p0.y ^= 0x80808080u; p0.z ^= 0x80808080u; p0.w ^= 0x80808080u;
        p1.x ^= 0x80808080u; p1.y ^= 0x80808080u; p1.z ^= 0x80808080u; p1.w ^= 0x80808080u;
        bl_v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(bl_v4i, p0), TH0, seed16, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(bl_v4i, p1), TH1, acc, 0, 0, 0);
#pragma unroll
        for (int w = 0; w < 4; w++) {
            lo[w] = (int)(bl_pack_byte(acc[4 * w], acc[4 * w + 1], acc[4 * w + 2], acc[4 * w + 3], 0) ^ 0x80808080u);
            hi[w] = (int)bl_pack_byte(acc[4 * w], acc[4 * w + 1], acc[4 * w + 2], acc[4 * w + 3], 1);
        }
    };
    bl_v4i Hh[3], Hl[3];                                       // tiles ty-1, ty, ty+1
    Hh[0] = zero4; Hl[0] = zero4;
    // Pixel registers: three sets, set k holds a tile with index = k (mod 3).  Tile t+1 is consumed in step t and its set is refilled
    // with tile t+4 right away, so a load has three steps to arrive; the sets are addressed statically (the main loop is unrolled by three)
    uint4 P0a, P0b, P1a, P1b, P2a, P2b;
    load_px(0, P0a, P0b);
    load_px(min(1, ntile - 1), P1a, P1b);
    load_px(min(2, ntile - 1), P2a, P2b);
    htile(P0a, P0b, Hh[1], Hl[1]);
    load_px(min(3, ntile - 1), P0a, P0b);
    const int hpad = (a.h + 7) & ~7;                            // tiled planes are whole 8-row tile rows
    auto do_tile = [&](int ty, uint4& pa, uint4& pb, uint4 t0, uint4 t1, uint4 t2) __attribute__((always_inline)) {     // (pa, pb) = the set of tile ty + 1
        if (ty + 1 < ntile) htile(pa, pb, Hh[2], Hl[2]); else { Hh[2] = zero4; Hl[2] = zero4; }
        if (ty + 4 < ntile) load_px(ty + 4, pa, pb);
        bl_v16i sh = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(bl_v4i, t0), Hh[0], zero16, 0, 0, 0);
        bl_v16i sl = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(bl_v4i, t0), Hl[0], zero16, 0, 0, 0);
        sh = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(bl_v4i, t1), Hh[1], sh, 0, 0, 0);
        sl = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(bl_v4i, t1), Hl[1], sl, 0, 0, 0);
        sh = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(bl_v4i, t2), Hh[2], sh, 0, 0, 0);
        sl = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(bl_v4i, t2), Hl[2], sl, 0, 0, 0);
        // out = min(255, (256 S_hi + S_lo + vconst) >> 16) = byte 2 of the clamped sum; out - 128 as the transpose's A operand
        bl_v4i A3;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = min((sh[4 * w + k] << 8) + sl[4 * w + k] + vconst, 0x00ffffff);
            A3[w] = (int)(bl_pack_byte(v[0], v[1], v[2], v[3], 2) ^ 0x80808080u);
        }
        const bl_v16i T = __builtin_amdgcn_mfma_i32_32x32x32_i8(A3, ID, zero16, 0, 0, 0);      // lane (row li, half lh): x = (r&3) + 8(r>>2) + 4 lh
        // the two half-waves hold interleaved 4-pixel groups of the same row: swap two dwords so that each lane owns 16 contiguous
        // pixels (half 0: x 0..15, half 1: x 16..31) — one 16-byte row of a destination tile
        uint32_t d[4];
#pragma unroll
        for (int g = 0; g < 4; g++) d[g] = bl_pack_byte(T[4 * g], T[4 * g + 1], T[4 * g + 2], T[4 * g + 3], 0) ^ 0x80808080u;
        const uint32_t r0 = (uint32_t)__shfl_xor((int)(lh ? d[0] : d[2]), 32, 64), r1 = (uint32_t)__shfl_xor((int)(lh ? d[1] : d[3]), 32, 64);
        const uint4 o = lh ? make_uint4(r0, d[2], r1, d[3]) : make_uint4(d[0], r0, d[1], r1);
        const int y = 32 * ty + li, x = x0 + 16 * lh;
        if (y < hpad && x < a.dpitch) *reinterpret_cast<uint4*>(dst + tiled_off(x, y, a.dpitch)) = o;
        Hh[0] = Hh[1]; Hl[0] = Hl[1]; Hh[1] = Hh[2]; Hl[1] = Hl[2];
    };
    auto tv = [&](int ty, int o) __attribute__((always_inline)) { return tabV[(size_t)(ty * 3 + o) * 64 + lane]; };
    auto do_tile_any = [&](int ty, uint4 t0, uint4 t1, uint4 t2) __attribute__((always_inline)) {          // set of tile ty + 1 chosen at run time (wave-uniform)
        const int k = (ty + 1) % 3;
        if (k == 0) do_tile(ty, P0a, P0b, t0, t1, t2); else if (k == 1) do_tile(ty, P1a, P1b, t0, t1, t2); else do_tile(ty, P2a, P2b, t0, t1, t2);
    };
    // Tv blocks: every tile whose 7-row windows stay inside the image uses the same three blocks; they stay in registers for the main
    // loop.  The first tile and the last two read theirs from the table.
    int tlast = ntile;                                         // interior tiles are [1, tlast)
    while (tlast > 1 && 32 * (tlast - 1) + 34 >= a.h) tlast--;
    do_tile(0, P1a, P1b, tv(0, 0), tv(0, 1), tv(0, 2));
    int ty = 1;
    if (tlast > 1) {
        const uint4 iv0 = tv(1, 0), iv1 = tv(1, 1), iv2 = tv(1, 2);
        asm volatile("" ::"v"(iv0.x), "v"(iv0.y), "v"(iv0.z), "v"(iv0.w), "v"(iv1.x), "v"(iv1.y), "v"(iv1.z), "v"(iv1.w), "v"(iv2.x), "v"(iv2.y),
                     "v"(iv2.z), "v"(iv2.w));
        for (; ty + 3 <= tlast; ty += 3) {                     // ty = 1 (mod 3): tiles ty+1, ty+2, ty+3 live in sets 2, 0, 1
            do_tile(ty, P2a, P2b, iv0, iv1, iv2);
            do_tile(ty + 1, P0a, P0b, iv0, iv1, iv2);
            do_tile(ty + 2, P1a, P1b, iv0, iv1, iv2);
        }
        for (; ty < tlast; ty++) do_tile_any(ty, iv0, iv1, iv2);
    }
    for (; ty < ntile; ty++) do_tile_any(ty, tv(ty, 0), tv(ty, 1), tv(ty, 2));
}

// ---- host: operand tables of k_blur7_mfma ----
static int bl_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}
static int bl_coef(const int q[7], int out, int in, int len) {          // weight of input position `in` in output position `out`
    if (out < 0 || out >= len || in < 0 || in >= len) return 0;
    int c = 0;
    for (int t = 0; t < 7; t++) if (bl_reflect101(out + t - 3, len) == in) c += q[t];
    return c;
}
static inline int bl_slot_row(int b, int half) { return (b & 3) + 8 * (b >> 2) + 4 * half; }
static uint4 bl_pack16(const int c[16]) {
    uint32_t w[4] = {0, 0, 0, 0};
    for (int i = 0; i < 16; i++) w[i >> 2] |= (uint32_t)(c[i] & 0xff) << (8 * (i & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// appends the level's Th blocks (2 per 32-column strip) and Tv blocks (3 per 32-row tile) to `tab`; false = this level cannot take the
// matrix-core form (narrower than 64 columns, a folded coefficient above 127, a window that does not hold its mirrored columns)
bool blur_mfma_tables(int w, int h, const int q[7], std::vector<uint4>& tab, size_t& offH, size_t& offV) {
    if (w < 64 || h < 8) return false;
    int sum = 0;
    for (int t = 0; t < 7; t++) { if (q[t] < 0 || q[t] > 127) return false; sum += q[t]; }
    if (sum < 1 || sum > 257) return false;
    const int nstrip = (w + 31) / 32, ntile = (h + 31) / 32;
    const size_t keep = tab.size();
    offH = tab.size();
    for (int s = 0; s < nstrip; s++) {
        const int x0 = 32 * s, ws = std::min(std::max(x0 - 16, 0), w - 64);
        for (int kb = 0; kb < 2; kb++)
            for (int l = 0; l < 64; l++) {
                int c[16];
                for (int bb = 0; bb < 16; bb++) {
                    c[bb] = bl_coef(q, x0 + (l & 31), ws + 32 * kb + 16 * (l >> 5) + bb, w);
                    if (c[bb] > 127) { tab.resize(keep); return false; }
                }
                tab.push_back(bl_pack16(c));
            }
        // every input column an output column of this strip needs must lie inside the 64-column window
        for (int j = 0; j < 32 && x0 + j < w; j++)
            for (int t = 0; t < 7; t++) { const int in = bl_reflect101(x0 + j + t - 3, w); if (in < ws || in >= ws + 64) { tab.resize(keep); return false; } }
    }
    offV = tab.size();
    for (int ty = 0; ty < ntile; ty++)
        for (int o = 0; o < 3; o++)
            for (int l = 0; l < 64; l++) {
                int c[16];
                for (int bb = 0; bb < 16; bb++) {
                    c[bb] = bl_coef(q, 32 * ty + (l & 31), 32 * (ty + o - 1) + bl_slot_row(bb, l >> 5), h);
                    if (c[bb] > 127) { tab.resize(keep); return false; }
                }
                tab.push_back(bl_pack16(c));
            }
    return true;
}
void blur_mfma_ident(std::vector<uint4>& tab, size_t& offI) {
    offI = tab.size();
    for (int l = 0; l < 64; l++) {
        int c[16];
        for (int bb = 0; bb < 16; bb++) c[bb] = bl_slot_row(bb, l >> 5) == (l & 31) ? 1 : 0;
        tab.push_back(bl_pack16(c));
    }
}

// ------------------------------------------------------------------------------------------------
// K3: grid FAST.  One 256-thread block per strip of 4 horizontally adjacent 30-px grid cells: the strip's ROI rows (+3 px ring)
// are staged once in LDS (shared halos), every interior pixel gets its FAST-9 score (largest threshold at which it is still a
// corner), 3x3 strict-maximum NMS runs inside each cell only (as cv::FAST on the sub-Mat does), the 20 -> 7 threshold
// fallback is decided per cell, survivors are appended to the level's candidate list.
// Candidate payload: py<<20 | px<<8 | score   (px,py border-relative as in ORBextractor.cpp:871-872).
// ------------------------------------------------------------------------------------------------
// ring order = reference makeOffsets(), ORBextractor.cpp:365-369
#define FAST_RING_X {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1}
#define FAST_RING_Y {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3}

// Profiling builds only (tools/fast_phase_pmc.sh): MYSLAM_FAST_PHASE = n truncates the DENSE path of the strip kernel after phase n so that
// the hardware counters of consecutive builds difference into per-phase instruction counts.  1: block decode + staging + score-map
// clearing; 2: + the scoring loop's control and window loads (score replaced by an XOR over the 24 window dwords); 3: + the byte-pair
// picks (score = XOR over the 17 picks of a pixel pair); 4: + the min / max network (= the whole scoring phase); 5: + NMS and the
// strip's record list; 0 (the product): + path statistics, filter and global append.
#ifndef MYSLAM_FAST_PHASE
#define MYSLAM_FAST_PHASE 0
#endif

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }

// FAST-9 score = largest threshold at which the pixel is still a corner (cv::cornerScore<16> without the threshold seed):
// max over the 16 arcs of 9 of min(v - p) and of min(p - v), minus 1.
// Two-pixel packing: register i holds ring position i of two horizontally adjacent pixels (raw u8 values in 16-bit lanes), so no
// difference to the centre is taken until the end:
//   bright = max over arcs (min over the arc of p) - v,   dark = v - min over arcs (max over the arc of p).
// The 16 arcs of 9 are covered from the 8 even-aligned windows of 8: arc [2k, 2k+8] = m8[k] + p[2k+8] and
// arc [2k-1, 2k+7] = p[2k-1] + m8[k], and max(min(m, a), min(m, b)) = min(m, max(a, b)) — 36 packed ops per polarity
// for two pixels.  Returns z = score - (minTh - 1) for corners at minTh, 0 otherwise (an order-preserving shift: the
// NMS compares z, the append adds minTh - 1 back).  PAIR selects pixels (2 PAIR, 2 PAIR + 1) of the window's four.
// gfx950 has three-input packed minimum / maximum only for f16.  Ring values are zero-extended bytes in 16-bit lanes, i.e. the f16
// bit patterns of +0 and the positive denormals 1 .. 255, which order exactly like the integers; the kernels run with f16 denormals
// preserved (the AMDGPU default, amdhsa_float_denorm_mode_16_64 = 3) and a minimum / maximum only selects one of its operands, so the
// two-input steps stay v_pk_min/max_i16, the three-input steps are v_pk_minimum3/maximum3_f16 on the same registers, and differences
// are plain 16-bit integer subtractions.  (Round 1 carried 0x6400 | p — the normal numbers 1024 + p — which cost an extra OR on every
// byte pair that straddles two dwords.)
__device__ __forceinline__ s16x2 pmin3(s16x2 a, s16x2 b, s16x2 c) { s16x2 d; asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ s16x2 pmax3(s16x2 a, s16x2 b, s16x2 c) { s16x2 d; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

template <int PAIR, int ROW = 0, int NR = 7>
__device__ __forceinline__ s16x2 fast9_score_pair(const uint32_t (&r)[NR][3], s16x2 thv) {
    constexpr int RX[16] = FAST_RING_X;
    constexpr int RY[16] = FAST_RING_Y;
    constexpr int xc = 3 + 2 * PAIR;
    auto pick = [&](int row, int a) -> s16x2 {          // bytes a, a+1 of window row `row`, zero-extended into two 16-bit lanes
        const int d0 = a >> 2, d1 = ((a & 3) == 3) ? d0 + 1 : d0;
        const uint32_t u = __builtin_amdgcn_perm(r[row][d1], r[row][d0],
                                                 0x0c000c00u | ((((a & 3) == 3) ? 4u : (uint32_t)(a & 3) + 1u) << 16) | (uint32_t)(a & 3));
        s16x2 q; __builtin_memcpy(&q, &u, 4);
        return q;
    };
    const s16x2 vv = pick(ROW + 3, xc);
    s16x2 p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = pick(ROW + 3 + RY[i], xc + RX[i]);
#if MYSLAM_FAST_PHASE == 3
    { s16x2 x = vv ^ thv; for (int i = 0; i < 16; i++) x = x ^ p[i]; return x; }
#endif
    s16x2 n2[8], x2[8], n4[8], x4[8], ub[8], ud[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { n2[k] = pmin(p[2 * k], p[2 * k + 1]); x2[k] = pmax(p[2 * k], p[2 * k + 1]); }
#pragma unroll
    for (int k = 0; k < 8; k++) { n4[k] = pmin(n2[k], n2[(k + 1) & 7]); x4[k] = pmax(x2[k], x2[(k + 1) & 7]); }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const s16x2 a = p[(2 * k + 8) & 15], c = p[(2 * k + 15) & 15];
        ub[k] = pmin3(n4[k], n4[(k + 2) & 7], pmax(a, c));              // min over ring positions 2k .. 2k+7 and max(a, c)
        ud[k] = pmax3(x4[k], x4[(k + 2) & 7], pmin(a, c));
    }
    const s16x2 brt = pmax(pmax3(pmax3(ub[0], ub[1], ub[2]), pmax3(ub[3], ub[4], ub[5]), ub[6]), ub[7]);
    const s16x2 drk = pmin(pmin3(pmin3(ud[0], ud[1], ud[2]), pmin3(ud[3], ud[4], ud[5]), ud[6]), ud[7]);
    const s16x2 sraw = pmax(brt - vv, vv - drk);             // score + 1
    return pmax(sraw, thv) - thv;
}

// bytes A, A+1 (A = 0..3) of the dword pair (lo, hi) as two zero-extended 16-bit lanes
template <int A>
__device__ __forceinline__ s16x2 pick16(uint32_t lo, uint32_t hi) {
    const uint32_t u = __builtin_amdgcn_perm(hi, lo, 0x0c000c00u | ((uint32_t)(A + 1) << 16) | (uint32_t)A);
    s16x2 q; __builtin_memcpy(&q, &u, 4);
    return q;
}

// Necessary condition for a FAST-9 corner at threshold th, on two pixels at once: every arc of 9 contains two CYCLICALLY ADJACENT
// compass points of the ring (positions 0, 4, 8, 12 = (0,3), (3,0), (0,-3), (-3,0)), so a corner has an adjacent compass pair whose
// two pixels are both darker than v - th or both brighter than v + th (the in-tree isFastCorner pre-tests, ORBextractor.cpp:464-478,
// are the same idea on opposite pairs).  19 packed ops and 5 byte-pair picks against 77 + 17 for the score.  Returns true when either
// pixel of the pair passes.
__device__ __forceinline__ bool fast9_compass_pair(s16x2 v, s16x2 p0, s16x2 p4, s16x2 p8, s16x2 p12, s16x2 thv) {
    const s16x2 dmin = pmin(pmin(pmax(p0, p4), pmax(p4, p8)), pmin(pmax(p8, p12), pmax(p12, p0)));
    const s16x2 bmax = pmax(pmax(pmin(p0, p4), pmin(p4, p8)), pmax(pmin(p8, p12), pmin(p12, p0)));
    const s16x2 t1 = (v - thv) - dmin;                       // > 0  <=>  some adjacent compass pair is darker than v - th
    const s16x2 t2 = bmax - (v + thv);                       // > 0  <=>  some adjacent compass pair is brighter than v + th
    const s16x2 zero = {0, 0};
    const s16x2 m = pmax(pmax(t1, t2), zero);
    uint32_t u; __builtin_memcpy(&u, &m, 4);
    return u != 0;
}

// 3x3 strict-maximum test for pixels (2 PAIR, 2 PAIR + 1) of the lane's four, on 16-bit lanes: 9 byte-pair picks,
// 7 packed max, one packed subtract.  m = 3 rows x 12 bytes of the score map, the lane's pixels at bytes 4..7.
// Returns bit 0 / bit 1 = pixel is a strict maximum (which implies its score is non-zero); zc = the two centre scores.
template <int PAIR, int ROW = 0, int NR = 3>
__device__ __forceinline__ int nms_pair(const uint32_t (&m)[NR][3], uint32_t& zc) {
    constexpr int x = 4 + 2 * PAIR;
    auto pick = [&](int row, int a) -> s16x2 {
        const int d0 = a >> 2, d1 = ((a & 3) == 3) ? d0 + 1 : d0;
        const uint32_t u = __builtin_amdgcn_perm(m[row][d1], m[row][d0],
                                                 0x0c000c00u | ((((a & 3) == 3) ? 4u : (uint32_t)(a & 3) + 1u) << 16) | (uint32_t)(a & 3));
        s16x2 q; __builtin_memcpy(&q, &u, 4);
        return q;
    };
    const s16x2 c = pick(ROW + 1, x);
    s16x2 nb = pmax(pick(ROW, x - 1), pick(ROW, x));
    nb = pmax(nb, pick(ROW, x + 1));
    nb = pmax(nb, pmax(pick(ROW + 1, x - 1), pick(ROW + 1, x + 1)));
    nb = pmax(nb, pmax(pick(ROW + 2, x - 1), pick(ROW + 2, x)));
    nb = pmax(nb, pick(ROW + 2, x + 1));
    const s16x2 d = c - nb;
    __builtin_memcpy(&zc, &c, 4);
    return ((int)d.x > 0 ? 1 : 0) | ((int)d.y > 0 ? 2 : 0);
}

// Dense-path NMS, separable form.  One score-map row of a lane = 3 dwords (pixels -4..-1 | 0..3 | 4..7, the lane's four in the middle).
// With the five pairs of adjacent pixels P(-1) .. P3 (zero-extended bytes in 16-bit lanes), pixel pair A = (0, 1) has left = P(-1),
// centre = P0, right = P1 and pair B = (2, 3) has P1, P2, P3:  lr = max(left, right), h = max(lr, centre).  The largest of the 8
// neighbours of a pixel in row M is then max(h[M-1], h[M+1], lr[M]): 5 picks + 4 packed max per row and 4 per output row,
// against 9 picks + 7 max per pixel pair when every pair gathers its own neighbourhood.
struct NmsRow { s16x2 hA, hB, lrA, lrB, cA, cB; };
__device__ __forceinline__ NmsRow nms_row(uint32_t d0, uint32_t d1, uint32_t d2) {
    const s16x2 pm1 = pick16<3>(d0, d1), p0 = pick16<0>(d1, d1), p1 = pick16<1>(d1, d1), p2 = pick16<2>(d1, d1), p3 = pick16<3>(d1, d2);
    NmsRow r;
    r.lrA = pmax(pm1, p1); r.hA = pmax(r.lrA, p0);
    r.lrB = pmax(p1, p3);  r.hB = pmax(r.lrB, p2);
    r.cA = p0; r.cB = p2;
    return r;
}
// rows that only lend their neighbourhood maxima (the row above and the row below a 4-row block): h alone, one three-input maximum per pixel pair (round 6)
struct NmsRowH { s16x2 hA, hB; };
__device__ __forceinline__ NmsRowH nms_row_h(uint32_t d0, uint32_t d1, uint32_t d2) {
    const s16x2 pm1 = pick16<3>(d0, d1), p0 = pick16<0>(d1, d1), p1 = pick16<1>(d1, d1), p2 = pick16<2>(d1, d1), p3 = pick16<3>(d1, d2);
    NmsRowH r;
    r.hA = pmax3(pm1, p0, p1); r.hB = pmax3(p1, p2, p3);
    return r;
}
// the four z bytes of row M with everything that is not a STRICT 3x3 maximum set to zero: t = sat(c - neighbours) is non-zero exactly
// at strict maxima and t << 8 >= 256 > c there, so min(c, t << 8) keeps c at maxima and gives 0 elsewhere.  zacc collects the largest
// surviving z of the caller's rows (two 16-bit lanes).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// (U / D: any record with hA, hB — a full NmsRow or the h-only form.  The three-input maxima are v_pk_maximum3_f16 on zero-extended bytes, as in the scoring network.)
template <class RU, class RD>
__device__ __forceinline__ uint32_t nms_strict4(const RU& U, const NmsRow& M, const RD& D, u16x2& zacc) {
    auto keep = [](s16x2 c, s16x2 nb) -> s16x2 {
        u16x2 cu, nu; __builtin_memcpy(&cu, &c, 4); __builtin_memcpy(&nu, &nb, 4);
        const u16x2 t = __builtin_elementwise_sub_sat(cu, nu);
        const u16x2 k = __builtin_elementwise_min(cu, (u16x2)(t << (unsigned short)8));
        s16x2 ks; __builtin_memcpy(&ks, &k, 4);
        return ks;
    };
    const s16x2 ka = keep(M.cA, pmax3(U.hA, D.hA, M.lrA)), kb = keep(M.cB, pmax3(U.hB, D.hB, M.lrB));
    s16x2 zs; __builtin_memcpy(&zs, &zacc, 4);
    zs = pmax3(zs, ka, kb);                                             // surviving z are bytes: the f16 order is the integer order
    __builtin_memcpy(&zacc, &zs, 4);
    uint32_t za, zb; __builtin_memcpy(&za, &ka, 4); __builtin_memcpy(&zb, &kb, 4);
    return __builtin_amdgcn_perm(zb, za, 0x06040200u);
}

// inclusive prefix sum over the 64 lanes on the DPP network (row_shr 1/2/4/8, then the two row broadcasts)
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);        // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);        // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);        // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);        // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);        // row_bcast15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);        // row_bcast31 -> rows 2, 3
    return v;
}

// number of set bits of a 64-bit ballot below this lane, added to acc: two v_mbcnt (the mask-and-popcount form costs five)
__device__ __forceinline__ int rank_below(unsigned long long ballot, int acc) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, (uint32_t)acc));
}

// Path selection of the strip kernel (per level): `prev` = what the previous launch of this extractor handle measured,
// `cur` = what this launch accumulates (zeroed by the host): [level][4] = {pixel pairs that survived the pre-test (two-phase path) or
// 4-pixel rows holding a corner (dense path), pixel pairs looked at, path used, -}.  force: -1 = choose, 0 = two-phase, 1 = dense.
// Both paths produce the same candidate SET, so the choice only moves time.  Measured on MI355X (ms per 512 images of 1241 x 376,
// two-phase / dense, against the fraction of the pixel pairs that survive the pre-test; tools/fast_path_table.py prints the fractions,
// bench.py --fast-mode the times): 0.14: 1.00 / 1.41, 0.29: 1.26 / 1.42, 0.45: 1.53 / 1.44, 0.54: 1.68 / 1.45, 0.62: 1.80 / 1.46,
// 0.76: 2.02 / 1.48 — the two-phase path costs 0.78 + 1.63 s, break-even at s = 0.40.  The switch goes to dense above 0.41 surviving
// pairs and back when fewer than 0.165 corner rows per pixel pair are seen (the dense path's own statistic, free on the scalar unit:
// 0.174 at the break-even).
struct FastCtl { const uint32_t* prev; uint32_t* cur; int force; };

// ---- strip kernel: one block scores G horizontally adjacent cells -----------------------------------
// NMS and the 20 -> 7 fallback never cross a cell border; the ROI rows of the G cells are staged once, every cell at its own 16-byte
// aligned LDS offset (unaligned 16-byte global loads), so that a lane's register windows are dword-aligned; block dispatch, barriers
// and the global append are amortised over G cells.
//   two-phase path:  1. compass pre-test on every pixel (work item = 4 pixels x 2 rows from a 6-row x 12-byte register window),
//                       surviving pixel PAIRS are ballot-compacted into an LDS list (one LDS atomic per wave and iteration);
//                    2. one lane per listed pair: 7 x 8-byte window, packed score, 16-bit store into the score map;
//                    3. one lane per listed pair: 3x3 strict-maximum test on the score map.
//   dense path:      every pixel is scored (work item = 4 pixels x 2 rows: the two 7-row windows share 6 rows); NMS on 4 x 4 blocks in
//                    separable form; every 4-pixel row with a strict maximum leaves one list record, expanded by the append phase.
// Blocks are handed out XCD-aware: consecutive strips (which share halo rows and columns) go to the same XCD's L2.
// CW / CH = the widest / tallest grid cell of the plan (columns cost LDS in 16-byte steps, rows one by one)
#ifndef MYSLAM_FAST_LDS_BLOCK                                  // A/B builds (tools/build_variants.sh): another block footprint, e.g. 24300 = just over 160 KB / 7
#define MYSLAM_FAST_LDS_BLOCK (163840 / 6 - 128)
#endif
#ifndef MYSLAM_FAST_BLOCKS_PER_CU                              // A/B builds that set another MYSLAM_FAST_LDS_BLOCK say how many blocks it is meant to admit
#define MYSLAM_FAST_BLOCKS_PER_CU 6
#endif
// LDS a block of k_fast_strip<CW, G, CH> declares (an upper estimate: the arrays below plus their alignment) and the PAD that brings the block to
// MYSLAM_FAST_LDS_BLOCK.  The pad is DYNAMIC LDS, passed by the launcher (round 6): as a static array it made the compiler conclude "six waves per SIMD at
// most" and — to honour that bound by registers as well — raise the kernel's register allocation from its 68 to 73, i.e. from 72 to 80 registers per lane
// in the hardware's granules of 8.  Alone that costs nothing (LDS admits six blocks either way); under the pipeline the blocks of the other streams hold
// part of every CU's register file and the 8 registers decide how many FAST blocks fit beside them (beside two descriptor blocks: 5 instead of 4).
// NS > 1 (round 6, the multi-strip form): a second tile buffer and a second set of the strip's counters; such blocks are sized for MYSLAM_FAST_MS_BLOCKS_PER_CU per CU
#ifndef MYSLAM_FAST_MS_BLOCKS_PER_CU
#define MYSLAM_FAST_MS_BLOCKS_PER_CU 5
#endif
template <int CW, int G, int CH, int NS = 1>
struct FastLds {
    static constexpr int CP = (CW + 6 + 15) & ~15, TP = G * CP, TROWS = CH + 6 + 1, SP = (CW + 4 + 8 + 3) & ~3, SROWS = CH + 2 + 4;
    static constexpr int NPAIR = G * CH * ((CW + 1) / 2);
    static constexpr int EST = (NS > 1 ? 2 : 1) * (TROWS * TP + 16) + G * (SROWS * SP + 16) + 2 * NPAIR + (NS > 1 ? 128 : 64);
    static constexpr int BLOCKS = NS > 1 ? MYSLAM_FAST_MS_BLOCKS_PER_CU : MYSLAM_FAST_BLOCKS_PER_CU;
    static constexpr int BLOCK = NS > 1 ? 163840 / MYSLAM_FAST_MS_BLOCKS_PER_CU - 128 : MYSLAM_FAST_LDS_BLOCK;
    static constexpr int PAD = (CW <= 32 && EST < (NS > 1 ? BLOCK : 163840 / 7)) ? BLOCK - EST : 0;
    static_assert(CW > 32 || (BLOCKS * (EST + PAD) <= 163840 && (BLOCKS + 1) * (EST + PAD) > 163840), "exactly BLOCKS blocks per CU");
};
#ifndef MYSLAM_FAST_WAVES_PER_EU
#define MYSLAM_FAST_WAVES_PER_EU 6
#endif
#ifndef MYSLAM_FAST_NS                                          // strips per block of the grid-FAST kernel (1 = one block per strip; > 1 = the multi-strip form for large launches)
#define MYSLAM_FAST_NS 1
#endif
#ifdef MYSLAM_FAST_NUM_VGPR
#define MYSLAM_FAST_VGPR_ATTR __attribute__((amdgpu_num_vgpr(MYSLAM_FAST_NUM_VGPR)))
#else
#define MYSLAM_FAST_VGPR_ATTR
#endif
template <int CW, int G, int CH = CW, int NS = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CW <= 40 ? MYSLAM_FAST_WAVES_PER_EU : 3))) MYSLAM_FAST_VGPR_ATTR void k_fast_strip(OrbPlan P, const uint8_t* __restrict__ pyr, size_t pyrStride,
                                                    const uint8_t* __restrict__ maskPyr,
                                                    uint32_t* __restrict__ cand, int32_t* __restrict__ candCount, FastCtl ctl, int batch) {
    constexpr int T = 256;
    constexpr int CP = (CW + 6 + 15) & ~15;                            // every cell's ROI (cell + 6 halo columns) is staged at its own 16-byte aligned offset:
    constexpr int TP = G * CP;                                         //   a lane's 12-byte windows are then dword-aligned and need no byte alignment
    constexpr int NQC = CP / 16;                                       // 16-byte groups per cell row
    constexpr int TROWS = CH + 6 + 1;                                  // + 1: the second row of a work item reads one row further
    constexpr int SP = (CW + 4 + 8 + 3) & ~3;
    constexpr int SROWS = CH + 2 + 4;                                  // + 4: the dense path's NMS reads whole 4-row blocks (rows past the cell stay zero)
    constexpr int NLIST = G * ((CW + 1) / 2) * ((CH + 1) / 2);         // a cell of a x b pixels holds at most ceil(a/2) ceil(b/2) strict maxima
    constexpr int NPAIR = G * CH * ((CW + 1) / 2);                     // pixel pairs of a strip
    static_assert(CW <= 63 && CH <= 63 && G <= 4, "pair list entry = cell << 11 | row << 5 | pair index");
    // NS > 1 (the multi-strip form, see the loop below): two tile buffers — the next strip's tile lands in one by LDS-DMA while the current strip is worked on in the
    // other — and two sets of the strip's small state (a wave that has nothing to append runs ahead into the next strip's set-up)
    constexpr int NB2 = NS > 1 ? 2 : 1;
    static_assert(NS == 1 || (CW <= 32 && G == 4 && MYSLAM_FAST_PHASE == 0), "the multi-strip form exists for the usual plans' instance only");
    __shared__ __attribute__((aligned(16))) uint8_t s_tile0[TROWS * TP + 16];
    // (the second tile buffer of the multi-strip form is DYNAMIC LDS, as the pad is: the compiler sizes the kernel's register allocation by the occupancy its STATIC
    // LDS admits — with 31 KB declared it concludes "five waves per SIMD at most" and spends 83 registers where the block's co-runners leave room for 72)
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ __attribute__((aligned(16))) uint8_t s_score[G][SROWS * SP + 16];
    __shared__ uint16_t s_pairs[NPAIR];
    // strict maxima of the strip (py<<20 | px<<8 | z) and their cells: the list lives in the tile's LDS, which is dead once the
    // scores are computed (5 bytes per entry, NLIST entries always fit: checked below)
    static_assert(NLIST * 5 <= TROWS * TP, "maxima list must fit into the staged tile");
    // dense path: one record per 4-pixel ROW that holds a strict maximum = its NMS-filtered z dword (s_recz, in the dead tile) + the
    // row's (cell << 10 | row << 4 | group) (s_pairs, which only the two-phase path uses otherwise).  A cell of a x b pixels has at
    // most ceil(a/2) ceil(b/2) strict maxima, hence at most NLIST such rows.
    static_assert(NLIST * 4 <= TROWS * TP && NLIST <= NPAIR, "row records must fit");
    __shared__ int s_ini2[NB2][G], s_wc2[NB2][G], s_cnt2[NB2][4];      // [.][0] = records / maxima listed, [1] = pairs listed, [2] = corner rows (the path statistic)
    // LDS footprint on purpose (1241 x 376: cells of at most 32 x 40 pixels): a block of this instance needs 22.4 KB, seven would fit a CU — and
    // with seven waves per SIMD FAST holds 504 of the 512 registers per lane (measured: FAST alone 3 % faster, the pipelined step 1 % slower).
    // Six blocks it is; the question is what the remaining LDS of a CU is open to.  History: 25.1 KB per block (9 KB free: only blur / resize
    // blocks fitted) -> round 3: just over 160 KB / 7 = 23.7 KB (22 KB free: a descriptor, conv1 or conv2 block moves in beside six FAST
    // blocks, +1.5 % then) -> end of round 4: 160 KB / 6 - 128 = 27.2 KB, NOTHING with LDS beside six FAST blocks.  Since the descriptor
    // kernel runs as a limited grid and the oct-tree kernel no longer ends on its long blocks, the kernels of the other streams find their
    // CUs where FAST blocks retire; a block that squeezes in beside six FAST blocks only slows the launch the whole step waits for
    // (block size 27.2 / 26.0 / 25.0 / 24.4 / 23.7 KB: 75.0 / 74.3 / 74.5 / 73.9 / 74.3 k frames/s, two runs each on one box).
    static_assert(FastLds<CW, G, CH, NS>::CP == CP && FastLds<CW, G, CH, NS>::TROWS == TROWS && FastLds<CW, G, CH, NS>::SP == SP && FastLds<CW, G, CH, NS>::SROWS == SROWS && FastLds<CW, G, CH, NS>::NPAIR == NPAIR,
                  "FastLds restates this kernel's LDS arrays");            // (the pad itself: dynamic LDS of FastLds<>::PAD bytes, see there)

    // XCD-aware block order (bijective remap of the 1-D grid): the dispatcher places block i on XCD i % 8 and every XCD has a private
    // L2; logical ids (image-major, strips row by row) are handed out so that each XCD walks a contiguous range of strips, whose
    // shared halo rows / columns then hit in that L2 instead of being fetched once per XCD.
    // (Round 6 tried a PERSISTENT grid — n blocks per CU, each walking several strips: the FAST launch got 25 % shorter under the pipeline and the
    // step no shorter, and the loop cost the kernel 26 spilled registers; profiles/r06_ab_fast_persistent_grid.json.  One block per strip it is.)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    MYSLAM_BT(0);
    // Round 6: the per-block trace (profiles/r06_fast_block_phases_alone.json) showed a block spending 2.1 of its 9.9 us between its start and its first tile
    // load: ~19 dependent scalar loads out of the kernel arguments (one per level of a level search, the level's fields, the call's scalars wherever a branch first
    // needed them) and ~250 scalar instructions deriving the strip's geometry — in each of the block's four waves.  Now the head is TWO dependent loads: the call's
    // scalars, then the strip's 32-byte record of the plan's per-strip table (orb_engine.hip make_plan); the level's full record follows for the later phases.
    int a_nstrips = P.nstrips, a_batch = batch, a_ext0N = P.ext0N, a_ext0Pitch = P.ext0Pitch;
    size_t a_ext0Stride = P.ext0Stride, a_pyrStride = pyrStride;
    // (pointers travel as integers through the pin and come back as GLOBAL / CONSTANT address-space pointers: a laundered generic pointer is loaded from with flat_ instructions)
    uintptr_t a_ext0i = (uintptr_t)P.ext0, a_pyri = (uintptr_t)pyr, a_tabi = (uintptr_t)P.stripTab;
    asm volatile("" : "+s"(a_nstrips), "+s"(a_batch), "+s"(a_ext0N), "+s"(a_ext0Pitch), "+s"(a_ext0Stride), "+s"(a_pyrStride), "+s"(a_ext0i), "+s"(a_pyri), "+s"(a_tabi));
    typedef const uint8_t __attribute__((address_space(1))) * GlobalU8;
    const GlobalU8 a_ext0 = (GlobalU8)a_ext0i, a_pyr = (GlobalU8)a_pyri;
    // ---- the strips of this block: one (NS == 1), or NS consecutive ones (the multi-strip form, round 6) ----
    // Multi-strip form: a FAST block lives ~8.3 us of which ~2.2 are the latency of its tile's global loads, and under the pipeline FAST holds only 2 - 3 blocks per CU
    // (the co-runners' registers), too few to hide it.  Here a block walks NS consecutive strips and the NEXT strip's tile lands in a second LDS buffer by LDS-DMA
    // (global_load_lds_dwordx4: no registers for the data) while the current strip is scored: only the first strip of a block waits for memory.
    const int total_strips = a_nstrips * a_batch;
    const int gs_first = logical * NS, gs_end = min(gs_first + NS, total_strips);
    if (gs_first >= total_strips) return;
    typedef const uint32_t __attribute__((address_space(4))) * ConstU32;
    // strip gs -> its image and the 8 dwords of its record in the plan's per-strip table (scalar: ONE s_load_dwordx8 through the constant address space)
    struct StripRec { uint32_t t[8]; int b; };
    auto load_rec = [&](int gs) __attribute__((always_inline)) -> StripRec {
        StripRec r;
        r.b = gs / a_nstrips;
        const int sidx = gs - r.b * a_nstrips;
        const ConstU32 tb = (ConstU32)(a_tabi + (uintptr_t)sidx * 32);
        uint32_t t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3], t4 = tb[4], t5 = tb[5], t6 = tb[6], t7 = tb[7];
        asm volatile("" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+s"(t4), "+s"(t5), "+s"(t6), "+s"(t7));      // all eight before the first branch: ONE load, one wait
        r.t[0] = t0; r.t[1] = t1; r.t[2] = t2; r.t[3] = t3; r.t[4] = t4; r.t[5] = t5; r.t[6] = t6; r.t[7] = t7;
        return r;
    };
    // multi-strip form: the tile of strip `rec` -> tile buffer q by LDS-DMA.  Lane i of a load writes LDS at (wave-uniform base) + 16 i, and a tile row is 12 pieces of
    // 16 bytes (4 cells x 3), so piece p = 256 j + thread lies at 16 p: row p / 12, cell (p % 12) / 3, 16-byte group p % 3 — the global address is per lane.  Rows below the
    // ROI, cells the strip does not have and columns past the row are CLAMPED to valid memory instead of staged as zeros: nothing ever reads them unmasked.
    auto issue_tile = [&](const StripRec& r, int q, int tid) __attribute__((always_inline)) {
        const int level = (int)(r.t[0] & 0xffu), hr = (int)(r.t[1] >> 16), iniY = (int)(r.t[1] & 0xffffu), iniX0 = (int)(r.t[2] & 0xffffu), wCell = (int)(r.t[2] >> 16);
        const bool ext = level == 0 && r.b < a_ext0N;
        const uint8_t* img = (const uint8_t*)(ext ? a_ext0 + (size_t)r.b * a_ext0Stride : a_pyr + (size_t)r.b * a_pyrStride + ((size_t)r.t[6] | ((size_t)r.t[7] << 32)));
        const int ipitch = ext ? a_ext0Pitch : (int)r.t[4];
        const uint32_t lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(q ? &s_dyn[0] : &s_tile0[0])) + 1024u * (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int j = 0; j < (TROWS * 12 + 255) / 256; j++) {
            const int pc = 256 * j + tid;
            const int rr = pc / 12, col = pc - 12 * rr, c = col / 3, k = col - 3 * c;
            const int x = min(iniX0 + c * wCell + 16 * k, ipitch - 16);
            const uint8_t* src = img + (size_t)(iniY + min(rr, hr - 1)) * ipitch + x;
            const uint32_t d = lds + 4096u * (uint32_t)j;
            uint32_t keep;
            if (pc < TROWS * 12)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(d) : "memory");
        }
    };
    StripRec rec = load_rec(gs_first);
    if constexpr (NS > 1) {
        static_assert(NS == 1 || (G * ((CW + 6 + 15) / 16) == 12), "a tile row of the multi-strip form is 12 pieces");
        if ((rec.t[0] >> 24) != 0) issue_tile(rec, 0, (int)threadIdx.x);
    }
    int par = 0;                                                       // tile buffer / state set of the current strip (NS > 1)
    bool fresh = true;                                                 // the current strip's tile was issued after this wave's last wait for its loads (NS > 1, wave-uniform)
  for (int gs = gs_first; gs < gs_end; gs++) {
    int tid_ = threadIdx.x;
    if constexpr (NS > 1) asm volatile("" : "+v"(tid_));               // (inside a loop the compiler would hoist every lane-dependent invariant of the body into registers of its own)
    const int tid = tid_;
    uint8_t* const s_tile = (NS > 1 && par) ? s_dyn : s_tile0;
    uint32_t* const s_list = reinterpret_cast<uint32_t*>(s_tile);
    uint8_t* const s_listc = s_tile + 4 * NLIST;
    uint32_t* const s_recz = reinterpret_cast<uint32_t*>(s_tile);
    int* const s_ini = s_ini2[par];
    int* const s_wc = s_wc2[par];
    int& s_nlist = s_cnt2[par][0]; int& s_npair = s_cnt2[par][1]; int& s_ncorner = s_cnt2[par][2];
    const int b = rec.b;
    StripRec rec_next = rec;
    if constexpr (NS > 1) { if (gs + 1 < gs_end) rec_next = load_rec(gs + 1); }      // scalar loads, in flight under this strip's work
    // what the next launch of this handle decides on is reported by a SAMPLE of the strips (a ratio of sums needs no more, and a few thousand
    // same-address atomics per launch cost nothing where 300 k of them serialise into milliseconds)
    const bool sampled = gs % max(1, total_strips >> 12) == 0;
    const uint4 e0 = make_uint4(rec.t[0], rec.t[1], rec.t[2], rec.t[3]), e1 = make_uint4(rec.t[4], rec.t[5], rec.t[6], rec.t[7]);
    const int level = (int)(e0.x & 0xffu), ci = (int)((e0.x >> 8) & 0xffu), cj0 = (int)((e0.x >> 16) & 0xffu), ncell = (int)(e0.x >> 24);
    if (ncell == 0) { rec = rec_next; if constexpr (NS > 1) { if (gs + 1 < gs_end && (rec.t[0] >> 24) != 0) { issue_tile(rec, par, tid); fresh = true; } } continue; }      // :843 / :852 (decided when the plan was made); nothing of this strip was in flight
    const int iniY = (int)(e0.y & 0xffffu), hr = (int)(e0.y >> 16), hc = hr - 6;
    const int iniX0 = (int)(e0.z & 0xffffu), t_wCell = (int)(e0.z >> 16);
    const int pairs_total = (int)e1.y;
    if (tid < G) s_wc[tid] = (int)((e0.w >> (8 * tid)) & 0xffu);      // per-cell widths are looked up from LDS (a register array would be indexed dynamically)
    auto wc_of = [&](int c) __attribute__((always_inline)) -> int { return s_wc[c]; };
    const bool ext = level == 0 && b < a_ext0N;                      // level 0 read in place (block-uniform)
    const uint8_t* img = (const uint8_t*)(ext ? a_ext0 + (size_t)b * a_ext0Stride : a_pyr + (size_t)b * a_pyrStride + ((size_t)e1.z | ((size_t)e1.w << 32)));
    const int ipitch = ext ? a_ext0Pitch : (int)e1.x;
    const LevelGeom g = P.lv[level];                                   // for the phases behind the tile loads (coordinates, list capacity, mask plane)
    // everything of the head that does not need the tile — clearing the score maps, the strip's counters, the path decision (three scalar loads of the previous
    // launch's statistics) — sits BETWEEN the issue of the tile's global loads and their stores to LDS (round 6: it used to follow the stores, i.e. the loads' latency)
    bool dense = false;
    auto prep = [&]() __attribute__((always_inline)) {
        {   // zero the score maps (16-byte stores; the dword tail only exists when the array size is not a multiple of 16)
            constexpr int NB = G * (SROWS * SP + 16), NV = NB / 16;
            for (int i = tid; i < NV; i += T) reinterpret_cast<uint4*>(&s_score[0][0])[i] = make_uint4(0, 0, 0, 0);
            if (NB % 16 != 0 && tid < (NB - 16 * NV) / 4) reinterpret_cast<uint32_t*>(&s_score[0][0])[4 * NV + tid] = 0;
        }
        if (tid < G) s_ini[tid] = 0;
        if (tid == 0) { s_nlist = 0; s_npair = 0; s_ncorner = 0; }
        // path of this launch: what the previous launch of the handle saw on this level decides (block-uniform)
        {
            // (scalar loads through the constant address space — the record was written by the handle's PREVIOUS launch: as a vector load its wait (vmcnt) also
            // waited for the tile loads issued just before)
            typedef const uint32_t __attribute__((address_space(4))) * ConstStat;
            const ConstStat pv = (ConstStat)((uintptr_t)ctl.prev + (uintptr_t)level * 16);
            const uint32_t pv0 = pv[0], pv1 = pv[1], pv2 = pv[2];
            const float ps = (float)pv0, pt = (float)pv1;
            const bool was_dense = pv2 != 0;
            dense = ctl.force >= 0 ? ctl.force != 0 : (pt > 0.f && (was_dense ? ps > 0.165f * pt : ps > 0.41f * pt));
        }
    };
    MYSLAM_BT_MARK(3);                                                 // (trace builds: the block's decode is done, its tile loads start here)
    if constexpr (NS > 1) {
        // this strip's tile is in flight (the block's first) or has landed (every wave waited for its own pieces before the previous strip's append)
        prep();
        if (fresh) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); fresh = false; }
        __syncthreads();
        // the next strip's tile -> the other buffer: the strip before this one kept its record list there, and every wave has left that strip (the barrier)
        if (gs + 1 < gs_end && (rec_next.t[0] >> 24) != 0) issue_tile(rec_next, par ^ 1, tid);
    } else if constexpr (G * NQC <= 16) {
        // all loads of a thread are issued before its LDS stores.  Unaligned 16-byte loads (a cell starts at any column); a load may run up
        // to 15 bytes past its row or, in the last row of the last plane, into the slack behind the pyramid block — those bytes are never used.
        // 16 lanes per tile row (G * NQC of them busy): (row, cell, 16-byte group) of a lane are bit fields of the thread index and its rows
        // are 16 apart — no division, one address increment per load (round 5: the index arithmetic of three loads was 66 of the ~120 vector
        // instructions a wave spent before the first barrier)
        constexpr int CPR = G * NQC, NIT = (TROWS + 15) / 16;
        const int col = threadIdx.x & 15, r0 = threadIdx.x >> 4;
        const int c = col / NQC, k = col - c * NQC;
        const int x = iniX0 + c * t_wCell + 16 * k;
        const bool lane_on = col < CPR && c < ncell, in_row = x < ipitch;
        const uint8_t* src = img + (size_t)(iniY + r0) * ipitch + x;
        uint4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; u++) {
            if (lane_on && in_row && r0 + 16 * u < hr) {
                uint4 t;
                __builtin_memcpy(&t, src + (size_t)(16 * u) * ipitch, 16);
                v[u] = t;
            } else v[u] = make_uint4(0, 0, 0, 0);
        }
        prep();
#pragma unroll
        for (int u = 0; u < NIT; u++)
            if (lane_on && r0 + 16 * u < hr) *reinterpret_cast<uint4*>(&s_tile[(r0 + 16 * u) * TP + 16 * col]) = v[u];      // c * CP + 16 k = 16 col
        __syncthreads();
    } else {
        constexpr int NIT = (TROWS * G * NQC + T - 1) / T;
        uint4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; u++) {
            const int i = threadIdx.x + u * T, r = i / (G * NQC), rem = i - r * (G * NQC), c = rem / NQC, k = rem - c * NQC;
            const int x = iniX0 + c * t_wCell + 16 * k;
            if (r < hr && c < ncell && x < ipitch) {
                const uint8_t* src = img + (size_t)(iniY + r) * ipitch + x;
                uint4 t;
                __builtin_memcpy(&t, src, 16);
                v[u] = t;
            } else v[u] = make_uint4(0, 0, 0, 0);
        }
        prep();
#pragma unroll
        for (int u = 0; u < NIT; u++) {
            const int i = threadIdx.x + u * T, r = i / (G * NQC), rem = i - r * (G * NQC), c = rem / NQC, k = rem - c * NQC;
            if (r < hr && c < ncell) *reinterpret_cast<uint4*>(&s_tile[r * TP + c * CP + 16 * k]) = v[u];
        }
        __syncthreads();
    }
    MYSLAM_BT_MARK(0);
#if MYSLAM_FAST_PHASE == 1
    if (dense) return;
#endif

    const s16x2 thv = {(short)P.minTh, (short)P.minTh};
    const int zoff = P.minTh - 1;                                      // the score map holds z = score - zoff
    // work item = 4 pixels x 2 rows
    const int hc2 = (hc + 1) >> 1;
    const int ngr = (g.wCell + 3) >> 2, per_cell = ngr * hc2, nitems = ncell * per_cell;
    // q -> (cell c, row pair cy2, 4-pixel group gi) without integer division: q < 2^12, so a float reciprocal + one fix-up is exact.
    // (Used by the two-phase path and by plans with cells wider than 32 pixels; the dense path of the usual plans maps by bit fields, below.)
    auto split = [&](int q, float inv_pc, float inv_ngr, int& c, int& cy, int& gi) __attribute__((always_inline)) {
        c = (int)(((float)q + 0.5f) * inv_pc);
        int rem = q - c * per_cell;
        if (rem < 0) { c--; rem += per_cell; } else if (rem >= per_cell) { c++; rem -= per_cell; }
        cy = (int)(((float)rem + 0.5f) * inv_ngr);
        gi = rem - cy * ngr;
        if (gi < 0) { cy--; gi += ngr; } else if (gi >= ngr) { cy++; gi -= ngr; }
    };
    const int lane = (unsigned)tid & 63;
    // append the strict maxima a lane found (bit k of mk = pixel k of its group, z of pixel k at bits [8 sh k, 8 sh k + 8) of zc) to the
    // strip's LDS list: ONE returning LDS atomic per wave and call (DPP prefix sum over the lanes' counts).  Wave-uniform call sites only.
    auto push_maxima = [&](int mk, uint32_t zc0, uint32_t zc1, int zshift, int c, int px0, int py0) __attribute__((always_inline)) {
        int incl = __popc(mk);
        const int cnt = incl;
        incl = wave_incl_scan_dpp(incl);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0) return;                                        // wave-uniform
        int base = 0;
        if (lane == 63) base = atomicAdd(&s_nlist, total);
        base = __builtin_amdgcn_readlane(base, 63);
        if (mk) {
            int dst = base + incl - cnt, zmax = 0, bits = mk;
            while (bits) {
                const int k2 = __ffs(bits) - 1;
                bits &= bits - 1;
                const uint32_t z = (((k2 & 4) ? zc1 : zc0) >> (zshift * (k2 & 3))) & 0xff;
                if (dst < NLIST) { s_list[dst] = ((uint32_t)(py0 + (k2 >> 2)) << 20) | ((uint32_t)(px0 + (k2 & 3)) << 8) | z; s_listc[dst] = (uint8_t)c; }
                dst++;
                zmax = max(zmax, (int)z);
            }
            if (zmax + zoff >= P.iniTh) s_ini[c] = 1;
        }
    };

    // Cells of at most 32 x CH pixels in strips of 4 (every level of a 1241 x 376 plan): eight 4-pixel groups span a cell row, so an item
    // index is the bit fields  group | cell << 3 | row block << 5  — a lane keeps its (cell, group) for the whole launch, its items are
    // 8 row blocks apart, and the loops carry one add.  (Round 5, from the per-phase counter split profiles/r05_fast_phase_valu.json: the
    // float-reciprocal decomposition, its two divisions and the carry chain that advanced it cost ~150 of the ~1 300 vector instructions of a wave.)
    constexpr bool BITMAP = CW <= 32 && G == 4;
    if (!dense) {
        const float inv_pc = 1.0f / (float)per_cell, inv_ngr = 1.0f / (float)ngr;
        // ---- 1. compass pre-test, surviving pixel pairs -> s_pairs ----
        for (int q0 = 0; q0 < nitems; q0 += T) {                       // uniform trip count: the wave-wide scan needs every lane
            const int q = q0 + (unsigned)tid;
            int pm = 0, c = 0, cy = 0, cx = 0;                         // pm bit 2 r + k: pair k of row cy + r survives
            if (q < nitems) {
                int cy2, gi;
                split(q, inv_pc, inv_ngr, c, cy2, gi);
                cx = 4 * gi; cy = 2 * cy2;
                const int wc = wc_of(c);
                if (cx < wc) {
                    const uint8_t* base = &s_tile[cy * TP + c * CP + cx];     // ROI column cx of cell c: dword-aligned
                    uint32_t rc[2][3], ru[2][2], rd[2][2];             // centre rows (12 bytes), rows above / below the centres (8 bytes)
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const uint32_t* pu = reinterpret_cast<const uint32_t*>(base + (r + 0) * TP);
                        const uint32_t* pc = reinterpret_cast<const uint32_t*>(base + (r + 3) * TP);
                        const uint32_t* pd = reinterpret_cast<const uint32_t*>(base + (r + 6) * TP);
                        ru[r][0] = pu[0]; ru[r][1] = pu[1];
                        rc[r][0] = pc[0]; rc[r][1] = pc[1]; rc[r][2] = pc[2];
                        rd[r][0] = pd[0]; rd[r][1] = pd[1];
                    }
                    // pixel i of the group sits at window byte 3 + i: ring position 0 = (0, +3) in row + 6, 8 = (0, -3) in row + 0,
                    // 4 = (+3, 0) and 12 = (-3, 0) in the centre row.  Pair 0 = bytes (3, 4), pair 1 = bytes (5, 6).
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const uint32_t c0 = rc[r][0], c1 = rc[r][1], c2 = rc[r][2];
                        const bool a = fast9_compass_pair(pick16<3>(c0, c1), pick16<3>(rd[r][0], rd[r][1]), pick16<2>(c1, c1), pick16<3>(ru[r][0], ru[r][1]),
                                                          pick16<0>(c0, c0), thv);
                        const bool bb = fast9_compass_pair(pick16<1>(c1, c1), pick16<1>(rd[r][1], rd[r][1]), pick16<0>(c2, c2), pick16<1>(ru[r][1], ru[r][1]),
                                                           pick16<2>(c0, c0), thv);
                        pm |= (a ? 1 : 0) << (2 * r);
                        pm |= (bb ? 2 : 0) << (2 * r);
                    }
                    if (cx + 2 >= wc) pm &= 5;                         // second pair outside the cell interior
                    if (cy + 1 >= hc) pm &= 3;                         // second row outside
                }
            }
            int incl = __popc(pm);
            const int cnt = incl;
            incl = wave_incl_scan_dpp(incl);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (total == 0) continue;                                  // wave-uniform
            int base = 0;
            if (lane == 63) base = atomicAdd(&s_npair, total);
            base = __builtin_amdgcn_readlane(base, 63);
            int dst = base + incl - cnt;
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++)
                if (pm & (1 << k2)) s_pairs[dst++] = (uint16_t)((c << 11) | ((cy + (k2 >> 1)) << 5) | ((cx >> 1) + (k2 & 1)));
        }
        __syncthreads();
        const int np = min(s_npair, NPAIR);
        // ---- 2. score of the listed pairs ----
        for (int q = (unsigned)tid; q < np; q += T) {
            const uint32_t e = s_pairs[q];
            const int c = (int)(e >> 11), row = (int)((e >> 5) & 63), cx = 2 * (int)(e & 31);
            const int col = c * CP + cx;
            const uint32_t sh = (uint32_t)(col & 3);
            uint32_t r[7][3];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                const uint32_t* rp = reinterpret_cast<const uint32_t*>(&s_tile[(row + j) * TP + (col & ~3)]);
                const uint32_t w0 = rp[0], w1 = rp[1], w2 = rp[2];
                r[j][0] = __builtin_amdgcn_alignbyte(w1, w0, sh);
                r[j][1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
                r[j][2] = 0u;
            }
            const s16x2 z = fast9_score_pair<0, 0, 7>(r, thv);
            uint32_t u;
            __builtin_memcpy(&u, &z, 4);
            uint32_t zb = __builtin_amdgcn_perm(u, u, 0x0c0c0200u);      // the two z bytes
            if (cx + 1 >= wc_of(c)) zb &= 0xffu;
            if (zb) *reinterpret_cast<uint16_t*>(&s_score[c][(row + 1) * SP + 4 + cx]) = (uint16_t)zb;
        }
        __syncthreads();
        // ---- 3. NMS of the listed pairs ----
        for (int q0 = 0; q0 < np; q0 += T) {                           // uniform trip count
            const int q = q0 + (unsigned)tid;
            int mk = 0, c = 0, row = 0, cx = 0;
            uint32_t zc = 0;
            if (q < np) {
                const uint32_t e = s_pairs[q];
                c = (int)(e >> 11); row = (int)((e >> 5) & 63); cx = 2 * (int)(e & 31);
                if (*reinterpret_cast<const uint16_t*>(&s_score[c][(row + 1) * SP + 4 + cx]) != 0) {
                    const uint32_t sh2 = (uint32_t)(cx & 2);
                    uint32_t m[3][3];                                  // score-map rows row-1 .. row+1, the pair at bytes 4, 5
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const uint32_t* rp = reinterpret_cast<const uint32_t*>(&s_score[c][(row + j) * SP + (cx & ~3)]);
                        const uint32_t w0 = rp[0], w1 = rp[1], w2 = rp[2];
                        m[j][0] = __builtin_amdgcn_alignbyte(w1, w0, sh2);
                        m[j][1] = __builtin_amdgcn_alignbyte(w2, w1, sh2);
                        m[j][2] = 0u;
                    }
                    uint32_t z16;
                    mk = nms_pair<0, 0, 3>(m, z16);
                    zc = z16;
                }
            }
            push_maxima(mk, zc, 0u, 16, c, cx + 3 + (cj0 + c) * g.wCell, row + 3 + ci * g.hCell);
        }
    } else {
        // ---- dense path: score every pixel ----
        auto score_item = [&](int c, int cy, int cx, uint32_t keepm) __attribute__((always_inline)) {
            uint32_t r[8][3];                                              // tile rows cy .. cy+7 (row cy+7 may lie below the ROI: staged as zeros / unused)
#pragma unroll
            for (int j = 0; j < 8; j++) {                                  // ROI column cx of cell c: dword-aligned
                const uint32_t* rp = reinterpret_cast<const uint32_t*>(&s_tile[(cy + j) * TP + c * CP + cx]);
                r[j][0] = rp[0]; r[j][1] = rp[1]; r[j][2] = rp[2];
            }
#if MYSLAM_FAST_PHASE == 2
            {
                uint32_t x = 0;
                for (int j = 0; j < 8; j++) x ^= r[j][0] ^ r[j][1] ^ r[j][2];
                *reinterpret_cast<uint32_t*>(&s_score[c][(cy + 1) * SP + 4 + cx]) = x & keepm;
                if (cy + 1 < hc) *reinterpret_cast<uint32_t*>(&s_score[c][(cy + 2) * SP + 4 + cx]) = ~x & keepm;
                return;
            }
#endif
            {
                const s16x2 za = fast9_score_pair<0, 0, 8>(r, thv), zb = fast9_score_pair<1, 0, 8>(r, thv);
                uint32_t ua, ub;
                __builtin_memcpy(&ua, &za, 4); __builtin_memcpy(&ub, &zb, 4);
                const uint32_t z4 = __builtin_amdgcn_perm(ub, ua, 0x06040200u) & keepm;   // the four z bytes
                *reinterpret_cast<uint32_t*>(&s_score[c][(cy + 1) * SP + 4 + cx]) = z4;
            }
            if (cy + 1 < hc) {
                const s16x2 za = fast9_score_pair<0, 1, 8>(r, thv), zb = fast9_score_pair<1, 1, 8>(r, thv);
                uint32_t ua, ub;
                __builtin_memcpy(&ua, &za, 4); __builtin_memcpy(&ub, &zb, 4);
                const uint32_t z4 = __builtin_amdgcn_perm(ub, ua, 0x06040200u) & keepm;
                *reinterpret_cast<uint32_t*>(&s_score[c][(cy + 2) * SP + 4 + cx]) = z4;
            }
        };
        if constexpr (BITMAP) {
            const int gi = tid & 7, c = (tid >> 3) & 3, cx = 4 * gi;
            const int wc = wc_of(c);
            if (cx < wc) {
                const uint32_t keepm = (wc - cx < 4) ? (1u << (8 * (wc - cx))) - 1u : 0xffffffffu;
                for (int cy2 = tid >> 5; cy2 < hc2; cy2 += T / 32) score_item(c, 2 * cy2, cx, keepm);
            }
        } else {
            // a lane's work items are q = tid, tid + T, ...: (cell, row pair, group) is split once and then advanced by the split of T
            const float inv_pc = 1.0f / (float)per_cell, inv_ngr = 1.0f / (float)ngr;
            int c, cy2, gi;
            split(tid, inv_pc, inv_ngr, c, cy2, gi);
            const int dc = T / per_cell, drem = T - dc * per_cell, dcy = drem / ngr, dgi = drem - dcy * ngr;       // block-uniform
            auto advance = [&](int& c_, int& cy2_, int& gi_) __attribute__((always_inline)) {
                gi_ += dgi; cy2_ += dcy; c_ += dc;
                if (gi_ >= ngr) { gi_ -= ngr; cy2_++; }
                if (cy2_ >= hc2) { cy2_ -= hc2; c_++; }
            };
            for (int q = (unsigned)tid; q < nitems; q += T, advance(c, cy2, gi)) {
                const int wc = wc_of(c);
                const int cx = 4 * gi, cy = 2 * cy2;
                if (cx >= wc) continue;
                score_item(c, cy, cx, (wc - cx < 4) ? (1u << (8 * (wc - cx))) - 1u : 0xffffffffu);
            }
        }
        __syncthreads();
        MYSLAM_BT_MARK(1);
#if MYSLAM_FAST_PHASE >= 2 && MYSLAM_FAST_PHASE <= 4
        return;
#endif
        int nquad = 0;                     // 4-pixel rows that hold a corner: counted on the scalar unit (ballot + s_bcnt1), the statistic of this path
        // NMS on 4 x 4 pixel blocks (6 score-map rows x 3 dwords per lane, separable 3x3 maximum).  Every 4-pixel row that holds a strict
        // maximum leaves ONE record (its filtered z dword) in the strip's list — four ballots and one LDS atomic per wave and
        // iteration; the records are expanded into candidates by the append phase below, where every lane has work.
        const int hc4 = (hc + 3) >> 2, per_cell4 = ngr * hc4;
        const int nitems4 = BITMAP ? 32 * hc4 : ncell * per_cell4;         // BITMAP: item = group | cell << 3 | row block << 5 (lanes of absent cells idle)
        int c, gi, cy4;
        int ec = 0, ecy = 0, egi = 0;                                      // !BITMAP: the split of T (block-uniform)
        if constexpr (BITMAP) {
            gi = tid & 7; c = (tid >> 3) & 3; cy4 = tid >> 5;
        } else {   // q -> (cell, row block, group) as split() does for the scoring items
            const float inv_ngr = 1.0f / (float)ngr;
            const int q = tid;
            c = (int)(((float)q + 0.5f) * (1.0f / (float)per_cell4));
            int rem = q - c * per_cell4;
            if (rem < 0) { c--; rem += per_cell4; } else if (rem >= per_cell4) { c++; rem -= per_cell4; }
            cy4 = (int)(((float)rem + 0.5f) * inv_ngr);
            gi = rem - cy4 * ngr;
            if (gi < 0) { cy4--; gi += ngr; } else if (gi >= ngr) { cy4++; gi -= ngr; }
            ec = T / per_cell4; const int erem = T - ec * per_cell4; ecy = erem / ngr; egi = erem - ecy * ngr;
        }
        const int wc_lane = BITMAP ? wc_of(c) : 0;
        const int zini = P.iniTh - zoff;                                   // z of a corner at the initial threshold
        for (int q0 = 0; q0 < nitems4; q0 += T) {                          // uniform trip count: the ballot needs every lane
            const int q = q0 + (unsigned)tid;
            uint32_t zm[4] = {0u, 0u, 0u, 0u};                             // rows 4 cy4 .. + 3: z where the pixel is a strict maximum, else 0
            u16x2 zacc = {0, 0};
            if (BITMAP ? (cy4 < hc4 && 4 * gi < wc_lane) : (q < nitems4 && 4 * gi < wc_of(c))) {
                uint32_t m[6][3];                                          // score-map rows 4 cy4 - 1 .. 4 cy4 + 4 (zero border rows around the cell)
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const uint32_t* rp = reinterpret_cast<const uint32_t*>(&s_score[c][(4 * cy4 + j) * SP + 4 * gi]);
                    m[j][0] = rp[0]; m[j][1] = rp[1]; m[j][2] = rp[2];
                }
                // the path statistic (4-pixel rows that hold a corner) is only reported by a sample of the strips (below): only those count it
                if (sampled)                                               // block-uniform
                    nquad += __popcll(__ballot(m[1][1] != 0)) + __popcll(__ballot(m[2][1] != 0)) + __popcll(__ballot(m[3][1] != 0)) + __popcll(__ballot(m[4][1] != 0));
                if ((m[1][1] | m[2][1] | m[3][1] | m[4][1]) != 0) {        // else none of the 16 pixels is a corner (rows past the cell hold zeros)
                    const NmsRowH Rt = nms_row_h(m[0][0], m[0][1], m[0][2]), Rb = nms_row_h(m[5][0], m[5][1], m[5][2]);
                    NmsRow R[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) R[j] = nms_row(m[j + 1][0], m[j + 1][1], m[j + 1][2]);
                    zm[0] = nms_strict4(Rt, R[0], R[1], zacc);
                    zm[1] = nms_strict4(R[0], R[1], R[2], zacc);
                    zm[2] = nms_strict4(R[1], R[2], R[3], zacc);
                    zm[3] = nms_strict4(R[2], R[3], Rb, zacc);
                }
            }
            const unsigned long long b0 = __ballot(zm[0] != 0), b1 = __ballot(zm[1] != 0), b2 = __ballot(zm[2] != 0), b3 = __ballot(zm[3] != 0);
            const int total = (int)(__popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3));
            if (total != 0) {                                              // wave-uniform
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_nlist, total);
                base = __builtin_amdgcn_readfirstlane(base);
                const int code = (c << 10) | (cy4 << 6) | gi;              // row = 4 cy4 + j
                // lane-major order (a lane's rows stay together, lanes = blocks adjacent in x): the oct-tree kernel that consumes the
                // candidate list is measurably faster on spatially coherent input (table lookups, scatter coalescing).
                // No capacity test: a record is a 4-pixel row with a strict 3x3 maximum, strict maxima are never 8-neighbours, so a cell of
                // a x b pixels holds at most ceil(a/2) ceil(b/2) of them and a strip at most NLIST (static_assert above).
                int pos = rank_below(b0, rank_below(b1, rank_below(b2, rank_below(b3, base))));
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (zm[j] != 0) {
                        s_recz[pos] = zm[j]; s_pairs[pos] = (uint16_t)(code | (j << 4));
                        pos++;
                    }
                if (max((int)zacc.x, (int)zacc.y) >= zini) s_ini[c] = 1;
            }
            if constexpr (BITMAP) cy4 += T / 32;
            else {
                gi += egi; cy4 += ecy; c += ec;
                if (gi >= ngr) { gi -= ngr; cy4++; }
                if (cy4 >= hc4) { cy4 -= hc4; c++; }
            }
        }
        if (lane == 0 && nquad) atomicAdd(&s_ncorner, nquad);
#if MYSLAM_FAST_PHASE == 5
        return;
#endif
    }
    __syncthreads();
    MYSLAM_BT_MARK(2);
    // multi-strip form: the next strip's tile was requested a whole scoring pass ago — this wave's pieces have landed by now; waiting for them HERE, in front of the
    // append's own memory operations, keeps the wait from ever covering this strip's candidate stores
    if constexpr (NS > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((unsigned)tid == 0 && sampled) {                                 // the path statistics of this launch (see `sampled` above)
        atomicAdd(&ctl.cur[level * 4], (uint32_t)(dense ? s_ncorner : min(s_npair, NPAIR)));
        atomicAdd(&ctl.cur[level * 4 + 1], (uint32_t)pairs_total);
        ctl.cur[level * 4 + 2] = dense ? 1u : 0u;
    }
    // filter (:858-865: th 20 if the cell has any such corner, else th 7; mask :873-877) and append: one global atomic per wave
    const uint8_t* mimg = maskPyr ? maskPyr + (size_t)b * pyrStride + g.imgOff : nullptr;
    uint32_t* out = cand + (size_t)b * P.totalKeyCap + g.keyOff;
    if (dense) {
        // every record = a 4-pixel row: two halves, at most one strict maximum each; all passes of a wave are appended with one global atomic
        const int nrec = min(s_nlist, NLIST);
        for (int i0 = 0; i0 < nrec; i0 += T) {
            if (i0 + (int)((unsigned)tid & ~63u) >= nrec) continue;          // this wave has no record in this round (no barrier in the loop)
            const int i = i0 + (unsigned)tid;
            uint32_t z = 0;
            int px0 = 0, py = 0, ini = 0;
            if (i < nrec) {
                const int code = s_pairs[i], c = code >> 10;
                z = s_recz[i];
                px0 = 4 * (code & 15) + 3 + (cj0 + c) * g.wCell;
                py = ((code >> 4) & 63) + 3 + ci * g.hCell;
                ini = s_ini[c];
            }
            uint32_t kp[2];
            unsigned long long bm[2];
            bool passk[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const uint32_t h = k ? z >> 16 : z & 0xffffu;
                const uint32_t second = h >> 8;                            // non-zero: the maximum is the half's second pixel
                const int sc = (int)(second ? second : h) + zoff;
                const int px = px0 + 2 * k + (second ? 1 : 0);
                bool pass = h != 0 && !(ini && sc < P.iniTh);
                if (pass && mimg) pass = mimg[(size_t)py * g.pitch + px] != 0;         // (no +16: reference quirk)
                kp[k] = ((uint32_t)py << 20) | ((uint32_t)px << 8) | (uint32_t)sc;
                bm[k] = __ballot(pass); passk[k] = pass;
            }
            const int total = (int)(__popcll(bm[0]) + __popcll(bm[1]));
            if (total == 0) continue;                                      // wave-uniform
            int gbase = 0;
            if (lane == 0) gbase = atomicAdd(&candCount[b * MAXL + level], total);
            gbase = __builtin_amdgcn_readfirstlane(gbase);
            int dst = rank_below(bm[0], rank_below(bm[1], gbase));                             // lane-major, as above
            if (passk[0]) {
                if (dst < g.keyCap) out[dst] = kp[0];
                dst++;
            }
            if (passk[1]) {
                if (dst < g.keyCap) out[dst] = kp[1];
            }
        }
    } else {
    const int nl = min(s_nlist, NLIST);
    for (int i0 = 0; i0 < nl; i0 += T) {
        const int i = i0 + (unsigned)tid;
        bool pass = false;
        uint32_t kp = 0;
        if (i < nl) {
            const uint32_t e = s_list[i];
            const int sc = (int)(e & 0xff) + zoff;
            kp = (e & 0xffffff00u) | (uint32_t)sc;
            pass = !(s_ini[s_listc[i]] && sc < P.iniTh);
            if (pass && mimg) pass = mimg[(size_t)(e >> 20) * g.pitch + ((e >> 8) & 0xfff)] != 0;      // (no +16: reference quirk)
        }
        const unsigned long long bm = __ballot(pass);
        if (bm == 0) continue;
        int gbase = 0;
        if (lane == 0) gbase = atomicAdd(&candCount[b * MAXL + level], __popcll(bm));
        gbase = __builtin_amdgcn_readfirstlane(gbase);
        if (pass) {
            const int dst = gbase + __popcll(bm & ((1ull << lane) - 1ull));
            if (dst < g.keyCap) out[dst] = kp;
        }
    }
    }
    rec = rec_next; par ^= (NS > 1 ? 1 : 0);
  }
}

// ------------------------------------------------------------------------------------------------
// K4: oct-tree keypoint distribution, one 256-thread block per (level, image), all levels in one launch.
//
// The reference walks a std::list of nodes, splitting nodes into 4 children and re-bucketing their
// key vectors (ORBextractor.cpp:586-810).  Here every key gets its quad-tree PATH CODE up front (root
// index + 2 bits per depth, derived with the reference's ceil-halving bounds).  Keys are bucketed by the
// first D levels of the code with ONE counting sort whose histogram lives in LDS (D chosen so that
// nIni*4^D <= 1024 buckets); the exclusive bucket offsets then give the key range of ANY node of depth
// <= D, and of its 4 children, by table lookup — the node-list simulation touches no global memory.
// A node deeper than D (only reached when many keys crowd into one ~10-px cell) is split on demand by
// partitioning its own key range in place on the next code digit.  Order inside a node is irrelevant:
// child counts do not depend on it and the best key per node is chosen by (max response, first in the
// reference's cell-major / row-major candidate order) explicitly (:795-804).
// The list order the reference produces (push_front of n1..n4, erase of the parent, size-sorted
// expansion near the budget, stop as soon as >= N nodes) is rebuilt with block-wide prefix sums.
// Tie-break of the size sort (:731 sorts pair<int,Node*>, i.e. by heap address) = creation order, the
// same deterministic choice the oracle makes.
// ------------------------------------------------------------------------------------------------
// threads per (image, level) block (template parameter OT of k_octree): the passes are separated by block barriers, so a block's time is
// its slowest wave's.  Batches run 256-thread blocks (five instead of four blocks per CU by LDS, +0.8 % frames/s under the pipeline
// although the kernel alone is slower, 0.87 against 0.78 ms per 1024 images); a lone frame has only nlevels blocks in flight and runs
// them 512 wide (53 us instead of 75)
constexpr int OT_MAXB = 1024;        // buckets of the counting sort

// inclusive prefix sum of a 64-bit value over the 64 lanes on the DPP network (both halves moved, one 64-bit add per step; lanes
// without a source add zero): 24 VALU operations instead of 12 dependent trips through the LDS crossbar
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v) {
#define MYSLAM_SCAN64_STEP(CTRL, ROWS)                                                                                   \
    {                                                                                                                    \
        const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWS, 0xf, false);          \
        const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROWS, 0xf, false);  \
        v += ((uint64_t)h2 << 32) | l2;                                                                                  \
    }
    MYSLAM_SCAN64_STEP(0x111, 0xf)          // row_shr:1
    MYSLAM_SCAN64_STEP(0x112, 0xf)          // row_shr:2
    MYSLAM_SCAN64_STEP(0x114, 0xf)          // row_shr:4
    MYSLAM_SCAN64_STEP(0x118, 0xf)          // row_shr:8
    MYSLAM_SCAN64_STEP(0x142, 0xa)          // row_bcast15 -> rows 1, 3
    MYSLAM_SCAN64_STEP(0x143, 0xc)          // row_bcast31 -> rows 2, 3
#undef MYSLAM_SCAN64_STEP
    return v;
}

// exclusive block scan of a packed 64-bit counter; s_w = OT/64 uint64 of LDS scratch
template <int OT>
__device__ __forceinline__ uint64_t block_excl_scan64(uint64_t v, uint64_t* s_w, uint64_t& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint64_t incl = wave_incl_scan64(v);
    __syncthreads();                       // previous users of s_w are done
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < OT / 64; w++) { const uint64_t x = s_w[w]; if (w < wid) base += x; tot += x; }
    total = tot;
    return base + incl - v;
}

__device__ __forceinline__ uint32_t oct_code(int px, int py, const LevelGeom& g) {
    int r = (int)__fdiv_rn((float)px, g.hX);                                  // :616
    r = min(max(r, 0), g.nIni - 1);
    int ULx = (int)__fmul_rn(g.hX, (float)r), URx = (int)__fmul_rn(g.hX, (float)(r + 1));   // :602-603
    int ULy = 0, BRy = g.maxBY - MIN_BORDER;
    uint32_t code = (uint32_t)r << ROOT_SHIFT;
    for (int k = 1; k <= g.ndepth; k++) {
        const int midX = ULx + ((URx - ULx + 1) >> 1);                        // ceil(w/2), :528-529
        const int midY = ULy + ((BRy - ULy + 1) >> 1);
        const uint32_t dx = px >= midX, dy = py >= midY;                      // :560-570
        code |= (dx | (dy << 1)) << (ROOT_SHIFT - 2 * k);
        if (dx) ULx = midX; else URx = midX;
        if (dy) ULy = midY; else BRy = midY;
    }
    return code;
}

// The sorted entries are 32-bit (the selection key, or the FAST payload on grids of more than 1024 cells); their quad-tree path codes
// live in a parallel array that only nodes deeper than the counting sort's depth ever read (many keys crowding into one ~10-px cell).
// in-place 4-way partition of (S, Cd)[lo,hi) on the 2-bit path-code digit at bit position sh (deep nodes only; one thread)
__device__ __noinline__ void partition4(uint32_t* __restrict__ S, uint32_t* __restrict__ Cd, int lo, int hi, int sh, int& b1, int& b2, int& b3) {
    int c0 = 0, c1 = 0, c2 = 0;
    for (int i = lo; i < hi; i++) {
        const int d = (int)((Cd[i] >> sh) & 3);
        c0 += (d == 0); c1 += (d == 1); c2 += (d == 2);
    }
    b1 = lo + c0; b2 = b1 + c1; b3 = b2 + c2;
    int p0 = lo, p1 = b1, p2 = b2, p3 = b3;                  // next free slot of every bucket
    auto take = [&](int d) -> int { int r; if (d == 0) r = p0++; else if (d == 1) r = p1++; else if (d == 2) r = p2++; else r = p3++; return r; };
    const int ends[4] = {b1, b2, b3, hi};
#pragma unroll
    for (int k = 0; k < 3; k++) {                            // bucket 3 is in place once 0..2 are
        int i = (k == 0) ? p0 : (k == 1) ? p1 : p2;
        while (i < ends[k]) {
            uint32_t v = S[i], cv = Cd[i];
            int d = (int)((cv >> sh) & 3);
            while (d != k) {                                 // cycle the element into its bucket
                const int j = take(d);
                const uint32_t w = S[j], cw = Cd[j];
                S[j] = v; Cd[j] = cv; v = w; cv = cw;
                d = (int)((cv >> sh) & 3);
            }
            S[i] = v; Cd[i] = cv;
            i++;
            if (k == 0) p0 = i; else if (k == 1) p1 = i; else p2 = i;
        }
    }
}

struct OctLds {
    uint32_t *lo0, *lo1, *hi0, *hi1;   // node key range [lo, hi) (double buffered)
    uint16_t *pfx0, *pfx1;      // node bucket prefix (valid while depth <= D)
    uint8_t *dep0, *dep1;       // node depth
    uint32_t *nb1, *nb2, *nb3;  // child boundaries (per list position / per sorted candidate)
    uint32_t *ckey0, *ckey1;    // candidate sort key: size<<14 | creation index (size < 2^18, <= 16383 candidates)
    uint16_t *cpos0, *cpos1;    // candidate -> list position
    uint16_t* ord;              // rank -> candidate
    uint16_t *kinc, *binc;      // inclusive sums of children / big children over sorted candidates
    uint8_t* kk;                // #non-empty children (0 = not expandable)
    uint8_t* mark;
    uint32_t* offs;             // exclusive bucket offsets [NB+1]
    __device__ __forceinline__ uint32_t* lo(int i) const { return i ? lo1 : lo0; }
    __device__ __forceinline__ uint32_t* hi(int i) const { return i ? hi1 : hi0; }
    __device__ __forceinline__ uint16_t* pfx(int i) const { return i ? pfx1 : pfx0; }
    __device__ __forceinline__ uint8_t* dep(int i) const { return i ? dep1 : dep0; }
    __device__ __forceinline__ uint32_t* ckey(int i) const { return i ? ckey1 : ckey0; }
    __device__ __forceinline__ uint16_t* cpos(int i) const { return i ? cpos1 : cpos0; }
};

// boundaries of the 4 children of node (depth d, prefix pfx, [lo,hi))
__device__ __forceinline__ void child_bounds(const OctLds& L, uint32_t* __restrict__ S, uint32_t* __restrict__ Cd, int D, int lo, int hi, int d, int pfx,
                                             int& b1, int& b2, int& b3) {
    if (d < D) {
        const int sb = 2 * (D - d - 1);
        const int base = pfx << 2;
        b1 = L.offs[(base + 1) << sb]; b2 = L.offs[(base + 2) << sb]; b3 = L.offs[(base + 3) << sb];
    } else {
        partition4(S, Cd, lo, hi, ROOT_SHIFT - 2 * (d + 1), b1, b2, b3);
    }
}

// maximum over aligned groups of 8 lanes (a half row of the DPP network: mirror, then the two quad exchanges), in every lane of the group
__device__ __forceinline__ uint32_t half_row_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));      // row_half_mirror: lane i <- lane 7 - i
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false));       // quad_perm [1, 0, 3, 2]
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false));       // quad_perm [2, 3, 0, 1]
    return v;
}

// tile bin of a selected key for the descriptor kernel's processing order (k_sel_order and phase E of k_octree): column high bits | Z-order of
// the low 4 + 4 tile bits
__device__ __forceinline__ int sel_order_bin(uint32_t pay, int ts) {
    const int tx = (int)((pay >> 8) & 0xfff) >> ts, ty = (int)(pay >> 20) >> ts;
    int z = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) z |= (((tx >> k) & 1) << (2 * k)) | (((ty >> k) & 1) << (2 * k + 1));
    return ((tx >> 4) << 8) | z;
}

constexpr int OCT_WIDE_BELOW = 64;      // batches smaller than this run 512-thread blocks
// (1024-thread blocks for a lone pair, end of round 5: 42.3 -> 41.1 us in the recorded one-pair chain — the time is the chain of phases, not
// their width; not kept.  Likewise FAST in strips of 2 cells instead of 4 for a lone pair, so that a block's chain of phases is shorter: its
// generic item mapping made the launch 91 us instead of 33; not kept.)
constexpr int OCT_FL = 4;              // candidates in flight per thread in the two candidate passes (8: same time, batched and one-frame)
template <int OT>
__global__ __launch_bounds__(OT) void k_octree(OrbPlan P, const uint32_t* __restrict__ cand,
                                               const int32_t* __restrict__ candCount, uint32_t* __restrict__ sortbuf,
                                               const uint32_t* __restrict__ octTab,
                                               uint32_t* __restrict__ selOut, int32_t* __restrict__ selCount,
                                               int32_t* __restrict__ status, int NCmax, uint16_t* __restrict__ order, BlurMulti BM) {
    MYSLAM_SIDE_PRIO();
    MYSLAM_BT(2);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int t = threadIdx.x;
    if constexpr (OT == 512) {                                  // small batches: rows of blocks behind the levels' are Gaussian bands (launch_octree)
        if ((int)blockIdx.y >= P.nlevels) {
            const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
            blur7_strip_wave(BM, ((int)blockIdx.y - P.nlevels) * 8 + wv, blockIdx.x, smem + wv * (16 * B3_TS));
            return;
        }
    }
    // images along x, levels along y: blocks are dispatched x-fastest, so all level-0 blocks (a third of the candidates sit there) start first
    // and the launch ends on the small top levels, not on a few long blocks
    const int level = blockIdx.y, b = blockIdx.x;
    const LevelGeom& g = P.lv[level];
    const int NC = NCmax;

    // carve LDS
    uint64_t* s_w = reinterpret_cast<uint64_t*>(smem);          // OT/64 x u64 scan scratch
    int* s_i = reinterpret_cast<int*>(smem + 128);              // [0]=m [1]=nc [2]=jstar
    uint8_t* pcur = smem + 192;
    uint32_t* s_cursor = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * OT_MAXB;
    OctLds L;
    L.offs = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * (OT_MAXB + 2);
    uint32_t* const s_tab = reinterpret_cast<uint32_t*>(pcur);      // phases A / B: the level's lookup tables, in the space of the node arrays (54 NC bytes)
    L.lo0 = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * NC;
    L.lo1 = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * NC;
    L.hi0 = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * NC;
    L.hi1 = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * NC;
    L.nb1 = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * NC;
    L.nb2 = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * NC;
    L.nb3 = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * NC;
    L.ckey0 = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * NC;
    L.ckey1 = reinterpret_cast<uint32_t*>(pcur); pcur += 4 * NC;
    L.pfx0 = reinterpret_cast<uint16_t*>(pcur); pcur += 2 * NC;
    L.pfx1 = reinterpret_cast<uint16_t*>(pcur); pcur += 2 * NC;
    L.cpos0 = reinterpret_cast<uint16_t*>(pcur); pcur += 2 * NC;
    L.cpos1 = reinterpret_cast<uint16_t*>(pcur); pcur += 2 * NC;
    L.ord = reinterpret_cast<uint16_t*>(pcur); pcur += 2 * NC;
    L.kinc = reinterpret_cast<uint16_t*>(pcur); pcur += 2 * NC;
    L.binc = reinterpret_cast<uint16_t*>(pcur); pcur += 2 * NC;
    L.dep0 = pcur; pcur += NC;
    L.dep1 = pcur; pcur += NC;
    L.kk = pcur; pcur += NC;
    L.mark = pcur; pcur += NC;

    const int cnt = candCount[b * MAXL + level];
    int32_t* myCount = selCount + b * MAXL + level;
    if (cnt > g.keyCap) {                      // candidate list overflowed: refuse rather than truncate
        if (t == 0) { *myCount = 0; status[b] = MYSLAM_ERR_CAPACITY; }
        return;
    }
    const int n = cnt;
    if (n == 0) { if (t == 0) *myCount = 0; return; }

    const uint32_t* keys = cand + (size_t)b * P.totalKeyCap + g.keyOff;
    uint32_t* S = sortbuf + ((size_t)b * P.totalKeyCap + g.keyOff) * 2;    // the level's entries in bucket order (32 bits each) ...
    uint32_t* Cd = S + g.keyCap;                                           // ... and their path codes
    const int D = g.sortDepth;                                   // bucket = code >> (ROOT_SHIFT - 2D)
    const int NB = g.nIni << (2 * D);
    const int bsh = ROOT_SHIFT - 2 * D;

    const uint32_t* xcode = octTab + g.tabOff; const uint32_t* ycode = xcode + g.tabX;
    const uint32_t* xcell = ycode + g.tabY; const uint32_t* ycell = xcell + g.tabX;
    // The passes over the candidates are bound by the texture addresser: per candidate four table gathers (one L1 access per lane
    // each) and two scattered stores.  The level's four tables (2 (w + h) dwords, 13 KB at 1241 x 376) are therefore staged in LDS —
    // in the space of the node arrays, which phase C initialises — whenever they fit; the gathers become LDS reads.
    const int ntab = 2 * (g.tabX + g.tabY);
    const bool tabLds = (size_t)ntab * 4 <= (size_t)54 * NC;            // block-uniform
    // ---- A: path codes + bucket histogram (LDS atomics) ----
    for (int i = t; i < NB; i += OT) s_cursor[i] = 0;
    if (tabLds) for (int i = t; i < ntab; i += OT) s_tab[i] = xcode[i];
    __syncthreads();
    // A sort entry is what phase D maximises per node.  With at most 1024 grid cells it is the complete selection key of the
    // reference (:795-804: largest response, then first in the cell-major / row-major candidate order):
    //     response << 22 | (0x3FFFFF - (cell << 12 | row in cell << 6 | column in cell))        (cells are at most 59 x 59)
    // so phase D is ONE 32-bit maximum per node and the winner's pixel is decoded from the key; otherwise the entry is the FAST
    // payload and phase D resolves the order with the cell tables in a second pass.
    // The candidate list is walked twice (histogram, then scatter: the second walk hits in L2) instead of parking 64-bit
    // (code, key) pairs in HBM between the passes, and phase D streams the 32-bit keys alone: 12-16 bytes of sort traffic per
    // candidate instead of 32.
    const bool rankkey = g.nCols * g.nRows <= 1024;
    const float inv_ncols = 1.0f / (float)g.nCols;
    auto passes = [&](auto in_lds) __attribute__((always_inline)) {
        constexpr bool LDS = decltype(in_lds)::value;
        const int oyc = g.tabX, oxl = g.tabX + g.tabY, oyl = 2 * g.tabX + g.tabY;
        auto XC = [&](uint32_t i) -> uint32_t { if constexpr (LDS) return s_tab[i]; else return xcode[i]; };
        auto YC = [&](uint32_t i) -> uint32_t { if constexpr (LDS) return s_tab[oyc + i]; else return ycode[i]; };
        auto XL = [&](uint32_t i) -> uint32_t { if constexpr (LDS) return s_tab[oxl + i]; else return xcell[i]; };
        auto YL = [&](uint32_t i) -> uint32_t { if constexpr (LDS) return s_tab[oyl + i]; else return ycell[i]; };
        for (int i0 = t; i0 < n; i0 += OCT_FL * OT) {                  // OCT_FL keys in flight per thread: the pass is latency-bound
            uint32_t pay[OCT_FL], cx[OCT_FL], cy[OCT_FL];
#pragma unroll
            for (int u = 0; u < OCT_FL; u++) pay[u] = (i0 + u * OT < n) ? keys[i0 + u * OT] : 0u;
#pragma unroll
            for (int u = 0; u < OCT_FL; u++) { cx[u] = XC((pay[u] >> 8) & 0xfff); cy[u] = YC(pay[u] >> 20); }
#pragma unroll
            for (int u = 0; u < OCT_FL; u++)
                if (i0 + u * OT < n) atomicAdd(&s_cursor[(cx[u] | cy[u]) >> bsh], 1u);     // code == oct_code(px, py, g), tabulated per axis at plan time
        }
        __syncthreads();
        // ---- B: exclusive offsets, then scatter (order inside a bucket is irrelevant) ----
        {
            const int c = (NB + OT - 1) / OT;
            const int beg = min(NB, t * c), end = min(NB, beg + c);
            uint64_t sum = 0;
            for (int i = beg; i < end; i++) sum += s_cursor[i];
            uint64_t total;
            uint64_t run = block_excl_scan64<OT>(sum, s_w, total);
            for (int i = beg; i < end; i++) {
                const uint32_t h = s_cursor[i];
                L.offs[i] = (uint32_t)run;
                s_cursor[i] = (uint32_t)run;
                run += h;
            }
            if (t == 0) L.offs[NB] = (uint32_t)n;
        }
        __syncthreads();
        for (int i0 = t; i0 < n; i0 += OCT_FL * OT) {
            uint32_t pay[OCT_FL], cx[OCT_FL], cy[OCT_FL], qx[OCT_FL], qy[OCT_FL];
#pragma unroll
            for (int u = 0; u < OCT_FL; u++) pay[u] = (i0 + u * OT < n) ? keys[i0 + u * OT] : 0u;
#pragma unroll
            for (int u = 0; u < OCT_FL; u++) {
                const uint32_t px = (pay[u] >> 8) & 0xfff, py = pay[u] >> 20;
                cx[u] = XC(px); cy[u] = YC(py);
                qx[u] = rankkey ? XL(px) : 0u; qy[u] = rankkey ? YL(py) : 0u;
            }
#pragma unroll
            for (int u = 0; u < OCT_FL; u++) {
                if (i0 + u * OT >= n) continue;
                uint32_t low = pay[u];
                if (rankkey) {
                    const int px = (int)((pay[u] >> 8) & 0xfff), py = (int)(pay[u] >> 20);
                    const int ci = (int)(((float)qy[u] + 0.5f) * inv_ncols);                  // ycell = ci * nCols (exact: < 2^11)
                    const uint32_t pxc = (uint32_t)(px - 3 - (int)qx[u] * g.wCell), pyc = (uint32_t)(py - 3 - ci * g.hCell);
                    low = ((pay[u] & 0xffu) << 22) | (0x3fffffu - (((qy[u] + qx[u]) << 12) | (pyc << 6) | pxc));
                }
                const uint32_t code = cx[u] | cy[u];
                const uint32_t pos = atomicAdd(&s_cursor[code >> bsh], 1u);
                S[pos] = low; Cd[pos] = code;
            }
        }
        __syncthreads();
    };
    if (tabLds) passes(std::true_type{}); else passes(std::false_type{});

    MYSLAM_BT_MARK(0);                                                 // (trace builds: the two candidate passes are done)
    MYSLAM_BT_AUX((unsigned long long)level | ((unsigned long long)n << 8));
    // ---- C: node list simulation ----
    // roots (:599-632): non-empty roots in index order
    if (t == 0) {
        int m = 0;
        for (int r = 0; r < g.nIni; r++) {
            const int lo = L.offs[r << (2 * D)], hi = L.offs[(r + 1) << (2 * D)];
            if (hi > lo) { L.lo0[m] = (uint32_t)lo; L.hi0[m] = (uint32_t)hi; L.dep0[m] = 0; L.pfx0[m] = (uint16_t)r; m++; }
        }
        s_i[0] = m; s_i[1] = 0;
    }
    __syncthreads();

    int cur = 0, ccur = 0, mode = 0;
    const int N = g.N;
    for (int round = 0; round < 96; round++) {
        const int m = s_i[0];
        const int prevSize = m;
        int newM;
        if (mode == 0) {
            // ---------- full round (:645-712): every node with >1 key is split ----------
            const int c = (m + OT - 1) / OT;
            const int beg = min(m, t * c), end = min(m, beg + c);
            uint64_t packed = 0;              // K | Non<<21 | Big<<42
            for (int p = beg; p < end; p++) {
                const int lo = (int)L.lo(cur)[p], hi = (int)L.hi(cur)[p], d = L.dep(cur)[p];
                int k = 0;
                if (hi - lo > 1 && d < g.ndepth) {
                    int b1, b2, b3;
                    child_bounds(L, S, Cd, D, lo, hi, d, L.pfx(cur)[p], b1, b2, b3);
                    L.nb1[p] = (uint32_t)b1; L.nb2[p] = (uint32_t)b2; L.nb3[p] = (uint32_t)b3;
                    const int c0 = b1 - lo, c1 = b2 - b1, c2 = b3 - b2, c3 = hi - b3;
                    k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                    const int big = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                    packed += (uint64_t)k | ((uint64_t)big << 42);
                } else {
                    packed += 1ull << 21;
                }
                L.kk[p] = (uint8_t)k;
            }
            uint64_t total;
            uint64_t run = block_excl_scan64<OT>(packed, s_w, total);
            const int Ktot = (int)(total & 0x1fffff), Ntot = (int)((total >> 21) & 0x1fffff), Btot = (int)(total >> 42);
            const int nxt = cur ^ 1, cnxt = ccur ^ 1;
            for (int p = beg; p < end; p++) {
                const int lo = (int)L.lo(cur)[p], hi = (int)L.hi(cur)[p], d = L.dep(cur)[p];
                const int pf = L.pfx(cur)[p];
                const int k = L.kk[p];
                if (k) {
                    const int bnd[5] = {lo, (int)L.nb1[p], (int)L.nb2[p], (int)L.nb3[p], hi};
                    const int rk = (int)(run & 0x1fffff);
                    int pos = Ktot - (rk + k);                           // children of later nodes come first
                    int cidx = (int)(run >> 42);
                    int childPos[4];
#pragma unroll
                    for (int q = 3; q >= 0; q--) {                       // list order n4,n3,n2,n1 (push_front)
                        childPos[q] = pos;
                        if (bnd[q + 1] > bnd[q]) {
                            L.lo(nxt)[pos] = (uint32_t)bnd[q]; L.hi(nxt)[pos] = (uint32_t)bnd[q + 1];
                            L.dep(nxt)[pos] = (uint8_t)(d + 1);
                            L.pfx(nxt)[pos] = (uint16_t)((pf << 2) | q);
                            pos++;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {                        // creation order n1..n4
                        const int sz = bnd[q + 1] - bnd[q];
                        if (sz > 1) {
                            L.ckey(cnxt)[cidx] = ((uint32_t)sz << 14) | (uint32_t)cidx;
                            L.cpos(cnxt)[cidx] = (uint16_t)childPos[q];
                            cidx++;
                        }
                    }
                    const int big = cidx - (int)(run >> 42);
                    run += (uint64_t)k | ((uint64_t)big << 42);
                } else {
                    const int rn = (int)((run >> 21) & 0x1fffff);
                    L.lo(nxt)[Ktot + rn] = (uint32_t)lo; L.hi(nxt)[Ktot + rn] = (uint32_t)hi;
                    L.dep(nxt)[Ktot + rn] = (uint8_t)d;
                    L.pfx(nxt)[Ktot + rn] = (uint16_t)pf;
                    run += 1ull << 21;
                }
            }
            newM = Ktot + Ntot;
            __syncthreads();
            if (t == 0) { s_i[0] = newM; s_i[1] = Btot; }
            cur = nxt; ccur = cnxt;
            __syncthreads();
            if (newM >= N || newM == prevSize) break;                    // :716
            if (newM + 3 * Btot > N) mode = 1;                           // :720
        } else {
            // ---------- budget-limited pass (:723-784): largest nodes first, stop at N ----------
            const int nc = s_i[1];
            if (nc == 0) break;                                          // nothing to split -> size unchanged
            // rank sort descending by (size, creation index)
            for (int ci = t; ci < nc; ci += OT) {
                const uint32_t key = L.ckey(ccur)[ci];
                int rank = 0;
                for (int cj = 0; cj < nc; cj++) rank += (L.ckey(ccur)[cj] > key);
                L.ord[rank] = (uint16_t)ci;
            }
            for (int p = t; p < m; p += OT) L.mark[p] = 0;
            if (t == 0) s_i[2] = nc - 1;
            __syncthreads();
            const int c = (nc + OT - 1) / OT;
            const int beg = min(nc, t * c), end = min(nc, beg + c);
            uint64_t packed = 0;                                         // K | Big<<32
            for (int j = beg; j < end; j++) {
                const int p = L.cpos(ccur)[L.ord[j]];
                const int lo = (int)L.lo(cur)[p], hi = (int)L.hi(cur)[p], d = L.dep(cur)[p];
                int k = 1, big = 1, b1 = hi, b2 = hi, b3 = hi;           // depth-exhausted node: one "child" = itself
                if (d < g.ndepth) {
                    child_bounds(L, S, Cd, D, lo, hi, d, L.pfx(cur)[p], b1, b2, b3);
                    const int c0 = b1 - lo, c1 = b2 - b1, c2 = b3 - b2, c3 = hi - b3;
                    k = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                    big = (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1);
                }
                L.nb1[j] = (uint32_t)b1; L.nb2[j] = (uint32_t)b2; L.nb3[j] = (uint32_t)b3;
                L.kk[j] = (uint8_t)(k | (big << 4));
                packed += (uint64_t)k | ((uint64_t)big << 32);
            }
            uint64_t total;
            uint64_t run = block_excl_scan64<OT>(packed, s_w, total);
            for (int j = beg; j < end; j++) {
                const int k = L.kk[j] & 15, big = L.kk[j] >> 4;
                run += (uint64_t)k | ((uint64_t)big << 32);
                const int kincl = (int)(run & 0xffffffffu);
                L.kinc[j] = (uint16_t)kincl;
                L.binc[j] = (uint16_t)(run >> 32);
                if (m + kincl - (j + 1) >= N) atomicMin(&s_i[2], j);     // :777 break as soon as size >= N
            }
            __syncthreads();
            const int ncmt = min(s_i[2] + 1, nc);
            const int Kc = L.kinc[ncmt - 1];
            const int nxt = cur ^ 1, cnxt = ccur ^ 1;
            for (int j = t; j < ncmt; j += OT) L.mark[L.cpos(ccur)[L.ord[j]]] = 1;
            __syncthreads();
            // committed candidates: children in front, reverse processing order
            for (int j = t; j < ncmt; j += OT) {
                const int p = L.cpos(ccur)[L.ord[j]];
                const int lo = (int)L.lo(cur)[p], hi = (int)L.hi(cur)[p], d = L.dep(cur)[p];
                const int pf = L.pfx(cur)[p];
                const int big = L.kk[j] >> 4;
                const int bnd[5] = {lo, (int)L.nb1[j], (int)L.nb2[j], (int)L.nb3[j], hi};
                int pos = Kc - L.kinc[j];
                int childPos[4];
                if (d < g.ndepth) {
#pragma unroll
                    for (int q = 3; q >= 0; q--) {
                        childPos[q] = pos;
                        if (bnd[q + 1] > bnd[q]) {
                            L.lo(nxt)[pos] = (uint32_t)bnd[q]; L.hi(nxt)[pos] = (uint32_t)bnd[q + 1];
                            L.dep(nxt)[pos] = (uint8_t)(d + 1);
                            L.pfx(nxt)[pos] = (uint16_t)((pf << 2) | q);
                            pos++;
                        }
                    }
                    int cidx = L.binc[j] - big;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int sz = bnd[q + 1] - bnd[q];
                        if (sz > 1) {
                            L.ckey(cnxt)[cidx] = ((uint32_t)sz << 14) | (uint32_t)cidx;
                            L.cpos(cnxt)[cidx] = (uint16_t)childPos[q];
                            cidx++;
                        }
                    }
                } else {                                                // cannot happen for distinct keys; keep node
                    L.lo(nxt)[pos] = (uint32_t)lo; L.hi(nxt)[pos] = (uint32_t)hi; L.dep(nxt)[pos] = (uint8_t)d; L.pfx(nxt)[pos] = (uint16_t)pf;
                    const int cidx = L.binc[j] - 1;
                    L.ckey(cnxt)[cidx] = ((uint32_t)(hi - lo) << 14) | (uint32_t)cidx;
                    L.cpos(cnxt)[cidx] = (uint16_t)pos;
                }
            }
            // untouched nodes keep their relative order behind the new children
            {
                const int c2 = (m + OT - 1) / OT;
                const int pb = min(m, t * c2), pe = min(m, pb + c2);
                uint64_t un = 0;
                for (int p = pb; p < pe; p++) un += (L.mark[p] == 0);
                uint64_t tot2;
                uint64_t ex = block_excl_scan64<OT>(un, s_w, tot2);
                for (int p = pb; p < pe; p++) {
                    if (L.mark[p] == 0) {
                        L.lo(nxt)[Kc + (int)ex] = L.lo(cur)[p]; L.hi(nxt)[Kc + (int)ex] = L.hi(cur)[p];
                        L.dep(nxt)[Kc + (int)ex] = L.dep(cur)[p];
                        L.pfx(nxt)[Kc + (int)ex] = L.pfx(cur)[p];
                        ex++;
                    }
                }
            }
            newM = Kc + (m - ncmt);
            const int newNc = L.binc[ncmt - 1];
            __syncthreads();
            if (t == 0) { s_i[0] = newM; s_i[1] = newNc; }
            cur = nxt; ccur = cnxt;
            __syncthreads();
            if (newM >= N || newM == prevSize) break;                    // :781
        }
    }

    MYSLAM_BT_MARK(1);
    // ---- D: best key per node (:788-807), list order.  8 or 16 lanes per node.  Payload levels: pass 1 finds the node's largest response;
    // pass 2 breaks ties by the reference's candidate order (cell-major, row-major inside a cell), so the cell tables
    // are only read for keys that carry that response. ----
    const int m = s_i[0];
    uint32_t* out = selOut + (size_t)b * P.totalOut + g.outBase;
    {
        // rank-key levels: 8 lanes per node; otherwise 16
        const int GLN = rankkey ? 8 : 16;
        const int sub = t & (GLN - 1);
        const int mm = min(m, g.nodeCap);
        const int ngrp = OT / GLN;
        for (int p0 = t / GLN; p0 < ((mm + ngrp - 1) / ngrp) * ngrp; p0 += ngrp) {
            const bool live = p0 < mm;
            const int lo = live ? (int)L.lo(cur)[p0] : 0, hi = live ? (int)L.hi(cur)[p0] : 0;
            if (rankkey) {                                               // block-uniform
                // four consecutive keys per lane and load (16-byte loads at 4-byte alignment; a load may run up to 3 entries past the node,
                // into the next node's keys or the code array behind S: masked), four loads in flight: 128 keys per round trip and group
                uint32_t bk = 0;
#pragma unroll 4
                for (int i = lo + 4 * sub; i < hi; i += 32) {
                    uint4 k4;
                    __builtin_memcpy(&k4, S + i, 16);
                    const int left = hi - i;                             // >= 1
                    bk = max(bk, k4.x);
                    bk = max(bk, left > 1 ? k4.y : 0u);
                    bk = max(bk, left > 2 ? k4.z : 0u);
                    bk = max(bk, left > 3 ? k4.w : 0u);
                }
                bk = half_row_max_u32(bk);
                if (live && sub == 0) {
                    const uint32_t rank = 0x3fffffu - (bk & 0x3fffffu), cell = rank >> 12;
                    const int ci = (int)(((float)cell + 0.5f) * inv_ncols), cj = (int)cell - ci * g.nCols;
                    const uint32_t px = (rank & 63u) + 3u + (uint32_t)(cj * g.wCell), py = ((rank >> 6) & 63u) + 3u + (uint32_t)(ci * g.hCell);
                    out[p0] = (py << 20) | (px << 8) | (bk >> 22);
                }
                continue;
            }
            uint32_t best = 0;
#pragma unroll 2
            for (int i = lo + sub; i < hi; i += 16) best = max(best, S[i] & 0xffu);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o, 64));
            uint64_t bo = ~0ull;
            for (int i = lo + sub; i < hi; i += 16) {
                const uint32_t pay = S[i];
                if ((pay & 0xffu) != best) continue;
                const uint32_t px = (pay >> 8) & 0xfff, py = pay >> 20;
                bo = min(bo, ((uint64_t)(ycell[py] + xcell[px]) << 24) | ((uint64_t)py << 12) | (uint64_t)px);
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) bo = min(bo, (uint64_t)__shfl_xor((unsigned long long)bo, o, 64));
            if (live && sub == 0) out[p0] = ((uint32_t)((bo >> 12) & 0xfff) << 20) | ((uint32_t)(bo & 0xfff) << 8) | best;
        }
    }
    if (t == 0) *myCount = min(m, g.nodeCap);
    MYSLAM_BT_MARK(2);
    // ---- E (batches): the descriptor kernel's processing order of this level's keys — Z-order of small pixel tiles, a counting sort over
    // <= 1024 tile bins (what the separate k_sel_order launch computes for other callers; fused here: one launch boundary less on the
    // extractor's chain).  The counting-sort buckets of phase B are free by now.
    if (order) {
        __threadfence_block();
        __syncthreads();
        int* s_hist = reinterpret_cast<int*>(s_cursor);                    // OT_MAXB = 1024 bins
        int* s_ws = reinterpret_cast<int*>(s_w);                           // 4 wave totals
        const int n = min(m, g.nodeCap);
        int ts = 4;
        while ((g.w >> ts) >= 64 || (g.h >> ts) >= 16) ts++;
        uint16_t* ord = order + (size_t)b * P.totalOut + g.outBase;
        for (int i = t; i < 1024; i += OT) s_hist[i] = 0;
        __syncthreads();
        // (the keys were stored by other lanes of this block a moment ago: read them back at agent scope, not through this CU's L1)
        for (int i = t; i < n; i += OT) atomicAdd(&s_hist[sel_order_bin(__hip_atomic_load(&out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ts)], 1);
        __syncthreads();
        int base = 0, v0 = 0, v1 = 0, v2 = 0, v = 0, incl = 0;
        if (t < 256) {                                                     // exclusive scan: 4 consecutive bins per thread of the first four waves
            v0 = s_hist[4 * t]; v1 = s_hist[4 * t + 1]; v2 = s_hist[4 * t + 2];
            v = v0 + v1 + v2 + s_hist[4 * t + 3];
            incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int nb = __shfl_up(incl, o, 64); if ((t & 63) >= o) incl += nb; }
            if ((t & 63) == 63) s_ws[t >> 6] = incl;
        }
        __syncthreads();
        if (t < 256) {
            base = incl - v;
            for (int w = 0; w < (t >> 6); w++) base += s_ws[w];
            s_hist[4 * t] = base; s_hist[4 * t + 1] = base + v0; s_hist[4 * t + 2] = base + v0 + v1; s_hist[4 * t + 3] = base + v0 + v1 + v2;
        }
        __syncthreads();
        for (int i = t; i < n; i += OT) ord[atomicAdd(&s_hist[sel_order_bin(__hip_atomic_load(&out[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), ts)], 1)] = (uint16_t)i;
    }
}

// ------------------------------------------------------------------------------------------------
// K5: orientation (intensity centroid) + steered BRIEF, one wave per keypoint.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {      // cv::fastAtan2 scalar form
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// Deterministic sin/cos shared (by construction, not by code) with the oracle: double Cody-Waite
// reduction + Taylor polynomials, plain IEEE mul/add in a fixed order, one final rounding to float.
__device__ __forceinline__ void det_sincos(float rad, float& s_out, float& c_out) {
    const double x = (double)rad;
    const double kd = rint(__dmul_rn(x, 0.6366197723675814));
    const int k = (int)kd;
    const double y = __dsub_rn(__dsub_rn(x, __dmul_rn(kd, 1.5707963267948966)), __dmul_rn(kd, 6.123233995736766e-17));
    const double y2 = __dmul_rn(y, y);
    double ps = -7.647163731819816e-13;
    ps = __dadd_rn(__dmul_rn(ps, y2), 1.6059043836821613e-10);
    ps = __dadd_rn(__dmul_rn(ps, y2), -2.505210838544172e-08);
    ps = __dadd_rn(__dmul_rn(ps, y2), 2.7557319223985893e-06);
    ps = __dadd_rn(__dmul_rn(ps, y2), -0.0001984126984126984);
    ps = __dadd_rn(__dmul_rn(ps, y2), 0.008333333333333333);
    ps = __dadd_rn(__dmul_rn(ps, y2), -0.16666666666666666);
    const double sn = __dadd_rn(y, __dmul_rn(y, __dmul_rn(y2, ps)));
    double pc = 4.779477332387385e-14;
    pc = __dadd_rn(__dmul_rn(pc, y2), -1.1470745597729725e-11);
    pc = __dadd_rn(__dmul_rn(pc, y2), 2.08767569878681e-09);
    pc = __dadd_rn(__dmul_rn(pc, y2), -2.755731922398589e-07);
    pc = __dadd_rn(__dmul_rn(pc, y2), 2.48015873015873e-05);
    pc = __dadd_rn(__dmul_rn(pc, y2), -0.001388888888888889);
    pc = __dadd_rn(__dmul_rn(pc, y2), 0.041666666666666664);
    pc = __dadd_rn(__dmul_rn(pc, y2), -0.5);
    const double cs = __dadd_rn(1.0, __dmul_rn(y2, pc));
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    s_out = (float)s;
    c_out = (float)c;
}

// IC_Angle over the 749-px disc, all 64 lanes; returns the angle on every lane
__device__ __forceinline__ float wave_ic_angle(const uint8_t* __restrict__ img, int pitch, int x, int y) {
    const int lane = threadIdx.x & 63;
    int m10 = 0, m01 = 0;
    // 31 rows; lanes 2r and 2r+1 share row r (v = r-15): left half incl. centre / right half
    const int r = lane >> 1;
    if (r < 31) {
        const int v = r - 15;
        const int d = c_umax[v < 0 ? -v : v];
        const uint8_t* row = img + (size_t)(y + v) * pitch + x;
        const int u0 = (lane & 1) ? 1 : -d, u1 = (lane & 1) ? d : 0;
        int sI = 0;
        for (int u = u0; u <= u1; u++) { const int I = row[u]; m10 += u * I; sI += I; }
        m01 = v * sI;
    }
    m10 = wave_reduce_sum(m10);
    m01 = wave_reduce_sum(m01);
    return fast_atan2_deg((float)m01, (float)m10);
}

// 256-bit rBRIEF; lane l produces bits 4l..4l+3; result bytes assembled with shuffles; lanes 0,8,..,56 hold a u32
// TILED: img is a blurred plane of the extractor (16 x 8-pixel tiles, tiled_off)
template <bool TILED = false>
__device__ __forceinline__ uint32_t wave_brief(const uint8_t* __restrict__ img, int pitch, int x, int y, float angle_deg) {
    const int lane = threadIdx.x & 63;
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ang = __fmul_rn(angle_deg, factorPI);
    float a, b;
    det_sincos(ang, b, a);
    const uint8_t* center = img + (size_t)y * pitch + x;
    uint32_t nib = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int8_t* pp = &c_pattern[(lane * 4 + j) * 4];
        const float x0 = (float)pp[0], y0 = (float)pp[1], x1 = (float)pp[2], y1 = (float)pp[3];
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = TILED ? img[tiled_off(x + c0, y + r0, pitch)] : center[r0 * pitch + c0];
        const int t1 = TILED ? img[tiled_off(x + c1, y + r1, pitch)] : center[r1 * pitch + c1];
        nib |= (uint32_t)(t0 < t1) << j;
    }
    uint32_t byte = nib | (__shfl_down(nib, 1, 64) << 4);            // valid on even lanes
    uint32_t w = byte | (__shfl_down(byte, 2, 64) << 8);
    w |= (__shfl_down(byte, 4, 64) << 16) | (__shfl_down(byte, 6, 64) << 24);   // valid on lanes % 8 == 0
    return w;
}

// LDS window of one keypoint (one wave): the 37x37 window of the blurred level that the steered 31x31 BRIEF pattern can reach
// (|rotated coord| <= 18), 64 bytes per row from a 16-byte aligned column (15 + 37 <= 64).  The kernel waits for L1 line fills and the
// texture addresser (TA_BUSY 72-78 % of its run time; tools/ta_probe.hip: a vector load costs max(16, ~1.25 x lines) cycles of a CU when
// the lines are in L1 and ~2.6-3 per line when they come from L2).  Round 2 fetched four aligned 16-byte pieces per row, lane = (row, piece):
// 3 instead of 7 instructions, ~55 lines per key-point; the blurred planes are now TILED (16 x 8 pixels per cache line, orb_plan.h), a
// window is 20-24 lines fetched as 512-byte runs (1.31 -> 1.16 ms per 1024 images).
constexpr int DB_R = 18, DB_ROWS = 2 * DB_R + 1, DB_P = 64;                       // row pitch of the LDS window (bytes)
constexpr int DB_N = DB_ROWS * 4;                                               // 148 pieces of 16 bytes
constexpr int DB_IT = 3;                                                        // 6 tile rows x 4 tile columns x 8 rows = 192 chunks of 16 bytes: 3 loads per lane

// K5: orientation + steered BRIEF, phased per block of 64 keypoints so that nothing scalar runs 64-wide:
//   0  thread per keypoint: slot -> (level, x, y, response)
//   A  16 keypoints / wave: intensity-centroid moments on the int8 matrix cores: [key-points x 64 k] x [64 k x {u, v weights}], the
//                           patch as it lies in memory is the A operand (p - 128; the masked weights sum to zero), 16 MFMAs
//   B  thread per keypoint: fastAtan2, deterministic sin/cos, cv::KeyPoint fields
//   C  wave per keypoint  : blurred 37x37 window -> LDS, 4 steered BRIEF tests per lane

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_i32(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_sum_lane63_i32(int v) {
    v = dpp_add_i32<0xB1, 0xf>(v); v = dpp_add_i32<0x4E, 0xf>(v); v = dpp_add_i32<0x141, 0xf>(v); v = dpp_add_i32<0x140, 0xf>(v);
    v = dpp_add_i32<0x142, 0xa>(v); v = dpp_add_i32<0x143, 0xc>(v);
    return v;
}

constexpr int KD_KPB = 64;                                       // keypoints per block
#ifndef MYSLAM_KD_GLDS                                              // windows a wave of the descriptor kernel keeps in flight in phase C (direct-to-LDS loads); 0 = the register-staged form
#define MYSLAM_KD_GLDS 2
#endif

// Processing order of the descriptor kernel: the selected keys of a level in Z-order of small pixel tiles (a counting sort over
// <= 1024 tile bins per (image, level); 32 x 32 pixels at 1241 x 376).  The oct-tree's list order scatters consecutive key-points all over the level; in tile order
// the 64 key-points of a block sit in a compact region, their 37 x 37 / 32 x 32 windows overlap and are fetched once per block
// instead of once per key-point.  Only the ORDER OF WORK changes: results go to the slots the list order prescribes.
__global__ __launch_bounds__(256) void k_sel_order(OrbPlan P, const uint32_t* __restrict__ selOut, const int32_t* __restrict__ selCount,
                                                   uint16_t* __restrict__ order, int batch) {
    constexpr int NB = 1024;                                           // tile bins: columns < 64, rows < 16
    __shared__ int s_hist[NB], s_wsum[4];
    const int level = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const LevelGeom& g = P.lv[level];
    const int n = min(selCount[b * MAXL + level], g.nodeCap);
    int ts = 4;                                                        // tile shift (bin = column high bits | Z-order of the low 4 + 4 bits)
    while ((g.w >> ts) >= 64 || (g.h >> ts) >= 16) ts++;
    const uint32_t* sel = selOut + (size_t)b * P.totalOut + g.outBase;
    uint16_t* out = order + (size_t)b * P.totalOut + g.outBase;
    for (int i = t; i < NB; i += 256) s_hist[i] = 0;
    __syncthreads();
    auto bin_of = [&](uint32_t pay) -> int { return sel_order_bin(pay, ts); };
    for (int i = t; i < n; i += 256) atomicAdd(&s_hist[bin_of(sel[i])], 1);
    __syncthreads();
    {   // exclusive scan of the bins: 4 consecutive bins per thread, wave scans, 4 wave totals
        const int v0 = s_hist[4 * t], v1 = s_hist[4 * t + 1], v2 = s_hist[4 * t + 2], v3 = s_hist[4 * t + 3];
        const int v = v0 + v1 + v2 + v3;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int nb = __shfl_up(incl, o, 64); if ((t & 63) >= o) incl += nb; }
        if ((t & 63) == 63) s_wsum[t >> 6] = incl;
        __syncthreads();
        int base = incl - v;
        for (int w = 0; w < (t >> 6); w++) base += s_wsum[w];
        s_hist[4 * t] = base; s_hist[4 * t + 1] = base + v0; s_hist[4 * t + 2] = base + v0 + v1; s_hist[4 * t + 3] = base + v0 + v1 + v2;
    }
    __syncthreads();
    for (int i = t; i < n; i += 256) out[atomicAdd(&s_hist[bin_of(sel[i])], 1)] = (uint16_t)i;
}

__device__ __forceinline__ void describe_block(const OrbPlan& P, const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                               size_t pyrStride, const uint32_t* __restrict__ selOut,
                                               const int32_t* __restrict__ selCount, myslam_keypoint* __restrict__ kps,
                                               uint8_t* __restrict__ desc, int32_t* __restrict__ counts,
                                               int32_t* __restrict__ status, int cap, int nchunk, int batch, int detectOnly,
                                               const uint16_t* __restrict__ order, const int logical_in MYSLAM_BT_PARAM) {
    const int logical = __builtin_amdgcn_readfirstlane(logical_in);      // block-uniform: everything derived from it stays on the scalar unit
    // per wave: phase A parks the 32 x 32 patches of four key-points here (4 x 64 pieces of 16 bytes), phase C the 37-row BRIEF window
#if MYSLAM_KD_GLDS
    // phase C keeps KD_NBUF windows per wave in flight (direct-to-LDS loads, below): KD_NBUF x 148 pieces per wave, and no less than phase A's 256
    constexpr int KD_WB = DB_N * MYSLAM_KD_GLDS > 256 ? DB_N * MYSLAM_KD_GLDS : 256;
    __shared__ __attribute__((aligned(16))) uint4 s_b[4][KD_WB];
#else
    __shared__ __attribute__((aligned(16))) uint4 s_b[4][256];
#endif
    static_assert(DB_N <= 256, "the BRIEF window must fit the per-wave buffer");
    __shared__ int s_x[KD_KPB], s_y[KD_KPB], s_lv[KD_KPB], s_m10[KD_KPB], s_m01[KD_KPB], s_out[KD_KPB];
    __shared__ float s_ca[KD_KPB], s_sb[KD_KPB];
    __shared__ __attribute__((aligned(16))) uint4 s_bw[16 * 4 * 2];      // IC-angle weights as MFMA B operands: [row pair][k block][x | y]
    __shared__ uint32_t s_pbase[KD_KPB], s_ppitch[KD_KPB];              // byte offset of patch(-15, -15) in the image's plane set (level 0 read in place: in the caller's image), row pitch
    // XCD-aware block order: the dispatcher places block i on XCD i % 8, each XCD has a private L2, and the ~32 blocks of one
    // image read overlapping windows of the same two pyramids.  Logical block ids (image-major) are handed out so that every
    // XCD walks a contiguous range of images (bijective remap, any grid size): without it every XCD pulls every image
    // through its own L2 (measured 3.2 GB instead of ~1.3 GB of HBM reads per 512 images).
    // the thread id passes through an opaque asm once per work item: inside the kernel's loop the compiler would otherwise hoist every
    // lane-dependent invariant (BRIEF pattern coordinates, operand roles: ~75 registers) out of the loop — 124 registers instead of 47
    int t_ = threadIdx.x;
    asm volatile("" : "+v"(t_));
    const int b = logical / nchunk, t = t_;
    if (b >= batch) return;
    if (t < 128) {      // weights of patch row 2 rp + (kq >> 1), columns 16 (kq & 1) .. +15: u (x moment) or v (y moment) inside the circular mask
        const int rp = t >> 3, kq = (t >> 1) & 3, jm = t & 1, row = 2 * rp + (kq >> 1), v = row - HALF_PATCH;
        const int d = (row <= 30) ? c_umax[v < 0 ? -v : v] : -1;
        uint32_t w[4] = {0, 0, 0, 0};
        for (int bb = 0; bb < 16; bb++) {
            const int u = 16 * (kq & 1) + bb - HALF_PATCH;
            const int val = (u >= -d && u <= d) ? (jm ? v : u) : 0;
            w[bb >> 2] |= (uint32_t)(val & 0xff) << (8 * (bb & 3));
        }
        s_bw[t] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;      // scalar: per-keypoint addressing goes to the SALU
    const int slot0 = (logical - b * nchunk) * KD_KPB;
    // ---- 0 ----
    float resp = 0.f;
    if (t < KD_KPB) {
        const int slot = slot0 + t;
        int level = -1, local = 0, total = 0;
        for (int l = 0; l < P.nlevels; l++) {
            const int c = selCount[b * MAXL + l];
            if (level < 0 && slot < total + c) { level = l; local = slot - total; }
            total += c;
        }
        if (slot == 0) {
            counts[b] = min(total, cap);
            if (total > cap && status) status[b] = MYSLAM_ERR_CAPACITY;
        }
        // work item `slot` = the local-th key of its level in PROCESSING order; it is written where the oct-tree's list order puts it
        int oslot = slot;
        if (level >= 0 && order) {
            const int orig = order[(size_t)b * P.totalOut + P.lv[level].outBase + local];
            oslot = slot - local + orig;
            local = orig;
        }
        const bool active = (level >= 0 && oslot < cap);
        s_out[t] = oslot;
        uint32_t pay = 0;
        if (active) pay = selOut[(size_t)b * P.totalOut + P.lv[level].outBase + local];
        s_lv[t] = active ? level : -1;
        s_x[t] = (int)((pay >> 8) & 0xfff) + MIN_BORDER; s_y[t] = (int)(pay >> 20) + MIN_BORDER;      // :897-898
        {   // keypoints keep >= 19 px from every border (ORBextractor.cpp:25): rows y-15 .. y+16 and columns x-15 .. x+16 exist
            const LevelGeom& gl = P.lv[active ? level : 0];
            const bool ext = active && level == 0 && b < P.ext0N;
            const uint32_t pp = ext ? (uint32_t)P.ext0Pitch : (uint32_t)gl.pitch;
            s_ppitch[t] = pp | (ext ? 0x80000000u : 0u);                  // bit 31: the patch lies in the caller's image
            s_pbase[t] = active ? (uint32_t)((ext ? 0 : gl.imgOff) + (size_t)(s_y[t] - HALF_PATCH) * pp + (s_x[t] - HALF_PATCH)) : 0u;
        }
        resp = (float)(pay & 0xff);
        if (detectOnly && active) {                // ORBextractor::Detect (:1067-1073): raw cv::FAST keypoints of level 0, no angle / descriptor
            myslam_keypoint kp;
            kp.x = (float)s_x[t]; kp.y = (float)s_y[t]; kp.size = 7.f; kp.angle = -1.f; kp.response = resp; kp.octave = 0; kp.class_id = -1;
            kps[(size_t)b * cap + oslot] = kp;
        }
    }
    if (detectOnly) return;                        // block-uniform
    __syncthreads();
    MYSLAM_BT_MARKP(0);
    // ---- A ----  intensity-centroid moments on the int8 matrix cores:
    //   [key-points x 64 k] x [64 k x {u weights, v weights}],  k = two patch rows of 32 pixels, v_mfma_i32_16x16x64_i8
    // A operand = the patch as it lies in memory (p - 128 as int8: the weights sum to zero over the symmetric mask, so the offset
    // cancels exactly); B operand = the masked weights (s_bw).
    // The patch of ONE key-point is fetched by one load instruction, lane = (row, 16-byte half): neighbouring lanes read neighbouring
    // bytes, so the texture addresser sees ~40 accesses per key-point (one or two per row) instead of the ~94 of loads that go
    // straight into the MFMA operand layout (lane = key-point x k block: 64 different rows per instruction).  Four key-points are
    // parked in LDS per round and read back in operand layout (rows 0 .. 3 of the 16-row tile; the matrix pipe has room for the idle
    // rows).  Phase timing (ms per 1024 images, cut-off builds): slot decode 0.05, this phase 0.60 -> 0.55, angle 0.02, BRIEF 0.68.
    {
        typedef int kd_v4i __attribute__((ext_vector_type(4)));
        const int r4 = lane & 15, kq = lane >> 4, jb = lane & 15;
        const int prow = lane >> 1, phalf = lane & 1;
        for (int bt = 0; bt < 4; bt++) {
            uint4 px[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int kk = wave * 16 + 4 * bt + j;
                const uint32_t ppk = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ppitch[kk]), ppitch = ppk & 0x7fffffffu;
                const uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_pbase[kk]);
                const uint8_t* pl = ((ppk >> 31) ? P.ext0 + (size_t)b * P.ext0Stride : pyr + (size_t)b * pyrStride) + pb;
                __builtin_memcpy(&px[j], pl + (uint32_t)(__mul24(prow, (int)ppitch) + 16 * phalf), 16);
            }
            __builtin_amdgcn_wave_barrier();                              // the previous round's operand reads are done (same wave)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint4 pv = px[j];
                pv.x ^= 0x80808080u; pv.y ^= 0x80808080u; pv.z ^= 0x80808080u; pv.w ^= 0x80808080u;
                s_b[wave][64 * j + lane] = pv;                            // [key-point][row][half]
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            kd_v4i acc = {0, 0, 0, 0};
#pragma unroll
            // operand row r4 (< 4) = key-point r4 of the round, k block kq.  Row i of the product depends on row i of A only and column j
            // on column j of B only, and only rows 0 .. 3 / columns 0, 1 are read back: the idle rows and columns simply repeat the live ones
            // (r4 & 3, jb & 1) — round 5: selecting zeros for them cost 8 v_mov + two exec-masked LDS reads + a full wait PER MFMA
            // (544 of a wave's ~3 000 vector instructions)
            for (int rp = 0; rp < 16; rp++) {
                const uint4 pv = s_b[wave][64 * (r4 & 3) + 2 * (2 * rp + (kq >> 1)) + (kq & 1)];
                const uint4 wv = s_bw[(rp * 4 + kq) * 2 + (jb & 1)];
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(kd_v4i, pv), __builtin_bit_cast(kd_v4i, wv), acc, 0, 0, 0);
            }
            // C/D: column = lane & 15 (0 -> m10, 1 -> m01), row (key-point of the round) = 4 (lane >> 4) + register: rows 0 .. 3 in lanes 0, 1
            if (lane < 2) {
                int* dstm = lane ? s_m01 : s_m10;
#pragma unroll
                for (int r = 0; r < 4; r++) dstm[wave * 16 + 4 * bt + r] = acc[r];
            }
        }
    }
    __syncthreads();
    MYSLAM_BT_MARKP(1);
    // ---- B ----
    if (t < KD_KPB && s_lv[t] >= 0) {
        const int level = s_lv[t];
        const LevelGeom& g = P.lv[level];
        const float angle = fast_atan2_deg((float)s_m01[t], (float)s_m10[t]);                      // :54
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        float ca, sb;
        det_sincos(__fmul_rn(angle, factorPI), sb, ca);
        s_ca[t] = ca; s_sb[t] = sb;
        const int x = s_x[t], y = s_y[t];
        myslam_keypoint kp;
        kp.x = (level != 0) ? __fmul_rn((float)x, g.scale) : (float)x;                            // :975-981
        kp.y = (level != 0) ? __fmul_rn((float)y, g.scale) : (float)y;
        kp.size = g.scaledPatch; kp.angle = angle; kp.response = resp; kp.octave = level; kp.class_id = -1;
        kps[(size_t)b * cap + s_out[t]] = kp;
    }
    __syncthreads();
    MYSLAM_BT_MARKP(2);
    // ---- C ----
    float pat[16];                                                    // this lane's 4 test pairs (x0 y0 x1 y1), ORBextractor.cpp:101-359
#pragma unroll
    for (int q = 0; q < 16; q++) pat[q] = (float)c_pattern[lane * 16 + q];
#pragma unroll
    for (int q = 0; q < 16; q++) asm volatile("" : "+v"(pat[q]));      // kept as floats: the compiler otherwise re-converts the packed int8 pattern for every key-point
#if MYSLAM_KD_GLDS
    // Direct-to-LDS form (global_load_lds_dwordx4): a wave keeps the windows of its next MYSLAM_KD_GLDS - 1 key-points in flight while it tests the current one.  Under
    // the pipeline this kernel runs two blocks per CU and a key-point costs its wave one memory latency (~1.5 us on the loaded chip: 25 of an item's 39 us); a
    // register prefetch one window deep hid nothing of that (profiles/r06_ab_describe_prefetch.json) and cost 14 registers, i.e. FAST blocks.  The LDS-DMA load
    // needs no registers for its data: lane i of a load writes LDS at (wave-uniform base) + 16 i, so the ROW-MAJOR window (row r at 64 r, as the test points
    // address it) is had by letting lane i fetch piece (row i >> 2, 16-byte column i & 3) from the TILED plane — the global address is per lane, any pattern.
    // Three loads per window as before: rows 0-15, 16-31, 32-36 (20 lanes).  Completion: vmcnt, waited for by hand (the compiler does not see the asm loads);
    // loads retire in order among themselves, so `3 x (windows issued after this one)` outstanding operations prove this window landed whatever the stores do.
    {
        constexpr int NBUF = MYSLAM_KD_GLDS, NJ = KD_KPB / 4;
        static_assert(NBUF >= 2 && NBUF <= 4, "windows in flight per wave");
        const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)&s_b[0][0]) + (uint32_t)wave * (uint32_t)sizeof(s_b[0]);
        auto level_of = [&](int j) -> int { return j < NJ ? __builtin_amdgcn_readfirstlane(s_lv[4 * j + wave]) : -1; };
        auto issue = [&](int j, int level) __attribute__((always_inline)) {            // window of key-point 4 j + wave -> buffer j % NBUF (level >= 0, wave-uniform)
            const int k = 4 * j + wave;
            const LevelGeom& g = P.lv[level];
            const int x = __builtin_amdgcn_readfirstlane(s_x[k]), y = __builtin_amdgcn_readfirstlane(s_y[k]);
            const int xb0 = (x - DB_R) & ~15, y0 = y - DB_R, pitch8 = g.pitch * 8;
            const uint8_t* base = blur + (size_t)b * pyrStride + g.imgOff + (size_t)(xb0 >> 4) * 128;
            const int yy = y0 + (lane >> 2);                                           // this lane's row of the first load; the others are 16 and 32 rows = 2 and 4 tile rows below
            const uint32_t voff = (uint32_t)__mul24(yy >> 3, pitch8) + (uint32_t)((yy & 7) << 4) + (uint32_t)((lane & 3) << 7);
            const uint32_t dst = lds0 + (uint32_t)(j % NBUF) * (uint32_t)(DB_N * 16);
#pragma unroll
            for (int q = 0; q < DB_IT; q++) {
                const uint8_t* src = base + (size_t)voff + (size_t)q * 2 * (size_t)pitch8;
                const uint32_t d = dst + 1024u * (uint32_t)q;
                uint32_t keep;
                if (q < 2 || lane < DB_N - 128)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(d) : "memory");
            }
        };
        int lv_cur = level_of(0);
        int lv_n[NBUF - 1];
#pragma unroll
        for (int u = 0; u < NBUF - 1; u++) { lv_n[u] = level_of(u + 1); }
        if (lv_cur >= 0) issue(0, lv_cur);
#pragma unroll
        for (int u = 0; u + 1 < NBUF - 1; u++) if (lv_n[u] >= 0) issue(u + 1, lv_n[u]);
        for (int j = 0; j < NJ; j++) {
            // the window NBUF - 1 key-points ahead goes into the buffer key-point j - 1 has just been tested from
            if (lv_n[NBUF - 2] >= 0) issue(j + NBUF - 1, lv_n[NBUF - 2]);
            int newer = 0;
#pragma unroll
            for (int u = 0; u < NBUF - 1; u++) newer += lv_n[u] >= 0 ? 1 : 0;
            const int level = lv_cur;
            lv_cur = lv_n[0];
#pragma unroll
            for (int u = 0; u + 1 < NBUF - 1; u++) lv_n[u] = lv_n[u + 1];
            lv_n[NBUF - 2] = level_of(j + NBUF);
            if (level < 0) continue;
            if (newer == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (newer == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (newer == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            const int k = 4 * j + wave;
            const int x = __builtin_amdgcn_readfirstlane(s_x[k]);
            const float ca = s_ca[k], sb = s_sb[k];
            const int offB = (x - DB_R) & 15;
            const uint8_t* center = reinterpret_cast<const uint8_t*>(s_b[wave]) + (j % NBUF) * (DB_N * 16) + DB_R * DB_P + offB + DB_R;
            uint32_t nib = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float x0 = pat[4 * q], y0 = pat[4 * q + 1], x1 = pat[4 * q + 2], y1 = pat[4 * q + 3];
                constexpr float RM = 12582912.f; constexpr uint32_t RK = 0x4B400000u;            // cvRound by the magic-number add, as in the register-staged form below
                const uint32_t r0 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(x0, sb), __fmul_rn(y0, ca)), RM));
                const uint32_t c0 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sb)), RM));
                const uint32_t r1 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(x1, sb), __fmul_rn(y1, ca)), RM));
                const uint32_t c1 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sb)), RM));
                static_assert(DB_P == 64, "row pitch of the LDS window as a shift");
                const int t0 = center[(int)((r0 << 6) + c0 - 65u * RK)], t1 = center[(int)((r1 << 6) + c1 - 65u * RK)];
                nib |= (uint32_t)(t0 < t1) << q;
            }
            const uint32_t byte = nib | (__shfl_down(nib, 1, 64) << 4);
            uint32_t w = byte | (__shfl_down(byte, 2, 64) << 8);
            w |= (__shfl_down(byte, 4, 64) << 16) | (__shfl_down(byte, 6, 64) << 24);
            if ((lane & 7) == 0) reinterpret_cast<uint32_t*>(desc + ((size_t)b * cap + s_out[k]) * 32)[lane >> 3] = w;
        }
    }
    return;
#endif
    for (int j = 0; j < KD_KPB / 4; j++) {
        const int k = 4 * j + wave;                                   // the four waves work on neighbouring key-points of the tile order: their windows overlap in L1
        const int level = __builtin_amdgcn_readfirstlane(s_lv[k]);
        if (level < 0) continue;
        const LevelGeom& g = P.lv[level];
        const int x = __builtin_amdgcn_readfirstlane(s_x[k]), y = __builtin_amdgcn_readfirstlane(s_y[k]);
        const float ca = s_ca[k], sb = s_sb[k];
        const int xb0 = (x - DB_R) & ~15, offB = (x - DB_R) - xb0;
        // The window in the TILED blurred plane: 4 tile columns (64 bytes at a 16-byte aligned column) x the 5 or 6 tile rows that hold
        // image rows y - 18 .. y + 18; chunk i = lane + 64 q in memory order (tile row i >> 5, tile column (i >> 3) & 3, row of the tile
        // i & 7): lanes 0-31 and 32-63 of a load each read 512 contiguous bytes = 4 whole cache lines; chunks of rows outside the window
        // are not requested.  The fourth tile column may lie past the image width (row padding or the next tile row): never read by a
        // test point.  The chunks land in LDS in row-major order (DB_P bytes per window row), where the test points address them.
        const int y0 = y - DB_R;
        const uint8_t* bl = blur + (size_t)b * pyrStride + g.imgOff + (size_t)(y0 >> 3) * g.pitch * 8 + (size_t)(xb0 >> 4) * 128;
        uint4 rb[DB_IT];
#pragma unroll
        for (int q = 0; q < DB_IT; q++) {
            const int i = lane + 64 * q;
            const int wr = ((i >> 5) << 3) + (i & 7) - (y0 & 7);              // window row of the chunk
            rb[q] = (wr >= 0 && wr < DB_ROWS) ? *reinterpret_cast<const uint4*>(bl + (uint32_t)(__mul24(i >> 5, g.pitch * 8) + 16 * (i & 31))) : make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();                                  // previous keypoint's window reads are done (same wave)
#pragma unroll
        for (int q = 0; q < DB_IT; q++) {
            const int i = lane + 64 * q;
            const int wr = ((i >> 5) << 3) + (i & 7) - (y0 & 7);
            if (wr >= 0 && wr < DB_ROWS) s_b[wave][wr * 4 + ((i >> 3) & 3)] = rb[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint8_t* center = reinterpret_cast<const uint8_t*>(s_b[wave]) + DB_R * DB_P + offB + DB_R;
        uint32_t nib = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float x0 = pat[4 * q], y0 = pat[4 * q + 1], x1 = pat[4 * q + 2], y1 = pat[4 * q + 3];
            // cvRound = round to nearest even: for |s| < 2^22, s + 1.5 * 2^23 has ulp 1, so its low mantissa bits ARE the rounded integer
            // (offset by the constant's bit pattern, folded into the address): one add instead of v_rndne + v_cvt per coordinate
            constexpr float RM = 12582912.f; constexpr uint32_t RK = 0x4B400000u;
            const uint32_t r0 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(x0, sb), __fmul_rn(y0, ca)), RM));
            const uint32_t c0 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(x0, ca), __fmul_rn(y0, sb)), RM));
            const uint32_t r1 = __float_as_uint(__fadd_rn(__fadd_rn(__fmul_rn(x1, sb), __fmul_rn(y1, ca)), RM));
            const uint32_t c1 = __float_as_uint(__fadd_rn(__fsub_rn(__fmul_rn(x1, ca), __fmul_rn(y1, sb)), RM));
            static_assert(DB_P == 64, "row pitch of the LDS window as a shift");
            const int t0 = center[(int)((r0 << 6) + c0 - 65u * RK)], t1 = center[(int)((r1 << 6) + c1 - 65u * RK)];
            nib |= (uint32_t)(t0 < t1) << q;
        }
        const uint32_t byte = nib | (__shfl_down(nib, 1, 64) << 4);            // valid on even lanes
        uint32_t w = byte | (__shfl_down(byte, 2, 64) << 8);
        w |= (__shfl_down(byte, 4, 64) << 16) | (__shfl_down(byte, 6, 64) << 24);   // valid on lanes % 8 == 0
        if ((lane & 7) == 0) reinterpret_cast<uint32_t*>(desc + ((size_t)b * cap + s_out[k]) * 32)[lane >> 3] = w;
    }
}

// ------------------------------------------------------------------------------------------------
// K6/K7: per-keypoint operators of the loop-closing path (one image, n keypoints; wave per keypoint)
// ------------------------------------------------------------------------------------------------
// isFastCorner (ORBextractor.cpp:449-511): > 8 contiguous ring pixels darker / brighter than v -/+ th
// The kernel: a LIMITED grid of blocks, each walking a contiguous share of the (image, 64-key-point chunk) work items of its XCD.
// Why limited (round 4): under the pipeline this kernel runs beside the other handle's grid-FAST launch, which is what the step is bound by
// (VALU issue).  A block here lives ~40 us against ~13 us of a FAST block, so with an unlimited grid every slot a FAST block frees is taken by
// a long-lived block sooner or later: the latency-bound kernel ends up holding most of the CUs while the VALU-bound one starves (kernel
// timeline of round 4: FAST made 20 % of its progress in the 2.5 ms the other handle's kernels ran, 80 % in the 1.4 ms after them).  With
// at most `gridDim.x / 256` blocks per CU the rest of the CU stays FAST's.
// XCD-aware order: the dispatcher places block i on XCD i % 8, each XCD has a private L2, and the ~32 blocks of one image read overlapping
// windows of the same two pyramids: every XCD walks a contiguous range of images (measured 3.2 GB -> ~1.3 GB of HBM reads per 512 images).
__global__ __launch_bounds__(256) void k_describe2(OrbPlan P, const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                   size_t pyrStride, const uint32_t* __restrict__ selOut,
                                                   const int32_t* __restrict__ selCount, myslam_keypoint* __restrict__ kps,
                                                   uint8_t* __restrict__ desc, int32_t* __restrict__ counts,
                                                   int32_t* __restrict__ status, int cap, int nchunk, int batch, int detectOnly,
                                                   const uint16_t* __restrict__ order) {
    MYSLAM_SIDE_PRIO();
    const int total = nchunk * batch, nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3, per = (nwg + 7 - xcd) >> 3;              // blocks of this XCD: ids xcd, xcd + 8, ...
    const int tq = total >> 3, tr = total & 7;
    const int lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, n = tq + (xcd < tr ? 1 : 0);      // this XCD's contiguous range
#pragma nounroll
    for (int k = j; k < n; k += per) {
        MYSLAM_BT(4);
        describe_block(P, pyr, blur, pyrStride, selOut, selCount, kps, desc, counts, status, cap, nchunk, batch, detectOnly, order, lo + k MYSLAM_BT_ARG);
        __syncthreads();                               // the next item reuses the block's LDS
    }
}

__device__ __forceinline__ bool is_fast_corner(const uint8_t* __restrict__ img, int pitch, int x, int y, int th) {
    constexpr int RX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int RY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    th = min(max(th, 0), 255);
    const uint8_t* p = img + (size_t)y * pitch + x;
    const int v = p[0];
    unsigned dm = 0, bm = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int pv = p[RX[k] + RY[k] * pitch];
        dm |= (unsigned)(pv < v - th) << k;
        bm |= (unsigned)(pv > v + th) << k;
    }
    auto run9 = [](unsigned m) {
        unsigned x = m | (m << 16);       // unrolled ring
        unsigned r = x & (x >> 1);
        r &= r >> 2;
        r &= r >> 4;                      // runs of 8
        r &= x >> 8;                      // runs of 9
        return (r & 0xffffu) != 0;
    };
    return run9(dm) || run9(bm);
}

// ScreenAndComputeKPsParams (ORBextractor.cpp:1098-1127): in/out keypoint i, keep flag
__global__ __launch_bounds__(256) void k_screen(OrbPlan P, const uint8_t* __restrict__ pyr, myslam_keypoint* __restrict__ kin,
                                                int n, myslam_keypoint* __restrict__ kout, uint8_t* __restrict__ keep) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    myslam_keypoint k = kin[i];
    const int level = k.octave;
    if (level < 0 || level >= P.nlevels) { if (lane == 0) keep[i] = 0; return; }
    const LevelGeom& g = P.lv[level];
    const float scale = g.scale;
    k.x = __fdiv_rn(k.x, scale); k.y = __fdiv_rn(k.y, scale);                                   // :1104
    bool ok = (__fsub_rn(k.y, (float)EDGE_THRESHOLD) >= 0 && __fadd_rn(k.y, (float)EDGE_THRESHOLD) < (float)g.h &&
               __fsub_rn(k.x, (float)EDGE_THRESHOLD) >= 0 && __fadd_rn(k.x, (float)EDGE_THRESHOLD) < (float)g.w);
    const uint8_t* img = pyr + g.imgOff;
    int px = 0, py = 0;
    if (ok) {
        px = __float2int_rn(k.x); py = __float2int_rn(k.y);
        ok = is_fast_corner(img, g.pitch, px, py, P.minTh);                                     // :1112
    }
    if (ok) {                                                                                   // wave-uniform
        k.angle = wave_ic_angle(img, g.pitch, px, py);                                          // :1118
        k.size = __fmul_rn((float)PATCH_SIZE, scale);                                           // :1121
    }
    k.x = __fmul_rn(k.x, scale); k.y = __fmul_rn(k.y, scale);                                   // :1108/1113/1123
    if (lane == 0) { kin[i] = k; kout[i] = k; keep[i] = ok ? 1 : 0; }
}

// CalcDescriptors (ORBextractor.cpp:1210-1223)
__global__ __launch_bounds__(256) void k_calc_desc(OrbPlan P, const uint8_t* __restrict__ blur, const myslam_keypoint* __restrict__ kps,
                                                   int n, uint8_t* __restrict__ desc) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const myslam_keypoint k = kps[i];
    const int level = min(max(k.octave, 0), P.nlevels - 1);
    const LevelGeom& g = P.lv[level];
    const int px = __float2int_rn(__fdiv_rn(k.x, g.scale)), py = __float2int_rn(__fdiv_rn(k.y, g.scale));
    const uint32_t w = wave_brief<true>(blur + g.imgOff, g.pitch, px, py, k.angle);
    if ((lane & 7) == 0) reinterpret_cast<uint32_t*>(desc + (size_t)i * 32)[lane >> 3] = w;
}

// unpack a level's candidate list for the debug tap
__global__ void k_unpack_cands(const uint32_t* __restrict__ cand, int n, int32_t* xs, int32_t* ys, int32_t* sc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = cand[i];
    xs[i] = (p >> 8) & 0xfff; ys[i] = p >> 20; sc[i] = p & 0xff;
}

// ------------------------------------------------------------------------------------------------
// K0: ingest level 0 from the caller's buffer (any pitch/alignment) into the 64-byte pitched plane
// ------------------------------------------------------------------------------------------------
// The per-call counters (candidate / selection counts, status words, FAST statistics) are cleared by the FIRST kernel of a call — this
// one — instead of by hipMemsetAsync: no launch of their own (a one-frame call is a chain of short dependent launches), and kernel nodes
// are the part of a captured HIP graph that replays reliably (memset nodes of a replayed graph left garbage in the counters on ROCm 7.2).
struct ZeroArgs { uint32_t* p[4]; int n[4]; };
__device__ __forceinline__ void zero_part(const ZeroArgs& z, int tid, int T) {
#pragma unroll
    for (int k = 0; k < 4; k++)
        for (int i = tid; i < z.n[k]; i += T) z.p[k][i] = 0u;
}
__device__ __forceinline__ void ingest_part(const uint8_t* __restrict__ src, int rows, int cols, int step, size_t sstride,
                                            uint8_t* __restrict__ dst, int dpitch, size_t dstride, int tpr, int rpb, int bx, int by, int b) {
    // 16 destination bytes per thread (one aligned 16-byte store) from 5 aligned source dwords + v_alignbyte; tpr threads per row
    const int y = by * rpb + (int)threadIdx.x / tpr;
    const int x16 = (bx * 256 + (int)threadIdx.x % tpr) * 16;
    if ((int)threadIdx.x >= tpr * rpb || y >= rows || x16 >= cols) return;
    const uint8_t* s = src + (size_t)b * sstride + (size_t)y * step + x16;
    const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(s) & 3);
    const uint32_t* sa = reinterpret_cast<const uint32_t*>(s - sh);
    const int nvalid = cols - x16;                                   // bytes of this row still to copy (>= 1)
    uint32_t d[5];
#pragma unroll
    for (int j = 0; j < 5; j++) d[j] = (4 * j < (int)sh + nvalid) ? sa[j] : 0u;     // a dword is read only if it holds a byte of the row
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(d[1], d[0], sh); o.y = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
    o.z = __builtin_amdgcn_alignbyte(d[3], d[2], sh); o.w = __builtin_amdgcn_alignbyte(d[4], d[3], sh);
    uint8_t* q = dst + (size_t)b * dstride + (size_t)y * dpitch + x16;
    if (x16 + 16 <= dpitch) *reinterpret_cast<uint4*>(q) = o;        // pitch is a multiple of 64: padding bytes may be written
    else { uint32_t w[4] = {o.x, o.y, o.z, o.w}; for (int j = 0; j < 4 && x16 + 4 * j < dpitch; j++) reinterpret_cast<uint32_t*>(q)[j] = w[j]; }
}
__global__ __launch_bounds__(256) void k_ingest(const uint8_t* __restrict__ src, int rows, int cols, int step, size_t sstride,
                                                uint8_t* __restrict__ dst, int dpitch, size_t dstride, int tpr, int rpb, ZeroArgs z) {
    zero_part(z, (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + (int)threadIdx.x, (int)(gridDim.x * gridDim.y * gridDim.z) * 256);
    ingest_part(src, rows, cols, step, sstride, dst, dpitch, dstride, tpr, rpb, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The first launch of a call with little work (a live stream's frame): level-0 ingest of the images that are not read in place + the per-call
// counters + the first pyramid levels of ALL images, computed from the caller's buffers (k_resize_chain's per-pixel byte loads never read past
// a row, so the last image of a batch needs no copy for THEM) — one node of a recorded step instead of three.
struct IngestPart { const uint8_t* src; int rows, cols, step; size_t sstride; uint8_t* dst; int dpitch; size_t dstride; int tpr, rpb, gx, b0; };
__global__ __launch_bounds__(256) void k_pyr_head(ResizeChain c, IngestPart g, ZeroArgs z) {
    const int blk = blockIdx.x, b = blockIdx.y;
    zero_part(z, (int)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + (int)threadIdx.x, (int)(gridDim.x * gridDim.y) * 256);
    if (blk < c.blk0[RC_MAX]) { rc_block(c, blk, b); return; }
    if (b < g.b0) return;                                            // read in place: no level-0 copy
    const int ib = blk - c.blk0[RC_MAX];
    ingest_part(g.src, g.rows, g.cols, g.step, g.sstride, g.dst, g.dpitch, g.dstride, g.tpr, g.rpb, ib % g.gx, ib / g.gx, b);
}

__global__ __launch_bounds__(256) void k_zero_u32(ZeroArgs a) {             // the same clearing on its own (debug entry points)
    const int i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (a.p[k] && i < a.n[k]) a.p[k][i] = 0u;
}
void launch_zero_u32(uint32_t* p0, int n0, uint32_t* p1, int n1, uint32_t* p2, int n2, uint32_t* p3, int n3, hipStream_t s) {
    ZeroArgs a{{p0, p1, p2, p3}, {n0, n1, n2, n3}};
    const int n = max(max(n0, n1), max(n2, n3));
    if (n > 0) hipLaunchKernelGGL(k_zero_u32, dim3((n + 255) / 256), dim3(256), 0, s, a);
}

static void ingest_launch(const uint8_t* src, int rows, int cols, int step, size_t sstride, uint8_t* dst, int dpitch,
                          size_t dstride, int batch, const ZeroArgs& z, hipStream_t s) {
    const int t = (cols + 15) / 16;
    const int tpr = t < 256 ? t : 256, rpb = t < 256 ? 256 / t : 1;
    dim3 grid((t + 255) / 256, (rows + rpb - 1) / rpb, batch);
    hipLaunchKernelGGL(k_ingest, grid, dim3(256), 0, s, src, rows, cols, step, sstride, dst, dpitch, dstride, tpr, rpb, z);
}
void launch_ingest(const uint8_t* src, int rows, int cols, int step, size_t sstride, uint8_t* dst, int dpitch,
                   size_t dstride, int batch, hipStream_t s) {
    ingest_launch(src, rows, cols, step, sstride, dst, dpitch, dstride, batch, ZeroArgs{{nullptr, nullptr, nullptr, nullptr}, {0, 0, 0, 0}}, s);
}
// the call's first launch: level-0 ingest + the per-call counters
void launch_ingest_clear(const uint8_t* src, int rows, int cols, int step, size_t sstride, uint8_t* dst, int dpitch, size_t dstride, int batch,
                         uint32_t* p0, int n0, uint32_t* p1, int n1, uint32_t* p2, int n2, uint32_t* p3, int n3, hipStream_t s) {
    ingest_launch(src, rows, cols, step, sstride, dst, dpitch, dstride, batch, ZeroArgs{{p0, p1, p2, p3}, {n0, n1, n2, n3}}, s);
}

// ------------------------------------------------------------------------------------------------
// launch helpers (called from orb_engine.hip)
// ------------------------------------------------------------------------------------------------
bool resize_uses_strips(const ResizeArgs& a) { return (double)RS_R * a.scale_y + 2.0 <= (double)RS_MAXR && a.scale_x <= 1.6; }

bool resize_is_little(const ResizeArgs& a, int batch) { return a.n0 == 0 && (size_t)batch * a.dw * a.dh < (size_t)1500000; }

#ifndef MYSLAM_RESIZE_CHAIN_MAX        // A/B builds (tools/build_variants.sh): levels per launch of the small-batch pyramid (1 = a launch per level)
#define MYSLAM_RESIZE_CHAIN_MAX 3
#endif
int resize_chain_max() { return MYSLAM_RESIZE_CHAIN_MAX < RC_MAX ? MYSLAM_RESIZE_CHAIN_MAX : RC_MAX; }

static ResizeChain make_chain(const ResizeArgs* lv, int n) {
    ResizeChain c;
    c.src = lv[0].src; c.sw = lv[0].sw; c.sh = lv[0].sh; c.spitch = lv[0].spitch; c.sstride = lv[0].sstride;
    c.n = n; c.dstride = lv[0].dstride; c.blk0[0] = 0;
    for (int j = 0; j < RC_MAX; j++) {
        const ResizeArgs& a = lv[j < n ? j : n - 1];
        c.dst[j] = a.dst; c.dw[j] = a.dw; c.dh[j] = a.dh; c.dpitch[j] = a.dpitch; c.scale_x[j] = a.scale_x; c.scale_y[j] = a.scale_y;
        c.blk0[j + 1] = c.blk0[j] + (j < n ? (((a.dw + 3) / 4) * (4 / rc_px(j + 1)) * a.dh + 255) / 256 : 0);
    }
    return c;
}
// n consecutive levels (lv[j].src is lv[j - 1].dst), every one of them "little": one launch
void launch_resize_chain(const ResizeArgs* lv, int n, int batch, hipStream_t s) {
    const ResizeChain c = make_chain(lv, n);
    hipLaunchKernelGGL(k_resize_chain, dim3(c.blk0[RC_MAX], batch), dim3(256), 0, s, c);
}
// levels 1 .. n from the caller's images (lv[0].src / spitch / sstride = the caller's buffer) + ingest of images b >= b0 + the per-call counters
void launch_pyr_head(const ResizeArgs* lv, int n, int rows, int cols, uint8_t* dst0, int dpitch0, size_t dstride0, int b0, int batch,
                     uint32_t* p0, int n0, uint32_t* p1, int n1, uint32_t* p2, int n2, uint32_t* p3, int n3, hipStream_t s) {
    const ResizeChain c = make_chain(lv, n);
    const int t = (cols + 15) / 16;
    IngestPart g{lv[0].src, rows, cols, lv[0].spitch, lv[0].sstride, dst0, dpitch0, dstride0, t < 256 ? t : 256, t < 256 ? 256 / t : 1, (t + 255) / 256, b0};
    const int iblocks = g.gx * ((rows + g.rpb - 1) / g.rpb);
    hipLaunchKernelGGL(k_pyr_head, dim3(c.blk0[RC_MAX] + iblocks, batch), dim3(256), 0, s, c, g, ZeroArgs{{p0, p1, p2, p3}, {n0, n1, n2, n3}});
}
// (measured at 1 pair x 16 lanes, tools/ab_stream_mode.sh, us per step / graph nodes: no head launch, chains of 3: 89.3 / 18; head of 2 levels + chains of 3:
// 85.2 / 17; head of 3 + chains of 2: 86.9 / 17; head of 3 + one chain of 4: 86.5 / 16 — a node costs ~4.6 us, and the 21 / 85 interpolations per
// pixel of a third / fourth chained level begin to cost as much)
#ifndef MYSLAM_PYR_HEAD_LEVELS         // A/B builds: levels the head launch of a small batch produces (0 = no head launch)
#define MYSLAM_PYR_HEAD_LEVELS 2
#endif
int pyr_head_levels() { return MYSLAM_PYR_HEAD_LEVELS < 3 ? MYSLAM_PYR_HEAD_LEVELS : 3; }

void launch_resize(const ResizeArgs& a, int batch, hipStream_t s) {
    // register strips cover pyramid scale factors up to 1.25 (rows) / 1.6 (columns); larger steps take the generic kernel — and so do
    // launches with little work (a one-frame call: a level is ~100 strip waves that each walk their band serially, 8.5 us per level and
    // seven dependent levels; one thread per four destination pixels finishes a level in a third of that).  Same arithmetic, bit for bit
    // (tests/test_gpu_fallbacks.py); images read in place (n0 > 0: batched calls only) need the strip form.
    const bool little = resize_is_little(a, batch);
    if (resize_uses_strips(a) && !little) {
        const int nstrips = (a.dw + 255) / 256, nbands = (a.dh + RS_R - 1) / RS_R;
        const int ngroups = (nbands + RS_NB - 1) / RS_NB;
        hipLaunchKernelGGL(k_resize_strip, dim3((nstrips * ngroups + 3) / 4, 1, batch), dim3(256), 0, s, a, nstrips, nbands);
        return;
    }
    dim3 grid((a.dw + 255) / 256, (a.dh + 3) / 4, batch);
    hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, s, a);
}

bool blur_uses_strips(const BlurArgs& a) {
    return a.w >= 8 && a.h >= 8 && (a.spitch & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.src) | a.sstride) & 3) == 0;
}

// n levels (n <= MAXL): the ones the register-strip kernel can take share one launch, the others (unaligned rows, images narrower
// than 8 pixels) go through the LDS-tiled kernel one by one
void launch_blur_levels(const BlurArgs* lv, int n, int batch, hipStream_t s) {
    BlurMulti M; M.n = 0; M.wave0[0] = 0;
    BlurMfmaMulti X; X.n = 0; X.wave0[0] = 0;
    for (int i = 0; i < n; i++) {
        const BlurArgs& a = lv[i];
        if (a.tabH && a.tabV && a.ident && a.dtiled && (a.dpitch & 15) == 0 && ((reinterpret_cast<uintptr_t>(a.dst) | a.dstride) & 15) == 0) {
            // matrix-core form: one wave per 32-column strip of the level, all such levels in one launch.  (tabH / tabV already point at
            // the level's blocks: the per-level offsets of the multi-launch stay 0)
            X.a[X.n] = a; X.tabHOff[X.n] = 0; X.tabVOff[X.n] = 0; X.wave0[X.n + 1] = X.wave0[X.n] + (a.w + 31) / 32; X.n++;
        } else if (blur_uses_strips(a)) {
            const int nstrips = (a.w + 255) / 256, nbands = (a.h + B3_R - 1) / B3_R;
            M.a[M.n] = a; M.nstrips[M.n] = nstrips; M.wave0[M.n + 1] = M.wave0[M.n] + nstrips * nbands; M.n++;
        } else {
            dim3 grid((a.w + B2_W - 1) / B2_W, (a.h + B2_H - 1) / B2_H, batch);
            hipLaunchKernelGGL(k_blur7_dot, grid, dim3(256), 0, s, a);
        }
    }
    if (M.n) hipLaunchKernelGGL(k_blur7_strip, dim3((M.wave0[M.n] + 3) / 4, 1, batch), dim3(256), 0, s, M);
    if (X.n) hipLaunchKernelGGL(k_blur7_mfma, dim3((X.wave0[X.n] + 3) / 4, 1, batch), dim3(256), 0, s, X);
}
void launch_blur(const BlurArgs& a, int batch, hipStream_t s) { launch_blur_levels(&a, 1, batch, s); }

#ifndef MYSLAM_BLUR_WITH_OCTREE        // A/B builds (tools/build_variants.sh): 0 = the Gaussian keeps its own launch
#define MYSLAM_BLUR_WITH_OCTREE 1
#endif
// small batches: all levels as register strips in the oct-tree's launch — possible when every level takes the strip form (no matrix-core option, aligned planes)
bool blur_multi_for_octree(const BlurArgs* lv, int n, int batch, BlurMulti& M) {
    // (up to 4 pairs per call: +5 % frames/s at 1 - 2 pairs, even at 4; at 8 and 16 pairs the bands, which then wait for 512-thread blocks with the
    // oct-tree's LDS, cost 1 - 3 %: profiles/r05_ab_pairs_8_16.log)
    if (!MYSLAM_BLUR_WITH_OCTREE || batch >= OCT_WIDE_BELOW || batch >= 16) return false;
    M.n = 0; M.wave0[0] = 0;
    for (int i = 0; i < n; i++) {
        const BlurArgs& a = lv[i];
        if ((a.tabH && a.tabV) || !blur_uses_strips(a)) return false;
        const int nstrips = (a.w + 255) / 256, nbands = (a.h + B3_R - 1) / B3_R;
        M.a[M.n] = a; M.nstrips[M.n] = nstrips; M.wave0[M.n + 1] = M.wave0[M.n] + nstrips * nbands; M.n++;
    }
    return M.n > 0;
}

void launch_fast(const OrbPlan& P, const uint8_t* pyr, size_t pyrStride, const uint8_t* maskPyr, uint32_t* cand,
                 int32_t* candCount, const uint32_t* statPrev, uint32_t* statCur, int forceMode, int batch, hipStream_t s) {
    int cw = 0, ch = 0;
    for (int l = 0; l < P.nlevels; l++) { cw = max(cw, P.lv[l].wCell); ch = max(ch, P.lv[l].hCell); }
    const FastCtl ctl{statPrev, statCur, forceMode};
    const dim3 grid((unsigned)P.nstrips * (unsigned)batch);
    constexpr int pad = FastLds<32, 4, 40>::PAD;                          // dynamic LDS: see FastLds
#if MYSLAM_FAST_NS > 1
    // the multi-strip form (blocks of MYSLAM_FAST_NS consecutive strips, the next strip's tile prefetched by LDS-DMA) for launches that fill the chip several times over
    if (cw <= 32 && ch <= 40 && (size_t)P.nstrips * batch >= (size_t)MYSLAM_FAST_NS * 256 * 6 * 4) {
        const dim3 gridms((unsigned)(((size_t)P.nstrips * batch + MYSLAM_FAST_NS - 1) / MYSLAM_FAST_NS));
        constexpr int padms = FastLds<32, 4, 40, MYSLAM_FAST_NS>::PAD + FastLds<32, 4, 40, MYSLAM_FAST_NS>::TROWS * FastLds<32, 4, 40, MYSLAM_FAST_NS>::TP + 16;      // the second tile buffer + the pad
        hipLaunchKernelGGL((k_fast_strip<32, 4, 40, MYSLAM_FAST_NS>), gridms, dim3(256), padms, s, P, pyr, pyrStride, maskPyr, cand, candCount, ctl, batch);
        return;
    }
#endif
    if (cw <= 32 && ch <= 40) hipLaunchKernelGGL((k_fast_strip<32, 4, 40>), grid, dim3(256), pad, s, P, pyr, pyrStride, maskPyr, cand, candCount, ctl, batch);
    else if (max(cw, ch) <= 40) hipLaunchKernelGGL((k_fast_strip<40, 4>), grid, dim3(256), 0, s, P, pyr, pyrStride, maskPyr, cand, candCount, ctl, batch);
    else hipLaunchKernelGGL((k_fast_strip<MAX_CELL, 4>), grid, dim3(256), 0, s, P, pyr, pyrStride, maskPyr, cand, candCount, ctl, batch);
}

size_t octree_lds_bytes(int nodeCap) { return 192 + 4 * (size_t)OT_MAXB + 4 * (size_t)(OT_MAXB + 2) + (size_t)nodeCap * 54 + 16; }

bool blur_multi_for_octree(const BlurArgs* lv, int n, int batch, BlurMulti& M);
// returns true when the Gaussian of the levels blurLv[0 .. nBlur) rode in the launch (small batches; the caller then skips its blur launch)
bool launch_octree(const OrbPlan& P, const uint32_t* cand, const int32_t* candCount, uint32_t* sortbuf, const uint32_t* octTab, uint32_t* selOut,
                   int32_t* selCount, int32_t* status, int batch, uint16_t* order, hipStream_t s, const BlurArgs* blurLv, int nBlur) {
    BlurMulti BMh;
    const BlurMulti* blurWith = (blurLv && nBlur > 0 && blur_multi_for_octree(blurLv, nBlur, batch, BMh)) ? &BMh : nullptr;
    // one launch for all levels, grid (image, level): dispatched x-fastest, the long level-0 blocks all start first and the small levels fill
    // the gaps at the end (with the level along x the launch ended on the last images' level-0 blocks: 0.37 instead of 0.22 ms per 512 images)
    int ncmax = 0;
    for (int l = 0; l < P.nlevels; l++) ncmax = max(ncmax, P.lv[l].nodeCap);
    const size_t lds = std::max(octree_lds_bytes(ncmax), (size_t)8 * 16 * B3_TS);
    // The limit is state of the FUNCTION (per device and process), not of a launch: it is always raised to the device's whole LDS, never
    // to this launch's own size — a handle with a smaller plan (or another thread) would otherwise lower it under a launch that is still
    // to come, e.g. the replay of a captured HIP graph (a memory fault, found with two extractor handles of different budgets).
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_octree<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_octree<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // (a grid limit as in launch_describe was measured for this kernel too, round 4: 1 / 2 / 3 blocks per CU gave 7.27 / 7.11 / 7.15 ms per step
    // against 7.11 unlimited, and the loop itself cost 0.1 ms — not built in)
    if (batch >= OCT_WIDE_BELOW)
        hipLaunchKernelGGL(k_octree<256>, dim3(batch, P.nlevels), dim3(256), lds, s, P, cand, candCount, sortbuf, octTab, selOut, selCount, status, ncmax, order, BlurMulti{});
    else
    {   // small batches: the Gaussian of all levels rides in the same launch (blocks behind the oct-tree's, 8 bands each) — it depends on the
        // pyramid only, and a recorded step of a few frames is bound by the number of its launches
        BlurMulti M = blurWith ? *blurWith : BlurMulti{};
        if (!blurWith) M.n = 0;
        const int brows = blurWith ? (M.wave0[M.n] + 7) / 8 : 0;
        hipLaunchKernelGGL(k_octree<512>, dim3(batch, P.nlevels + brows), dim3(512), lds, s, P, cand, candCount, sortbuf, octTab, selOut, selCount, status, ncmax, order, M);
    }
    return blurWith != nullptr;
}

bool describe_uses_tile_order(bool have_order, int detectOnly, int batch) { return have_order && !detectOnly && batch >= 8; }

void launch_describe(const OrbPlan& P, const uint8_t* pyr, const uint8_t* blur, size_t pyrStride, const uint32_t* selOut,
                     const int32_t* selCount, myslam_keypoint* kps, uint8_t* desc, int32_t* counts, int32_t* status,
                     int cap, int detectOnly, int batch, uint16_t* order, bool order_ready, int blocks_per_cu, hipStream_t s) {
    const int slots = min(cap, P.totalOut);
    const int nchunk = (slots + KD_KPB - 1) / KD_KPB;
    // the tile-order permutation pays when thousands of windows compete for L1 / L2; a handful of images is a few dozen blocks: list order
    const bool tiled = describe_uses_tile_order(order != nullptr, detectOnly, batch);
    if (tiled && !order_ready) hipLaunchKernelGGL(k_sel_order, dim3(P.nlevels, batch), dim3(256), 0, s, P, selOut, selCount, order, batch);
    const int total = nchunk * batch;
    const int grid = blocks_per_cu > 0 ? min(total, 256 * blocks_per_cu) : total;       // MYSLAM_ORB_OPT_SIDE_BLOCKS_PER_CU
    hipLaunchKernelGGL(k_describe2, dim3(grid), dim3(256), 0, s, P, pyr, blur, pyrStride, selOut, selCount,
                       kps, desc, counts, status, cap, nchunk, batch, detectOnly, tiled ? order : nullptr);
}

void launch_screen(const OrbPlan& P, const uint8_t* pyr, myslam_keypoint* kin, int n, myslam_keypoint* kout, uint8_t* keep,
                   hipStream_t s) {
    hipLaunchKernelGGL(k_screen, dim3((n + 3) / 4), dim3(256), 0, s, P, pyr, kin, n, kout, keep);
}

void launch_calc_desc(const OrbPlan& P, const uint8_t* blur, const myslam_keypoint* kps, int n, uint8_t* desc, hipStream_t s) {
    hipLaunchKernelGGL(k_calc_desc, dim3((n + 3) / 4), dim3(256), 0, s, P, blur, kps, n, desc);
}

void launch_unpack_cands(const uint32_t* cand, int n, int32_t* xs, int32_t* ys, int32_t* sc, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_unpack_cands, dim3((n + 255) / 256), dim3(256), 0, s, cand, n, xs, ys, sc);
}

}  // namespace myslam_hip

#ifdef MYSLAM_BLOCK_TRACE
// profiling builds only (see BlockTrace above): d_buf = cap_records x 32 bytes of device memory, nullptr = stop recording; *n = records written so far
extern "C" int myslam_debug_block_trace(void* d_buf, unsigned cap_records, unsigned* n) {
    using namespace myslam_hip;
    unsigned long long* p = (unsigned long long*)d_buf;
    const unsigned zero = 0;
    if (n) MYSLAM_HIP_CHECK(hipMemcpyFromSymbol(n, HIP_SYMBOL(g_bt_n), sizeof(unsigned)));
    if (d_buf) MYSLAM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_bt_n), &zero, sizeof(unsigned)));
    MYSLAM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_bt_cap), &cap_records, sizeof(unsigned)));
    MYSLAM_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_bt_buf), &p, sizeof(p)));
    return MYSLAM_OK;
}
#endif
