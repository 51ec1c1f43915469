// match_tri.hip — 256-bit Hamming brute-force matcher and stereo triangulation for gfx950.
//
// Hamming: replaces cv::BFMatcher(NORM_HAMMING)::match (reference src/loopclosing.cpp:33,172): one
// best train row per query row, ties -> lowest train index: a +-1 product on the FP4 matrix cores.
//
// Triangulation: replaces triangulation() (reference include/myslam/algorithm.h:16-33) for the
// stereo rig (src/system.cpp:108-116,141-145): DLT 4x4, smallest right singular vector by one-sided
// Jacobi in f64, accept iff sigma3/sigma2 < 1e-2 and z > 0 (src/frontend.cpp:400,471).
#include <algorithm>
#include <utility>
#include <vector>

#include "common.h"

namespace myslam_hip {

// ------------------------------------------------------------------------------------------------
// Hamming brute force on the matrix cores.  With every descriptor bit b encoded as s = 1 - 2b,
//     sum_k s_q[k] s_t[k] = 256 - 2 * hamming(q, t),
// so the 2000 x 2000 distance matrix of a stereo pair is a [nt x 256] x [256 x nq] product.
//   * block = 4 waves x HQ_TILES query tiles of 32; the query operand (B) is expanded ONCE into registers and reused against
//     every train chunk
//   * train descriptors stay bit-packed in HBM; each chunk of 32 is expanded into LDS by the whole block through a 256-entry
//     byte table, double buffered, one barrier per chunk
//   * A (train) and B (query) use the same (lane, byte) -> k map, so the hardware's k order inside an operand is irrelevant
//   * the 32x32 result puts ONE query column in every lane (16 train rows in its registers): the running best is a single
//     register per tile, key = (dot + 256) << 20 | (0xFFFFF - train index), maximised with v_max3 — highest dot = smallest
//     distance, ties to the lowest train index (BFMatcher keeps the first minimum); the two half-waves merge at the end.
// (History: xor / popcount on the VALU ran at its issue peak, 1.00 ms per 512 pairs; int8 MFMA 0.40 ms; this FP4 form 0.23 ms.)
constexpr int HQ_TILES = 4;                       // query tiles of 32 per wave -> 512 queries per block
constexpr int HQ_BLOCK = 4 * HQ_TILES * 32;
constexpr int HQ_NARROW_BELOW = 16;               // fewer pairs than this per call: one query tile per wave (k_hamming_fp4<1, HQ_SPLIT>)
#ifndef MYSLAM_HQ_SPLIT                           // A/B builds (tools/build_variants.sh)
#define MYSLAM_HQ_SPLIT 4
#endif
constexpr int HQ_SPLIT = MYSLAM_HQ_SPLIT;         // ... and this many wave groups per block, each on every HQ_SPLIT-th train chunk

// The product runs on the FP4 path of the matrix cores (v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 per instruction, twice the int8
// rate): E2M1 represents +-1 exactly (0x2 / 0xA), the factor 32 of the query operand is its E8M0 block scale (2^5), sums of at most
// 256 terms of +-32 plus the (31 - row) start value are exact in f32.  One descriptor dword (32 bits) expands to the 16 operand bytes
// of a lane; 4 MFMAs per 32 x 32 tile instead of 8, 16 query registers per tile instead of 32, 4 KB of LDS per train chunk.
// the triangulation of ONE left key-point against its match (defined below, behind tri_solve): also the tail of the one-pair matcher
struct TriTail {
    const myslam_keypoint* kl; const myslam_keypoint* kr;        // kl == nullptr: no tail
    double fx, fy, cx, cy, baseline; double* xyz; uint8_t* ok;
};
__device__ __forceinline__ void tri_point(const TriTail& tt, int p, int cap, int i, int j);

typedef int hq_v8i __attribute__((ext_vector_type(8)));
typedef float hq_v16f __attribute__((ext_vector_type(16)));
constexpr int HF_ROWB = 144;                      // LDS bytes per expanded train row (128 + 16)

// TILES = query tiles of 32 per wave: 4 (512 queries per block) for batches; 1 (128 per block) for a handful of pairs — the block's loop
// over the train chunks is a chain of barrier-separated steps, and with one tile per wave a step is a quarter as long while four times
// as many blocks run side by side (one pair of 2 000 x 2 000: 50 -> ~25 us).  Same bits either way (integer arg-min, ties by row).
// SPLIT = groups of 4 waves per block, each taking every SPLIT-th train chunk for the block's queries (round 5, one-pair calls: 16 blocks
// walk 63 chunks as a chain of barrier-separated steps — 32 us on 16 of 256 CUs; with 4 groups the chain is 16 steps and the groups' keys
// merge through LDS at the end.  The key is an integer maximum: the same bits whatever the split).
// TRI: the block triangulates its 128 queries against the matches it has just found (myslam_hamming_match_triangulate_batch: a recorded
// one-pair step is bound by the number of its launches, and the triangulation of a query needs nothing but that query's match).
template <int TILES, int SPLIT = 1, bool TRI = false>
__global__ __launch_bounds__(256 * SPLIT) void k_hamming_fp4(const uint8_t* __restrict__ q, const int32_t* __restrict__ nqv,
                                                     const uint8_t* __restrict__ tr, const int32_t* __restrict__ ntv,
                                                     int cap, int nq_single, int nt_single,
                                                     int32_t* __restrict__ out_idx, int32_t* __restrict__ out_dist, TriTail tt) {
    MYSLAM_SIDE_PRIO();
    __shared__ __attribute__((aligned(16))) uint8_t s_exp[SPLIT][2][32 * HF_ROWB];
    __shared__ uint32_t s_key[SPLIT > 1 ? SPLIT : 1][SPLIT > 1 ? 128 * TILES : 1];
    __shared__ int32_t s_match[TRI ? 128 * TILES : 1];
    __shared__ uint32_t s_lut[256];                // byte -> 8 FP4 codes (bit i -> nibble i): bit 1 -> -1.0 (0xA), bit 0 -> +1.0 (0x2)
    const int p = blockIdx.y;
    const int nq = nqv ? min(nqv[p], cap) : nq_single;
    const int nt = ntv ? min(ntv[p], cap) : nt_single;
    constexpr int QBLOCK = 4 * TILES * 32;
    const int q0 = blockIdx.x * QBLOCK;
    if (q0 >= nq) return;
    const int grp = SPLIT > 1 ? (int)threadIdx.x >> 8 : 0;      // train split of this wave group
    const int tid = threadIdx.x & 255, lane = tid & 63, wv = tid >> 6;
    if (grp == 0) {
        uint32_t v = 0;
        for (int i = 0; i < 8; i++) v |= (((tid >> i) & 1) ? 0xAu : 0x2u) << (4 * i);
        s_lut[tid] = v;
    }
    __syncthreads();
    const uint32_t* Q = reinterpret_cast<const uint32_t*>(q + (size_t)p * cap * 32);
    const uint32_t* T = reinterpret_cast<const uint32_t*>(tr + (size_t)p * cap * 32);
    auto expand32 = [&](uint32_t w) { return make_uint4(s_lut[w & 0xff], s_lut[(w >> 8) & 0xff], s_lut[(w >> 16) & 0xff], s_lut[w >> 24]); };
    // query operands: k block m of tile t takes descriptor dword 2 m + (lane >> 5) of query qb + 32 t + (lane & 31)
    const int qb = q0 + wv * (TILES * 32);
    hq_v8i B[TILES][4];
#pragma unroll
    for (int t = 0; t < TILES; t++) {
        const int qi = qb + 32 * t + (lane & 31);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint4 e = expand32((qi < nq) ? Q[(size_t)qi * 8 + 2 * m + (lane >> 5)] : 0u);
            B[t][m] = hq_v8i{(int)e.x, (int)e.y, (int)e.z, (int)e.w, 0, 0, 0, 0};
        }
    }
    const int rbase = 4 * (lane >> 5);                       // row of accumulator register r: (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    hq_v16f cinit, ctail;
    const int last0 = ((nt - 1) >> 5) << 5;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        cinit[r] = (float)(31 - row);
        ctail[r] = (last0 + row < nt) ? (float)(31 - row) : -16777216.f;
    }
    int bestv[TILES], bestor[TILES], bestc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; t++) { bestv[t] = -(1 << 30); bestor[t] = -(1 << 30); bestc[t] = 0; }
    const int er = tid >> 3, ed = tid & 7;                   // expansion job: dword ed of chunk row er
    auto expand = [&](int buf, uint32_t w) { *reinterpret_cast<uint4*>(&s_exp[grp][buf][er * HF_ROWB + ed * 16]) = expand32(w); };
    const int nchunk = (nt + 31) >> 5;
    const int nstep = (nchunk + SPLIT - 1) / SPLIT;          // block-uniform: every group meets every barrier
    uint32_t wnext = (32 * grp + er < nt) ? T[(size_t)(32 * grp + er) * 8 + ed] : 0u;
    if (grp < nchunk) expand(0, wnext);
    for (int i = 0; i < nstep; i++) {
        const int c = i * SPLIT + grp;                       // this group's chunk of the step (may lie past the end)
        const int t0 = c << 5;
        if (c + SPLIT < nchunk) { const int row = t0 + 32 * SPLIT + er; wnext = (row < nt) ? T[(size_t)row * 8 + ed] : 0u; }
        __syncthreads();
        if (c + SPLIT < nchunk) expand((i + 1) & 1, wnext);
        if (c >= nchunk) continue;
        const uint8_t* sb = &s_exp[grp][i & 1][(lane & 31) * HF_ROWB + (lane >> 5) * 16];
        hq_v8i A[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint4 e = *reinterpret_cast<const uint4*>(sb + m * 32);
            A[m] = hq_v8i{(int)e.x, (int)e.y, (int)e.z, (int)e.w, 0, 0, 0, 0};
        }
        const hq_v16f c0 = (c + 1 == nchunk) ? ctail : cinit;
#pragma unroll
        for (int t = 0; t < TILES; t++) {
            // A: FP4, scale 2^0 (E8M0 127); B: FP4, scale 2^5 (132)
            hq_v16f acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[0], B[t][0], c0, 4, 4, 0, 127, 0, 132);
#pragma unroll
            for (int m = 1; m < 4; m++) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[m], B[t][m], acc, 4, 4, 0, 127, 0, 132);
            float vf = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) vf = fmaxf(fmaxf(vf, acc[r]), acc[r + 1]);
            vf = fmaxf(vf, acc[15]);
            const int v = (int)vf;
            const bool better = v > bestor[t];
            bestv[t] = better ? v : bestv[t];
            bestor[t] = better ? (v | 31) : bestor[t];
            bestc[t] = better ? c : bestc[t];
        }
    }
#pragma unroll
    for (int t = 0; t < TILES; t++) {
        const int dot = bestv[t] >> 5, row = 31 - (bestv[t] & 31);
        const uint32_t key = bestv[t] < -(1 << 20) ? 0u
                                                     : ((uint32_t)(dot + 256) << 20) | (0xfffffu - (uint32_t)(bestc[t] * 32 + row));
        uint32_t b = max(key, (uint32_t)__shfl_xor((int)key, 32, 64));
        if (SPLIT > 1) {                                     // the groups' keys of this query: maximum through LDS, written by group 0
            if (lane < 32) s_key[grp][(wv * TILES + t) * 32 + lane] = b;
            __syncthreads();
            if (grp == 0 && lane < 32)
#pragma unroll
                for (int g2 = 1; g2 < SPLIT; g2++) b = max(b, s_key[g2][(wv * TILES + t) * 32 + lane]);
        }
        const int qi = qb + 32 * t + lane;
        if (grp == 0 && lane < 32 && qi < nq) {
            const bool any = nt > 0;
            out_idx[(size_t)p * cap + qi] = any ? (int32_t)(0xfffffu - (b & 0xfffffu)) : -1;
            out_dist[(size_t)p * cap + qi] = any ? (int32_t)((512u - (b >> 20)) >> 1) : -1;
            if (TRI) s_match[(wv * TILES + t) * 32 + lane] = any ? (int32_t)(0xfffffu - (b & 0xfffffu)) : -1;
        }
    }
    if (TRI) {                                               // whole waves: thread i of the block takes query q0 + i
        __syncthreads();
        const int i = threadIdx.x;
        if (i < 128 * TILES && q0 + i < nq) tri_point(tt, p, cap, q0 + i, s_match[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// one-sided Jacobi SVD of the 4x4 DLT matrix (f64), fully unrolled pair loop (no runtime register indexing)
__device__ __forceinline__ void tri_solve(const double (&P)[2][12], const double (&pt)[2][2], double* xyz, double& ratio) {
    double A[4][4], V[4][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            A[2 * i][c] = pt[i][0] * P[i][8 + c] - P[i][c];            // algorithm.h:23
            A[2 * i + 1][c] = pt[i][1] * P[i][8 + c] - P[i][4 + c];    // algorithm.h:24
        }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
    // Per column pair: one sqrt, one division and one reciprocal square root (f64 sqrt / division cost ~25 instructions each on this
    // hardware; the textbook form zeta -> t -> c needs two of each, plus two more for a normalised convergence measure).  The rotation
    // angle is the same: tan(theta) = 2 gamma / (d + sign(d) hypot(d, 2 gamma)) with d = beta - alpha; convergence is tested on squares.
    for (int sweep = 0; sweep < 60; sweep++) {
        double off2 = 0;                                     // largest gamma^2 / (alpha beta) of the sweep, kept as a pair of products
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
                const double ab = alpha * beta, g2 = gamma * gamma;
                // |gamma| > 1e-30 + 1e-17 sqrt(alpha beta)  (the pair is numerically orthogonal otherwise), on squares
                if (g2 > 1e-60 && g2 > 1e-34 * ab) {
                    rotated = true;
                    if (g2 > off2 * ab) off2 = g2 / (ab + 1e-300);
                    const double d = beta - alpha, tg = 2.0 * gamma;
                    const double hyp = sqrt(d * d + tg * tg);
                    const double t = tg / (d + (d >= 0 ? hyp : -hyp));
                    const double c = rsqrt(1.0 + t * t), s = c * t;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const double ap = A[i][p], aq = A[i][q];
                        A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
                        const double vp = V[i][p], vq = V[i][q];
                        V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
                    }
                }
            }
        if (!rotated || off2 < 1e-30) break;                 // off = sqrt(off2) < 1e-15
    }
    double sv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) sv[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j] + A[3][j] * A[3][j]);
    // smallest and second smallest singular value; V column of the smallest
    double s_min = sv[0], v0 = V[0][0], v1 = V[1][0], v2 = V[2][0], v3 = V[3][0];
#pragma unroll
    for (int j = 1; j < 4; j++)
        if (sv[j] < s_min) { s_min = sv[j]; v0 = V[0][j]; v1 = V[1][j]; v2 = V[2][j]; v3 = V[3][j]; }
    double s_2nd = 1e300;
    bool skipped = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!skipped && sv[j] == s_min) { skipped = true; continue; }
        s_2nd = fmin(s_2nd, sv[j]);
    }
    xyz[0] = v0 / v3; xyz[1] = v1 / v3; xyz[2] = v2 / v3;              // algorithm.h:27
    ratio = s_min / s_2nd;                                             // algorithm.h:29
}

__device__ __forceinline__ void tri_point(const TriTail& tt, int p, int cap, int i, int j) {
    const size_t o = (size_t)p * cap + i;
    if (j < 0 || j >= cap) { tt.ok[o] = 0;       // no match (or an index outside the pair's slots: treated as none)
        tt.xyz[3 * o] = tt.xyz[3 * o + 1] = tt.xyz[3 * o + 2] = 0; return; }
    const double ul = tt.kl[o].x, vl = tt.kl[o].y;
    const myslam_keypoint r = tt.kr[(size_t)p * cap + j];
    const double ur = r.x, vr = r.y;
    const double P[2][12] = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0},
                             {1, 0, 0, -tt.baseline, 0, 1, 0, 0, 0, 0, 1, 0}};      // system.cpp:108-116,141-145
    const double pt[2][2] = {{(ul - tt.cx) / tt.fx, (vl - tt.cy) / tt.fy}, {(ur - tt.cx) / tt.fx, (vr - tt.cy) / tt.fy}};   // camera.cpp:22-26
    double X[3], ratio;
    tri_solve(P, pt, X, ratio);
    tt.xyz[3 * o] = X[0]; tt.xyz[3 * o + 1] = X[1]; tt.xyz[3 * o + 2] = X[2];
    tt.ok[o] = (ratio < 1e-2 && X[2] > 0) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_triangulate(const float* __restrict__ xl, const float* __restrict__ yl,
                                                     const float* __restrict__ xr, const float* __restrict__ yr,
                                                     const myslam_keypoint* __restrict__ kl, const myslam_keypoint* __restrict__ kr,
                                                     const int32_t* __restrict__ match, const int32_t* __restrict__ nlv,
                                                     int cap, int n_single, double fx, double fy, double cx, double cy,
                                                     double baseline, double* __restrict__ xyz, uint8_t* __restrict__ ok) {
    MYSLAM_SIDE_PRIO();
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = nlv ? min(nlv[p], cap) : n_single;
    if (i >= n) return;
    const size_t o = (size_t)p * cap + i;
    if (kl) { tri_point(TriTail{kl, kr, fx, fy, cx, cy, baseline, xyz, ok}, p, cap, i, match[o]); return; }      // (the same code as the one-pair matcher's tail)
    const double ul = xl[i], vl = yl[i], ur = xr[i], vr = yr[i];
    const double P[2][12] = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0},
                             {1, 0, 0, -baseline, 0, 1, 0, 0, 0, 0, 1, 0}};      // system.cpp:108-116,141-145
    const double pt[2][2] = {{(ul - cx) / fx, (vl - cy) / fy}, {(ur - cx) / fx, (vr - cy) / fy}};   // camera.cpp:22-26
    double X[3], ratio;
    tri_solve(P, pt, X, ratio);
    xyz[3 * o] = X[0]; xyz[3 * o + 1] = X[1]; xyz[3 * o + 2] = X[2];
    ok[o] = (ratio < 1e-2 && X[2] > 0) ? 1 : 0;
}

}  // namespace myslam_hip

using namespace myslam_hip;

extern "C" {

int myslam_hamming_match_batch(const uint8_t* d_q, const int32_t* d_nq, const uint8_t* d_t, const int32_t* d_nt, int batch,
                               int cap, int32_t* d_train_idx, int32_t* d_dist, void* hip_stream) {
    if (!d_q || !d_t || !d_nq || !d_nt || batch <= 0 || cap <= 0 || !d_train_idx || !d_dist) return MYSLAM_ERR_INVALID;
    if (cap >= (1 << 20)) return MYSLAM_ERR_UNSUPPORTED;          // the running minimum packs (distance, train index) into 32 bits
    hipStream_t s = (hipStream_t)hip_stream;
    ScopedProf sp(P_MATCH, s);
    if (batch < HQ_NARROW_BELOW)
        hipLaunchKernelGGL((k_hamming_fp4<1, HQ_SPLIT>), dim3((cap + 127) / 128, batch), dim3(256 * HQ_SPLIT), 0, s, d_q, d_nq, d_t, d_nt, cap, 0, 0, d_train_idx, d_dist, TriTail{});
    else
        hipLaunchKernelGGL(k_hamming_fp4<HQ_TILES>, dim3((cap + HQ_BLOCK - 1) / HQ_BLOCK, batch), dim3(256), 0, s, d_q, d_nq, d_t, d_nt, cap, 0, 0,
                           d_train_idx, d_dist, TriTail{});
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

int myslam_hamming_match(const uint8_t* query, int nq, const uint8_t* train, int nt, int32_t* train_idx, int32_t* dist) {
    if (nq < 0 || nt < 0) return MYSLAM_ERR_INVALID;
    if (nq == 0) return MYSLAM_OK;
    if (!query || !train_idx || !dist || (nt > 0 && !train)) return MYSLAM_ERR_INVALID;
    if (nt >= (1 << 20)) return MYSLAM_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    const int cap = std::max(nq, std::max(nt, 1));
    HostCall hc;                                               // thread-local staging: no allocation, one copy each way
    const int pq = hc.in(query, (size_t)nq * 32), pt = hc.in(train, (size_t)nt * 32);
    const int pi = hc.out(train_idx, (size_t)nq), pd = hc.out(dist, (size_t)nq);
    int rc = hc.upload();
    if (rc) return rc;
    const uint8_t* dq = hc.dev<uint8_t>(pq); const uint8_t* dt = hc.dev<uint8_t>(pt);
    int32_t* di = hc.dev<int32_t>(pi); int32_t* dd = hc.dev<int32_t>(pd);
    {
        ScopedProf sp(P_MATCH, hc.stream());
        hipLaunchKernelGGL((k_hamming_fp4<1, HQ_SPLIT>), dim3((nq + 127) / 128, 1), dim3(256 * HQ_SPLIT), 0, hc.stream(), dq, (const int32_t*)nullptr, dt,
                           (const int32_t*)nullptr, cap, nq, nt, di, dd, TriTail{});
    }
    MYSLAM_HIP_CHECK(hipGetLastError());
    return hc.download();
}

// src/loopclosing.cpp:175-186 — host bookkeeping (a handful of compares)
int myslam_hamming_filter(const int32_t* dist, int n, uint8_t* keep, int* min_dist) {
    if (n < 0 || (n > 0 && (!dist || !keep))) return MYSLAM_ERR_INVALID;
    if (n == 0) { if (min_dist) *min_dist = 0; return MYSLAM_OK; }
    int mn = dist[0];
    for (int i = 1; i < n; i++) mn = std::min(mn, dist[i]);
    const double lim = std::max(2.0 * (double)mn, 30.0);
    for (int i = 0; i < n; i++) keep[i] = ((double)dist[i] <= lim) ? 1 : 0;
    if (min_dist) *min_dist = mn;
    return MYSLAM_OK;
}

// LoopClosing::ProcessNewKF, src/loopclosing.cpp:94-105: every single-layer feature of the key-frame becomes nlevels pyramid key-points
// (octave = level, response = -1, class_id = index of the feature) — the input of ScreenAndComputeKPsParams.  Host bookkeeping.
int myslam_expand_pyramid_keypoints(const myslam_keypoint* feats, int n, int nlevels, myslam_keypoint* out /* n * nlevels */) {
    if (n < 0 || nlevels < 1 || (n > 0 && (!feats || !out))) return MYSLAM_ERR_INVALID;
    for (int i = 0; i < n; i++)
        for (int l = 0; l < nlevels; l++) {
            myslam_keypoint kp = feats[i];
            kp.octave = l; kp.response = -1.f; kp.class_id = i;
            out[(size_t)i * nlevels + l] = kp;
        }
    return MYSLAM_OK;
}

// LoopClosing::MatchFeatures, src/loopclosing.cpp:172-203, after the matcher: keep matches with distance <= max(2 * min_dist, 30),
// map both sides to their FEATURE ids through class_id and insert (current, loop) into a std::set — one entry per feature pair, in the
// set's order (ascending current id, then loop id), which is the order ComputeCorrectPose walks (:215-238).  query = loop key-frame's
// pyramid key-points, train = current key-frame's.  Returns MYSLAM_OK; the caller applies the `< 10 matches` rule (:196).
int myslam_match_feature_pairs(const int32_t* train_idx, const int32_t* dist, int n_query, const myslam_keypoint* loop_pyr_kps,
                               const myslam_keypoint* cur_pyr_kps, int n_train, int32_t* pairs /* n_query x 2: (current, loop) */, int* n_pairs) {
    if (n_query < 0 || !n_pairs || (n_query > 0 && (!train_idx || !dist || !loop_pyr_kps || !cur_pyr_kps || !pairs))) return MYSLAM_ERR_INVALID;
    *n_pairs = 0;
    if (n_query == 0) return MYSLAM_OK;
    int mn = dist[0];
    for (int i = 1; i < n_query; i++) mn = std::min(mn, dist[i]);
    const double lim = std::max(2.0 * (double)mn, 30.0);
    std::vector<std::pair<int, int>> v;
    for (int i = 0; i < n_query; i++) {
        if (train_idx[i] < 0 || train_idx[i] >= n_train) return MYSLAM_ERR_INVALID;
        if ((double)dist[i] <= lim) v.emplace_back(cur_pyr_kps[train_idx[i]].class_id, loop_pyr_kps[i].class_id);
    }
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
    for (size_t k = 0; k < v.size(); k++) { pairs[2 * k] = v[k].first; pairs[2 * k + 1] = v[k].second; }
    *n_pairs = (int)v.size();
    return MYSLAM_OK;
}

int myslam_triangulate_stereo_batch(const myslam_keypoint* d_kps_l, const myslam_keypoint* d_kps_r, const int32_t* d_match,
                                    const int32_t* d_nl, int batch, int cap, double fx, double fy, double cx, double cy,
                                    double baseline, double* d_xyz, uint8_t* d_ok, void* hip_stream) {
    if (!d_kps_l || !d_kps_r || !d_match || !d_nl || batch <= 0 || cap <= 0 || !d_xyz || !d_ok) return MYSLAM_ERR_INVALID;
    hipStream_t s = (hipStream_t)hip_stream;
    ScopedProf sp(P_TRI, s);
    hipLaunchKernelGGL(k_triangulate, dim3((cap + 255) / 256, batch), dim3(256), 0, s, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, d_kps_l, d_kps_r, d_match, d_nl, cap, 0, fx, fy, cx, cy,
                       baseline, d_xyz, d_ok);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

// BFMatcher.match + triangulation() of every match in one call: what the two calls above / below do, in ONE launch for a handful of pairs
int myslam_hamming_match_triangulate_batch(const uint8_t* d_q, const int32_t* d_nq, const uint8_t* d_t, const int32_t* d_nt,
                                           const myslam_keypoint* d_kps_l, const myslam_keypoint* d_kps_r, int batch, int cap,
                                           double fx, double fy, double cx, double cy, double baseline,
                                           int32_t* d_train_idx, int32_t* d_dist, double* d_xyz, uint8_t* d_ok, void* hip_stream) {
    if (!d_kps_l || !d_kps_r || !d_xyz || !d_ok) return MYSLAM_ERR_INVALID;
    if (batch >= HQ_NARROW_BELOW) {
        const int rc = myslam_hamming_match_batch(d_q, d_nq, d_t, d_nt, batch, cap, d_train_idx, d_dist, hip_stream);
        if (rc) return rc;
        return myslam_triangulate_stereo_batch(d_kps_l, d_kps_r, d_train_idx, d_nq, batch, cap, fx, fy, cx, cy, baseline, d_xyz, d_ok, hip_stream);
    }
    if (!d_q || !d_t || !d_nq || !d_nt || batch <= 0 || cap <= 0 || !d_train_idx || !d_dist) return MYSLAM_ERR_INVALID;
    if (cap >= (1 << 20)) return MYSLAM_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)hip_stream;
    ScopedProf sp(P_MATCH, s);
    hipLaunchKernelGGL((k_hamming_fp4<1, HQ_SPLIT, true>), dim3((cap + 127) / 128, batch), dim3(256 * HQ_SPLIT), 0, s, d_q, d_nq, d_t, d_nt, cap, 0, 0,
                       d_train_idx, d_dist, TriTail{d_kps_l, d_kps_r, fx, fy, cx, cy, baseline, d_xyz, d_ok});
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

int myslam_triangulate_stereo(const float* xl, const float* yl, const float* xr, const float* yr, int n, double fx, double fy,
                              double cx, double cy, double baseline, double* xyz, uint8_t* ok) {
    if (n < 0) return MYSLAM_ERR_INVALID;
    if (n == 0) return MYSLAM_OK;
    if (!xl || !yl || !xr || !yr || !xyz || !ok) return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    HostCall hc;
    const int p0 = hc.in(xl, (size_t)n), p1 = hc.in(yl, (size_t)n), p2 = hc.in(xr, (size_t)n), p3 = hc.in(yr, (size_t)n);
    const int px = hc.out(xyz, (size_t)n * 3), po = hc.out(ok, (size_t)n);
    int rc = hc.upload();
    if (rc) return rc;
    {
        ScopedProf sp(P_TRI, hc.stream());
        hipLaunchKernelGGL(k_triangulate, dim3((n + 255) / 256, 1), dim3(256), 0, hc.stream(), hc.dev<float>(p0), hc.dev<float>(p1),
                           hc.dev<float>(p2), hc.dev<float>(p3), (const myslam_keypoint*)nullptr, (const myslam_keypoint*)nullptr,
                           (const int32_t*)nullptr, (const int32_t*)nullptr, n, n, fx, fy, cx, cy, baseline, hc.dev<double>(px), hc.dev<uint8_t>(po));
    }
    MYSLAM_HIP_CHECK(hipGetLastError());
    return hc.download();
}

}  // extern "C"
