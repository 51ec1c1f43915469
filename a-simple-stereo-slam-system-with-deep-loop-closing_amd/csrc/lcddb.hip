// lcddb.hip — key-frame descriptor database and cosine scan on gfx950.
// Replaces LoopClosing::_mvDatabase + DetectLoop()/AddToDatabase() (reference include/myslam/loopclosing.h:67,120;
// src/loopclosing.cpp:124-161, 651-659): ascending-id scan, stop at the first KF with cur_id - id < 20,
// maxScore (init 0, strict '>': lowest id wins ties), cnt = #{score > thr_low}.
//
// HBM layout: row-major f32 [capacity][1064] (4256-byte rows), ids kept on the host (ascending).
// Scan kernel: one wave per database row, the row lives in registers (17 floats per lane) and is dotted
// against every query of the batch, so a batch of queries streams the database from HBM exactly once.
#include <algorithm>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"

namespace myslam_hip {

constexpr int DIM = MYSLAM_LCD_DIM;     // 1064 = 16*64 + 40
constexpr int DB_WAVES = 4;             // waves per block
constexpr int DB_ROWS_PER_WAVE = 8;

struct Partial { float score; int32_t idx; int32_t cnt; };

// nstart (may be null = 0 for every query): first row a query looks at — a query's rows are [nstart, nvalid) (round 6: the scan BEHIND the cut-off window of a
// shard that owns interleaved ids, myslam_lcddb_query_batch_owned)
__global__ __launch_bounds__(256) void k_db_scan(const float* __restrict__ db, const float* __restrict__ q, int nq,
                                                 const int32_t* __restrict__ nvalid, const int32_t* __restrict__ nstart, float thr_low,
                                                 Partial* __restrict__ partials /*[nblocks][nq]*/) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_db[];
    Partial* s_p = reinterpret_cast<Partial*>(smem_db);          // [DB_WAVES][nq]
    int* s_lim = reinterpret_cast<int*>(s_p + DB_WAVES * nq);    // [nq]: the row limits, read ONCE (a recorded step reads them from pinned host memory)
    int* s_beg = s_lim + nq;                                     // [nq]: first rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int qi = threadIdx.x; qi < nq; qi += 256) { s_lim[qi] = nvalid[qi]; s_beg[qi] = nstart ? nstart[qi] : 0; }
    __syncthreads();
    {   // a launch covers the allocation (recorded steps outlive appends): blocks behind every query's row limit leave an empty partial
        int lim = 0;
        for (int qi = 0; qi < nq; qi++) lim = max(lim, s_lim[qi]);
        if ((int)blockIdx.x * DB_WAVES * DB_ROWS_PER_WAVE >= lim) {
            for (int qi = threadIdx.x; qi < nq; qi += 256) partials[(size_t)blockIdx.x * nq + qi] = {0.f, -1, 0};
            return;
        }
    }
    for (int i = threadIdx.x; i < DB_WAVES * nq; i += 256) s_p[i] = {0.f, -1, 0};
    __syncthreads();
    const int row0 = (blockIdx.x * DB_WAVES + wave) * DB_ROWS_PER_WAVE;
    for (int r = row0; r < row0 + DB_ROWS_PER_WAVE; r++) {
        const float* row = db + (size_t)r * DIM;
        float v[17];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = row[k * 64 + lane];
        v[16] = (lane < DIM - 1024) ? row[1024 + lane] : 0.f;
        for (int qi = 0; qi < nq; qi++) {
            if (r >= s_lim[qi] || r < s_beg[qi]) continue;      // cut-off rule, loopclosing.cpp:133 (uniform per wave)
            const float* qq = q + (size_t)qi * DIM;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 16; k++) acc += v[k] * qq[k * 64 + lane];
            if (lane < DIM - 1024) acc += v[16] * qq[1024 + lane];
            acc = wave_reduce_sum(acc);
            if (lane == 0) {
                Partial& p = s_p[wave * nq + qi];               // rows ascend within a wave: strict '>' keeps the lowest
                if (acc > p.score) { p.score = acc; p.idx = r; }
                if (acc > thr_low) p.cnt++;
            }
        }
    }
    __syncthreads();
    for (int qi = threadIdx.x; qi < nq; qi += 256) {
        Partial best = s_p[qi];
        for (int w = 1; w < DB_WAVES; w++) {                    // waves own ascending row ranges
            const Partial p = s_p[w * nq + qi];
            if (p.score > best.score) { best.score = p.score; best.idx = p.idx; }
            best.cnt += p.cnt;
        }
        partials[(size_t)blockIdx.x * nq + qi] = best;
    }
}

// ---- batched scan as a matrix-core GEMM: scores[rows x queries] = DB[rows x 1064] * Q^T, reduced per query in the
// epilogue.  Block tile 128 rows x 128 queries, 4 waves as 2x2 of 64x64, BK = 16, register-staged LDS double buffering.
// The database streams from HBM once per 128 queries.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GM = 128, GN = 128, GK = 16;

__device__ __forceinline__ bool better(float s, int row, float bs, int brow) {
    return s > bs || (s == bs && brow >= 0 && row < brow);       // strict '>' scan: on equal scores the lower row came first
}

// The GEMM runs on the bf16 matrix cores with f32 accuracy (see k_conv2_bf16x6 in calc.hip): an f32-input MFMA runs at the f32
// vector rate and competes with the VALU-bound ORB kernels of the other stream (0.28 against 0.15 ms per 512 queries x 10 k rows).  Database rows and queries are split exactly into
// three bf16 pieces while they are staged into LDS; hh + hm + mh + hl + lh + mm are accumulated in f32.
typedef __bf16 db_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void db_split3(float a0, float a1, uint32_t& h, uint32_t& m, uint32_t& l) {
    const __bf16 h0 = (__bf16)a0, h1 = (__bf16)a1;
    const float r0 = a0 - (float)h0, r1 = a1 - (float)h1;
    const __bf16 m0 = (__bf16)r0, m1 = (__bf16)r1;
    const float q0 = r0 - (float)m0, q1 = r1 - (float)m1;
    const __bf16 l0 = (__bf16)q0, l1 = (__bf16)q1;
    h = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
    m = (uint32_t)__builtin_bit_cast(unsigned short, m0) | ((uint32_t)__builtin_bit_cast(unsigned short, m1) << 16);
    l = (uint32_t)__builtin_bit_cast(unsigned short, l0) | ((uint32_t)__builtin_bit_cast(unsigned short, l1) << 16);
}

__global__ __launch_bounds__(256) void k_db_scan_bf16x6(const float* __restrict__ db, int rows_alloc, const float* __restrict__ q,
                                                        int nq, const int32_t* __restrict__ nvalid, const int32_t* __restrict__ nstart, float thr_low,
                                                        Partial* __restrict__ partials) {
    MYSLAM_SIDE_PRIO();
    __shared__ uint4 s_a[2][3][GM * 2];                // [piece][row][k half] x 8 bf16
    __shared__ uint4 s_b[2][3][GN * 2];
    __shared__ Partial s_p[2][GN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = blockIdx.x * GM, n0 = blockIdx.y * GN;
    {   // blocks behind the row limits of all of their queries leave empty partials (see k_db_scan)
        const int n = n0 + (t & (GN - 1));
        const bool live = n < nq && nvalid[n] > m0 && (!nstart || nstart[n] < m0 + GM);
        if (__syncthreads_or(live ? 1 : 0) == 0) {
            if (t < GN && n0 + t < nq) partials[(size_t)blockIdx.x * nq + n0 + t] = {0.f, -1, 0};
            return;
        }
    }
    const int row = t >> 1, kk = (t & 1) * 8;
    const bool a_ok = (m0 + row) < rows_alloc, b_ok = (n0 + row) < nq;
    // unconditional loads from clamped rows + a select at store time keep the prefetch in registers
    const float* arow = db + (size_t)(a_ok ? m0 + row : 0) * DIM + kk;
    const float* brow = q + (size_t)(b_ok ? n0 + row : 0) * DIM + kk;
    float4 ra0, ra1, rb0, rb1;
    bool kv = true;
#define DB_LOAD_STAGE(ST)                                                                  \
    {                                                                                      \
        const int k0_ = (ST) * GK;                                                         \
        kv = (k0_ + kk) < DIM;                                                             \
        const int ko_ = kv ? k0_ : 0;                                                      \
        const float4* pa_ = reinterpret_cast<const float4*>(arow + ko_); ra0 = pa_[0]; ra1 = pa_[1]; \
        const float4* pb_ = reinterpret_cast<const float4*>(brow + ko_); rb0 = pb_[0]; rb1 = pb_[1]; \
    }
    auto store_stage = [&](int buf) {
        const float za = (a_ok && kv) ? 1.f : 0.f, zb = (b_ok && kv) ? 1.f : 0.f;
        uint4 h, m, l;
        db_split3(ra0.x * za, ra0.y * za, h.x, m.x, l.x); db_split3(ra0.z * za, ra0.w * za, h.y, m.y, l.y);
        db_split3(ra1.x * za, ra1.y * za, h.z, m.z, l.z); db_split3(ra1.z * za, ra1.w * za, h.w, m.w, l.w);
        s_a[buf][0][t] = h; s_a[buf][1][t] = m; s_a[buf][2][t] = l;
        db_split3(rb0.x * zb, rb0.y * zb, h.x, m.x, l.x); db_split3(rb0.z * zb, rb0.w * zb, h.y, m.y, l.y);
        db_split3(rb1.x * zb, rb1.y * zb, h.z, m.z, l.z); db_split3(rb1.z * zb, rb1.w * zb, h.w, m.w, l.w);
        s_b[buf][0][t] = h; s_b[buf][1][t] = m; s_b[buf][2][t] = l;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    constexpr int NST = (DIM + GK - 1) / GK;                     // 67
    DB_LOAD_STAGE(0)
    store_stage(0);
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    for (int st = 0; st < NST; st++) {
        const int buf = st & 1;
        if (st + 1 < NST) DB_LOAD_STAGE(st + 1)
        db_bf16x8 A[2][3], B[2][3];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int p = 0; p < 3; p++) A[i][p] = __builtin_bit_cast(db_bf16x8, s_a[buf][p][(wm + 32 * i + lr) * 2 + lk]);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int p = 0; p < 3; p++) B[j][p] = __builtin_bit_cast(db_bf16x8, s_b[buf][p][(wn + 32 * j + lr) * 2 + lk]);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][2], B[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][0], c, 0, 0, 0);
                acc[i][j] = c;
            }
        if (st + 1 < NST) store_stage(buf ^ 1);
        __syncthreads();
    }
#undef DB_LOAD_STAGE
    // epilogue: identical to k_db_scan_mfma
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int n = n0 + wn + j * 32 + lr;
        const int nv = (n < nq) ? nvalid[n] : 0;
        const int ns = (nstart && n < nq) ? nstart[n] : 0;
        float bs = 0.f; int bi = -1, cnt = 0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float sc = acc[i][j][r];
                if (m < nv && m >= ns) {
                    if (better(sc, m, bs, bi)) { bs = sc; bi = m; }
                    cnt += (sc > thr_low);
                }
            }
        const float os = __shfl_xor(bs, 32, 64); const int oi = __shfl_xor(bi, 32, 64); const int oc = __shfl_xor(cnt, 32, 64);
        if (oi >= 0 && better(os, oi, bs, bi)) { bs = os; bi = oi; }
        cnt += oc;
        if (lk == 0 && (wave >> 1) == 1) s_p[0][wn + j * 32 + lr] = {bs, bi, cnt};
        __syncthreads();
        if (lk == 0 && (wave >> 1) == 0) {
            const Partial o = s_p[0][wn + j * 32 + lr];
            if (o.idx >= 0 && better(o.score, o.idx, bs, bi)) { bs = o.score; bi = o.idx; }
            cnt += o.cnt;
            if (n < nq) partials[(size_t)blockIdx.x * nq + n] = {bs, bi, cnt};
        }
        __syncthreads();
    }
}

// One WAVE per query (round 5: a thread per query walked the block partials one dependent load after the other — 29 us for the 313
// partials of a single query against 10 000 rows, a fifth of a live stream's per-frame chain).  Lanes take partials lane, lane + 64, ...
// in ascending order (strict '>' keeps the lowest block), then the 64 candidates are merged: larger score wins, equal scores -> lower row
// (blocks own ascending row ranges), exactly what the ascending strict-'>' walk returns.
__global__ __launch_bounds__(256) void k_db_reduce(const Partial* __restrict__ partials, int nblocks, int nq,
                                                   const uint64_t* __restrict__ ids, uint64_t* __restrict__ best_id,
                                                   float* __restrict__ max_score, int32_t* __restrict__ cnt) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (qi >= nq) return;
    float ms = 0.f; int bi = -1; int c = 0;
    for (int b = lane; b < nblocks; b += 64) {
        const Partial p = partials[(size_t)b * nq + qi];
        if (p.score > ms) { ms = p.score; bi = p.idx; }
        c += p.cnt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float os = __shfl_xor(ms, o, 64); const int oi = __shfl_xor(bi, o, 64);
        c += __shfl_xor(c, o, 64);
        if (os > ms || (os == ms && oi >= 0 && (bi < 0 || oi < bi))) { ms = os; bi = oi; }
    }
    if (lane == 0) {
        best_id[qi] = (bi >= 0) ? ids[bi] : 0;                  // bestId initialised to 0, loopclosing.cpp:129
        max_score[qi] = ms; cnt[qi] = c;
    }
}

// per-shard result of a query in the layout that travels between ranks: cnt bit 31 = this shard's scan hit the break (:133)
__global__ __launch_bounds__(256) void k_db_pack_candidates(const uint64_t* __restrict__ best_id, const float* __restrict__ max_score,
                                                            const int32_t* __restrict__ cnt, const int32_t* __restrict__ nvalid,
                                                            const int32_t* __restrict__ nrows_p, int nq, myslam_lcd_candidate* __restrict__ out) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    const int nrows = *nrows_p;                                   // rows the shard held when the limits were written (device memory: replays see appends)
    myslam_lcd_candidate c;
    c.best_id = best_id[qi]; c.max_score = max_score[qi];
    c.cnt = (cnt[qi] & 0x7fffffff) | (nvalid[qi] < nrows ? (int32_t)0x80000000 : 0);
    out[qi] = c;
}

// The reference's ONE ascending scan (loopclosing.cpp:124-161) over shards that own ascending id ranges: strict '>' keeps the
// first (= lowest-id) maximum, counts add up, and the first shard whose own scan hit the break ends the whole scan.
__host__ __device__ inline void merge_one(const myslam_lcd_candidate* g, int nshards, int nq, int qi, uint64_t& best, float& ms, int32_t& cnt) {
    best = 0; ms = 0.f; cnt = 0;
    for (int s = 0; s < nshards; s++) {
        const myslam_lcd_candidate c = g[(size_t)s * nq + qi];
        if (c.max_score > ms) { ms = c.max_score; best = c.best_id; }
        cnt += c.cnt & 0x7fffffff;
        if (c.cnt < 0) break;
    }
}

__global__ __launch_bounds__(256) void k_db_merge_candidates(const myslam_lcd_candidate* __restrict__ g, int nshards, int nq,
                                                             uint64_t* __restrict__ best_id, float* __restrict__ max_score,
                                                             int32_t* __restrict__ cnt) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint64_t b; float m; int32_t c;
    merge_one(g, nshards, nq, qi, b, m, c);
    best_id[qi] = b; max_score[qi] = m; cnt[qi] = c;
}

// ---- shards that own INTERLEAVED ids (round 6: a sharded database that grows) -----------------------------------------------------------------------
// With contiguous id ranges every new key-frame lands on the last rank.  When ownership is by arrival order (key-frame k of the job goes to rank k mod N,
// any rule that puts every id into exactly one shard will do) a shard's ids interleave with the others', and "the first shard that broke ends the scan"
// no longer describes the reference's ONE ascending scan (loopclosing.cpp:124-161).  What does: the scan looks at every id BELOW the cut-off window
// W = {id : cur - id < 20 (mod 2^64)}, stops if the map holds any id inside W, and otherwise goes on with every id above cur.  So a shard reports both
// parts and whether it holds an id in W; the merge adds the parts above cur only when NO shard holds one.  Ties: strict '>' in ascending order keeps the
// LOWEST id among equal scores — compared explicitly, since shard order no longer is id order.
__global__ __launch_bounds__(256) void k_db_pack_owned(const uint64_t* __restrict__ bestP, const float* __restrict__ maxP, const int32_t* __restrict__ cntP,
                                                       const uint64_t* __restrict__ bestS, const float* __restrict__ maxS, const int32_t* __restrict__ cntS,
                                                       const int32_t* __restrict__ broke, int has_suffix, int nq, myslam_lcd_owned_candidate* __restrict__ out) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    myslam_lcd_owned_candidate c;
    c.pre_best_id = bestP[qi]; c.pre_max_score = maxP[qi];
    c.pre_cnt = (cntP[qi] & 0x7fffffff) | (broke[qi] ? (int32_t)0x80000000 : 0);
    c.suf_best_id = has_suffix ? bestS[qi] : 0; c.suf_max_score = has_suffix ? maxS[qi] : 0.f; c.suf_cnt = has_suffix ? cntS[qi] : 0;
    out[qi] = c;
}

__host__ __device__ inline void merge_owned_one(const myslam_lcd_owned_candidate* g, int nshards, int nq, int qi, uint64_t& best, float& ms, int32_t& cnt) {
    best = 0; ms = 0.f; cnt = 0;
    bool broke = false;
    for (int s = 0; s < nshards; s++) {                            // everything below the window: highest score, lowest id among equals
        const myslam_lcd_owned_candidate c = g[(size_t)s * nq + qi];
        if (c.pre_max_score > ms || (c.pre_max_score == ms && ms > 0.f && c.pre_best_id < best)) { ms = c.pre_max_score; best = c.pre_best_id; }
        cnt += c.pre_cnt & 0x7fffffff;
        broke = broke || c.pre_cnt < 0;
    }
    if (broke) return;                                             // the map holds an id with cur - id < 20: the scan ended there (:133)
    float ss = 0.f; uint64_t sb = 0;
    for (int s = 0; s < nshards; s++) {                            // the ids above cur, scanned after all of the above
        const myslam_lcd_owned_candidate c = g[(size_t)s * nq + qi];
        if (c.suf_max_score > ss || (c.suf_max_score == ss && ss > 0.f && c.suf_best_id < sb)) { ss = c.suf_max_score; sb = c.suf_best_id; }
        cnt += c.suf_cnt;
    }
    if (ss > ms) { ms = ss; best = sb; }                           // strict: an equal score further up the map does not replace the earlier one
}

__global__ __launch_bounds__(256) void k_db_merge_owned(const myslam_lcd_owned_candidate* __restrict__ g, int nshards, int nq,
                                                        uint64_t* __restrict__ best_id, float* __restrict__ max_score, int32_t* __restrict__ cnt) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint64_t b; float m; int32_t c;
    merge_owned_one(g, nshards, nq, qi, b, m, c);
    best_id[qi] = b; max_score[qi] = m; cnt[qi] = c;
}

}  // namespace myslam_hip

using namespace myslam_hip;

// ---- storage and query contexts ------------------------------------------------------------------------------------------------
// LoopClosing::_mvDatabase is ONE std::map that every key-frame of the process goes into (loopclosing.h:120, loopclosing.cpp:651-659).
// The device form is split accordingly (round 5): `myslam_lcddb` owns the descriptor matrix and the ids — appends go there — and a
// `myslam_lcddb_query_ctx` owns what ONE stream of queries needs (row-limit staging, partial results, shard scratch, recorded-step
// state).  L concurrent streams (L cameras of one GPU, or L lanes of a bench) scan ONE matrix through L contexts; the handle itself
// carries a built-in context on its own stream, which is what the myslam_lcddb_query* entry points use.
//
// Ordering rules:
//   * appends copy on the storage's stream and return when the rows are in HBM; a scan only reads rows below its own row limits
//     (fixed on the host at call time), so appends never race with scans in flight on other streams;
//   * growth (db_reserve) moves the matrix: it waits for every context's stream AND for the replays of every recorded step that
//     captured a scan (DbGraphLink::wait), bumps the generation, and parks the old matrix until no recorded step can read it (a
//     stale step's launch fails with MYSLAM_ERR_CAPACITY instead of reading freed memory);
//   * a recorded scan covers the whole ALLOCATION (blocks behind the row limits return at once), so appends inside the capacity
//     need no re-recording — only myslam_lcddb_ctx_update_query_limits before the next replay.
struct myslam_lcddb_query_ctx {
    myslam_lcddb* db = nullptr;
    hipStream_t stream = nullptr;
    bool builtin = false;
    Partial* d_partials = nullptr; size_t partialsCap = 0;
    int32_t* d_nvalid = nullptr; int nvalidCap = 0;                 // nq row limits + 1: the row count they were computed against
    int32_t* h_nvalid = nullptr; hipEvent_t nvEvent = nullptr;      // pinned staging of the per-query row limits + "copy done" event
    // round 6: eager uploads go through a RING of pinned slots behind slot 0 (slot 0 is what recorded scans read at their replays): a call whose limits changed — a
    // database that grows every step — used to wait for the previous call's upload to leave the one pinned buffer, i.e. for the stream to drain up to it; now it waits for
    // the upload three calls back, which has long gone
    // (16 slots since the end of round 6: with three the host could enqueue at most three steps ahead of the side stream — the step's upload sits at the END of its
    // DeepLCD chain — so any hiccup of the host thread stalled the device: the growing-database pass of a 20-step region read 6.3 or 9.5 ms per step, bimodal)
    static constexpr int NV_RING = 16;
    hipEvent_t nvRingEv[NV_RING] = {}; bool nvRingPending[NV_RING] = {}; int nvSlot = 0;
    const int32_t* limits = nullptr;                                // what the launches of the current call read: d_nvalid (eager) or h_nvalid itself (recorded)
    uint64_t* d_bestS = nullptr; float* d_maxS = nullptr; int32_t* d_cntS = nullptr; int shardCap = 0;     // scratch of the sharded query
    // scratch of the owned-shard query (round 6): row ranges [0, pLim) and [sBeg, sLim) + break flags per query (pinned staging + device copy), results of both parts
    int32_t* h_own = nullptr; int32_t* d_own = nullptr; uint64_t* d_bestO = nullptr; float* d_maxO = nullptr; int32_t* d_cntO = nullptr; int ownCap = 0;
    hipEvent_t ownEvent = nullptr; bool ownPending = false;
    int graphRows = 0, graphQueries = 0;      // > 0: a query of this context was recorded into a HIP graph covering this many rows / queries
    uint64_t graphGen = 0;                    // generation of the matrix that recording reads
    std::vector<int32_t> lastLimits, scratchLimits; int lastRows = -1; bool nvFresh = false, nvPending = false;      // what d_nvalid holds (skip identical uploads)
    std::shared_ptr<DbGraphLink> link;        // shared with the recorded steps (graph.hip): generation check at launch, replay-done events
};

struct myslam_lcddb {
    hipStream_t stream = nullptr;
    int capacity = 0, n = 0;
    float* d_db = nullptr;
    uint64_t* d_ids = nullptr;
    std::vector<uint64_t> ids;
    uint64_t generation = 0;                  // bumped whenever the matrix moves
    std::vector<std::pair<float*, uint64_t*>> retired;      // matrices a recorded step of an older generation may still name
    std::vector<myslam_lcddb_query_ctx*> ctxs;              // every live context (ctxs[0] = the built-in one)
    std::mutex mu;                            // host state: ids, n, capacity, pointers, context list
    float* d_q1 = nullptr; uint64_t* d_best1 = nullptr; float* d_max1 = nullptr; int32_t* d_cnt1 = nullptr;
    // pinned staging of the ids of asynchronous appends (a copy out of pageable host memory makes the runtime wait for the stream): a ring of slots, each with its event
    static constexpr int ID_RING = 16, ID_SLOT = 1024;
    uint64_t* h_idring = nullptr; hipEvent_t idEv[ID_RING] = {}; bool idPending[ID_RING] = {}; int idSlot = 0;

    // index of the first row the reference's scan does NOT look at: it breaks at the first id with
    // (cur - id) < 20 in unsigned arithmetic (loopclosing.cpp:133), i.e. id in [cur-19, cur] mod 2^64
    int first_in(uint64_t a, uint64_t b) const {
        auto it = std::lower_bound(ids.begin(), ids.end(), a);
        if (it != ids.end() && *it <= b) return (int)(it - ids.begin());
        return (int)ids.size();
    }
    // the three row ranges of the reference's scan for one query (ids ascending): rows [0, p) lie below the cut-off window, `broke` = row p is inside it,
    // rows [sb, se) are the ids above cur that the scan reaches when nothing is inside the window (for cur < 19 the window wraps: ids >= 2^64 - (19 - cur) end the scan too)
    void scan_ranges(uint64_t cur, int& p, bool& broke, int& sb, int& se) const {
        const uint64_t lo = cur >= 19 ? cur - 19 : 0;
        auto it = std::lower_bound(ids.begin(), ids.end(), lo);
        p = (int)(it - ids.begin());
        broke = it != ids.end() && *it <= cur;
        sb = (int)(std::upper_bound(ids.begin(), ids.end(), cur) - ids.begin());
        se = (int)ids.size();
        if (cur < 19) {
            auto hi = std::lower_bound(ids.begin(), ids.end(), UINT64_MAX - (18 - cur));
            if (hi != ids.end()) { se = (int)(hi - ids.begin()); }
        }
        if (se < sb) se = sb;
    }
    int n_valid(uint64_t cur) const {
        if (cur >= 19) return first_in(cur - 19, cur);
        return std::min(first_in(0, cur), first_in(UINT64_MAX - (18 - cur), UINT64_MAX));
    }
};

static void ctx_free(myslam_lcddb_query_ctx* c) {
    void* ptrs[] = {c->d_partials, c->d_nvalid, c->d_bestS, c->d_maxS, c->d_cntS, c->d_own, c->d_bestO, c->d_maxO, c->d_cntO};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (c->h_nvalid) (void)hipHostFree(c->h_nvalid);
    if (c->h_own) (void)hipHostFree(c->h_own);
    if (c->nvEvent) (void)hipEventDestroy(c->nvEvent);
    for (hipEvent_t e : c->nvRingEv) if (e) (void)hipEventDestroy(e);
    if (c->ownEvent) (void)hipEventDestroy(c->ownEvent);
    if (c->link) c->link->invalidate();       // recorded steps that captured this context can no longer be launched
    delete c;
}

// everything that may still be reading a context's pinned limits or the matrix through it: its stream + the replays of recorded steps
static int ctx_quiesce(myslam_lcddb_query_ctx* c) {
    // a step that scans through this context is being RECORDED (between myslam_graph_begin and _end, possibly on another thread): synchronising the
    // capturing stream would invalidate the capture, and nothing of the recording is in flight yet — the caller's change (growth, scratch, stream) is
    // refused until the recording has ended
    if (c->link && c->link->captures_open.load() > 0) return MYSLAM_ERR_UNSUPPORTED;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->link) { const int rc = c->link->wait(); if (rc) return rc; }
    return MYSLAM_OK;
}

// The context's own scratch (pinned / device row limits, partial results, shard scratch) is about to be freed.  A recorded step names those buffers by
// address and its matrix generation is unchanged: give the link a new scratch epoch so that myslam_graph_launch refuses the old step
// (MYSLAM_ERR_CAPACITY: record it again) instead of replaying reads and writes of freed memory.  Called after ctx_quiesce (no replay in flight).
static void ctx_scratch_moves(myslam_lcddb_query_ctx* c) {
    if (c->graphRows > 0) {
        c->link->scratch_epoch.fetch_add(1);
        c->graphRows = 0; c->graphQueries = 0;
    }
}

static myslam_lcddb_query_ctx* ctx_new(myslam_lcddb* db, hipStream_t s, bool builtin) {
    myslam_lcddb_query_ctx* c = new myslam_lcddb_query_ctx();
    c->db = db; c->stream = s; c->builtin = builtin;
    c->link = std::make_shared<DbGraphLink>();
    c->link->generation.store(db->generation);
    return c;
}

extern "C" {

int myslam_lcddb_create(myslam_lcddb** out, int capacity) {
    if (!out || capacity < 1) return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    myslam_lcddb* h = new myslam_lcddb();
    h->ctxs.push_back(ctx_new(h, nullptr, true));
    const int rowsPerBlock = DB_WAVES * DB_ROWS_PER_WAVE;
    h->capacity = (capacity + rowsPerBlock - 1) / rowsPerBlock * rowsPerBlock;      // scan reads whole blocks of rows
    auto alloc_all = [&]() -> int {
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_db, (size_t)h->capacity * DIM * sizeof(float)));
        {   // (not hipMemset: the legacy stream may not be touched while another thread records a graph on a blocking stream, common.h)
            const hipStream_t us = host_call_stream();
            if (!us) return MYSLAM_ERR_HIP;
            MYSLAM_HIP_CHECK(hipMemsetAsync(h->d_db, 0, (size_t)h->capacity * DIM * sizeof(float), us));
            MYSLAM_HIP_CHECK(hipStreamSynchronize(us));
        }
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_ids, (size_t)h->capacity * sizeof(uint64_t)));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_q1, DIM * sizeof(float)));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_best1, 8)); MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_max1, 4));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_cnt1, 4));
        return MYSLAM_OK;
    };
    const int rc = alloc_all();
    if (rc) { (void)myslam_lcddb_destroy(h); return rc; }      // nothing half-built survives an allocation failure
    *out = h;
    return MYSLAM_OK;
}

int myslam_lcddb_destroy(myslam_lcddb* h) {
    if (!h) return MYSLAM_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    for (myslam_lcddb_query_ctx* c : h->ctxs) { (void)ctx_quiesce(c); ctx_free(c); }      // contexts die with their database
    void* ptrs[] = {h->d_db, h->d_ids, h->d_q1, h->d_best1, h->d_max1, h->d_cnt1};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int r = 0; r < myslam_lcddb::ID_RING; r++) { if (h->idPending[r]) (void)hipEventSynchronize(h->idEv[r]); if (h->idEv[r]) (void)hipEventDestroy(h->idEv[r]); }
    if (h->h_idring) (void)hipHostFree(h->h_idring);
    for (auto& r : h->retired) { (void)hipFree(r.first); (void)hipFree(r.second); }
    delete h;
    return MYSLAM_OK;
}

int myslam_lcddb_set_stream(myslam_lcddb* h, void* s) {
    if (!h) return MYSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipStreamSynchronize(h->stream);
    (void)ctx_quiesce(h->ctxs[0]);
    h->stream = (hipStream_t)s;
    h->ctxs[0]->stream = (hipStream_t)s;
    return MYSLAM_OK;
}

int myslam_lcddb_size(const myslam_lcddb* h) { return h ? h->n : MYSLAM_ERR_INVALID; }

int myslam_lcddb_generation(const myslam_lcddb* h) { return h ? (int)h->generation : MYSLAM_ERR_INVALID; }

int myslam_lcddb_query_ctx_create(myslam_lcddb_query_ctx** out, myslam_lcddb* db, void* hip_stream) {
    if (!out || !db) return MYSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(db->mu);
    myslam_lcddb_query_ctx* c = ctx_new(db, (hipStream_t)hip_stream, false);
    db->ctxs.push_back(c);
    *out = c;
    return MYSLAM_OK;
}

int myslam_lcddb_query_ctx_destroy(myslam_lcddb_query_ctx* c) {
    if (!c || c->builtin) return MYSLAM_ERR_INVALID;
    myslam_lcddb* db = c->db;
    std::lock_guard<std::mutex> lk(db->mu);
    const int rc = ctx_quiesce(c);
    auto it = std::find(db->ctxs.begin(), db->ctxs.end(), c);
    if (it != db->ctxs.end()) db->ctxs.erase(it);
    ctx_free(c);
    return rc;
}

// LoopClosing::_mvDatabase is an unbounded std::map (loopclosing.h:120, loopclosing.cpp:651-659): the device matrix grows with it.
// New storage, one device-to-device copy of the rows held so far, zeroed tail (the scan kernels read whole blocks of rows).
// Caller holds h->mu.
static int db_reserve(myslam_lcddb* h, long long rows) {
    const int rowsPerBlock = DB_WAVES * DB_ROWS_PER_WAVE;
    if (rows <= h->capacity) return MYSLAM_OK;
    if (rows > (long long)INT32_MAX - rowsPerBlock) return MYSLAM_ERR_CAPACITY;
    const int cap = (int)((rows + rowsPerBlock - 1) / rowsPerBlock * rowsPerBlock);
    for (myslam_lcddb_query_ctx* c : h->ctxs)                              // before anything synchronises: a recording in progress is not disturbed (see ctx_quiesce)
        if (c->link && c->link->captures_open.load() > 0) return MYSLAM_ERR_UNSUPPORTED;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    bool recorded = false;
    for (myslam_lcddb_query_ctx* c : h->ctxs) {                           // nothing in flight reads the old matrix: every context's stream and
        const int rc = ctx_quiesce(c);                                    // every replay of a recorded step that scans through it
        if (rc) return rc;
        recorded = recorded || c->graphRows > 0;
    }
    float* nd = nullptr; uint64_t* ni = nullptr;
    if (hipMalloc((void**)&nd, (size_t)cap * DIM * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return MYSLAM_ERR_CAPACITY; }
    if (hipMalloc((void**)&ni, (size_t)cap * sizeof(uint64_t)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(nd); return MYSLAM_ERR_CAPACITY; }
    auto move_rows = [&]() -> int {
        if (h->n) {
            MYSLAM_HIP_CHECK(hipMemcpyAsync(nd, h->d_db, (size_t)h->n * DIM * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
            MYSLAM_HIP_CHECK(hipMemcpyAsync(ni, h->d_ids, (size_t)h->n * sizeof(uint64_t), hipMemcpyDeviceToDevice, h->stream));
        }
        MYSLAM_HIP_CHECK(hipMemsetAsync(nd + (size_t)h->n * DIM, 0, (size_t)(cap - h->n) * DIM * sizeof(float), h->stream));
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        return MYSLAM_OK;
    };
    const int rc = move_rows();
    if (rc) { (void)hipFree(nd); (void)hipFree(ni); return rc; }
    // A recorded step names the old pointers in its kernel nodes.  Its launch is refused from now on (generation check in
    // myslam_graph_launch), but a caller that ignores the status must still not touch freed memory: the old matrix stays allocated
    // until the database is destroyed (geometric growth: all retired matrices together are smaller than the live one).
    if (recorded) h->retired.emplace_back(h->d_db, h->d_ids);
    else { (void)hipFree(h->d_db); (void)hipFree(h->d_ids); }
    h->d_db = nd; h->d_ids = ni; h->capacity = cap;
    h->generation++;
    for (myslam_lcddb_query_ctx* c : h->ctxs) c->link->generation.store(h->generation);
    return MYSLAM_OK;
}

static int db_append(myslam_lcddb* h, const uint64_t* ids, const float* src, int n, hipMemcpyKind kind) {
    if (!h || !ids || !src || n < 0) return MYSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    for (int i = 0; i < n; i++) {
        const uint64_t prev = (i == 0) ? (h->ids.empty() ? 0 : h->ids.back()) : ids[i - 1];
        const bool first = (i == 0 && h->ids.empty());
        if (!first && ids[i] <= prev) return MYSLAM_ERR_INVALID;          // std::map order: strictly ascending keys
    }
    if ((long long)h->n + n > h->capacity) {                              // geometric growth: AddToDatabase never fails for lack of room
        const int rc = db_reserve(h, std::max<long long>((long long)h->n + n, 2LL * h->capacity));
        if (rc) return rc;
    }
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_db + (size_t)h->n * DIM, src, (size_t)n * DIM * sizeof(float), kind, h->stream));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_ids + h->n, ids, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));                    // ids is a host pointer; the rows are in HBM when the call returns
    h->ids.insert(h->ids.end(), ids, ids + n);
    h->n += n;
    return MYSLAM_OK;
}

// AddToDatabase inside a pipelined step (round 6): the rows are copied on the CALLER's stream and the call does not wait for them — the caller orders later scans
// behind the copy (same stream, or an event), as it orders everything else of the step.  The host state (ids, row count) is updated at once, so the row limits of the next
// query already cover the new rows.  Never moves the matrix: MYSLAM_ERR_CAPACITY when the rows do not fit (myslam_lcddb_reserve ahead of the run).
static int db_append_async(myslam_lcddb* h, const uint64_t* ids, const float* d_src, int n, hipStream_t s) {
    if (!h || !ids || !d_src || n < 0) return MYSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    for (int i = 0; i < n; i++) {
        const uint64_t prev = (i == 0) ? (h->ids.empty() ? 0 : h->ids.back()) : ids[i - 1];
        if (!(i == 0 && h->ids.empty()) && ids[i] <= prev) return MYSLAM_ERR_INVALID;
    }
    if ((long long)h->n + n > h->capacity) return MYSLAM_ERR_CAPACITY;
    if (n == 0) return MYSLAM_OK;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_db + (size_t)h->n * DIM, d_src, (size_t)n * DIM * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (n <= myslam_lcddb::ID_SLOT) {                                  // ids through a pinned ring slot: the call never waits for the stream
        if (!h->h_idring) {
            MYSLAM_HIP_CHECK(hipHostMalloc((void**)&h->h_idring, sizeof(uint64_t) * myslam_lcddb::ID_RING * myslam_lcddb::ID_SLOT));
            for (auto& e : h->idEv) MYSLAM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const int r = h->idSlot; h->idSlot = (h->idSlot + 1) % myslam_lcddb::ID_RING;
        if (h->idPending[r]) MYSLAM_HIP_CHECK(hipEventSynchronize(h->idEv[r]));
        uint64_t* st = h->h_idring + (size_t)r * myslam_lcddb::ID_SLOT;
        memcpy(st, ids, sizeof(uint64_t) * n);
        MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_ids + h->n, st, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        MYSLAM_HIP_CHECK(hipEventRecord(h->idEv[r], s)); h->idPending[r] = true;
    } else {
        MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_ids + h->n, ids, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, s));      // pageable source: staged before the call returns
    }
    h->ids.insert(h->ids.end(), ids, ids + n);
    h->n += n;
    return MYSLAM_OK;
}

int myslam_lcddb_capacity(const myslam_lcddb* h) { return h ? h->capacity : MYSLAM_ERR_INVALID; }

int myslam_lcddb_append_batch_async(myslam_lcddb* h, const uint64_t* ids, const float* d_descr, int n, void* hip_stream) {
    return db_append_async(h, ids, d_descr, n, (hipStream_t)hip_stream);
}

int myslam_lcddb_reserve(myslam_lcddb* h, int rows) {
    if (!h || rows < 0) return MYSLAM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    return db_reserve(h, rows);
}

int myslam_lcddb_append(myslam_lcddb* h, uint64_t id, const float* descr) { return db_append(h, &id, descr, 1, hipMemcpyHostToDevice); }

int myslam_lcddb_append_batch(myslam_lcddb* h, const uint64_t* ids, const float* d_descr, int n) {
    return db_append(h, ids, d_descr, n, hipMemcpyDeviceToDevice);
}

}  // extern "C"

static bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}

static int db_query(myslam_lcddb_query_ctx* c, const float* d_q, const uint64_t* cur_ids_host, int nq, float thr_low, uint64_t* d_best,
                    float* d_max, int32_t* d_cnt) {
    myslam_lcddb* h = c->db;
    const int rowsPerBlock = DB_WAVES * DB_ROWS_PER_WAVE;
    // Recorded into a HIP graph (graph.hip)?  Then nothing here may synchronise or allocate, the copy node of the row limits reads the
    // pinned buffer at EVERY replay (myslam_lcddb_ctx_update_query_limits rewrites it), and the launch covers every row of the
    // ALLOCATION (not only the rows today's limits reach), so that later limits and appends stay inside the captured grid.
    const bool cap = stream_is_capturing(c->stream);
    if (nq > c->nvalidCap) {
        if (cap) return MYSLAM_ERR_UNSUPPORTED;                          // first call with this many queries: run it once outside the capture
        const int rq = ctx_quiesce(c);
        if (rq) return rq;
        ctx_scratch_moves(c);
        if (c->d_nvalid) (void)hipFree(c->d_nvalid);
        if (c->h_nvalid) (void)hipHostFree(c->h_nvalid);
        c->d_nvalid = nullptr; c->h_nvalid = nullptr; c->nvalidCap = 0;
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_nvalid, sizeof(int32_t) * (nq + 1)));
        MYSLAM_HIP_CHECK(hipHostMalloc((void**)&c->h_nvalid, sizeof(int32_t) * (nq + 1) * (1 + myslam_lcddb_query_ctx::NV_RING)));
        if (!c->nvEvent) MYSLAM_HIP_CHECK(hipEventCreateWithFlags(&c->nvEvent, hipEventDisableTiming));
        for (int r = 0; r < myslam_lcddb_query_ctx::NV_RING; r++) {
            if (!c->nvRingEv[r]) MYSLAM_HIP_CHECK(hipEventCreateWithFlags(&c->nvRingEv[r], hipEventDisableTiming));
            c->nvRingPending[r] = false;                              // (ctx_quiesce above: nothing of the old buffer is in flight)
        }
        c->nvalidCap = nq; c->nvFresh = false;
    }
    // the row limits of this call; when they equal what the device buffer already holds (the same cur_ids against the same rows, the
    // usual case of a batch of queries per step) nothing is uploaded and the host never waits for the device
    // h->mu is held from here until the scan and the reduce are ENQUEUED (round 6): the launches name the matrix by address, and an append from another
    // thread may grow the database (db_reserve) — it takes the same mutex, then synchronises this context's stream, which now covers these launches;
    // released earlier, the old matrix could be freed between the pointer read and the launch
    int maxv = 0, rows_now, cap_now; float* d_db; uint64_t* d_ids; uint64_t gen;
    bool same;
    std::lock_guard<std::mutex> lk(h->mu);
    {
        rows_now = h->n; cap_now = h->capacity; d_db = h->d_db; d_ids = h->d_ids; gen = h->generation;
        same = c->nvFresh && !cap && (int)c->lastLimits.size() == nq && c->lastRows == rows_now;
        c->scratchLimits.resize(nq);
        for (int i = 0; i < nq; i++) {
            const int v = h->n_valid(cur_ids_host[i]);
            c->scratchLimits[i] = v; maxv = std::max(maxv, v);
            if (same && c->lastLimits[i] != v) same = false;
        }
    }
    if (cap) maxv = cap_now;
    const int nblocks = std::max(1, (maxv + rowsPerBlock - 1) / rowsPerBlock);
    // partial results: sized for the whole allocation and this call's kernel (a later recording of the same query covers the allocation
    // and must find its scratch in place: nothing may be allocated inside a capture)
    const size_t need = (size_t)std::max(1, nq >= 32 ? (cap_now + GM - 1) / GM : (cap_now + rowsPerBlock - 1) / rowsPerBlock) * nq;
    if (need > c->partialsCap) {
        if (cap) return MYSLAM_ERR_UNSUPPORTED;
        const int rq = ctx_quiesce(c);
        if (rq) return rq;
        ctx_scratch_moves(c);
        if (c->d_partials) (void)hipFree(c->d_partials);
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_partials, need * sizeof(Partial)));
        c->partialsCap = need;
    }
    if (cap) {
        c->graphRows = maxv; c->graphQueries = nq; c->graphGen = gen;
        const int rn = graph_note_db_link(c->link, gen);                 // the step being recorded learns which matrix it reads (graph.hip)
        if (rn) return rn;
    }
    if (!same) {
        // eager: one small upload out of a ring slot (slots are nvalidCap + 1 ints apart; slot 0 belongs to the recorded scans).  Recorded: NO copy node — the
        // recorded kernels read slot 0 of the pinned buffer itself (device-visible host memory) at every replay; a recorded one-pair step is bound by the number of
        // its nodes (~4.6 us each, tools/node_count_probe.sh), and a few hundred 4-byte reads over the host link cost the scan ~2 us
        int32_t* stage = c->h_nvalid;
        int ring = -1;
        if (!cap) {
            ring = c->nvSlot; c->nvSlot = (c->nvSlot + 1) % myslam_lcddb_query_ctx::NV_RING;
            if (c->nvRingPending[ring]) MYSLAM_HIP_CHECK(hipEventSynchronize(c->nvRingEv[ring]));      // the upload NV_RING calls back has left this slot
            stage = c->h_nvalid + (size_t)(1 + ring) * (c->nvalidCap + 1);
        }
        memcpy(stage, c->scratchLimits.data(), sizeof(int32_t) * nq);
        stage[nq] = rows_now;
        if (!cap) {
            MYSLAM_HIP_CHECK(hipMemcpyAsync(c->d_nvalid, stage, sizeof(int32_t) * (nq + 1), hipMemcpyHostToDevice, c->stream));      // pinned -> no host sync
            MYSLAM_HIP_CHECK(hipEventRecord(c->nvRingEv[ring], c->stream)); c->nvRingPending[ring] = true;
        }
        c->lastLimits = c->scratchLimits; c->lastRows = rows_now; c->nvFresh = !cap;             // (a recorded scan reads whatever the pinned buffer holds at its replay)
    }
    c->limits = cap ? c->h_nvalid : c->d_nvalid;
    {
        ScopedProf sp(P_DBSCAN, c->stream);
        int nparts = nblocks;
        if (nq >= 32) {           // batched: GEMM on the matrix cores with the per-query reduction fused into the epilogue
            nparts = std::max(1, (maxv + GM - 1) / GM);
            hipLaunchKernelGGL(k_db_scan_bf16x6, dim3(nparts, (nq + GN - 1) / GN), dim3(256), 0, c->stream, d_db, cap_now, d_q, nq,
                               c->limits, (const int32_t*)nullptr, thr_low, c->d_partials);
        } else {                  // a few queries: bandwidth-bound GEMV, one wave per database row
            const size_t lds = sizeof(Partial) * DB_WAVES * nq + 2 * sizeof(int) * nq;
            hipLaunchKernelGGL(k_db_scan, dim3(nblocks), dim3(256), lds, c->stream, d_db, d_q, nq, c->limits, (const int32_t*)nullptr, thr_low, c->d_partials);
        }
        hipLaunchKernelGGL(k_db_reduce, dim3((nq + 3) / 4), dim3(256), 0, c->stream, c->d_partials, nparts, nq, d_ids, d_best,
                           d_max, d_cnt);
    }
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

static int db_update_limits(myslam_lcddb_query_ctx* c, const uint64_t* cur_ids, int nq) {
    myslam_lcddb* h = c->db;
    if (!c->graphRows || nq > c->graphQueries || nq > c->nvalidCap) return MYSLAM_ERR_INVALID;      // no recorded query to feed
    std::lock_guard<std::mutex> lk(h->mu);
    if (c->graphGen != h->generation || h->n > c->graphRows) return MYSLAM_ERR_CAPACITY;      // the database moved: record the step again
    const int rw = c->link->wait();                                   // no replay — on whatever stream it was launched — is reading the pinned buffer
    if (rw) return rw;
    MYSLAM_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < nq; i++) c->h_nvalid[i] = h->n_valid(cur_ids[i]);
    c->h_nvalid[c->graphQueries] = h->n;
    c->nvFresh = false;                                               // the next replay rewrites d_nvalid behind the eager path's back
    return MYSLAM_OK;
}

static int db_query_sharded(myslam_lcddb_query_ctx* c, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low, myslam_lcd_candidate* d_cand) {
    if (nq > c->shardCap) {
        if (stream_is_capturing(c->stream)) return MYSLAM_ERR_UNSUPPORTED;
        const int rq = ctx_quiesce(c);
        if (rq) return rq;
        ctx_scratch_moves(c);
        void* old[] = {c->d_bestS, c->d_maxS, c->d_cntS};
        for (void* p : old) if (p) (void)hipFree(p);
        c->d_bestS = nullptr; c->d_maxS = nullptr; c->d_cntS = nullptr; c->shardCap = 0;
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_bestS, sizeof(uint64_t) * nq));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_maxS, sizeof(float) * nq));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_cntS, sizeof(int32_t) * nq));
        c->shardCap = nq;
    }
    int rc = db_query(c, d_q, cur_ids, nq, thr_low, c->d_bestS, c->d_maxS, c->d_cntS);
    if (rc) return rc;
    // the row count sits behind the nq limits of THIS call in d_nvalid (a recorded step's count is refreshed with its limits)
    hipLaunchKernelGGL(k_db_pack_candidates, dim3((nq + 255) / 256), dim3(256), 0, c->stream, c->d_bestS, c->d_maxS, c->d_cntS, c->limits,
                       c->limits + nq, nq, d_cand);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

// One shard of a database whose ids interleave across shards: both parts of the reference's scan + the break flag, 32 bytes per query.
// Two scans of the same kernels (rows below the window; rows above cur — launched only when some query has such rows), not recordable into a step graph.
static int db_query_owned(myslam_lcddb_query_ctx* c, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low, myslam_lcd_owned_candidate* d_cand) {
    myslam_lcddb* h = c->db;
    if (stream_is_capturing(c->stream)) return MYSLAM_ERR_UNSUPPORTED;
    const int rowsPerBlock = DB_WAVES * DB_ROWS_PER_WAVE;
    if (nq > c->ownCap) {
        const int rq = ctx_quiesce(c);
        if (rq) return rq;
        void* old[] = {c->d_own, c->d_bestO, c->d_maxO, c->d_cntO};
        for (void* p : old) if (p) (void)hipFree(p);
        if (c->h_own) (void)hipHostFree(c->h_own);
        c->d_own = nullptr; c->h_own = nullptr; c->d_bestO = nullptr; c->d_maxO = nullptr; c->d_cntO = nullptr; c->ownCap = 0; c->ownPending = false;
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_own, sizeof(int32_t) * 4 * nq));
        MYSLAM_HIP_CHECK(hipHostMalloc((void**)&c->h_own, sizeof(int32_t) * 4 * nq));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_bestO, sizeof(uint64_t) * 2 * nq));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_maxO, sizeof(float) * 2 * nq));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_cntO, sizeof(int32_t) * 2 * nq));
        if (!c->ownEvent) MYSLAM_HIP_CHECK(hipEventCreateWithFlags(&c->ownEvent, hipEventDisableTiming));
        c->ownCap = nq;
    }
    if (c->ownPending) MYSLAM_HIP_CHECK(hipEventSynchronize(c->ownEvent));          // the previous call's upload has left the pinned block
    std::lock_guard<std::mutex> lk(h->mu);                                          // held until everything is enqueued (see db_query)
    const int cap_now = h->capacity;
    int32_t* pLim = c->h_own; int32_t* sBeg = pLim + nq; int32_t* sLim = sBeg + nq; int32_t* brk = sLim + nq;
    int maxP = 0, maxS = 0; bool any_suffix = false;
    for (int i = 0; i < nq; i++) {
        int p, sb, se; bool broke;
        h->scan_ranges(cur_ids[i], p, broke, sb, se);
        pLim[i] = p; sBeg[i] = sb; sLim[i] = se; brk[i] = broke ? 1 : 0;
        maxP = std::max(maxP, p);
        if (se > sb) { any_suffix = true; maxS = std::max(maxS, se); }
    }
    MYSLAM_HIP_CHECK(hipMemcpyAsync(c->d_own, c->h_own, sizeof(int32_t) * 4 * nq, hipMemcpyHostToDevice, c->stream));
    MYSLAM_HIP_CHECK(hipEventRecord(c->ownEvent, c->stream)); c->ownPending = true;
    const size_t need = (size_t)std::max(1, nq >= 32 ? (cap_now + GM - 1) / GM : (cap_now + rowsPerBlock - 1) / rowsPerBlock) * nq;
    if (need > c->partialsCap) {
        const int rq = ctx_quiesce(c);
        if (rq) return rq;
        ctx_scratch_moves(c);
        if (c->d_partials) (void)hipFree(c->d_partials);
        c->d_partials = nullptr; c->partialsCap = 0;
        MYSLAM_HIP_CHECK(hipMalloc((void**)&c->d_partials, need * sizeof(Partial)));
        c->partialsCap = need;
    }
    const int32_t* dP = c->d_own; const int32_t* dSb = dP + nq; const int32_t* dSl = dSb + nq; const int32_t* dBr = dSl + nq;
    auto scan = [&](const int32_t* lim, const int32_t* beg, int maxv, uint64_t* best, float* mx, int32_t* cnt) {
        int nparts;
        if (nq >= 32) {
            nparts = std::max(1, (maxv + GM - 1) / GM);
            hipLaunchKernelGGL(k_db_scan_bf16x6, dim3(nparts, (nq + GN - 1) / GN), dim3(256), 0, c->stream, h->d_db, cap_now, d_q, nq, lim, beg, thr_low, c->d_partials);
        } else {
            nparts = std::max(1, (maxv + rowsPerBlock - 1) / rowsPerBlock);
            const size_t lds = sizeof(Partial) * DB_WAVES * nq + 2 * sizeof(int) * nq;
            hipLaunchKernelGGL(k_db_scan, dim3(nparts), dim3(256), lds, c->stream, h->d_db, d_q, nq, lim, beg, thr_low, c->d_partials);
        }
        hipLaunchKernelGGL(k_db_reduce, dim3((nq + 3) / 4), dim3(256), 0, c->stream, c->d_partials, nparts, nq, h->d_ids, best, mx, cnt);
    };
    {
        ScopedProf sp(P_DBSCAN, c->stream);
        scan(dP, nullptr, maxP, c->d_bestO, c->d_maxO, c->d_cntO);
        if (any_suffix) scan(dSl, dSb, maxS, c->d_bestO + nq, c->d_maxO + nq, c->d_cntO + nq);       // (the partials are reused: same stream, in order)
    }
    hipLaunchKernelGGL(k_db_pack_owned, dim3((nq + 255) / 256), dim3(256), 0, c->stream, c->d_bestO, c->d_maxO, c->d_cntO, c->d_bestO + nq, c->d_maxO + nq,
                       c->d_cntO + nq, dBr, any_suffix ? 1 : 0, nq, d_cand);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

extern "C" {

int myslam_lcddb_query_batch_owned(myslam_lcddb* h, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low, myslam_lcd_owned_candidate* d_cand) {
    if (!h || !d_q || !cur_ids || nq < 1 || nq > 65536 || !d_cand) return MYSLAM_ERR_INVALID;
    return db_query_owned(h->ctxs[0], d_q, cur_ids, nq, thr_low, d_cand);
}

int myslam_lcddb_ctx_query_batch_owned(myslam_lcddb_query_ctx* c, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low,
                                       myslam_lcd_owned_candidate* d_cand) {
    if (!c || !d_q || !cur_ids || nq < 1 || nq > 65536 || !d_cand) return MYSLAM_ERR_INVALID;
    return db_query_owned(c, d_q, cur_ids, nq, thr_low, d_cand);
}

int myslam_lcd_merge_owned_candidates(const myslam_lcd_owned_candidate* gathered, int nshards, int nq, uint64_t* best_id, float* max_score, int32_t* cnt) {
    if (!gathered || nshards < 1 || nq < 0 || !best_id || !max_score || !cnt) return MYSLAM_ERR_INVALID;
    for (int qi = 0; qi < nq; qi++) merge_owned_one(gathered, nshards, nq, qi, best_id[qi], max_score[qi], cnt[qi]);
    return MYSLAM_OK;
}

int myslam_lcd_merge_owned_candidates_device(const myslam_lcd_owned_candidate* d_gathered, int nshards, int nq, uint64_t* d_best_id, float* d_max_score,
                                             int32_t* d_cnt, void* hip_stream) {
    if (!d_gathered || nshards < 1 || nq < 1 || !d_best_id || !d_max_score || !d_cnt) return MYSLAM_ERR_INVALID;
    hipLaunchKernelGGL(k_db_merge_owned, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, d_gathered, nshards, nq, d_best_id, d_max_score, d_cnt);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

int myslam_lcddb_query_batch(myslam_lcddb* h, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low,
                             uint64_t* d_best_id, float* d_max_score, int32_t* d_cnt) {
    if (!h || !d_q || !cur_ids || nq < 1 || nq > 65536 || !d_best_id || !d_max_score || !d_cnt) return MYSLAM_ERR_INVALID;
    return db_query(h->ctxs[0], d_q, cur_ids, nq, thr_low, d_best_id, d_max_score, d_cnt);
}

int myslam_lcddb_ctx_query_batch(myslam_lcddb_query_ctx* c, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low,
                                 uint64_t* d_best_id, float* d_max_score, int32_t* d_cnt) {
    if (!c || !d_q || !cur_ids || nq < 1 || nq > 65536 || !d_best_id || !d_max_score || !d_cnt) return MYSLAM_ERR_INVALID;
    return db_query(c, d_q, cur_ids, nq, thr_low, d_best_id, d_max_score, d_cnt);
}

int myslam_lcddb_update_query_limits(myslam_lcddb* h, const uint64_t* cur_ids, int nq) {
    if (!h || !cur_ids || nq < 1) return MYSLAM_ERR_INVALID;
    return db_update_limits(h->ctxs[0], cur_ids, nq);
}

int myslam_lcddb_ctx_update_query_limits(myslam_lcddb_query_ctx* c, const uint64_t* cur_ids, int nq) {
    if (!c || !cur_ids || nq < 1) return MYSLAM_ERR_INVALID;
    return db_update_limits(c, cur_ids, nq);
}

int myslam_lcddb_query_batch_sharded(myslam_lcddb* h, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low,
                                     myslam_lcd_candidate* d_cand) {
    if (!h || !d_q || !cur_ids || nq < 1 || nq > 65536 || !d_cand) return MYSLAM_ERR_INVALID;
    return db_query_sharded(h->ctxs[0], d_q, cur_ids, nq, thr_low, d_cand);
}

int myslam_lcddb_ctx_query_batch_sharded(myslam_lcddb_query_ctx* c, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low,
                                         myslam_lcd_candidate* d_cand) {
    if (!c || !d_q || !cur_ids || nq < 1 || nq > 65536 || !d_cand) return MYSLAM_ERR_INVALID;
    return db_query_sharded(c, d_q, cur_ids, nq, thr_low, d_cand);
}

int myslam_lcd_merge_candidates(const myslam_lcd_candidate* gathered, int nshards, int nq, uint64_t* best_id, float* max_score, int32_t* cnt) {
    if (!gathered || nshards < 1 || nq < 0 || !best_id || !max_score || !cnt) return MYSLAM_ERR_INVALID;
    for (int qi = 0; qi < nq; qi++) merge_one(gathered, nshards, nq, qi, best_id[qi], max_score[qi], cnt[qi]);
    return MYSLAM_OK;
}

int myslam_lcd_merge_candidates_device(const myslam_lcd_candidate* d_gathered, int nshards, int nq, uint64_t* d_best_id, float* d_max_score,
                                       int32_t* d_cnt, void* hip_stream) {
    if (!d_gathered || nshards < 1 || nq < 1 || !d_best_id || !d_max_score || !d_cnt) return MYSLAM_ERR_INVALID;
    hipLaunchKernelGGL(k_db_merge_candidates, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, d_gathered, nshards, nq, d_best_id,
                       d_max_score, d_cnt);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

int myslam_lcddb_query(myslam_lcddb* h, const float* descr, uint64_t cur_id, float thr_low, uint64_t* best_id, float* max_score,
                       int* cnt) {
    if (!h || !descr || !best_id || !max_score || !cnt) return MYSLAM_ERR_INVALID;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_q1, descr, DIM * sizeof(float), hipMemcpyHostToDevice, h->stream));
    int rc = db_query(h->ctxs[0], h->d_q1, &cur_id, 1, thr_low, h->d_best1, h->d_max1, h->d_cnt1);
    if (rc) return rc;
    int32_t c = 0;
    MYSLAM_HIP_CHECK(hipMemcpyAsync(best_id, h->d_best1, 8, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(max_score, h->d_max1, 4, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(&c, h->d_cnt1, 4, hipMemcpyDeviceToHost, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
    *cnt = c;
    return MYSLAM_OK;
}

}  // extern "C"
