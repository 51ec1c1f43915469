// lcddb.hip — key-frame descriptor database and cosine scan on gfx950.
// Replaces LoopClosing::_mvDatabase + DetectLoop()/AddToDatabase() (reference include/myslam/loopclosing.h:67,120;
// src/loopclosing.cpp:124-161, 651-659): ascending-id scan, stop at the first KF with cur_id - id < 20,
// maxScore (init 0, strict '>': lowest id wins ties), cnt = #{score > thr_low}.
//
// HBM layout: row-major f32 [capacity][1064] (4256-byte rows), ids kept on the host (ascending).
// Scan kernel: one wave per database row, the row lives in registers (17 floats per lane) and is dotted
// against every query of the batch, so a batch of queries streams the database from HBM exactly once.
#include <algorithm>
#include <vector>

#include "common.h"

namespace myslam_hip {

constexpr int DIM = MYSLAM_LCD_DIM;     // 1064 = 16*64 + 40
constexpr int DB_WAVES = 4;             // waves per block
constexpr int DB_ROWS_PER_WAVE = 8;

struct Partial { float score; int32_t idx; int32_t cnt; };

__global__ __launch_bounds__(256) void k_db_scan(const float* __restrict__ db, const float* __restrict__ q, int nq,
                                                 const int32_t* __restrict__ nvalid, float thr_low,
                                                 Partial* __restrict__ partials /*[nblocks][nq]*/) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_db[];
    Partial* s_p = reinterpret_cast<Partial*>(smem_db);          // [DB_WAVES][nq]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
            if (r >= nvalid[qi]) continue;                      // cut-off rule, loopclosing.cpp:133 (uniform per wave)
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
                                                        int nq, const int32_t* __restrict__ nvalid, float thr_low,
                                                        Partial* __restrict__ partials) {
    MYSLAM_SIDE_PRIO();
    __shared__ uint4 s_a[2][3][GM * 2];                // [piece][row][k half] x 8 bf16
    __shared__ uint4 s_b[2][3][GN * 2];
    __shared__ Partial s_p[2][GN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int m0 = blockIdx.x * GM, n0 = blockIdx.y * GN;
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
        float bs = 0.f; int bi = -1, cnt = 0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float sc = acc[i][j][r];
                if (m < nv) {
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

__global__ __launch_bounds__(256) void k_db_reduce(const Partial* __restrict__ partials, int nblocks, int nq,
                                                   const uint64_t* __restrict__ ids, uint64_t* __restrict__ best_id,
                                                   float* __restrict__ max_score, int32_t* __restrict__ cnt) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    float ms = 0.f; int bi = -1; int c = 0;
    for (int b = 0; b < nblocks; b++) {                         // blocks own ascending row ranges
        const Partial p = partials[(size_t)b * nq + qi];
        if (p.score > ms) { ms = p.score; bi = p.idx; }
        c += p.cnt;
    }
    best_id[qi] = (bi >= 0) ? ids[bi] : 0;                      // bestId initialised to 0, loopclosing.cpp:129
    max_score[qi] = ms; cnt[qi] = c;
}

// per-shard result of a query in the layout that travels between ranks: cnt bit 31 = this shard's scan hit the break (:133)
__global__ __launch_bounds__(256) void k_db_pack_candidates(const uint64_t* __restrict__ best_id, const float* __restrict__ max_score,
                                                            const int32_t* __restrict__ cnt, const int32_t* __restrict__ nvalid, int nrows,
                                                            int nq, myslam_lcd_candidate* __restrict__ out) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
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

}  // namespace myslam_hip

using namespace myslam_hip;

struct myslam_lcddb {
    hipStream_t stream = nullptr;
    int capacity = 0, n = 0;
    float* d_db = nullptr;
    uint64_t* d_ids = nullptr;
    std::vector<uint64_t> ids;
    Partial* d_partials = nullptr; size_t partialsCap = 0;
    int32_t* d_nvalid = nullptr; int nvalidCap = 0;
    int32_t* h_nvalid = nullptr; hipEvent_t nvEvent = nullptr;      // pinned staging of the per-query row limits + "copy done" event
    float* d_q1 = nullptr; uint64_t* d_best1 = nullptr; float* d_max1 = nullptr; int32_t* d_cnt1 = nullptr;
    uint64_t* d_bestS = nullptr; float* d_maxS = nullptr; int32_t* d_cntS = nullptr; int shardCap = 0;     // scratch of the sharded query
    int graphRows = 0, graphQueries = 0;      // > 0: a query of this handle was recorded into a HIP graph covering this many rows / queries
    std::vector<int32_t> lastLimits, scratchLimits; bool nvFresh = false, nvPending = false;      // what d_nvalid holds (skip identical uploads)
    bool graphStale = false;                  // the matrix moved (db_reserve) after a query was recorded: the recorded step reads freed memory

    // index of the first row the reference's scan does NOT look at: it breaks at the first id with
    // (cur - id) < 20 in unsigned arithmetic (loopclosing.cpp:133), i.e. id in [cur-19, cur] mod 2^64
    int first_in(uint64_t a, uint64_t b) const {
        auto it = std::lower_bound(ids.begin(), ids.end(), a);
        if (it != ids.end() && *it <= b) return (int)(it - ids.begin());
        return (int)ids.size();
    }
    int n_valid(uint64_t cur) const {
        if (cur >= 19) return first_in(cur - 19, cur);
        return std::min(first_in(0, cur), first_in(UINT64_MAX - (18 - cur), UINT64_MAX));
    }
};

extern "C" {

int myslam_lcddb_create(myslam_lcddb** out, int capacity) {
    if (!out || capacity < 1) return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    myslam_lcddb* h = new myslam_lcddb();
    const int rowsPerBlock = DB_WAVES * DB_ROWS_PER_WAVE;
    h->capacity = (capacity + rowsPerBlock - 1) / rowsPerBlock * rowsPerBlock;      // scan reads whole blocks of rows
    auto alloc_all = [&]() -> int {
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_db, (size_t)h->capacity * DIM * sizeof(float)));
        MYSLAM_HIP_CHECK(hipMemset(h->d_db, 0, (size_t)h->capacity * DIM * sizeof(float)));
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
    void* ptrs[] = {h->d_db, h->d_ids, h->d_partials, h->d_nvalid, h->d_q1, h->d_best1, h->d_max1, h->d_cnt1, h->d_bestS, h->d_maxS, h->d_cntS};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->h_nvalid) (void)hipHostFree(h->h_nvalid);
    if (h->nvEvent) (void)hipEventDestroy(h->nvEvent);
    delete h;
    return MYSLAM_OK;
}

int myslam_lcddb_set_stream(myslam_lcddb* h, void* s) {
    if (!h) return MYSLAM_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    h->stream = (hipStream_t)s;
    return MYSLAM_OK;
}

int myslam_lcddb_size(const myslam_lcddb* h) { return h ? h->n : MYSLAM_ERR_INVALID; }

// LoopClosing::_mvDatabase is an unbounded std::map (loopclosing.h:120, loopclosing.cpp:651-659): the device matrix grows with it.
// New storage, one device-to-device copy of the rows held so far, zeroed tail (the scan kernels read whole blocks of rows).
static int db_reserve(myslam_lcddb* h, long long rows) {
    const int rowsPerBlock = DB_WAVES * DB_ROWS_PER_WAVE;
    if (rows <= h->capacity) return MYSLAM_OK;
    if (rows > (long long)INT32_MAX - rowsPerBlock) return MYSLAM_ERR_CAPACITY;
    const int cap = (int)((rows + rowsPerBlock - 1) / rowsPerBlock * rowsPerBlock);
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));                    // nothing in flight reads the old matrix
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
    (void)hipFree(h->d_db); (void)hipFree(h->d_ids);
    h->d_db = nd; h->d_ids = ni; h->capacity = cap;
    if (h->graphRows) h->graphStale = true;
    return MYSLAM_OK;
}

static int db_append(myslam_lcddb* h, const uint64_t* ids, const float* src, int n, hipMemcpyKind kind) {
    if (!h || !ids || !src || n < 0) return MYSLAM_ERR_INVALID;
    if ((long long)h->n + n > h->capacity) {                              // geometric growth: AddToDatabase never fails for lack of room
        const int rc = db_reserve(h, std::max<long long>((long long)h->n + n, 2LL * h->capacity));
        if (rc) return rc;
    }
    for (int i = 0; i < n; i++) {
        const uint64_t prev = (i == 0) ? (h->ids.empty() ? 0 : h->ids.back()) : ids[i - 1];
        const bool first = (i == 0 && h->ids.empty());
        if (!first && ids[i] <= prev) return MYSLAM_ERR_INVALID;          // std::map order: strictly ascending keys
    }
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_db + (size_t)h->n * DIM, src, (size_t)n * DIM * sizeof(float), kind, h->stream));
    MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_ids + h->n, ids, (size_t)n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));                    // ids is a host pointer
    h->ids.insert(h->ids.end(), ids, ids + n);
    h->n += n;
    return MYSLAM_OK;
}

int myslam_lcddb_capacity(const myslam_lcddb* h) { return h ? h->capacity : MYSLAM_ERR_INVALID; }

int myslam_lcddb_reserve(myslam_lcddb* h, int rows) {
    if (!h || rows < 0) return MYSLAM_ERR_INVALID;
    return db_reserve(h, rows);
}

int myslam_lcddb_append(myslam_lcddb* h, uint64_t id, const float* descr) { return db_append(h, &id, descr, 1, hipMemcpyHostToDevice); }

int myslam_lcddb_append_batch(myslam_lcddb* h, const uint64_t* ids, const float* d_descr, int n) {
    return db_append(h, ids, d_descr, n, hipMemcpyDeviceToDevice);
}

static bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusActive;
}

static int db_query(myslam_lcddb* h, const float* d_q, const uint64_t* cur_ids_host, int nq, float thr_low, uint64_t* d_best,
                    float* d_max, int32_t* d_cnt) {
    const int rowsPerBlock = DB_WAVES * DB_ROWS_PER_WAVE;
    // Recorded into a HIP graph (graph.hip)?  Then nothing here may synchronise or allocate, the copy node of the row limits reads the
    // pinned buffer at EVERY replay (myslam_lcddb_update_query_limits rewrites it), and the launch covers every row the database holds
    // (not only the rows today's limits reach), so that later limits stay inside the captured grid.
    const bool cap = stream_is_capturing(h->stream);
    if (nq > h->nvalidCap) {
        if (cap) return MYSLAM_ERR_UNSUPPORTED;                          // first call with this many queries: run it once outside the capture
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->d_nvalid) (void)hipFree(h->d_nvalid);
        if (h->h_nvalid) (void)hipHostFree(h->h_nvalid);
        h->d_nvalid = nullptr; h->h_nvalid = nullptr; h->nvalidCap = 0;
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_nvalid, sizeof(int32_t) * nq));
        MYSLAM_HIP_CHECK(hipHostMalloc((void**)&h->h_nvalid, sizeof(int32_t) * nq));
        if (!h->nvEvent) MYSLAM_HIP_CHECK(hipEventCreateWithFlags(&h->nvEvent, hipEventDisableTiming));
        h->nvalidCap = nq; h->nvFresh = false;
    }
    // the row limits of this call; when they equal what the device buffer already holds (the same cur_ids against the same rows, the
    // usual case of a batch of queries per step) nothing is uploaded and the host never waits for the device
    int maxv = 0;
    bool same = h->nvFresh && !cap && (int)h->lastLimits.size() == nq;
    h->scratchLimits.resize(nq);
    for (int i = 0; i < nq; i++) {
        const int v = h->n_valid(cur_ids_host[i]);
        h->scratchLimits[i] = v; maxv = std::max(maxv, v);
        if (same && h->lastLimits[i] != v) same = false;
    }
    if (cap) { maxv = std::max(maxv, h->n); h->graphRows = maxv; h->graphQueries = nq; h->graphStale = false; }
    const int nblocks = std::max(1, (maxv + rowsPerBlock - 1) / rowsPerBlock);
    const size_t need = (size_t)std::max(nblocks, std::max(1, (maxv + GM - 1) / GM)) * nq;
    if (need > h->partialsCap) {
        if (cap) return MYSLAM_ERR_UNSUPPORTED;
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->d_partials) (void)hipFree(h->d_partials);
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_partials, need * sizeof(Partial)));
        h->partialsCap = need;
    }
    if (!same) {
        if (!cap) {
            if (h->graphRows) MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));      // a recorded step also reads the pinned buffer: wait for its replays
            else if (h->nvPending) MYSLAM_HIP_CHECK(hipEventSynchronize(h->nvEvent));   // the previous upload has left the pinned buffer
        }
        memcpy(h->h_nvalid, h->scratchLimits.data(), sizeof(int32_t) * nq);
        MYSLAM_HIP_CHECK(hipMemcpyAsync(h->d_nvalid, h->h_nvalid, sizeof(int32_t) * nq, hipMemcpyHostToDevice, h->stream));      // pinned -> no host sync
        if (!cap) { MYSLAM_HIP_CHECK(hipEventRecord(h->nvEvent, h->stream)); h->nvPending = true; }
        h->lastLimits = h->scratchLimits; h->nvFresh = !cap;             // (a recorded copy re-runs at every replay with whatever the pinned buffer holds then)
    }
    {
        ScopedProf sp(P_DBSCAN, h->stream);
        int nparts = nblocks;
        if (nq >= 32) {           // batched: GEMM on the matrix cores with the per-query reduction fused into the epilogue
            nparts = std::max(1, (maxv + GM - 1) / GM);
            hipLaunchKernelGGL(k_db_scan_bf16x6, dim3(nparts, (nq + GN - 1) / GN), dim3(256), 0, h->stream, h->d_db, h->capacity, d_q, nq,
                               h->d_nvalid, thr_low, h->d_partials);
        } else {                  // a few queries: bandwidth-bound GEMV, one wave per database row
            const size_t lds = sizeof(Partial) * DB_WAVES * nq;
            hipLaunchKernelGGL(k_db_scan, dim3(nblocks), dim3(256), lds, h->stream, h->d_db, d_q, nq, h->d_nvalid, thr_low, h->d_partials);
        }
        hipLaunchKernelGGL(k_db_reduce, dim3((nq + 255) / 256), dim3(256), 0, h->stream, h->d_partials, nparts, nq, h->d_ids, d_best,
                           d_max, d_cnt);
    }
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

int myslam_lcddb_query_batch(myslam_lcddb* h, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low,
                             uint64_t* d_best_id, float* d_max_score, int32_t* d_cnt) {
    if (!h || !d_q || !cur_ids || nq < 1 || nq > 65536 || !d_best_id || !d_max_score || !d_cnt) return MYSLAM_ERR_INVALID;
    return db_query(h, d_q, cur_ids, nq, thr_low, d_best_id, d_max_score, d_cnt);
}

int myslam_lcddb_update_query_limits(myslam_lcddb* h, const uint64_t* cur_ids, int nq) {
    if (!h || !cur_ids || nq < 1) return MYSLAM_ERR_INVALID;
    if (!h->graphRows || nq > h->graphQueries || nq > h->nvalidCap) return MYSLAM_ERR_INVALID;      // no recorded query to feed
    if (h->graphStale || h->n > h->graphRows) return MYSLAM_ERR_CAPACITY;      // the database moved or outgrew the recorded launch: record the step again
    MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));               // no replay is reading the pinned buffer
    for (int i = 0; i < nq; i++) h->h_nvalid[i] = h->n_valid(cur_ids[i]);
    h->nvFresh = false;                                              // the next replay rewrites d_nvalid behind the eager path's back
    return MYSLAM_OK;
}

int myslam_lcddb_query_batch_sharded(myslam_lcddb* h, const float* d_q, const uint64_t* cur_ids, int nq, float thr_low,
                                     myslam_lcd_candidate* d_cand) {
    if (!h || !d_q || !cur_ids || nq < 1 || nq > 65536 || !d_cand) return MYSLAM_ERR_INVALID;
    if (nq > h->shardCap) {
        MYSLAM_HIP_CHECK(hipStreamSynchronize(h->stream));
        void* old[] = {h->d_bestS, h->d_maxS, h->d_cntS};
        for (void* p : old) if (p) (void)hipFree(p);
        h->d_bestS = nullptr; h->d_maxS = nullptr; h->d_cntS = nullptr; h->shardCap = 0;
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_bestS, sizeof(uint64_t) * nq));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_maxS, sizeof(float) * nq));
        MYSLAM_HIP_CHECK(hipMalloc((void**)&h->d_cntS, sizeof(int32_t) * nq));
        h->shardCap = nq;
    }
    int rc = db_query(h, d_q, cur_ids, nq, thr_low, h->d_bestS, h->d_maxS, h->d_cntS);
    if (rc) return rc;
    hipLaunchKernelGGL(k_db_pack_candidates, dim3((nq + 255) / 256), dim3(256), 0, h->stream, h->d_bestS, h->d_maxS, h->d_cntS, h->d_nvalid,
                       h->n, nq, d_cand);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
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
    int rc = db_query(h, h->d_q1, &cur_id, 1, thr_low, h->d_best1, h->d_max1, h->d_cnt1);
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
