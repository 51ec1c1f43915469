// sharded_db_rccl.cpp — the multi-GPU loop-database exchange with NO Python in the loop: plain C++, the HIP runtime, librccl and the C ABI.
// One process per GPU (WORLD_SIZE / RANK / LOCAL_RANK from the environment, as torch.distributed.run or mpirun set them); the reference's
// side would be LoopClosing::DetectLoop, /root/reference/src/loopclosing.cpp:124-161, over a std::map that no longer fits one device.
//   1. every rank holds the contiguous id range [r n/W, (r+1) n/W) of the key-frame descriptors in a myslam_lcddb and P query descriptors
//   2. ncclAllGather of the queries (W x P x 1064 f32)
//   3. myslam_lcddb_query_batch_sharded: every shard scores every query -> one 16-byte myslam_lcd_candidate per query (device memory)
//   4. ncclAllGather of the raw candidate bytes (W x NQ x 16)
//   5. myslam_lcd_merge_candidates_device -> (best id, max score, count) of ONE ascending scan over the whole database
// Rank 0 also holds the whole database in one handle and checks 5. against its single scan.
// After the check the same exchange is repeated `reps` times with HIP events around its four stages: every rank prints one line
// "rank r: allgather_queries_ms … shard_scan_ms … allgather_candidates_ms … merge_ms …" — the per-step collective / scan cost a multi-GPU
// run of bench.py reports under the same names (DESIGN.md section 4 holds the predicted values for N = 2 / 4 / 8).
// Built by build.py into bin/sharded_db_rccl (g++; -D__HIP_PLATFORM_AMD__ is what the HIP runtime headers need from a plain host compiler).
//   usage: sharded_db_rccl <db.f32> <queries.f32> <cur_ids.u64> <n_db> <nq> [reps = 20] [grow_steps = 0]      (nq divisible by WORLD_SIZE)
//   grow_steps > 0: after the static exchange, the GROWING database of round 6 (see below): appends and queries interleaved for that many steps, ownership by
//   arrival over WORLD_SIZE x $MYSLAM_LOCAL_SHARDS shards, every answer checked against one map on rank 0
//   the ncclUniqueId travels through the file $MYSLAM_NCCL_ID_FILE (rank 0 writes it; a launcher with MPI would broadcast it instead)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "myslam_hip.h"

#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", g_rank, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NCCLOK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", g_rank, #x, ncclGetErrorString(r_)); return 3; } } while (0)
#define MYOK(x) do { int c_ = (x); if (c_ != MYSLAM_OK) { fprintf(stderr, "rank %d: %s -> %d\n", g_rank, #x, c_); return 4; } } while (0)
static int g_rank = 0;

template <class T>
static bool read_all(const char* path, std::vector<T>& v, size_t n) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    v.resize(n);
    const size_t got = fread(v.data(), sizeof(T), n, f);
    fclose(f);
    return got == n;
}

static int env_int(const char* k, int def) { const char* s = getenv(k); return s ? atoi(s) : def; }

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s db.f32 queries.f32 cur_ids.u64 n_db nq\n", argv[0]); return 1; }
    const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local = env_int("LOCAL_RANK", rank);
    g_rank = rank;
    const int n_db = atoi(argv[4]), nq = atoi(argv[5]), reps = argc > 6 ? atoi(argv[6]) : 20;
    if (world < 1 || rank < 0 || rank >= world || nq % world != 0 || n_db < world) { fprintf(stderr, "bad sizes\n"); return 1; }
    const int P = nq / world, D = MYSLAM_LCD_DIM;
    std::vector<float> db, q; std::vector<uint64_t> cur;
    if (!read_all(argv[1], db, (size_t)n_db * D) || !read_all(argv[2], q, (size_t)nq * D) || !read_all(argv[3], cur, (size_t)nq)) { fprintf(stderr, "cannot read the inputs\n"); return 1; }
    int ndev = 0;
    HIPOK(hipGetDeviceCount(&ndev));
    if (local >= ndev) { fprintf(stderr, "rank %d: LOCAL_RANK %d but %d device(s)\n", rank, local, ndev); return 1; }
    HIPOK(hipSetDevice(local));

    // ---- communicator -----------------------------------------------------------------------------------------------------------
    ncclUniqueId id;
    const char* idfile = getenv("MYSLAM_NCCL_ID_FILE");
    if (world > 1 && !idfile) { fprintf(stderr, "MYSLAM_NCCL_ID_FILE is not set\n"); return 1; }
    if (rank == 0) {
        NCCLOK(ncclGetUniqueId(&id));
        if (world > 1) {
            const std::string tmp = std::string(idfile) + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&id, sizeof(id), 1, f) != 1) return 1;
            fclose(f);
            if (rename(tmp.c_str(), idfile) != 0) return 1;
        }
    } else {
        FILE* f = nullptr;
        for (int t = 0; t < 600 && !(f = fopen(idfile, "rb")); t++) usleep(100000);
        if (!f || fread(&id, sizeof(id), 1, f) != 1) { fprintf(stderr, "rank %d: no unique id\n", rank); return 1; }
        fclose(f);
    }
    ncclComm_t comm;
    NCCLOK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t s;
    HIPOK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

    // ---- this rank's shard: ids lo .. hi-1 (key-frame id = row index) ----------------------------------------------------------------
    const int lo = (int)((long long)n_db * rank / world), hi = (int)((long long)n_db * (rank + 1) / world);
    myslam_lcddb* shard = nullptr;
    MYOK(myslam_lcddb_create(&shard, 16));                          // grows as the reference's std::map does
    MYOK(myslam_lcddb_set_stream(shard, s));
    for (int i = lo; i < hi; i++) MYOK(myslam_lcddb_append(shard, (uint64_t)i, db.data() + (size_t)i * D));     // LoopClosing::AddToDatabase, one key-frame at a time
    if (myslam_lcddb_size(shard) != hi - lo) return 5;

    float *d_myq = nullptr, *d_allq = nullptr; myslam_lcd_candidate *d_cand = nullptr, *d_gath = nullptr;
    uint64_t* d_best = nullptr; float* d_max = nullptr; int32_t* d_cnt = nullptr;
    HIPOK(hipMalloc((void**)&d_myq, sizeof(float) * P * D)); HIPOK(hipMalloc((void**)&d_allq, sizeof(float) * nq * D));
    HIPOK(hipMalloc((void**)&d_cand, sizeof(myslam_lcd_candidate) * nq)); HIPOK(hipMalloc((void**)&d_gath, sizeof(myslam_lcd_candidate) * nq * world));
    HIPOK(hipMalloc((void**)&d_best, 8 * nq)); HIPOK(hipMalloc((void**)&d_max, 4 * nq)); HIPOK(hipMalloc((void**)&d_cnt, 4 * nq));
    HIPOK(hipMemcpyAsync(d_myq, q.data() + (size_t)rank * P * D, sizeof(float) * P * D, hipMemcpyHostToDevice, s));      // this rank's own P queries

    // ---- the exchange: two all-gathers around the two C-ABI calls, everything on one stream, no host synchronisation in between ---------
    const float thr_low = 0.92f;
    NCCLOK(ncclAllGather(d_myq, d_allq, (size_t)P * D, ncclFloat, comm, s));
    MYOK(myslam_lcddb_query_batch_sharded(shard, d_allq, cur.data(), nq, thr_low, d_cand));
    NCCLOK(ncclAllGather(d_cand, d_gath, sizeof(myslam_lcd_candidate) * (size_t)nq, ncclChar, comm, s));
    MYOK(myslam_lcd_merge_candidates_device(d_gath, world, nq, d_best, d_max, d_cnt, s));
    std::vector<uint64_t> best(nq); std::vector<float> mx(nq); std::vector<int32_t> cnt(nq);
    HIPOK(hipMemcpyAsync(best.data(), d_best, 8 * nq, hipMemcpyDeviceToHost, s));
    HIPOK(hipMemcpyAsync(mx.data(), d_max, 4 * nq, hipMemcpyDeviceToHost, s));
    HIPOK(hipMemcpyAsync(cnt.data(), d_cnt, 4 * nq, hipMemcpyDeviceToHost, s));
    HIPOK(hipStreamSynchronize(s));

    // ---- rank 0: the same queries against ONE scan of the whole database -------------------------------------------------------------
    int bad = 0;
    if (rank == 0) {
        myslam_lcddb* whole = nullptr;
        MYOK(myslam_lcddb_create(&whole, n_db));
        MYOK(myslam_lcddb_set_stream(whole, s));
        std::vector<uint64_t> ids(n_db);
        for (int i = 0; i < n_db; i++) ids[i] = (uint64_t)i;
        float* d_db = nullptr;
        HIPOK(hipMalloc((void**)&d_db, sizeof(float) * (size_t)n_db * D));
        HIPOK(hipMemcpy(d_db, db.data(), sizeof(float) * (size_t)n_db * D, hipMemcpyHostToDevice));
        MYOK(myslam_lcddb_append_batch(whole, ids.data(), d_db, n_db));
        uint64_t* d_b1 = nullptr; float* d_m1 = nullptr; int32_t* d_c1 = nullptr;
        HIPOK(hipMalloc((void**)&d_b1, 8 * nq)); HIPOK(hipMalloc((void**)&d_m1, 4 * nq)); HIPOK(hipMalloc((void**)&d_c1, 4 * nq));
        MYOK(myslam_lcddb_query_batch(whole, d_allq, cur.data(), nq, thr_low, d_b1, d_m1, d_c1));
        std::vector<uint64_t> b1(nq); std::vector<float> m1(nq); std::vector<int32_t> c1(nq);
        HIPOK(hipMemcpyAsync(b1.data(), d_b1, 8 * nq, hipMemcpyDeviceToHost, s));
        HIPOK(hipMemcpyAsync(m1.data(), d_m1, 4 * nq, hipMemcpyDeviceToHost, s));
        HIPOK(hipMemcpyAsync(c1.data(), d_c1, 4 * nq, hipMemcpyDeviceToHost, s));
        HIPOK(hipStreamSynchronize(s));
        int loops = 0, breaks = 0;
        for (int i = 0; i < nq; i++) {
            if (best[i] != b1[i] || cnt[i] != c1[i] || memcmp(&mx[i], &m1[i], 4) != 0) {
                if (bad++ < 5) fprintf(stderr, "query %d: sharded (%llu, %.9g, %d) vs one scan (%llu, %.9g, %d)\n", i, (unsigned long long)best[i], mx[i], cnt[i],
                                       (unsigned long long)b1[i], m1[i], c1[i]);
            }
            loops += (mx[i] >= 0.94f && cnt[i] <= 3);               // DetectLoop's decision, loopclosing.cpp:147
            breaks += cur[i] < (uint64_t)n_db + 19;
        }
        // the host form of the merge on the gathered bytes (a CPU-side consumer needs no device)
        std::vector<myslam_lcd_candidate> hg((size_t)nq * world);
        HIPOK(hipMemcpy(hg.data(), d_gath, sizeof(myslam_lcd_candidate) * hg.size(), hipMemcpyDeviceToHost));
        std::vector<uint64_t> b2(nq); std::vector<float> m2(nq); std::vector<int32_t> c2(nq);
        MYOK(myslam_lcd_merge_candidates(hg.data(), world, nq, b2.data(), m2.data(), c2.data()));
        for (int i = 0; i < nq; i++) bad += (b2[i] != best[i] || c2[i] != cnt[i] || memcmp(&m2[i], &mx[i], 4) != 0);
        printf("%s ranks=%d n_db=%d queries=%d (P=%d per rank) accepted_loops=%d queries_hitting_the_cut_off=%d mismatches=%d\n",
               bad ? "SHARDED DB RCCL FAILED" : "SHARDED DB RCCL OK", world, n_db, nq, P, loops, breaks, bad);
        (void)myslam_lcddb_destroy(whole);
        (void)hipFree(d_db); (void)hipFree(d_b1); (void)hipFree(d_m1); (void)hipFree(d_c1);
    }
    // ---- the exchange again, timed stage by stage (HIP events on the stream all four stages run on) -----------------------------------
    if (reps > 0 && !bad) {
        hipEvent_t ev[5];
        for (auto& e : ev) HIPOK(hipEventCreate(&e));
        double acc[4] = {0, 0, 0, 0};
        for (int it = -2; it < reps; it++) {                        // two untimed warm-up rounds
            HIPOK(hipEventRecord(ev[0], s));
            NCCLOK(ncclAllGather(d_myq, d_allq, (size_t)P * D, ncclFloat, comm, s));
            HIPOK(hipEventRecord(ev[1], s));
            MYOK(myslam_lcddb_query_batch_sharded(shard, d_allq, cur.data(), nq, thr_low, d_cand));
            HIPOK(hipEventRecord(ev[2], s));
            NCCLOK(ncclAllGather(d_cand, d_gath, sizeof(myslam_lcd_candidate) * (size_t)nq, ncclChar, comm, s));
            HIPOK(hipEventRecord(ev[3], s));
            MYOK(myslam_lcd_merge_candidates_device(d_gath, world, nq, d_best, d_max, d_cnt, s));
            HIPOK(hipEventRecord(ev[4], s));
            HIPOK(hipStreamSynchronize(s));
            if (it < 0) continue;
            for (int k = 0; k < 4; k++) { float ms = 0; HIPOK(hipEventElapsedTime(&ms, ev[k], ev[k + 1])); acc[k] += ms; }
        }
        printf("rank %d: allgather_queries_ms %.4f (%zu B per rank) shard_scan_ms %.4f (%d rows x %d queries) allgather_candidates_ms %.4f (%zu B per rank) merge_ms %.4f  [mean of %d]\n",
               rank, acc[0] / reps, sizeof(float) * (size_t)P * D, acc[1] / reps, hi - lo, nq, acc[2] / reps, sizeof(myslam_lcd_candidate) * (size_t)nq, acc[3] / reps, reps);
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    // ---- a database that GROWS under the ranks (round 6) ------------------------------------------------------------------------------------------
    // The reference appends per key-frame (LoopClosing::AddToDatabase, src/loopclosing.cpp:651-659).  `grow_steps` steps; in each, every rank brings K new
    // key-frames (id, descriptor), all ranks all-gather them, every SHARD scores every query (myslam_lcddb_query_batch_owned: 32-byte records, the shard's
    // ids interleave with the others'), the records are all-gathered and merged (myslam_lcd_merge_owned_candidates_device), and then the step's key-frames
    // are appended: the j-th in id order goes to shard (appended so far + j) mod S — every rank computes the same owners from the same gathered list, no
    // further collective, row counts never more than 1 apart.  S = WORLD_SIZE x MYSLAM_LOCAL_SHARDS (L shards per process: on a one-GPU box L > 1 still
    // interleaves ownership for real).  Rank 0 keeps ONE map with everything in it and checks every merged answer against its single scan, bit for bit.
    const int grow_steps = argc > 7 ? atoi(argv[7]) : 0;
    if (grow_steps > 0 && !bad) {
        const int L = std::max(1, env_int("MYSLAM_LOCAL_SHARDS", 1)), S = world * L, K = 2, NQg = world * K;
        std::vector<myslam_lcddb*> gs(L, nullptr);
        for (auto& g : gs) { MYOK(myslam_lcddb_create(&g, 16)); MYOK(myslam_lcddb_set_stream(g, s)); }
        myslam_lcddb* one = nullptr;
        if (rank == 0) { MYOK(myslam_lcddb_create(&one, 16)); MYOK(myslam_lcddb_set_stream(one, s)); }
        float *d_new = nullptr, *d_alln = nullptr; uint64_t *d_ids = nullptr, *d_allids = nullptr;
        myslam_lcd_owned_candidate *d_rec = nullptr, *d_grec = nullptr;
        uint64_t *d_gb = nullptr, *d_rb = nullptr; float *d_gm = nullptr, *d_rm = nullptr; int32_t *d_gc = nullptr, *d_rc = nullptr;
        HIPOK(hipMalloc((void**)&d_new, sizeof(float) * K * D)); HIPOK(hipMalloc((void**)&d_alln, sizeof(float) * NQg * D));
        HIPOK(hipMalloc((void**)&d_ids, 8 * K)); HIPOK(hipMalloc((void**)&d_allids, 8 * NQg));
        HIPOK(hipMalloc((void**)&d_rec, sizeof(myslam_lcd_owned_candidate) * L * NQg)); HIPOK(hipMalloc((void**)&d_grec, sizeof(myslam_lcd_owned_candidate) * S * NQg));
        HIPOK(hipMalloc((void**)&d_gb, 8 * NQg)); HIPOK(hipMalloc((void**)&d_gm, 4 * NQg)); HIPOK(hipMalloc((void**)&d_gc, 4 * NQg));
        HIPOK(hipMalloc((void**)&d_rb, 8 * NQg)); HIPOK(hipMalloc((void**)&d_rm, 4 * NQg)); HIPOK(hipMalloc((void**)&d_rc, 4 * NQg));
        long long total = 0; int gbad = 0, gq = 0, gloop = 0, spread = 0;
        std::vector<uint64_t> hid(K), allid(NQg), gb(NQg), rb(NQg); std::vector<float> gm(NQg), rm(NQg); std::vector<int32_t> gc(NQg), rc(NQg);
        for (int st = 0; st < grow_steps && !gbad; st++) {
            // this rank's K new key-frames: rows of the database file (they repeat when the file is exhausted: equal rows in different shards, the tie rule)
            for (int j = 0; j < K; j++) {
                hid[j] = (uint64_t)st * (uint64_t)(NQg + 3) + (uint64_t)((long long)(rank * K + j) * (NQg - 1) % NQg);      // ids grow step by step; within a step the ranks' ids interleave (x -> -x mod NQg: distinct)
                const size_t row = ((size_t)st * NQg + (size_t)rank * K + j) % (size_t)std::min(n_db, 97);
                HIPOK(hipMemcpyAsync(d_new + (size_t)j * D, db.data() + row * D, sizeof(float) * D, hipMemcpyHostToDevice, s));
            }
            HIPOK(hipMemcpyAsync(d_ids, hid.data(), 8 * K, hipMemcpyHostToDevice, s));
            NCCLOK(ncclAllGather(d_new, d_alln, (size_t)K * D, ncclFloat, comm, s));
            NCCLOK(ncclAllGather(d_ids, d_allids, (size_t)K, ncclUint64, comm, s));
            HIPOK(hipMemcpyAsync(allid.data(), d_allids, 8 * NQg, hipMemcpyDeviceToHost, s));
            HIPOK(hipStreamSynchronize(s));                                     // the row ranges of a query are computed on the host from its id
            for (int l = 0; l < L; l++) MYOK(myslam_lcddb_query_batch_owned(gs[l], d_alln, allid.data(), NQg, thr_low, d_rec + (size_t)l * NQg));
            NCCLOK(ncclAllGather(d_rec, d_grec, sizeof(myslam_lcd_owned_candidate) * (size_t)L * NQg, ncclChar, comm, s));
            MYOK(myslam_lcd_merge_owned_candidates_device(d_grec, S, NQg, d_gb, d_gm, d_gc, s));
            if (rank == 0) {
                if (myslam_lcddb_size(one) > 0) {
                    MYOK(myslam_lcddb_query_batch(one, d_alln, allid.data(), NQg, thr_low, d_rb, d_rm, d_rc));
                    HIPOK(hipMemcpyAsync(gb.data(), d_gb, 8 * NQg, hipMemcpyDeviceToHost, s)); HIPOK(hipMemcpyAsync(gm.data(), d_gm, 4 * NQg, hipMemcpyDeviceToHost, s));
                    HIPOK(hipMemcpyAsync(gc.data(), d_gc, 4 * NQg, hipMemcpyDeviceToHost, s)); HIPOK(hipMemcpyAsync(rb.data(), d_rb, 8 * NQg, hipMemcpyDeviceToHost, s));
                    HIPOK(hipMemcpyAsync(rm.data(), d_rm, 4 * NQg, hipMemcpyDeviceToHost, s)); HIPOK(hipMemcpyAsync(rc.data(), d_rc, 4 * NQg, hipMemcpyDeviceToHost, s));
                    HIPOK(hipStreamSynchronize(s));
                    for (int i = 0; i < NQg; i++) {
                        gq++;
                        if (gb[i] != rb[i] || gc[i] != rc[i] || memcmp(&gm[i], &rm[i], 4) != 0) {
                            if (gbad++ < 5) fprintf(stderr, "growing step %d query %d (id %llu): shards (%llu, %.9g, %d) vs one map (%llu, %.9g, %d)\n", st, i, (unsigned long long)allid[i],
                                                    (unsigned long long)gb[i], gm[i], gc[i], (unsigned long long)rb[i], rm[i], rc[i]);
                        }
                        gloop += (gm[i] >= 0.94f && gc[i] <= 3);
                    }
                }
            }
            // AddToDatabase after DetectLoop: the step's key-frames in id order, the j-th to shard (total + j) mod S
            std::vector<int> order(NQg);
            for (int i = 0; i < NQg; i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](int a, int b) { return allid[a] < allid[b]; });
            for (int j = 0; j < NQg; j++) {
                const int slot = order[j], owner = (int)((total + j) % S);
                if (j && allid[order[j]] == allid[order[j - 1]]) { fprintf(stderr, "duplicate key-frame id in one step\n"); return 7; }
                if (owner / L == rank) MYOK(myslam_lcddb_append_batch(gs[owner % L], &allid[slot], d_alln + (size_t)slot * D, 1));
                if (rank == 0) MYOK(myslam_lcddb_append_batch(one, &allid[slot], d_alln + (size_t)slot * D, 1));
            }
            total += NQg;
            int lo_rows = INT32_MAX, hi_rows = 0;
            for (auto g : gs) { lo_rows = std::min(lo_rows, myslam_lcddb_size(g)); hi_rows = std::max(hi_rows, myslam_lcddb_size(g)); }
            spread = std::max(spread, hi_rows - lo_rows);
        }
        // every shard of the job within one row of every other: the counts travel once at the end (the owners were computed identically everywhere)
        std::vector<int32_t> cnts(L), allc((size_t)S); int32_t *d_c = nullptr, *d_ac = nullptr;
        for (int l = 0; l < L; l++) cnts[l] = myslam_lcddb_size(gs[l]);
        HIPOK(hipMalloc((void**)&d_c, 4 * L)); HIPOK(hipMalloc((void**)&d_ac, 4 * S));
        HIPOK(hipMemcpyAsync(d_c, cnts.data(), 4 * L, hipMemcpyHostToDevice, s));
        NCCLOK(ncclAllGather(d_c, d_ac, (size_t)L, ncclInt32, comm, s));
        HIPOK(hipMemcpyAsync(allc.data(), d_ac, 4 * S, hipMemcpyDeviceToHost, s)); HIPOK(hipStreamSynchronize(s));
        const int cmin = *std::min_element(allc.begin(), allc.end()), cmax = *std::max_element(allc.begin(), allc.end());
        long long csum = 0; for (int v : allc) csum += v;
        if (cmax - cmin > 1 || csum != total || spread > 1) { fprintf(stderr, "rank %d: unbalanced shards (%d .. %d rows, sum %lld of %lld)\n", rank, cmin, cmax, csum, total); gbad++; }
        if (rank == 0)
            printf("%s shards=%d (ranks=%d x local=%d) steps=%d key_frames=%lld queries_checked=%d accepted_loops=%d rows_per_shard=%d..%d mismatches=%d\n",
                   gbad ? "GROWING SHARDED DB FAILED" : "GROWING SHARDED DB OK", S, world, L, grow_steps, total, gq, gloop, cmin, cmax, gbad);
        bad += gbad;
        for (auto g : gs) (void)myslam_lcddb_destroy(g);
        if (one) (void)myslam_lcddb_destroy(one);
        void* fg[] = {d_new, d_alln, d_ids, d_allids, d_rec, d_grec, d_gb, d_gm, d_gc, d_rb, d_rm, d_rc, d_c, d_ac};
        for (void* p : fg) (void)hipFree(p);
    }
    (void)myslam_lcddb_destroy(shard);
    void* fr[] = {d_myq, d_allq, d_cand, d_gath, d_best, d_max, d_cnt};
    for (void* p : fr) (void)hipFree(p);
    NCCLOK(ncclCommDestroy(comm));
    (void)hipStreamDestroy(s);
    return bad ? 6 : 0;
}
