// bench_oracle.cpp — TEST INFRASTRUCTURE (CPU baseline leg of bench.py only): runs the oracle's per-frame pipeline over a
// batch of stereo pairs with a std::thread pool, one frame per task, and reports the wall time plus per-frame stage times.
// The stages and their order are the GPU workload's: ORB DetectAndCompute on left and right, Hamming match + stereo
// triangulation, CALC descriptor of the left image + loop-database scan, local-BA block build (and, optionally, the
// OptimizeActiveMap solve stage).  Protocol of SURVEY.md §8(d): the first `n_warmup` frames are processed untimed, the wall
// clock covers the remaining frames, per-stage medians are taken by the caller from `stage_seconds`.
#include <malloc.h>

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "oracle.h"

namespace {
struct BaWindows {        // nwin windows with common capacities; frame i uses window i % nwin
    const double* poses; const double* points; const int32_t* ep; const int32_t* el; const double* obs; const uint8_t* fixed;
    const int32_t* sizes; int nwin, maxP, maxL, maxE;
};
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// n_tasks frames are processed (task i uses stereo pair i % n_pairs), the first n_warmup of them untimed.
extern "C" int orc_bench_frames(const uint8_t* frames /*n_pairs x 2 x rows x cols*/, int n_pairs, int n_tasks, int rows, int cols, int nfeatures,
                                double fx, double fy, double cx, double cy, double baseline,
                                const float* weights, size_t nweights, const float* db, const uint64_t* ids, int n_db,
                                const double* ba_poses, const double* ba_points, const int32_t* ep, const int32_t* el, const double* obs,
                                const uint8_t* fixed, const int32_t* ba_sizes /*nwin x 3*/, int nwin, int maxP, int maxL, int maxE,
                                int stages /*1 orb+match+tri, 2 +lcd, 3 +ba build, 4 +ba solve*/, int threads, int n_warmup,
                                double* seconds, double* stage_seconds /*n_tasks x 5 or NULL*/) {
    if (!frames || n_pairs < 1 || n_tasks < 1 || threads < 1 || !seconds || n_warmup < 0 || n_warmup >= n_tasks) return -1;
    // every frame allocates a few MB of pyramid / blurred planes: above glibc's mmap threshold each of them is an mmap + page faults +
    // munmap under the process-wide mm lock, which serialises a frame-parallel run (measured: 18x longer per frame on 256 threads).
    // Keep those blocks in the per-thread arenas instead — a frame's planes are then reused by the thread's next frame.
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    if (stages >= 3 && (nwin < 1 || !ba_sizes)) return -1;
    const BaWindows W{ba_poses, ba_points, ep, el, obs, fixed, ba_sizes, nwin, maxP, maxL, maxE};
    std::atomic<int> next{0}, fail{0};
    int limit = 0;
    const size_t img = (size_t)rows * cols;
    auto worker = [&]() {
        orc_orb_params p{nfeatures, 1.2f, 8, 20, 7};
        const int cap = 2 * nfeatures + 64;
        const int cE = maxE > 0 ? maxE : 1, cP = maxP > 0 ? maxP : 1, cL = maxL > 0 ? maxL : 1;
        std::vector<orc_keypoint> kl(cap), kr(cap);
        std::vector<uint8_t> dl((size_t)cap * 32), dr((size_t)cap * 32), ok(cap), tmp(img), out(cE);
        std::vector<int32_t> idx(cap), dist(cap);
        std::vector<float> xl(cap), yl(cap), xr(cap), yr(cap), net_in(120 * 160), descr(1064);
        std::vector<double> xyz((size_t)cap * 3), Hpp((size_t)cP * 36), Hll((size_t)cL * 9), Hpl((size_t)cE * 18), bp((size_t)cP * 6),
            bl((size_t)cL * 3), chi((size_t)cE), sp, sx;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= limit) break;
            double* st = stage_seconds ? stage_seconds + (size_t)i * 5 : nullptr;
            const uint8_t* L = frames + (size_t)(i % n_pairs) * 2 * img; const uint8_t* R = L + img;
            int nl = 0, nr = 0;
            double t0 = now_s();
            if (orc_detect_and_compute(&p, L, rows, cols, cols, nullptr, 0, kl.data(), dl.data(), cap, &nl) ||
                orc_detect_and_compute(&p, R, rows, cols, cols, nullptr, 0, kr.data(), dr.data(), cap, &nr)) { fail++; continue; }
            double t1 = now_s();
            if (st) st[0] = t1 - t0;
            if (nl > 0 && nr > 0) {
                orc_hamming_match(dl.data(), nl, dr.data(), nr, idx.data(), dist.data());
                for (int k = 0; k < nl; k++) { xl[k] = kl[k].x; yl[k] = kl[k].y; xr[k] = kr[idx[k]].x; yr[k] = kr[idx[k]].y; }
                orc_triangulate_stereo(xl.data(), yl.data(), xr.data(), yr.data(), nl, fx, fy, cx, cy, baseline, xyz.data(), ok.data());
            }
            t0 = now_s();
            if (st) st[1] = t0 - t1;
            if (stages >= 2) {
                std::copy(L, L + img, tmp.begin());
                orc_calc_preproc(tmp.data(), rows, cols, cols, 1, net_in.data());
                if (orc_calc_forward(weights, nweights, net_in.data(), descr.data())) { fail++; continue; }
                uint64_t best; float mx; int cnt;
                orc_lcddb_query(db, ids, n_db, descr.data(), (uint64_t)n_db + 20, 0.92f, &best, &mx, &cnt);
            }
            t1 = now_s();
            if (st) st[2] = t1 - t0;
            const int w = W.nwin > 0 ? i % W.nwin : 0;
            const int nP = stages >= 3 ? W.sizes[3 * w] : 0, nL = stages >= 3 ? W.sizes[3 * w + 1] : 0, nE = stages >= 3 ? W.sizes[3 * w + 2] : 0;
            const double* wp = W.poses + (size_t)w * W.maxP * 7; const double* wx = W.points + (size_t)w * W.maxL * 3;
            const int32_t* wep = W.ep + (size_t)w * W.maxE; const int32_t* wel = W.el + (size_t)w * W.maxE;
            const double* wo = W.obs + (size_t)w * W.maxE * 2; const uint8_t* wf = W.fixed + (size_t)w * W.maxL;
            if (stages >= 3)
                orc_ba_build(wp, nP, wx, nL, wep, wel, wo, nE, wf, fx, fy, cx, cy, 5.991, Hpp.data(), Hll.data(), Hpl.data(), bp.data(), bl.data(),
                             chi.data());
            t0 = now_s();
            if (st) st[3] = t0 - t1;
            if (stages >= 4) {
                sp.assign(wp, wp + (size_t)nP * 7); sx.assign(wx, wx + (size_t)nL * 3);
                int rd, no;
                orc_ba_optimize_active_map(sp.data(), nP, sx.data(), nL, wep, wel, wo, nE, wf, fx, fy, cx, cy, 5.991, 5.991, 5, 10, chi.data(),
                                           out.data(), &rd, &no);
            }
            if (st) st[4] = now_s() - t0;
        }
    };
    auto run_pool = [&](int from, int upto) {
        next.store(from); limit = upto;
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
    };
    if (n_warmup > 0) run_pool(0, n_warmup);       // frames [0, n_warmup): untimed
    const auto t0 = std::chrono::steady_clock::now();
    run_pool(n_warmup, n_tasks);                   // tasks [n_warmup, n_tasks)
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return fail.load() ? -2 : 0;
}

// How many threads does this host really run at the same time?  Containers often see every CPU of the machine in their affinity mask
// while a CPU-time quota (or the neighbours) grants them a fraction: `threads` workers each do the same fixed amount of register-only
// integer work; the result is threads x (one worker's time alone) / (wall time of all of them together).
extern "C" double orc_cpu_capacity(int threads, int work_ms) {
    if (threads < 1) return 0;
    auto spin = [](uint64_t iters) { uint64_t x = 88172645463325252ull; for (uint64_t i = 0; i < iters; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; } return x; };
    // calibrate: iterations for ~work_ms on one thread
    uint64_t iters = 1 << 22; double t1 = 0; volatile uint64_t sink = 0;
    for (int rep = 0; rep < 8; rep++) {
        const double t0 = now_s(); sink = sink + spin(iters); t1 = now_s() - t0;
        if (t1 * 1e3 >= 0.5 * work_ms) break;
        iters *= 2;
    }
    const double t0 = now_s();
    std::vector<std::thread> pool;
    std::atomic<uint64_t> acc{0};
    for (int t = 0; t < threads; t++) pool.emplace_back([&]() { acc += spin(iters); });
    for (auto& th : pool) th.join();
    const double tw = now_s() - t0;
    return tw > 0 ? threads * t1 / tw : 0;
}
