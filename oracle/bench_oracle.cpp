// bench_oracle.cpp — TEST INFRASTRUCTURE (CPU baseline leg of bench.py only): runs the oracle's per-frame pipeline over a
// batch of stereo pairs with a std::thread pool, one frame per task, and reports the wall time.  The stages and their order are
// the GPU workload's: ORB DetectAndCompute on left and right, Hamming match, stereo triangulation, CALC descriptor of the
// left image, loop-database scan, local-BA block build (and, optionally, the OptimizeActiveMap solve stage).
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "oracle.h"

extern "C" int orc_bench_frames(const uint8_t* frames /*n_pairs x 2 x rows x cols*/, int n_pairs, int rows, int cols, int nfeatures,
                                double fx, double fy, double cx, double cy, double baseline,
                                const float* weights, size_t nweights, const float* db, const uint64_t* ids, int n_db,
                                const double* ba_poses, int nposes, const double* ba_points, int npts, const int32_t* ep,
                                const int32_t* el, const double* obs, int nedges, const uint8_t* fixed,
                                int stages /*1 orb+match+tri, 2 +lcd, 3 +ba build, 4 +ba solve*/, int threads, double* seconds) {
    if (!frames || n_pairs < 1 || threads < 1 || !seconds) return -1;
    std::atomic<int> next{0}, fail{0};
    const size_t img = (size_t)rows * cols;
    auto worker = [&]() {
        orc_orb_params p{nfeatures, 1.2f, 8, 20, 7};
        const int cap = 2 * nfeatures + 64;
        std::vector<orc_keypoint> kl(cap), kr(cap);
        std::vector<uint8_t> dl((size_t)cap * 32), dr((size_t)cap * 32), ok(cap), tmp(img), out(nedges > 0 ? nedges : 1);
        std::vector<int32_t> idx(cap), dist(cap);
        std::vector<float> xl(cap), yl(cap), xr(cap), yr(cap), net_in(120 * 160), descr(1064);
        std::vector<double> xyz((size_t)cap * 3), Hpp((size_t)nposes * 36), Hll((size_t)npts * 9), Hpl((size_t)(nedges > 0 ? nedges : 1) * 18),
            bp((size_t)nposes * 6), bl((size_t)npts * 3), chi((size_t)(nedges > 0 ? nedges : 1)), sp, sx;
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n_pairs) break;
            const uint8_t* L = frames + (size_t)i * 2 * img; const uint8_t* R = L + img;
            int nl = 0, nr = 0;
            if (orc_detect_and_compute(&p, L, rows, cols, cols, nullptr, 0, kl.data(), dl.data(), cap, &nl) ||
                orc_detect_and_compute(&p, R, rows, cols, cols, nullptr, 0, kr.data(), dr.data(), cap, &nr)) { fail++; continue; }
            if (nl > 0 && nr > 0) {
                orc_hamming_match(dl.data(), nl, dr.data(), nr, idx.data(), dist.data());
                for (int k = 0; k < nl; k++) { xl[k] = kl[k].x; yl[k] = kl[k].y; xr[k] = kr[idx[k]].x; yr[k] = kr[idx[k]].y; }
                orc_triangulate_stereo(xl.data(), yl.data(), xr.data(), yr.data(), nl, fx, fy, cx, cy, baseline, xyz.data(), ok.data());
            }
            if (stages >= 2) {
                std::copy(L, L + img, tmp.begin());
                orc_calc_preproc(tmp.data(), rows, cols, cols, 1, net_in.data());
                if (orc_calc_forward(weights, nweights, net_in.data(), descr.data())) { fail++; continue; }
                uint64_t best; float mx; int cnt;
                orc_lcddb_query(db, ids, n_db, descr.data(), (uint64_t)n_db + 20, 0.92f, &best, &mx, &cnt);
            }
            if (stages >= 3)
                orc_ba_build(ba_poses, nposes, ba_points, npts, ep, el, obs, nedges, fixed, fx, fy, cx, cy, 5.991, Hpp.data(), Hll.data(),
                             Hpl.data(), bp.data(), bl.data(), chi.data());
            if (stages >= 4) {
                sp.assign(ba_poses, ba_poses + (size_t)nposes * 7); sx.assign(ba_points, ba_points + (size_t)npts * 3);
                int rd, no;
                orc_ba_optimize_active_map(sp.data(), nposes, sx.data(), npts, ep, el, obs, nedges, fixed, fx, fy, cx, cy, 5.991, 5.991, 5, 10,
                                           chi.data(), out.data(), &rd, &no);
            }
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return fail.load() ? -2 : 0;
}
