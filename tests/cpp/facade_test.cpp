// C++ parity test of the facade (product) against the oracle (checker).  Built and run by tests/test_gpu_facade.py.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../a-simple-stereo-slam-system-with-deep-loop-closing_amd/host/myslam_hip.hpp"
#include "../../oracle/oracle.h"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main() {
    const int H = 300, W = 420;
    std::vector<uint8_t> img((size_t)H * W);
    std::mt19937 rng(7);
    // blocky texture: random rectangles + noise (corner rich)
    for (auto& p : img) p = 128;
    for (int k = 0; k < 1500; k++) {
        int x0 = rng() % W, y0 = rng() % H, w = 3 + rng() % 22, h = 3 + rng() % 22, v = (int)(rng() % 181) - 90;
        for (int y = y0; y < std::min(H, y0 + h); y++)
            for (int x = x0; x < std::min(W, x0 + w); x++) img[(size_t)y * W + x] = (uint8_t)std::min(255, std::max(0, img[(size_t)y * W + x] + v));
    }
    for (auto& p : img) p = (uint8_t)std::min(255, std::max(0, (int)p + (int)(rng() % 7) - 3));

    myslam::ORBextractor ext(600, 1.2f, 8, 20, 7);
    EXPECT(ext.GetLevels() == 8 && std::fabs(ext.GetScaleFactor() - 1.2f) < 1e-6f);
    myslam::ImageView iv{img.data(), H, W, W}, nomask;
    std::vector<myslam::KeyPoint> kps; myslam::Descriptors desc;
    ext.DetectAndCompute(iv, nomask, kps, desc);
    orc_orb_params p{600, 1.2f, 8, 20, 7};
    std::vector<orc_keypoint> rk(2000); std::vector<uint8_t> rd(2000 * 32); int rn = 0;
    EXPECT(orc_detect_and_compute(&p, img.data(), H, W, W, nullptr, 0, rk.data(), rd.data(), 2000, &rn) == 0);
    EXPECT((int)kps.size() == rn && rn > 300);
    EXPECT(std::memcmp(kps.data(), rk.data(), sizeof(orc_keypoint) * rn) == 0);
    EXPECT(std::memcmp(desc.data(), rd.data(), (size_t)32 * rn) == 0);

    std::vector<myslam::KeyPoint> k0; ext.Detect(iv, nomask, k0);
    std::vector<orc_keypoint> r0(2000); int n0 = 0;
    orc_orb_params p0{600, 1.2f, 8, 20, 7};
    EXPECT(orc_detect(&p0, img.data(), H, W, W, nullptr, 0, r0.data(), 2000, &n0) == 0);
    EXPECT((int)k0.size() == n0 && std::memcmp(k0.data(), r0.data(), sizeof(orc_keypoint) * n0) == 0);

    std::vector<myslam::DMatch> matches;
    myslam::BFMatcherHamming::match(desc, desc, matches);
    EXPECT(matches.size() == kps.size());
    for (size_t i = 0; i < matches.size(); i++) EXPECT(matches[i].distance == 0.f && matches[i].queryIdx == (int)i);

    // empty inputs are silent no-ops, like the reference
    myslam::ImageView empty;
    ext.DetectAndCompute(empty, nomask, kps, desc);
    EXPECT(kps.empty() && desc.empty());

    // loop database decision rule
    std::vector<myslam::DeepLCD::DescrVector> db(80);
    for (auto& d : db) { double s = 0; for (auto& v : d) { v = std::fabs((float)(rng() % 1000) / 1000.f); s += v * v; } for (auto& v : d) v /= (float)std::sqrt(s); }
    myslam::LoopDatabase D(128);
    for (size_t i = 0; i < db.size(); i++) D.AddToDatabase(i, db[i]);
    unsigned long loop = 999; float mx = 0;
    EXPECT(D.DetectLoop(200, db[33], loop, &mx) && loop == 33 && mx > 0.999f);
    EXPECT(!D.DetectLoop(40, db[33], loop));          // id 33 is younger than cur-20 -> never scanned

    // local BA: 6 key-frames on a line looking down +z, 80 landmarks, every landmark seen by every key-frame, noisy observations
    {
        myslam::LocalBA ba;
        const int P = 6, L = 80;
        ba.fx = 718.856; ba.fy = 718.856; ba.cx = 607.1928; ba.cy = 185.2157;
        std::uniform_real_distribution<double> U(-1, 1);
        std::vector<double> tx(P);
        for (int p = 0; p < P; p++) { tx[p] = 0.4 * p; const double q[7] = {0, 0, 0, 1, -tx[p], 0, 0}; ba.poses.insert(ba.poses.end(), q, q + 7); }
        for (int l = 0; l < L; l++) {
            const double X[3] = {1.0 + 4 * U(rng), 1.5 * U(rng), 12 + 6 * U(rng)};
            for (int p = 0; p < P; p++) {
                const double xc = X[0] - tx[p], u = ba.fx * xc / X[2] + ba.cx, v = ba.fy * X[1] / X[2] + ba.cy;
                ba.edge_pose.push_back(p); ba.edge_pt.push_back(l);
                ba.obs.push_back(u + 0.5 * U(rng) + (l % 17 == 0 && p == 2 ? 25.0 : 0.0)); ba.obs.push_back(v + 0.5 * U(rng));
            }
            ba.points.push_back(X[0] + 0.05 * U(rng)); ba.points.push_back(X[1] + 0.05 * U(rng)); ba.points.push_back(X[2] + 0.05 * U(rng));
            ba.fixed.push_back(l % 9 == 0);
        }
        const int E = (int)ba.edge_pose.size();
        std::vector<double> rp = ba.poses, rx = ba.points, rchi(E); std::vector<uint8_t> rout(E); int rr = 0, rn2 = 0;
        EXPECT(orc_ba_optimize_active_map(rp.data(), P, rx.data(), L, ba.edge_pose.data(), ba.edge_pt.data(), ba.obs.data(), E, ba.fixed.data(),
                                          ba.fx, ba.fy, ba.cx, ba.cy, 5.991, 5.991, 5, 10, rchi.data(), rout.data(), &rr, &rn2) == 0);
        const int nout = ba.OptimizeActiveMap();
        EXPECT(nout == rn2 && nout >= 5);
        double dmax = 0;
        for (size_t i = 0; i < rp.size(); i++) dmax = std::max(dmax, std::fabs(rp[i] - ba.poses[i]));
        for (size_t i = 0; i < rx.size(); i++) dmax = std::max(dmax, std::fabs(rx[i] - ba.points[i]));
        EXPECT(dmax < 1e-7);
        int flagdiff = 0;
        for (int k = 0; k < E; k++) flagdiff += (ba.outlier[k] != rout[k]) && std::fabs(rchi[k] - 5.991) > 1e-6;
        EXPECT(flagdiff == 0);

        // the same window entered through the Map tables (Backend::OptimizeActiveMap's graph build, backend.cpp:139-206): key-frames and map
        // points in scrambled container order with arbitrary ids, plus an outlier map point, an outlier feature and a landmark whose
        // first observer left the window; Flatten() must reproduce the flat arrays above (minus the skipped rows) and the solve must agree
        myslam::LocalBA fb;
        fb.fx = ba.fx; fb.fy = ba.fy; fb.cx = ba.cx; fb.cy = ba.cy;
        std::vector<double> p0v(7 * P), x0v(3 * L);
        for (int pp = 0; pp < P; pp++) { const double q[7] = {0, 0, 0, 1, -tx[pp], 0, 0}; std::copy(q, q + 7, p0v.begin() + 7 * pp); }
        const int korder[6] = {3, 0, 5, 1, 4, 2};
        for (int i = 0; i < P; i++) fb.AddKeyFrame(100 + 10 * korder[i], &p0v[7 * korder[i]]);            // ids 100, 110, ... ascending with the slot
        std::vector<int> lorder(L);
        for (int l = 0; l < L; l++) lorder[l] = (l * 37) % L;                                                // 37 is coprime to 80
        // (landmark positions: `ba`'s optimised points serve as the second problem's initial estimate)
        for (int i = 0; i < L; i++) {
            const int l = lorder[i];
            fb.AddMapPoint(1000 + l, &ba.points[3 * l], /*isOutlier=*/l == 5, /*firstObserverKFId=*/(l % 9 == 0) ? 7 : 100);
        }
        for (int i = 0; i < L; i++) {
            const int l = lorder[i];
            for (int pp = 0; pp < P; pp++)
                fb.AddObservation(1000 + l, 100 + 10 * pp, (float)ba.obs[2 * (l * P + pp)], (float)ba.obs[2 * (l * P + pp) + 1], /*featureIsOutlier=*/l == 8 && pp == 3);
        }
        fb.Flatten();
        EXPECT((int)fb.poses.size() == 7 * P && (int)fb.points.size() == 3 * (L - 1) && (int)fb.edge_pose.size() == E - P - 1);
        bool ok = true;
        for (int pp = 0; pp < P; pp++) ok = ok && fb.kf_ids[fb.pose_src[pp]] == (uint64_t)(100 + 10 * pp);
        for (size_t j = 0; j < fb.pt_src.size(); j++) {
            const int l = (int)fb.mp_ids[fb.pt_src[j]] - 1000;
            ok = ok && l != 5 && (j == 0 || fb.mp_ids[fb.pt_src[j]] > fb.mp_ids[fb.pt_src[j - 1]]) && fb.fixed[j] == (l % 9 == 0 ? 1 : 0);
        }
        for (size_t k = 0; k < fb.edge_pose.size(); k++) {
            const int row = fb.edge_src[k], l = (int)fb.obs_mp_id[row] - 1000, pp = (int)(fb.obs_kf_id[row] - 100) / 10;
            ok = ok && fb.edge_pose[k] == pp && (int)fb.mp_ids[fb.pt_src[fb.edge_pt[k]]] - 1000 == l && !(l == 8 && pp == 3) && l != 5;
            ok = ok && fb.obs[2 * k] == (double)(float)ba.obs[2 * (l * P + pp)] && (k == 0 || fb.edge_pt[k] >= fb.edge_pt[k - 1]);
        }
        EXPECT(ok);
        // solve through the facade and through the oracle on the flattened arrays
        std::vector<double> fp = fb.poses, fx2 = fb.points, fchi(fb.edge_pose.size()); std::vector<uint8_t> fout(fb.edge_pose.size()); int fr = 0, fn = 0;
        EXPECT(orc_ba_optimize_active_map(fp.data(), P, fx2.data(), (int)fb.points.size() / 3, fb.edge_pose.data(), fb.edge_pt.data(), fb.obs.data(),
                                          (int)fb.edge_pose.size(), fb.fixed.data(), fb.fx, fb.fy, fb.cx, fb.cy, 5.991, 5.991, 5, 10, fchi.data(), fout.data(), &fr, &fn) == 0);
        const int fnout = fb.OptimizeActiveMap();
        EXPECT(fnout == fn);
        double fd = 0;
        for (size_t i = 0; i < fp.size(); i++) fd = std::max(fd, std::fabs(fp[i] - fb.poses[i]));
        for (size_t i = 0; i < fx2.size(); i++) fd = std::max(fd, std::fabs(fx2[i] - fb.points[i]));
        EXPECT(fd < 1e-7);
        // a window beyond the solve kernels' pose limit is refused with a status, not truncated
        myslam::LocalBA big = fb;
        for (int extra = 0; extra < 6; extra++) big.poses.insert(big.poses.end(), p0v.begin(), p0v.begin() + 7);
        bool threw = false;
        try { big.OptimizeActiveMap(); } catch (const std::exception&) { threw = true; }
        EXPECT(threw && (int)big.poses.size() / 7 > MYSLAM_BA_MAX_WINDOW_POSES);
    }

    // LK tracker: the image against itself shifted by 2 px
    {
        std::vector<uint8_t> img2(img.size());
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) img2[(size_t)y * W + x] = img[(size_t)y * W + std::min(W - 1, x + 2)];
        myslam::ImageView iv2{img2.data(), H, W, W};
        std::vector<myslam::Point2f> p0, p1; std::vector<uint8_t> st; std::vector<float> er;
        for (size_t i = 0; i < k0.size() && i < 150; i++) p0.push_back({k0[i].x, k0[i].y});
        p1 = p0;
        myslam::PyrLKTracker lk;
        lk.calcOpticalFlowPyrLK(iv, iv2, p0, p1, st, er);
        std::vector<float> r1(2 * p0.size()); std::vector<uint8_t> rs(p0.size()); std::vector<float> re(p0.size());
        for (size_t i = 0; i < p0.size(); i++) { r1[2 * i] = p0[i].x; r1[2 * i + 1] = p0[i].y; }
        EXPECT(orc_lk_track(img.data(), img2.data(), H, W, W, W, reinterpret_cast<const float*>(p0.data()), r1.data(), (int)p0.size(), 11, 3, 30, 0.01f, 1e-4f,
                            rs.data(), re.data()) >= 0);
        EXPECT(std::memcmp(r1.data(), p1.data(), sizeof(float) * r1.size()) == 0 && std::memcmp(rs.data(), st.data(), rs.size()) == 0);
        int good = 0; for (size_t i = 0; i < p0.size(); i++) good += st[i] && std::fabs(p1[i].x - p0[i].x + 2.f) < 0.2f;
        EXPECT(good > (int)p0.size() * 8 / 10);
    }

    // loop correction: a 30 key-frame circle whose odometry drifts, closed by one loop edge
    {
        const int N = 30;
        myslam::PoseGraph pg;
        std::vector<double> gt((size_t)N * 7);
        auto compose = [](const double* a, const double* b, bool invb, double* out) { orc_se3_compose(a, b, invb ? 1 : 0, out); };
        for (int i = 0; i < N; i++) {
            const double ang = 2 * M_PI * i / (N - 2), xi[6] = {0, 0, 0, 0, -ang, 0};
            double R[7]; orc_se3_exp(xi, R);
            const double twc[3] = {20 * std::cos(ang), 0, 20 * std::sin(ang)};
            // Tcw = (Rwc^T, -Rwc^T twc) with Rwc = exp(+ang about y): here R already holds the inverse rotation
            double T[7] = {R[0], R[1], R[2], R[3], 0, 0, 0}, neg[7] = {0, 0, 0, 1, -twc[0], -twc[1], -twc[2]};
            compose(T, neg, false, &gt[(size_t)7 * i]);
        }
        std::uniform_real_distribution<double> U(-1, 1);
        std::vector<double> est(gt.begin(), gt.begin() + 7);
        for (int i = 1; i < N; i++) {
            double M[7], noise[7], Mn[7], Ti[7];
            compose(&gt[(size_t)7 * i], &gt[(size_t)7 * (i - 1)], true, M);
            const double xi[6] = {0.02 * U(rng), 0.02 * U(rng), 0.02 * U(rng), 0.003 * U(rng), 0.003 * U(rng), 0.003 * U(rng)};
            orc_se3_exp(xi, noise); compose(noise, M, false, Mn);
            compose(Mn, &est[(size_t)7 * (i - 1)], false, Ti);
            est.insert(est.end(), Ti, Ti + 7);
            pg.AddEdge(i, i - 1, Mn);
        }
        for (int i = 0; i < N; i++) pg.AddKeyFrame(&est[(size_t)7 * i], i == 0 || i == 1);
        double Ml0[7], Ml[7], nl[7]; compose(&gt[(size_t)7 * (N - 1)], &gt[7], true, Ml0);
        const double xl[6] = {0.01, -0.02, 0.015, 0.002, -0.001, 0.0015};      // a noisy loop measurement: the optimum keeps a residual
        orc_se3_exp(xl, nl); compose(nl, Ml0, false, Ml);
        pg.AddEdge(N - 1, 1, Ml);
        std::vector<double> rp = pg.poses; double rchi = 0, gchi = 0; int rit = 0;
        EXPECT(orc_pose_graph_optimize(rp.data(), N, pg.fixed.data(), pg.edge_v0.data(), pg.edge_v1.data(), pg.meas.data(), (int)pg.edge_v0.size(), 20, &rchi, &rit) == 0);
        const std::vector<double> before = pg.poses;
        const int git = pg.Optimize(20, &gchi);
        if (git != rit) printf("pose graph: device %d iterations chi2 %.12g, oracle %d iterations chi2 %.12g\n", git, gchi, rit, rchi);
        EXPECT(git == rit || std::fabs(gchi - rchi) <= 1e-9 * rchi);      // at the rounding floor Levenberg gives up at a noise-dependent iteration
        EXPECT(std::fabs(gchi - rchi) <= 1e-3 * rchi + 1e-12);
        double dmax = 0;
        for (size_t i = 0; i < rp.size(); i++) dmax = std::max(dmax, std::fabs(rp[i] - pg.poses[i]));
        EXPECT(dmax < 5e-4);                 // the operator's numeric-Jacobian noise floor, see tests/test_gpu_pgo.py
        std::vector<double> pts = {1, 2, 3, -4, 0.5, 9, 7, 7, 7}, rpts = pts; std::vector<int32_t> kf = {5, -1, 29};
        myslam::PoseGraph::CorrectMapPoints(before, pg.poses, kf, pts);
        EXPECT(orc_correct_map_points(before.data(), pg.poses.data(), N, kf.data(), rpts.data(), 3) == 0);
        for (int i = 0; i < 9; i++) EXPECT(std::fabs(pts[i] - rpts[i]) < 1e-10);
        EXPECT(pts[3] == -4 && pts[4] == 0.5 && pts[5] == 9);
    }

    // loop verification: PnP-RANSAC on 80 map points seen from a known pose, every fifth match wrong
    {
        const double fx = 718.856, fy = 718.856, cx = 607.1928, cy = 185.2157;
        const double xi[6] = {0.8, -0.1, 1.5, 0.02, 0.3, -0.01};
        double T[7]; orc_se3_exp(xi, T);
        const double qx = T[0], qy = T[1], qz = T[2], qw = T[3];
        const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw), 2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz),
                             2 * (qy * qz - qx * qw), 2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
        std::uniform_real_distribution<double> U(-1, 1);
        std::vector<myslam::Point3f> p3; std::vector<myslam::Point2f> p2;
        for (int i = 0; i < 80; i++) {
            const double pc[3] = {10 * U(rng), 2.5 * U(rng), 22 + 16 * U(rng)}, d[3] = {pc[0] - T[4], pc[1] - T[5], pc[2] - T[6]};
            p3.push_back({(float)(R[0] * d[0] + R[3] * d[1] + R[6] * d[2]), (float)(R[1] * d[0] + R[4] * d[1] + R[7] * d[2]), (float)(R[2] * d[0] + R[5] * d[1] + R[8] * d[2])});
            double u = fx * pc[0] / pc[2] + cx + 0.4 * U(rng), v = fy * pc[1] / pc[2] + cy + 0.4 * U(rng);
            if (i % 5 == 0) { u = 620 + 600 * U(rng); v = 188 + 180 * U(rng); }
            p2.push_back({(float)u, (float)v});
        }
        double gp[7], rp[7]; std::vector<uint8_t> gin, rin(80); int rn = 0;
        EXPECT(myslam::solvePnPRansac(p3, p2, fx, fy, cx, cy, gp, &gin));
        EXPECT(orc_solve_pnp_ransac(reinterpret_cast<const float*>(p3.data()), reinterpret_cast<const float*>(p2.data()), 80, fx, fy, cx, cy, 100, (double)5.991f, 0.99,
                                    rp, rin.data(), &rn) == 0);
        EXPECT(gin == rin && rn >= 60);
        for (int i = 0; i < 7; i++) EXPECT(std::fabs(gp[i] - rp[i]) < 1e-9);
        for (int i = 0; i < 3; i++) EXPECT(std::fabs(gp[4 + i] - T[4 + i]) < 0.05);
        std::vector<myslam::Point3f> few(p3.begin(), p3.begin() + 4); std::vector<myslam::Point2f> few2(p2.begin(), p2.begin() + 4);
        EXPECT(!myslam::solvePnPRansac(few, few2, fx, fy, cx, cy, gp));
    }

    // loop closer bookkeeping (a26): KeyFrame::{mvPyramidKeyPoints, mORBDescriptors} of two key-frames of the same scene, MatchFeatures,
    // LoopLocalFusion — the reference's own call order (ProcessNewKF blurs the image in place first)
    {
        std::vector<uint8_t> imgB = img;                                  // "current" key-frame: the scene shifted by 3 px
        for (int y = 0; y < H; y++) for (int x = 0; x < W - 3; x++) imgB[(size_t)y * W + x] = img[(size_t)y * W + x + 3];
        myslam::ORBextractor det(150, 1.2f, 8, 20, 7);
        std::vector<uint8_t> imgs[2] = {img, imgB};
        myslam::KeyFrameFeatures kf[2];
        std::vector<std::vector<orc_keypoint>> rpyr(2); std::vector<std::vector<uint8_t>> rdesc(2);
        std::vector<float> w(myslam_lcd_nweights());
        std::mt19937 wr(3); std::normal_distribution<float> N01(0.f, 0.05f);
        for (auto& v : w) v = N01(wr);
        myslam::DeepLCD lcd(w.data(), w.size());
        for (int k = 0; k < 2; k++) {
            myslam::ImageView v{imgs[k].data(), H, W, W};
            std::vector<myslam::KeyPoint> feats; det.Detect(v, nomask, feats);
            std::vector<uint8_t> ref = imgs[k];
            (void)lcd.calcDescrOriginalImg(v);                           // blurs imgs[k] in place (reference quirk)
            std::vector<float> net(120 * 160);
            EXPECT(orc_calc_preproc(ref.data(), H, W, W, 1, net.data()) == 0);
            EXPECT(ref == imgs[k]);
            kf[k].Compute(ext, v, feats);
            // the oracle's chain: expand by hand, screen, describe
            orc_orb_params p8{600, 1.2f, 8, 20, 7};
            std::vector<orc_keypoint> pyr(feats.size() * 8), out(feats.size() * 8 + 1);
            for (size_t i = 0; i < feats.size(); i++)
                for (int l = 0; l < 8; l++) { orc_keypoint q; std::memcpy(&q, &feats[i], sizeof(q)); q.octave = l; q.response = -1; q.class_id = (int)i; pyr[i * 8 + l] = q; }
            int no = 0;
            EXPECT(orc_screen(&p8, ref.data(), H, W, W, pyr.data(), (int)pyr.size(), out.data(), (int)out.size(), &no) == 0);
            out.resize(no); rpyr[k] = out; rdesc[k].resize((size_t)no * 32);
            EXPECT(orc_calc_descriptors(&p8, ref.data(), H, W, W, out.data(), no, rdesc[k].data()) == 0);
            EXPECT((int)kf[k].mvPyramidKeyPoints.size() == no && no > 100);
            EXPECT(std::memcmp(kf[k].mvPyramidKeyPoints.data(), out.data(), sizeof(orc_keypoint) * no) == 0);
            EXPECT(kf[k].mORBDescriptors == rdesc[k]);
        }
        std::vector<std::pair<int, int>> pairs;
        const bool enough = myslam::MatchFeatures(kf[0], kf[1], pairs);
        // reference restatement of :172-194 on the oracle's matcher output
        const int nq = (int)rpyr[0].size(), nt = (int)rpyr[1].size();
        std::vector<int32_t> ri(nq), rdst(nq);
        EXPECT(orc_hamming_match(rdesc[0].data(), nq, rdesc[1].data(), nt, ri.data(), rdst.data()) == 0);
        int mn = rdst[0]; for (int i = 1; i < nq; i++) mn = std::min(mn, rdst[i]);
        std::vector<std::pair<int, int>> want;
        for (int i = 0; i < nq; i++) if ((double)rdst[i] <= std::max(2.0 * mn, 30.0)) want.emplace_back(rpyr[1][ri[i]].class_id, rpyr[0][i].class_id);
        std::sort(want.begin(), want.end()); want.erase(std::unique(want.begin(), want.end()), want.end());
        EXPECT(pairs == want && enough == (want.size() >= 10) && want.size() >= 10);
        // LoopLocalFusion on three active key-frames
        std::vector<double> act = {0, 0, 0, 1, 0, 0, 0,  0, 0.01, 0, 1, 0.5, 0, 0.1,  0.01, 0, 0, 1, 1.0, 0.02, 0.2}, ract = act;
        const double corr[7] = {0.0, 0.02, 0.0, 1.0, 1.1, 0.0, 0.25};
        std::vector<double> pts = {1, 2, 13, -4, 0.5, 9, 7, 1, 17}, rpts = pts; std::vector<int32_t> first = {0, -1, 2};
        myslam::LoopLocalFusion(act, 2, corr, first, pts);
        EXPECT(orc_loop_local_fusion(ract.data(), 3, 2, corr, first.data(), rpts.data(), 3) == 0);
        for (size_t i = 0; i < act.size(); i++) EXPECT(std::fabs(act[i] - ract[i]) < 1e-12);
        for (size_t i = 0; i < pts.size(); i++) EXPECT(std::fabs(pts[i] - rpts[i]) < 1e-10);
    }

    printf(fails ? "FACADE TEST FAILED (%d)\n" : "FACADE TEST OK (%d failures)\n", fails);
    return fails ? 1 : 0;
}
