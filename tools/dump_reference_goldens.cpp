// dump_reference_goldens.cpp — step 2 of the reference pin kit: the IN-TREE functions of the reference, driven through its real libmyslam.so
// (and g2o's Levenberg iterations as the reference configures them), on the repo's synthetic inputs.  It cannot be built in this repo's
// build environment (OpenCV 3.4.8, Eigen, Sophus, g2o, glog are absent: DESIGN.md section 5, "parity unpinned"); it is for a maintainer
// whose machine has the reference built.  Nothing in the product, in bench.py or in the -m gpu tests depends on it.
//
//   python tools/make_reference_inputs.py                                   # tests/golden/reference/in_*.npy, in_*.pgm
//   g++ -O2 -std=c++14 tools/dump_reference_goldens.cpp -o dump_reference_goldens   (one line:)
//       -I<reference>/include -I/usr/include/eigen3 $(pkg-config --cflags opencv) -I<Sophus> -I<g2o include> \
//       -L<reference>/lib -lmyslam $(pkg-config --libs opencv) -lg2o_core -lg2o_stuff -lg2o_solver_csparse -lg2o_csparse_extension -lcxsparse -lglog
//   ./dump_reference_goldens tests/golden/reference                         # writes ref_*.npy there
//   python -m pytest tests/test_reference_pin.py -q                         # the XFAILs turn into real comparisons
//
// What it records (reference file:line of what is called):
//   ref_dac_kps / ref_dac_desc       ORBextractor(2000,1.2,8,20,7).DetectAndCompute(left)          src/ORBextractor.cpp:922-985
//   ref_det_kps                      ORBextractor(300, ...).Detect(left, mask)                      src/ORBextractor.cpp:989-1074
//   ref_screen_kps / ref_calc_desc   ScreenAndComputeKPsParams + CalcDescriptors on Detect's points  src/ORBextractor.cpp:1083-1129, 1180-1226
//   ref_tri_xyz / ref_tri_ok         triangulation() on 64 two-view cases                           include/myslam/algorithm.h:16-33
//   ref_<w>_edge_err / _jxi / _jxj   EdgeProjection::computeError / linearizeOplus per edge          include/myslam/g2o_types.h:115-144
//   ref_<w>_trace                    per Levenberg iteration: activeRobustChi2(), currentLambda(), levenbergIteration()
//   ref_<w>_poses / _points / _edge_chi2 / _rounds      the solve stage of Backend::OptimizeActiveMap  src/backend.cpp:126-232
// for the two windows w = ba (3 % gross outliers) and ba_bad (60 %: every round fails the inlier test).  Key-points are stored as rows of
// 7 floats (pt.x, pt.y, size, angle, response, octave, class_id), the layout of the repo's myslam_keypoint.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#ifndef MYSLAM_NPY_SELFTEST            // -DMYSLAM_NPY_SELFTEST: only the NPY reader / writer below, round-tripped by tests/test_reference_pin.py (no OpenCV needed)
#include <opencv2/opencv.hpp>

#include "myslam/ORBextractor.h"
#include "myslam/algorithm.h"
#include "myslam/g2o_types.h"

using namespace myslam;
#endif

// ---- minimal NPY v1 reader / writer (little-endian, C order) ------------------------------------------------------------------------------
struct Npy { std::vector<size_t> shape; std::string descr; std::vector<char> data; size_t count() const { size_t n = 1; for (size_t s : shape) n *= s; return n; } };

static bool npy_read(const std::string& path, Npy& a) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    char magic[10];
    f.read(magic, 10);
    if (std::memcmp(magic, "\x93NUMPY", 6) != 0) return false;
    const size_t hlen = (unsigned char)magic[8] | ((unsigned char)magic[9] << 8);
    std::string h(hlen, ' ');
    f.read(&h[0], hlen);
    const size_t d0 = h.find("'descr': '") + 10;
    a.descr = h.substr(d0, h.find('\'', d0) - d0);
    const size_t s0 = h.find('(', h.find("'shape'")) + 1, s1 = h.find(')', s0);
    a.shape.clear();
    std::string dims = h.substr(s0, s1 - s0);
    size_t pos = 0;
    while (pos < dims.size()) {
        while (pos < dims.size() && (dims[pos] == ' ' || dims[pos] == ',')) pos++;
        if (pos >= dims.size()) break;
        a.shape.push_back(std::stoul(dims.substr(pos)));
        while (pos < dims.size() && dims[pos] != ',') pos++;
    }
    const size_t item = std::stoul(a.descr.substr(2));
    a.data.resize(a.count() * item);
    f.read(a.data.data(), a.data.size());
    return (bool)f;
}

static void npy_write(const std::string& path, const char* descr, const std::vector<size_t>& shape, const void* data, size_t bytes) {
    std::string dims;
    for (size_t i = 0; i < shape.size(); i++) dims += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
    std::string h = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (" + dims + "), }";
    while ((10 + h.size() + 1) % 64 != 0) h += ' ';
    h += '\n';
    std::ofstream f(path, std::ios::binary);
    const char magic[8] = {'\x93', 'N', 'U', 'M', 'P', 'Y', 1, 0};
    f.write(magic, 8);
    const uint16_t hl = (uint16_t)h.size();
    f.write((const char*)&hl, 2);
    f.write(h.data(), h.size());
    f.write((const char*)data, bytes);
}
template <class T> static const T* as(const Npy& a) { return reinterpret_cast<const T*>(a.data.data()); }

#ifdef MYSLAM_NPY_SELFTEST
int main(int argc, char** argv) {          // copies every argv[i] (an .npy file) to argv[i] + ".copy.npy" through the reader and the writer
    for (int i = 1; i < argc; i++) {
        Npy a;
        if (!npy_read(argv[i], a)) { std::fprintf(stderr, "cannot read %s\n", argv[i]); return 1; }
        npy_write(std::string(argv[i]) + ".copy.npy", a.descr.c_str(), a.shape, a.data.data(), a.data.size());
    }
    return 0;
}
#else
static void save_kps(const std::string& path, const std::vector<cv::KeyPoint>& k) {
    std::vector<float> r(k.size() * 7);
    for (size_t i = 0; i < k.size(); i++) {
        float* p = &r[7 * i];
        p[0] = k[i].pt.x; p[1] = k[i].pt.y; p[2] = k[i].size; p[3] = k[i].angle; p[4] = k[i].response; p[5] = (float)k[i].octave; p[6] = (float)k[i].class_id;
    }
    npy_write(path, "<f4", {k.size(), 7}, r.data(), r.size() * 4);
}
static void save_desc(const std::string& path, const cv::Mat& d) {
    cv::Mat c = d.isContinuous() ? d : d.clone();
    npy_write(path, "|u1", {(size_t)c.rows, (size_t)c.cols}, c.data, (size_t)c.rows * c.cols);
}

// ---- g2o: what the post-iteration hook of Backend::OptimizeActiveMap's optimizer sees ------------------------------------------------------
struct Trace : public g2o::HyperGraphAction {
    g2o::SparseOptimizer* opt; g2o::OptimizationAlgorithmLevenberg* alg; std::vector<double> rows;
    HyperGraphAction* operator()(const g2o::HyperGraph*, Parameters*) override {
        rows.push_back(opt->activeRobustChi2()); rows.push_back(alg->currentLambda()); rows.push_back((double)alg->levenbergIteration());
        return this;
    }
};

static int dump_window(const std::string& dir, const std::string& w, const double* K4) {
    Npy poses, points, ep, el, obs, fixed;
    if (!npy_read(dir + "/in_" + w + "_poses.npy", poses) || !npy_read(dir + "/in_" + w + "_points.npy", points) || !npy_read(dir + "/in_" + w + "_edge_pose.npy", ep) ||
        !npy_read(dir + "/in_" + w + "_edge_point.npy", el) || !npy_read(dir + "/in_" + w + "_obs.npy", obs) || !npy_read(dir + "/in_" + w + "_fixed.npy", fixed)) return 1;
    const size_t P = poses.shape[0], L = points.shape[0], E = ep.shape[0];
    // the optimiser of src/backend.cpp:128-133
    typedef g2o::BlockSolver_6_3 BlockSolverType;
    typedef g2o::LinearSolverCSparse<BlockSolverType::PoseMatrixType> LinearSolverType;
    auto solver = new g2o::OptimizationAlgorithmLevenberg(g2o::make_unique<BlockSolverType>(g2o::make_unique<LinearSolverType>()));
    g2o::SparseOptimizer optimizer;
    optimizer.setAlgorithm(solver);
    Mat33 camK; camK << K4[0], 0, K4[2], 0, K4[1], K4[3], 0, 0, 1;
    const SE3 camExt;                                                     // the left camera's pose in the rig: identity (src/system.cpp:108-116)
    std::vector<VertexPose*> vp(P); std::vector<VertexXYZ*> vl(L);
    for (size_t p = 0; p < P; p++) {                                      // :140-151 — in_*_poses rows are (qx qy qz qw tx ty tz), Tcw
        const double* q = as<double>(poses) + 7 * p;
        VertexPose* v = new VertexPose();
        v->setId((int)p);
        v->setEstimate(SE3(Eigen::Quaterniond(q[3], q[0], q[1], q[2]), Vec3(q[4], q[5], q[6])));
        optimizer.addVertex(v); vp[p] = v;
    }
    const double chi2_th = 5.991;
    std::vector<EdgeProjection*> edges(E);
    for (size_t l = 0; l < L; l++) {                                      // :161-181
        VertexXYZ* v = new VertexXYZ;
        v->setEstimate(Vec3(as<double>(points)[3 * l], as<double>(points)[3 * l + 1], as<double>(points)[3 * l + 2]));
        v->setId((int)(P + l));
        v->setMarginalized(true);
        if (as<uint8_t>(fixed)[l]) v->setFixed(true);
        optimizer.addVertex(v); vl[l] = v;
    }
    for (size_t k = 0; k < E; k++) {                                      // :184-204 (edges grouped by landmark, as the loop over map points builds them)
        EdgeProjection* e = new EdgeProjection(camK, camExt);
        e->setId((int)k + 1);
        e->setVertex(0, vp[as<int32_t>(ep)[k]]);
        e->setVertex(1, vl[as<int32_t>(el)[k]]);
        e->setMeasurement(Vec2(as<double>(obs)[2 * k], as<double>(obs)[2 * k + 1]));
        e->setInformation(Mat22::Identity());
        auto rk = new g2o::RobustKernelHuber();
        rk->setDelta(chi2_th);
        e->setRobustKernel(rk);
        optimizer.addEdge(e); edges[k] = e;
    }
    // EdgeProjection at the initial estimate: error, dE/dxi (2x6), dE/dp (2x3), row-major
    std::vector<double> err(2 * E), jxi(12 * E), jxj(6 * E);
    for (size_t k = 0; k < E; k++) {
        edges[k]->computeError(); edges[k]->linearizeOplus();
        err[2 * k] = edges[k]->error()[0]; err[2 * k + 1] = edges[k]->error()[1];
        for (int r = 0; r < 2; r++) {
            for (int c = 0; c < 6; c++) jxi[12 * k + 6 * r + c] = edges[k]->jacobianOplusXi()(r, c);
            for (int c = 0; c < 3; c++) jxj[6 * k + 3 * r + c] = edges[k]->jacobianOplusXj()(r, c);
        }
    }
    npy_write(dir + "/ref_" + w + "_edge_err.npy", "<f8", {E, 2}, err.data(), err.size() * 8);
    npy_write(dir + "/ref_" + w + "_jxi.npy", "<f8", {E, 2, 6}, jxi.data(), jxi.size() * 8);
    npy_write(dir + "/ref_" + w + "_jxj.npy", "<f8", {E, 2, 3}, jxj.data(), jxj.size() * 8);
    // the rounds of :208-232, with the iteration trace
    Trace tr; tr.opt = &optimizer; tr.alg = solver;
    optimizer.addPostIterationAction(&tr);
    int iteration = 0;
    while (iteration < 5) {
        optimizer.initializeOptimization();
        optimizer.optimize(10);
        int cntOutlier = 0, cntInlier = 0;
        for (size_t k = 0; k < E; k++) { if (edges[k]->chi2() > chi2_th) cntOutlier++; else cntInlier++; }
        const double inlierRatio = cntInlier / double(cntInlier + cntOutlier);
        if (inlierRatio > 0.5) break;
        iteration++;
    }
    std::vector<double> op(7 * P), ol(3 * L), ochi(E);
    for (size_t p = 0; p < P; p++) {
        const SE3 T = vp[p]->estimate();
        const Eigen::Quaterniond q = T.unit_quaternion();
        double* o = &op[7 * p];
        o[0] = q.x(); o[1] = q.y(); o[2] = q.z(); o[3] = q.w(); o[4] = T.translation()[0]; o[5] = T.translation()[1]; o[6] = T.translation()[2];
    }
    for (size_t l = 0; l < L; l++) for (int c = 0; c < 3; c++) ol[3 * l + c] = vl[l]->estimate()[c];
    for (size_t k = 0; k < E; k++) ochi[k] = edges[k]->chi2();
    const int32_t rounds = iteration;
    npy_write(dir + "/ref_" + w + "_trace.npy", "<f8", {tr.rows.size() / 3, 3}, tr.rows.data(), tr.rows.size() * 8);
    npy_write(dir + "/ref_" + w + "_poses.npy", "<f8", {P, 7}, op.data(), op.size() * 8);
    npy_write(dir + "/ref_" + w + "_points.npy", "<f8", {L, 3}, ol.data(), ol.size() * 8);
    npy_write(dir + "/ref_" + w + "_edge_chi2.npy", "<f8", {E}, ochi.data(), ochi.size() * 8);
    npy_write(dir + "/ref_" + w + "_rounds.npy", "<i4", {1}, &rounds, 4);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <tests/golden/reference>\n", argv[0]); return 1; }
    const std::string dir = argv[1];
    const cv::Mat left = cv::imread(dir + "/in_left.pgm", cv::IMREAD_GRAYSCALE), mask = cv::imread(dir + "/in_mask.pgm", cv::IMREAD_GRAYSCALE);
    if (left.empty() || mask.empty()) { std::fprintf(stderr, "run tools/make_reference_inputs.py first\n"); return 1; }

    {   // DetectAndCompute, the KITTI config's extractor parameters at 2000 features
        ORBextractor ext(2000, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> kps; cv::Mat desc;
        ext.DetectAndCompute(left, cv::Mat(), kps, desc);
        save_kps(dir + "/ref_dac_kps.npy", kps); save_desc(dir + "/ref_dac_desc.npy", desc);
    }
    {   // Detect with the frontend's mask, then the loop closer's two calls on those points (LoopClosing::ProcessNewKF, src/loopclosing.cpp:91-112)
        ORBextractor ext(300, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> kps, screened; cv::Mat desc;
        ext.Detect(left, mask, kps);
        save_kps(dir + "/ref_det_kps.npy", kps);
        ext.ScreenAndComputeKPsParams(left, kps, screened);
        ext.CalcDescriptors(left, screened, desc);
        save_kps(dir + "/ref_screen_kps.npy", screened); save_desc(dir + "/ref_calc_desc.npy", desc);
    }
    {   // triangulation()
        Npy P34, pts;
        if (!npy_read(dir + "/in_tri_poses34.npy", P34) || !npy_read(dir + "/in_tri_points.npy", pts)) return 1;
        const size_t n = pts.shape[0];
        std::vector<SE3> poses;
        for (int c = 0; c < 2; c++) {
            const double* m = as<double>(P34) + 12 * c;
            Mat33 R; R << m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10];
            poses.push_back(SE3(R, Vec3(m[3], m[7], m[11])));
        }
        std::vector<double> xyz(3 * n); std::vector<uint8_t> ok(n);
        for (size_t i = 0; i < n; i++) {
            const double* p = as<double>(pts) + 6 * i;
            std::vector<Vec3> pn{Vec3(p[0], p[1], p[2]), Vec3(p[3], p[4], p[5])};
            Vec3 pw = Vec3::Zero();
            ok[i] = triangulation(poses, pn, pw) ? 1 : 0;
            for (int c = 0; c < 3; c++) xyz[3 * i + c] = pw[c];
        }
        npy_write(dir + "/ref_tri_xyz.npy", "<f8", {n, 3}, xyz.data(), xyz.size() * 8);
        npy_write(dir + "/ref_tri_ok.npy", "|u1", {n}, ok.data(), n);
    }
    Npy K;
    if (!npy_read(dir + "/in_K.npy", K)) return 1;
    if (dump_window(dir, "ba", as<double>(K)) || dump_window(dir, "ba_bad", as<double>(K))) return 1;
    const std::string ver = CV_VERSION;
    npy_write(dir + "/ref_opencv_version.npy", "|u1", {ver.size()}, ver.data(), ver.size());
    std::printf("wrote ref_*.npy into %s (OpenCV %s)\n", dir.c_str(), CV_VERSION);
    return 0;
}
#endif
