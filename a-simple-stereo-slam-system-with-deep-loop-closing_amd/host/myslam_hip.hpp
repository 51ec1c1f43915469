// myslam_hip.hpp — C++ facade over the C ABI (include/myslam_hip.h) with the reference's own class and method
// names, so that Frontend / LoopClosing / Backend code of libmyslam.so keeps compiling against it:
//
//   myslam::ORBextractor   include/myslam/ORBextractor.h:52-110   (Detect, DetectAndCompute,
//                                                                  ScreenAndComputeKPsParams, CalcDescriptors, getters)
//   myslam::DeepLCD        include/myslam/deeplcd.h:21-48          (calcDescrOriginalImg, calcDescr, score, DescrVector)
//   myslam::BFMatcherHamming::match                                 (cv::BFMatcher use at src/loopclosing.cpp:33,172)
//   myslam::triangulation  include/myslam/algorithm.h:16-33        (stereo rig form)
//   myslam::LoopDatabase   LoopClosing::DetectLoop / AddToDatabase  src/loopclosing.cpp:124-161, 651-659
//   myslam::LocalBA::{Build,Optimize,OptimizeActiveMap}  Backend::OptimizeActiveMap  src/backend.cpp:126-243
//   myslam::PyrLKTracker::calcOpticalFlowPyrLK        cv::calcOpticalFlowPyrLK call sites        src/frontend.cpp:150-153, 358-361
//   myslam::EstimateCurrentPose                       g2o stage of Frontend::EstimateCurrentPose src/frontend.cpp:176-276
//   myslam::KeyFrameFeatures / MatchFeatures          KeyFrame::{mvPyramidKeyPoints, mORBDescriptors} + LoopClosing::ProcessNewKF /
//                                                     MatchFeatures bookkeeping                   src/loopclosing.cpp:83-121, 163-203
//   myslam::LoopLocalFusion                           arithmetic of LoopClosing::LoopLocalFusion  src/loopclosing.cpp:466-507
//
// No OpenCV / Eigen / g2o: images are (data, rows, cols, step) views, cv::KeyPoint is the layout-compatible
// myslam_keypoint, DescrVector is std::array<float,1064>.  Errors the reference reports by logging + return keep
// that behaviour (empty inputs are no-ops); everything else throws std::runtime_error.  There is no CPU fallback.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <algorithm>
#include <vector>

#include "../../include/myslam_hip.h"

namespace myslam {

struct ImageView {            // cv::Mat CV_8UC1 stand-in
    uint8_t* data = nullptr;
    int rows = 0, cols = 0, step = 0;
    bool empty() const { return !data || rows <= 0 || cols <= 0; }
};
using KeyPoint = myslam_keypoint;                       // cv::KeyPoint layout
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; };
using Descriptors = std::vector<uint8_t>;               // N x 32, row-major (cv::Mat CV_8UC1 N x 32)

inline void check(int rc, const char* what) {
    if (rc != MYSLAM_OK) throw std::runtime_error(std::string(what) + " failed with status " + std::to_string(rc));
}

class ORBextractor {
   public:
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : nfeatures_(nfeatures), nlevels_(nlevels) {
        check(myslam_orb_create(&h_, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST), "myslam_orb_create");
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mnFeaturesPerLevel.resize(nlevels); umax.resize(16);
        check(myslam_orb_get_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mnFeaturesPerLevel.data(), umax.data()),
              "myslam_orb_get_tables");
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) { mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i]; }
    }
    ~ORBextractor() { if (h_) myslam_orb_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // ORBextractor.h:61-63
    void DetectAndCompute(const ImageView& image, const ImageView& mask, std::vector<KeyPoint>& keypoints, Descriptors& descriptors) {
        keypoints.clear(); descriptors.clear();
        if (image.empty()) return;                                               // ORBextractor.cpp:924
        const int cap = myslam_orb_max_keypoints_for(h_, image.rows, image.cols);
        keypoints.resize(cap); descriptors.resize((size_t)cap * 32);
        int n = 0;
        check(myslam_orb_detect_and_compute(h_, image.data, image.rows, image.cols, image.step, mask.empty() ? nullptr : mask.data,
                                            mask.step, keypoints.data(), descriptors.data(), cap, &n), "myslam_orb_detect_and_compute");
        keypoints.resize(n); descriptors.resize((size_t)n * 32);
    }
    // ORBextractor.h:70-71
    void Detect(const ImageView& image, const ImageView& mask, std::vector<KeyPoint>& keypoints) {
        keypoints.clear();
        if (image.empty()) return;                                               // ORBextractor.cpp:990
        const int cap = myslam_orb_max_keypoints_for(h_, image.rows, image.cols);
        keypoints.resize(cap);
        int n = 0;
        check(myslam_orb_detect(h_, image.data, image.rows, image.cols, image.step, mask.empty() ? nullptr : mask.data, mask.step,
                                keypoints.data(), cap, &n), "myslam_orb_detect");
        keypoints.resize(n);
    }
    // ORBextractor.h:83-84 (keypoints is updated in place exactly as the reference's loop does)
    void ScreenAndComputeKPsParams(const ImageView& image, std::vector<KeyPoint>& keypoints, std::vector<KeyPoint>& out_keypoints) {
        out_keypoints.clear();
        if (image.empty() || keypoints.empty()) return;                          // ORBextractor.cpp:1085-1088
        out_keypoints.resize(keypoints.size());
        int n = 0;
        check(myslam_orb_screen_and_compute_params(h_, image.data, image.rows, image.cols, image.step, keypoints.data(),
                                                   (int)keypoints.size(), out_keypoints.data(), (int)out_keypoints.size(), &n),
              "myslam_orb_screen_and_compute_params");
        out_keypoints.resize(n);
    }
    // ORBextractor.h:65-67
    void CalcDescriptors(const ImageView& image, const std::vector<KeyPoint>& keypoints, Descriptors& descriptors) {
        descriptors.clear();
        if (image.empty() || keypoints.empty()) return;                          // ORBextractor.cpp:1183-1186
        descriptors.resize(keypoints.size() * 32);
        check(myslam_orb_calc_descriptors(h_, image.data, image.rows, image.cols, image.step, keypoints.data(), (int)keypoints.size(),
                                          descriptors.data()), "myslam_orb_calc_descriptors");
    }
    // scheduling / parity knobs (no reference counterpart; see include/myslam_hip.h)
    void SetOption(int option, int value) { check(myslam_orb_set_option(h_, option, value), "myslam_orb_set_option"); }
    void SetGaussTaps(const int32_t* q7) { check(myslam_orb_set_gauss_taps(h_, q7), "myslam_orb_set_gauss_taps"); }
    // getters, ORBextractor.h:87-107
    int GetLevels() const { return nlevels_; }
    float GetScaleFactor() const { return mvScaleFactor.size() > 1 ? mvScaleFactor[1] : 1.f; }
    const std::vector<float>& GetScaleFactors() const { return mvScaleFactor; }
    const std::vector<float>& GetInverseScaleFactors() const { return mvInvScaleFactor; }
    const std::vector<float>& GetScaleSigmaSquares() const { return mvLevelSigma2; }
    const std::vector<float>& GetInverseScaleSigmaSquares() const { return mvInvLevelSigma2; }
    myslam_orb* handle() { return h_; }

    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;

   private:
    myslam_orb* h_ = nullptr;
    int nfeatures_, nlevels_;
};

class DeepLCD {
   public:
    using DescrVector = std::array<float, MYSLAM_LCD_DIM>;     // Eigen::Matrix<float,1064,1>, deeplcd.h:25
    // DeepLCD(network_definition_file, pre_trained_model_file, gpu_id) deeplcd.h:33: the reference's own two files, read without Caffe
    DeepLCD(const std::string& network_definition_file, const std::string& pre_trained_model_file, int /*gpu_id*/ = -1) {
        check(myslam_lcd_create_from_caffe(&h_, network_definition_file.c_str(), pre_trained_model_file.c_str()), "myslam_lcd_create_from_caffe");
    }
    // own model file (CALCW1 / CALCW2, csrc/calc.hip)
    explicit DeepLCD(const std::string& weights_file) { check(myslam_lcd_create_from_file(&h_, weights_file.c_str()), "myslam_lcd_create_from_file"); }
    bool UsesFusedKernels() const { return myslam_lcd_uses_fused_kernels(h_) == 1; }
    DeepLCD(const float* weights, size_t n) { check(myslam_lcd_create(&h_, weights, n), "myslam_lcd_create"); }
    ~DeepLCD() { if (h_) myslam_lcd_destroy(h_); }
    DeepLCD(const DeepLCD&) = delete;
    DeepLCD& operator=(const DeepLCD&) = delete;
    float score(const DescrVector& d1, const DescrVector& d2) const { return myslam_lcd_score(d1.data(), d2.data()); }   // deeplcd.cpp:35-39
    // deeplcd.cpp:43-52 — blurs originalImg in place, like the reference
    DescrVector calcDescrOriginalImg(const ImageView& originalImg) {
        DescrVector d;
        check(myslam_lcd_calc_descr_original_img(h_, originalImg.data, originalImg.rows, originalImg.cols, originalImg.step, 1, d.data()),
              "myslam_lcd_calc_descr_original_img");
        return d;
    }
    DescrVector calcDescr(const ImageView& im160x120) {            // deeplcd.cpp:55-91
        DescrVector d;
        check(myslam_lcd_calc_descr(h_, im160x120.data, im160x120.step, d.data()), "myslam_lcd_calc_descr");
        return d;
    }

   private:
    myslam_lcd* h_ = nullptr;
};

struct BFMatcherHamming {      // cv::DescriptorMatcher::create("BruteForce-Hamming")->match(query, train, matches)
    static void match(const Descriptors& query, const Descriptors& train, std::vector<DMatch>& matches) {
        const int nq = (int)(query.size() / 32), nt = (int)(train.size() / 32);
        std::vector<int32_t> idx(nq), dist(nq);
        check(myslam_hamming_match(query.data(), nq, train.data(), nt, idx.data(), dist.data()), "myslam_hamming_match");
        matches.resize(nq);
        for (int i = 0; i < nq; i++) matches[i] = DMatch{i, idx[i], 0, (float)dist[i]};
    }
};

// What LoopClosing keeps per key-frame for loop verification (include/myslam/keyframe.h:44-52): the pyramid key-points that survived
// ScreenAndComputeKPsParams (class_id = index of the feature they belong to) and their descriptors.
struct KeyFrameFeatures {
    std::vector<KeyPoint> mvPyramidKeyPoints;
    Descriptors mORBDescriptors;
    // LoopClosing::ProcessNewKF, src/loopclosing.cpp:94-112.  `image` is the key-frame's left image AFTER DeepLCD::calcDescrOriginalImg
    // blurred it in place (the reference's order of calls); featureKeyPoints = mvpFeaturesLeft[i]->mkpPosition.
    void Compute(ORBextractor& extractor, const ImageView& image, const std::vector<KeyPoint>& featureKeyPoints) {
        const int nl = extractor.GetLevels();
        std::vector<KeyPoint> pyr(featureKeyPoints.size() * (size_t)nl);
        check(myslam_expand_pyramid_keypoints(featureKeyPoints.data(), (int)featureKeyPoints.size(), nl, pyr.data()), "myslam_expand_pyramid_keypoints");
        extractor.ScreenAndComputeKPsParams(image, pyr, mvPyramidKeyPoints);
        extractor.CalcDescriptors(image, mvPyramidKeyPoints, mORBDescriptors);
    }
};

// LoopClosing::MatchFeatures, src/loopclosing.cpp:163-203: BFMatcher(loop -> current), distance filter, (current feature, loop feature)
// pairs in the order of the reference's std::set.  false when fewer than 10 pairs survive (:196).
inline bool MatchFeatures(const KeyFrameFeatures& loopKF, const KeyFrameFeatures& currentKF, std::vector<std::pair<int, int>>& validFeatureMatches) {
    validFeatureMatches.clear();
    const int nq = (int)loopKF.mvPyramidKeyPoints.size(), nt = (int)currentKF.mvPyramidKeyPoints.size();
    if (nq == 0 || nt == 0) return false;
    std::vector<int32_t> idx(nq), dist(nq), pairs((size_t)nq * 2);
    check(myslam_hamming_match(loopKF.mORBDescriptors.data(), nq, currentKF.mORBDescriptors.data(), nt, idx.data(), dist.data()), "myslam_hamming_match");
    int np = 0;
    check(myslam_match_feature_pairs(idx.data(), dist.data(), nq, loopKF.mvPyramidKeyPoints.data(), currentKF.mvPyramidKeyPoints.data(), nt,
                                     pairs.data(), &np), "myslam_match_feature_pairs");
    for (int k = 0; k < np; k++) validFeatureMatches.emplace_back(pairs[2 * k], pairs[2 * k + 1]);
    return np >= 10;
}

// LoopClosing::LoopLocalFusion, src/loopclosing.cpp:466-507 (the arithmetic; re-linking observations :509-532 stays with the Map):
// activePoses (n x 7, in/out) move rigidly with the corrected current key-frame, points (n x 3, in/out) follow firstActiveKF[i] (< 0 = skip)
inline void LoopLocalFusion(std::vector<double>& activePoses, int currentIndex, const double correctedCurrentPose[7],
                            const std::vector<int32_t>& firstActiveKF, std::vector<double>& points) {
    check(myslam_loop_local_fusion(activePoses.data(), (int)(activePoses.size() / 7), currentIndex, correctedCurrentPose, firstActiveKF.data(),
                                   points.data(), (int)firstActiveKF.size()), "myslam_loop_local_fusion");
}

// triangulation() of algorithm.h:16-33 for the stereo rig (left ext = I, right ext t = (-baseline,0,0)); ok = success && z > 0
inline void triangulation(const std::vector<float>& xl, const std::vector<float>& yl, const std::vector<float>& xr,
                          const std::vector<float>& yr, double fx, double fy, double cx, double cy, double baseline,
                          std::vector<std::array<double, 3>>& pts, std::vector<uint8_t>& ok) {
    const int n = (int)xl.size();
    pts.resize(n); ok.resize(n);
    check(myslam_triangulate_stereo(xl.data(), yl.data(), xr.data(), yr.data(), n, fx, fy, cx, cy, baseline,
                                    reinterpret_cast<double*>(pts.data()), ok.data()), "myslam_triangulate_stereo");
}

class LoopDatabase {           // LoopClosing::_mvDatabase + DetectLoop + AddToDatabase
   public:
    explicit LoopDatabase(int capacity, float thresHigh = 0.94f, float thresLow = 0.92f) : th1_(thresHigh), th2_(thresLow) {
        check(myslam_lcddb_create(&h_, capacity), "myslam_lcddb_create");
    }
    ~LoopDatabase() { if (h_) myslam_lcddb_destroy(h_); }
    LoopDatabase(const LoopDatabase&) = delete;
    LoopDatabase& operator=(const LoopDatabase&) = delete;
    void AddToDatabase(unsigned long kfId, const DeepLCD::DescrVector& d) { check(myslam_lcddb_append(h_, kfId, d.data()), "myslam_lcddb_append"); }
    size_t size() const { return (size_t)myslam_lcddb_size(h_); }
    size_t capacity() const { return (size_t)myslam_lcddb_capacity(h_); }          // grows by itself (std::map has no bound); reserve() avoids the moves
    void reserve(int rows) { check(myslam_lcddb_reserve(h_, rows), "myslam_lcddb_reserve"); }
    // loopclosing.cpp:124-161: true + loop KF id when maxScore >= high threshold and at most 3 scores exceed the low one
    bool DetectLoop(unsigned long curKFId, const DeepLCD::DescrVector& d, unsigned long& loopKFId, float* maxScore = nullptr) {
        uint64_t best = 0; float mx = 0; int cnt = 0;
        check(myslam_lcddb_query(h_, d.data(), curKFId, th2_, &best, &mx, &cnt), "myslam_lcddb_query");
        if (maxScore) *maxScore = mx;
        if (mx < th1_ || cnt > 3) return false;
        loopKFId = best;
        return true;
    }

   private:
    myslam_lcddb* h_ = nullptr;
    float th1_, th2_;
};

struct LocalBA {               // flat-array form of the graph Backend::OptimizeActiveMap builds (backend.cpp:139-206)
    std::vector<double> poses, points, obs;          // nposes x 7 (qx qy qz qw tx ty tz = KeyFrame::Pose()), npts x 3, nedges x 2
    std::vector<int32_t> edge_pose, edge_pt;
    std::vector<uint8_t> fixed;                      // setFixed rule of backend.cpp:175-177
    double fx = 0, fy = 0, cx = 0, cy = 0, huber_delta = 5.991;   // backend.cpp:155,199
    std::vector<double> Hpp, Hll, Hpl, bp, bl, chi2;
    // --- Map -> flat arrays (backend.cpp:139-206): copy the Map's tables into the *Table members, call Flatten(), and the arrays above
    // are filled; after OptimizeActiveMap() pose slot p belongs to key-frame kf_ids[pose_src[p]], landmark slot j to map point
    // mp_ids[pt_src[j]] and edge k to observation row edge_src[k] (outlier[k] -> that feature's mbIsOutlier, :234-250)
    std::vector<uint64_t> kf_ids; std::vector<double> kf_pose7;                                       // Map::GetActiveKeyFrames(): mnKFId, Pose()
    std::vector<uint64_t> mp_ids, mp_first_observer_kf; std::vector<uint8_t> mp_outlier; std::vector<double> mp_pos;   // GetActiveMapPoints()
    std::vector<uint64_t> obs_mp_id, obs_kf_id; std::vector<float> obs_uv; std::vector<uint8_t> obs_feat_outlier;    // GetActiveObservations()
    std::vector<int32_t> pose_src, pt_src, edge_src;
    void AddKeyFrame(uint64_t kfId, const double pose7[7]) { kf_ids.push_back(kfId); kf_pose7.insert(kf_pose7.end(), pose7, pose7 + 7); }
    void AddMapPoint(uint64_t id, const double pos[3], bool isOutlier, uint64_t firstObserverKFId) {
        mp_ids.push_back(id); mp_pos.insert(mp_pos.end(), pos, pos + 3); mp_outlier.push_back(isOutlier); mp_first_observer_kf.push_back(firstObserverKFId);
    }
    void AddObservation(uint64_t mpId, uint64_t kfId, float u, float v, bool featureIsOutlier) {
        obs_mp_id.push_back(mpId); obs_kf_id.push_back(kfId); obs_uv.push_back(u); obs_uv.push_back(v); obs_feat_outlier.push_back(featureIsOutlier);
    }
    void Flatten() {
        const int nk = (int)kf_ids.size(), nm = (int)mp_ids.size(), no = (int)obs_mp_id.size();
        pose_src.assign(nk, 0); pt_src.assign(nm, 0); fixed.assign(nm, 0);
        edge_pose.assign(no, 0); edge_pt.assign(no, 0); edge_src.assign(no, 0); obs.assign((size_t)no * 2, 0);
        int32_t L = 0, E = 0;
        check(myslam_ba_flatten_window(kf_ids.data(), nk, mp_ids.data(), mp_outlier.data(), mp_first_observer_kf.data(), nm, obs_mp_id.data(),
                                       obs_kf_id.data(), obs_uv.data(), obs_feat_outlier.data(), no, pose_src.data(), pt_src.data(), &L,
                                       edge_pose.data(), edge_pt.data(), obs.data(), edge_src.data(), &E, fixed.data()), "myslam_ba_flatten_window");
        pt_src.resize(L); fixed.resize(L); edge_pose.resize(E); edge_pt.resize(E); edge_src.resize(E); obs.resize((size_t)E * 2);
        poses.resize((size_t)nk * 7); points.resize((size_t)L * 3);
        for (int p = 0; p < nk; p++) std::copy(kf_pose7.begin() + 7 * pose_src[p], kf_pose7.begin() + 7 * pose_src[p] + 7, poses.begin() + 7 * p);
        for (int j = 0; j < L; j++) std::copy(mp_pos.begin() + 3 * pt_src[j], mp_pos.begin() + 3 * pt_src[j] + 3, points.begin() + 3 * j);
    }
    void Build() {
        const int P = (int)(poses.size() / 7), L = (int)(points.size() / 3), E = (int)edge_pose.size();
        Hpp.assign((size_t)P * 36, 0); Hll.assign((size_t)L * 9, 0); Hpl.assign((size_t)E * 18, 0);
        bp.assign((size_t)P * 6, 0); bl.assign((size_t)L * 3, 0); chi2.assign(E, 0);
        check(myslam_ba_build(poses.data(), P, points.data(), L, edge_pose.data(), edge_pt.data(), obs.data(), E,
                              fixed.empty() ? nullptr : fixed.data(), fx, fy, cx, cy, huber_delta, Hpp.data(), Hll.data(), Hpl.data(),
                              bp.data(), bl.data(), chi2.data()), "myslam_ba_build");
    }
    // optimizer.optimize(n) of backend.cpp:212-214 on the device: poses / points are updated in place
    int Optimize(int iterations = 10, double* final_chi2 = nullptr) {
        const int P = (int)(poses.size() / 7), L = (int)(points.size() / 3), E = (int)edge_pose.size();
        int it = 0;
        check(myslam_ba_optimize(poses.data(), P, points.data(), L, edge_pose.data(), edge_pt.data(), obs.data(), E,
                                 fixed.empty() ? nullptr : fixed.data(), fx, fy, cx, cy, huber_delta, iterations, final_chi2, &it),
              "myslam_ba_optimize");
        return it;
    }
    // the whole solve stage, backend.cpp:208-243: rounds of optimize(10) until the inlier ratio passes 0.5, then the outlier
    // flags (outlier[k] != 0 <=> the reference sets feature->mbIsOutlier and detaches it, :232-249); chi2[k] = edge->chi2()
    std::vector<uint8_t> outlier;
    int OptimizeActiveMap(double chi2_th = 5.991, int max_rounds = 5, int iters_per_round = 10) {
        const int P = (int)(poses.size() / 7), L = (int)(points.size() / 3), E = (int)edge_pose.size();
        chi2.assign(E, 0); outlier.assign(E, 0);
        int rounds = 0, nout = 0;
        check(myslam_ba_optimize_active_map(poses.data(), P, points.data(), L, edge_pose.data(), edge_pt.data(), obs.data(), E,
                                            fixed.empty() ? nullptr : fixed.data(), fx, fy, cx, cy, huber_delta, chi2_th, max_rounds,
                                            iters_per_round, chi2.data(), outlier.data(), &rounds, &nout),
              "myslam_ba_optimize_active_map");
        return nout;
    }
};

// flat-array form of the graph LoopClosing::PoseGraphOptimization builds (loopclosing.cpp:546-601) and of its write-back (:612-640)
struct PoseGraph {
    std::vector<double> poses;                       // nKF x 7 (qx qy qz qw tx ty tz) = KeyFrame::Pose() of every key-frame, index = position in the list
    std::vector<uint8_t> fixed;                      // :557-562: active key-frames, the loop key-frame, key-frame 0
    std::vector<int32_t> edge_v0, edge_v1;           // EdgePoseGraph vertices 0 / 1 (:581-582, :594-595)
    std::vector<double> meas;                        // nedges x 7: mRelativePoseToLastKF / mRelativePoseToLoopKF (:583, :596)
    void AddKeyFrame(const double pose7[7], bool isFixed) { poses.insert(poses.end(), pose7, pose7 + 7); fixed.push_back(isFixed); }
    void AddEdge(int kf, int otherKf, const double relPose7[7]) { edge_v0.push_back(kf); edge_v1.push_back(otherKf); meas.insert(meas.end(), relPose7, relPose7 + 7); }
    // optimizer.initializeOptimization(); optimizer.optimize(20) (:605-606); poses are replaced by the optimised estimates (:636-638)
    int Optimize(int iterations = 20, double* final_chi2 = nullptr) {
        int it = 0;
        check(myslam_pose_graph_optimize(poses.data(), (int)(poses.size() / 7), fixed.data(), edge_v0.data(), edge_v1.data(), meas.data(),
                                         (int)edge_v0.size(), iterations, final_chi2, &it), "myslam_pose_graph_optimize");
        return it;
    }
    // :621-633: points (n x 3, in/out) keep their camera-frame position in first_kf[i] (index into poses; < 0 = skip)
    static void CorrectMapPoints(const std::vector<double>& old_poses, const std::vector<double>& new_poses, const std::vector<int32_t>& first_kf,
                                 std::vector<double>& points) {
        check(myslam_correct_map_points(old_poses.data(), new_poses.data(), (int)(old_poses.size() / 7), first_kf.data(), points.data(),
                                        (int)first_kf.size()), "myslam_correct_map_points");
    }
};

// cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(11,11), 3, TermCriteria(COUNT+EPS,30,0.01),
// OPTFLOW_USE_INITIAL_FLOW) as Frontend::TrackLastFrame / FindFeaturesInRight call it (src/frontend.cpp:150-153, 358-361)
struct Point2f { float x, y; };
class PyrLKTracker {
    myslam_lk* h_ = nullptr;
public:
    explicit PyrLKTracker(int win = 11, int maxLevel = 3, int maxCount = 30, float epsilon = 0.01f, float minEigThreshold = 1e-4f) {
        check(myslam_lk_create(&h_, win, maxLevel, maxCount, epsilon, minEigThreshold), "myslam_lk_create");
    }
    ~PyrLKTracker() { if (h_) myslam_lk_destroy(h_); }
    PyrLKTracker(const PyrLKTracker&) = delete; PyrLKTracker& operator=(const PyrLKTracker&) = delete;
    // nextPts carries the initial flow on entry (the reference always provides one)
    void calcOpticalFlowPyrLK(const ImageView& prevImg, const ImageView& nextImg, const std::vector<Point2f>& prevPts,
                              std::vector<Point2f>& nextPts, std::vector<uint8_t>& status, std::vector<float>& err) {
        const int n = (int)prevPts.size();
        if ((int)nextPts.size() != n) nextPts = prevPts;
        status.assign(n, 0); err.assign(n, 0.f);
        if (n == 0) return;
        check(myslam_lk_track(h_, prevImg.data, nextImg.data, prevImg.rows, prevImg.cols, prevImg.step, nextImg.step,
                              reinterpret_cast<const float*>(prevPts.data()), reinterpret_cast<float*>(nextPts.data()), n, status.data(), err.data()),
              "myslam_lk_track");
    }
    // the same with the handle's two-image cache (myslam_lk_track_cached): a non-zero token names an image whose bytes never change
    void calcOpticalFlowPyrLK(const ImageView& prevImg, uint64_t prevToken, const ImageView& nextImg, uint64_t nextToken, const std::vector<Point2f>& prevPts,
                              std::vector<Point2f>& nextPts, std::vector<uint8_t>& status, std::vector<float>& err) {
        const int n = (int)prevPts.size();
        if ((int)nextPts.size() != n) nextPts = prevPts;
        status.assign(n, 0); err.assign(n, 0.f);
        if (n == 0) return;
        check(myslam_lk_track_cached(h_, prevImg.data, prevToken, nextImg.data, nextToken, prevImg.rows, prevImg.cols, prevImg.step, nextImg.step,
                                     reinterpret_cast<const float*>(prevPts.data()), reinterpret_cast<float*>(nextPts.data()), n, status.data(), err.data()),
              "myslam_lk_track_cached");
    }
    // asynchronous upload + pyramid of the image the next cached call will name by `token` (it must stay valid and unchanged until then)
    void Prefetch(const ImageView& img, uint64_t token) { check(myslam_lk_prefetch(h_, img.data, token, img.rows, img.cols, img.step), "myslam_lk_prefetch"); }
};

// the g2o stage of Frontend::EstimateCurrentPose (src/frontend.cpp:176-276); returns features.size() - cntOutliers.
// preOptimize = 1 gives LoopClosing::OptimizeCurrentPose (src/loopclosing.cpp:339-433)
inline int EstimateCurrentPose(double pose_qt[7], const std::vector<double>& mapPoints /*n x 3*/, const std::vector<double>& pixels /*n x 2*/,
                               double fx, double fy, double cx, double cy, std::vector<uint8_t>& isOutlier,
                               double chi2_th = 5.991, int numIterations = 4, int optimizeIters = 10, int preOptimize = 0) {
    const int n = (int)(mapPoints.size() / 3);
    isOutlier.assign(n, 0);
    int inl = 0;
    check(myslam_pose_only_optimize(pose_qt, mapPoints.data(), pixels.data(), n, fx, fy, cx, cy, chi2_th, numIterations, optimizeIters,
                                    preOptimize, n ? isOutlier.data() : nullptr, &inl), "myslam_pose_only_optimize");
    return inl;
}

// cv::solvePnPRansac(points3d, points2d, K, cv::Mat(), rvec, tvec, false, 100, 5.991, 0.99) + cv::Rodrigues as
// LoopClosing::ComputeCorrectPose uses them (src/loopclosing.cpp:262-272).  pose7 = (qx qy qz qw tx ty tz) of SE3d(R, t).
// false where OpenCV returns false (fewer than 5 matches, no model) — the reference wraps the call in try / catch and gives up.
struct Point3f { float x, y, z; };
inline bool solvePnPRansac(const std::vector<Point3f>& objectPoints, const std::vector<Point2f>& imagePoints, double fx, double fy, double cx,
                           double cy, double pose7[7], std::vector<uint8_t>* inliers = nullptr, int iterationsCount = 100,
                           float reprojectionError = 5.991f, double confidence = 0.99) {
    if (objectPoints.size() != imagePoints.size()) throw std::runtime_error("solvePnPRansac: size mismatch");
    const int n = (int)objectPoints.size();
    std::vector<uint8_t> mask((size_t)std::max(n, 1));
    int ninl = 0;
    const int rc = myslam_solve_pnp_ransac(reinterpret_cast<const float*>(objectPoints.data()), reinterpret_cast<const float*>(imagePoints.data()), n,
                                           fx, fy, cx, cy, iterationsCount, (double)reprojectionError, confidence, pose7, mask.data(), &ninl);
    if (rc == MYSLAM_ERR_UNSUPPORTED) return false;
    check(rc, "myslam_solve_pnp_ransac");
    if (inliers) { mask.resize(n); *inliers = mask; }
    return true;
}

}  // namespace myslam
