// myslam_system.hpp — the reference's Frontend / Backend / LoopClosing / Map as ONE sequential C++ schedule over the C ABI
// (include/myslam_hip.h) through the facade classes of myslam_hip.hpp.  It is the compiled twin of the package's chain.py (same
// decisions, same order of operator calls; tests/test_gpu_runner.py requires the two to write the same trajectory) and what
// app/run_kitti_stereo.cpp — BASELINE configs[0]'s entry point as a compiled program — is built from.
//
// What follows the reference (plain objects, shared_ptr for what the reference shares):
//   Frontend::GrabStereoImage / Track / TrackLastFrame / EstimateCurrentPose / StereoInit / DetectFeatures / FindFeaturesInRight /
//   BuildInitMap / InsertKeyFrame / TriangulateNewPoints                                   src/frontend.cpp:41-488
//   KeyFrame::CreateKF                                                                     src/keyframe.cpp:6-44
//   Map::InsertKeyFrame / RemoveOldActiveKeyframe / RemoveOldActiveMapPoints / RemoveAllOutlierMapPoints / RemoveMapPoint
//                                                                                          src/map.cpp:15-167
//   MapPoint::Add / Remove(Active)Observation                                              src/mappoint.cpp:21-57
//   Backend::ProcessNewKeyFrame / OptimizeActiveMap                                        src/backend.cpp:105-266
//   LoopClosing::InsertNewKeyFrame / LoopClosingRun / ProcessNewKF / DetectLoop / MatchFeatures / ComputeCorrectPose /
//   OptimizeCurrentPose / LoopCorrect / LoopLocalFusion / PoseGraphOptimization / AddToDatabase        src/loopclosing.cpp:51-687
//   System::GetCamera / SaveTrajectory / SaveLoopEdges                                     src/system.cpp:101-224
//
// What a sequential program has to decide (the reference runs three threads whose interleaving is a race) — as chain.py:
//   * a new key-frame goes through Map::InsertKeyFrame, LoopClosing::InsertNewKeyFrame, OptimizeActiveMap and then the loop closer's turn
//     before the next frame is tracked;
//   * DeepLCD blurs the key-frame's image in place and that image IS the frame's left image (cv::Mat copies share pixels): the next
//     TrackLastFrame sees the blurred pixels (`lcdBlurReachesTracker`, default true; false = the frontend wins the race);
//   * unordered_map iteration order is taken as ascending id.
// Two places where this host deliberately does NOT do what the reference's text does (chain.py: the same):
//   * LoopLocalFusion, a matched pair whose two features already share ONE map point: skipped.  The reference (loopclosing.cpp:516-527)
//     would re-add that point's observations to itself and then RemoveMapPoint() the point both key-frames use — it flags its own live
//     landmark as an outlier; a sequential host has no later pass that would notice, so the pair is left alone;
//   * OptimizeActiveMap, an active map point with no observation at all: an outlier one is skipped first, as backend.cpp:163 does before
//     it ever looks at GetObservations().front() (:175); a NON-outlier one would be undefined behaviour there (front() of an empty list)
//     and is an error here.
// Every arithmetic operator is a call into libmyslam_hip.so; what is computed here is bookkeeping and a handful of 4x4 products.
#pragma once
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "myslam_hip.hpp"

namespace myslam {

// ---- SE3 as (qx qy qz qw tx ty tz) = Tcw, and as a row-major 4x4 -------------------------------------------------------------------
struct Pose7 { double v[7] = {0, 0, 0, 1, 0, 0, 0}; };
struct Mat4 {
    double m[4][4];
    static Mat4 Identity() { Mat4 a; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) a.m[i][j] = i == j ? 1.0 : 0.0; return a; }
};
inline Mat4 operator*(const Mat4& a, const Mat4& b) {
    Mat4 c;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = a.m[i][0] * b.m[0][j];          // ascending k, plain multiplies and adds: chain.py's mm() uses the same order
            for (int k = 1; k < 4; k++) s += a.m[i][k] * b.m[k][j];
            c.m[i][j] = s;
        }
    return c;
}
inline Mat4 T_of(const Pose7& p) {
    const double n = std::sqrt(p.v[0] * p.v[0] + p.v[1] * p.v[1] + p.v[2] * p.v[2] + p.v[3] * p.v[3]);
    const double x = p.v[0] / n, y = p.v[1] / n, z = p.v[2] / n, w = p.v[3] / n;
    Mat4 T = Mat4::Identity();
    T.m[0][0] = 1 - 2 * (y * y + z * z); T.m[0][1] = 2 * (x * y - z * w);     T.m[0][2] = 2 * (x * z + y * w);
    T.m[1][0] = 2 * (x * y + z * w);     T.m[1][1] = 1 - 2 * (x * x + z * z); T.m[1][2] = 2 * (y * z - x * w);
    T.m[2][0] = 2 * (x * z - y * w);     T.m[2][1] = 2 * (y * z + x * w);     T.m[2][2] = 1 - 2 * (x * x + y * y);
    T.m[0][3] = p.v[4]; T.m[1][3] = p.v[5]; T.m[2][3] = p.v[6];
    return T;
}
inline Pose7 p7_of(const Mat4& T) {
    const double (*R)[4] = T.m;
    const double t = R[0][0] + R[1][1] + R[2][2];
    double q[4];
    if (t > 0) {
        const double s = std::sqrt(t + 1.0) * 2;
        q[0] = (R[2][1] - R[1][2]) / s; q[1] = (R[0][2] - R[2][0]) / s; q[2] = (R[1][0] - R[0][1]) / s; q[3] = 0.25 * s;
    } else if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) {
        const double s = std::sqrt(1.0 + R[0][0] - R[1][1] - R[2][2]) * 2;
        q[0] = 0.25 * s; q[1] = (R[0][1] + R[1][0]) / s; q[2] = (R[0][2] + R[2][0]) / s; q[3] = (R[2][1] - R[1][2]) / s;
    } else if (R[1][1] > R[2][2]) {
        const double s = std::sqrt(1.0 + R[1][1] - R[0][0] - R[2][2]) * 2;
        q[0] = (R[0][1] + R[1][0]) / s; q[1] = 0.25 * s; q[2] = (R[1][2] + R[2][1]) / s; q[3] = (R[0][2] - R[2][0]) / s;
    } else {
        const double s = std::sqrt(1.0 + R[2][2] - R[0][0] - R[1][1]) * 2;
        q[0] = (R[0][2] + R[2][0]) / s; q[1] = (R[1][2] + R[2][1]) / s; q[2] = 0.25 * s; q[3] = (R[1][0] - R[0][1]) / s;
    }
    Pose7 p;
    const double sg = q[3] >= 0 ? 1.0 : -1.0;
    for (int i = 0; i < 4; i++) p.v[i] = sg * q[i];
    p.v[4] = T.m[0][3]; p.v[5] = T.m[1][3]; p.v[6] = T.m[2][3];
    return p;
}
inline Mat4 T_inv(const Mat4& T) {
    Mat4 I = Mat4::Identity();
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) I.m[i][j] = T.m[j][i];
        double s = -T.m[0][i] * T.m[0][3];
        for (int k = 1; k < 3; k++) s += -T.m[k][i] * T.m[k][3];
        I.m[i][3] = s;
    }
    return I;
}
// |Sophus::SE3d::log()|: Map::RemoveOldActiveKeyframe's distance (map.cpp:88), the `error > 1` test of ComputeCorrectPose (loopclosing.cpp:283)
inline double se3_log_norm(const Mat4& T) {
    const double (*R)[4] = T.m;
    const double t[3] = {T.m[0][3], T.m[1][3], T.m[2][3]};
    const double c = std::min(1.0, std::max(-1.0, (R[0][0] + R[1][1] + R[2][2] - 1.0) / 2.0));
    const double th = std::acos(c);
    const double w[3] = {R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1]};
    double om[3];
    if (th < 1e-10) {
        for (int i = 0; i < 3; i++) om[i] = 0.5 * w[i];
    } else if (M_PI - th < 1e-6) {                 // near pi: axis from the diagonal
        double A[3][3], ax[3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] = (R[i][j] + (i == j ? 1.0 : 0.0)) / 2.0;
        for (int i = 0; i < 3; i++) ax[i] = std::sqrt(std::max(A[i][i], 0.0));
        int k = 0;
        for (int i = 1; i < 3; i++) if (ax[i] > ax[k]) k = i;
        const double d = std::max(ax[k], 1e-300);
        double col[3] = {A[0][k] / d, A[1][k] / d, A[2][k] / d};
        const double nn = std::sqrt(col[0] * col[0] + col[1] * col[1] + col[2] * col[2]);
        for (int i = 0; i < 3; i++) col[i] /= nn;
        if (col[0] * w[0] + col[1] * w[1] + col[2] * w[2] < 0) for (int i = 0; i < 3; i++) col[i] = -col[i];
        for (int i = 0; i < 3; i++) om[i] = th * col[i];
    } else {
        const double f = th / (2.0 * std::sin(th));
        for (int i = 0; i < 3; i++) om[i] = f * w[i];
    }
    const double Om[3][3] = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}};
    double Om2[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = Om[i][0] * Om[0][j]; for (int k = 1; k < 3; k++) s += Om[i][k] * Om[k][j]; Om2[i][j] = s; }
    double coef;
    if (th < 1e-10) coef = 1.0 / 12.0;
    else { const double h = 0.5 * th; coef = (1.0 - th * std::cos(h) / (2.0 * std::sin(h))) / (th * th); }
    double u[3];
    for (int i = 0; i < 3; i++) {
        double V[3];
        for (int j = 0; j < 3; j++) V[j] = ((i == j ? 1.0 : 0.0) - 0.5 * Om[i][j]) + coef * Om2[i][j];
        u[i] = (V[0] * t[0] + V[1] * t[1]) + V[2] * t[2];
    }
    return std::sqrt((u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) + (om[0] * om[0] + om[1] * om[1] + om[2] * om[2]));
}

// System::GetCamera (system.cpp:101-146): BOTH cameras take the Camera.right.* keys (reference quirk 8), every value passes through a
// float, baseline = bf / fx in float
struct StereoCamera {
    double fx = 0, fy = 0, cx = 0, cy = 0, bf = 0, baseline = 0;
    template <class Cfg>
    static StereoCamera FromConfig(const Cfg& cfg) {
        auto f = [&](const char* k) { return (float)cfg.template Get<double>(k); };
        const float fx = f("Camera.right.fx"), fy = f("Camera.right.fy"), cx = f("Camera.right.cx"), cy = f("Camera.right.cy"), bf = f("Camera.bf");
        StereoCamera c;
        c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy; c.bf = bf; c.baseline = (double)(float)(bf / fx);
        return c;
    }
};

// the reference's config/stereo/gray/KITTI00-02.yaml values; the YAML given on the command line overrides them
struct SystemConfig {
    int nInitGood = 100, nTrackingGood = 50, nTrackingBad = 10;            // numFeatures.*
    int nInitFeatures = 300, nNewFeatures = 100, nLevels = 8, iniThFAST = 20, minThFAST = 7;   // ORBextractor.*
    float scaleFactor = 1.2f;
    int activeMapSize = 7;                                                 // Map.activeMap.size
    float lcdThresHigh = 0.94f, lcdThresLow = 0.92f;                       // LCD.similarityScoreThreshold.*
    int lcdMinDatabase = 50;                                               // LCD.nDatabaseMinSize
    int kfEvery = 0;                       // > 0 replaces the inlier-count rule by "every n-th frame" (chain.py's option)
    bool lcdBlurReachesTracker = true;
    double correctThreshold = 1.0;         // loopclosing.cpp:283-288
    template <class Cfg>
    void Override(const Cfg& c) {
        auto I = [&](const char* k, int& v) { if (c.Has(k)) v = c.template Get<int>(k); };
        auto F = [&](const char* k, float& v) { if (c.Has(k)) v = (float)c.template Get<double>(k); };
        I("numFeatures.initGood", nInitGood); I("numFeatures.trackingGood", nTrackingGood); I("numFeatures.trackingBad", nTrackingBad);
        I("ORBextractor.nInitFeatures", nInitFeatures); I("ORBextractor.nNewFeatures", nNewFeatures); I("ORBextractor.nLevels", nLevels);
        I("ORBextractor.iniThFAST", iniThFAST); I("ORBextractor.minThFAST", minThFAST); F("ORBextractor.scaleFactor", scaleFactor);
        I("Map.activeMap.size", activeMapSize); I("LCD.nDatabaseMinSize", lcdMinDatabase);
        F("LCD.similarityScoreThreshold.high", lcdThresHigh); F("LCD.similarityScoreThreshold.low", lcdThresLow);
    }
};

// ---- the map's objects -------------------------------------------------------------------------------------------------------------
struct Image {                              // cv::Mat CV_8UC1; shared_ptr<Image> copies share pixels as cv::Mat copies do
    std::vector<uint8_t> px; int rows = 0, cols = 0;
    uint64_t token = 0;                     // names these BYTES for the tracker's image cache (myslam_lk_track_cached): new token whenever px changes
    ImageView view() { return ImageView{px.data(), rows, cols, cols}; }
};
struct MapPoint; struct KeyFrame;
struct Feature {                            // include/myslam/feature.h:14-35
    float x, y;                             // mkpPosition.pt
    std::shared_ptr<MapPoint> mp;           // mpMapPoint (weak_ptr in the reference: Live() is lock() != nullptr)
    KeyFrame* kf = nullptr;                 // mpKF (key-frames are never destroyed)
    bool outlier = false;                   // mbIsOutlier
    Feature(float x_, float y_) : x(x_), y(y_) {}
    inline MapPoint* Live() const;
};
struct MapPoint {                           // include/myslam/mappoint.h:13-61
    unsigned long id; double pos[3];
    std::vector<Feature*> obs, activeObs;   // mlistObservations, mlistActiveObservations (features of key-frames only)
    bool outlier = false;
    bool alive = true;                      // false once the Map dropped its shared_ptr (the only owner): every weak_ptr expires
};
inline MapPoint* Feature::Live() const { return (mp && mp->alive) ? mp.get() : nullptr; }
using FeatureList = std::vector<std::shared_ptr<Feature>>;
struct Frame {                              // include/myslam/frame.h:12-49
    unsigned long id = 0; double ts = 0;
    std::shared_ptr<Image> L, R;
    FeatureList feats;                      // mvpFeaturesLeft
    std::vector<Point2f> right; std::vector<uint8_t> hasRight;   // mvpFeaturesRight (nullptr where LK failed)
    Mat4 rel = Mat4::Identity();            // RelativePose(): pose relative to the reference key-frame
};
struct KeyFrame {                           // include/myslam/keyframe.h:14-60
    unsigned long id = 0, frameId = 0; double ts = 0;
    std::shared_ptr<Image> img;             // mImageLeft: the SAME pixels as the frame's mLeftImg
    FeatureList feats;                      // the shared Feature objects
    Pose7 pose;
    KeyFrame* lastKF = nullptr; Pose7 relToLast;
    KeyFrame* loopKF = nullptr; Pose7 relToLoop;
    DeepLCD::DescrVector descr{};           // mpDescrVector
    KeyFrameFeatures pyr;                   // mvPyramidKeyPoints + mORBDescriptors
};

class StereoSystem {
   public:
    enum Status { INITING, TRACKING_GOOD, TRACKING_BAD, LOST };
    struct Counters { long lkInitFromProjection = 0, lkInitFromLast = 0, poseOnly = 0, ba = 0, lcd = 0, detectLoop = 0, pnp = 0, pgo = 0, lkPrefetched = 0;
                      double secLK = 0, secPoseOnly = 0, secKeyFrame = 0, secTrackHost = 0, secPoseHost = 0, secGrab = 0, secRelease = 0, secInit = 0; } stats;     // wall time inside the tracker's two per-frame calls and inside key-frame insertion

    StereoSystem(const StereoCamera& cam, const SystemConfig& cfg, std::unique_ptr<DeepLCD> lcd)
        : K_(cam), c_(cfg),
          orbInit_(cfg.nInitFeatures, cfg.scaleFactor, cfg.nLevels, cfg.iniThFAST, cfg.minThFAST),     // Frontend::_mpORBextractorInit (frontend.cpp:34)
          orb_(cfg.nNewFeatures, cfg.scaleFactor, cfg.nLevels, cfg.iniThFAST, cfg.minThFAST),          // System::_mpORBextractor, shared by Frontend and LoopClosing
          lcd_(std::move(lcd)), db_(64, cfg.lcdThresHigh, cfg.lcdThresLow) {}

    // Frontend::GrabStereoImage (frontend.cpp:41-80); false = the tracker is LOST (the reference quits)
    // nextLeft (optional): the left image of the FOLLOWING frame when the caller already has it — it is uploaded and down-sampled on the
    // tracker's stream while this frame's pose is optimised (the reference reads one pair per call, app/run_kitti_stereo.cpp:66-67; a reader
    // that decodes ahead can hand the next image over).  Results do not depend on it.
    bool GrabStereoImage(std::shared_ptr<Image> left, std::shared_ptr<Image> right, double timestamp, std::shared_ptr<Image> nextLeft = nullptr) {
        const auto tg0 = std::chrono::steady_clock::now();
        if (left && !left->token) left->token = ++imageTokens_;
        if (nextLeft && !nextLeft->token) nextLeft->token = ++imageTokens_;
        nextLeft_ = std::move(nextLeft);
        cur_ = std::make_shared<Frame>();
        cur_->id = nextFrameId_++; cur_->ts = timestamp; cur_->L = std::move(left); cur_->R = std::move(right);
        if (status_ == INITING) {                    // (the first call also pays for the library's lazy set-up: code objects, recorded graphs, tables)
            StereoInit();
            stats.secInit += std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count();
        }
        else if (status_ == TRACKING_GOOD || status_ == TRACKING_BAD) Track();
        else return false;
        framePoses.push_back(refKF_ ? p7_of(cur_->rel * T_of(refKF_->pose)) : Pose7());
        const auto tr0 = std::chrono::steady_clock::now();
        last_ = cur_;                                // releases the frame before last (its images, unless a key-frame holds them)
        const auto tg1 = std::chrono::steady_clock::now();
        stats.secRelease += std::chrono::duration<double>(tg1 - tr0).count(); stats.secGrab += std::chrono::duration<double>(tg1 - tg0).count();
        return true;
    }

    // System::SaveTrajectory / SaveLoopEdges (system.cpp:153-224), through the library's writers
    void Save(const std::string& dir) const {
        std::vector<uint64_t> ids; std::vector<double> ts, poses;
        for (const auto& kv : allKFs_) { ids.push_back(kv.first); ts.push_back(kv.second->ts); poses.insert(poses.end(), kv.second->pose.v, kv.second->pose.v + 7); }
        check(myslam_io_save_trajectory((dir + "/trajectory.txt").c_str(), ids.data(), ts.data(), poses.data(), (int)ids.size()), "myslam_io_save_trajectory");
        std::vector<uint64_t> ci, li; std::vector<double> ct, lt, cp, lp;
        for (const auto& kv : allKFs_) {
            const KeyFrame* k = kv.second.get();
            if (!k->loopKF) continue;
            ci.push_back(k->id); ct.push_back(k->ts); cp.insert(cp.end(), k->pose.v, k->pose.v + 7);
            li.push_back(k->loopKF->id); lt.push_back(k->loopKF->ts); lp.insert(lp.end(), k->loopKF->pose.v, k->loopKF->pose.v + 7);
        }
        check(myslam_io_save_loop_edges((dir + "/loopEdges.txt").c_str(), ci.data(), ct.data(), cp.data(), li.data(), lt.data(), lp.data(), (int)ci.size()),
              "myslam_io_save_loop_edges");
    }

    size_t NumKeyFrames() const { return allKFs_.size(); }
    size_t NumMapPoints() const { return allMPs_.size(); }
    size_t NumLoops() const { return loops_.size(); }
    size_t DatabaseSize() const { return db_.size(); }
    std::vector<unsigned long> keyFrameFrames;      // frame id of every key-frame
    std::vector<Pose7> framePoses;                  // Tcw of every tracked frame, as estimated when the frame was processed
    std::vector<int> poseOnlyInliers;               // per tracked frame

   private:
    // ------------------------------------------------------------ cameras (camera.cpp:7-45; left extrinsics = identity, right = (-baseline, 0, 0))
    void World2Pixel(const double pw[3], const Mat4& Tcw, bool right, double uv[2]) const {
        double pc[3];
        for (int i = 0; i < 3; i++) pc[i] = (Tcw.m[i][0] * pw[0] + Tcw.m[i][1] * pw[1] + Tcw.m[i][2] * pw[2]) + Tcw.m[i][3];
        if (right) pc[0] = pc[0] + -K_.baseline;
        uv[0] = K_.fx * pc[0] / pc[2] + K_.cx; uv[1] = K_.fy * pc[1] / pc[2] + K_.cy;
    }

    // ------------------------------------------------------------ Frontend
    bool StereoInit() {                              // frontend.cpp:281-295
        DetectFeatures();
        if (FindFeaturesInRight() < c_.nInitGood) return false;
        BuildInitMap();
        status_ = TRACKING_GOOD;
        return true;
    }

    void Track() {                                   // frontend.cpp:85-124
        cur_->rel = relMotion_ * last_->rel;
        TrackLastFrame();
        const int nInl = EstimateCurrentPose();
        poseOnlyInliers.push_back(nInl);
        status_ = nInl > c_.nTrackingGood ? TRACKING_GOOD : (nInl > c_.nTrackingBad ? TRACKING_BAD : LOST);
        relMotion_ = cur_->rel * T_inv(last_->rel);
        const bool insert = c_.kfEvery <= 0 ? status_ == TRACKING_BAD : (status_ != LOST && cur_->id % (unsigned long)c_.kfEvery == 0);
        if (insert) {
            const auto ti0 = std::chrono::steady_clock::now();
            DetectFeatures();
            FindFeaturesInRight();
            TriangulateNewPoints();
            InsertKeyFrame();
            stats.secKeyFrame += std::chrono::duration<double>(std::chrono::steady_clock::now() - ti0).count();
        }
    }

    void TrackLastFrame() {                          // frontend.cpp:129-172
        const auto th0 = std::chrono::steady_clock::now();
        const Mat4 Tcw = cur_->rel * T_of(refKF_->pose);
        const size_t n = last_->feats.size();
        std::vector<Point2f> p0(n), p1(n);
        for (size_t i = 0; i < n; i++) {
            const Feature& f = *last_->feats[i];
            p0[i] = Point2f{f.x, f.y};
            const MapPoint* mp = f.Live();
            if (mp && !mp->outlier) {                // initial flow = the re-projection with the predicted pose
                double uv[2]; World2Pixel(mp->pos, Tcw, false, uv);
                p1[i] = Point2f{(float)uv[0], (float)uv[1]};
                stats.lkInitFromProjection++;
            } else { p1[i] = p0[i]; stats.lkInitFromLast++; }
        }
        std::vector<uint8_t> st; std::vector<float> err;
        auto a = last_->L->view(), b = cur_->L->view();
        // the last frame's image is still on the device under its token (it was this call's `next` one frame ago) unless DeepLCD has blurred
        // it in place since (a key-frame: new token, uploaded again)
        const auto tk0 = std::chrono::steady_clock::now();
        uploader_.Wait();                                                   // the handle serves one thread at a time
        lk_.calcOpticalFlowPyrLK(a, last_->L->token, b, cur_->L->token, p0, p1, st, err);
        const auto tk1 = std::chrono::steady_clock::now();
        stats.secLK += std::chrono::duration<double>(tk1 - tk0).count();
        if (nextLeft_) { uploader_.Post(&lk_, nextLeft_); stats.lkPrefetched++; }     // uploaded by a helper thread beside EstimateCurrentPose
        for (size_t i = 0; i < n; i++)
            if (st[i] && last_->feats[i]->Live()) {  // status && !mpMapPoint.expired()
                auto g = std::make_shared<Feature>(p1[i].x, p1[i].y);
                g->mp = last_->feats[i]->mp;
                cur_->feats.push_back(std::move(g));
            }
        stats.secTrackHost += std::chrono::duration<double>((tk0 - th0) + (std::chrono::steady_clock::now() - tk1)).count();
    }

    int EstimateCurrentPose() {                      // frontend.cpp:176-276
        const auto th0 = std::chrono::steady_clock::now();
        std::vector<Feature*> feats;
        for (auto& f : cur_->feats) if (f->Live() && !f->mp->outlier) feats.push_back(f.get());
        std::vector<double> p3, obs;
        for (Feature* f : feats) { p3.insert(p3.end(), f->mp->pos, f->mp->pos + 3); obs.push_back((double)f->x); obs.push_back((double)f->y); }
        Pose7 pose = p7_of(cur_->rel * T_of(refKF_->pose));
        std::vector<uint8_t> outl;
        const auto tp0 = std::chrono::steady_clock::now();
        const int nInl = myslam::EstimateCurrentPose(pose.v, p3, obs, K_.fx, K_.fy, K_.cx, K_.cy, outl);
        const auto tp1 = std::chrono::steady_clock::now();
        stats.secPoseOnly += std::chrono::duration<double>(tp1 - tp0).count();
        stats.poseOnly++;
        cur_->rel = T_of(pose) * T_inv(T_of(refKF_->pose));
        for (size_t i = 0; i < feats.size(); i++)
            if (outl[i]) {
                Feature* f = feats[i];
                MapPoint* mp = f->Live();
                if (mp && cur_->id - refKF_->frameId <= 2) {       // a map point that fails right after its creation leaves the map
                    mp->outlier = true; outlierMPs_.push_back(mp->id);
                }
                f->mp.reset(); f->outlier = false;
            }
        stats.secPoseHost += std::chrono::duration<double>((tp0 - th0) + (std::chrono::steady_clock::now() - tp1)).count();
        return nInl;
    }

    static int cvRoundf(float v) { return (int)std::lrintf(v); }          // cv::Point2f -> cv::Point

    int DetectFeatures() {                           // frontend.cpp:300-330
        Image& L = *cur_->L;
        std::vector<uint8_t> mask((size_t)L.rows * L.cols, 255);
        for (const auto& f : cur_->feats) {          // cv::rectangle(pt - (20, 20), pt + (20, 20), 0, CV_FILLED)
            const int x0 = std::max(cvRoundf(f->x - 20.f), 0), x1 = std::min(std::max(cvRoundf(f->x + 20.f) + 1, 0), L.cols);
            const int y0 = std::max(cvRoundf(f->y - 20.f), 0), y1 = std::min(std::max(cvRoundf(f->y + 20.f) + 1, 0), L.rows);
            for (int y = y0; y < y1; y++) for (int x = x0; x < x1; x++) mask[(size_t)y * L.cols + x] = 0;
        }
        std::vector<KeyPoint> kps;
        (status_ == INITING ? orbInit_ : orb_).Detect(L.view(), ImageView{mask.data(), L.rows, L.cols, L.cols}, kps);
        for (const KeyPoint& k : kps) cur_->feats.push_back(std::make_shared<Feature>(k.x, k.y));
        return (int)kps.size();
    }

    int FindFeaturesInRight() {                      // frontend.cpp:335-379
        const Mat4 Tcw = refKF_ ? cur_->rel * T_of(refKF_->pose) : Mat4::Identity();
        const size_t n = cur_->feats.size();
        std::vector<Point2f> p0(n), p1(n);
        for (size_t i = 0; i < n; i++) {
            const Feature& f = *cur_->feats[i];
            p0[i] = Point2f{f.x, f.y};
            const MapPoint* mp = f.Live();
            if (mp && !mp->outlier) { double uv[2]; World2Pixel(mp->pos, Tcw, true, uv); p1[i] = Point2f{(float)uv[0], (float)uv[1]}; }
            else p1[i] = p0[i];
        }
        std::vector<float> err;
        auto a = cur_->L->view(), b = cur_->R->view();
        uploader_.Wait();
        lk_.calcOpticalFlowPyrLK(a, b, p0, p1, cur_->hasRight, err);
        cur_->right = p1;
        int cnt = 0;
        for (uint8_t s : cur_->hasRight) cnt += s ? 1 : 0;
        return cnt;
    }

    void Triangulate(const std::vector<int>& idx, std::vector<std::array<double, 3>>& xyz, std::vector<uint8_t>& ok) {
        std::vector<float> xl, yl, xr, yr;
        for (int i : idx) { xl.push_back(cur_->feats[i]->x); yl.push_back(cur_->feats[i]->y); xr.push_back(cur_->right[i].x); yr.push_back(cur_->right[i].y); }
        triangulation(xl, yl, xr, yr, K_.fx, K_.fy, K_.cx, K_.cy, K_.baseline, xyz, ok);      // triangulation() && z > 0 (algorithm.h:16-33, frontend.cpp:401,470)
    }

    void NewMapPoint(const double pos[3], Feature& feat) {
        auto mp = std::make_shared<MapPoint>();
        mp->id = nextMPId_++; mp->pos[0] = pos[0]; mp->pos[1] = pos[1]; mp->pos[2] = pos[2];
        feat.mp = mp;
        allMPs_[mp->id] = mp;                        // Map::InsertMapPoint
    }

    void BuildInitMap() {                            // frontend.cpp:385-417
        std::vector<int> idx;
        for (size_t i = 0; i < cur_->feats.size(); i++) if (cur_->hasRight[i]) idx.push_back((int)i);
        if (!idx.empty()) {
            std::vector<std::array<double, 3>> xyz; std::vector<uint8_t> ok;
            Triangulate(idx, xyz, ok);
            for (size_t j = 0; j < idx.size(); j++) if (ok[j]) NewMapPoint(xyz[j].data(), *cur_->feats[idx[j]]);
        }
        InsertKeyFrame();
    }

    void TriangulateNewPoints() {                    // frontend.cpp:451-488
        const Mat4 Twc = T_inv(cur_->rel * T_of(refKF_->pose));
        std::vector<int> idx;
        for (size_t i = 0; i < cur_->feats.size(); i++) if (!cur_->feats[i]->Live() && cur_->hasRight[i]) idx.push_back((int)i);     // !expired() -> skip
        if (idx.empty()) return;
        std::vector<std::array<double, 3>> xyz; std::vector<uint8_t> ok;
        Triangulate(idx, xyz, ok);
        for (size_t j = 0; j < idx.size(); j++)
            if (ok[j]) {
                double pw[3];
                for (int r = 0; r < 3; r++) pw[r] = (Twc.m[r][0] * xyz[j][0] + Twc.m[r][1] * xyz[j][1] + Twc.m[r][2] * xyz[j][2]) + Twc.m[r][3];
                NewMapPoint(pw, *cur_->feats[idx[j]]);
            }
    }

    void InsertKeyFrame() {                          // frontend.cpp:424-447 + KeyFrame::CreateKF (keyframe.cpp:29-44)
        auto kf = std::make_shared<KeyFrame>();
        kf->id = nextKFId_++; kf->frameId = cur_->id; kf->ts = cur_->ts; kf->img = cur_->L; kf->feats = cur_->feats;
        for (auto& f : kf->feats) {
            f->kf = kf.get();
            if (MapPoint* mp = f->Live()) mp->obs.push_back(f.get());       // MapPoint::AddObservation
        }
        if (status_ != INITING) {
            kf->pose = p7_of(cur_->rel * T_of(refKF_->pose));
            kf->lastKF = refKF_; kf->relToLast = p7_of(cur_->rel);
        }
        refKF_ = kf.get();
        cur_->rel = Mat4::Identity();
        keyFrameFrames.push_back(cur_->id);
        BackendNewKeyFrame(kf);
    }

    // ------------------------------------------------------------ Backend + Map
    void BackendNewKeyFrame(const std::shared_ptr<KeyFrame>& kf) {        // Backend::ProcessNewKeyFrame + the optimisation it triggers (backend.cpp:82-121)
        MapInsertKeyFrame(kf);
        const bool queued = LoopClosingInsertNewKeyFrame(*kf);
        OptimizeActiveMap();
        if (queued) LoopClosingTurn(*kf);
    }

    void MapInsertKeyFrame(const std::shared_ptr<KeyFrame>& kf) {         // map.cpp:15-45
        mapCurKF_ = kf.get();
        allKFs_[kf->id] = kf; activeKFs_[kf->id] = kf.get();
        for (auto& f : kf->feats)
            if (MapPoint* mp = f->Live()) { mp->activeObs.push_back(f.get()); activeMPs_[mp->id] = f->mp; }      // AddActiveObservation
        if ((int)activeKFs_.size() > c_.activeMapSize) { RemoveOldActiveKeyframe(); RemoveOldActiveMapPoints(); }
    }

    void RemoveOldActiveKeyframe() {                 // map.cpp:75-120
        const Mat4 Twc = T_inv(T_of(mapCurKF_->pose));
        double maxDis = 0.0, minDis = 9999.0; unsigned long maxId = 0, minId = 0;
        for (const auto& kv : activeKFs_) {
            if (kv.second == mapCurKF_) continue;
            const double dis = se3_log_norm(T_of(kv.second->pose) * Twc);
            if (dis > maxDis) { maxDis = dis; maxId = kv.first; }
            else if (dis < minDis) { minDis = dis; minId = kv.first; }
        }
        KeyFrame* gone = minDis < 0.2 ? activeKFs_.at(minId) : activeKFs_.at(maxId);
        activeKFs_.erase(gone->id);
        for (auto& f : gone->feats) if (MapPoint* mp = f->Live()) RemoveActiveObservation(*mp, f.get());
    }
    static void RemoveActiveObservation(MapPoint& mp, Feature* f) {       // mappoint.cpp:36-45
        for (size_t i = 0; i < mp.activeObs.size(); i++) if (mp.activeObs[i] == f) { mp.activeObs.erase(mp.activeObs.begin() + i); break; }
    }
    static void RemoveObservation(MapPoint& mp, Feature* f) {             // mappoint.cpp:48-58
        for (size_t i = 0; i < mp.obs.size(); i++) if (mp.obs[i] == f) { mp.obs.erase(mp.obs.begin() + i); f->mp.reset(); break; }
    }
    void RemoveOldActiveMapPoints() {                // map.cpp:124-137
        for (auto it = activeMPs_.begin(); it != activeMPs_.end();) it = it->second->activeObs.empty() ? activeMPs_.erase(it) : std::next(it);
    }
    void RemoveAllOutlierMapPoints() {               // map.cpp:163-171
        for (unsigned long id : outlierMPs_) {
            auto it = allMPs_.find(id);
            if (it != allMPs_.end()) { it->second->alive = false; allMPs_.erase(it); }
            activeMPs_.erase(id);
        }
        outlierMPs_.clear();
    }
    void RemoveMapPoint(MapPoint& mp) {              // map.cpp:141-149
        mp.alive = false;
        const unsigned long id = mp.id;
        activeMPs_.erase(id); allMPs_.erase(id);     // (the caller holds no owning reference: `mp` may dangle after this line)
    }

    void OptimizeActiveMap() {                       // backend.cpp:126-266
        if (activeKFs_.empty() || activeMPs_.empty()) return;
        std::vector<KeyFrame*> kfs; std::vector<MapPoint*> mps;
        for (auto& kv : activeKFs_) kfs.push_back(kv.second);
        for (auto& kv : activeMPs_) mps.push_back(kv.second.get());
        // the graph-build rules of :139-206 live behind the C ABI (LocalBA::Flatten = myslam_ba_flatten_window): skip outlier map points /
        // features, fix the landmarks whose first observer left the window, vertices by id, edges grouped by landmark
        LocalBA ba;
        ba.fx = K_.fx; ba.fy = K_.fy; ba.cx = K_.cx; ba.cy = K_.cy;
        for (KeyFrame* k : kfs) ba.AddKeyFrame(k->id, k->pose.v);
        std::vector<std::pair<MapPoint*, Feature*>> rows;
        for (MapPoint* m : mps) {
            // an outlier point is dropped by Flatten (:163) before :175 could touch obs.front(): only a live point must have an observation
            if (!m->outlier && m->obs.empty() && m->activeObs.empty()) throw std::runtime_error("active map point without observations");
            const unsigned long first = !m->obs.empty() ? m->obs[0]->kf->id : !m->activeObs.empty() ? m->activeObs[0]->kf->id : 0ul;      // GetObservations().front()->mpKF (:175)
            ba.AddMapPoint(m->id, m->pos, m->outlier, first);
            for (Feature* f : m->activeObs) { ba.AddObservation(m->id, f->kf->id, f->x, f->y, f->outlier); rows.emplace_back(m, f); }
        }
        ba.Flatten();
        if (ba.edge_src.empty()) return;
        ba.OptimizeActiveMap();
        stats.ba++;
        std::vector<std::shared_ptr<MapPoint>> keep;         // owners of the window's landmarks while flags are written back
        for (int32_t s : ba.pt_src) keep.push_back(activeMPs_.at(mps[s]->id));
        for (size_t e = 0; e < ba.edge_src.size(); e++) {    // :234-250
            MapPoint* mp = rows[ba.edge_src[e]].first; Feature* f = rows[ba.edge_src[e]].second;
            if (ba.outlier[e]) {
                f->outlier = true;
                RemoveActiveObservation(*mp, f);
                RemoveObservation(*mp, f);
                if (mp->obs.empty()) { mp->outlier = true; outlierMPs_.push_back(mp->id); }
                f->mp.reset();
            } else f->outlier = false;
        }
        for (size_t p = 0; p < ba.pose_src.size(); p++) std::copy(ba.poses.begin() + 7 * p, ba.poses.begin() + 7 * p + 7, kfs[ba.pose_src[p]]->pose.v);   // :252-266
        for (size_t j = 0; j < ba.pt_src.size(); j++) std::copy(ba.points.begin() + 3 * j, ba.points.begin() + 3 * j + 3, keep[j]->pos);
        RemoveAllOutlierMapPoints();
        RemoveOldActiveMapPoints();
    }

    // ------------------------------------------------------------ LoopClosing
    bool LoopClosingInsertNewKeyFrame(KeyFrame& kf) {                    // loopclosing.cpp:671-681: the 5 key-frames after a closed loop are skipped
        if (!lastClosedKF_ || kf.id - lastClosedKF_->id > 5) return true;
        kf.img.reset();
        return false;
    }

    void LoopClosingTurn(KeyFrame& kf) {             // one pass of LoopClosingRun's body (loopclosing.cpp:51-77)
        ProcessNewKF(kf);
        bool confirmed = false;
        if ((int)dbKFs_.size() > c_.lcdMinDatabase) {
            if (KeyFrame* loop = DetectLoop(kf)) {
                std::vector<std::pair<int, int>> pairs;
                if (MatchFeatures(loop->pyr, kf.pyr, pairs)) {           // query = loop key-frame, train = current key-frame (:172)
                    confirmed = ComputeCorrectPose(kf, *loop, pairs);
                    if (confirmed) LoopCorrect(kf, *loop);
                }
            }
        }
        if (!confirmed) { dbKFs_[kf.id] = &kf; db_.AddToDatabase(kf.id, kf.descr); }       // AddToDatabase (:651-659)
    }

    void ProcessNewKF(KeyFrame& kf) {                // loopclosing.cpp:83-121
        if (!c_.lcdBlurReachesTracker) kf.img = std::make_shared<Image>(*kf.img);          // the frontend wins the race: the blur hits a private copy
        kf.descr = lcd_->calcDescrOriginalImg(kf.img->view());          // blurs the key-frame's image in place (reference quirk 7)
        kf.img->token = ++imageTokens_;                                 // other bytes now: the tracker's cached copy of the sharp image is stale
        std::vector<KeyPoint> feats(kf.feats.size());
        for (size_t i = 0; i < feats.size(); i++) feats[i] = KeyPoint{kf.feats[i]->x, kf.feats[i]->y, 7.f, -1.f, 0.f, 0, -1};
        kf.pyr.Compute(orb_, kf.img->view(), feats);
        kf.img.reset();                              // LoopClosing.bShowResult 0: mImageLeft.release() (:117-120)
        stats.lcd++;
    }

    KeyFrame* DetectLoop(KeyFrame& kf) {             // loopclosing.cpp:124-161
        unsigned long best = 0;
        stats.detectLoop++;
        if (!db_.DetectLoop(kf.id, kf.descr, best)) return nullptr;
        return dbKFs_.at(best);
    }

    bool ComputeCorrectPose(KeyFrame& kf, KeyFrame& loop, const std::vector<std::pair<int, int>>& pairs) {      // loopclosing.cpp:208-335
        std::vector<std::pair<int, int>> valid;
        for (const auto& p : pairs) if (loop.feats[p.second]->Live()) valid.push_back(p);        // matches without a map point leave the set
        if (valid.size() < 10) return false;
        std::vector<Point3f> p3; std::vector<Point2f> p2; std::vector<double> P3, P2;
        for (const auto& v : valid) {
            const MapPoint* mp = loop.feats[v.second]->mp.get();
            p3.push_back(Point3f{(float)mp->pos[0], (float)mp->pos[1], (float)mp->pos[2]});     // cv::Point3f
            p2.push_back(Point2f{kf.feats[v.first]->x, kf.feats[v.first]->y});
            P3.insert(P3.end(), mp->pos, mp->pos + 3); P2.push_back((double)kf.feats[v.first]->x); P2.push_back((double)kf.feats[v.first]->y);
        }
        Pose7 pose;
        try {                                        // the reference's try / catch around solvePnPRansac (:262-270)
            if (!solvePnPRansac(p3, p2, K_.fx, K_.fy, K_.cx, K_.cy, pose.v)) return false;
        } catch (...) { return false; }
        stats.pnp++;
        std::vector<uint8_t> outl;                   // OptimizeCurrentPose (:339-433): the map points in double, the pixels through toVec2
        myslam::EstimateCurrentPose(pose.v, P3, P2, K_.fx, K_.fy, K_.cx, K_.cy, outl, 5.991, 4, 10, 1);
        std::vector<std::pair<int, int>> kept;
        for (size_t i = 0; i < valid.size(); i++) if (!outl[i]) kept.push_back(valid[i]);
        if (kept.size() < 10) return false;
        needCorrect_ = se3_log_norm(T_of(kf.pose) * T_inv(T_of(pose))) > c_.correctThreshold;
        kf.loopKF = &loop;
        kf.relToLoop = p7_of(T_of(pose) * T_inv(T_of(loop.pose)));
        lastClosedKF_ = &kf;
        loops_.emplace_back(kf.id, loop.id);
        corrected_ = pose; validPairs_ = kept;
        return true;
    }

    void LoopCorrect(KeyFrame& kf, KeyFrame& loop) { // loopclosing.cpp:438-462
        if (!needCorrect_) return;
        LoopLocalFusionStep(kf, loop);
        PoseGraphOptimization(loop);
    }

    void LoopLocalFusionStep(KeyFrame& kf, KeyFrame& loop) {             // loopclosing.cpp:466-533
        std::vector<KeyFrame*> act; std::map<unsigned long, int> slot;
        for (auto& kv : activeKFs_) { slot[kv.first] = (int)act.size(); act.push_back(kv.second); }
        if (!slot.count(kf.id)) throw std::runtime_error("the current key-frame left the active window before its loop was closed");
        std::vector<std::shared_ptr<MapPoint>> mps;
        for (auto& kv : activeMPs_) mps.push_back(kv.second);
        std::vector<int32_t> first; std::vector<double> pts, poses;
        for (auto& m : mps) {
            int s = -1;
            if (!m->activeObs.empty()) { auto it = slot.find(m->activeObs[0]->kf->id); if (it != slot.end()) s = it->second; }
            first.push_back(s); pts.insert(pts.end(), m->pos, m->pos + 3);
        }
        for (KeyFrame* k : act) poses.insert(poses.end(), k->pose.v, k->pose.v + 7);
        LoopLocalFusion(poses, slot[kf.id], corrected_.v, first, pts);
        for (size_t i = 0; i < act.size(); i++) std::copy(poses.begin() + 7 * i, poses.begin() + 7 * i + 7, act[i]->pose.v);
        for (size_t j = 0; j < mps.size(); j++) std::copy(pts.begin() + 3 * j, pts.begin() + 3 * j + 3, mps[j]->pos);
        for (const auto& v : validPairs_) {          // the current key-frame's map points are replaced by the loop key-frame's (:510-532)
            Feature& lf = *loop.feats[v.second]; Feature& cf = *kf.feats[v.first];
            if (!lf.Live()) continue;
            std::shared_ptr<MapPoint> loopMP = lf.mp;
            if (cf.Live()) {
                std::shared_ptr<MapPoint> curMP = cf.mp;
                if (curMP == loopMP) continue;
                const std::vector<Feature*> obs = curMP->obs;
                for (Feature* g : obs) { loopMP->obs.push_back(g); g->mp = loopMP; }
                RemoveMapPoint(*curMP);
            } else cf.mp = loopMP;
        }
    }

    void PoseGraphOptimization(KeyFrame& loop) {     // loopclosing.cpp:537-646
        std::vector<KeyFrame*> kfs; std::map<unsigned long, int> idx;
        for (auto& kv : allKFs_) { idx[kv.first] = (int)kfs.size(); kfs.push_back(kv.second.get()); }
        PoseGraph g;
        for (KeyFrame* k : kfs) g.AddKeyFrame(k->pose.v, activeKFs_.count(k->id) || k->id == loop.id || k->id == 0);
        for (KeyFrame* k : kfs) {
            if (k->lastKF) g.AddEdge(idx[k->id], idx[k->lastKF->id], k->relToLast.v);
            if (k->loopKF) g.AddEdge(idx[k->id], idx[k->loopKF->id], k->relToLoop.v);
        }
        const std::vector<double> old = g.poses;
        g.Optimize(20);
        stats.pgo++;
        std::vector<MapPoint*> mps; std::vector<int32_t> first; std::vector<double> pts;
        for (auto& kv : allMPs_) {                   // map points outside the active map follow the key-frame that first observed them (:612-633)
            MapPoint* m = kv.second.get();
            if (activeMPs_.count(kv.first) || m->obs.empty()) continue;
            auto it = idx.find(m->obs[0]->kf->id);
            mps.push_back(m); first.push_back(it == idx.end() ? -1 : it->second); pts.insert(pts.end(), m->pos, m->pos + 3);
        }
        if (!mps.empty()) {
            PoseGraph::CorrectMapPoints(old, g.poses, first, pts);
            for (size_t j = 0; j < mps.size(); j++) std::copy(pts.begin() + 3 * j, pts.begin() + 3 * j + 3, mps[j]->pos);
        }
        for (size_t i = 0; i < kfs.size(); i++) std::copy(g.poses.begin() + 7 * i, g.poses.begin() + 7 * i + 7, kfs[i]->pose.v);
    }

    StereoCamera K_; SystemConfig c_;
    ORBextractor orbInit_, orb_;
    PyrLKTracker lk_;
    uint64_t imageTokens_ = 0;
    std::shared_ptr<Image> nextLeft_;
    // The upload of the next frame's left image (myslam_lk_prefetch) copies 466 KB out of pageable memory before it returns: a helper thread
    // makes that call, so the tracking thread goes straight on to the pose optimisation.  One job in flight; Wait() before the next call on
    // the handle (it serves one thread at a time) — by then the job is long done.
    class Uploader {
        std::thread th_; std::mutex mu_; std::condition_variable cv_;
        PyrLKTracker* lk_ = nullptr; std::shared_ptr<Image> job_; bool busy_ = false, stop_ = false; std::string error_;
        void Run() {
            for (;;) {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || busy_; });
                if (stop_) return;
                std::shared_ptr<Image> img = job_; PyrLKTracker* t = lk_;
                lk.unlock();
                std::string err;
                try { t->Prefetch(img->view(), img->token); } catch (const std::exception& e) { err = e.what(); }
                lk.lock();
                busy_ = false; job_.reset(); if (!err.empty()) error_ = err;
                cv_.notify_all();
            }
        }
    public:
        ~Uploader() { if (th_.joinable()) { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); th_.join(); } }
        void Post(PyrLKTracker* t, std::shared_ptr<Image> img) {
            Wait();
            { std::lock_guard<std::mutex> lk(mu_); lk_ = t; job_ = std::move(img); busy_ = true; }
            if (!th_.joinable()) th_ = std::thread([this] { Run(); });
            cv_.notify_all();
        }
        void Wait() {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return !busy_; });
            if (!error_.empty()) { const std::string e = error_; error_.clear(); throw std::runtime_error("image upload: " + e); }
        }
    } uploader_;
    std::unique_ptr<DeepLCD> lcd_;
    LoopDatabase db_;
    // Frontend
    Status status_ = INITING;
    std::shared_ptr<Frame> cur_, last_;
    KeyFrame* refKF_ = nullptr;
    Mat4 relMotion_ = Mat4::Identity();              // _mseRelativeMotion
    unsigned long nextFrameId_ = 0, nextKFId_ = 0, nextMPId_ = 0;        // the static id factories (frame.cpp:10, keyframe.cpp:14, mappoint.cpp:8)
    // Map
    std::map<unsigned long, std::shared_ptr<KeyFrame>> allKFs_;
    std::map<unsigned long, KeyFrame*> activeKFs_;
    std::map<unsigned long, std::shared_ptr<MapPoint>> allMPs_, activeMPs_;
    std::vector<unsigned long> outlierMPs_;          // _mlistOutlierMapPoints
    KeyFrame* mapCurKF_ = nullptr;
    // LoopClosing
    std::map<unsigned long, KeyFrame*> dbKFs_;       // _mvDatabase (the descriptors live in the library's device matrix)
    KeyFrame* lastClosedKF_ = nullptr;
    std::vector<std::pair<unsigned long, unsigned long>> loops_;
    bool needCorrect_ = false;
    Pose7 corrected_; std::vector<std::pair<int, int>> validPairs_;
};

}  // namespace myslam
