/*
 * oracle.h — C interface of the CPU ORACLE (test infrastructure, NOT product code).
 *
 * The oracle is a dependency-free, single-threaded CPU restatement of the reference's
 * per-frame dense path.  It exists to CHECK the HIP path:
 *   - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 *   - nothing under the product package includes, links or calls anything in oracle/.
 *
 * PARITY STATUS: **parity unpinned** for every stage whose arithmetic lives in
 * un-vendored third-party code (cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2,
 * BFMatcher, Caffe forward, g2o, Eigen bdcSvd) — the reference has no tests, no golden
 * vectors and cannot be built here (no OpenCV/Eigen/g2o/Caffe).  Those
 * stages follow the published algorithms as restated in SURVEY.md Appendix A.  Stages
 * whose arithmetic is in-tree in the reference follow the cited lines verbatim and are
 * pinned by hand-derivable known-answer tests (tests/test_oracle_kat.py).
 *
 * All functions return 0 on success, negative on error.
 */
#ifndef MYSLAM_ORACLE_H
#define MYSLAM_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layout-compatible with cv::KeyPoint (28 bytes) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

typedef struct {
    int nfeatures;
    float scale_factor;
    int nlevels;
    int ini_th_fast;
    int min_th_fast;
} orc_orb_params;

/* ---- ORB tables: ORBextractor::ORBextractor, src/ORBextractor.cpp:384-445 ---- */
int orc_orb_tables(const orc_orb_params* p, float* scale /*nlevels*/, float* inv_scale,
                   int* n_per_level, int* umax16);
/* pyramid level size: src/ORBextractor.cpp:1237-1238 */
int orc_level_size(int cols, int rows, float inv_scale, int* w, int* h);
/* rBRIEF pattern bytes (1024 int8) */
const int8_t* orc_orb_pattern(void);

/* ---- image primitives (OpenCV 3.4.8 restated) ---- */
int orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstep,
                         uint8_t* dst, int dw, int dh, int dstep);
/* kind 0: 7x7 sigma=2 (ORB, ORBextractor.cpp:966); kind 1: 7x7 sigma=0 -> fixed small table
 * (DeepLCD, deeplcd.cpp:46).  BORDER_REFLECT_101.  src may equal dst. */
int orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstep,
                          uint8_t* dst, int dstep, int kind);
int orc_build_pyramid(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step,
                      uint8_t** levels /*nlevels, caller-allocated w*h tight*/);

/* ---- FAST (cv::FAST restated, Appendix A.1; predicate mirrored at ORBextractor.cpp:449-511) ---- */
/* score map of a whole ROI at threshold th: out[y*w+x] = score (>=th) or 0; no NMS. */
/* replace the sigma = 2 Q8 taps (7 ints, 0..255, sum 1..257; NULL restores the default [18,34,49,55,49,34,18] = OpenCV 3.4.8's per-tap rounding) — process global */
int orc_set_gauss_taps(const int* q7);
/* FAST score of one pixel: the oracle's definition, and OpenCV's cornerScore<16> including its threshold seed */
int orc_fast_score_px(const uint8_t* img, int step, int x, int y);
int orc_fast_score_seeded(const uint8_t* img, int step, int x, int y, int threshold);
int orc_fast_score_map(const uint8_t* img, int w, int h, int step, int th, uint8_t* out);
/* cv::FAST(img, th, nonmax=true): row-major list of (x,y,score) */
int orc_fast_detect(const uint8_t* img, int w, int h, int step, int th,
                    int* xs, int* ys, int* scores, int cap, int* n);
int orc_is_fast_corner(const uint8_t* img, int step, int x, int y, int th);
/* grid FAST of one level (ORBextractor.cpp:814-883): border-relative candidates in reference order */
int orc_grid_fast(const uint8_t* img, int w, int h, int step,
                  const uint8_t* mask, int mstep, int ini_th, int min_th,
                  int* xs, int* ys, int* scores, int cap, int* n);

/* ---- oct-tree (ORBextractor.cpp:526-810); out_idx = indices into the candidate list, list order.
 *      Tie-break of the size sort (:731) = creation order (deterministic restatement). ---- */
int orc_distribute_octree(const int* xs, const int* ys, const int* scores, int n,
                          int minX, int maxX, int minY, int maxY, int N,
                          int* out_idx, int cap, int* nout);

/* ---- orientation / descriptor (ORBextractor.cpp:27-98) ---- */
float orc_fast_atan2(float y, float x);
float orc_ic_angle(const uint8_t* img, int step, int x, int y);
/* deterministic sin/cos of a float angle in radians (see orb_oracle.cpp) */
void orc_sincos(float rad, float* s, float* c);
int orc_brief(const uint8_t* blurred, int step, int x, int y, float angle_deg, uint8_t* desc32);

/* ---- full operators (ORBextractor.cpp:922-985, 989-1074, 1083-1129, 1180-1226) ---- */
int orc_detect_and_compute(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step,
                           const uint8_t* mask, int mstep,
                           orc_keypoint* kps, uint8_t* desc, int cap, int* n);
int orc_detect(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step,
               const uint8_t* mask, int mstep, orc_keypoint* kps, int cap, int* n);
int orc_screen(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step,
               orc_keypoint* kps_in, int n_in, orc_keypoint* kps_out, int cap, int* n_out);
int orc_calc_descriptors(const orc_orb_params* p, const uint8_t* img, int rows, int cols, int step,
                         const orc_keypoint* kps, int n, uint8_t* desc);

/* ---- Hamming brute force (BFMatcher NORM_HAMMING, loopclosing.cpp:33,172; App. A.5) ---- */
int orc_hamming_match(const uint8_t* q, int nq, const uint8_t* t, int nt,
                      int32_t* train_idx, int32_t* dist);
/* filter of loopclosing.cpp:175-194: keep[i] = dist[i] <= max(2*min_dist, 30) */
int orc_hamming_filter(const int32_t* dist, int n, uint8_t* keep, int* min_dist);

/* ---- triangulation (algorithm.h:16-33); poses = nviews x 12 (row-major 3x4 [R|t]), pts = nviews x 3 ---- */
int orc_triangulate(const double* poses, const double* pts, int nviews, double* xyz, double* sv_ratio);
/* stereo batch: left ext = I, right ext t = (-baseline,0,0) (system.cpp:108-116,141-145);
 * pixel2camera camera.cpp:22-26; ok = ratio<1e-2 && z>0 (frontend.cpp:400-401, 471-472) */
int orc_triangulate_stereo(const float* xl, const float* yl, const float* xr, const float* yr, int n,
                           double fx, double fy, double cx, double cy, double baseline,
                           double* xyz, uint8_t* ok);

/* ---- CALC / DeepLCD (deeplcd.cpp:43-91; architecture Appendix A.6) ---- */
/* weights blob layout: see calc_oracle.cpp header. */
int orc_calc_preproc(uint8_t* img, int rows, int cols, int step, int blur_in_place, float* out /*120*160*/);
int orc_calc_forward(const float* weights, size_t nweights, const float* in /*120*160*/, float* out1064);
/* one layer as deploy.prototxt describes it: type 1 Convolution / 2 ReLU / 3 Pooling MAX / 4 LRN (across channels) */
typedef struct orc_calc_layer { int32_t type, num_output, kernel, stride, pad, local_size; float alpha, beta, k; } orc_calc_layer;
int orc_calc_forward_net(const orc_calc_layer* layers, int nlayers, const float* weights, size_t nweights, const float* in, float* out1064);
size_t orc_calc_nweights(void);
float orc_lcd_score(const float* a, const float* b);
/* DetectLoop scan (loopclosing.cpp:124-161): ids ascending */
int orc_lcddb_query(const float* db, const uint64_t* ids, int n, const float* q, uint64_t cur_id,
                    float thr_low, uint64_t* best_id, float* max_score, int* cnt);

/* ---- BA (g2o_types.h:115-144, backend.cpp:126-232, Appendix A.7) ---- */
/* poses: nposes x 7 (qx,qy,qz,qw,tx,ty,tz) Tcw; points: npts x 3; edges: pose_idx, pt_idx, obs(u,v).
 * fixed_pt[npts] flags.  Outputs: Hpp nposes*36, Hll npts*9, Hpl nedges*18 (6x3 row-major),
 * bp nposes*6, bl npts*3, chi2 nedges (robustified: e2 raw in chi2_raw). */
int orc_ba_build(const double* poses, int nposes, const double* points, int npts,
                 const int32_t* edge_pose, const int32_t* edge_pt, const double* obs, int nedges,
                 const uint8_t* fixed_pt, double fx, double fy, double cx, double cy, double huber_delta,
                 double* Hpp, double* Hll, double* Hpl, double* bp, double* bl, double* chi2_raw);
/* Levenberg-Marquardt as A.7; poses/points updated in place; returns iterations done in *iters */
int orc_ba_optimize(double* poses, int nposes, double* points, int npts,
                    const int32_t* edge_pose, const int32_t* edge_pt, const double* obs, int nedges,
                    const uint8_t* fixed_pt, double fx, double fy, double cx, double cy, double huber_delta,
                    int max_iters, double* final_chi2, int* iters);
/* Backend::OptimizeActiveMap outer loop (src/backend.cpp:208-243); *rounds = rounds that failed the inlier-ratio test */
int orc_ba_optimize_active_map(double* poses, int nposes, double* points, int npts,
                               const int32_t* edge_pose, const int32_t* edge_pt, const double* obs, int nedges,
                               const uint8_t* fixed_pt, double fx, double fy, double cx, double cy, double huber_delta,
                               double chi2_th, int max_rounds, int iters_per_round,
                               double* edge_chi2, uint8_t* outlier, int* rounds, int* n_outliers);
void orc_se3_exp(const double* xi6, double* q_t7);
/* Frontend::EstimateCurrentPose (src/frontend.cpp:176-276): pose7 = (qx qy qz qw tx ty tz) Tcw in/out */
int orc_pose_only_optimize(double* pose7, const double* pts3d, const double* obs, int n, double fx, double fy, double cx, double cy,
                           double chi2_th, int rounds, int iters, int pre_optimize, uint8_t* outlier, int* n_inliers);

/* ---- pyramidal LK tracker (cv::calcOpticalFlowPyrLK as called at frontend.cpp:150-153, 358-361; lk_oracle.cpp) ---- */
int orc_pyr_down(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep);
int orc_lk_track(const uint8_t* prev, const uint8_t* next, int rows, int cols, int pstep, int nstep,
                 const float* prev_pts, float* next_pts, int n, int win, int max_level, int max_iters, float eps,
                 float min_eig_threshold, uint8_t* status, float* err);

/* ---- loop correction (pgo_oracle.cpp): LoopClosing::PoseGraphOptimization, src/loopclosing.cpp:537-646 ---- */
/* poses n x 7 (qx qy qz qw tx ty tz) Tcw in/out; edge k: error = log(meas_k^-1 * T[e0] * T[e1]^-1) (g2o_types.h:157-167) */
int orc_loop_local_fusion(double* active_poses, int n_active, int cur, const double* corrected_cur, const int32_t* first_active_kf,
                          double* points, int n_points);
int orc_pose_graph_optimize(double* poses, int n, const uint8_t* fixed, const int32_t* e0, const int32_t* e1,
                            const double* meas, int E, int max_iters, double* final_chi2, int* iters);
/* :621-633: p <- T_new[kf]^-1 * (T_old[kf] * p); kf < 0 leaves the point alone */
int orc_correct_map_points(const double* old_poses, const double* new_poses, int nposes, const int32_t* kf, double* pts, int npts);
int orc_se3_log(const double* q_t7, double* xi6);
int orc_se3_compose(const double* a7, const double* b7, int invert_b, double* out7);

/* ---- loop verification (pnp_oracle.cpp): cv::solvePnPRansac as called at src/loopclosing.cpp:262-268 ---- */
/* pts3d n x 3, pts2d n x 2 (float, as the reference passes them); pose7 = (qx qy qz qw tx ty tz) world -> camera; inlier n flags.
 * returns 0, -2 (fewer than 5 points), -3 (no model found) */
int orc_solve_pnp_ransac(const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx, double cy, int iterations,
                         double reproj_error, double confidence, double* pose7, uint8_t* inlier, int* n_inliers);
/* EPnP on n >= 4 correspondences (pw n x 3, uv n x 2): R (row-major 3x3), t */
int orc_epnp(const double* pw, const double* uv, int n, double fx, double fy, double cx, double cy, double* R9, double* t3);
/* `count` draws of cv::RNG(seed).uniform(a, b) */
int orc_cv_rng_uniform(uint64_t seed, int a, int b, int count, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif
