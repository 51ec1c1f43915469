/*
 * myslam_hip.h — C ABI of libmyslam_hip.so: the MI355X (gfx950) implementation of the per-frame
 * dense path of "A Simple Stereo SLAM System with Deep Loop Closing".
 *
 * The reference has no FFI/plugin layer: the path sits behind C++ member functions of libmyslam.so
 * (SURVEY.md §8b).  Each entry point below names the reference interface it replaces
 * (file:line under the reference tree).  A maintainer's binding is shown in INTEGRATION.md; the
 * C++ facade with the reference's own class names lives in <pkg>/host/.
 *
 * Conventions: caller-owned buffers; plain pointers and sizes; int status (0 = ok, <0 = error);
 * no exceptions cross the ABI.  "_batch" entry points take DEVICE pointers, are asynchronous on
 * the handle's HIP stream and never synchronise; the others take HOST pointers and return when the
 * result is in the caller's buffers.  There is NO CPU fallback: every call fails with
 * MYSLAM_ERR_HIP when no gfx950 device is usable.
 *
 * Threading: a handle owns device scratch and one HIP stream, so it serves one thread at a time; handle-free functions are
 * re-entrant.  The reference shares ONE ORBextractor between the frontend and the loop-closing thread (src/system.cpp:31,54,66):
 * create one handle per thread instead.  Work of different handles on different streams runs concurrently on the device.
 */
#ifndef MYSLAM_HIP_H
#define MYSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MYSLAM_OK 0
#define MYSLAM_ERR_INVALID (-1)     /* bad argument */
#define MYSLAM_ERR_HIP (-2)         /* HIP runtime error / no device */
#define MYSLAM_ERR_CAPACITY (-3)    /* an output or internal buffer was too small */
#define MYSLAM_ERR_UNSUPPORTED (-4) /* image too small for the 30-px FAST grid etc. */

/* cv::KeyPoint layout (28 bytes): pt.x pt.y size angle response octave class_id */
typedef struct myslam_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} myslam_keypoint;

#define MYSLAM_DESC_BYTES 32   /* 256-bit rBRIEF */
#define MYSLAM_LCD_DIM 1064    /* DeepLCD::DescrVector, include/myslam/deeplcd.h:25 */

/* ------------------------------------------------------------------------------------------
 * Library-wide
 * ------------------------------------------------------------------------------------------ */
int myslam_hip_device_count(void);
/* "myslam_hip <version> (gfx950) build <digest>": the digest covers every source file the library was built from (build.py), so a measurement
 * file can name the build it was taken on (profiles/r<NN>_pmc_*.json "build_id"; bench.py sets roofline.traffic_stale when it differs) */
const char* myslam_hip_version(void);
/* The shader clock the device runs at NOW: one wave spins for spin_us (0 < spin_us <= 1e6) on the given stream and compares the shader cycle counter
 * with the constant 100 MHz counter; synchronises the stream.  bench.py samples it around its timed regions ("clock_mhz"): box-to-box and
 * run-to-run differences of a few per cent are clock states more often than code */
int myslam_prof_shader_clock_mhz(void* hip_stream, float spin_us, float* mhz);
/* per-kernel HIP-event timing: enable, run, synchronise, then read (name, total ms, launches) */
int myslam_prof_enable(int on);
int myslam_prof_reset(void);
int myslam_prof_count(void);
int myslam_prof_get(int i, const char** name, double* total_ms, long* launches);
/* debug: device->host copy of n bytes through this library's HIP runtime */
int myslam_debug_peek(const void* d_ptr, void* out, size_t n);

/* ------------------------------------------------------------------------------------------
 * ORB extractor — replaces class ORBextractor (include/myslam/ORBextractor.h:52-110)
 * ------------------------------------------------------------------------------------------ */
typedef struct myslam_orb myslam_orb;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)  ORBextractor.h:55-56 */
int myslam_orb_create(myslam_orb** out, int nfeatures, float scale_factor, int nlevels,
                      int ini_th_fast, int min_th_fast);
int myslam_orb_destroy(myslam_orb* h);
/* all work of this handle is ordered on `hip_stream` (a hipStream_t; NULL = default stream).  The batched extractor may run its
 * Gaussian-pyramid launches on an internal stream; they are fenced by events against `hip_stream` on both sides, so for the caller
 * every call still starts after, and completes before, its neighbours on `hip_stream`. */
int myslam_orb_set_stream(myslam_orb* h, void* hip_stream);
/* pipelining aid for callers that run several extractor handles on several streams (two cameras, or the left and the right images
 * of a batch of stereo pairs): `hip_event` (a hipEvent_t, NULL = off) is
 * recorded on the handle's stream right after the grid-FAST stage of every following batched call.  FAST is the VALU-bound
 * stage, the oct-tree / descriptor stages after it are latency-bound: a second handle whose stream waits for this event runs its
 * own FAST under them instead of beside the first handle's FAST.  No reference counterpart (scheduling only). */
int myslam_orb_set_fast_event(myslam_orb* h, void* hip_event);
/* the matching gate: before the grid-FAST stage of every following batched call the handle's stream waits for `hip_event`
 * (a hipEvent_t, NULL = off) as it was last recorded when the call is made; the pyramid stages before FAST are not held back.
 * Two handles that gate each other with their fast events take turns on the VALU-bound stage. */
int myslam_orb_set_fast_gate(myslam_orb* h, void* hip_event);
/* scheduling / debugging knobs of a handle (no reference counterpart; none of them changes a result):
 *   FAST_MODE        -1 (default) = the grid-FAST kernel picks its path per pyramid level from what the handle's previous launch measured:
 *                    a compass pre-test + compaction of the surviving pixel pairs (imagery with a few % of corners: a tenth of the pixels
 *                    is scored) or full scoring of every pixel (noise-like texture where most pairs survive the pre-test);
 *                    0 / 1 force the two-phase / the dense path
 *   INTERNAL_STREAM  1 (default) = the Gaussian pyramid runs on an internal stream beside the oct-tree kernel (forked after FAST),
 *                    2 = forked right after the image pyramid (beside FAST), 0 = everything on the handle's stream
 *   COPY_INPUT       0 (default) = a *_batch call reads the full-resolution level of every image but the last IN PLACE (no copy into the
 *                    pyramid block): the input buffer must then stay unchanged until the call has completed on the handle's stream;
 *                    1 = every image is copied first, the input may be overwritten as soon as the copy kernel has run
 *   STOP_AFTER       debug: a batched call returns after stage 1 ingest / 2 pyramid / 3 oct-tree / 4 blur (0 = complete call)
 *   BLUR_MFMA        1 = the 7 x 7 Gaussian pyramid runs on the int8 matrix cores (two banded Toeplitz products per 32 x 32 tile, bit-identical
 *                    to the register-strip kernel; levels narrower than 64 columns or tap tables with a folded coefficient above 127 keep
 *                    the strip kernel), 0 = register-strip kernel on the vector units.  The step is VALU-issue bound: see DESIGN.md section 6
 *   SIDE_BLOCKS_PER_CU  n > 0: the descriptor kernel of a batched call is launched as a LIMITED grid of 256 n blocks, each walking several
 *                    (image, 64-key-point chunk) work items.  For callers that run two handles beside each other: a descriptor block lives
 *                    three times as long as a FAST block, so an unlimited grid gradually takes the CUs from the other handle's FAST launch —
 *                    the VALU-bound kernel starves under the latency-bound one (measured: 7.12 -> 7.02 ms per 512-pair step with n = 2, round 4).
 *                    Since the kernel fetches its BRIEF windows by LDS-DMA (end of round 6) its blocks are short and the unlimited grid measures
 *                    faster (6.25 -> 6.09 ms): 0 (default) = one block per work item */
#define MYSLAM_ORB_OPT_FAST_MODE 1
#define MYSLAM_ORB_OPT_INTERNAL_STREAM 2
#define MYSLAM_ORB_OPT_STOP_AFTER 3
#define MYSLAM_ORB_OPT_COPY_INPUT 4
#define MYSLAM_ORB_OPT_BLUR_MFMA 5
#define MYSLAM_ORB_OPT_SIDE_BLOCKS_PER_CU 6
int myslam_orb_set_option(myslam_orb* h, int option, int value);
/* The 7 x 7 sigma = 2 Gaussian before rBRIEF (ORBextractor.cpp:966, :1197) runs in OpenCV's 8-bit fixed-point form; how OpenCV 3.4.8
 * rounds the taps to Q8 could not be checked in the build environment (parity unpinned).  Default
 * [18,34,49,55,49,34,18]: every normalised tap rounded to nearest on its own, as getFixedpointGaussianKernel of the 3.4.8 era does —
 * the sum is 257 and the u8 result saturates as ufixedpoint32 -> uint8_t does.  A maintainer who measures another table with
 * tools/dump_opencv_goldens.py sets it here (7 ints, 0..255, sum 1..257 so that the Q8.8 row sums fit 16 bits; NULL restores the
 * default; [18,34,49,54,49,34,18] is the sum-256 table rounds 1-2 of this library used). */
int myslam_orb_set_gauss_taps(myslam_orb* h, const int32_t* q7);
/* getters ORBextractor.h:87-107 */
int myslam_orb_get_tables(const myslam_orb* h, float* scale, float* inv_scale, int* features_per_level, int* umax16);
/* upper bound of keypoints DetectAndCompute / Detect can return for one image: sum over levels of
 * max(N_level + 3, 32) — the oct-tree stops only after a split pushed it to >= N, and its first round is unconditional */
int myslam_orb_max_keypoints(const myslam_orb* h);
/* the same bound for one image size, exact for ANY aspect ratio (a level with nIni = round(width/height) root nodes can return
 * 4*nIni key-points even when its budget is smaller, ORBextractor.cpp:645-716); use it to size kps / desc for that size */
int myslam_orb_max_keypoints_for(const myslam_orb* h, int rows, int cols);

/* void DetectAndCompute(image, mask, keypoints, descriptors)   ORBextractor.h:61-63, .cpp:922-985
 * mask may be NULL (= all 255).  desc: cap x 32 bytes. */
int myslam_orb_detect_and_compute(myslam_orb* h, const uint8_t* img, int rows, int cols, int step,
                                  const uint8_t* mask, int mask_step,
                                  myslam_keypoint* kps, uint8_t* desc, int cap, int* n);
/* void Detect(image, mask, keypoints)   ORBextractor.h:70-71, .cpp:989-1074 (level 0 only) */
int myslam_orb_detect(myslam_orb* h, const uint8_t* img, int rows, int cols, int step,
                      const uint8_t* mask, int mask_step, myslam_keypoint* kps, int cap, int* n);
/* void ScreenAndComputeKPsParams(image, keypoints, out_keypoints)   ORBextractor.h:83-84, .cpp:1083-1129
 * kps_in is modified in place exactly as the reference does (pt /= scale ... pt *= scale). */
int myslam_orb_screen_and_compute_params(myslam_orb* h, const uint8_t* img, int rows, int cols, int step,
                                         myslam_keypoint* kps_in, int n_in,
                                         myslam_keypoint* kps_out, int cap, int* n_out);
/* void CalcDescriptors(image, keypoints, descriptors)   ORBextractor.h:65-67, .cpp:1180-1226 */
int myslam_orb_calc_descriptors(myslam_orb* h, const uint8_t* img, int rows, int cols, int step,
                                const myslam_keypoint* kps, int n, uint8_t* desc);

/* Batched DetectAndCompute over `batch` same-sized images resident in HBM.
 *   d_imgs : batch images, image b at d_imgs + b*img_stride, row pitch `step` bytes
 *   d_masks: NULL or same layout
 *   d_kps  : batch x cap keypoints, d_desc: batch x cap x 32, d_counts: batch
 *   d_status: batch int32 (0 ok / MYSLAM_ERR_CAPACITY), may be NULL
 * level_only0 != 0 runs Detect() semantics (level 0, N = nfeatures, no angle/descriptor). */
int myslam_orb_detect_and_compute_batch(myslam_orb* h, const uint8_t* d_imgs, int batch, int rows, int cols,
                                        int step, size_t img_stride, const uint8_t* d_masks,
                                        myslam_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts,
                                        int32_t* d_status, int cap);
int myslam_orb_detect_batch(myslam_orb* h, const uint8_t* d_imgs, int batch, int rows, int cols,
                            int step, size_t img_stride, const uint8_t* d_masks,
                            myslam_keypoint* d_kps, int32_t* d_counts, int32_t* d_status, int cap);

/* debug/inspection taps used by the stage-level parity tests (host buffers, one image) */
int myslam_orb_debug_pyramid(myslam_orb* h, const uint8_t* img, int rows, int cols, int step,
                             int level, int blurred, uint8_t* out, int out_step, int* w, int* hgt);
int myslam_orb_debug_candidates(myslam_orb* h, const uint8_t* img, int rows, int cols, int step,
                                const uint8_t* mask, int mask_step, int level,
                                int32_t* xs, int32_t* ys, int32_t* scores, int cap, int* n);

/* raw readback of the engine's HBM buffers after a *_batch call: what = 0 pyramid plane (w*h, tight), 1 blurred
 * plane, 2 candidate count (i32), 3 candidate payloads (u32 py<<20|px<<8|score), 4 selected count, 5 selected payloads,
 * 6 the grid-FAST statistics of the last launch on `level`: 4 x u32 {pixel pairs that survived the pre-test (two-phase path) or 4-pixel
 * rows holding a corner (dense path), pixel pairs looked at — both over the sampled strips —, path used (0 two-phase, 1 dense), 0} */
int myslam_orb_debug_readback(myslam_orb* h, int what, int b, int level, void* out, size_t cap_bytes, int detect_plan);

/* ------------------------------------------------------------------------------------------
 * Hamming brute force — replaces cv::BFMatcher(NORM_HAMMING)::match as used at
 * src/loopclosing.cpp:33,172 and the filter at :175-194.  One match per query row, ties -> lowest
 * train index.  nt == 0 -> train_idx = -1, dist = -1.
 * ------------------------------------------------------------------------------------------ */
int myslam_hamming_match(const uint8_t* query, int nq, const uint8_t* train, int nt,
                         int32_t* train_idx, int32_t* dist);
/* batch: pair p uses d_q + p*cap*32 (d_nq[p] rows) vs d_t + p*cap*32 (d_nt[p] rows) */
int myslam_hamming_match_batch(const uint8_t* d_q, const int32_t* d_nq, const uint8_t* d_t, const int32_t* d_nt,
                               int batch, int cap, int32_t* d_train_idx, int32_t* d_dist, void* hip_stream);
/* keep[i] = dist[i] <= max(2*min_dist, 30.0)   (loopclosing.cpp:175-186); host-side bookkeeping */
int myslam_hamming_filter(const int32_t* dist, int n, uint8_t* keep, int* min_dist);

/* Key-frame / feature bookkeeping of the loop closer around the two calls above (KeyFrame::mvPyramidKeyPoints,
 * cv::KeyPoint::class_id as the feature index).  Host functions, no device work.
 * expand: LoopClosing::ProcessNewKF src/loopclosing.cpp:94-105 — out[i * nlevels + l] = feats[i] with octave = l, response = -1,
 *         class_id = i (the input of myslam_orb_screen_and_compute_params).
 * pairs : LoopClosing::MatchFeatures :175-194 — matches with distance <= max(2 * min_dist, 30) mapped to (current feature id, loop
 *         feature id) through class_id, de-duplicated and ordered as the reference's std::set<std::pair<int,int>> iterates. */
int myslam_expand_pyramid_keypoints(const myslam_keypoint* feats, int n, int nlevels, myslam_keypoint* out);
int myslam_match_feature_pairs(const int32_t* train_idx, const int32_t* dist, int n_query, const myslam_keypoint* loop_pyr_kps,
                               const myslam_keypoint* cur_pyr_kps, int n_train, int32_t* pairs, int* n_pairs);

/* ------------------------------------------------------------------------------------------
 * Triangulation — replaces triangulation() include/myslam/algorithm.h:16-33 with the stereo rig
 * of src/system.cpp:108-116,141-145 and Camera::pixel2camera src/camera.cpp:22-26.
 * ok[i] = (sigma3/sigma2 < 1e-2) && z > 0   (frontend.cpp:400, 471).
 * ------------------------------------------------------------------------------------------ */
int myslam_triangulate_stereo(const float* xl, const float* yl, const float* xr, const float* yr, int n,
                              double fx, double fy, double cx, double cy, double baseline,
                              double* xyz, uint8_t* ok);
/* batch: left keypoint i of pair p is matched to right keypoint d_match[p*cap+i] (<0 = none) */
int myslam_triangulate_stereo_batch(const myslam_keypoint* d_kps_l, const myslam_keypoint* d_kps_r,
                                    const int32_t* d_match, const int32_t* d_nl, int batch, int cap,
                                    double fx, double fy, double cx, double cy, double baseline,
                                    double* d_xyz, uint8_t* d_ok, void* hip_stream);

/* cv::BFMatcher::match (src/loopclosing.cpp:172-173) followed by triangulation() (include/myslam/algorithm.h:16-33) of every matched left
 * key-point, as Frontend::FindFeaturesInRight + TriangulateNewPoints do in sequence (src/frontend.cpp:344-420): exactly
 * myslam_hamming_match_batch(d_q = left descriptors, d_t = right descriptors, ...) and then myslam_triangulate_stereo_batch(d_kps_l, d_kps_r,
 * d_train_idx, d_nq, ...) — same outputs, bit for bit.  For fewer than 16 pairs per call it is ONE launch (each matcher block triangulates
 * its 128 queries once their matches are known): a live stream's step is bound by the number of its dependent launches. */
int myslam_hamming_match_triangulate_batch(const uint8_t* d_q, const int32_t* d_nq, const uint8_t* d_t, const int32_t* d_nt,
                                           const myslam_keypoint* d_kps_l, const myslam_keypoint* d_kps_r, int batch, int cap,
                                           double fx, double fy, double cx, double cy, double baseline,
                                           int32_t* d_train_idx, int32_t* d_dist, double* d_xyz, uint8_t* d_ok, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * DeepLCD — replaces class DeepLCD (include/myslam/deeplcd.h:21-48, src/deeplcd.cpp:10-91)
 *
 * The model is DATA: a list of layer records (what calc_model/deploy.prototxt says) + the convolution weights (what
 * calc_model/calc.caffemodel holds).  Three sources:
 *   myslam_lcd_create_from_caffe  the reference's two files, read by a dependency-free parser (host/myslam_caffe.hpp)
 *   myslam_lcd_create_from_layers records + flat weights from the caller
 *   myslam_lcd_create / _from_file the SURVEY A.6 layer list (myslam_lcd_default_layers) + a flat blob / the CALCW1 / CALCW2 files
 * weights: flat f32, per Convolution layer in order: w[OC][IC][K][K] then b[OC] (Caffe's blob order).  For the default list:
 *          conv1.w[64][1][5][5] conv1.b[64] conv2.w[128][64][4][4] conv2.b[128] conv3.w[4][128][3][3] conv3.b[4] (137476 floats).
 * Supported layers: Convolution (square kernel, group 1, with bias), ReLU, Pooling MAX (Caffe ceil mode, pad 0), LRN ACROSS_CHANNELS;
 * input 1 x 120 x 160 (the reference resizes every frame to that, deeplcd.cpp:50); 1064 outputs (deeplcd.cpp:80 asserts it).
 * Anything else: MYSLAM_ERR_UNSUPPORTED.  A list with the SURVEY A.6 geometry (any LRN alpha / beta / k, ReLUs present or not) runs
 * on fused kernels; other lists run layer by layer on generic kernels (myslam_lcd_uses_fused_kernels tells which).
 * ------------------------------------------------------------------------------------------ */
#define MYSLAM_CALC_CONV 1
#define MYSLAM_CALC_RELU 2
#define MYSLAM_CALC_POOL_MAX 3
#define MYSLAM_CALC_LRN 4
typedef struct myslam_calc_layer {
    int32_t type;                              /* MYSLAM_CALC_* */
    int32_t num_output, kernel, stride, pad;   /* Convolution (convolution_param); Pooling uses kernel / stride / pad (pooling_param) */
    int32_t local_size;                        /* LRN (lrn_param): y = x * (k + alpha / local_size * sum x^2)^-beta */
    float alpha, beta, k;
} myslam_calc_layer;
typedef struct myslam_lcd myslam_lcd;
/* the SURVEY A.6 list (10 records); layers == NULL returns the count only */
int myslam_lcd_default_layers(myslam_calc_layer* layers, int cap);
int myslam_lcd_create(myslam_lcd** out, const float* weights, size_t nweights);
int myslam_lcd_create_from_layers(myslam_lcd** out, const myslam_calc_layer* layers, int nlayers, const float* weights, size_t nweights);
/* DeepLCD::DeepLCD(prototxt_path, caffemodel_path, gpu_id)  deeplcd.h:33, deeplcd.cpp:10-31 */
int myslam_lcd_create_from_caffe(myslam_lcd** out, const char* prototxt_path, const char* caffemodel_path);
/* host only (no device needed): parse + validate the two Caffe files into records and the flat weight blob; NULL outputs are skipped */
int myslam_calc_parse_caffe(const char* prototxt_path, const char* caffemodel_path, myslam_calc_layer* layers, int cap, int* nlayers,
                            float* weights, size_t wcap, size_t* nweights);
/* own files: "CALCW1" (weights of the default list) or "CALCW2" (records + weights), see csrc/calc.hip */
int myslam_lcd_create_from_file(myslam_lcd** out, const char* path);
/* 1 = the layer list runs on the fused kernels, 0 = on the generic layer kernels */
int myslam_lcd_uses_fused_kernels(const myslam_lcd* h);
/* partial products per multiply-add of the fused path's conv2: 3 = f16 x 3 (k_conv2_f16x3: the model's ranges fit f16 — |conv2 weight| < 31 and
 * sum |conv1 weights| + |bias| < 60000 with an LRN that cannot amplify), 6 = bf16 x 6 (k_conv2_bf16x6), 0 = generic kernels.  Both matrix-core forms
 * reach f32-level accuracy (max-normalised error against f64: 1.1e-6 / 1.5e-6). */
int myslam_lcd_conv2_products(const myslam_lcd* h);
#define MYSLAM_LCD_OPT_GENERIC_KERNELS 1       /* value != 0: run even a fusable list on the generic kernels (tests, diagnosis) */
#define MYSLAM_LCD_OPT_CONV2_BF16X6 2          /* value != 0: conv2 of the fused path on the six-product bf16 kernel and conv1 on its f32 vector kernel even when
                                                * the model's ranges allow the three-product f16 matrix-core kernels (all reach f32-level accuracy; tests and
                                                * tools/gpu_fuzz_lcd.py compare the two families) */
#define MYSLAM_LCD_OPT_SKIP_KERNELS 3          /* DIAGNOSIS, timing only (results are garbage): bit mask of fused-path kernels that are not launched —
                                                * 1 input, 2 conv1, 4 conv2, 8 pool2, 16 conv3 + norm: what each costs a step that runs beside the extractor */
int myslam_lcd_set_option(myslam_lcd* h, int option, int value);
int myslam_lcd_destroy(myslam_lcd* h);
int myslam_lcd_set_stream(myslam_lcd* h, void* hip_stream);
size_t myslam_lcd_nweights(void);
/* DescrVector calcDescrOriginalImg(const cv::Mat&)  deeplcd.cpp:43-52.  blur_in_place != 0 reproduces
 * the reference's side effect (the caller's image is Gaussian-blurred). */
int myslam_lcd_calc_descr_original_img(myslam_lcd* h, uint8_t* img, int rows, int cols, int step,
                                       int blur_in_place, float* descr1064);
/* const DescrVector calcDescr(const cv::Mat& im)  deeplcd.cpp:55-91; im = 160x120 u8 */
int myslam_lcd_calc_descr(myslam_lcd* h, const uint8_t* img160x120, int step, float* descr1064);
/* const float score(d1, d2)  deeplcd.cpp:35-39 */
float myslam_lcd_score(const float* d1, const float* d2);
int myslam_lcd_describe_batch(myslam_lcd* h, uint8_t* d_imgs, int batch, int rows, int cols, int step,
                              size_t img_stride, int blur_in_place, float* d_descr /* batch x 1064 */);
/* stage taps for parity tests (host buffers, one image) */
int myslam_lcd_debug_forward(myslam_lcd* h, const float* in120x160, float* out_stage, int stage, size_t cap_floats);

/* ------------------------------------------------------------------------------------------
 * Loop database — replaces LoopClosing::_mvDatabase + DetectLoop()/AddToDatabase()
 * (include/myslam/loopclosing.h:67,120; src/loopclosing.cpp:124-161, 651-659).
 * ids must be appended in ascending order (std::map iteration order).
 * query: scan ascending, stop at the first id with cur_id - id < 20, max score (strict >, init 0),
 *        cnt = #{score > thr_low}.  The caller applies maxScore >= thr_high && cnt <= 3.
 * ------------------------------------------------------------------------------------------ */
typedef struct myslam_lcddb myslam_lcddb;
int myslam_lcddb_create(myslam_lcddb** out, int capacity);
int myslam_lcddb_destroy(myslam_lcddb* h);
int myslam_lcddb_set_stream(myslam_lcddb* h, void* hip_stream);
int myslam_lcddb_size(const myslam_lcddb* h);
/* `capacity` of myslam_lcddb_create is the first allocation only: _mvDatabase is an unbounded std::map (loopclosing.h:120), so append
 * grows the device matrix geometrically (x2, one device-to-device copy of the rows held so far; the handle's stream is synchronised
 * while it moves) and fails with MYSLAM_ERR_CAPACITY only when the device has no memory left.  myslam_lcddb_reserve makes room for
 * `rows` key-frames ahead of time (never shrinks); myslam_lcddb_capacity = rows the current allocation holds. */
int myslam_lcddb_capacity(const myslam_lcddb* h);
int myslam_lcddb_reserve(myslam_lcddb* h, int rows);
int myslam_lcddb_append(myslam_lcddb* h, uint64_t id, const float* descr1064);
int myslam_lcddb_append_batch(myslam_lcddb* h, const uint64_t* ids /*host*/, const float* d_descr, int n);
/* AddToDatabase inside a pipelined step (round 6): as myslam_lcddb_append_batch, but the device copies are enqueued on `hip_stream` and the call does NOT wait for them — the
 * caller orders later scans of those rows behind the copy (same stream, or an event).  Ids and row count are updated before the call returns (the next query's row limits
 * cover the new rows).  Never moves the matrix: MYSLAM_ERR_CAPACITY when the rows do not fit the allocation (myslam_lcddb_reserve ahead of the run). */
int myslam_lcddb_append_batch_async(myslam_lcddb* h, const uint64_t* ids /*host*/, const float* d_descr, int n, void* hip_stream);
int myslam_lcddb_query(myslam_lcddb* h, const float* descr1064, uint64_t cur_id, float thr_low,
                       uint64_t* best_id, float* max_score, int* cnt);
/* nq queries at once: d_q nq x 1064 (device), cur_ids nq (HOST); outputs nq each (device).  nq <= 65536; no host synchronisation. */
int myslam_lcddb_query_batch(myslam_lcddb* h, const float* d_q, const uint64_t* cur_ids /*host*/, int nq, float thr_low,
                             uint64_t* d_best_id, float* d_max_score, int32_t* d_cnt);

/* For a query recorded into a HIP graph (myslam_graph_begin / _end below): the per-query row limits (the `cur - id < 20` cut-off of
 * loopclosing.cpp:133 for each cur_id) sit in a pinned host buffer that the recorded copy reads at EVERY replay.  This call rewrites them
 * for new cur_ids or after appends; it waits for the handle's stream first.  MYSLAM_ERR_CAPACITY = the database moved (it grew beyond its
 * allocation) or holds more rows than the recorded launch covers: record the step again.  nq <= the recorded query count. */
int myslam_lcddb_update_query_limits(myslam_lcddb* h, const uint64_t* cur_ids /*host*/, int nq);

/* Query contexts: ONE database, several concurrent streams of queries.  LoopClosing::_mvDatabase is a single std::map that every
 * key-frame of the process goes into (include/myslam/loopclosing.h:120, src/loopclosing.cpp:651-659); with L cameras on one GPU the
 * L loop-closing streams must scan the SAME matrix, not L copies of it.  A myslam_lcddb owns the descriptor matrix and the ids
 * (append / reserve); a myslam_lcddb_query_ctx owns what one stream of scans needs (row-limit staging, partial results, shard
 * scratch, recorded-step state) and is bound to one HIP stream.  The handle's own entry points above use a built-in context on the
 * handle's stream.  Rules:
 *   - appends return when the new rows are in HBM; scans in flight on any context only read rows below the limits they were
 *     issued with, so appends and scans need no ordering between them;
 *   - growth beyond the allocation moves the matrix: append / reserve wait for every context's stream and for every replay of a
 *     recorded step that scans through a context, then bump myslam_lcddb_generation();
 *   - a scan recorded into a step graph covers the whole ALLOCATION (blocks behind the row limits return at once): appends inside the
 *     capacity only need myslam_lcddb_ctx_update_query_limits before the next replay.  After a move, myslam_graph_launch of a step
 *     that captured a scan returns MYSLAM_ERR_CAPACITY (record it again); the old matrix stays allocated until the database is
 *     destroyed, so even an unchecked replay reads valid memory;
 *   - a recorded scan also names the CONTEXT's scratch (row limits, partial results): an eager call on the same context that needs more of it
 *     (more queries than any call before) replaces that scratch after waiting for the step's replays, and the step is refused from then on
 *     (MYSLAM_ERR_CAPACITY: record it again) — round 6;
 *   - while a step that scans through a context is being RECORDED (myslam_graph_begin .. _end, on any thread), nothing may synchronise that
 *     context's stream: an append / reserve that would have to move the matrix, and myslam_lcddb_set_stream, return MYSLAM_ERR_UNSUPPORTED until
 *     the recording has ended (appends inside the allocation are unaffected) — round 6;
 *   - a query holds the database's host lock until its launches are enqueued, so a growing append from another thread waits for them (its stream
 *     synchronisation then covers the scan) instead of freeing the matrix under a launch that is about to be issued — round 6;
 *   - contexts are destroyed before their database (myslam_lcddb_destroy frees any that are left: their pointers die with it).
 * One thread per context; append / reserve may come from any thread. */
typedef struct myslam_lcddb_query_ctx myslam_lcddb_query_ctx;
int myslam_lcddb_query_ctx_create(myslam_lcddb_query_ctx** out, myslam_lcddb* db, void* hip_stream);
int myslam_lcddb_query_ctx_destroy(myslam_lcddb_query_ctx* c);
int myslam_lcddb_ctx_query_batch(myslam_lcddb_query_ctx* c, const float* d_q, const uint64_t* cur_ids /*host*/, int nq, float thr_low,
                                 uint64_t* d_best_id, float* d_max_score, int32_t* d_cnt);
int myslam_lcddb_ctx_update_query_limits(myslam_lcddb_query_ctx* c, const uint64_t* cur_ids /*host*/, int nq);
/* number of times the descriptor matrix has moved (growth): a recorded step is valid for the generation it was recorded in */
int myslam_lcddb_generation(const myslam_lcddb* h);

/* Multi-GPU form: the database is sharded by contiguous key-frame id range, shard r on rank r, every shard scores
 * every query.  A shard's answer travels as one 16-byte record; the records of all shards (rank order = id order) are reduced to what
 * ONE scan of src/loopclosing.cpp:124-161 over the whole std::map returns: strict '>' keeps the first (lowest-id) maximum, counts
 * add up, and the first shard whose own scan hit the `cur - id < 20` break (:133) ends the scan for all shards behind it.
 * The all-gather between the two calls is the caller's (RCCL ncclAllGather on the raw bytes). */
typedef struct myslam_lcd_candidate {
    uint64_t best_id;      /* 0 when nothing scored above 0 (loopclosing.cpp:129) */
    float max_score;
    int32_t cnt;           /* bits 0..30: #{score > thr_low} in this shard; bit 31: this shard's scan stopped at the break */
} myslam_lcd_candidate;
/* as myslam_lcddb_query_batch, one record per query (device memory, asynchronous on the handle's stream) */
int myslam_lcddb_query_batch_sharded(myslam_lcddb* h, const float* d_q, const uint64_t* cur_ids /*host*/, int nq, float thr_low,
                                     myslam_lcd_candidate* d_cand);
int myslam_lcddb_ctx_query_batch_sharded(myslam_lcddb_query_ctx* c, const float* d_q, const uint64_t* cur_ids /*host*/, int nq, float thr_low,
                                     myslam_lcd_candidate* d_cand);
/* gathered: [nshards][nq] records, shard-major in ascending id-range order.  Host pointers (plain C++, no device needed). */
int myslam_lcd_merge_candidates(const myslam_lcd_candidate* gathered, int nshards, int nq, uint64_t* best_id, float* max_score, int32_t* cnt);
/* the same reduce on device pointers, asynchronous on hip_stream */
int myslam_lcd_merge_candidates_device(const myslam_lcd_candidate* d_gathered, int nshards, int nq, uint64_t* d_best_id,
                                       float* d_max_score, int32_t* d_cnt, void* hip_stream);

/* A sharded database that GROWS (round 6).  With contiguous id ranges every new key-frame belongs to the last rank; a job that appends (the reference
 * does, per key-frame: LoopClosing::AddToDatabase, src/loopclosing.cpp:651-659) spreads the rows instead — e.g. the k-th key-frame of the job goes to rank
 * k mod N (sharded_db.py GrowingShardedDatabase; app/sharded_db_rccl.cpp) — and a shard's ids then INTERLEAVE with the others'.  Any rule works that puts
 * every id into exactly one shard and keeps each shard's own ids ascending.  The reference's one ascending scan (:124-161) looks at every id below the
 * cut-off window {id : cur - id < 20}, stops if the map holds an id inside it, and otherwise goes on with the ids above cur; so a shard reports BOTH parts
 * and whether it holds an id inside the window (32 bytes per query), and the merge — highest score, LOWEST ID among equal scores (shard order is not id
 * order here), counts added — takes the parts above cur only when no shard reported the break.  Bit-identical to one scan of the whole map. */
typedef struct myslam_lcd_owned_candidate {
    uint64_t pre_best_id;  /* over this shard's ids below the window: 0 when nothing scored above 0 */
    float pre_max_score;
    int32_t pre_cnt;       /* bits 0..30: #{score > thr_low}; bit 31: this shard holds an id with cur - id < 20 */
    uint64_t suf_best_id;  /* over this shard's ids above cur */
    float suf_max_score;
    int32_t suf_cnt;
} myslam_lcd_owned_candidate;
/* one record per query (device memory, asynchronous on the stream; not recordable into a step graph) */
int myslam_lcddb_query_batch_owned(myslam_lcddb* h, const float* d_q, const uint64_t* cur_ids /*host*/, int nq, float thr_low,
                                   myslam_lcd_owned_candidate* d_cand);
int myslam_lcddb_ctx_query_batch_owned(myslam_lcddb_query_ctx* c, const float* d_q, const uint64_t* cur_ids /*host*/, int nq, float thr_low,
                                       myslam_lcd_owned_candidate* d_cand);
/* gathered: [nshards][nq] records, shard-major, ANY shard order.  Host pointers / device pointers (asynchronous on hip_stream). */
int myslam_lcd_merge_owned_candidates(const myslam_lcd_owned_candidate* gathered, int nshards, int nq, uint64_t* best_id, float* max_score, int32_t* cnt);
int myslam_lcd_merge_owned_candidates_device(const myslam_lcd_owned_candidate* d_gathered, int nshards, int nq, uint64_t* d_best_id,
                                             float* d_max_score, int32_t* d_cnt, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Local BA linear-system build — replaces the per-edge work g2o does for
 * Backend::OptimizeActiveMap (src/backend.cpp:126-232): EdgeProjection::computeError /
 * linearizeOplus (include/myslam/g2o_types.h:115-144), Huber(delta) and the block quadratic form.
 * poses  nposes x 7  (qx qy qz qw tx ty tz), Tcw;  points npts x 3;  obs nedges x 2
 * Hpp nposes x 36, Hll npts x 9, Hpl nedges x 18 (6x3 row-major), bp nposes x 6, bl npts x 3,
 * chi2 nedges (un-robustified e^T e, what edge->chi2() returns at backend.cpp:219,238)
 * ------------------------------------------------------------------------------------------ */
int myslam_ba_build(const double* poses, int nposes, const double* points, int npts,
                    const int32_t* edge_pose, const int32_t* edge_pt, const double* obs, int nedges,
                    const uint8_t* fixed_pt, double fx, double fy, double cx, double cy, double huber_delta,
                    double* Hpp, double* Hll, double* Hpl, double* bp, double* bl, double* chi2);
/* batch of `nwin` windows with identical capacities (max_poses, max_pts, max_edges); window w's arrays
 * start at base + w*capacity*elemsize; d_sizes = nwin x 3 (nposes, npts, nedges).  A window whose sizes exceed the capacities is
 * skipped: its blocks are zero and its chi2[0] = -1. */
int myslam_ba_build_batch(const double* d_poses, const double* d_points, const int32_t* d_edge_pose,
                          const int32_t* d_edge_pt, const double* d_obs, const uint8_t* d_fixed,
                          const int32_t* d_sizes, int nwin, int max_poses, int max_pts, int max_edges,
                          double fx, double fy, double cx, double cy, double huber_delta,
                          double* d_Hpp, double* d_Hll, double* d_Hpl, double* d_bp, double* d_bl, double* d_chi2,
                          void* hip_stream);

/* Map -> flat arrays: the graph-build rules of Backend::OptimizeActiveMap (src/backend.cpp:139-206) as a host function (plain C++,
 * no device needed), so that a maintainer's OptimizeActiveMap body is { copy the Map's tables; flatten; optimize_active_map; write back }.
 * Inputs (host pointers), one row per object of the reference's containers:
 *   active_kf_ids[n_kf]                 Map::GetActiveKeyFrames() keys (any order; :136,139-150)
 *   mp_ids / mp_outlier / mp_first_observer_kf [n_mp]   Map::GetActiveMapPoints(): mnId, mbIsOutlier, and the key-frame id of
 *                                       GetObservations().front() (:163-177)
 *   obs_mp_id / obs_kf_id / obs_uv / obs_feat_outlier [n_obs]   every MapPoint::GetActiveObservations() entry (list order within a map
 *                                       point): the map point's id, feature->mpKF->mnKFId, feature->mkpPosition.pt, feature->mbIsOutlier (:181-189)
 * Rules reproduced: outlier map points get no vertex and no edges (:163); outlier features get no edge (:189); a map point whose FIRST
 * observer is not an active key-frame is fixed (:175-177); an observation from a key-frame outside the active set is the reference's
 * assert (:187) -> MYSLAM_ERR_INVALID, as is an observation of an unknown map point.  Orders (the reference iterates unordered_maps, g2o
 * then sorts vertices by id): pose slots by ascending key-frame id, landmark slots by ascending map-point id, edges grouped by landmark
 * (what myslam_ba_optimize* need) in observation-list order.  Map points left without an edge are dropped (g2o never activates them).
 * Outputs: pose_src[n_kf] / pt_src[*n_pts] = index of the INPUT row each slot was taken from (to gather poses / positions and to write
 * results back: :252-258), edge_pose / edge_pt / edge_obs (double, as toVec2) / edge_src (index of the observation row, for the outlier
 * write-back :234-250) [*n_edges <= n_obs], fixed_pt[*n_pts]. */
int myslam_ba_flatten_window(const uint64_t* active_kf_ids, int n_kf, const uint64_t* mp_ids, const uint8_t* mp_outlier,
                             const uint64_t* mp_first_observer_kf, int n_mp, const uint64_t* obs_mp_id, const uint64_t* obs_kf_id,
                             const float* obs_uv, const uint8_t* obs_feat_outlier, int n_obs, int32_t* pose_src, int32_t* pt_src,
                             int32_t* n_pts, int32_t* edge_pose, int32_t* edge_pt, double* edge_obs, int32_t* edge_src, int32_t* n_edges,
                             uint8_t* fixed_pt);

/* The solve kernels keep a window's pose system on one wavefront: at most MYSLAM_BA_MAX_WINDOW_POSES key-frames per window
 * (Map.activeMap.size is 7 in every config the reference ships, config/stereo/gray/KITTI00-02.yaml:73); a larger window returns
 * MYSLAM_ERR_UNSUPPORTED from myslam_ba_optimize* (myslam_ba_build* has no such limit). */
#define MYSLAM_BA_MAX_WINDOW_POSES 10

/* Levenberg-Marquardt with Schur complement on device — replaces optimizer.optimize(n) of
 * Backend::OptimizeActiveMap (src/backend.cpp:212-214: g2o OptimizationAlgorithmLevenberg + BlockSolver_6_3 +
 * CSparse).  poses/points are updated in place.  Edges must be grouped by landmark
 * (backend.cpp:161-205 builds them that way; myslam_ba_flatten_window does); max_poses <= MYSLAM_BA_MAX_WINDOW_POSES.  d_scratch: nwin x max_edges x 18 doubles.
 * Windows too large for the all-in-LDS solver (more than ~470 landmarks at 10 key-frames) keep 22 doubles per landmark behind the edge list in
 * that scratch: the batch entry points return MYSLAM_ERR_CAPACITY when max_edges x 18 < max_edges / 2 + 22 max_pts + 10 — windows of many
 * short-lived landmarks with one or two observations each; max_edges is only a capacity, raise it.  The host-pointer entry points size
 * their scratch themselves and have no such limit. */
int myslam_ba_optimize(double* poses, int nposes, double* points, int npts, const int32_t* edge_pose, const int32_t* edge_pt,
                       const double* obs, int nedges, const uint8_t* fixed_pt, double fx, double fy, double cx, double cy,
                       double huber_delta, int max_iters, double* final_chi2, int* iters);
int myslam_ba_optimize_batch(double* d_poses, double* d_points, const int32_t* d_edge_pose, const int32_t* d_edge_pt,
                             const double* d_obs, const uint8_t* d_fixed, const int32_t* d_sizes, int nwin, int max_poses,
                             int max_pts, int max_edges, double fx, double fy, double cx, double cy, double huber_delta,
                             int max_iters, double* d_scratch, double* d_final_chi2, int32_t* d_iters, int32_t* d_status,
                             void* hip_stream);

/* The whole solve stage of Backend::OptimizeActiveMap (src/backend.cpp:208-243): up to max_rounds (5) times
 * { initializeOptimization(); optimize(iters_per_round = 10) }, stopping as soon as more than half of the edges have
 * chi2() <= chi2_th (5.991); then the outlier flags of :232-249.  edge_chi2[k] is what edge->chi2() returns there: e^T e of
 * the last error evaluation (the last Levenberg trial, accepted or not — a g2o property the reference inherits).
 * *rounds = the reference's `iteration` counter (rounds that failed the inlier test).  *rounds == max_rounds means EVERY round failed:
 * the window did NOT converge to a majority-inlier solution (the reference carries on with the result regardless, backend.cpp:232-266;
 * on such windows the Levenberg iteration is chaotic — one ulp in an observation moves the final poses by decimetres in the reference
 * arithmetic itself — so poses / flags agree with another implementation only in distribution: tests/golden/ba_chaotic_window.npz).
 * MYSLAM_BA_CONVERGED(rounds, max_rounds) spells the test.  Edges grouped by landmark, max_poses <= MYSLAM_BA_MAX_WINDOW_POSES. */
#define MYSLAM_BA_CONVERGED(rounds, max_rounds) ((rounds) < (max_rounds))
int myslam_ba_optimize_active_map(double* poses, int nposes, double* points, int npts, const int32_t* edge_pose, const int32_t* edge_pt,
                                  const double* obs, int nedges, const uint8_t* fixed_pt, double fx, double fy, double cx, double cy,
                                  double huber_delta, double chi2_th, int max_rounds, int iters_per_round,
                                  double* edge_chi2, uint8_t* outlier, int* rounds, int* n_outliers);
/* batched / device-resident form; d_edge_chi2, d_outlier: nwin x max_edges; d_rounds, d_n_outliers, d_status: nwin */
int myslam_ba_optimize_active_map_batch(double* d_poses, double* d_points, const int32_t* d_edge_pose, const int32_t* d_edge_pt,
                                        const double* d_obs, const uint8_t* d_fixed, const int32_t* d_sizes, int nwin, int max_poses,
                                        int max_pts, int max_edges, double fx, double fy, double cx, double cy, double huber_delta,
                                        double chi2_th, int max_rounds, int iters_per_round, double* d_scratch, double* d_edge_chi2,
                                        uint8_t* d_outlier, int32_t* d_rounds, int32_t* d_n_outliers, int32_t* d_status, void* hip_stream);

/* Scheduling knob of the solve kernel (process-wide, no reference counterpart; results do not depend on it beyond f64 rounding of
 * nothing: the arithmetic and its order are the same).  MYSLAM_BA_OPT_LANDMARKS_IN_HBM: 0 (default) = a window's per-landmark state lives
 * in LDS whenever it fits (133 KB for 10 key-frames x 300 landmarks: one window per CU and almost nothing beside it); 1 = always in the
 * window's HBM scratch (81 KB of LDS: the form for a solve that runs BESIDE other kernels, e.g. on the Backend's stream under the
 * extractor — its CU keeps room for their blocks).  d_scratch sizes are the same in both forms. */
#define MYSLAM_BA_OPT_LANDMARKS_IN_HBM 1
/* MYSLAM_BA_OPT_BUILD_POSE_ATOMICS (block build, myslam_ba_build[_batch]): 0 (default) = calls of fewer than 32 windows stage a window's arrays in
 * LDS and sum the pose blocks by one wave per pose over a counting-sorted edge list whenever that fits (33 bytes of LDS per edge slot + 24 per
 * landmark slot); 1 = always the form batches use (every edge's 27 pose terms added into per-wave LDS copies with ds_add_f64).  Both forms are
 * bit-reproducible; they differ from each other in the last bits of Hpp / bp (another summation order), in nothing else. */
#define MYSLAM_BA_OPT_BUILD_POSE_ATOMICS 2
int myslam_ba_set_option(int option, int value);

/* ------------------------------------------------------------------------------------------
 * Pose-only optimisation of the current frame — replaces the g2o part of Frontend::EstimateCurrentPose
 * (src/frontend.cpp:176-276): one VertexPose, one EdgeProjectionPoseOnly (include/myslam/g2o_types.h:62-100) per tracked
 * feature with a map point, Huber(1), Levenberg; `rounds` (4) x { optimize(iters = 10) over the admitted edges; classify every
 * edge by chi2 > chi2_th (5.991), exclude / re-admit }; the robust kernel is dropped after round rounds-2.   [§8(f) rank 1]
 * pose7 = (qx qy qz qw tx ty tz) Tcw in/out; pts3d n x 3 (MapPoint::Pos()); obs n x 2 (feature pixel); outlier[i] = the
 * feature's mbIsOutlier at :231-241; *n_inliers = features.size() - cntOutliers (:275).  n <= 4096.
 * pre_optimize = number of plain optimize(iters) calls before the rounds: 0 for the frontend; 1 reproduces
 * LoopClosing::OptimizeCurrentPose (src/loopclosing.cpp:339-433, extra optimize at :395-396; outlier[i] = vEdgeIsOutlier[i]).
 * ------------------------------------------------------------------------------------------ */
int myslam_pose_only_optimize(double* pose7, const double* pts3d, const double* obs, int n, double fx, double fy, double cx, double cy,
                              double chi2_th, int rounds, int iters, int pre_optimize, uint8_t* outlier, int* n_inliers);
/* `batch` frames: d_poses batch x 7, d_pts3d batch x cap x 3, d_obs batch x cap x 2, d_counts batch, d_outlier batch x cap */
int myslam_pose_only_optimize_batch(double* d_poses, const double* d_pts3d, const double* d_obs, const int32_t* d_counts, int batch, int cap,
                                    double fx, double fy, double cx, double cy, double chi2_th, int rounds, int iters, int pre_optimize,
                                    uint8_t* d_outlier, int32_t* d_n_inliers, int32_t* d_status, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Pyramidal Lucas-Kanade tracker — replaces
 *   cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(win,win), max_level,
 *                            TermCriteria(COUNT+EPS, max_iters, eps), OPTFLOW_USE_INITIAL_FLOW)
 * as called by Frontend::TrackLastFrame (src/frontend.cpp:150-153) and Frontend::FindFeaturesInRight (:358-361) with
 * win 11, max_level 3, max_iters 30, eps 0.01 (minEigThreshold = OpenCV's default 1e-4).   [SURVEY.md §8(f) rank 1]
 * next_pts is in/out: the initial flow on entry (the reference always passes one), the tracked positions on return;
 * status[i] = 1 where the flow was found; err[i] (optional) = mean absolute patch difference / 32 at level 0.
 * ------------------------------------------------------------------------------------------ */
typedef struct myslam_lk myslam_lk;
int myslam_lk_create(myslam_lk** out, int win, int max_level, int max_iters, float eps, float min_eig_threshold);
int myslam_lk_destroy(myslam_lk* h);
int myslam_lk_set_stream(myslam_lk* h, void* hip_stream);
/* host pointers (uploads, runs, downloads, synchronises); pts = n x (x, y) float */
int myslam_lk_track(myslam_lk* h, const uint8_t* prev, const uint8_t* next, int rows, int cols, int prev_step, int next_step,
                    const float* prev_pts, float* next_pts, int n, uint8_t* status, float* err);
/* The same call for a TRACKER that sees one new image per frame (Frontend::TrackLastFrame, src/frontend.cpp:129-172: the `next` image of frame
 * t is the `prev` image of frame t + 1): the handle keeps the device copy and the pyramid of the two images it saw last, named by caller
 * tokens.  A non-zero token promises "the bytes behind this token never change" — an image found under its token is neither uploaded nor
 * down-sampled again; 0 = do not look up, do not remember.  A caller that modifies an image in place (DeepLCD blurs a key-frame's image,
 * src/deeplcd.cpp:46, and cv::Mat copies share pixels) gives it a new token.  Results are bit-identical to myslam_lk_track.
 * myslam_lk_prefetch uploads an image and builds its pyramid asynchronously on the handle's stream into the slot the next
 * myslam_lk_track_cached call does not track FROM (the image must stay valid and unchanged until that call returns): the upload of frame
 * t + 1 then runs beside whatever the caller does with frame t (its pose optimisation). */
int myslam_lk_track_cached(myslam_lk* h, const uint8_t* prev, uint64_t prev_token, const uint8_t* next, uint64_t next_token, int rows, int cols,
                           int prev_step, int next_step, const float* prev_pts, float* next_pts, int n, uint8_t* status, float* err);
int myslam_lk_prefetch(myslam_lk* h, const uint8_t* img, uint64_t token, int rows, int cols, int step);
/* device pointers, asynchronous on the handle's stream: `batch` image pairs (image b at base + b*stride), points of pair b at
 * pts + b*cap*2, d_counts[b] of them valid; d_status batch x cap, d_err batch x cap or NULL */
int myslam_lk_track_batch(myslam_lk* h, const uint8_t* d_prev, const uint8_t* d_next, int batch, int rows, int cols, int step, size_t stride,
                          const float* d_prev_pts, float* d_next_pts, const int32_t* d_counts, int cap, uint8_t* d_status, float* d_err);

/* ------------------------------------------------------------------------------------------
 * Loop correction — replaces the g2o part of LoopClosing::PoseGraphOptimization (src/loopclosing.cpp:537-610) and the map-point
 * write-back that follows it (:612-640).   [SURVEY.md §8(f) rank 3]
 * poses: n x 7 (qx qy qz qw tx ty tz) Tcw of every key-frame, in/out (VertexPose, :548-565); fixed[i] != 0 for the key-frames the
 * reference fixes (:557-562: active window, loop key-frame, key-frame 0); edge k = EdgePoseGraph between poses edge_v0[k] and
 * edge_v1[k] with measurement meas[k] (7 doubles, = T[v0] * T[v1]^-1 when the edge is satisfied: mRelativePoseToLastKF :577-588,
 * mRelativePoseToLoopKF :590-601), information I6, error log(meas^-1 * T[v0] * T[v1]^-1) (include/myslam/g2o_types.h:157-167),
 * Jacobians by g2o's central differences (delta 1e-9; the analytic linearizeOplus is commented out at g2o_types.h:168-182).
 * Runs g2o's Levenberg for max_iters (20, :606) iterations.  *final_chi2 = sum of e^T e at the returned poses, *iters = iterations
 * done.  Host pointers; uploads, runs and downloads synchronously on the null stream.
 * Solver structure: key-frames in index order form the chain; every edge that does not join neighbouring free key-frames adds
 * one separator key-frame to a dense Schur block (at most 96 separators: MYSLAM_ERR_UNSUPPORTED beyond, i.e. graphs far from
 * chain + loops); long chain runs are cut by further separators so that the serial block-tridiagonal sweeps run in parallel.
 * ------------------------------------------------------------------------------------------ */
int myslam_pose_graph_optimize(double* poses, int n, const uint8_t* fixed, const int32_t* edge_v0, const int32_t* edge_v1,
                               const double* meas, int n_edges, int max_iters, double* final_chi2, int* iters);
/* src/loopclosing.cpp:621-633: points[i] <- T_new[first_kf[i]]^-1 * (T_old[first_kf[i]] * points[i]) — each map point outside the
 * active window keeps its camera-frame position in the key-frame that first observed it; first_kf[i] < 0 leaves the point alone
 * (the :625-629 skip).  old_poses / new_poses: n_poses x 7 as above; points n_points x 3 in/out. */
int myslam_correct_map_points(const double* old_poses, const double* new_poses, int n_poses, const int32_t* first_kf, double* points, int n_points);
/* LoopClosing::LoopLocalFusion, src/loopclosing.cpp:466-507 — the arithmetic of it (re-linking the current key-frame's observations to the
 * loop key-frame's map points, :509-532, is Map bookkeeping and stays with the caller).  active_poses: n_active x 7 Tcw of the active
 * key-frames, in/out; cur = index of the current key-frame among them; corrected_cur_pose7 = _mseCorrectedCurrentPose.
 * Every active key-frame moves rigidly with the current one (Ta' = Ta * Tc^-1 * Tc'); every map point i keeps its camera-frame position
 * in the active key-frame first_active_kf[i] that first observes it (< 0: left alone).  Host pointers. */
int myslam_loop_local_fusion(double* active_poses, int n_active, int cur, const double* corrected_cur_pose7, const int32_t* first_active_kf,
                             double* points, int n_points);
/* device pointers, asynchronous on hip_stream; *d_status (zeroed by the caller) receives MYSLAM_ERR_INVALID for an index >= n_poses */
int myslam_correct_map_points_device(const double* d_old_poses, const double* d_new_poses, int n_poses, const int32_t* d_first_kf,
                                     double* d_points, int n_points, int32_t* d_status, void* hip_stream);

/* ------------------------------------------------------------------------------------------
 * Loop verification — replaces cv::solvePnPRansac(vLoopPoints3d, vCurrentPoints2d, K, cv::Mat(), rvec, tvec, false, 100, 5.991, 0.99)
 * followed by cv::Rodrigues in LoopClosing::ComputeCorrectPose (src/loopclosing.cpp:262-272).   [SURVEY.md §8(f) rank 3]
 * pts3d n x 3 / pts2d n x 2 floats (cv::Point3f / cv::Point2f arrays as the reference builds them, :215-231); iterations = 100,
 * reproj_error = 5.991 (pixels), confidence = 0.99 at the call site.  pose7 = (qx qy qz qw tx ty tz): Sophus::SE3d(R, t) of :270-272.
 * inlier (optional, n flags) / *n_inliers: the RANSAC consensus set (the reference does not ask OpenCV for it).
 * OpenCV 3.4 semantics kept: cv::RNG((uint64)-1) drives 5-point samples, EPnP per sample, squared reprojection error against
 * threshold^2 in float, best-count bookkeeping with the shrinking iteration budget, then a least-squares refinement on the
 * consensus set (Levenberg-Marquardt from the RANSAC model instead of OpenCV's DLT restart: the same minimiser).
 * Returns MYSLAM_ERR_UNSUPPORTED when n < 5 or no model was found (OpenCV returns false there).  Host pointers, synchronous.
 * ------------------------------------------------------------------------------------------ */
int myslam_solve_pnp_ransac(const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx, double cy, int iterations,
                            double reproj_error, double confidence, double* pose7, uint8_t* inlier, int* n_inliers);

/* ------------------------------------------------------------------------------------------
 * Host-side formats of the reference's runner — plain host code, no device needed; the C++ forms live in
 * host/myslam_io.hpp and host/myslam_png.hpp.
 * ------------------------------------------------------------------------------------------ */
/* cv::imread(file, cv::IMREAD_GRAYSCALE) for the 8-bit single-channel PNGs of KITTI image_0 / image_1 (app/run_kitti_stereo.cpp:66-67).
 * out == NULL: size query (rows, cols only).  Non-interlaced grey 8 / 16 bit, RGB / RGBA 8 bit (BGR2GRAY fixed point); else INVALID. */
int myslam_io_read_png_gray(const char* path, uint8_t* out, size_t cap_bytes, int* rows, int* cols);
/* LoadImages (app/run_kitti_stereo.cpp:114-144): *n = number of frames listed in <sequence>/times.txt; timestamps may be NULL */
int myslam_io_load_images(const char* sequence_path, double* timestamps, int cap, int* n);
/* "<sequence>/image_0/000123.png" (right != 0: image_1), :135-141 */
int myslam_io_image_path(const char* sequence_path, int index, int right, char* buf, size_t cap);
/* System::SaveTrajectory (src/system.cpp:153-180): one line "id timestamp tx ty tz qx qy qz qw" per key-frame in ascending id order,
 * std::fixed / setprecision(6), pose = KeyFrame::Pose().inverse().  poses7_cw = Tcw as qx qy qz qw tx ty tz (what every entry point
 * of this library calls a pose). */
int myslam_io_save_trajectory(const char* path, const uint64_t* ids, const double* timestamps, const double* poses7_cw, int n);
/* System::SaveLoopEdges (src/system.cpp:188-224): two lines per loop (current key-frame, loop key-frame), ordered by the current id */
int myslam_io_save_loop_edges(const char* path, const uint64_t* cur_ids, const double* cur_ts, const double* cur_poses7_cw,
                              const uint64_t* loop_ids, const double* loop_ts, const double* loop_poses7_cw, int n);

/* ------------------------------------------------------------------------------------------
 * A whole batched step as ONE HIP graph — the per-frame call pattern of the reference (one frame at a time: src/frontend.cpp:302-328,
 * src/loopclosing.cpp:83-121) is launch-bound on a GPU: ~90 launches on 4-5 streams cost the host ~0.6 ms however few frames they carry.
 * Every *_batch entry point is asynchronous and allocation-free after its first call with a given shape, so a caller records one step
 * between begin and end — the calls it would make anyway, on the streams it would use — and replays it with one launch per step.
 *   begin: starts a thread-local capture on origin_stream and forks the side streams into it (they must be distinct from the origin);
 *   end:   joins the side streams back, ends the capture and instantiates the graph.
 * Rules: run the step once eagerly first (lazy allocations); profiling must be off (MYSLAM_ERR_UNSUPPORTED); events passed to
 * myslam_orb_set_fast_gate must be recorded INSIDE the capture (the first handle of a ring takes no gate); buffers, handles, options and
 * streams the step uses must stay as they were; host code is not replayed — record TWO consecutive steps and replay them alternately so
 * that the extractor's FAST statistics keep ping-ponging, and feed the loop database's row limits through
 * myslam_lcddb_update_query_limits.  Results are bit-identical to the eager step (tests/test_gpu_graph.py).
 * Other threads: HIP's legacy NULL stream synchronises with every blocking stream, so while ANY thread records on a blocking stream no thread may
 * issue work on the legacy stream (HIP refuses it and the recording is lost).  The library stays off the legacy stream: handles and plans, the batch
 * entry points, and the synchronous host-pointer calls (PnP-RANSAC, pose graph, map-point correction: a non-blocking stream per calling thread) —
 * tests/test_gpu_threads.py runs four threads, each recording its own one-frame calls while the others create plans.  Only the myslam_*_debug_*
 * getters copy through it.
 * ------------------------------------------------------------------------------------------ */
typedef struct myslam_step_graph myslam_step_graph;
int myslam_graph_begin(void* origin_stream, void* const* side_streams, int n_side);
int myslam_graph_end(void* origin_stream, void* const* side_streams, int n_side, myslam_step_graph** out);
int myslam_graph_launch(myslam_step_graph* g, void* hip_stream);
int myslam_graph_node_count(const myslam_step_graph* g);
int myslam_graph_destroy(myslam_step_graph* g);

#ifdef __cplusplus
}
#endif
#endif /* MYSLAM_HIP_H */
