// run_kitti_stereo — BASELINE configs[0]'s entry point as a compiled program: the reference's
//     run_kitti_stereo <config.yaml> <sequence dir>                                              (app/run_kitti_stereo.cpp:20-111)
// with the per-frame dense path running through libmyslam_hip.so.  Plain C++17 over the C ABI: no OpenCV, no Eigen, no g2o, no Caffe.
//
//   run_kitti_stereo config/stereo/gray/KITTI00-02.yaml /data/kitti/sequences/00 [--frames 200] [--out result]
//                    [--calc-prototxt calc_model/deploy.prototxt --calc-model calc_model/calc.caffemodel | --calc-weights file.calcw]
//                    [--kf-every N] [--frontend-wins-race] [--correct-threshold X] [--frame-poses] [--no-prefetch]
//
// Reads <sequence>/times.txt and image_0 / image_1/%06d.png (LoadImages + cv::imread(…, IMREAD_GRAYSCALE): host/myslam_io.hpp,
// host/myslam_png.hpp), tracks every frame through host/myslam_system.hpp (Frontend / Backend / LoopClosing / Map as one sequential
// schedule), writes <out>/trajectory.txt and <out>/loopEdges.txt in the reference's format (src/system.cpp:153-224).  The CALC model
// defaults to the reference's calc_model/ files relative to the working directory (include/myslam/deeplcd.h:33).
// Built by build.py (g++, links libmyslam_hip.so); tests/test_gpu_runner.py runs it on a rendered 200-frame KITTI-layout sequence and
// compares its trajectory with the Python chain's.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <malloc.h>
#include <fstream>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "../host/myslam_io.hpp"
#include "../host/myslam_png.hpp"
#include "../host/myslam_system.hpp"

static std::shared_ptr<myslam::Image> imread_gray(const std::string& path) {
    auto im = std::make_shared<myslam::Image>();
    if (!myslam::io::ReadPngGray(path, im->px, im->rows, im->cols)) return nullptr;
    return im;
}

// cv::imread per step (app/run_kitti_stereo.cpp:66-67), decoded AHEAD of the tracker by a pool of threads: task 2 i + side = image `side` of
// frame i, handed out in order, at most `window` frames ahead of the frame the tracker has taken (a 1241 x 376 PNG takes ~4 ms to decode, a
// tracked frame 0.5 ms: see main() for the thread count).  (Round 5; one std::async per image cost the tracking loop two thread creations per frame.)
class ImageReader {
    const std::vector<std::string>&left_, &right_;
    const int n_, window_;
    std::vector<std::shared_ptr<myslam::Image>> img_;        // 2 n slots
    std::vector<char> done_;
    std::mutex mu_; std::condition_variable cv_;
    int nextTask_ = 0, taken_ = 0; bool stop_ = false;
    std::vector<std::thread> workers_;
    void work() {
        for (;;) {
            int task;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (nextTask_ < 2 * n_ && nextTask_ / 2 < taken_ + window_); });
                if (stop_) return;
                task = nextTask_++;
            }
            auto im = imread_gray((task & 1) ? right_[task >> 1] : left_[task >> 1]);
            { std::lock_guard<std::mutex> lk(mu_); img_[task] = std::move(im); done_[task] = 1; }
            cv_.notify_all();
        }
    }
public:
    ImageReader(const std::vector<std::string>& left, const std::vector<std::string>& right, int n, int window, int threads)
        : left_(left), right_(right), n_(n), window_(window), img_(2 * (size_t)n), done_(2 * (size_t)n, 0) {
        for (int i = 0; i < threads; i++) workers_.emplace_back([this] { work(); });
    }
    ~ImageReader() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); for (auto& t : workers_) t.join(); }
    // blocks until both images of frame i are decoded; the tracker has then taken frames 0 .. i
    void Take(int i, std::shared_ptr<myslam::Image>& L, std::shared_ptr<myslam::Image>& R) {
        std::unique_lock<std::mutex> lk(mu_);
        taken_ = std::max(taken_, i + 1);
        cv_.notify_all();
        cv_.wait(lk, [&] { return done_[2 * i] && done_[2 * i + 1]; });
        L = std::move(img_[2 * i]); R = std::move(img_[2 * i + 1]);
    }
    // the left image of frame i if it is decoded already (it stays in its slot for Take)
    std::shared_ptr<myslam::Image> PeekLeft(int i) {
        if (i >= n_) return nullptr;
        std::lock_guard<std::mutex> lk(mu_);
        return done_[2 * i] ? img_[2 * i] : nullptr;
    }
};

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "Usage: %s path_to_config path_to_sequence [--frames N] [--out dir] [--calc-prototxt P --calc-model M | --calc-weights F] "
                             "[--kf-every N] [--frontend-wins-race] [--correct-threshold X] [--frame-poses] [--no-prefetch]\n", argv[0]);
        return 1;
    }
    // A 1241 x 376 image is 467 KB: above glibc's mmap threshold every decoded image is its own mapping, and the tracker's release of the frame
    // before last is two munmap calls per frame — with a dozen decoder threads running, each one interrupts every core of the process (measured:
    // 0.19 ms of the tracker's 0.7 ms per frame).  Image buffers come from the heap instead and stay there.
    mallopt(M_MMAP_THRESHOLD, 32 << 20); mallopt(M_TRIM_THRESHOLD, 512 << 20);
    const std::string configPath = argv[1], sequence = argv[2];
    std::string out = "result", proto = "calc_model/deploy.prototxt", model = "calc_model/calc.caffemodel", weights;
    int frames = 0; bool framePoses = false, prefetch = true;
    myslam::SystemConfig sc;
    for (int i = 3; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(1); } return argv[++i]; };
        if (a == "--frames") frames = std::atoi(next());
        else if (a == "--out") out = next();
        else if (a == "--calc-prototxt") proto = next();
        else if (a == "--calc-model") model = next();
        else if (a == "--calc-weights") weights = next();
        else if (a == "--kf-every") sc.kfEvery = std::atoi(next());
        else if (a == "--frontend-wins-race") sc.lcdBlurReachesTracker = false;
        else if (a == "--correct-threshold") sc.correctThreshold = std::atof(next());
        else if (a == "--frame-poses") framePoses = true;
        else if (a == "--no-prefetch") prefetch = false;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
    }
    myslam::io::Config cfg;
    if (!cfg.SetParameterFile(configPath)) { std::fprintf(stderr, "parameter file %s does not exist.\n", configPath.c_str()); return 1; }
    sc.Override(cfg);
    const myslam::StereoCamera cam = myslam::StereoCamera::FromConfig(cfg);

    std::vector<std::string> left, right; std::vector<double> ts;
    const int listed = myslam::io::LoadImages(sequence, left, right, ts);
    const int n = frames <= 0 ? listed : std::min(frames, listed);
    if (n < 2) { std::fprintf(stderr, "%s: times.txt lists %d frames\n", sequence.c_str(), listed); return 1; }

    try {
        std::unique_ptr<myslam::DeepLCD> lcd(weights.empty() ? new myslam::DeepLCD(proto, model) : new myslam::DeepLCD(weights));
        myslam::StereoSystem slam(cam, sc, std::move(lcd));
        double tRead = 0.0;
        int done = 0, rows = 0, cols = 0;
        const auto t0 = std::chrono::steady_clock::now();
        // `--no-prefetch`: read the two images in the loop, as the reference does.  Otherwise a reader pool decodes some frames ahead, and a
        // following frame whose left image is ready early is handed to the tracker as `nextLeft` (uploaded beside this frame's pose optimisation)
        std::unique_ptr<ImageReader> reader;
        // (a 1241 x 376 PNG takes ~4 ms to decode and a tracked frame 0.5 ms: 16 decodes in flight keep up, a quarter of the machine's cores up to
        // 32 leave a margin; the window is what that many threads can hold in flight plus two frames)
        const int readers = std::min(32, std::max(4, (int)std::thread::hardware_concurrency() / 4));
        if (prefetch) reader.reset(new ImageReader(left, right, n, readers / 2 + 2, readers));
        for (int i = 0; i < n; i++) {
            const auto r0 = std::chrono::steady_clock::now();
            std::shared_ptr<myslam::Image> L, R, nextL;
            if (prefetch) { reader->Take(i, L, R); nextL = reader->PeekLeft(i + 1); }
            else { L = imread_gray(left[i]); R = imread_gray(right[i]); }
            tRead += std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
            if (!L || !R) { std::fprintf(stderr, "Failed to load image at: %s\n", left[i].c_str()); return 1; }
            rows = L->rows; cols = L->cols;
            if (!slam.GrabStereoImage(std::move(L), std::move(R), ts[i], std::move(nextL))) {
                std::printf("System failed, now quited (frame %d: tracking LOST)\n", i);            // app/run_kitti_stereo.cpp:83-86
                break;
            }
            done++;
        }
        const double tRun = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - tRead;
        std::filesystem::create_directories(out);
        slam.Save(out);
        if (framePoses) {
            std::ofstream f(out + "/frame_poses_cw.txt");
            f.precision(17);
            for (const auto& p : slam.framePoses) { for (int k = 0; k < 7; k++) f << p.v[k] << (k == 6 ? "\n" : " "); }
            std::ofstream g(out + "/key_frame_frames.txt");
            for (unsigned long id : slam.keyFrameFrames) g << id << "\n";
        }
        std::printf("per tracked frame: LK call %.3f ms, pose-only call %.3f ms, host bookkeeping %.3f ms (%.3f around LK, %.3f around the pose call, %.3f releasing the frame before last, %.3f "
                    "outside GrabStereoImage); stereo initialisation (first frame, with the library's lazy set-up) %.1f ms; key-frame insertion (detect, right image, triangulation, local BA, loop closer) %.2f ms per key-frame\n",
                    1e3 * slam.stats.secLK / std::max(1L, slam.stats.poseOnly), 1e3 * slam.stats.secPoseOnly / std::max(1L, slam.stats.poseOnly),
                    1e3 * (tRun - slam.stats.secLK - slam.stats.secPoseOnly - slam.stats.secKeyFrame - slam.stats.secInit) / std::max(1, done),
                    1e3 * slam.stats.secTrackHost / std::max(1L, slam.stats.poseOnly), 1e3 * slam.stats.secPoseHost / std::max(1L, slam.stats.poseOnly),
                    1e3 * slam.stats.secRelease / std::max(1, done), 1e3 * (tRun - slam.stats.secGrab) / std::max(1, done), 1e3 * slam.stats.secInit,
                    1e3 * slam.stats.secKeyFrame / std::max<size_t>(1, slam.NumKeyFrames()));
        // the reference's closing lines (app/run_kitti_stereo.cpp:101-105), for scripts that scrape them
        std::printf("\n-------\nsystem stop.\ntotal time cost: %g, average fps: %g\n", tRun + tRead, done / std::max(tRun + tRead, 1e-9));
        std::printf("%d frames (%dx%d), %zu key-frames, %zu map points, %zu loops; waited %.2f s for images, tracked + mapped in %.2f s = %.1f frames/s (%.1f after the first frame), %.1f frames/s end to end "
                    "(compiled host, one call per operator and frame; %ld pose-only, %ld local-BA, %ld DeepLCD, %ld loop queries, %ld next images uploaded ahead); wrote %s/trajectory.txt, loopEdges.txt\n",
                    done, cols, rows, slam.NumKeyFrames(), slam.NumMapPoints(), slam.NumLoops(), tRead, tRun, done / std::max(tRun, 1e-9), (done - 1) / std::max(tRun - slam.stats.secInit, 1e-9), done / std::max(tRun + tRead, 1e-9),
                    slam.stats.poseOnly, slam.stats.ba, slam.stats.lcd, slam.stats.detectLoop, slam.stats.lkPrefetched, out.c_str());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "run_kitti_stereo: %s\n", e.what());
        return 2;
    }
    return 0;
}
