// io.hip — the host-side formats of host/myslam_io.hpp / host/myslam_png.hpp behind the C ABI (plain host code, no device needed), so
// that a non-C++ host (tools/run_kitti_stereo.py) reads the KITTI sequences and writes the trajectory files exactly as a C++ host does.
#include <string.h>

#include <string>
#include <vector>

#include "../../include/myslam_hip.h"
#include "../host/myslam_io.hpp"
#include "../host/myslam_png.hpp"

extern "C" {

int myslam_io_read_png_gray(const char* path, uint8_t* out, size_t cap_bytes, int* rows, int* cols) {
    if (!path || !rows || !cols) return MYSLAM_ERR_INVALID;
    std::vector<uint8_t> px; int r = 0, c = 0;
    if (!myslam::io::ReadPngGray(path, px, r, c)) return MYSLAM_ERR_INVALID;
    *rows = r; *cols = c;
    if (!out) return MYSLAM_OK;                                  // size query
    if (cap_bytes < px.size()) return MYSLAM_ERR_CAPACITY;
    memcpy(out, px.data(), px.size());
    return MYSLAM_OK;
}

int myslam_io_load_images(const char* sequence_path, double* timestamps, int cap, int* n) {
    if (!sequence_path || !n) return MYSLAM_ERR_INVALID;
    std::vector<std::string> L, R; std::vector<double> T;
    *n = myslam::io::LoadImages(sequence_path, L, R, T);
    if (!timestamps) return MYSLAM_OK;
    if (cap < *n) return MYSLAM_ERR_CAPACITY;
    for (int i = 0; i < *n; i++) timestamps[i] = T[i];
    return MYSLAM_OK;
}

int myslam_io_image_path(const char* sequence_path, int index, int right, char* buf, size_t cap) {
    if (!sequence_path || !buf || index < 0 || index > 999999) return MYSLAM_ERR_INVALID;
    char name[16]; snprintf(name, sizeof name, "%06d", index);                                 // setfill('0') << setw(6), run_kitti_stereo.cpp:135-141
    const std::string p = std::string(sequence_path) + (right ? "/image_1/" : "/image_0/") + name + ".png";
    if (p.size() + 1 > cap) return MYSLAM_ERR_CAPACITY;
    memcpy(buf, p.c_str(), p.size() + 1);
    return MYSLAM_OK;
}

// Tcw (qx qy qz qw tx ty tz, the KeyFrame::Pose() the other entry points use) -> the Twc record SaveTrajectory writes (system.cpp:166-172)
static myslam::io::KeyFramePose to_record(uint64_t id, double ts, const double* p) {
    double x = p[0], y = p[1], z = p[2], w = p[3];
    const double nq = std::sqrt(x * x + y * y + z * z + w * w);
    x /= nq; y /= nq; z /= nq; w /= nq;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    myslam::io::KeyFramePose k;
    k.id = (unsigned long)id; k.timestamp = ts;
    for (int i = 0; i < 3; i++) k.t[i] = -(R[0 * 3 + i] * p[4] + R[1 * 3 + i] * p[5] + R[2 * 3 + i] * p[6]);      // -R^T t
    k.q[0] = -x; k.q[1] = -y; k.q[2] = -z; k.q[3] = w;                                                             // conjugate = rotation of Twc
    if (k.q[3] < 0) for (double& c : k.q) c = -c;       // Eigen::Quaterniond(Matrix3d) returns w >= 0
    return k;
}

int myslam_io_save_trajectory(const char* path, const uint64_t* ids, const double* timestamps, const double* poses7_cw, int n) {
    if (!path || n < 0 || (n && (!ids || !timestamps || !poses7_cw))) return MYSLAM_ERR_INVALID;
    std::vector<myslam::io::KeyFramePose> v;
    for (int i = 0; i < n; i++) v.push_back(to_record(ids[i], timestamps[i], poses7_cw + 7 * i));
    return myslam::io::SaveTrajectory(path, v) ? MYSLAM_OK : MYSLAM_ERR_INVALID;
}

int myslam_io_save_loop_edges(const char* path, const uint64_t* cur_ids, const double* cur_ts, const double* cur_poses7_cw,
                              const uint64_t* loop_ids, const double* loop_ts, const double* loop_poses7_cw, int n) {
    if (!path || n < 0 || (n && (!cur_ids || !cur_ts || !cur_poses7_cw || !loop_ids || !loop_ts || !loop_poses7_cw))) return MYSLAM_ERR_INVALID;
    std::vector<std::pair<myslam::io::KeyFramePose, myslam::io::KeyFramePose>> v;
    for (int i = 0; i < n; i++) v.emplace_back(to_record(cur_ids[i], cur_ts[i], cur_poses7_cw + 7 * i), to_record(loop_ids[i], loop_ts[i], loop_poses7_cw + 7 * i));
    return myslam::io::SaveLoopEdges(path, v) ? MYSLAM_OK : MYSLAM_ERR_INVALID;
}

}  // extern "C"
