// myslam_io.hpp — host-side formats either side of the device path, header only, no dependencies:
//   myslam::io::Config          the OpenCV-FileStorage YAML subset the reference's configs use ("%YAML:1.0", flat `key: value`
//                               lines, '#' comments)                                  src/config.cpp, include/myslam/config.h, config/*.yaml
//   myslam::io::LoadImages      KITTI sequence listing: times.txt + image_0 / image_1 paths       app/run_kitti_stereo.cpp:114-144
//   myslam::io::SaveTrajectory  "keyframe id, timestamp, tx ty tz qx qy qz qw" (std::fixed, setprecision(6))   src/system.cpp:153-180
//   myslam::io::SaveLoopEdges   two such lines per loop edge (current, then loop key-frame)                    src/system.cpp:188-224
// PNG decoding (cv::imread(file, IMREAD_GRAYSCALE) of the KITTI images) lives next door in myslam_png.hpp, equally dependency-free.
#pragma once
#include <fstream>
#include <iomanip>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace myslam {
namespace io {

class Config {
    std::map<std::string, std::string> kv_;
    static std::string trim(const std::string& s) {
        const size_t a = s.find_first_not_of(" \t\r\n\""), b = s.find_last_not_of(" \t\r\n\"");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    }
public:
    // Config::SetParameterFile: false (and nothing loaded) when the file cannot be opened
    bool SetParameterFile(const std::string& filename) {
        std::ifstream f(filename);
        if (!f.is_open()) return false;
        kv_.clear();
        std::string line;
        while (std::getline(f, line)) {
            const size_t hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            if (line.empty() || line[0] == '%' || line.compare(0, 3, "---") == 0) continue;
            const size_t colon = line.find(':');
            if (colon == std::string::npos) continue;
            const std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
            if (!key.empty() && !val.empty()) kv_[key] = val;
        }
        return true;
    }
    bool Has(const std::string& key) const { return kv_.count(key) != 0; }
    // Config::Get<T>(key): cv::FileNode conversion; a missing key converts to T() as an empty FileNode does
    template <typename T>
    T Get(const std::string& key) const {
        const auto it = kv_.find(key);
        T v = T();
        if (it == kv_.end()) return v;
        std::istringstream ss(it->second);
        ss >> v;
        return v;
    }
    size_t size() const { return kv_.size(); }
};
template <>
inline std::string Config::Get<std::string>(const std::string& key) const {
    const auto it = kv_.find(key);
    return it == kv_.end() ? std::string() : it->second;
}

// run_kitti_stereo.cpp:114-144; returns the number of frames listed in <sequence>/times.txt
inline int LoadImages(const std::string& strPathToSequence, std::vector<std::string>& vstrImageLeft,
                      std::vector<std::string>& vstrImageRight, std::vector<double>& vTimestamps) {
    vTimestamps.clear();
    std::ifstream fTimes(strPathToSequence + "/times.txt");
    std::string s;
    while (std::getline(fTimes, s)) {
        if (s.empty()) continue;
        std::stringstream ss(s);
        double t;
        if (ss >> t) vTimestamps.push_back(t);
    }
    const int n = (int)vTimestamps.size();
    vstrImageLeft.resize(n); vstrImageRight.resize(n);
    for (int i = 0; i < n; i++) {
        std::stringstream ss;
        ss << std::setfill('0') << std::setw(6) << i;
        vstrImageLeft[i] = strPathToSequence + "/image_0/" + ss.str() + ".png";
        vstrImageRight[i] = strPathToSequence + "/image_1/" + ss.str() + ".png";
    }
    return n;
}

struct KeyFramePose {            // Twc = KeyFrame::Pose().inverse(): translation, then Eigen quaternion coeffs (x y z w)
    unsigned long id; double timestamp; double t[3]; double q[4];
};

inline void write_pose_line(std::ostream& o, const KeyFramePose& k) {
    o << std::setprecision(6) << k.id << " " << k.timestamp << " " << k.t[0] << " " << k.t[1] << " " << k.t[2] << " "
      << k.q[0] << " " << k.q[1] << " " << k.q[2] << " " << k.q[3] << std::endl;
}

// system.cpp:153-180 — key-frames in ascending id order
inline bool SaveTrajectory(const std::string& save_file, const std::vector<KeyFramePose>& keyframes) {
    std::ofstream outfile(save_file, std::ios_base::out | std::ios_base::trunc);
    if (!outfile.is_open()) return false;
    outfile << std::fixed;
    std::map<unsigned long, const KeyFramePose*> sorted;
    for (const auto& k : keyframes) sorted[k.id] = &k;
    for (const auto& kv : sorted) write_pose_line(outfile, *kv.second);
    return true;
}

// system.cpp:188-224 — (current key-frame, loop key-frame) pairs, ordered by the current key-frame's id
inline bool SaveLoopEdges(const std::string& save_file, const std::vector<std::pair<KeyFramePose, KeyFramePose>>& edges) {
    std::ofstream outfile(save_file, std::ios_base::out | std::ios_base::trunc);
    if (!outfile.is_open()) return false;
    outfile << std::fixed;
    std::map<unsigned long, const std::pair<KeyFramePose, KeyFramePose>*> sorted;
    for (const auto& e : edges) sorted[e.first.id] = &e;
    for (const auto& kv : sorted) { write_pose_line(outfile, kv.second->first); write_pose_line(outfile, kv.second->second); }
    return true;
}

}  // namespace io
}  // namespace myslam
