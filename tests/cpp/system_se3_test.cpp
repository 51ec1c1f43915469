// CPU check of the SE3 helpers of host/myslam_system.hpp against chain.py's numpy functions (tests/test_chain_host.py feeds poses on
// stdin: "qx qy qz qw tx ty tz" per line, two lines per case, and compares the printed numbers).
#include <cstdio>
#include "../../a-simple-stereo-slam-system-with-deep-loop-closing_amd/host/myslam_system.hpp"

int main() {
    myslam::Pose7 a, b;
    while (std::scanf("%lf %lf %lf %lf %lf %lf %lf", &a.v[0], &a.v[1], &a.v[2], &a.v[3], &a.v[4], &a.v[5], &a.v[6]) == 7 &&
           std::scanf("%lf %lf %lf %lf %lf %lf %lf", &b.v[0], &b.v[1], &b.v[2], &b.v[3], &b.v[4], &b.v[5], &b.v[6]) == 7) {
        const myslam::Mat4 A = myslam::T_of(a), B = myslam::T_of(b);
        const myslam::Mat4 C = A * myslam::T_inv(B);                 // the product the chain forms everywhere (relative poses, distances)
        const myslam::Pose7 c = myslam::p7_of(C);
        for (int k = 0; k < 7; k++) std::printf("%.17g ", c.v[k]);
        std::printf("%.17g\n", myslam::se3_log_norm(C));
    }
    return 0;
}
