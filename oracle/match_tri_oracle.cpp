/*
 * match_tri_oracle.cpp — CPU ORACLE (test infrastructure only).
 *
 * Hamming brute force (cv::BFMatcher(NORM_HAMMING)::match as used at src/loopclosing.cpp:33,172;
 * semantics SURVEY.md Appendix A.5 — third-party, PARITY UNPINNED but trivial), the match filter of
 * src/loopclosing.cpp:175-194, and triangulation (include/myslam/algorithm.h:16-33 — in-tree; the
 * SVD itself is Eigen bdcSvd, replaced by a one-sided Jacobi SVD, checked against LAPACK in
 * tests/test_oracle_kat.py).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

inline int popcnt256(const uint8_t* a, const uint8_t* b) {
    int d = 0;
    for (int k = 0; k < 4; k++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * k, 8);
        memcpy(&y, b + 8 * k, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

// one-sided Jacobi SVD of an m x 4 matrix (m <= 8).  A is overwritten by U*S; V (4x4) accumulated.
void jacobi_svd_mx4(double* A, int m, double V[16], double sv[4]) {
    for (int i = 0; i < 16; i++) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < m; i++) {
                    double ap = A[i * 4 + p], aq = A[i * 4 + q];
                    alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
                }
                if (gamma == 0.0) continue;
                double lim = 1e-30 + 1e-17 * sqrt(alpha * beta);
                if (fabs(gamma) <= lim) continue;
                off = std::max(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < m; i++) {
                    double ap = A[i * 4 + p], aq = A[i * 4 + q];
                    A[i * 4 + p] = c * ap - s * aq;
                    A[i * 4 + q] = s * ap + c * aq;
                }
                for (int i = 0; i < 4; i++) {
                    double vp = V[i * 4 + p], vq = V[i * 4 + q];
                    V[i * 4 + p] = c * vp - s * vq;
                    V[i * 4 + q] = s * vp + c * vq;
                }
            }
        if (off < 1e-15) break;
    }
    for (int j = 0; j < 4; j++) {
        double n = 0;
        for (int i = 0; i < m; i++) n += A[i * 4 + j] * A[i * 4 + j];
        sv[j] = sqrt(n);
    }
    // sort descending (selection sort, permuting V columns)
    for (int a = 0; a < 3; a++) {
        int best = a;
        for (int b = a + 1; b < 4; b++) if (sv[b] > sv[best]) best = b;
        if (best != a) {
            std::swap(sv[a], sv[best]);
            for (int i = 0; i < 4; i++) std::swap(V[i * 4 + a], V[i * 4 + best]);
        }
    }
}

// algorithm.h:16-33
int triangulate(const double* poses, const double* pts, int nviews, double* xyz, double* ratio) {
    if (nviews < 2 || nviews > 4) return -1;
    double A[8 * 4];
    const int m = 2 * nviews;
    for (int i = 0; i < nviews; i++) {
        const double* P = poses + 12 * i;        // row-major 3x4
        const double x = pts[3 * i + 0], y = pts[3 * i + 1];
        for (int c = 0; c < 4; c++) {
            A[(2 * i) * 4 + c] = x * P[8 + c] - P[0 + c];          // :23
            A[(2 * i + 1) * 4 + c] = y * P[8 + c] - P[4 + c];      // :24
        }
    }
    double V[16], sv[4];
    jacobi_svd_mx4(A, m, V, sv);
    const double w = V[3 * 4 + 3];
    xyz[0] = V[0 * 4 + 3] / w; xyz[1] = V[1 * 4 + 3] / w; xyz[2] = V[2 * 4 + 3] / w;   // :27
    *ratio = sv[3] / sv[2];                                                            // :29
    return 0;
}

}  // namespace

extern "C" {

int orc_hamming_match(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* train_idx, int32_t* dist) {
    if (nq < 0 || nt < 0) return -1;
    for (int i = 0; i < nq; i++) {
        int best = -1, bd = 1 << 30;
        for (int j = 0; j < nt; j++) {
            int d = popcnt256(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < bd) { bd = d; best = j; }    // first j wins ties
        }
        train_idx[i] = best;
        dist[i] = (best < 0) ? -1 : bd;
    }
    return 0;
}

int orc_hamming_filter(const int32_t* dist, int n, uint8_t* keep, int* min_dist) {
    if (n <= 0) { if (min_dist) *min_dist = 0; return 0; }
    int mn = dist[0];
    for (int i = 1; i < n; i++) mn = std::min(mn, dist[i]);
    double lim = std::max(2.0 * (double)mn, 30.0);     // loopclosing.cpp:184
    for (int i = 0; i < n; i++) keep[i] = ((double)dist[i] <= lim) ? 1 : 0;
    if (min_dist) *min_dist = mn;
    return 0;
}

int orc_triangulate(const double* poses, const double* pts, int nviews, double* xyz, double* sv_ratio) {
    return triangulate(poses, pts, nviews, xyz, sv_ratio);
}

int orc_triangulate_stereo(const float* xl, const float* yl, const float* xr, const float* yr, int n,
                           double fx, double fy, double cx, double cy, double baseline,
                           double* xyz, uint8_t* ok) {
    double poses[24] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0,
                        1, 0, 0, -baseline, 0, 1, 0, 0, 0, 0, 1, 0};      // system.cpp:108-116,141-145
    for (int i = 0; i < n; i++) {
        double pts[6] = {((double)xl[i] - cx) / fx, ((double)yl[i] - cy) / fy, 1.0,     // camera.cpp:22-26
                         ((double)xr[i] - cx) / fx, ((double)yr[i] - cy) / fy, 1.0};
        double r;
        int rc = triangulate(poses, pts, 2, xyz + 3 * i, &r);
        if (rc) return rc;
        ok[i] = (r < 1e-2 && xyz[3 * i + 2] > 0) ? 1 : 0;                    // algorithm.h:29, frontend.cpp:400
    }
    return 0;
}

}  // extern "C"
