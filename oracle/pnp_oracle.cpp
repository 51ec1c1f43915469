/*
 * pnp_oracle.cpp — CPU ORACLE (test infrastructure only).
 *
 * Loop verification: cv::solvePnPRansac(points3d, points2d, K, noArray(), rvec, tvec, false, 100, 5.991, 0.99) as called by
 * LoopClosing::ComputeCorrectPose (src/loopclosing.cpp:262-268).  Everything below is OpenCV 3.4.x (calib3d), absent from
 * /root/reference: restated from its published algorithm, PARITY UNPINNED.
 *   sampling     RANSACPointSetRegistrator::run / getSubset (ptsetreg.cpp): cv::RNG seeded with (uint64)-1 (multiply-with-carry,
 *                coefficient 4164903690), 5 distinct uniform indices per iteration drawn from ONE generator across iterations
 *   kernel       EPnP (Lepetit, Moreno-Noguer, Fua 2009; epnp.cpp) on the 5 sampled points: control points from the PCA of the
 *                sample, barycentric coordinates, the 4 smallest eigenvectors of M^T M, the three beta approximations each
 *                polished by 5 Gauss-Newton steps, absolute orientation (Arun), the candidate with the least mean reprojection error
 *   scoring      inlier <=> squared reprojection error (float) <= (float)(threshold^2); a model replaces the best when its count
 *                exceeds max(best, 4); the iteration budget shrinks with RANSACUpdateNumIters(confidence, outlier ratio, 5, niters)
 *   refinement   OpenCV re-solves the inliers with SOLVEPNP_ITERATIVE (DLT start + its own Levenberg-Marquardt, 20 iterations,
 *                eps = FLT_EPSILON); here: Levenberg-Marquardt on the same cost (sum of squared pixel residuals over the inliers,
 *                left-multiplied SE3 update) started from the RANSAC model and run to convergence — the same minimiser whenever
 *                both converge, which is what the caller (an initial value for OptimizeCurrentPose) relies on.
 * The small dense kernels (symmetric Jacobi eigen-solver, Householder least squares) stand in for cvSVD / cvSolve.
 */
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

// cyclic Jacobi: A (n x n symmetric, destroyed) = V diag(w) V^T; eigenvalues descending, eigenvector k = column k of V (row-major)
void jacobi_eigh(int n, double* A, double* V, double* w) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j]; }
        if (off <= 1e-30 * diag || off == 0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    for (int i = 0; i < n - 1; i++) {                      // selection sort, descending
        int m = i;
        for (int j = i + 1; j < n; j++) if (w[j] > w[m]) m = j;
        if (m != i) { std::swap(w[i], w[m]); for (int k = 0; k < n; k++) std::swap(V[k * n + i], V[k * n + m]); }
    }
}

// min |A x - b|, A m x n row-major (m >= n), Householder QR; A and b are destroyed
void ls_solve(int m, int n, double* A, double* b, double* x) {
    for (int k = 0; k < n; k++) {
        double nrm = 0;
        for (int i = k; i < m; i++) nrm += A[i * n + k] * A[i * n + k];
        nrm = sqrt(nrm);
        if (nrm == 0) continue;
        const double alpha = A[k * n + k] > 0 ? -nrm : nrm;
        double v[12];
        for (int i = k; i < m; i++) v[i] = A[i * n + k];
        v[k] -= alpha;
        double vn = 0;
        for (int i = k; i < m; i++) vn += v[i] * v[i];
        if (vn == 0) continue;
        for (int j = k; j < n; j++) {
            double d = 0;
            for (int i = k; i < m; i++) d += v[i] * A[i * n + j];
            d = 2 * d / vn;
            for (int i = k; i < m; i++) A[i * n + j] -= d * v[i];
        }
        double d = 0;
        for (int i = k; i < m; i++) d += v[i] * b[i];
        d = 2 * d / vn;
        for (int i = k; i < m; i++) b[i] -= d * v[i];
    }
    for (int k = n - 1; k >= 0; k--) {
        double s = b[k];
        for (int j = k + 1; j < n; j++) s -= A[k * n + j] * x[j];
        x[k] = A[k * n + k] != 0 ? s / A[k * n + k] : 0.0;
    }
}

struct Cam { double fu, fv, uc, vc; };

double reproj_mean(const Cam& K, const double* R, const double* t, const double* pw, const double* uv, int n) {
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double* p = pw + 3 * i;
        const double Xc = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + t[0], Yc = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + t[1];
        const double inv = 1.0 / (R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + t[2]);
        const double ue = K.uc + K.fu * Xc * inv, ve = K.vc + K.fv * Yc * inv;
        sum += sqrt((uv[2 * i] - ue) * (uv[2 * i] - ue) + (uv[2 * i + 1] - ve) * (uv[2 * i + 1] - ve));
    }
    return sum / n;
}

// EPnP on n (= 5) correspondences; returns false when the sample is degenerate
bool epnp(const Cam& K, const double* pw, const double* uv, int n, double* Rout, double* tout) {
    double cws[4][3] = {{0}};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) cws[0][j] += pw[3 * i + j];
    for (int j = 0; j < 3; j++) cws[0][j] /= n;
    double C[9] = {0};
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) C[a * 3 + b] += (pw[3 * i + a] - cws[0][a]) * (pw[3 * i + b] - cws[0][b]);
    double V3[9], w3[3];
    jacobi_eigh(3, C, V3, w3);
    for (int i = 1; i < 4; i++) {
        const double k = sqrt(std::max(w3[i - 1], 0.0) / n);
        for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * V3[j * 3 + (i - 1)];
    }
    // barycentric coordinates
    double cc[9];
    for (int j = 0; j < 3; j++) for (int i = 1; i < 4; i++) cc[3 * j + i - 1] = cws[i][j] - cws[0][j];
    const double det = cc[0] * (cc[4] * cc[8] - cc[5] * cc[7]) - cc[1] * (cc[3] * cc[8] - cc[5] * cc[6]) + cc[2] * (cc[3] * cc[7] - cc[4] * cc[6]);
    if (!(fabs(det) > 1e-300)) return false;
    const double ci[9] = {(cc[4] * cc[8] - cc[5] * cc[7]) / det, (cc[2] * cc[7] - cc[1] * cc[8]) / det, (cc[1] * cc[5] - cc[2] * cc[4]) / det,
                          (cc[5] * cc[6] - cc[3] * cc[8]) / det, (cc[0] * cc[8] - cc[2] * cc[6]) / det, (cc[2] * cc[3] - cc[0] * cc[5]) / det,
                          (cc[3] * cc[7] - cc[4] * cc[6]) / det, (cc[1] * cc[6] - cc[0] * cc[7]) / det, (cc[0] * cc[4] - cc[1] * cc[3]) / det};
    std::vector<double> alphas(4 * n);
    for (int i = 0; i < n; i++) {
        double* a = &alphas[4 * i];
        for (int j = 0; j < 3; j++)
            a[1 + j] = ci[3 * j] * (pw[3 * i] - cws[0][0]) + ci[3 * j + 1] * (pw[3 * i + 1] - cws[0][1]) + ci[3 * j + 2] * (pw[3 * i + 2] - cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    // M^T M and its eigenvectors
    double MtM[144] = {0};
    for (int i = 0; i < n; i++) {
        double m1[12], m2[12];
        const double* a = &alphas[4 * i];
        for (int j = 0; j < 4; j++) {
            m1[3 * j] = a[j] * K.fu; m1[3 * j + 1] = 0.0; m1[3 * j + 2] = a[j] * (K.uc - uv[2 * i]);
            m2[3 * j] = 0.0; m2[3 * j + 1] = a[j] * K.fv; m2[3 * j + 2] = a[j] * (K.vc - uv[2 * i + 1]);
        }
        for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) MtM[r * 12 + c] += m1[r] * m1[c] + m2[r] * m2[c];
    }
    double V[144], w[12];
    jacobi_eigh(12, MtM, V, w);
    double v[4][12];                                      // the eigenvectors of the 4 smallest eigenvalues, smallest first
    for (int k = 0; k < 4; k++) for (int r = 0; r < 12; r++) v[k][r] = V[r * 12 + (11 - k)];
    // L (6 x 10) and rho
    static const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double L[60], rho[6];
    for (int j = 0; j < 6; j++) {
        double dv[4][3];
        for (int k = 0; k < 4; k++) for (int c = 0; c < 3; c++) dv[k][c] = v[k][3 * pa[j] + c] - v[k][3 * pb[j] + c];
        auto dot = [&](int a, int b) { return dv[a][0] * dv[b][0] + dv[a][1] * dv[b][1] + dv[a][2] * dv[b][2]; };
        double* r = L + 10 * j;
        r[0] = dot(0, 0); r[1] = 2 * dot(0, 1); r[2] = dot(1, 1); r[3] = 2 * dot(0, 2); r[4] = 2 * dot(1, 2);
        r[5] = dot(2, 2); r[6] = 2 * dot(0, 3); r[7] = 2 * dot(1, 3); r[8] = 2 * dot(2, 3); r[9] = dot(3, 3);
        rho[j] = 0;
        for (int c = 0; c < 3; c++) rho[j] += (cws[pa[j]][c] - cws[pb[j]][c]) * (cws[pa[j]][c] - cws[pb[j]][c]);
    }
    double best = 1e300;
    bool any = false;
    for (int variant = 1; variant <= 3; variant++) {
        double betas[4] = {0, 0, 0, 0};
        if (variant == 1) {                                // betas10 columns [B11 B12 B13 B14]
            static const int cols[4] = {0, 1, 3, 6};
            double A[24], b[6], x[4];
            for (int j = 0; j < 6; j++) { for (int c = 0; c < 4; c++) A[j * 4 + c] = L[10 * j + cols[c]]; b[j] = rho[j]; }
            ls_solve(6, 4, A, b, x);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = -x[1] / betas[0]; betas[2] = -x[2] / betas[0]; betas[3] = -x[3] / betas[0]; }
            else { betas[0] = sqrt(x[0]); betas[1] = x[1] / betas[0]; betas[2] = x[2] / betas[0]; betas[3] = x[3] / betas[0]; }
        } else if (variant == 2) {                         // [B11 B12 B22]
            double A[18], b[6], x[3];
            for (int j = 0; j < 6; j++) { for (int c = 0; c < 3; c++) A[j * 3 + c] = L[10 * j + c]; b[j] = rho[j]; }
            ls_solve(6, 3, A, b, x);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { betas[0] = sqrt(x[0]); betas[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) betas[0] = -betas[0];
        } else {                                           // [B11 B12 B22 B13 B23]
            double A[30], b[6], x[5];
            for (int j = 0; j < 6; j++) { for (int c = 0; c < 5; c++) A[j * 5 + c] = L[10 * j + c]; b[j] = rho[j]; }
            ls_solve(6, 5, A, b, x);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { betas[0] = sqrt(x[0]); betas[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) betas[0] = -betas[0];
            betas[2] = x[3] / betas[0];
        }
        for (int it = 0; it < 5; it++) {                   // Gauss-Newton on the 6 distance constraints
            double A[24], b[6], x[4];
            for (int j = 0; j < 6; j++) {
                const double* r = L + 10 * j;
                A[j * 4 + 0] = 2 * r[0] * betas[0] + r[1] * betas[1] + r[3] * betas[2] + r[6] * betas[3];
                A[j * 4 + 1] = r[1] * betas[0] + 2 * r[2] * betas[1] + r[4] * betas[2] + r[7] * betas[3];
                A[j * 4 + 2] = r[3] * betas[0] + r[4] * betas[1] + 2 * r[5] * betas[2] + r[8] * betas[3];
                A[j * 4 + 3] = r[6] * betas[0] + r[7] * betas[1] + r[8] * betas[2] + 2 * r[9] * betas[3];
                b[j] = rho[j] - (r[0] * betas[0] * betas[0] + r[1] * betas[0] * betas[1] + r[2] * betas[1] * betas[1] + r[3] * betas[0] * betas[2] +
                                 r[4] * betas[1] * betas[2] + r[5] * betas[2] * betas[2] + r[6] * betas[0] * betas[3] + r[7] * betas[1] * betas[3] +
                                 r[8] * betas[2] * betas[3] + r[9] * betas[3] * betas[3]);
            }
            ls_solve(6, 4, A, b, x);
            for (int k = 0; k < 4; k++) betas[k] += x[k];
        }
        // control points and sample points in the camera frame
        double ccs[4][3];
        for (int i = 0; i < 4; i++) for (int c = 0; c < 3; c++) ccs[i][c] = betas[0] * v[0][3 * i + c] + betas[1] * v[1][3 * i + c] + betas[2] * v[2][3 * i + c] + betas[3] * v[3][3 * i + c];
        std::vector<double> pcs(3 * n);
        for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) {
            const double* a = &alphas[4 * i];
            pcs[3 * i + c] = a[0] * ccs[0][c] + a[1] * ccs[1][c] + a[2] * ccs[2][c] + a[3] * ccs[3][c];
        }
        if (pcs[2] < 0) for (auto& p : pcs) p = -p;
        // absolute orientation (Arun): R = U V^T of ABt = sum (pc - pc0)(pw - pw0)^T
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
        for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) { pc0[c] += pcs[3 * i + c]; pw0[c] += pw[3 * i + c]; }
        for (int c = 0; c < 3; c++) { pc0[c] /= n; pw0[c] /= n; }
        double ABt[9] = {0};
        for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) ABt[a * 3 + b] += (pcs[3 * i + a] - pc0[a]) * (pw[3 * i + b] - pw0[b]);
        double AtA[9], Vr[9], s2[3];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) AtA[a * 3 + b] = ABt[0 * 3 + a] * ABt[0 * 3 + b] + ABt[1 * 3 + a] * ABt[1 * 3 + b] + ABt[2 * 3 + a] * ABt[2 * 3 + b];
        jacobi_eigh(3, AtA, Vr, s2);
        double U[9];                                       // columns u_k = ABt v_k / s_k; the last one completed by the cross product
        bool ok = true;
        for (int k = 0; k < 2; k++) {
            double u[3], nn = 0;
            for (int a = 0; a < 3; a++) { u[a] = ABt[a * 3] * Vr[0 * 3 + k] + ABt[a * 3 + 1] * Vr[1 * 3 + k] + ABt[a * 3 + 2] * Vr[2 * 3 + k]; nn += u[a] * u[a]; }
            nn = sqrt(nn);
            if (!(nn > 1e-300)) { ok = false; break; }
            for (int a = 0; a < 3; a++) U[a * 3 + k] = u[a] / nn;
        }
        if (!ok) continue;
        {
            double u[3], nn = 0;
            for (int a = 0; a < 3; a++) { u[a] = ABt[a * 3] * Vr[0 * 3 + 2] + ABt[a * 3 + 1] * Vr[1 * 3 + 2] + ABt[a * 3 + 2] * Vr[2 * 3 + 2]; nn += u[a] * u[a]; }
            nn = sqrt(nn);
            const double cx = U[3] * U[7] - U[6] * U[4], cy = U[6] * U[1] - U[0] * U[7], cz = U[0] * U[4] - U[3] * U[1];      // u0 x u1
            if (nn > 1e-12 * sqrt(std::max(s2[0], 0.0))) for (int a = 0; a < 3; a++) U[a * 3 + 2] = u[a] / nn;
            else { U[2] = cx; U[5] = cy; U[8] = cz; }
        }
        double R[9], t[3];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) R[a * 3 + b] = U[a * 3] * Vr[b * 3] + U[a * 3 + 1] * Vr[b * 3 + 1] + U[a * 3 + 2] * Vr[b * 3 + 2];
        const double dR = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (dR < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
        for (int a = 0; a < 3; a++) t[a] = pc0[a] - (R[a * 3] * pw0[0] + R[a * 3 + 1] * pw0[1] + R[a * 3 + 2] * pw0[2]);
        const double err = reproj_mean(K, R, t, pw, uv, n);
        if (std::isfinite(err) && err < best) { best = err; memcpy(Rout, R, sizeof(R)); memcpy(tout, t, sizeof(t)); any = true; }
    }
    return any;
}

struct CvRng {                                             // cv::RNG
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffu) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

int ransac_update_iters(double p, double ep, int model_points, int max_iters) {
    p = std::min(std::max(p, 0.0), 1.0); ep = std::min(std::max(ep, 0.0), 1.0);
    double num = std::max(1.0 - p, DBL_MIN), denom = 1.0 - pow(1.0 - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

void R_to_q(const double* R, double* q) {
    const double tr = R[0] + R[4] + R[8];
    double x, y, z, w;
    if (tr > 0) { const double s = sqrt(tr + 1.0) * 2; w = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s; }
    else if (R[0] > R[4] && R[0] > R[8]) { const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; x = 0.25 * s; w = (R[7] - R[5]) / s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s; }
    else if (R[4] > R[8]) { const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; y = 0.25 * s; w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; z = (R[5] + R[7]) / s; }
    else { const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; z = 0.25 * s; w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

// exp of (upsilon, omega) as rotation matrix + translation (Rodrigues / SE3 V matrix)
void se3_exp_R(const double* d, double* R, double* t) {
    const double wx = d[3], wy = d[4], wz = d[5], th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double A, B, C;
    if (th < 1e-8) { A = 1 - th2 / 6; B = 0.5 - th2 / 24; C = 1.0 / 6 - th2 / 120; }
    else { A = sin(th) / th; B = (1 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W2[i * 3 + j] = W[i * 3] * W[j] + W[i * 3 + 1] * W[3 + j] + W[i * 3 + 2] * W[6 + j];
    double V[9];
    for (int i = 0; i < 9; i++) { const double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + A * W[i] + B * W2[i]; V[i] = I + B * W[i] + C * W2[i]; }
    for (int i = 0; i < 3; i++) t[i] = V[i * 3] * d[0] + V[i * 3 + 1] * d[1] + V[i * 3 + 2] * d[2];
}

double refine_cost(const Cam& K, const double* R, const double* t, const float* p3, const float* p2, const uint8_t* mask, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) {
        if (!mask[i]) continue;
        const double X = p3[3 * i], Y = p3[3 * i + 1], Z = p3[3 * i + 2];
        const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0], yc = R[3] * X + R[4] * Y + R[5] * Z + t[1], zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        const double eu = p2[2 * i] - (K.fu * xc / zc + K.uc), ev = p2[2 * i + 1] - (K.fv * yc / zc + K.vc);
        s += eu * eu + ev * ev;
    }
    return s;
}

}  // namespace

extern "C" {

int orc_epnp(const double* pw, const double* uv, int n, double fx, double fy, double cx, double cy, double* R9, double* t3) {
    if (n < 4 || n > 64) return -1;
    const Cam K{fx, fy, cx, cy};
    return epnp(K, pw, uv, n, R9, t3) ? 0 : -2;
}

int orc_cv_rng_uniform(uint64_t seed, int a, int b, int count, int32_t* out) {
    CvRng r(seed);
    for (int i = 0; i < count; i++) out[i] = r.uniform(a, b);
    return 0;
}

int orc_solve_pnp_ransac(const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx, double cy, int iterations,
                         double reproj_error, double confidence, double* pose7, uint8_t* inlier, int* n_inliers) {
    if (n < 0 || iterations < 1 || (n > 0 && (!pts3d || !pts2d)) || !pose7) return -1;
    if (n_inliers) *n_inliers = 0;
    if (inlier) memset(inlier, 0, (size_t)n);
    const int MP = 5;
    if (n < MP) return -2;                                 // cv: CV_Assert / "count < modelPoints -> false"
    const Cam K{fx, fy, cx, cy};
    CvRng rng((uint64_t)-1);
    std::vector<uint8_t> mask(n), best_mask(n, 0);
    double bestR[9], bestt[3];
    int max_good = 0, niters = iterations;
    const float thr = (float)(reproj_error * reproj_error);
    for (int iter = 0; iter < niters; iter++) {
        int idx[MP];
        for (int i = 0; i < MP; i++) {
            for (;;) {
                idx[i] = rng.uniform(0, n);
                int j = 0;
                for (; j < i; j++) if (idx[j] == idx[i]) break;
                if (j == i) break;
            }
        }
        double pw[3 * MP], uv[2 * MP], R[9], t[3];
        for (int i = 0; i < MP; i++) {
            for (int c = 0; c < 3; c++) pw[3 * i + c] = pts3d[3 * idx[i] + c];
            for (int c = 0; c < 2; c++) uv[2 * i + c] = pts2d[2 * idx[i] + c];
        }
        if (!epnp(K, pw, uv, MP, R, t)) continue;
        int good = 0;
        for (int i = 0; i < n; i++) {
            const double X = pts3d[3 * i], Y = pts3d[3 * i + 1], Z = pts3d[3 * i + 2];
            const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0], yc = R[3] * X + R[4] * Y + R[5] * Z + t[1], zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
            const float pu = (float)(K.fu * xc / zc + K.uc), pv = (float)(K.fv * yc / zc + K.vc);
            const float du = pts2d[2 * i] - pu, dv = pts2d[2 * i + 1] - pv;
            const float e = du * du + dv * dv;
            mask[i] = e <= thr;                            // NaN (zc == 0) compares false
            good += mask[i];
        }
        if (good > std::max(max_good, MP - 1)) {
            best_mask = mask; memcpy(bestR, R, sizeof(R)); memcpy(bestt, t, sizeof(t));
            max_good = good;
            niters = ransac_update_iters(confidence, (double)(n - good) / n, MP, niters);
        }
    }
    if (max_good <= 0) return -3;
    // refinement on the inliers: Levenberg-Marquardt, left-multiplied update, start = the RANSAC model
    double R[9], t[3];
    memcpy(R, bestR, sizeof(R)); memcpy(t, bestt, sizeof(t));
    double cost = refine_cost(K, R, t, pts3d, pts2d, best_mask.data(), n), lambda = 1e-3;
    for (int it = 0; it < 50; it++) {
        double H[36] = {0}, g[6] = {0};
        for (int i = 0; i < n; i++) {
            if (!best_mask[i]) continue;
            const double X = pts3d[3 * i], Y = pts3d[3 * i + 1], Z = pts3d[3 * i + 2];
            const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0], yc = R[3] * X + R[4] * Y + R[5] * Z + t[1], zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
            const double zi = 1.0 / zc, zi2 = zi * zi;
            const double eu = pts2d[2 * i] - (K.fu * xc * zi + K.uc), ev = pts2d[2 * i + 1] - (K.fv * yc * zi + K.vc);
            const double J[12] = {-K.fu * zi, 0, K.fu * xc * zi2, K.fu * xc * yc * zi2, -K.fu - K.fu * xc * xc * zi2, K.fu * yc * zi,
                                  0, -K.fv * zi, K.fv * yc * zi2, K.fv + K.fv * yc * yc * zi2, -K.fv * xc * yc * zi2, -K.fv * xc * zi};
            for (int r = 0; r < 6; r++) {
                for (int c = 0; c < 6; c++) H[r * 6 + c] += J[r] * J[c] + J[6 + r] * J[6 + c];
                g[r] -= J[r] * eu + J[6 + r] * ev;
            }
        }
        bool improved = false;
        double dxn = 0;
        for (int trial = 0; trial < 10 && !improved; trial++) {
            double A[36], x[6];
            memcpy(A, H, sizeof(A));
            for (int a = 0; a < 6; a++) A[a * 7] += lambda * (H[a * 7] > 0 ? H[a * 7] : 1.0);
            bool ok = true;
            for (int j = 0; j < 6 && ok; j++) {
                double d = A[j * 6 + j];
                for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k];
                if (!(d > 0)) { ok = false; break; }
                A[j * 6 + j] = sqrt(d);
                for (int i = j + 1; i < 6; i++) {
                    double v = A[i * 6 + j];
                    for (int k = 0; k < j; k++) v -= A[i * 6 + k] * A[j * 6 + k];
                    A[i * 6 + j] = v / A[j * 6 + j];
                }
            }
            if (ok) {
                for (int i = 0; i < 6; i++) { double v = g[i]; for (int k = 0; k < i; k++) v -= A[i * 6 + k] * x[k]; x[i] = v / A[i * 7]; }
                for (int i = 5; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < 6; k++) v -= A[k * 6 + i] * x[k]; x[i] = v / A[i * 7]; }
                double dR[9], dt[3], Rn[9], tn[3];
                se3_exp_R(x, dR, dt);
                for (int a = 0; a < 3; a++) {
                    for (int b = 0; b < 3; b++) Rn[a * 3 + b] = dR[a * 3] * R[b] + dR[a * 3 + 1] * R[3 + b] + dR[a * 3 + 2] * R[6 + b];
                    tn[a] = dR[a * 3] * t[0] + dR[a * 3 + 1] * t[1] + dR[a * 3 + 2] * t[2] + dt[a];
                }
                const double c2 = refine_cost(K, Rn, tn, pts3d, pts2d, best_mask.data(), n);
                if (std::isfinite(c2) && c2 <= cost) {
                    memcpy(R, Rn, sizeof(R)); memcpy(t, tn, sizeof(t)); cost = c2; improved = true;
                    lambda = std::max(lambda * 0.1, 1e-12);
                    dxn = 0; for (int a = 0; a < 6; a++) dxn = std::max(dxn, fabs(x[a]));
                    continue;
                }
            }
            lambda *= 10;
        }
        if (!improved || dxn < 1e-12) break;
    }
    R_to_q(R, pose7);
    for (int k = 0; k < 3; k++) pose7[4 + k] = t[k];
    if (inlier) memcpy(inlier, best_mask.data(), n);
    if (n_inliers) *n_inliers = max_good;
    return 0;
}

}  // extern "C"
