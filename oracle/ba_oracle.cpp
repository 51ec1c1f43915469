/*
 * ba_oracle.cpp — CPU ORACLE (test infrastructure only).
 *
 * Sliding-window local BA.  In-tree arithmetic followed verbatim:
 *   EdgeProjection::computeError / linearizeOplus   include/myslam/g2o_types.h:115-144
 *   VertexPose::oplusImpl (T <- exp(d) * T, d=(v,w)) include/myslam/g2o_types.h:32-37
 *   VertexXYZ::oplusImpl                             include/myslam/g2o_types.h:50-54
 *   graph construction conventions                   src/backend.cpp:126-206 (left cam, ext = I,
 *                                                    information = I2, Huber delta = 5.991)
 * g2o internals (Huber, quadratic form, Schur, Levenberg) restated from SURVEY.md Appendix A.7 —
 * third-party, PARITY UNPINNED.  Jacobians are checked against finite differences in tests.
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct Pose { double R[9]; double t[3]; };

void quat_to_R(const double* q, double* R) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double n = sqrt(x * x + y * y + z * z + w * w);
    x /= n; y /= n; z /= n; w /= n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

void R_to_quat(const double* R, double* q) {
    double tr = R[0] + R[4] + R[8];
    double x, y, z, w;
    if (tr > 0) {
        double s = sqrt(tr + 1.0) * 2; w = 0.25 * s;
        x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; x = 0.25 * s;
        w = (R[7] - R[5]) / s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; y = 0.25 * s;
        w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; z = (R[5] + R[7]) / s;
    } else {
        double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; z = 0.25 * s;
        w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s;
    }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

// Sophus SE3d::exp, tangent = (upsilon, omega)
void se3_exp(const double* d, double* R, double* t) {
    const double wx = d[3], wy = d[4], wz = d[5];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double A, B, C;   // R = I + A W + B W^2 ; V = I + B W + C W^2
    if (th < 1e-8) { A = 1 - th2 / 6; B = 0.5 - th2 / 24; C = 1.0 / 6 - th2 / 120; }
    else { A = sin(th) / th; B = (1 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += W[i * 3 + k] * W[k * 3 + j];
            W2[i * 3 + j] = s;
        }
    double V[9];
    for (int i = 0; i < 9; i++) {
        double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + A * W[i] + B * W2[i];
        V[i] = I + B * W[i] + C * W2[i];
    }
    for (int i = 0; i < 3; i++) t[i] = V[i * 3] * d[0] + V[i * 3 + 1] * d[1] + V[i * 3 + 2] * d[2];
}

void pose_oplus(Pose& T, const double* d) {       // g2o_types.h:32-37
    double Rd[9], td[3];
    se3_exp(d, Rd, td);
    Pose N;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += Rd[i * 3 + k] * T.R[k * 3 + j];
            N.R[i * 3 + j] = s;
        }
        N.t[i] = Rd[i * 3] * T.t[0] + Rd[i * 3 + 1] * T.t[1] + Rd[i * 3 + 2] * T.t[2] + td[i];
    }
    T = N;
}

struct Cam { double fx, fy, cx, cy; };

// g2o_types.h:115-144; returns e (2), Jxi (2x6), Jp (2x3)
inline void edge_eval(const Pose& T, const double* pw, const double* z, const Cam& K,
                      double* e, double* Jxi, double* Jp) {
    double pc[3];
    for (int i = 0; i < 3; i++) pc[i] = T.R[i * 3] * pw[0] + T.R[i * 3 + 1] * pw[1] + T.R[i * 3 + 2] * pw[2] + T.t[i];
    const double X = pc[0], Y = pc[1], Z = pc[2];
    e[0] = z[0] - (K.fx * X / Z + K.cx);          // K*p / p.z  (:119-121)
    e[1] = z[1] - (K.fy * Y / Z + K.cy);
    if (!Jxi) return;
    const double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;     // :133-134
    Jxi[0] = -K.fx * Zinv; Jxi[1] = 0; Jxi[2] = K.fx * X * Zinv2; Jxi[3] = K.fx * X * Y * Zinv2;
    Jxi[4] = -K.fx - K.fx * X * X * Zinv2; Jxi[5] = K.fx * Y * Zinv;
    Jxi[6] = 0; Jxi[7] = -K.fy * Zinv; Jxi[8] = K.fy * Y * Zinv2; Jxi[9] = K.fy + K.fy * Y * Y * Zinv2;
    Jxi[10] = -K.fy * X * Y * Zinv2; Jxi[11] = -K.fy * X * Zinv;
    for (int r = 0; r < 2; r++)                                     // :140-141 (cam_ext = I)
        for (int c = 0; c < 3; c++)
            Jp[r * 3 + c] = Jxi[r * 6 + 0] * T.R[0 * 3 + c] + Jxi[r * 6 + 1] * T.R[1 * 3 + c] + Jxi[r * 6 + 2] * T.R[2 * 3 + c];
}

// RobustKernelHuber::robustify: rho0 (value), rho1 (weight)
inline void huber(double e2, double delta, double& rho0, double& rho1) {
    const double dsqr = delta * delta;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
    else { double sq = sqrt(e2); rho0 = 2 * sq * delta - dsqr; rho1 = delta / sq; }
}

struct Problem {
    std::vector<Pose> poses; std::vector<double> pts;
    const int32_t* ep; const int32_t* el; const double* obs; int ne;
    const uint8_t* fixed; Cam K; double delta;
};

// computeActiveErrors() + activeRobustChi2(); edge_chi2 (optional) receives what edge->chi2() would return afterwards
double robust_chi2(const Problem& P, double* edge_chi2 = nullptr) {
    double s = 0;
    for (int k = 0; k < P.ne; k++) {
        double e[2];
        edge_eval(P.poses[P.ep[k]], &P.pts[3 * P.el[k]], P.obs + 2 * k, P.K, e, nullptr, nullptr);
        double r0, r1; huber(e[0] * e[0] + e[1] * e[1], P.delta, r0, r1);
        if (edge_chi2) edge_chi2[k] = e[0] * e[0] + e[1] * e[1];
        s += r0;
    }
    return s;
}

void build(const Problem& P, double* Hpp, double* Hll, double* Hpl, double* bp, double* bl, double* chi2_raw) {
    const int np = (int)P.poses.size(), nl = (int)P.pts.size() / 3;
    memset(Hpp, 0, sizeof(double) * np * 36);
    memset(Hll, 0, sizeof(double) * nl * 9);
    memset(Hpl, 0, sizeof(double) * P.ne * 18);
    memset(bp, 0, sizeof(double) * np * 6);
    memset(bl, 0, sizeof(double) * nl * 3);
    for (int k = 0; k < P.ne; k++) {
        const int ip = P.ep[k], il = P.el[k];
        double e[2], J[12], Jp[6];
        edge_eval(P.poses[ip], &P.pts[3 * il], P.obs + 2 * k, P.K, e, J, Jp);
        const double e2 = e[0] * e[0] + e[1] * e[1];
        if (chi2_raw) chi2_raw[k] = e2;
        double r0, w; huber(e2, P.delta, r0, w);
        for (int a = 0; a < 6; a++) {
            for (int b = 0; b < 6; b++) Hpp[ip * 36 + a * 6 + b] += w * (J[a] * J[b] + J[6 + a] * J[6 + b]);
            bp[ip * 6 + a] += -w * (J[a] * e[0] + J[6 + a] * e[1]);
        }
        if (P.fixed && P.fixed[il]) continue;
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++) Hll[il * 9 + a * 3 + b] += w * (Jp[a] * Jp[b] + Jp[3 + a] * Jp[3 + b]);
            bl[il * 3 + a] += -w * (Jp[a] * e[0] + Jp[3 + a] * e[1]);
        }
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 3; b++) Hpl[k * 18 + a * 3 + b] = w * (J[a] * Jp[b] + J[6 + a] * Jp[3 + b]);
    }
}

bool inv3(const double* A, double* I) {
    double det = A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
    if (det == 0 || !std::isfinite(det)) return false;
    double id = 1.0 / det;
    I[0] = (A[4] * A[8] - A[5] * A[7]) * id; I[1] = (A[2] * A[7] - A[1] * A[8]) * id; I[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    I[3] = (A[5] * A[6] - A[3] * A[8]) * id; I[4] = (A[0] * A[8] - A[2] * A[6]) * id; I[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    I[6] = (A[3] * A[7] - A[4] * A[6]) * id; I[7] = (A[1] * A[6] - A[0] * A[7]) * id; I[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return true;
}

// dense Cholesky solve, n x n SPD
bool chol_solve(std::vector<double>& A, std::vector<double>& b, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k]; b[i] = s / A[i * n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k]; b[i] = s / A[i * n + i]; }
    return true;
}

// Solve (H + lambda I) x = b via Schur complement on the landmarks (BlockSolver_6_3 semantics)
bool schur_solve(const Problem& P, const double* Hpp, const double* Hll, const double* Hpl, const double* bp,
                 const double* bl, double lambda, std::vector<double>& xp, std::vector<double>& xl) {
    const int np = (int)P.poses.size(), nl = (int)P.pts.size() / 3, n = 6 * np;
    std::vector<double> S((size_t)n * n, 0.0), rhs(bp, bp + n);
    for (int p = 0; p < np; p++)
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) S[(size_t)(6 * p + a) * n + 6 * p + b] = Hpp[p * 36 + a * 6 + b] + (a == b ? lambda : 0.0);
    std::vector<double> Hinv((size_t)nl * 9, 0.0);
    std::vector<char> active(nl, 0);
    // group edges per landmark
    std::vector<std::vector<int>> byl(nl);
    for (int k = 0; k < P.ne; k++) byl[P.el[k]].push_back(k);
    for (int l = 0; l < nl; l++) {
        if ((P.fixed && P.fixed[l]) || byl[l].empty()) continue;
        double A[9];
        for (int i = 0; i < 9; i++) A[i] = Hll[l * 9 + i] + ((i % 4 == 0) ? lambda : 0.0);
        if (!inv3(A, &Hinv[l * 9])) return false;
        active[l] = 1;
        // W_p = sum of Hpl over edges (p,l)  (normally one)
        std::vector<int> ps;
        std::vector<double> Wp;
        for (int k : byl[l]) {
            int p = P.ep[k];
            size_t idx = std::find(ps.begin(), ps.end(), p) - ps.begin();
            if (idx == ps.size()) { ps.push_back(p); Wp.resize(Wp.size() + 18, 0.0); }
            for (int i = 0; i < 18; i++) Wp[idx * 18 + i] += Hpl[k * 18 + i];
        }
        for (size_t i1 = 0; i1 < ps.size(); i1++) {
            double WH[18];   // W1 * Hinv (6x3)
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int c = 0; c < 3; c++) s += Wp[i1 * 18 + a * 3 + c] * Hinv[l * 9 + c * 3 + b];
                    WH[a * 3 + b] = s;
                }
            for (int a = 0; a < 6; a++) {
                double s = 0;
                for (int c = 0; c < 3; c++) s += WH[a * 3 + c] * bl[l * 3 + c];
                rhs[6 * ps[i1] + a] -= s;
            }
            for (size_t i2 = 0; i2 < ps.size(); i2++)
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 6; b++) {
                        double s = 0;
                        for (int c = 0; c < 3; c++) s += WH[a * 3 + c] * Wp[i2 * 18 + b * 3 + c];
                        S[(size_t)(6 * ps[i1] + a) * n + 6 * ps[i2] + b] -= s;
                    }
        }
    }
    if (!chol_solve(S, rhs, n)) return false;
    xp = rhs;
    xl.assign((size_t)nl * 3, 0.0);
    for (int l = 0; l < nl; l++) {
        if (!active[l]) continue;
        double r[3] = {bl[l * 3], bl[l * 3 + 1], bl[l * 3 + 2]};
        for (int k : byl[l]) {
            int p = P.ep[k];
            for (int b = 0; b < 3; b++) {
                double s = 0;
                for (int a = 0; a < 6; a++) s += Hpl[k * 18 + a * 3 + b] * xp[6 * p + a];
                r[b] -= s;
            }
        }
        for (int a = 0; a < 3; a++) xl[l * 3 + a] = Hinv[l * 9 + a * 3] * r[0] + Hinv[l * 9 + a * 3 + 1] * r[1] + Hinv[l * 9 + a * 3 + 2] * r[2];
    }
    return true;
}

void load_problem(Problem& P, const double* poses, int nposes, const double* points, int npts,
                  const int32_t* ep, const int32_t* el, const double* obs, int ne, const uint8_t* fixed,
                  double fx, double fy, double cx, double cy, double delta) {
    P.poses.resize(nposes);
    for (int i = 0; i < nposes; i++) {
        quat_to_R(poses + 7 * i, P.poses[i].R);
        for (int k = 0; k < 3; k++) P.poses[i].t[k] = poses[7 * i + 4 + k];
    }
    P.pts.assign(points, points + 3 * npts);
    P.ep = ep; P.el = el; P.obs = obs; P.ne = ne; P.fixed = fixed;
    P.K = {fx, fy, cx, cy}; P.delta = delta;
}

}  // namespace

extern "C" {

void orc_se3_exp(const double* xi6, double* q_t7) {
    double R[9], t[3];
    se3_exp(xi6, R, t);
    R_to_quat(R, q_t7);
    q_t7[4] = t[0]; q_t7[5] = t[1]; q_t7[6] = t[2];
}

int orc_ba_build(const double* poses, int nposes, const double* points, int npts,
                 const int32_t* edge_pose, const int32_t* edge_pt, const double* obs, int nedges,
                 const uint8_t* fixed_pt, double fx, double fy, double cx, double cy, double huber_delta,
                 double* Hpp, double* Hll, double* Hpl, double* bp, double* bl, double* chi2_raw) {
    for (int k = 0; k < nedges; k++)
        if (edge_pose[k] < 0 || edge_pose[k] >= nposes || edge_pt[k] < 0 || edge_pt[k] >= npts) return -1;
    Problem P;
    load_problem(P, poses, nposes, points, npts, edge_pose, edge_pt, obs, nedges, fixed_pt, fx, fy, cx, cy, huber_delta);
    build(P, Hpp, Hll, Hpl, bp, bl, chi2_raw);
    return 0;
}

// Iteration trace for the maintainer-side pin (tools/dump_reference_goldens.cpp records the same three numbers from g2o's post-iteration
// hook: activeRobustChi2() — the robust chi2 of the LAST Levenberg trial, accepted or not —, currentLambda(), levenbergIteration()).
// Rows of 4 doubles: {robust chi2 of the last trial, chi2 of the accepted state, lambda after the iteration, trials}.  Test infrastructure.
static double* g_trace = nullptr; static int g_trace_cap = 0, g_trace_n = 0;
extern "C" void orc_ba_set_trace(double* rows4, int cap_rows) { g_trace = rows4; g_trace_cap = rows4 ? cap_rows : 0; g_trace_n = 0; }
extern "C" int orc_ba_trace_rows() { return g_trace_n; }

// g2o OptimizationAlgorithmLevenberg::solve x max_iters (Appendix A.7)
static int lm_optimize(double* poses, int nposes, double* points, int npts,
                       const int32_t* edge_pose, const int32_t* edge_pt, const double* obs, int nedges,
                       const uint8_t* fixed_pt, double fx, double fy, double cx, double cy, double huber_delta,
                       int max_iters, double* final_chi2, int* iters, double* edge_chi2) {
    for (int k = 0; k < nedges; k++)
        if (edge_pose[k] < 0 || edge_pose[k] >= nposes || edge_pt[k] < 0 || edge_pt[k] >= npts) return -1;
    Problem P;
    load_problem(P, poses, nposes, points, npts, edge_pose, edge_pt, obs, nedges, fixed_pt, fx, fy, cx, cy, huber_delta);
    std::vector<double> Hpp((size_t)nposes * 36), Hll((size_t)npts * 9), Hpl((size_t)nedges * 18), bp((size_t)nposes * 6), bl((size_t)npts * 3);
    double lambda = 0, ni = 2;
    int it = 0;
    for (; it < max_iters; it++) {
        double currentChi = robust_chi2(P, edge_chi2), tempChi = currentChi;
        build(P, Hpp.data(), Hll.data(), Hpl.data(), bp.data(), bl.data(), nullptr);
        if (it == 0) {          // computeLambdaInit: tau * max diagonal
            double mx = 0;
            for (int p = 0; p < nposes; p++) for (int a = 0; a < 6; a++) mx = std::max(mx, fabs(Hpp[p * 36 + a * 7]));
            for (int l = 0; l < npts; l++) { if (fixed_pt && fixed_pt[l]) continue; for (int a = 0; a < 3; a++) mx = std::max(mx, fabs(Hll[l * 9 + a * 4])); }
            lambda = 1e-5 * mx; ni = 2;
        }
        double rho = 0;
        int qmax = 0;
        do {
            std::vector<Pose> savedPoses = P.poses;
            std::vector<double> savedPts = P.pts;
            std::vector<double> xp, xl;
            bool ok = schur_solve(P, Hpp.data(), Hll.data(), Hpl.data(), bp.data(), bl.data(), lambda, xp, xl);
            if (ok) {
                for (int p = 0; p < nposes; p++) pose_oplus(P.poses[p], &xp[6 * p]);
                for (int l = 0; l < npts; l++) for (int a = 0; a < 3; a++) P.pts[3 * l + a] += xl[3 * l + a];
                tempChi = robust_chi2(P, edge_chi2);      // the edges keep THIS error even if the step is rejected below
            } else tempChi = 1e300;
            rho = currentChi - tempChi;
            double scale = 1e-3;
            if (ok) {
                for (int i = 0; i < 6 * nposes; i++) scale += xp[i] * (lambda * xp[i] + bp[i]);
                for (int l = 0; l < npts; l++) { if (fixed_pt && fixed_pt[l]) continue; for (int a = 0; a < 3; a++) scale += xl[3 * l + a] * (lambda * xl[3 * l + a] + bl[3 * l + a]); }
            }
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi) && ok) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni; ni *= 2;
                P.poses = savedPoses; P.pts = savedPts;
                if (!std::isfinite(lambda)) break;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        if (g_trace && g_trace_n < g_trace_cap) { double* r = g_trace + 4 * g_trace_n++; r[0] = tempChi; r[1] = currentChi; r[2] = lambda; r[3] = qmax; }
        if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) { it++; break; }
    }
    for (int i = 0; i < nposes; i++) {
        R_to_quat(P.poses[i].R, poses + 7 * i);
        for (int k = 0; k < 3; k++) poses[7 * i + 4 + k] = P.poses[i].t[k];
    }
    memcpy(points, P.pts.data(), sizeof(double) * 3 * npts);
    if (final_chi2) *final_chi2 = robust_chi2(P);
    if (iters) *iters = it;
    return 0;
}

int orc_ba_optimize(double* poses, int nposes, double* points, int npts,
                    const int32_t* edge_pose, const int32_t* edge_pt, const double* obs, int nedges,
                    const uint8_t* fixed_pt, double fx, double fy, double cx, double cy, double huber_delta,
                    int max_iters, double* final_chi2, int* iters) {
    return lm_optimize(poses, nposes, points, npts, edge_pose, edge_pt, obs, nedges, fixed_pt, fx, fy, cx, cy, huber_delta, max_iters,
                       final_chi2, iters, nullptr);
}

/* Backend::OptimizeActiveMap, src/backend.cpp:208-243: up to max_rounds x { initializeOptimization(); optimize(iters) }, after
 * each round count edges with chi2() > chi2_th and stop as soon as the inlier ratio exceeds 0.5; then flag the outliers.
 * edge->chi2() is e^T e of the LAST error evaluation g2o made (App. A.7: the last Levenberg trial, accepted or not). */
int orc_ba_optimize_active_map(double* poses, int nposes, double* points, int npts,
                               const int32_t* edge_pose, const int32_t* edge_pt, const double* obs, int nedges,
                               const uint8_t* fixed_pt, double fx, double fy, double cx, double cy, double huber_delta,
                               double chi2_th, int max_rounds, int iters_per_round,
                               double* edge_chi2, uint8_t* outlier, int* rounds, int* n_outliers) {
    if (nedges < 1 || !edge_chi2 || !outlier) return -1;
    int r = 0, cntOut = 0;
    while (r < max_rounds) {
        int rc = lm_optimize(poses, nposes, points, npts, edge_pose, edge_pt, obs, nedges, fixed_pt, fx, fy, cx, cy, huber_delta,
                             iters_per_round, nullptr, nullptr, edge_chi2);
        if (rc) return rc;
        cntOut = 0;
        for (int k = 0; k < nedges; k++) cntOut += edge_chi2[k] > chi2_th;
        const double inlierRatio = (nedges - cntOut) / double(nedges);
        if (inlierRatio > 0.5) break;
        r++;
    }
    for (int k = 0; k < nedges; k++) outlier[k] = edge_chi2[k] > chi2_th;
    if (rounds) *rounds = r;              // the reference's `iteration` counter: rounds that FAILED the inlier test
    if (n_outliers) *n_outliers = cntOut;
    return 0;
}

/* Frontend::EstimateCurrentPose, src/frontend.cpp:176-276 (and LoopClosing::OptimizeCurrentPose, src/loopclosing.cpp:339-433, which
 * runs one extra optimize(10) first): one VertexPose, one EdgeProjectionPoseOnly (g2o_types.h:62-100) per
 * tracked feature that has a map point, Huber (default delta = 1), g2o Levenberg with a dense 6x6 solve; `rounds` (4) times
 * { initializeOptimization(); optimize(iters = 10) } over the level-0 edges, then every edge is classified by chi2() > chi2_th
 * (an edge that was excluded gets computeError() at the new estimate first, :231-233) and excluded / re-admitted for the next
 * round (:234-241); after round rounds-2 the robust kernel is removed (:244-246).  chi2() of an ACTIVE edge is e^T e of the
 * last error evaluation g2o made (the last Levenberg trial, accepted or not).  Third-party internals: PARITY UNPINNED. */
int orc_pose_only_optimize(double* pose7, const double* pts3d, const double* obs, int n, double fx, double fy, double cx, double cy,
                           double chi2_th, int rounds, int iters, int pre_optimize, uint8_t* outlier, int* n_inliers) {
    if (!pose7 || n < 0 || (n > 0 && (!pts3d || !obs || !outlier)) || rounds < 1 || iters < 1 || pre_optimize < 0) return -1;
    Pose T;
    quat_to_R(pose7, T.R);
    for (int k = 0; k < 3; k++) T.t[k] = pose7[4 + k];
    std::vector<uint8_t> level(n, 0);                 // 1 = excluded from the optimisation (e->setLevel(1))
    std::vector<double> echi(n, 0.0);
    for (int i = 0; i < n; i++) outlier[i] = 0;
    bool robust = true;
    // e = z - (K (T p)) / (K (T p)).z   (:71-75)
    auto edge_err = [&](const Pose& P, int i, double* e) {
        double pc[3];
        for (int r = 0; r < 3; r++) pc[r] = P.R[r * 3] * pts3d[3 * i] + P.R[r * 3 + 1] * pts3d[3 * i + 1] + P.R[r * 3 + 2] * pts3d[3 * i + 2] + P.t[r];
        const double u = fx * pc[0] + 0.0 * pc[1] + cx * pc[2], v = 0.0 * pc[0] + fy * pc[1] + cy * pc[2];
        e[0] = obs[2 * i] - u / pc[2]; e[1] = obs[2 * i + 1] - v / pc[2];
    };
    auto active_chi2 = [&](const Pose& P) {            // computeActiveErrors + activeRobustChi2
        double s = 0;
        for (int i = 0; i < n; i++) {
            if (level[i]) continue;
            double e[2]; edge_err(P, i, e);
            const double e2 = e[0] * e[0] + e[1] * e[1];
            echi[i] = e2;
            double r0 = e2, r1 = 1;
            if (robust) huber(e2, 1.0, r0, r1);
            s += r0;
        }
        return s;
    };
    int cntOut = 0;
    // pre_optimize: unclassified optimize(iters) calls before the rounds (LoopClosing::OptimizeCurrentPose has one, loopclosing.cpp:395-396)
    for (int round = -pre_optimize; round < rounds; round++) {
        int nact = 0;
        for (int i = 0; i < n; i++) nact += !level[i];
        if (nact > 0) {
            double lambda = 0, ni = 2;
            for (int it = 0; it < iters; it++) {
                double currentChi = active_chi2(T), tempChi = currentChi;
                double H[36] = {0}, b[6] = {0};
                for (int i = 0; i < n; i++) {
                    if (level[i]) continue;
                    double pc[3];
                    for (int r = 0; r < 3; r++) pc[r] = T.R[r * 3] * pts3d[3 * i] + T.R[r * 3 + 1] * pts3d[3 * i + 1] + T.R[r * 3 + 2] * pts3d[3 * i + 2] + T.t[r];
                    const double X = pc[0], Y = pc[1], Z = pc[2], Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;     // :79-92
                    const double J[12] = {-fx * Zinv, 0, fx * X * Zinv2, fx * X * Y * Zinv2, -fx - fx * X * X * Zinv2, fx * Y * Zinv,
                                          0, -fy * Zinv, fy * Y * Zinv2, fy + fy * Y * Y * Zinv2, -fy * X * Y * Zinv2, -fy * X * Zinv};
                    double e[2]; edge_err(T, i, e);
                    double r0, w = 1;
                    if (robust) huber(e[0] * e[0] + e[1] * e[1], 1.0, r0, w);
                    for (int r = 0; r < 6; r++) {
                        for (int c = 0; c < 6; c++) H[r * 6 + c] += w * (J[r] * J[c] + J[6 + r] * J[6 + c]);
                        b[r] += -w * (J[r] * e[0] + J[6 + r] * e[1]);
                    }
                }
                if (it == 0) {
                    double mx = 0;
                    for (int a = 0; a < 6; a++) mx = std::max(mx, fabs(H[a * 7]));
                    lambda = 1e-5 * mx; ni = 2;
                }
                double rho = 0; int qmax = 0;
                do {
                    const Pose saved = T;
                    double A[36], x[6];
                    memcpy(A, H, sizeof(A));
                    for (int a = 0; a < 6; a++) A[a * 7] += lambda;
                    bool ok = true;                     // dense Cholesky solve (LinearSolverDense)
                    for (int j = 0; j < 6 && ok; j++) {
                        double d = A[j * 6 + j];
                        for (int k2 = 0; k2 < j; k2++) d -= A[j * 6 + k2] * A[j * 6 + k2];
                        if (!(d > 0)) { ok = false; break; }
                        A[j * 6 + j] = sqrt(d);
                        for (int i2 = j + 1; i2 < 6; i2++) {
                            double v = A[i2 * 6 + j];
                            for (int k2 = 0; k2 < j; k2++) v -= A[i2 * 6 + k2] * A[j * 6 + k2];
                            A[i2 * 6 + j] = v / A[j * 6 + j];
                        }
                    }
                    if (ok) {
                        for (int i2 = 0; i2 < 6; i2++) { double v = b[i2]; for (int k2 = 0; k2 < i2; k2++) v -= A[i2 * 6 + k2] * x[k2]; x[i2] = v / A[i2 * 7]; }
                        for (int i2 = 5; i2 >= 0; i2--) { double v = x[i2]; for (int k2 = i2 + 1; k2 < 6; k2++) v -= A[k2 * 6 + i2] * x[k2]; x[i2] = v / A[i2 * 7]; }
                        pose_oplus(T, x);
                        tempChi = active_chi2(T);
                    } else tempChi = 1e300;
                    rho = currentChi - tempChi;
                    double scale = 1e-3;
                    if (ok) for (int a = 0; a < 6; a++) scale += x[a] * (lambda * x[a] + b[a]);
                    rho /= scale;
                    if (rho > 0 && std::isfinite(tempChi) && ok) {
                        double alpha = 1. - pow(2 * rho - 1, 3);
                        alpha = std::min(alpha, 2. / 3.);
                        lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
                    } else {
                        lambda *= ni; ni *= 2; T = saved;
                        if (!std::isfinite(lambda)) break;
                    }
                    qmax++;
                } while (rho < 0 && qmax < 10);
                if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) break;
            }
        }
        if (round < 0) continue;
        cntOut = 0;
        for (int i = 0; i < n; i++) {
            if (outlier[i]) { double e[2]; edge_err(T, i, e); echi[i] = e[0] * e[0] + e[1] * e[1]; }      // :231-233
            if (echi[i] > chi2_th) { outlier[i] = 1; level[i] = 1; cntOut++; }
            else { outlier[i] = 0; level[i] = 0; }
        }
        if (round == rounds - 2) robust = false;                                                           // :244-246
    }
    R_to_quat(T.R, pose7);
    for (int k2 = 0; k2 < 3; k2++) pose7[4 + k2] = T.t[k2];
    if (n_inliers) *n_inliers = n - cntOut;
    return 0;
}

}  // extern "C"
