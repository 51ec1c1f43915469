/*
 * pgo_oracle.cpp — CPU ORACLE (test infrastructure only).
 *
 * Loop correction: LoopClosing::PoseGraphOptimization (src/loopclosing.cpp:537-646).
 *   vertices   one VertexPose per key-frame (estimate = Tcw, left-multiplied update, g2o_types.h:32-37); the active
 *              key-frames, the loop key-frame and key-frame 0 are fixed (:557-562) — here: the caller's `fixed` flags
 *   edges      EdgePoseGraph (g2o_types.h:157-190): error = log(M^-1 * v0 * v1^-1), information = I6, no robust kernel;
 *              one per (KF, previous KF) with M = mRelativePoseToLastKF (:577-588) and one per (KF, loop KF) with
 *              M = mRelativePoseToLoopKF (:590-601)
 *   Jacobians  linearizeOplus is commented out in the reference (g2o_types.h:168-182), so g2o's numeric central
 *              difference runs: delta = 1e-9 per tangent coordinate of each non-fixed vertex, column = (e+ - e-) / (2 delta)
 *   solver     g2o Levenberg over BlockSolver<6,6> + sparse Cholesky (:538-543), optimize(20) (:606)
 *   write-back map points move rigidly with the key-frame that first observed them (:621-633)
 *
 * Sophus (SE3 = unit quaternion + translation; exp / log with the small-angle branches at 1e-10) and g2o internals are
 * third-party and absent from /root/reference: restated from their published algorithms, PARITY UNPINNED.
 * The linear solve here is an envelope (sky-line) Cholesky over the free vertices in index order — any exact solver gives
 * the same Levenberg trajectory up to rounding; the numeric Jacobian itself carries ~1e-7 relative noise.
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct SE3q { double q[4]; double t[3]; };      // q = (x, y, z, w)

const double kEps = 1e-10;                      // Sophus::Constants<double>::epsilon()

inline void rot(const double* q, const double* v, double* out) {       // Eigen: v + w*(2 qv x v) + qv x (2 qv x v)
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

inline void qnormalize(double* q) {
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

inline SE3q mul(const SE3q& a, const SE3q& b) {
    SE3q r;
    r.q[3] = a.q[3] * b.q[3] - a.q[0] * b.q[0] - a.q[1] * b.q[1] - a.q[2] * b.q[2];
    r.q[0] = a.q[3] * b.q[0] + a.q[0] * b.q[3] + a.q[1] * b.q[2] - a.q[2] * b.q[1];
    r.q[1] = a.q[3] * b.q[1] - a.q[0] * b.q[2] + a.q[1] * b.q[3] + a.q[2] * b.q[0];
    r.q[2] = a.q[3] * b.q[2] + a.q[0] * b.q[1] - a.q[1] * b.q[0] + a.q[2] * b.q[3];
    qnormalize(r.q);
    double rt[3];
    rot(a.q, b.t, rt);
    for (int k = 0; k < 3; k++) r.t[k] = a.t[k] + rt[k];
    return r;
}

inline SE3q inv(const SE3q& a) {
    SE3q r;
    r.q[0] = -a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = a.q[3];
    double rt[3];
    rot(r.q, a.t, rt);
    for (int k = 0; k < 3; k++) r.t[k] = -rt[k];
    return r;
}

// Sophus SE3::exp, tangent = (upsilon, omega)
inline SE3q se3q_exp(const double* d) {
    SE3q r;
    const double wx = d[3], wy = d[4], wz = d[5];
    const double th2 = wx * wx + wy * wy + wz * wz;
    double imag, real, th = 0;
    if (th2 < kEps * kEps) {
        const double th4 = th2 * th2;
        imag = 0.5 - th2 / 48.0 + th4 / 3840.0;
        real = 1.0 - th2 / 8.0 + th4 / 384.0;
    } else {
        th = sqrt(th2);
        const double h = 0.5 * th;
        imag = sin(h) / th;
        real = cos(h);
    }
    r.q[0] = imag * wx; r.q[1] = imag * wy; r.q[2] = imag * wz; r.q[3] = real;
    qnormalize(r.q);
    // t = V * upsilon, V = I + B W + C W^2
    double B, C;
    if (th2 < kEps * kEps) { B = 0.5; C = 1.0 / 6.0; }
    else { B = (1.0 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double u[3] = {d[0], d[1], d[2]};
    const double wu[3] = {wy * u[2] - wz * u[1], wz * u[0] - wx * u[2], wx * u[1] - wy * u[0]};
    const double wwu[3] = {wy * wu[2] - wz * wu[1], wz * wu[0] - wx * wu[2], wx * wu[1] - wy * wu[0]};
    for (int k = 0; k < 3; k++) r.t[k] = u[k] + B * wu[k] + C * wwu[k];
    return r;
}

// Sophus SE3::log -> (upsilon, omega)
inline void se3q_log(const SE3q& T, double* d) {
    const double n2 = T.q[0] * T.q[0] + T.q[1] * T.q[1] + T.q[2] * T.q[2], w = T.q[3];
    double f;                                                   // 2 atan(n / w) / n
    if (n2 < kEps * kEps) f = 2.0 / w - 2.0 / 3.0 * n2 / (w * w * w);
    else {
        const double n = sqrt(n2);
        if (fabs(w) < kEps) f = (w > 0 ? M_PI : -M_PI) / n;
        else f = 2.0 * atan(n / w) / n;
    }
    const double wx = f * T.q[0], wy = f * T.q[1], wz = f * T.q[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double C;                                                   // V^-1 = I - W/2 + C W^2
    if (th < kEps) C = 1.0 / 12.0;
    else { const double h = 0.5 * th; C = (1.0 - th * cos(h) / (2.0 * sin(h))) / th2; }
    const double* t = T.t;
    const double wt[3] = {wy * t[2] - wz * t[1], wz * t[0] - wx * t[2], wx * t[1] - wy * t[0]};
    const double wwt[3] = {wy * wt[2] - wz * wt[1], wz * wt[0] - wx * wt[2], wx * wt[1] - wy * wt[0]};
    for (int k = 0; k < 3; k++) d[k] = t[k] - 0.5 * wt[k] + C * wwt[k];
    d[3] = wx; d[4] = wy; d[5] = wz;
}

inline SE3q load7(const double* p) {
    SE3q T;
    for (int k = 0; k < 4; k++) T.q[k] = p[k];
    qnormalize(T.q);
    for (int k = 0; k < 3; k++) T.t[k] = p[4 + k];
    return T;
}
inline void store7(const SE3q& T, double* p) {
    for (int k = 0; k < 4; k++) p[k] = T.q[k];
    for (int k = 0; k < 3; k++) p[4 + k] = T.t[k];
}

// g2o_types.h:161-167
inline void edge_error(const SE3q& Minv, const SE3q& v0, const SE3q& v1, double* e) {
    se3q_log(mul(mul(Minv, v0), inv(v1)), e);
}

// lower-triangular envelope storage: row i holds columns first[i]..i
struct Skyline {
    int n = 0;
    std::vector<int> first;
    std::vector<size_t> off;
    std::vector<double> a;
    double& at(int i, int j) { return a[off[i] + (size_t)(j - first[i])]; }
    void init(const std::vector<int>& f) {
        n = (int)f.size(); first = f; off.resize(n + 1); off[0] = 0;
        for (int i = 0; i < n; i++) off[i + 1] = off[i] + (size_t)(i - first[i] + 1);
        a.assign(off[n], 0.0);
    }
    bool cholesky() {                                            // in place, A = L L^T
        for (int i = 0; i < n; i++) {
            for (int j = first[i]; j <= i; j++) {
                double s = at(i, j);
                const int k0 = std::max(first[i], first[j]);
                const double* ri = &a[off[i] + (size_t)(k0 - first[i])];
                const double* rj = &a[off[j] + (size_t)(k0 - first[j])];
                for (int k = 0; k < j - k0; k++) s -= ri[k] * rj[k];
                if (j < i) at(i, j) = s / at(j, j);
                else { if (!(s > 0)) return false; at(i, i) = sqrt(s); }
            }
        }
        return true;
    }
    void solve(std::vector<double>& x) {                         // x <- A^-1 x
        for (int i = 0; i < n; i++) {
            double s = x[i];
            for (int k = first[i]; k < i; k++) s -= at(i, k) * x[k];
            x[i] = s / at(i, i);
        }
        for (int i = n - 1; i >= 0; i--) {
            x[i] /= at(i, i);
            for (int k = first[i]; k < i; k++) x[k] -= at(i, k) * x[i];
        }
    }
};

}  // namespace

extern "C" {

int orc_se3_log(const double* q_t7, double* xi6) {
    se3q_log(load7(q_t7), xi6);
    return 0;
}

int orc_se3_compose(const double* a7, const double* b7, int invert_b, double* out7) {
    SE3q b = load7(b7);
    if (invert_b) b = inv(b);
    store7(mul(load7(a7), b), out7);
    return 0;
}

// LoopClosing::LoopLocalFusion, src/loopclosing.cpp:466-507 (arithmetic part): the active key-frames move rigidly with the corrected
// current key-frame (Ta' = Ta * Tc^-1 * Tc'), every active map point keeps its camera-frame position in the active key-frame that
// first observes it (p' = Ta'^-1 * (Ta * p)).
int orc_loop_local_fusion(double* active_poses, int n_active, int cur, const double* corrected_cur, const int32_t* first_active_kf,
                          double* points, int n_points) {
    if (n_active < 1 || cur < 0 || cur >= n_active || !active_poses || !corrected_cur || n_points < 0) return -1;
    std::vector<SE3q> oldp(n_active), newp(n_active);
    for (int a = 0; a < n_active; a++) oldp[a] = load7(active_poses + 7 * a);
    const SE3q Tc_inv = inv(oldp[cur]), Tcc = load7(corrected_cur);
    for (int a = 0; a < n_active; a++) newp[a] = (a == cur) ? Tcc : mul(mul(oldp[a], Tc_inv), Tcc);          // :480-482
    for (int i = 0; i < n_points; i++) {
        const int a = first_active_kf[i];
        if (a < 0) continue;
        if (a >= n_active) return -1;
        double pc[3], pw[3];
        rot(oldp[a].q, points + 3 * i, pc);
        for (int k = 0; k < 3; k++) pc[k] += oldp[a].t[k];                                                     // :499
        const SE3q Ti = inv(newp[a]);
        rot(Ti.q, pc, pw);
        for (int k = 0; k < 3; k++) points[3 * i + k] = pw[k] + Ti.t[k];                                       // :501
    }
    for (int a = 0; a < n_active; a++) store7(newp[a], active_poses + 7 * a);                                  // :505-507
    return 0;
}

int orc_pose_graph_optimize(double* poses, int n, const uint8_t* fixed, const int32_t* e0, const int32_t* e1,
                            const double* meas, int E, int max_iters, double* final_chi2, int* iters) {
    if (n < 0 || E < 0 || max_iters < 0 || (n > 0 && !poses) || (E > 0 && (!e0 || !e1 || !meas))) return -1;
    for (int k = 0; k < E; k++)
        if (e0[k] < 0 || e0[k] >= n || e1[k] < 0 || e1[k] >= n || e0[k] == e1[k]) return -1;
    std::vector<SE3q> V(n), Minv(E);
    for (int i = 0; i < n; i++) V[i] = load7(poses + 7 * i);
    for (int k = 0; k < E; k++) Minv[k] = inv(load7(meas + 7 * k));
    std::vector<int> fidx(n, -1);
    int nf = 0;
    for (int i = 0; i < n; i++) if (!(fixed && fixed[i])) fidx[i] = nf++;

    auto chi2_all = [&]() {
        double s = 0;
        for (int k = 0; k < E; k++) {
            double e[6]; edge_error(Minv[k], V[e0[k]], V[e1[k]], e);
            double c = 0; for (int a = 0; a < 6; a++) c += e[a] * e[a];
            s += c;
        }
        return s;
    };
    int it = 0;
    if (nf > 0 && E > 0) {
        // envelope of the free system in vertex order
        std::vector<int> first(6 * nf);
        for (int i = 0; i < nf; i++) for (int a = 0; a < 6; a++) first[6 * i + a] = 6 * i;
        for (int k = 0; k < E; k++) {
            const int a = fidx[e0[k]], b = fidx[e1[k]];
            if (a < 0 || b < 0) continue;
            const int hi = std::max(a, b), lo = std::min(a, b);
            for (int r = 0; r < 6; r++) first[6 * hi + r] = std::min(first[6 * hi + r], 6 * lo);
        }
        Skyline H, A;
        H.init(first);
        std::vector<double> b(6 * nf), x(6 * nf);
        double lambda = 0, ni = 2;
        for (; it < max_iters; it++) {
            double currentChi = chi2_all(), tempChi = currentChi;
            std::fill(H.a.begin(), H.a.end(), 0.0);
            std::fill(b.begin(), b.end(), 0.0);
            for (int k = 0; k < E; k++) {
                const int vi[2] = {e0[k], e1[k]};
                double e[6]; edge_error(Minv[k], V[vi[0]], V[vi[1]], e);
                double J[2][36];                                 // 6x6 row-major per side
                for (int s = 0; s < 2; s++) {
                    if (fidx[vi[s]] < 0) continue;
                    for (int d = 0; d < 6; d++) {
                        double add[6] = {0, 0, 0, 0, 0, 0}, ep[6], em[6];
                        SE3q v[2] = {V[vi[0]], V[vi[1]]};
                        add[d] = 1e-9;  v[s] = mul(se3q_exp(add), V[vi[s]]); edge_error(Minv[k], v[0], v[1], ep);
                        add[d] = -1e-9; v[s] = mul(se3q_exp(add), V[vi[s]]); edge_error(Minv[k], v[0], v[1], em);
                        const double scalar = 1.0 / (2 * 1e-9);
                        for (int r = 0; r < 6; r++) J[s][r * 6 + d] = scalar * (ep[r] - em[r]);
                    }
                }
                for (int s = 0; s < 2; s++) {
                    const int fa = fidx[vi[s]];
                    if (fa < 0) continue;
                    for (int r = 0; r < 6; r++) {
                        double g = 0;
                        for (int m = 0; m < 6; m++) g += J[s][m * 6 + r] * e[m];
                        b[6 * fa + r] -= g;
                        for (int c = 0; c <= r; c++) {
                            double h = 0;
                            for (int m = 0; m < 6; m++) h += J[s][m * 6 + r] * J[s][m * 6 + c];
                            H.at(6 * fa + r, 6 * fa + c) += h;
                        }
                    }
                }
                const int fa = fidx[vi[0]], fb = fidx[vi[1]];
                if (fa >= 0 && fb >= 0) {
                    const int sh = fa > fb ? 0 : 1, sl = 1 - sh, hi = std::max(fa, fb), lo = std::min(fa, fb);
                    for (int r = 0; r < 6; r++)
                        for (int c = 0; c < 6; c++) {
                            double h = 0;
                            for (int m = 0; m < 6; m++) h += J[sh][m * 6 + r] * J[sl][m * 6 + c];
                            H.at(6 * hi + r, 6 * lo + c) += h;
                        }
                }
            }
            if (it == 0) {
                double mx = 0;
                for (int i = 0; i < 6 * nf; i++) mx = std::max(mx, fabs(H.at(i, i)));
                lambda = 1e-5 * mx; ni = 2;
            }
            double rho = 0; int qmax = 0;
            do {
                std::vector<SE3q> saved = V;
                A = H;
                for (int i = 0; i < 6 * nf; i++) A.at(i, i) += lambda;
                const bool ok = A.cholesky();
                if (ok) {
                    x = b; A.solve(x);
                    for (int i = 0; i < n; i++) if (fidx[i] >= 0) V[i] = mul(se3q_exp(&x[6 * fidx[i]]), V[i]);
                    tempChi = chi2_all();
                } else tempChi = 1e300;
                rho = currentChi - tempChi;
                double scale = 1e-3;
                if (ok) for (int i = 0; i < 6 * nf; i++) scale += x[i] * (lambda * x[i] + b[i]);
                rho /= scale;
                if (rho > 0 && std::isfinite(tempChi) && ok) {
                    double alpha = 1. - pow(2 * rho - 1, 3);
                    alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha); ni = 2; currentChi = tempChi;
                } else {
                    lambda *= ni; ni *= 2; V = saved;
                    if (!std::isfinite(lambda)) break;
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0 || !std::isfinite(lambda)) { it++; break; }
        }
    }
    for (int i = 0; i < n; i++) store7(V[i], poses + 7 * i);
    if (final_chi2) *final_chi2 = chi2_all();
    if (iters) *iters = it;
    return 0;
}

/* src/loopclosing.cpp:621-633: p <- T_new^-1 * (T_old * p) with T = the pose of the key-frame that first observed the point;
 * kf[i] < 0 leaves the point alone (the :625-629 skip). */
int orc_correct_map_points(const double* old_poses, const double* new_poses, int nposes, const int32_t* kf, double* pts, int npts) {
    for (int i = 0; i < npts; i++) {
        if (kf[i] < 0) continue;
        if (kf[i] >= nposes) return -1;
        const SE3q To = load7(old_poses + 7 * kf[i]), Tn = inv(load7(new_poses + 7 * kf[i]));
        double pc[3], pw[3];
        rot(To.q, pts + 3 * i, pc);
        for (int k = 0; k < 3; k++) pc[k] += To.t[k];
        rot(Tn.q, pc, pw);
        for (int k = 0; k < 3; k++) pts[3 * i + k] = pw[k] + Tn.t[k];
    }
    return 0;
}

}  // extern "C"
