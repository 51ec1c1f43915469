// pnp.hip — loop verification on gfx950: cv::solvePnPRansac(points3d, points2d, K, noArray(), rvec, tvec, false, 100, 5.991, 0.99)
// as LoopClosing::ComputeCorrectPose calls it (src/loopclosing.cpp:262-268)   [SURVEY.md §8(f) rank 3]
//
// OpenCV 3.4 runs the hypotheses one after another (sample 5 matches with cv::RNG, EPnP, count inliers, shrink the iteration budget
// when a better model appears).  The sample sets do not depend on the models — one generator, drawn in order — so here
//   host      draws all `iterations` 5-subsets exactly as RANSACPointSetRegistrator::getSubset would (cv::RNG((uint64)-1))
//   k_pnp_hypotheses   one wave per hypothesis, all in flight at once: EPnP (Lepetit et al. 2009 as in calib3d/epnp.cpp: PCA
//             control points, barycentric coordinates, M^T M, its 4 smallest eigenvectors by a wave-cooperative cyclic Jacobi in
//             LDS, three beta approximations + 5 Gauss-Newton steps each, Arun's absolute orientation, least mean reprojection
//             error wins), then the wave scores every match (squared reprojection error in float against (float)(thr^2))
//   k_pnp_select_refine   replays OpenCV's sequential bookkeeping over the per-hypothesis counts (a model wins when its count
//             exceeds max(best, 4); RANSACUpdateNumIters shrinks the budget; hypotheses past the budget are ignored), rebuilds the
//             winner's inlier mask and refines the pose on the inliers by Levenberg-Marquardt (left-multiplied SE3 update, run to
//             convergence; OpenCV re-solves with SOLVEPNP_ITERATIVE = DLT start + its own LM on the same cost).
// Built with -ffp-contract=off and the oracle's operation order: the RANSAC stage has no transcendental function in it, so counts,
// winner and mask are reproducible bit for bit; only the refinement (sin / cos in exp) differs in the last bits.
#include <float.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace myslam_hip {

constexpr int PNP_MP = 5;            // model points of the RANSAC kernel (EPnP)

#define PNP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                             __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Cyclic Jacobi on a symmetric n x n matrix in LDS, the whole wave cooperating: lane k owns element k of the rotated rows /
// columns; every lane evaluates the (uniform) rotation parameters.  A is destroyed, V = eigenvectors in columns.
__device__ void pnp_wave_jacobi(int n, double* A, double* V, int lane) {
    for (int i = lane; i < n * n; i += 64) V[i] = ((i / n) == (i % n)) ? 1.0 : 0.0;
    PNP_WAVE_SYNC();
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j]; }
        if (off <= 1e-30 * diag || off == 0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0) continue;                    // uniform
                const double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                PNP_WAVE_SYNC();
                if (lane < n) {
                    const int k = lane;
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
                }
                PNP_WAVE_SYNC();
                if (lane < n) {
                    const int k = lane;
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
                }
                PNP_WAVE_SYNC();
            }
    }
}

// single-lane version for the 3 x 3 problems, eigenvalues sorted descending (w), eigenvector k = column k of V
__device__ void pnp_jacobi3(double* A, double* V, double* w) {
    const int n = 3;
    for (int i = 0; i < 9; i++) V[i] = ((i / 3) == (i % 3)) ? 1.0 : 0.0;
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
                for (int k = 0; k < n; k++) { const double akp = A[k * n + p], akq = A[k * n + q]; A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq; }
                for (int k = 0; k < n; k++) { const double apk = A[p * n + k], aqk = A[q * n + k]; A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk; }
                for (int k = 0; k < n; k++) { const double vkp = V[k * n + p], vkq = V[k * n + q]; V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq; }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    for (int i = 0; i < n - 1; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++) if (w[j] > w[m]) m = j;
        if (m != i) {
            const double tw = w[i]; w[i] = w[m]; w[m] = tw;
            for (int k = 0; k < n; k++) { const double tv = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = tv; }
        }
    }
}

// min |A x - b| by Householder QR (m = 6 rows, n <= 5 columns), everything in LDS scratch, single lane
__device__ void pnp_ls_solve(int m, int n, double* A, double* b, double* x, double* v) {
    for (int k = 0; k < n; k++) {
        double nrm = 0;
        for (int i = k; i < m; i++) nrm += A[i * n + k] * A[i * n + k];
        nrm = sqrt(nrm);
        if (nrm == 0) continue;
        const double alpha = A[k * n + k] > 0 ? -nrm : nrm;
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

struct PnpCam { double fu, fv, uc, vc; };

struct PnpLds {                       // one wave's EPnP workspace
    double A[144], V[144];
    double pw[3 * PNP_MP], uv[2 * PNP_MP], alphas[4 * PNP_MP], pcs[3 * PNP_MP];
    double cws[12], v4[48], L[60], rho[6];
    double lsA[30], lsb[6], lsx[5], lsv[12];
    double m3[9], V3[9], w3[3], U[9];
    double R[9], t[3], bestR[9], bestt[3];
    int ok;
};

// EPnP phase A (lane 0): control points, barycentric coordinates, M^T M -> S.A
__device__ bool pnp_epnp_setup(PnpLds& S, const PnpCam& K) {
    const int n = PNP_MP;
    double* cws = S.cws;
    for (int j = 0; j < 12; j++) cws[j] = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) cws[j] += S.pw[3 * i + j];
    for (int j = 0; j < 3; j++) cws[j] /= n;
    double* C = S.m3;
    for (int a = 0; a < 9; a++) C[a] = 0;
    for (int i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) C[a * 3 + b] += (S.pw[3 * i + a] - cws[a]) * (S.pw[3 * i + b] - cws[b]);
    pnp_jacobi3(C, S.V3, S.w3);
    for (int i = 1; i < 4; i++) {
        const double k = sqrt(fmax(S.w3[i - 1], 0.0) / n);
        for (int j = 0; j < 3; j++) cws[3 * i + j] = cws[j] + k * S.V3[j * 3 + (i - 1)];
    }
    double cc[9];
    for (int j = 0; j < 3; j++) for (int i = 1; i < 4; i++) cc[3 * j + i - 1] = cws[3 * i + j] - cws[j];
    const double det = cc[0] * (cc[4] * cc[8] - cc[5] * cc[7]) - cc[1] * (cc[3] * cc[8] - cc[5] * cc[6]) + cc[2] * (cc[3] * cc[7] - cc[4] * cc[6]);
    if (!(fabs(det) > 1e-300)) return false;
    const double ci[9] = {(cc[4] * cc[8] - cc[5] * cc[7]) / det, (cc[2] * cc[7] - cc[1] * cc[8]) / det, (cc[1] * cc[5] - cc[2] * cc[4]) / det,
                          (cc[5] * cc[6] - cc[3] * cc[8]) / det, (cc[0] * cc[8] - cc[2] * cc[6]) / det, (cc[2] * cc[3] - cc[0] * cc[5]) / det,
                          (cc[3] * cc[7] - cc[4] * cc[6]) / det, (cc[1] * cc[6] - cc[0] * cc[7]) / det, (cc[0] * cc[4] - cc[1] * cc[3]) / det};
    for (int i = 0; i < n; i++) {
        double* a = S.alphas + 4 * i;
        for (int j = 0; j < 3; j++)
            a[1 + j] = ci[3 * j] * (S.pw[3 * i] - cws[0]) + ci[3 * j + 1] * (S.pw[3 * i + 1] - cws[1]) + ci[3 * j + 2] * (S.pw[3 * i + 2] - cws[2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
    for (int i = 0; i < 144; i++) S.A[i] = 0;
    for (int i = 0; i < n; i++) {
        double m1[12], m2[12];
        const double* a = S.alphas + 4 * i;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            m1[3 * j] = a[j] * K.fu; m1[3 * j + 1] = 0.0; m1[3 * j + 2] = a[j] * (K.uc - S.uv[2 * i]);
            m2[3 * j] = 0.0; m2[3 * j + 1] = a[j] * K.fv; m2[3 * j + 2] = a[j] * (K.vc - S.uv[2 * i + 1]);
        }
#pragma unroll
        for (int r = 0; r < 12; r++)
#pragma unroll
            for (int c = 0; c < 12; c++) S.A[r * 12 + c] += m1[r] * m1[c] + m2[r] * m2[c];
    }
    return true;
}

__device__ double pnp_reproj_mean(const PnpCam& K, const double* R, const double* t, const double* pw, const double* uv, int n) {
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

// EPnP phase B (lane 0): eigenvectors in S.V / eigenvalues on the diagonal of S.A -> best (R, t)
__device__ bool pnp_epnp_finish(PnpLds& S, const PnpCam& K) {
    const int n = PNP_MP;
    // the 4 smallest eigenvalues, smallest first; ties keep the order a stable descending sort would give (later column first)
    int order[12];
    for (int i = 0; i < 12; i++) order[i] = i;
    {   // the oracle's selection sort, descending, on (w, column permutation)
        double w[12];
        for (int i = 0; i < 12; i++) w[i] = S.A[i * 12 + i];
        for (int i = 0; i < 11; i++) {
            int m = i;
            for (int j = i + 1; j < 12; j++) if (w[j] > w[m]) m = j;
            if (m != i) { const double tw = w[i]; w[i] = w[m]; w[m] = tw; const int to = order[i]; order[i] = order[m]; order[m] = to; }
        }
    }
    double* v = S.v4;
    for (int k = 0; k < 4; k++) for (int r = 0; r < 12; r++) v[k * 12 + r] = S.V[r * 12 + order[11 - k]];
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pb[6] = {1, 2, 3, 2, 3, 3};
    double* L = S.L;
    double* rho = S.rho;
    const double* cws = S.cws;
    for (int j = 0; j < 6; j++) {
        double dv[4][3];
        for (int k = 0; k < 4; k++) for (int c = 0; c < 3; c++) dv[k][c] = v[k * 12 + 3 * pa[j] + c] - v[k * 12 + 3 * pb[j] + c];
#define PNP_DOT(a, b) (dv[a][0] * dv[b][0] + dv[a][1] * dv[b][1] + dv[a][2] * dv[b][2])
        double* r = L + 10 * j;
        r[0] = PNP_DOT(0, 0); r[1] = 2 * PNP_DOT(0, 1); r[2] = PNP_DOT(1, 1); r[3] = 2 * PNP_DOT(0, 2); r[4] = 2 * PNP_DOT(1, 2);
        r[5] = PNP_DOT(2, 2); r[6] = 2 * PNP_DOT(0, 3); r[7] = 2 * PNP_DOT(1, 3); r[8] = 2 * PNP_DOT(2, 3); r[9] = PNP_DOT(3, 3);
#undef PNP_DOT
        rho[j] = 0;
        for (int c = 0; c < 3; c++) rho[j] += (cws[3 * pa[j] + c] - cws[3 * pb[j] + c]) * (cws[3 * pa[j] + c] - cws[3 * pb[j] + c]);
    }
    double best = 1e300;
    bool any = false;
    for (int variant = 1; variant <= 3; variant++) {
        double betas[4] = {0, 0, 0, 0};
        double* A = S.lsA; double* b = S.lsb; double* x = S.lsx;
        if (variant == 1) {
            const int cols[4] = {0, 1, 3, 6};
            for (int j = 0; j < 6; j++) { for (int c = 0; c < 4; c++) A[j * 4 + c] = L[10 * j + cols[c]]; b[j] = rho[j]; }
            pnp_ls_solve(6, 4, A, b, x, S.lsv);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = -x[1] / betas[0]; betas[2] = -x[2] / betas[0]; betas[3] = -x[3] / betas[0]; }
            else { betas[0] = sqrt(x[0]); betas[1] = x[1] / betas[0]; betas[2] = x[2] / betas[0]; betas[3] = x[3] / betas[0]; }
        } else if (variant == 2) {
            for (int j = 0; j < 6; j++) { for (int c = 0; c < 3; c++) A[j * 3 + c] = L[10 * j + c]; b[j] = rho[j]; }
            pnp_ls_solve(6, 3, A, b, x, S.lsv);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { betas[0] = sqrt(x[0]); betas[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) betas[0] = -betas[0];
        } else {
            for (int j = 0; j < 6; j++) { for (int c = 0; c < 5; c++) A[j * 5 + c] = L[10 * j + c]; b[j] = rho[j]; }
            pnp_ls_solve(6, 5, A, b, x, S.lsv);
            if (x[0] < 0) { betas[0] = sqrt(-x[0]); betas[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { betas[0] = sqrt(x[0]); betas[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) betas[0] = -betas[0];
            betas[2] = x[3] / betas[0];
        }
        for (int it = 0; it < 5; it++) {
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
            pnp_ls_solve(6, 4, A, b, x, S.lsv);
            for (int k = 0; k < 4; k++) betas[k] += x[k];
        }
        double ccs[12];
        for (int i = 0; i < 4; i++) for (int c = 0; c < 3; c++)
            ccs[3 * i + c] = betas[0] * v[3 * i + c] + betas[1] * v[12 + 3 * i + c] + betas[2] * v[24 + 3 * i + c] + betas[3] * v[36 + 3 * i + c];
        double* pcs = S.pcs;
        for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) {
            const double* a = S.alphas + 4 * i;
            pcs[3 * i + c] = a[0] * ccs[c] + a[1] * ccs[3 + c] + a[2] * ccs[6 + c] + a[3] * ccs[9 + c];
        }
        if (pcs[2] < 0) for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
        for (int i = 0; i < n; i++) for (int c = 0; c < 3; c++) { pc0[c] += pcs[3 * i + c]; pw0[c] += S.pw[3 * i + c]; }
        for (int c = 0; c < 3; c++) { pc0[c] /= n; pw0[c] /= n; }
        double ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; i++) for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) ABt[a * 3 + c] += (pcs[3 * i + a] - pc0[a]) * (S.pw[3 * i + c] - pw0[c]);
        double* AtA = S.m3; double* Vr = S.V3; double* s2 = S.w3; double* U = S.U;
        for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) AtA[a * 3 + c] = ABt[0 * 3 + a] * ABt[0 * 3 + c] + ABt[1 * 3 + a] * ABt[1 * 3 + c] + ABt[2 * 3 + a] * ABt[2 * 3 + c];
        pnp_jacobi3(AtA, Vr, s2);
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
            const double cx = U[3] * U[7] - U[6] * U[4], cy = U[6] * U[1] - U[0] * U[7], cz = U[0] * U[4] - U[3] * U[1];
            if (nn > 1e-12 * sqrt(fmax(s2[0], 0.0))) for (int a = 0; a < 3; a++) U[a * 3 + 2] = u[a] / nn;
            else { U[2] = cx; U[5] = cy; U[8] = cz; }
        }
        double* R = S.R; double* t = S.t;
        for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) R[a * 3 + c] = U[a * 3] * Vr[c * 3] + U[a * 3 + 1] * Vr[c * 3 + 1] + U[a * 3 + 2] * Vr[c * 3 + 2];
        const double dR = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (dR < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
        for (int a = 0; a < 3; a++) t[a] = pc0[a] - (R[a * 3] * pw0[0] + R[a * 3 + 1] * pw0[1] + R[a * 3 + 2] * pw0[2]);
        const double err = pnp_reproj_mean(K, R, t, S.pw, S.uv, n);
        if (isfinite(err) && err < best) { best = err; for (int i = 0; i < 9; i++) S.bestR[i] = R[i]; for (int i = 0; i < 3; i++) S.bestt[i] = t[i]; any = true; }
    }
    return any;
}

__device__ __forceinline__ bool pnp_is_inlier(const PnpCam& K, const double* R, const double* t, const float* p3, const float* p2, float thr) {
    const double X = p3[0], Y = p3[1], Z = p3[2];
    const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0], yc = R[3] * X + R[4] * Y + R[5] * Z + t[1], zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    const float pu = (float)(K.fu * xc / zc + K.uc), pv = (float)(K.fv * yc / zc + K.vc);
    const float du = p2[0] - pu, dv = p2[1] - pv;
    const float e = du * du + dv * dv;
    return e <= thr;
}

// one wave per hypothesis: models[h] = {R (9), t (3)}, counts[h] = inliers (or -1 when EPnP found no model)
__global__ void __launch_bounds__(64) k_pnp_hypotheses(const float* __restrict__ p3, const float* __restrict__ p2, int n, const int32_t* __restrict__ samples,
                                                       PnpCam K, float thr, double* __restrict__ models, int32_t* __restrict__ counts) {
    __shared__ PnpLds S;
    const int lane = threadIdx.x, h = blockIdx.x;
    if (lane == 0) {
        for (int i = 0; i < PNP_MP; i++) {
            const int id = samples[h * PNP_MP + i];
            for (int c = 0; c < 3; c++) S.pw[3 * i + c] = p3[3 * id + c];
            for (int c = 0; c < 2; c++) S.uv[2 * i + c] = p2[2 * id + c];
        }
        S.ok = pnp_epnp_setup(S, K) ? 1 : 0;
    }
    PNP_WAVE_SYNC();
    if (S.ok) {
        pnp_wave_jacobi(12, S.A, S.V, lane);
        PNP_WAVE_SYNC();
        if (lane == 0) S.ok = pnp_epnp_finish(S, K) ? 1 : 0;
        PNP_WAVE_SYNC();
    }
    if (!S.ok) { if (lane == 0) counts[h] = -1; return; }
    double R[9], t[3];
    for (int i = 0; i < 9; i++) R[i] = S.bestR[i];
    for (int i = 0; i < 3; i++) t[i] = S.bestt[i];
    int good = 0;
    for (int i = lane; i < n; i += 64) good += pnp_is_inlier(K, R, t, p3 + 3 * i, p2 + 2 * i, thr) ? 1 : 0;
    good = wave_reduce_sum(good);
    if (lane == 0) {
        counts[h] = good;
        for (int i = 0; i < 9; i++) models[12 * h + i] = R[i];
        for (int i = 0; i < 3; i++) models[12 * h + 9 + i] = t[i];
    }
}

__device__ int pnp_update_iters(double p, double ep, int model_points, int max_iters) {
    p = fmin(fmax(p, 0.0), 1.0); ep = fmin(fmax(ep, 0.0), 1.0);
    double num = fmax(1.0 - p, DBL_MIN), denom = 1.0 - pow(1.0 - ep, (double)model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

__device__ void pnp_se3_exp(const double* d, double* R, double* t) {
    const double wx = d[3], wy = d[4], wz = d[5], th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double A, B, C;
    if (th < 1e-8) { A = 1 - th2 / 6; B = 0.5 - th2 / 24; C = 1.0 / 6 - th2 / 120; }
    else { A = sin(th) / th; B = (1 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9], V[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) W2[i * 3 + j] = W[i * 3] * W[j] + W[i * 3 + 1] * W[3 + j] + W[i * 3 + 2] * W[6 + j];
    for (int i = 0; i < 9; i++) { const double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + A * W[i] + B * W2[i]; V[i] = I + B * W[i] + C * W2[i]; }
    for (int i = 0; i < 3; i++) t[i] = V[i * 3] * d[0] + V[i * 3 + 1] * d[1] + V[i * 3 + 2] * d[2];
}

// sum over the inliers of this wave's matches, fixed order: lane partial sums, then the xor tree
__device__ double pnp_cost(const PnpCam& K, const double* R, const double* t, const float* p3, const float* p2, const uint8_t* mask, int n, int lane) {
    double s = 0;
    for (int i = lane; i < n; i += 64) {
        if (!mask[i]) continue;
        const double X = p3[3 * i], Y = p3[3 * i + 1], Z = p3[3 * i + 2];
        const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0], yc = R[3] * X + R[4] * Y + R[5] * Z + t[1], zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        const double eu = p2[2 * i] - (K.fu * xc / zc + K.uc), ev = p2[2 * i + 1] - (K.fv * yc / zc + K.vc);
        s += eu * eu + ev * ev;
    }
    return wave_reduce_sum(s);
}

// out: pose7 (qx qy qz qw tx ty tz), inlier mask, result[0] = inlier count (0 = no model), result[1] = winning hypothesis
__global__ void __launch_bounds__(64) k_pnp_select_refine(const float* __restrict__ p3, const float* __restrict__ p2, int n, const double* __restrict__ models,
                                                          const int32_t* __restrict__ counts, int iterations, PnpCam K, float thr, double confidence,
                                                          double* __restrict__ pose7, uint8_t* __restrict__ mask, int32_t* __restrict__ result) {
    const int lane = threadIdx.x;
    int best = -1, max_good = 0, niters = iterations;
    for (int it = 0; it < niters; it++) {              // uniform replay of RANSACPointSetRegistrator::run's bookkeeping
        const int good = counts[it];
        if (good < 0) continue;
        if (good > max(max_good, PNP_MP - 1)) {
            best = it; max_good = good;
            niters = pnp_update_iters(confidence, (double)(n - good) / n, PNP_MP, niters);
        }
    }
    if (best < 0) { if (lane == 0) { result[0] = 0; result[1] = -1; } for (int i = lane; i < n; i += 64) mask[i] = 0; return; }
    double R[9], t[3];
    for (int i = 0; i < 9; i++) R[i] = models[12 * best + i];
    for (int i = 0; i < 3; i++) t[i] = models[12 * best + 9 + i];
    for (int i = lane; i < n; i += 64) mask[i] = pnp_is_inlier(K, R, t, p3 + 3 * i, p2 + 2 * i, thr) ? 1 : 0;
    PNP_WAVE_SYNC();
    __threadfence_block();
    double cost = pnp_cost(K, R, t, p3, p2, mask, n, lane), lambda = 1e-3;
    for (int it = 0; it < 50; it++) {
        double H[21], g[6];
#pragma unroll
        for (int a = 0; a < 21; a++) H[a] = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) g[a] = 0;
        for (int i = lane; i < n; i += 64) {
            if (!mask[i]) continue;
            const double X = p3[3 * i], Y = p3[3 * i + 1], Z = p3[3 * i + 2];
            const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0], yc = R[3] * X + R[4] * Y + R[5] * Z + t[1], zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
            const double zi = 1.0 / zc, zi2 = zi * zi;
            const double eu = p2[2 * i] - (K.fu * xc * zi + K.uc), ev = p2[2 * i + 1] - (K.fv * yc * zi + K.vc);
            const double J[12] = {-K.fu * zi, 0, K.fu * xc * zi2, K.fu * xc * yc * zi2, -K.fu - K.fu * xc * xc * zi2, K.fu * yc * zi,
                                  0, -K.fv * zi, K.fv * yc * zi2, K.fv + K.fv * yc * yc * zi2, -K.fv * xc * yc * zi2, -K.fv * xc * zi};
#pragma unroll
            for (int r = 0; r < 6; r++) {
#pragma unroll
                for (int c = 0; c <= r; c++) H[r * (r + 1) / 2 + c] += J[r] * J[c] + J[6 + r] * J[6 + c];
                g[r] -= J[r] * eu + J[6 + r] * ev;
            }
        }
#pragma unroll
        for (int a = 0; a < 21; a++) H[a] = wave_reduce_sum(H[a]);
#pragma unroll
        for (int a = 0; a < 6; a++) g[a] = wave_reduce_sum(g[a]);
        bool improved = false;
        double dxn = 0;
        for (int trial = 0; trial < 10 && !improved; trial++) {
            double A[36], x[6];
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c <= r; c++) A[r * 6 + c] = H[r * (r + 1) / 2 + c];
#pragma unroll
            for (int a = 0; a < 6; a++) A[a * 7] += lambda * (H[a * (a + 1) / 2 + a] > 0 ? H[a * (a + 1) / 2 + a] : 1.0);
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                double d = A[j * 6 + j];
#pragma unroll
                for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k];
                if (!(d > 0)) { ok = false; d = 1.0; }
                A[j * 6 + j] = sqrt(d);
#pragma unroll
                for (int i = j + 1; i < 6; i++) {
                    double v = A[i * 6 + j];
#pragma unroll
                    for (int k = 0; k < j; k++) v -= A[i * 6 + k] * A[j * 6 + k];
                    A[i * 6 + j] = v / A[j * 6 + j];
                }
            }
            if (ok) {
#pragma unroll
                for (int i = 0; i < 6; i++) { double v = g[i]; for (int k = 0; k < i; k++) v -= A[i * 6 + k] * x[k]; x[i] = v / A[i * 7]; }
#pragma unroll
                for (int i = 5; i >= 0; i--) { double v = x[i]; for (int k = i + 1; k < 6; k++) v -= A[k * 6 + i] * x[k]; x[i] = v / A[i * 7]; }
                double dR[9], dt[3], Rn[9], tn[3];
                pnp_se3_exp(x, dR, dt);
                for (int a = 0; a < 3; a++) {
                    for (int b = 0; b < 3; b++) Rn[a * 3 + b] = dR[a * 3] * R[b] + dR[a * 3 + 1] * R[3 + b] + dR[a * 3 + 2] * R[6 + b];
                    tn[a] = dR[a * 3] * t[0] + dR[a * 3 + 1] * t[1] + dR[a * 3 + 2] * t[2] + dt[a];
                }
                const double c2 = pnp_cost(K, Rn, tn, p3, p2, mask, n, lane);
                if (isfinite(c2) && c2 <= cost) {
                    for (int a = 0; a < 9; a++) R[a] = Rn[a];
                    for (int a = 0; a < 3; a++) t[a] = tn[a];
                    cost = c2; improved = true;
                    lambda = fmax(lambda * 0.1, 1e-12);
                    dxn = 0;
                    for (int a = 0; a < 6; a++) dxn = fmax(dxn, fabs(x[a]));
                    continue;
                }
            }
            lambda *= 10;
        }
        if (!improved || dxn < 1e-12) break;
    }
    if (lane == 0) {
        const double tr = R[0] + R[4] + R[8];
        double x, y, z, w;
        if (tr > 0) { const double s = sqrt(tr + 1.0) * 2; w = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s; }
        else if (R[0] > R[4] && R[0] > R[8]) { const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; x = 0.25 * s; w = (R[7] - R[5]) / s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s; }
        else if (R[4] > R[8]) { const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; y = 0.25 * s; w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; z = (R[5] + R[7]) / s; }
        else { const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; z = 0.25 * s; w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; }
        pose7[0] = x; pose7[1] = y; pose7[2] = z; pose7[3] = w; pose7[4] = t[0]; pose7[5] = t[1]; pose7[6] = t[2];
        result[0] = max_good; result[1] = best;
    }
}

namespace {
struct CvRng {                                             // cv::RNG (multiply-with-carry)
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffu) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
struct PnpBuf {
    std::vector<void*> ptrs;
    ~PnpBuf() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T> hipError_t alloc(T** p, size_t count) {
        hipError_t e = hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
};
}  // namespace
}  // namespace myslam_hip

using namespace myslam_hip;

extern "C" {

int myslam_solve_pnp_ransac(const float* pts3d, const float* pts2d, int n, double fx, double fy, double cx, double cy, int iterations,
                            double reproj_error, double confidence, double* pose7, uint8_t* inlier, int* n_inliers) {
    if (n < 0 || iterations < 1 || iterations > 100000 || (n > 0 && (!pts3d || !pts2d)) || !pose7) return MYSLAM_ERR_INVALID;
    if (n_inliers) *n_inliers = 0;
    if (inlier && n) memset(inlier, 0, (size_t)n);
    if (n < PNP_MP) return MYSLAM_ERR_UNSUPPORTED;          // OpenCV: fewer points than the model needs -> false
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    // RANSACPointSetRegistrator::getSubset: 5 distinct uniform indices per iteration, one generator for the whole run
    std::vector<int32_t> samples((size_t)iterations * PNP_MP);
    CvRng rng((uint64_t)-1);
    for (int it = 0; it < iterations; it++) {
        int32_t* idx = &samples[(size_t)it * PNP_MP];
        for (int i = 0; i < PNP_MP; i++) {
            for (;;) {
                idx[i] = rng.uniform(0, n);
                int j = 0;
                for (; j < i; j++) if (idx[j] == idx[i]) break;
                if (j == i) break;
            }
        }
    }
    const hipStream_t st = host_call_stream();             // everything below runs on this thread's own non-blocking stream (never the legacy stream: common.h)
    if (!st) return MYSLAM_ERR_HIP;
    PnpBuf mem;
    float *d_p3, *d_p2; int32_t *d_samples, *d_counts, *d_result; double *d_models, *d_pose; uint8_t* d_mask;
    MYSLAM_HIP_CHECK(mem.alloc(&d_p3, (size_t)3 * n)); MYSLAM_HIP_CHECK(mem.alloc(&d_p2, (size_t)2 * n));
    MYSLAM_HIP_CHECK(mem.alloc(&d_samples, samples.size())); MYSLAM_HIP_CHECK(mem.alloc(&d_counts, (size_t)iterations));
    MYSLAM_HIP_CHECK(mem.alloc(&d_result, 2)); MYSLAM_HIP_CHECK(mem.alloc(&d_models, (size_t)12 * iterations));
    MYSLAM_HIP_CHECK(mem.alloc(&d_pose, 7)); MYSLAM_HIP_CHECK(mem.alloc(&d_mask, (size_t)n));
    { const int rc_ = copy_sync(d_p3, pts3d, sizeof(float) * 3 * n, hipMemcpyHostToDevice, st); if (rc_) return rc_; }
    { const int rc_ = copy_sync(d_p2, pts2d, sizeof(float) * 2 * n, hipMemcpyHostToDevice, st); if (rc_) return rc_; }
    { const int rc_ = copy_sync(d_samples, samples.data(), sizeof(int32_t) * samples.size(), hipMemcpyHostToDevice, st); if (rc_) return rc_; }
    const PnpCam K{fx, fy, cx, cy};
    const float thr = (float)(reproj_error * reproj_error);
    hipLaunchKernelGGL(k_pnp_hypotheses, dim3(iterations), dim3(64), 0, st, d_p3, d_p2, n, d_samples, K, thr, d_models, d_counts);
    hipLaunchKernelGGL(k_pnp_select_refine, dim3(1), dim3(64), 0, st, d_p3, d_p2, n, d_models, d_counts, iterations, K, thr, confidence, d_pose,
                       d_mask, d_result);
    MYSLAM_HIP_CHECK(hipGetLastError());
    int32_t res[2];
    { const int rc_ = copy_sync(res, d_result, sizeof(res), hipMemcpyDeviceToHost, st); if (rc_) return rc_; }
    if (res[0] <= 0) return MYSLAM_ERR_UNSUPPORTED;         // no model: OpenCV returns false and leaves rvec / tvec alone
    { const int rc_ = copy_sync(pose7, d_pose, sizeof(double) * 7, hipMemcpyDeviceToHost, st); if (rc_) return rc_; }
    if (inlier) { const int rc_ = copy_sync(inlier, d_mask, (size_t)n, hipMemcpyDeviceToHost, st); if (rc_) return rc_; }
    if (n_inliers) *n_inliers = res[0];
    return MYSLAM_OK;
}

}  // extern "C"
