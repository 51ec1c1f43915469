// ba.hip — sliding-window local BA: residual + Jacobian + block normal-equation build on gfx950.
// Replaces the per-edge work g2o performs for Backend::OptimizeActiveMap (reference src/backend.cpp:126-232):
//   EdgeProjection::computeError / linearizeOplus      include/myslam/g2o_types.h:115-144
//   RobustKernelHuber (delta = 5.991, backend.cpp:198-200) and the weighted block quadratic form
//   (SURVEY.md Appendix A.7):  Hpp += w Jx^T Jx, Hll += w Jp^T Jp, Hpl = w Jx^T Jp, bp -= w Jx^T e, bl -= w Jp^T e.
// f64 throughout (g2o is f64).  One 256-thread block per window; the window's pose blocks (6x6 upper
// triangle + 6) and landmark blocks (3x3 upper + 3) are accumulated in LDS with ds_add_f64, the per-edge
// 6x3 Hpl blocks stream straight to HBM.  Summation order is not fixed -> results agree with the oracle to
// ~1e-12 relative, not bit-exactly (stated in the tests).
#include "common.h"

namespace myslam_hip {

struct BaArgs {
    const double* poses; const double* points; const int32_t* ep; const int32_t* el; const double* obs; const uint8_t* fixed;
    const int32_t* sizes;              // nwin x 3 or NULL (then n* below)
    int nposes, npts, nedges;
    int maxP, maxL, maxE;
    double fx, fy, cx, cy, delta;
    double *Hpp, *Hll, *Hpl, *bp, *bl, *chi2;
};

__global__ __launch_bounds__(256) void k_ba_build(BaArgs a) {
    extern __shared__ __attribute__((aligned(16))) double s_d[];
    const int w = blockIdx.x, t = threadIdx.x;
    const int P = a.sizes ? a.sizes[3 * w] : a.nposes;
    const int L = a.sizes ? a.sizes[3 * w + 1] : a.npts;
    const int E = a.sizes ? a.sizes[3 * w + 2] : a.nedges;
    double* sR = s_d;                        // maxP x 12 (R row-major, t)
    double* sHpp = sR + a.maxP * 12;         // maxP x 21 (upper triangle, row-major)
    double* sbp = sHpp + a.maxP * 21;        // maxP x 6
    double* sHll = sbp + a.maxP * 6;         // maxL x 6
    double* sbl = sHll + a.maxL * 6;         // maxL x 3
    const double* poses = a.poses + (size_t)w * a.maxP * 7;
    const double* pts = a.points + (size_t)w * a.maxL * 3;
    const int32_t* ep = a.ep + (size_t)w * a.maxE;
    const int32_t* el = a.el + (size_t)w * a.maxE;
    const double* obs = a.obs + (size_t)w * a.maxE * 2;
    const uint8_t* fixed = a.fixed ? a.fixed + (size_t)w * a.maxL : nullptr;

    for (int i = t; i < a.maxP * 27 + a.maxL * 9; i += 256) sHpp[i] = 0.0;
    for (int p = t; p < P; p += 256) {
        double x = poses[7 * p], y = poses[7 * p + 1], z = poses[7 * p + 2], q = poses[7 * p + 3];
        const double n = sqrt(x * x + y * y + z * z + q * q);
        x /= n; y /= n; z /= n; q /= n;
        double* R = sR + 12 * p;
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * q);     R[2] = 2 * (x * z + y * q);
        R[3] = 2 * (x * y + z * q);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * q);
        R[6] = 2 * (x * z - y * q);     R[7] = 2 * (y * z + x * q);     R[8] = 1 - 2 * (x * x + y * y);
        R[9] = poses[7 * p + 4]; R[10] = poses[7 * p + 5]; R[11] = poses[7 * p + 6];
    }
    __syncthreads();

    for (int k = t; k < E; k += 256) {
        const int ip = ep[k], il = el[k];
        double* hpl = a.Hpl + ((size_t)w * a.maxE + k) * 18;
        if (ip < 0 || ip >= P || il < 0 || il >= L) {             // malformed edge: contributes nothing
            a.chi2[(size_t)w * a.maxE + k] = 0.0;
#pragma unroll
            for (int i = 0; i < 18; i++) hpl[i] = 0.0;
            continue;
        }
        const double* R = sR + 12 * ip;
        const double pw0 = pts[3 * il], pw1 = pts[3 * il + 1], pw2 = pts[3 * il + 2];
        const double X = R[0] * pw0 + R[1] * pw1 + R[2] * pw2 + R[9];
        const double Y = R[3] * pw0 + R[4] * pw1 + R[5] * pw2 + R[10];
        const double Z = R[6] * pw0 + R[7] * pw1 + R[8] * pw2 + R[11];
        const double e0 = obs[2 * k] - (a.fx * X / Z + a.cx);      // g2o_types.h:119-121
        const double e1 = obs[2 * k + 1] - (a.fy * Y / Z + a.cy);
        const double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;   // :133-134
        double J[12];
        J[0] = -a.fx * Zinv; J[1] = 0; J[2] = a.fx * X * Zinv2; J[3] = a.fx * X * Y * Zinv2;
        J[4] = -a.fx - a.fx * X * X * Zinv2; J[5] = a.fx * Y * Zinv;
        J[6] = 0; J[7] = -a.fy * Zinv; J[8] = a.fy * Y * Zinv2; J[9] = a.fy + a.fy * Y * Y * Zinv2;
        J[10] = -a.fy * X * Y * Zinv2; J[11] = -a.fy * X * Zinv;
        double Jp[6];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) Jp[r * 3 + c] = J[r * 6] * R[c] + J[r * 6 + 1] * R[3 + c] + J[r * 6 + 2] * R[6 + c];   // :140-141
        const double e2 = e0 * e0 + e1 * e1;
        a.chi2[(size_t)w * a.maxE + k] = e2;
        const double wgt = (e2 <= a.delta * a.delta) ? 1.0 : a.delta / sqrt(e2);   // Huber rho'
        double* hp = sHpp + 21 * ip;
        int u = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
#pragma unroll
            for (int c = r; c < 6; c++) atomicAdd(&hp[u++], wgt * (J[r] * J[c] + J[6 + r] * J[6 + c]));
            atomicAdd(&sbp[6 * ip + r], -wgt * (J[r] * e0 + J[6 + r] * e1));
        }
        const bool fx_pt = fixed && fixed[il];
        if (!fx_pt) {
            double* hl = sHll + 6 * il;
            int v = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) {
#pragma unroll
                for (int c = r; c < 3; c++) atomicAdd(&hl[v++], wgt * (Jp[r] * Jp[c] + Jp[3 + r] * Jp[3 + c]));
                atomicAdd(&sbl[3 * il + r], -wgt * (Jp[r] * e0 + Jp[3 + r] * e1));
            }
        }
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) hpl[r * 3 + c] = fx_pt ? 0.0 : wgt * (J[r] * Jp[c] + J[6 + r] * Jp[3 + c]);
    }
    __syncthreads();

    for (int i = t; i < P * 36; i += 256) {
        const int p = i / 36, r = (i % 36) / 6, c = i % 6;
        const int rr = min(r, c), cc = max(r, c);
        const int u = rr * 6 - rr * (rr - 1) / 2 + (cc - rr);       // index in the row-major upper triangle
        a.Hpp[((size_t)w * a.maxP + p) * 36 + r * 6 + c] = sHpp[21 * p + u];
    }
    for (int i = t; i < P * 6; i += 256) a.bp[(size_t)w * a.maxP * 6 + i] = sbp[i];
    for (int i = t; i < L * 9; i += 256) {
        const int l = i / 9, r = (i % 9) / 3, c = i % 3;
        const int rr = min(r, c), cc = max(r, c);
        const int u = rr * 3 - rr * (rr - 1) / 2 + (cc - rr);
        a.Hll[((size_t)w * a.maxL + l) * 9 + r * 3 + c] = sHll[6 * l + u];
    }
    for (int i = t; i < L * 3; i += 256) a.bl[(size_t)w * a.maxL * 3 + i] = sbl[i];
}

static size_t ba_lds(int maxP, int maxL) { return sizeof(double) * ((size_t)maxP * 39 + (size_t)maxL * 9); }

static int ba_launch(const BaArgs& a, int nwin, hipStream_t s) {
    const size_t lds = ba_lds(a.maxP, a.maxL);
    if (lds > 150 * 1024) return MYSLAM_ERR_CAPACITY;
    static size_t attr = 0;
    if (lds > 48 * 1024 && lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    ScopedProf sp(P_BA, s);
    hipLaunchKernelGGL(k_ba_build, dim3(nwin), dim3(256), lds, s, a);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

}  // namespace myslam_hip

using namespace myslam_hip;

extern "C" {

int myslam_ba_build_batch(const double* d_poses, const double* d_points, const int32_t* d_edge_pose, const int32_t* d_edge_pt,
                          const double* d_obs, const uint8_t* d_fixed, const int32_t* d_sizes, int nwin, int max_poses,
                          int max_pts, int max_edges, double fx, double fy, double cx, double cy, double huber_delta,
                          double* d_Hpp, double* d_Hll, double* d_Hpl, double* d_bp, double* d_bl, double* d_chi2, void* hip_stream) {
    if (!d_poses || !d_points || !d_edge_pose || !d_edge_pt || !d_obs || !d_sizes || nwin < 1 || max_poses < 1 || max_pts < 1 ||
        max_edges < 1 || !d_Hpp || !d_Hll || !d_Hpl || !d_bp || !d_bl || !d_chi2)
        return MYSLAM_ERR_INVALID;
    BaArgs a{d_poses, d_points, d_edge_pose, d_edge_pt, d_obs, d_fixed, d_sizes, 0, 0, 0, max_poses, max_pts, max_edges,
             fx, fy, cx, cy, huber_delta, d_Hpp, d_Hll, d_Hpl, d_bp, d_bl, d_chi2};
    return ba_launch(a, nwin, (hipStream_t)hip_stream);
}

int myslam_ba_build(const double* poses, int nposes, const double* points, int npts, const int32_t* edge_pose,
                    const int32_t* edge_pt, const double* obs, int nedges, const uint8_t* fixed_pt, double fx, double fy, double cx,
                    double cy, double huber_delta, double* Hpp, double* Hll, double* Hpl, double* bp, double* bl, double* chi2) {
    if (!poses || !points || nposes < 1 || npts < 1 || nedges < 0 || !Hpp || !Hll || !bp || !bl) return MYSLAM_ERR_INVALID;
    if (nedges > 0 && (!edge_pose || !edge_pt || !obs || !Hpl || !chi2)) return MYSLAM_ERR_INVALID;
    for (int k = 0; k < nedges; k++)
        if (edge_pose[k] < 0 || edge_pose[k] >= nposes || edge_pt[k] < 0 || edge_pt[k] >= npts) return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    const int E = nedges > 0 ? nedges : 1;
    const size_t nd = (size_t)nposes * 7 + (size_t)npts * 3 + (size_t)E * 2 + (size_t)nposes * 36 + (size_t)npts * 9 +
                      (size_t)E * 18 + (size_t)nposes * 6 + (size_t)npts * 3 + (size_t)E;
    double* d = nullptr; int32_t* di = nullptr; uint8_t* df = nullptr;
    MYSLAM_HIP_CHECK(hipMalloc((void**)&d, nd * sizeof(double)));
    MYSLAM_HIP_CHECK(hipMalloc((void**)&di, (size_t)E * 2 * sizeof(int32_t)));
    MYSLAM_HIP_CHECK(hipMalloc((void**)&df, (size_t)npts));
    double* d_poses = d; double* d_pts = d_poses + (size_t)nposes * 7; double* d_obs = d_pts + (size_t)npts * 3;
    double* d_Hpp = d_obs + (size_t)E * 2; double* d_Hll = d_Hpp + (size_t)nposes * 36; double* d_Hpl = d_Hll + (size_t)npts * 9;
    double* d_bp = d_Hpl + (size_t)E * 18; double* d_bl = d_bp + (size_t)nposes * 6; double* d_chi = d_bl + (size_t)npts * 3;
    MYSLAM_HIP_CHECK(hipMemcpy(d_poses, poses, sizeof(double) * nposes * 7, hipMemcpyHostToDevice));
    MYSLAM_HIP_CHECK(hipMemcpy(d_pts, points, sizeof(double) * npts * 3, hipMemcpyHostToDevice));
    if (nedges) {
        MYSLAM_HIP_CHECK(hipMemcpy(d_obs, obs, sizeof(double) * nedges * 2, hipMemcpyHostToDevice));
        MYSLAM_HIP_CHECK(hipMemcpy(di, edge_pose, sizeof(int32_t) * nedges, hipMemcpyHostToDevice));
        MYSLAM_HIP_CHECK(hipMemcpy(di + E, edge_pt, sizeof(int32_t) * nedges, hipMemcpyHostToDevice));
    }
    if (fixed_pt) MYSLAM_HIP_CHECK(hipMemcpy(df, fixed_pt, npts, hipMemcpyHostToDevice));
    BaArgs a{d_poses, d_pts, di, di + E, d_obs, fixed_pt ? df : nullptr, nullptr, nposes, npts, nedges, nposes, npts, E,
             fx, fy, cx, cy, huber_delta, d_Hpp, d_Hll, d_Hpl, d_bp, d_bl, d_chi};
    int rc = ba_launch(a, 1, nullptr);
    if (rc) return rc;
    MYSLAM_HIP_CHECK(hipMemcpy(Hpp, d_Hpp, sizeof(double) * nposes * 36, hipMemcpyDeviceToHost));
    MYSLAM_HIP_CHECK(hipMemcpy(Hll, d_Hll, sizeof(double) * npts * 9, hipMemcpyDeviceToHost));
    MYSLAM_HIP_CHECK(hipMemcpy(bp, d_bp, sizeof(double) * nposes * 6, hipMemcpyDeviceToHost));
    MYSLAM_HIP_CHECK(hipMemcpy(bl, d_bl, sizeof(double) * npts * 3, hipMemcpyDeviceToHost));
    if (nedges) {
        MYSLAM_HIP_CHECK(hipMemcpy(Hpl, d_Hpl, sizeof(double) * nedges * 18, hipMemcpyDeviceToHost));
        MYSLAM_HIP_CHECK(hipMemcpy(chi2, d_chi, sizeof(double) * nedges, hipMemcpyDeviceToHost));
    }
    (void)hipFree(d); (void)hipFree(di); (void)hipFree(df);
    return MYSLAM_OK;
}

}  // extern "C"
