// ba.hip — sliding-window local BA: residual + Jacobian + block normal-equation build on gfx950.
// Replaces the per-edge work g2o performs for Backend::OptimizeActiveMap (reference src/backend.cpp:126-232):
//   EdgeProjection::computeError / linearizeOplus      include/myslam/g2o_types.h:115-144
//   RobustKernelHuber (delta = 5.991, backend.cpp:198-200) and the weighted block quadratic form
//   (SURVEY.md Appendix A.7):  Hpp += w Jx^T Jx, Hll += w Jp^T Jp, Hpl = w Jx^T Jp, bp -= w Jx^T e, bl -= w Jp^T e.
// f64 throughout (g2o is f64).  One 256-thread block per window; pose and landmark blocks are summed in a fixed order (two runs
// give the same bits, see k_ba_build), the per-edge 6x3 Hpl blocks stream straight to HBM.  The order differs
// from the oracle's plain edge loop -> agreement to ~1e-12 relative, not bit-exactly (stated in the tests).
#include <algorithm>
#include <atomic>
#include <unordered_map>
#include <vector>
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

// ------------------------------------------------------------------------------------------------
// Levenberg-Marquardt on device: what g2o's OptimizationAlgorithmLevenberg + BlockSolver_6_3 do for
// optimizer.optimize(n) in Backend::OptimizeActiveMap (src/backend.cpp:212-214):
// per iteration build H/b, then up to 10 trials of { (H + lambda I) x = b via Schur complement on the
// landmarks, dense Cholesky of the reduced 6P x 6P system, back-substitution, oplus (SE3 left update,
// g2o_types.h:32-37; additive points :50-54), gain ratio, lambda update }.  One block per window, the whole
// state (poses, points, backups, Hpp/Hll/Hinv, the reduced system) lives in LDS; only the per-edge 6x3
// blocks go through HBM scratch.  Edges must be grouped by landmark (as the reference builds them,
// backend.cpp:161-205: for each map point, its observations).
// ------------------------------------------------------------------------------------------------
struct BaOptArgs {
    double* poses; double* points; const int32_t* ep; const int32_t* el; const double* obs; const uint8_t* fixed;
    const int32_t* sizes; int nposes, npts, nedges; int maxP, maxL, maxE;
    double fx, fy, cx, cy, delta; int max_iters;
    double* W;            // scratch: nwin x maxE x 18
    double* final_chi2; int32_t* iters; int32_t* status;
    // Backend::OptimizeActiveMap outer loop (backend.cpp:208-243); edge_chi2 == nullptr -> a single optimize(max_iters)
    int rounds; double chi2_th; double* edge_chi2; uint8_t* outlier; int32_t* rounds_out; int32_t* nout_out;
    size_t wstride = 0;   // scratch doubles per window; 0 = maxE x 18 (the batch entry points' contract).  The host-pointer entry points size it
                          // themselves: a window with many short-lived landmarks (22 doubles each in the HBM form) may need more than its edges give
};

__device__ __forceinline__ void ba_edge(const double* R, const double* pw, const double* z, double fx, double fy, double cx, double cy,
                                        double& e0, double& e1, double* J, double* Jp) {
    const double X = R[0] * pw[0] + R[1] * pw[1] + R[2] * pw[2] + R[9];
    const double Y = R[3] * pw[0] + R[4] * pw[1] + R[5] * pw[2] + R[10];
    const double Z = R[6] * pw[0] + R[7] * pw[1] + R[8] * pw[2] + R[11];
    e0 = z[0] - (fx * X / Z + cx);
    e1 = z[1] - (fy * Y / Z + cy);
    if (!J) return;
    const double Zinv = 1.0 / (Z + 1e-18), Zinv2 = Zinv * Zinv;
    J[0] = -fx * Zinv; J[1] = 0; J[2] = fx * X * Zinv2; J[3] = fx * X * Y * Zinv2; J[4] = -fx - fx * X * X * Zinv2; J[5] = fx * Y * Zinv;
    J[6] = 0; J[7] = -fy * Zinv; J[8] = fy * Y * Zinv2; J[9] = fy + fy * Y * Y * Zinv2; J[10] = -fy * X * Y * Zinv2; J[11] = -fy * X * Zinv;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) Jp[r * 3 + c] = J[r * 6] * R[c] + J[r * 6 + 1] * R[3 + c] + J[r * 6 + 2] * R[6 + c];
}

typedef double ba_d4 __attribute__((ext_vector_type(4)));

// f64 wave sum on the DPP network (no LDS crossbar): the total lands in lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double wave_sum_lane63(double v) {
    v = dpp_shift_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_shift_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_shift_add<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_shift_add<0x140, 0xf>(v);     // row_mirror: every lane holds its 16-lane row sum
    v = dpp_shift_add<0x142, 0xa>(v);     // row_bcast15 into rows 1 and 3
    v = dpp_shift_add<0x143, 0xc>(v);     // row_bcast31 into rows 2 and 3
    return v;
}
// 32 f64 wave sums at once by halving: each step exchanges HALF of the values a lane still holds with a partner lane and adds the other half,
// so 32 values cost 16 + 8 + 4 + 2 + 1 + 1 exchange-and-add steps instead of 32 x 6 (v_permlane32/16_swap for the two widest steps: one
// instruction moves both directions).  On return lane l holds the total of value l >> 1 in h[0] (both lanes of a pair).
template <int CTRL>
__device__ __forceinline__ double dpp_fetch(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ void wave_sum32_halving(double* h, int lane) {
#pragma unroll
    for (int u = 0; u < 16; u++) {                        // lanes 0..31 <- value u of lanes l, l + 32; lanes 32..63 <- value u + 16
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(h[u]), (unsigned)__double2loint(h[u + 16]), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(h[u]), (unsigned)__double2hiint(h[u + 16]), false, false);
        h[u] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {                         // even rows of 16 lanes <- value u, odd rows <- value u + 8
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(h[u]), (unsigned)__double2loint(h[u + 8]), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(h[u]), (unsigned)__double2hiint(h[u + 8]), false, false);
        h[u] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int u = 0; u < 4; u++) h[u] = (b3 ? h[u + 4] : h[u]) + dpp_fetch<0x128>(b3 ? h[u] : h[u + 4]);      // row_ror:8 = lane ^ 8
#pragma unroll
    for (int u = 0; u < 2; u++) h[u] = (b2 ? h[u + 2] : h[u]) + dpp_fetch<0x141>(b2 ? h[u] : h[u + 2]);      // row_half_mirror: flips bits 0..2
    h[0] = (b1 ? h[1] : h[0]) + dpp_fetch<0x4E>(b1 ? h[0] : h[1]);                                          // quad_perm [2,3,0,1]
    h[0] += dpp_fetch<0xB1>(h[0]);                                                                          // quad_perm [1,0,3,2]
}
__device__ __forceinline__ double bcast_lane(double v, int srcLane) {       // srcLane must be wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), srcLane), __builtin_amdgcn_readlane(__double2loint(v), srcLane));
}

// Block build of one window with a fixed summation order (two runs give the same bits):
//   landmark blocks  the edges of a landmark normally form one contiguous run (the order Backend::OptimizeActiveMap emits them,
//                    backend.cpp:166-206): a lane pair walks the run (one half each), sums Hll / bl in registers and writes the
//                    edges' chi2 and Hpl;
//   pose blocks      LISTS (the default): the window's edges are counting-sorted by pose into an index list in LDS (integer ballots and
//                    prefix sums: a pose's edges in edge order), then ONE WAVE PER POSE walks its list — lane l takes entries l, l + 64, … —,
//                    evaluates each edge a second time and keeps the 27 pose terms (upper triangle of w Jx^T Jx, -w Jx^T e) in registers;
//                    the 27 wave sums are taken together by halving (wave_sum32_halving) and go straight to Hpp / bp.  Round 5: the
//                    earlier form added every edge's 27 terms into per-wave LDS copies with ds_add_f64 — 81 k atomic lane operations
//                    per window, most of them on the SAME few addresses (a landmark's run visits the poses in order, so half of a
//                    wave's lanes hit one pose at a time): the LDS pipe of the CU serialised them and was what the kernel waited for.
//                    A second evaluation of an edge (~100 f64 operations) is cheaper for a LONE window (63 -> 44 us; with the window's
//                    arrays staged in LDS first, see below, less); a batch of 512 windows is not bound there (0.167 -> 0.181 ms) and stays
//                    on the earlier form.
//                    !LISTS (batches, and windows whose arrays do not fit LDS): the earlier form — the same lanes add each edge's 27
//                    pose terms into ITS WAVE's private copy of the pose blocks in LDS (ds_add_f64: lanes of one instruction that hit
//                    the same address are served in lane order, instructions in program order — nothing depends on how the waves
//                    interleave); the copies are added in wave order.
//   A landmark whose edges are scattered over several runs takes the slow road: its edges are evaluated one per thread and its block
//   is summed by one thread scanning the edge list in order.
//   NT = threads per window: 256 for batches (one block per window fills the chip), 1024 for a handful of windows (a live stream builds ONE
//   window per key-frame: every landmark's lane pair then exists at once).
template <int NT, bool LISTS>
__global__ __launch_bounds__(NT) void k_ba_build(BaArgs a) {
    MYSLAM_SIDE_PRIO();
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) double s_d[];
    const int w = blockIdx.x, t = threadIdx.x, wv = t >> 6;
    int P = a.sizes ? a.sizes[3 * w] : a.nposes;
    int L = a.sizes ? a.sizes[3 * w + 1] : a.npts;
    int E = a.sizes ? a.sizes[3 * w + 2] : a.nedges;
    // a window whose sizes do not fit the common capacities would overrun LDS: it is skipped (outputs zero, chi2[0] = -1)
    const bool oversize = P < 0 || L < 0 || E < 0 || P > a.maxP || L > a.maxL || E > a.maxE;
    if (oversize) { P = 0; L = 0; E = 0; if (t == 0) a.chi2[(size_t)w * a.maxE] = -1.0; }
    double* sR = s_d;                                           // maxP x 12 (R row-major, t)
    double* sAcc = sR + a.maxP * 12;                            // !LISTS: NW waves x maxP x 27 (6x6 upper triangle row-major, then b)
    double* s_obs = sAcc;                                       // LISTS: maxE x 2 — the window's observations, landmarks and edge indices are
    double* s_pts = s_obs + 2 * a.maxE;                         //        staged once (coalesced): a lone window's time is a chain of dependent loads
    int* s_runs = reinterpret_cast<int*>(LISTS ? s_pts + 3 * a.maxL : sAcc + NW * a.maxP * 27);   // maxL: contiguous runs of each landmark in the edge list
    int* s_start = s_runs + a.maxL;                             // maxL: first edge of the landmark's run (meaningful when it has exactly one)
    int* s_len = s_start + a.maxL;                              // maxL: length of that run
    const int nch = (a.maxE + 63) >> 6;                         // LISTS: chunks of 64 consecutive edges
    int* s_cnt = s_len + a.maxL;                                // LISTS: maxP x nch — edges of pose p in chunk c, then their exclusive prefix
    int* s_ptot = s_cnt + a.maxP * nch;                         // LISTS: maxP — edges of pose p
    int* s_poff = s_ptot + a.maxP;                              // LISTS: maxP — where pose p's entries start in the list
    int* s_list = s_poff + a.maxP;                              // LISTS: maxE — edge indices grouped by pose, edge order within a pose
    int* s_el = s_list + a.maxE;                                // LISTS: maxE
    int* s_ep = s_el + a.maxE;                                  // LISTS: maxE
    const double* poses = a.poses + (size_t)w * a.maxP * 7;
    const double* gpts = a.points + (size_t)w * a.maxL * 3;
    const int32_t* gep = a.ep + (size_t)w * a.maxE;
    const int32_t* gel = a.el + (size_t)w * a.maxE;
    const double* gobs = a.obs + (size_t)w * a.maxE * 2;
    const double* pts = LISTS ? s_pts : gpts;
    const int32_t* ep = LISTS ? s_ep : gep;
    const int32_t* el = LISTS ? s_el : gel;
    const double* obs = LISTS ? s_obs : gobs;
    const uint8_t* fixed = a.fixed ? a.fixed + (size_t)w * a.maxL : nullptr;
    const double d2 = a.delta * a.delta;
    double* myAcc = sAcc + (size_t)wv * a.maxP * 27;

    if (LISTS) {
        for (int i = t; i < E; i += NT) { s_el[i] = gel[i]; s_ep[i] = gep[i]; }
        for (int i = t; i < 2 * E; i += NT) s_obs[i] = gobs[i];
        for (int i = t; i < 3 * L; i += NT) s_pts[i] = gpts[i];
    } else
        for (int i = t; i < NW * a.maxP * 27; i += NT) sAcc[i] = 0.0;
    for (int l = t; l < L; l += NT) s_runs[l] = 0;
    for (int p = t; p < P; p += NT) {
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
    // runs of every landmark: a run starts where the landmark index changes (integer counts: the order of the adds does not matter)
    for (int k = t; k < E; k += NT) {
        const int il = el[k];
        if (il >= 0 && il < L && (k == 0 || el[k - 1] != il)) {
            const int earlier = atomicAdd(&s_runs[il], 1);
            int n = 1;
            while (k + n < E && el[k + n] == il) n++;
            if (earlier == 0) { s_start[il] = k; s_len[il] = n; }      // one writer per landmark (a landmark with several runs does not use these)
        }
    }
    if (LISTS) {
        // counting sort of the well-formed edges by pose.  Pass 1: per chunk of 64 edges and pose, the number of edges (a ballot per pose).
        const int lane = t & 63, nchE = (E + 63) >> 6;
        for (int c = wv; c < nchE; c += NW) {
            const int k = c * 64 + lane;
            int ip = -1;
            if (k < E) { const int il = el[k]; ip = (il >= 0 && il < L) ? ep[k] : -1; }
            for (int p = 0; p < P; p++) {
                const unsigned long long m = __ballot(ip == p);
                if (lane == 0) s_cnt[p * nch + c] = __popcll(m);
            }
        }
        __syncthreads();
        // exclusive prefix over the chunks of each pose (a wave per pose, 64 chunks per step) and the pose totals
        for (int p = wv; p < P; p += NW) {
            int carry = 0;
            for (int c0 = 0; c0 < nchE; c0 += 64) {
                const int c = c0 + lane;
                const int v = c < nchE ? s_cnt[p * nch + c] : 0;
                int incl = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if (lane >= d) incl += u; }
                if (c < nchE) s_cnt[p * nch + c] = carry + incl - v;
                carry += __shfl(incl, 63, 64);
            }
            if (lane == 0) s_ptot[p] = carry;
        }
        __syncthreads();
        if (t < P) { int o = 0; for (int q = 0; q < t; q++) o += s_ptot[q]; s_poff[t] = o; }
        __syncthreads();
        // Pass 2: every edge to its slot (pose start + edges of the pose in earlier chunks + rank in its chunk)
        for (int c = wv; c < nchE; c += NW) {
            const int k = c * 64 + lane;
            int ip = -1;
            if (k < E) { const int il = el[k]; ip = (il >= 0 && il < L) ? ep[k] : -1; }
            for (int p = 0; p < P; p++) {
                const unsigned long long m = __ballot(ip == p);
                if (ip == p) s_list[s_poff[p] + s_cnt[p * nch + c] + __popcll(m & ((1ull << lane) - 1ull))] = k;
            }
        }
    }
    __syncthreads();
    // one edge: chi2, Hpl, (!LISTS) the pose terms into this wave's copy; hl (may be null) collects the landmark terms
    auto edge = [&](int k, int il, double* hl) {
        double* hpl = a.Hpl + ((size_t)w * a.maxE + k) * 18;
        const int ip = ep[k];
        if (ip < 0 || ip >= P) {                                // malformed edge: contributes nothing
            a.chi2[(size_t)w * a.maxE + k] = 0.0;
#pragma unroll
            for (int i = 0; i < 18; i++) hpl[i] = 0.0;
            return;
        }
        double e0, e1, J[12], Jp[6];
        ba_edge(sR + 12 * ip, pts + 3 * il, obs + 2 * k, a.fx, a.fy, a.cx, a.cy, e0, e1, J, Jp);
        const double e2 = e0 * e0 + e1 * e1;
        a.chi2[(size_t)w * a.maxE + k] = e2;
        const double wgt = (e2 <= d2) ? 1.0 : a.delta / sqrt(e2);   // Huber rho'
        if (!LISTS) {
            double* hp = myAcc + 27 * ip;
            int u = 0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
#pragma unroll
                for (int c = r; c < 6; c++) atomicAdd(&hp[u++], wgt * (J[r] * J[c] + J[6 + r] * J[6 + c]));
            }
#pragma unroll
            for (int r = 0; r < 6; r++) atomicAdd(&hp[21 + r], -wgt * (J[r] * e0 + J[6 + r] * e1));
        }
        const bool fx_pt = fixed && fixed[il];
        if (hl && !fx_pt) {
            int v = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) {
#pragma unroll
                for (int c = r; c < 3; c++) hl[v++] += wgt * (Jp[r] * Jp[c] + Jp[3 + r] * Jp[3 + c]);
            }
#pragma unroll
            for (int r = 0; r < 3; r++) hl[6 + r] += -wgt * (Jp[r] * e0 + Jp[3 + r] * e1);
        }
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) hpl[r * 3 + c] = fx_pt ? 0.0 : wgt * (J[r] * Jp[c] + J[6 + r] * Jp[3 + c]);
    };
    auto lm_store = [&](int il, const double* hl) {
        double* Hl = a.Hll + ((size_t)w * a.maxL + il) * 9;
        Hl[0] = hl[0]; Hl[1] = hl[1]; Hl[2] = hl[2]; Hl[3] = hl[1]; Hl[4] = hl[3]; Hl[5] = hl[4]; Hl[6] = hl[2]; Hl[7] = hl[4]; Hl[8] = hl[5];
        double* bl = a.bl + ((size_t)w * a.maxL + il) * 3;
        bl[0] = hl[6]; bl[1] = hl[7]; bl[2] = hl[8];
    };
    for (int k = t; k < E; k += NT) {                          // edges outside single-run landmarks: one per thread
        const int il = el[k];
        if (il < 0 || il >= L) {                                // malformed edge: contributes nothing
            a.chi2[(size_t)w * a.maxE + k] = 0.0;
            double* hpl = a.Hpl + ((size_t)w * a.maxE + k) * 18;
#pragma unroll
            for (int i = 0; i < 18; i++) hpl[i] = 0.0;
        } else if (s_runs[il] != 1) edge(k, il, nullptr);       // scattered landmark: its block is summed below
    }
    // single-run landmarks: a lane PAIR per landmark, each lane walks one half of the run (every lane of the wave has work; a thread
    // per run start would leave nine lanes in ten idle); the two halves are added first + second
    for (int i0 = 0; i0 < 2 * L; i0 += NT) {                   // uniform trip count: the pair exchange needs both lanes
        const int idx = i0 + t, il = idx >> 1, half = idx & 1;
        double hl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool own = il < L && s_runs[il] == 1;
        if (own) {
            const int k0 = s_start[il], n = s_len[il], n0 = (n + 1) >> 1;
            const int kb = half ? k0 + n0 : k0, ke = half ? k0 + n : k0 + n0;
            for (int kk = kb; kk < ke; kk++) edge(kk, il, hl);
        }
#pragma unroll
        for (int u = 0; u < 9; u++) hl[u] += __shfl_down(hl[u], 1, 64);       // even lane: first half + second half
        if (own && half == 0) lm_store(il, hl);
    }
    for (int il = t; il < L; il += NT) {                       // landmarks without edges (zeros) or with scattered edges (full scan, edge order)
        const int nr = s_runs[il];
        if (nr == 1) continue;
        double hl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (nr > 1 && !(fixed && fixed[il]))
            for (int kk = 0; kk < E; kk++) {
                const int ip = ep[kk];
                if (el[kk] != il || ip < 0 || ip >= P) continue;
                double e0, e1, J[12], Jp[6];
                ba_edge(sR + 12 * ip, pts + 3 * il, obs + 2 * kk, a.fx, a.fy, a.cx, a.cy, e0, e1, J, Jp);
                const double e2 = e0 * e0 + e1 * e1;
                const double wgt = (e2 <= d2) ? 1.0 : a.delta / sqrt(e2);
                int v = 0;
#pragma unroll
                for (int r = 0; r < 3; r++) {
#pragma unroll
                    for (int c = r; c < 3; c++) hl[v++] += wgt * (Jp[r] * Jp[c] + Jp[3 + r] * Jp[3 + c]);
                }
#pragma unroll
                for (int r = 0; r < 3; r++) hl[6 + r] += -wgt * (Jp[r] * e0 + Jp[3 + r] * e1);
            }
        lm_store(il, hl);
    }
    if (LISTS) {
        // pose blocks: a wave per pose walks the pose's edge list; nothing here depends on the passes above except the list
        const int lane = t & 63;
        for (int p = wv; p < P; p += NW) {
            const int n = s_ptot[p], o = s_poff[p];
            double h[32];
#pragma unroll
            for (int u = 0; u < 32; u++) h[u] = 0.0;
            for (int i = lane; i < n; i += 64) {
                const int k = s_list[o + i];
                double e0, e1, J[12], Jp[6];
                ba_edge(sR + 12 * p, pts + 3 * el[k], obs + 2 * k, a.fx, a.fy, a.cx, a.cy, e0, e1, J, Jp);
                const double e2 = e0 * e0 + e1 * e1;
                const double wgt = (e2 <= d2) ? 1.0 : a.delta / sqrt(e2);   // Huber rho'
                int u = 0;
#pragma unroll
                for (int r = 0; r < 6; r++) {
#pragma unroll
                    for (int c = r; c < 6; c++) h[u++] += wgt * (J[r] * J[c] + J[6 + r] * J[6 + c]);
                }
#pragma unroll
                for (int r = 0; r < 6; r++) h[21 + r] += -wgt * (J[r] * e0 + J[6 + r] * e1);
            }
            wave_sum32_halving(h, lane);                        // lane l: the total of value l >> 1
            const int u = lane >> 1;
            if (!(lane & 1) && u < 27) {
                if (u < 21) {
                    const int r = (u >= 6) + (u >= 11) + (u >= 15) + (u >= 18) + (u >= 20);
                    const int c = r + u - (r * 6 - r * (r - 1) / 2);
                    double* H = a.Hpp + ((size_t)w * a.maxP + p) * 36;
                    H[r * 6 + c] = h[0]; H[c * 6 + r] = h[0];
                } else a.bp[((size_t)w * a.maxP + p) * 6 + (u - 21)] = h[0];
            }
        }
        return;
    }
    __syncthreads();
    // pose blocks: the wave copies in wave order
    for (int i = t; i < P * 36; i += NT) {
        const int p = i / 36, r = (i % 36) / 6, c = i % 6;
        const int rr = min(r, c), cc = max(r, c);
        const int u = rr * 6 - rr * (rr - 1) / 2 + (cc - rr);       // index in the row-major upper triangle
        const size_t o = (size_t)27 * p + u, ws = (size_t)a.maxP * 27;
        double acc = sAcc[o];
#pragma unroll
        for (int q = 1; q < NW; q++) acc += sAcc[q * ws + o];
        a.Hpp[((size_t)w * a.maxP + p) * 36 + r * 6 + c] = acc;
    }
    for (int i = t; i < P * 6; i += NT) {
        const int p = i / 6, r = i % 6;
        const size_t o = (size_t)27 * p + 21 + r, ws = (size_t)a.maxP * 27;
        double acc = sAcc[o];
#pragma unroll
        for (int q = 1; q < NW; q++) acc += sAcc[q * ws + o];
        a.bp[(size_t)w * a.maxP * 6 + i] = acc;
    }
}

constexpr int BA_NT = 512;             // threads per window
constexpr int BA_NW = BA_NT / 64;
constexpr int BA_CL = 32;              // landmarks per Schur chunk
constexpr int BA_VS = 3 * BA_CL + 2;   // row stride of the staged V chunk (doubles): 2 mod 32 -> conflict-free MFMA operand reads

template <int NW>
__device__ __forceinline__ double block_sum_n(double v, double* s_red) {
    v = wave_reduce_sum(v);
    if (NW == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) s += s_red[i];
    return s;
}

__device__ __forceinline__ double block_sum(double v, double* s_red) {
    v = wave_reduce_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int i = 0; i < BA_NW; i++) s += s_red[i];
    return s;
}

// Sophus SE3d::exp(d) * T, d = (upsilon, omega); R|t stored as 12 doubles
__device__ void pose_oplus(double* T, const double* d) {
    const double wx = d[3], wy = d[4], wz = d[5];
    const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
    double A, B, C;
    if (th < 1e-8) { A = 1 - th2 / 6; B = 0.5 - th2 / 24; C = 1.0 / 6 - th2 / 120; }
    else { A = sin(th) / th; B = (1 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double Wm[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9], Rd[9], V[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) W2[i * 3 + j] = Wm[i * 3] * Wm[j] + Wm[i * 3 + 1] * Wm[3 + j] + Wm[i * 3 + 2] * Wm[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        Rd[i] = I + A * Wm[i] + B * W2[i];
        V[i] = I + B * Wm[i] + C * W2[i];
    }
    double td[3], N[12];
#pragma unroll
    for (int i = 0; i < 3; i++) td[i] = V[i * 3] * d[0] + V[i * 3 + 1] * d[1] + V[i * 3 + 2] * d[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) N[i * 3 + j] = Rd[i * 3] * T[j] + Rd[i * 3 + 1] * T[3 + j] + Rd[i * 3 + 2] * T[6 + j];
        N[9 + i] = Rd[i * 3] * T[9] + Rd[i * 3 + 1] * T[10] + Rd[i * 3 + 2] * T[11] + td[i];
    }
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = N[i];
}

// Work decomposition (one 512-thread block per window, nothing but the per-pose edge index leaves LDS):
//   build     pose pass: one wave per pose walks that pose's edge list (stable, built once by ballot scans), 27 register
//             accumulators (Hpp upper triangle + bp), wave reduction, no atomics;
//             landmark pass: one thread per landmark walks its edge group -> Hll, bl, robust chi2.
//   Schur     per trial: G_l with (Hll+lambda I)^-1 = G G^T (G = L^-T of the 3x3 Cholesky); landmarks are processed in
//             chunks of 32: V_{l,p} = W_{l,p} G_l (6x3, W recomputed from the edge) is staged in LDS, then thread
//             (pose pair p1>=p2, 4-landmark slice) accumulates the 6x6 block sum V_{l,p1} V_{l,p2}^T in registers over ALL
//             chunks and adds it to S once at the end (8 partial sums per entry).
//   Cholesky  6x6-blocked right-looking factorisation of the 6P x 6P lower triangle (3 barriers per block column).
//   solve     one wave, lane = row, shuffles; landmark back-substitution one thread per landmark.

// GL = false: the whole state lives in LDS (windows of up to ~700 landmarks at 7 key-frames).  GL = true: the per-landmark arrays
// (points, backups, Hll, bl, G, edge ranges: 22 doubles per landmark) live in the window's HBM scratch behind the pose-sorted edge list
// instead — slower, but a window of the reference's size (7 key-frames x ~150 new features each, backend.cpp:134-135) still fits.
template <bool GL>
__global__ __launch_bounds__(BA_NT) void k_ba_optimize(BaOptArgs a) {
    extern __shared__ __attribute__((aligned(16))) double s_d[];
    __shared__ double s_red[BA_NW];
    __shared__ double s_sc[8];      // [0] lambda [1] ni [2] curChi [5] ok
    __shared__ int s_poff[16];
    __shared__ int s_bad, s_dup;
    const int w = blockIdx.x, t = threadIdx.x, wv = t >> 6, lane = t & 63;
    const int P = a.sizes ? a.sizes[3 * w] : a.nposes;
    const int L = a.sizes ? a.sizes[3 * w + 1] : a.npts;
    const int E = a.sizes ? a.sizes[3 * w + 2] : a.nedges;
    const int n = 6 * P;
    double* sR = s_d;                         // maxP x 12
    double* sRb = sR + a.maxP * 12;
    double* sHpp = sRb + a.maxP * 12;         // maxP x 21
    double* sbp = sHpp + a.maxP * 21;         // maxP x 6
    double* const lmBase = GL ? a.W + (size_t)w * a.wstride + ((size_t)a.maxE + 1) / 2 + 8 : sbp + a.maxP * 6;      // behind plist / in LDS
    double* sPt = lmBase;                     // maxL x 3
    double* sPtb = sPt + a.maxL * 3;
    double* sHll = sPtb + a.maxL * 3;         // maxL x 6
    double* sbl = sHll + a.maxL * 6;          // maxL x 3
    double* sG = sbl + a.maxL * 3;            // maxL x 6   upper-triangular G (g00 g01 g02 g11 g12 g22)
    double* sS = GL ? sbp + a.maxP * 6 : sG + a.maxL * 6;             // (6 maxP)^2
    double* srhs = sS + 36 * a.maxP * a.maxP; // 6 maxP
    double* sLinv = srhs + 6 * a.maxP;        // maxP x 21  inverses of the diagonal blocks of chol(S)
    double* sy = sLinv + 21 * a.maxP;         // 3 BA_CL    G^T bl of the chunk's landmarks
    double* sV = sy + 3 * BA_CL;              // vrows x BA_VS: V of the chunk, row = 6 pose + r, column = 3 landmark + c
    const int vrows = 16 * ((6 * a.maxP + 15) / 16);
    int* lbeg = reinterpret_cast<int*>(GL ? sG + a.maxL * 6 : sV + vrows * BA_VS);   // maxL
    int* lend = lbeg + a.maxL;
    double* poses = a.poses + (size_t)w * a.maxP * 7;
    double* pts = a.points + (size_t)w * a.maxL * 3;
    const int32_t* ep = a.ep + (size_t)w * a.maxE;
    const int32_t* el = a.el + (size_t)w * a.maxE;
    const double* obs = a.obs + (size_t)w * a.maxE * 2;
    const uint8_t* fixed = a.fixed ? a.fixed + (size_t)w * a.maxL : nullptr;
    int* plist = reinterpret_cast<int*>(a.W + (size_t)w * a.wstride);       // edges sorted by pose (stable)
    const double d2 = a.delta * a.delta;

    // ---- load state, landmark -> edge range ----
    for (int p = t; p < P; p += BA_NT) {
        double x = poses[7 * p], y = poses[7 * p + 1], z = poses[7 * p + 2], q = poses[7 * p + 3];
        const double nn = sqrt(x * x + y * y + z * z + q * q);
        x /= nn; y /= nn; z /= nn; q /= nn;
        double* R = sR + 12 * p;
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * q);     R[2] = 2 * (x * z + y * q);
        R[3] = 2 * (x * y + z * q);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * q);
        R[6] = 2 * (x * z - y * q);     R[7] = 2 * (y * z + x * q);     R[8] = 1 - 2 * (x * x + y * y);
        R[9] = poses[7 * p + 4]; R[10] = poses[7 * p + 5]; R[11] = poses[7 * p + 6];
    }
    for (int i = t; i < 3 * L; i += BA_NT) sPt[i] = pts[i];
    for (int l = t; l < L; l += BA_NT) { lbeg[l] = 0; lend[l] = 0; }
    if (t == 0) { s_bad = 0; s_dup = 0; }
    __syncthreads();
    for (int k = t; k < E; k += BA_NT) {
        const int l = el[k], p = ep[k];
        if (l < 0 || l >= L || p < 0 || p >= P) { atomicAdd(&s_bad, 1); continue; }
        if (k == 0 || el[k - 1] != l) { lbeg[l] = k; atomicAdd(&lend[l], 1); }      // first edge of a group
    }
    __syncthreads();
    // lend[l] now counts the groups of landmark l: more than one group = edges not grouped by landmark
    for (int l = t; l < L; l += BA_NT) if (lend[l] > 1) atomicAdd(&s_bad, 1);
    __syncthreads();
    if (s_bad) { if (t == 0) { a.status[w] = MYSLAM_ERR_INVALID; if (a.iters) a.iters[w] = 0; if (a.final_chi2) a.final_chi2[w] = 0; } return; }
    for (int l = t; l < L; l += BA_NT) {
        if (lend[l] == 0) { lbeg[l] = 0; continue; }
        int k = lbeg[l];
        unsigned seen = 0;
        while (k < E && el[k] == l) {
            const unsigned bit = 1u << ep[k];
            if (seen & bit) s_dup = 1;               // two edges between the same (pose, landmark): staged V needs atomics
            seen |= bit;
            k++;
        }
        lend[l] = k;
    }
    // ---- per-pose edge lists, in edge order (ballot scan: deterministic) ----
    for (int p = wv; p < P; p += BA_NW) {
        int cnt = 0;
        for (int k0 = 0; k0 < E; k0 += 64) {
            const int k = k0 + lane;
            cnt += __popcll(__ballot(k < E && ep[k] == p));
        }
        if (lane == 0) s_poff[p + 1] = cnt;
    }
    __syncthreads();
    if (t == 0) { s_poff[0] = 0; for (int p = 0; p < P; p++) s_poff[p + 1] += s_poff[p]; }
    __syncthreads();
    for (int p = wv; p < P; p += BA_NW) {
        int base = s_poff[p];
        for (int k0 = 0; k0 < E; k0 += 64) {
            const int k = k0 + lane;
            const bool m = k < E && ep[k] == p;
            const unsigned long long bm = __ballot(m);
            if (m) plist[base + __popcll(bm & ((1ull << lane) - 1ull))] = k;
            base += __popcll(bm);
        }
    }
    __syncthreads();

    double* echi = a.edge_chi2 ? a.edge_chi2 + (size_t)w * a.maxE : nullptr;      // what edge->chi2() returns: e^T e of the last evaluation
    auto robust_chi2 = [&](bool record) -> double {          // computeActiveErrors() + activeRobustChi2()
        // (every edge loop of this kernel fetches the NEXT edge's indices and observation before it evaluates the current one: a block has two
        // waves per SIMD, so nothing else hides the L2 latency of these loads behind the ~250 f64 instructions of an evaluation — round 5)
        double acc = 0;
        int pn = 0, ln = 0; double zn0 = 0, zn1 = 0;
        if (t < E) { pn = ep[t]; ln = el[t]; zn0 = obs[2 * t]; zn1 = obs[2 * t + 1]; }
        for (int k = t; k < E; k += BA_NT) {
            const int pk = pn, lk = ln; const double z[2] = {zn0, zn1};
            if (k + BA_NT < E) { pn = ep[k + BA_NT]; ln = el[k + BA_NT]; zn0 = obs[2 * (k + BA_NT)]; zn1 = obs[2 * (k + BA_NT) + 1]; }
            double e0, e1;
            ba_edge(sR + 12 * pk, sPt + 3 * lk, z, a.fx, a.fy, a.cx, a.cy, e0, e1, nullptr, nullptr);
            const double e2 = e0 * e0 + e1 * e1;
            if (record && echi) echi[k] = e2;
            acc += (e2 <= d2) ? e2 : 2 * sqrt(e2) * a.delta - d2;
        }
        return block_sum(acc, s_red);
    };

    const int nb = (n + 15) / 16;                       // 16-row blocks of the reduced system

    int it = 0, rnd = 0, cntOut = 0;
  for (;;) {                                   // rounds of { initializeOptimization(); optimize(max_iters) }, backend.cpp:212-232
    it = 0;
    for (; it < a.max_iters; it++) {
        // ---- build H, b at the current estimate ----
        for (int p = wv; p < P; p += BA_NW) {
            double h[27];
#pragma unroll
            for (int u = 0; u < 27; u++) h[u] = 0.0;
            const int iend = s_poff[p + 1];
            int kn = 0, ln = 0; double zn0 = 0, zn1 = 0;
            if (s_poff[p] + lane < iend) { kn = plist[s_poff[p] + lane]; ln = el[kn]; zn0 = obs[2 * kn]; zn1 = obs[2 * kn + 1]; }
            for (int i = s_poff[p] + lane; i < iend; i += 64) {
                const int lk = ln; const double z[2] = {zn0, zn1};
                if (i + 64 < iend) { kn = plist[i + 64]; ln = el[kn]; zn0 = obs[2 * kn]; zn1 = obs[2 * kn + 1]; }
                double e0, e1, J[12], Jp[6];
                ba_edge(sR + 12 * p, sPt + 3 * lk, z, a.fx, a.fy, a.cx, a.cy, e0, e1, J, Jp);
                const double e2 = e0 * e0 + e1 * e1;
                const double wgt = (e2 <= d2) ? 1.0 : a.delta / sqrt(e2);
                int u = 0;
#pragma unroll
                for (int r = 0; r < 6; r++) {
#pragma unroll
                    for (int c = r; c < 6; c++) h[u++] += wgt * (J[r] * J[c] + J[6 + r] * J[6 + c]);
                }
#pragma unroll
                for (int r = 0; r < 6; r++) h[21 + r] += -wgt * (J[r] * e0 + J[6 + r] * e1);
            }
#pragma unroll
            for (int u = 0; u < 27; u++) h[u] = wave_sum_lane63(h[u]);
            if (lane == 63) {
#pragma unroll
                for (int u = 0; u < 21; u++) sHpp[21 * p + u] = h[u];
#pragma unroll
                for (int r = 0; r < 6; r++) sbp[6 * p + r] = h[21 + r];
            }
        }
        double acc = 0;
        for (int l = t; l < L; l += BA_NT) {
            double hl[9];
#pragma unroll
            for (int u = 0; u < 9; u++) hl[u] = 0.0;
            const bool fx_pt = fixed && fixed[l];
            const int kb = lbeg[l], ke = lend[l];
            int pn = 0; double zn0 = 0, zn1 = 0;
            if (kb < ke) { pn = ep[kb]; zn0 = obs[2 * kb]; zn1 = obs[2 * kb + 1]; }
            for (int k = kb; k < ke; k++) {
                const int pk = pn; const double z[2] = {zn0, zn1};
                if (k + 1 < ke) { pn = ep[k + 1]; zn0 = obs[2 * k + 2]; zn1 = obs[2 * k + 3]; }
                double e0, e1, J[12], Jp[6];
                ba_edge(sR + 12 * pk, sPt + 3 * l, z, a.fx, a.fy, a.cx, a.cy, e0, e1, J, Jp);
                const double e2 = e0 * e0 + e1 * e1;
                const double wgt = (e2 <= d2) ? 1.0 : a.delta / sqrt(e2);
                acc += (e2 <= d2) ? e2 : 2 * sqrt(e2) * a.delta - d2;
                if (echi) echi[k] = e2;
                int v = 0;
#pragma unroll
                for (int r = 0; r < 3; r++) {
#pragma unroll
                    for (int c = r; c < 3; c++) hl[v++] += wgt * (Jp[r] * Jp[c] + Jp[3 + r] * Jp[3 + c]);
                }
#pragma unroll
                for (int r = 0; r < 3; r++) hl[6 + r] += -wgt * (Jp[r] * e0 + Jp[3 + r] * e1);
            }
#pragma unroll
            for (int u = 0; u < 6; u++) sHll[6 * l + u] = fx_pt ? 0.0 : hl[u];
#pragma unroll
            for (int r = 0; r < 3; r++) sbl[3 * l + r] = fx_pt ? 0.0 : hl[6 + r];
        }
        const double curChi0 = block_sum(acc, s_red);
        if (it == 0) {                               // computeLambdaInit: tau * max |diag(H)|
            double mx = 0;
            for (int i = t; i < P * 6; i += BA_NT) { const int p = i / 6, r = i % 6; mx = fmax(mx, fabs(sHpp[21 * p + r * 6 - r * (r - 1) / 2])); }
            for (int i = t; i < L * 3; i += BA_NT) { const int l = i / 3, r = i % 3; if (!(fixed && fixed[l])) mx = fmax(mx, fabs(sHll[6 * l + r * 3 - r * (r - 1) / 2])); }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
            __syncthreads();
            if (lane == 0) s_red[wv] = mx;
            __syncthreads();
            if (t == 0) {
                double m = 0;
                for (int i = 0; i < BA_NW; i++) m = fmax(m, s_red[i]);
                s_sc[0] = 1e-5 * m; s_sc[1] = 2.0;
            }
        }
        if (t == 0) s_sc[2] = curChi0;
        __syncthreads();

        int qmax = 0;
        double rho = 0;
        do {
            const double lambda = s_sc[0];
            // backup (optimizer->push())
            for (int i = t; i < P * 12; i += BA_NT) sRb[i] = sR[i];
            for (int i = t; i < L * 3; i += BA_NT) sPtb[i] = sPt[i];
            // reduced system: S = Hpp + lambda I (lower triangle), rhs = bp
            for (int i = t; i < n * n; i += BA_NT) sS[i] = 0.0;
            if (t == 0) s_sc[5] = 1.0;
            __syncthreads();
            for (int i = t; i < P * 36; i += BA_NT) {
                const int p = i / 36, r = (i % 36) / 6, c = i % 6;
                const int rr = min(r, c), cc = max(r, c);
                sS[(6 * p + r) * n + 6 * p + c] = sHpp[21 * p + rr * 6 - rr * (rr - 1) / 2 + (cc - rr)] + (r == c ? lambda : 0.0);
            }
            for (int i = t; i < n; i += BA_NT) srhs[i] = sbp[i];
            // landmarks: (Hll + lambda I) = Lc Lc^T, G = Lc^-T (upper), so that (Hll + lambda I)^-1 = G G^T
            for (int l = t; l < L; l += BA_NT) {
                double* g = sG + 6 * l;
                if ((fixed && fixed[l]) || lend[l] <= lbeg[l]) { for (int i = 0; i < 6; i++) g[i] = 0.0; continue; }
                const double* h = sHll + 6 * l;
                const double a00 = h[0] + lambda, a01 = h[1], a02 = h[2], a11 = h[3] + lambda, a12 = h[4], a22 = h[5] + lambda;
                const double l00 = sqrt(a00), l10 = a01 / l00, l20 = a02 / l00;
                const double d11 = a11 - l10 * l10, l11 = sqrt(d11), l21 = (a12 - l20 * l10) / l11;
                const double d22 = a22 - l20 * l20 - l21 * l21, l22 = sqrt(d22);
                if (!(a00 > 0.0) || !(d11 > 0.0) || !(d22 > 0.0) || !isfinite(l22)) { s_sc[5] = 0.0; for (int i = 0; i < 6; i++) g[i] = 0.0; continue; }
                // M = Lc^-1 (lower): m00 = 1/l00, m11 = 1/l11, m22 = 1/l22, m10 = -l10 m00 m11, m21 = -l21 m11 m22,
                // m20 = -(l20 m00 + l21 m10) m22;  G = M^T
                const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;
                const double m10 = -l10 * m00 * m11, m21 = -l21 * m11 * m22, m20 = -(l20 * m00 + l21 * m10) * m22;
                g[0] = m00; g[1] = m10; g[2] = m20; g[3] = m11; g[4] = m21; g[5] = m22;
            }
            __syncthreads();
            // ---- Schur complement S -= V V^T, rhs -= V y on the f64 matrix cores, chunk by chunk ----
            // 16x16 tiles of the lower triangle, q -> (ti,tj): 0:(0,0) 1:(1,0) 2:(1,1) 3:(2,0) 4:(2,1) 5:(2,2) 6:(3,0) 7:(3,1) 8:(3,2) 9:(3,3).
            // wave wv owns tile wv for all k; tiles 8 and 9 (only when nb == 4) are split over k between the 8 waves, which
            // balances the four SIMDs (60 MFMAs per chunk each).  rhs rows ride along in the waves whose tile is (i,0)/(1,1).
            const int ntile = nb * (nb + 1) / 2;
            int ti = 0, tj = 0;
            { int q = wv; while (q >= ti + 1) { q -= ti + 1; ti++; } tj = q; }      // tile wv -> (ti, tj)
            const bool own = wv < ntile;
            ba_d4 acc = ba_d4{0.0, 0.0, 0.0, 0.0}, acc8 = ba_d4{0.0, 0.0, 0.0, 0.0}, acc9 = ba_d4{0.0, 0.0, 0.0, 0.0};
            double rpart = 0.0;
            const bool do_rhs = own && tj == 0;                                     // q = 0, 1, 3, 6 <-> row blocks 0..3
            for (int l0 = 0; l0 < L; l0 += BA_CL) {
                const int nl = min(BA_CL, L - l0);
                for (int i = t; i < 16 * nb * BA_VS; i += BA_NT) sV[i] = 0.0;
                __syncthreads();
                // stage V_{l,p} = sum over the (l,p) edges of W_k G_l, W_k = w J^T Jp;  y_l = G_l^T bl_l
                {
                    const int ll = t >> 4, l = l0 + ll;
                    const bool live = ll < nl && !(fixed && fixed[l]);
                    const double* g = sG + 6 * (live ? l : 0);
                    const double g00 = live ? g[0] : 0.0, g01 = live ? g[1] : 0.0, g02 = live ? g[2] : 0.0;
                    const double g11 = live ? g[3] : 0.0, g12 = live ? g[4] : 0.0, g22 = live ? g[5] : 0.0;
                    if ((t & 15) == 0) {
                        const double b0 = live ? sbl[3 * l] : 0.0, b1 = live ? sbl[3 * l + 1] : 0.0, b2 = live ? sbl[3 * l + 2] : 0.0;
                        sy[3 * ll] = g00 * b0; sy[3 * ll + 1] = g01 * b0 + g11 * b1; sy[3 * ll + 2] = g02 * b0 + g12 * b1 + g22 * b2;
                    }
                    if (live) {
                        for (int k = lbeg[l] + (t & 15); k < lend[l]; k += 16) {
                            const int p = ep[k];
                            double e0, e1, J[12], Jp[6];
                            ba_edge(sR + 12 * p, sPt + 3 * l, obs + 2 * k, a.fx, a.fy, a.cx, a.cy, e0, e1, J, Jp);
                            const double e2 = e0 * e0 + e1 * e1;
                            const double wgt = (e2 <= d2) ? 1.0 : a.delta / sqrt(e2);
                            double* v = sV + (size_t)(6 * p) * BA_VS + 3 * ll;
                            const double q0 = wgt * (Jp[0] * g00), q1 = wgt * (Jp[0] * g01 + Jp[1] * g11), q2 = wgt * (Jp[0] * g02 + Jp[1] * g12 + Jp[2] * g22);
                            const double q3 = wgt * (Jp[3] * g00), q4 = wgt * (Jp[3] * g01 + Jp[4] * g11), q5 = wgt * (Jp[3] * g02 + Jp[4] * g12 + Jp[5] * g22);
                            if (!s_dup) {
#pragma unroll
                                for (int r = 0; r < 6; r++) {                   // V = J^T (w Jp G)
                                    v[r * BA_VS] = J[r] * q0 + J[6 + r] * q3;
                                    v[r * BA_VS + 1] = J[r] * q1 + J[6 + r] * q4;
                                    v[r * BA_VS + 2] = J[r] * q2 + J[6 + r] * q5;
                                }
                            } else {                                            // an (l,p) pair carries several edges: accumulate
#pragma unroll
                                for (int r = 0; r < 6; r++) {
                                    atomicAdd(&v[r * BA_VS], J[r] * q0 + J[6 + r] * q3);
                                    atomicAdd(&v[r * BA_VS + 1], J[r] * q1 + J[6 + r] * q4);
                                    atomicAdd(&v[r * BA_VS + 2], J[r] * q2 + J[6 + r] * q5);
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                if (own) {
                    const double* pa = sV + (size_t)(16 * ti + (lane & 15)) * BA_VS + (lane >> 4);
                    const double* pb = sV + (size_t)(16 * tj + (lane & 15)) * BA_VS + (lane >> 4);
#pragma unroll 4
                    for (int ks = 0; ks < 3 * BA_CL / 4; ks++) {
                        const double av = pa[4 * ks], bv = pb[4 * ks];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                        if (do_rhs) rpart += av * sy[4 * ks + (lane >> 4)];
                    }
                }
                if (ntile == 10) {
                    const double* p2 = sV + (size_t)(32 + (lane & 15)) * BA_VS + (lane >> 4);
                    const double* p3 = sV + (size_t)(48 + (lane & 15)) * BA_VS + (lane >> 4);
#pragma unroll
                    for (int kk = 0; kk < 3; kk++) {
                        const int ks = wv + BA_NW * kk;
                        const double a2 = p2[4 * ks], a3 = p3[4 * ks];
                        acc8 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, a2, acc8, 0, 0, 0);
                        acc9 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, a3, acc9, 0, 0, 0);
                    }
                }
                __syncthreads();
            }
            // flush: D row = (lane>>4) + 4 v, col = lane & 15.  Owned tiles subtract directly; the split tiles go through LDS.
            if (own) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int row = 16 * ti + (lane >> 4) + 4 * v, col = 16 * tj + (lane & 15);
                    if (row < n && col < n && col <= row) sS[row * n + col] -= acc[v];
                }
                if (do_rhs) {
                    rpart += __shfl_xor(rpart, 16, 64);
                    rpart += __shfl_xor(rpart, 32, 64);
                    const int row = 16 * ti + lane;
                    if (lane < 16 && row < n) srhs[row] -= rpart;
                }
            }
            if (ntile == 10) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    sV[(wv * 2) * 256 + v * 64 + lane] = acc8[v];
                    sV[(wv * 2 + 1) * 256 + v * 64 + lane] = acc9[v];
                }
                __syncthreads();
                {
                    const int tile = t >> 8, e = t & 255, v = e >> 6, ln = e & 63;      // 512 threads = 2 tiles x 256 entries
                    double sum = 0;
#pragma unroll
                    for (int u = 0; u < BA_NW; u++) sum += sV[(u * 2 + tile) * 256 + e];
                    const int row = 48 + (ln >> 4) + 4 * v, col = (tile ? 48 : 32) + (ln & 15);
                    if (row < n && col < n && col <= row) sS[row * n + col] -= sum;
                }
            }
            __syncthreads();
            // ---- blocked Cholesky of S (lower triangle), 6x6 blocks.  Every panel thread factors the diagonal block
            // redundantly in registers (same latency as one thread doing it, one barrier less per block column) ----
            for (int jb = 0; jb < P; jb++) {
                const int j0 = 6 * jb;
                const int m = n - j0 - 6;                          // rows below the diagonal block
                if (t <= m) {
                    double Ld[21], iv[6];
#pragma unroll
                    for (int r = 0; r < 6; r++)
#pragma unroll
                        for (int c = 0; c <= r; c++) Ld[r * (r + 1) / 2 + c] = sS[(j0 + r) * n + j0 + c];
                    double x[6];
                    if (t < m) {
#pragma unroll
                        for (int c = 0; c < 6; c++) x[c] = sS[(size_t)(j0 + 6 + t) * n + j0 + c];
                    }
                    bool good = true;
#pragma unroll
                    for (int c = 0; c < 6; c++) {                  // right-looking inside the block: short dependency chains
                        double d = Ld[c * (c + 1) / 2 + c];
                        if (!(d > 0.0)) { good = false; d = 1.0; }
                        const double inv = rsqrt(d);
                        iv[c] = inv;
                        Ld[c * (c + 1) / 2 + c] = d * inv;
#pragma unroll
                        for (int r = c + 1; r < 6; r++) Ld[r * (r + 1) / 2 + c] *= inv;
#pragma unroll
                        for (int r = c + 1; r < 6; r++)
#pragma unroll
                            for (int k = c + 1; k <= r; k++) Ld[r * (r + 1) / 2 + k] -= Ld[r * (r + 1) / 2 + c] * Ld[k * (k + 1) / 2 + c];
                    }
                    if (t < m) {                                   // panel row: x L_jj^T = a
#pragma unroll
                        for (int c = 0; c < 6; c++) {
                            double v = x[c];
#pragma unroll
                            for (int k = 0; k < c; k++) v -= x[k] * Ld[c * (c + 1) / 2 + k];
                            x[c] = v * iv[c];
                        }
#pragma unroll
                        for (int c = 0; c < 6; c++) sS[(size_t)(j0 + 6 + t) * n + j0 + c] = x[c];
                    } else {                                       // t == m: publishes the inverse of the diagonal block (lower), which is
                        if (!good) s_sc[5] = 0.0;                  // all the substitutions need of it (sS keeps the unfactored block)
                        double M[21];
#pragma unroll
                        for (int c = 0; c < 6; c++) {
                            M[c * (c + 1) / 2 + c] = iv[c];
#pragma unroll
                            for (int r = c + 1; r < 6; r++) {
                                double v = 0;
#pragma unroll
                                for (int k = c; k < r; k++) v += Ld[r * (r + 1) / 2 + k] * M[k * (k + 1) / 2 + c];
                                M[r * (r + 1) / 2 + c] = -v * iv[r];
                            }
                        }
#pragma unroll
                        for (int i = 0; i < 21; i++) sLinv[21 * jb + i] = M[i];
                    }
                }
                __syncthreads();
                for (int idx = t; idx < m * (m + 1) / 2; idx += BA_NT) {     // trailing update, lower triangle: idx = i (i+1)/2 + k
                    int i = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
                    if (i * (i + 1) / 2 > idx) i--;
                    if ((i + 1) * (i + 2) / 2 <= idx) i++;
                    const int k = idx - i * (i + 1) / 2;
                    const double* ri = sS + (size_t)(j0 + 6 + i) * n + j0;
                    const double* rk = sS + (size_t)(j0 + 6 + k) * n + j0;
                    double v = 0;
#pragma unroll
                    for (int c = 0; c < 6; c++) v += ri[c] * rk[c];
                    sS[(size_t)(j0 + 6 + i) * n + j0 + 6 + k] -= v;
                }
                __syncthreads();
            }
            // block forward / backward substitution by one wave (lane = row); values move over readlane
            if (t < 64) {
                double y = (t < n) ? srhs[t] : 0.0;
                const int tr = min(t, n - 1);
                for (int jb = 0; jb < P; jb++) {
                    const int j0 = 6 * jb;
                    const double* M = sLinv + 21 * jb;
                    double yb[6], xb[6];
#pragma unroll
                    for (int c = 0; c < 6; c++) yb[c] = bcast_lane(y, j0 + c);
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        double v = 0;
#pragma unroll
                        for (int c = 0; c <= r; c++) v += M[r * (r + 1) / 2 + c] * yb[c];
                        xb[r] = v;
                    }
                    double upd = 0;
#pragma unroll
                    for (int c = 0; c < 6; c++) upd += sS[tr * n + j0 + c] * xb[c];
                    if (t >= j0 + 6 && t < n) y -= upd;
#pragma unroll
                    for (int c = 0; c < 6; c++) if (t == j0 + c) y = xb[c];
                }
                for (int jb = P - 1; jb >= 0; jb--) {
                    const int j0 = 6 * jb;
                    const double* M = sLinv + 21 * jb;
                    double yb[6], xb[6];
#pragma unroll
                    for (int c = 0; c < 6; c++) yb[c] = bcast_lane(y, j0 + c);
#pragma unroll
                    for (int r = 0; r < 6; r++) {                  // x = M^T y
                        double v = 0;
#pragma unroll
                        for (int c = r; c < 6; c++) v += M[c * (c + 1) / 2 + r] * yb[c];
                        xb[r] = v;
                    }
                    double upd = 0;
#pragma unroll
                    for (int c = 0; c < 6; c++) upd += sS[(j0 + c) * n + tr] * xb[c];
                    if (t < j0) y -= upd;
#pragma unroll
                    for (int c = 0; c < 6; c++) if (t == j0 + c) y = xb[c];
                }
                if (t < n) srhs[t] = y;               // xp
            }
            __syncthreads();
            const bool ok = s_sc[5] != 0.0;
            // xl = Hinv (bl - W^T xp); scale = x^T (lambda x + b); oplus on the landmarks
            double sc = 0;
            for (int l = t; l < L; l += BA_NT) {
                const int kb = lbeg[l], ke = lend[l];
                if ((fixed && fixed[l]) || ke <= kb) continue;
                double r0 = sbl[3 * l], r1 = sbl[3 * l + 1], r2 = sbl[3 * l + 2];
                int pn = ep[kb]; double zn0 = obs[2 * kb], zn1 = obs[2 * kb + 1];
                for (int k = kb; k < ke; k++) {
                    const int p = pn; const double z[2] = {zn0, zn1};
                    if (k + 1 < ke) { pn = ep[k + 1]; zn0 = obs[2 * k + 2]; zn1 = obs[2 * k + 3]; }
                    double e0, e1, J[12], Jp[6];
                    ba_edge(sRb + 12 * p, sPtb + 3 * l, z, a.fx, a.fy, a.cx, a.cy, e0, e1, J, Jp);
                    const double e2 = e0 * e0 + e1 * e1;
                    const double wgt = (e2 <= d2) ? 1.0 : a.delta / sqrt(e2);
                    const double* xp = srhs + 6 * p;
                    double s0 = 0, s1 = 0;
#pragma unroll
                    for (int r = 0; r < 6; r++) { s0 += J[r] * xp[r]; s1 += J[6 + r] * xp[r]; }
                    r0 -= wgt * (Jp[0] * s0 + Jp[3] * s1); r1 -= wgt * (Jp[1] * s0 + Jp[4] * s1); r2 -= wgt * (Jp[2] * s0 + Jp[5] * s1);
                }
                const double* g = sG + 6 * l;
                const double y0 = g[0] * r0, y1 = g[1] * r0 + g[3] * r1, y2 = g[2] * r0 + g[4] * r1 + g[5] * r2;   // G^T r
                const double x0 = g[0] * y0 + g[1] * y1 + g[2] * y2, x1 = g[3] * y1 + g[4] * y2, x2 = g[5] * y2;    // G y
                sc += x0 * (lambda * x0 + sbl[3 * l]) + x1 * (lambda * x1 + sbl[3 * l + 1]) + x2 * (lambda * x2 + sbl[3 * l + 2]);
                if (ok) { sPt[3 * l] += x0; sPt[3 * l + 1] += x1; sPt[3 * l + 2] += x2; }
            }
            for (int i = t; i < n; i += BA_NT) sc += srhs[i] * (lambda * srhs[i] + sbp[i]);
            const double scale = block_sum(sc, s_red) + 1e-3;
            if (ok) for (int p = t; p < P; p += BA_NT) pose_oplus(sR + 12 * p, srhs + 6 * p);
            __syncthreads();
            const double tmpChi = ok ? robust_chi2(true) : 1e300;       // the edges keep this error even if the step is rejected
            rho = (s_sc[2] - tmpChi) / scale;
            __syncthreads();
            if (rho > 0 && isfinite(tmpChi) && ok) {
                if (t == 0) {
                    double alpha = 1. - pow(2 * rho - 1, 3);
                    alpha = fmin(alpha, 2. / 3.);
                    s_sc[0] = lambda * fmax(1. / 3., alpha); s_sc[1] = 2.0; s_sc[2] = tmpChi;
                }
            } else {
                if (t == 0) { s_sc[0] = lambda * s_sc[1]; s_sc[1] *= 2.0; }
                for (int i = t; i < P * 12; i += BA_NT) sR[i] = sRb[i];      // optimizer->pop()
                for (int i = t; i < L * 3; i += BA_NT) sPt[i] = sPtb[i];
            }
            __syncthreads();
            qmax++;
        } while (rho < 0 && qmax < 10 && isfinite(s_sc[0]));
        if (qmax == 10 || rho == 0 || !isfinite(s_sc[0])) { it++; break; }
    }
    if (!echi) break;
    {   // every thread reads back the entries it wrote itself (same k = t + j BA_NT mapping everywhere)
        double c = 0;
        for (int k = t; k < E; k += BA_NT) c += (echi[k] > a.chi2_th) ? 1.0 : 0.0;
        cntOut = (int)block_sum(c, s_red);
    }
    if ((double)(E - cntOut) / (double)E > 0.5) break;          // inlierRatio > 0.5, backend.cpp:225-227
    rnd++;
    if (rnd >= a.rounds) break;
    __syncthreads();
  }
    if (echi) {
        for (int k = t; k < E; k += BA_NT) a.outlier[(size_t)w * a.maxE + k] = echi[k] > a.chi2_th ? 1 : 0;
        if (t == 0) { a.rounds_out[w] = rnd; a.nout_out[w] = cntOut; }
    }
    // ---- write back: R -> quaternion, points, final chi2 ----
    const double fin = robust_chi2(false);
    for (int p = t; p < P; p += BA_NT) {
        const double* R = sR + 12 * p;
        const double tr = R[0] + R[4] + R[8];
        double x, y, z, q;
        if (tr > 0) { const double s = sqrt(tr + 1.0) * 2; q = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s; }
        else if (R[0] > R[4] && R[0] > R[8]) { const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; x = 0.25 * s; q = (R[7] - R[5]) / s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s; }
        else if (R[4] > R[8]) { const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; y = 0.25 * s; q = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; z = (R[5] + R[7]) / s; }
        else { const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; z = 0.25 * s; q = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; }
        poses[7 * p] = x; poses[7 * p + 1] = y; poses[7 * p + 2] = z; poses[7 * p + 3] = q;
        poses[7 * p + 4] = R[9]; poses[7 * p + 5] = R[10]; poses[7 * p + 6] = R[11];
    }
    for (int i = t; i < 3 * L; i += BA_NT) pts[i] = sPt[i];
    if (t == 0) { if (a.final_chi2) a.final_chi2[w] = fin; if (a.iters) a.iters[w] = it; a.status[w] = MYSLAM_OK; }
}

static size_t ba_opt_lds(int maxP, int maxL, bool gl) {
    const size_t vrows = 16 * (((size_t)6 * maxP + 15) / 16);
    const size_t lm = gl ? 0 : (size_t)maxL;           // per-landmark arrays in HBM scratch instead
    return sizeof(double) * ((size_t)maxP * (12 + 12 + 21 + 6) + lm * (3 + 3 + 6 + 3 + 6) + 36 * (size_t)maxP * maxP + (6 + 21) * (size_t)maxP +
                             3 * BA_CL + vrows * BA_VS) +
           sizeof(int) * (2 * lm);
}

static std::atomic<int> g_ba_landmarks_in_hbm{0};         // MYSLAM_BA_OPT_LANDMARKS_IN_HBM

// scratch doubles one window needs: the pose-sorted edge list, and in the HBM form the per-landmark arrays behind it
static size_t ba_opt_scratch_need(int maxL, int maxE, bool gl) { return ((size_t)maxE + 1) / 2 + 8 + (gl ? 22 * (size_t)maxL + 2 : 0); }

static int ba_opt_launch(const BaOptArgs& a_in, int nwin, hipStream_t s) {
    BaOptArgs a = a_in;
    if (!a.wstride) a.wstride = (size_t)a.maxE * 18;
    if (a.maxP > MYSLAM_BA_MAX_WINDOW_POSES) return MYSLAM_ERR_UNSUPPORTED;          // substitution runs on one wave: 6P <= 64; 55 pose pairs x 8 slices <= 512 threads
    size_t lds = ba_opt_lds(a.maxP, a.maxL, false);
    const bool gl = lds > 160 * 1024 - 512 || g_ba_landmarks_in_hbm.load(std::memory_order_relaxed) != 0;   // large window (or by option): per-landmark state goes to the HBM scratch
    if (gl) {
        lds = ba_opt_lds(a.maxP, a.maxL, true);
        // the scratch (wstride doubles per window) must hold the pose-sorted edge list + 22 doubles per landmark
        if (ba_opt_scratch_need(a.maxL, a.maxE, true) > a.wstride || lds > 160 * 1024 - 512) return MYSLAM_ERR_CAPACITY;
    }
    const void* fn = gl ? reinterpret_cast<const void*>(k_ba_optimize<true>) : reinterpret_cast<const void*>(k_ba_optimize<false>);
    // the limit belongs to the function (per device and process), not to a launch: always the device's whole LDS minus the kernel's static
    // part, so that concurrent callers with windows of different sizes cannot lower it under each other's launches
    MYSLAM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    ScopedProf sp(P_BA, s);
    if (gl) hipLaunchKernelGGL(k_ba_optimize<true>, dim3(nwin), dim3(BA_NT), lds, s, a);
    else hipLaunchKernelGGL(k_ba_optimize<false>, dim3(nwin), dim3(BA_NT), lds, s, a);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

// ------------------------------------------------------------------------------------------------
// Pose-only optimisation of the current frame: Frontend::EstimateCurrentPose, src/frontend.cpp:176-276 [SURVEY.md §8(f) rank 1].
// One VertexPose, one EdgeProjectionPoseOnly (g2o_types.h:62-100) per tracked feature with a map point, Huber(delta = 1),
// g2o Levenberg + dense 6x6 solve; `rounds` x { optimize(iters) over the level-0 edges; classify every edge by chi2() > chi2_th
// (excluded edges are re-evaluated at the new estimate first), exclude / re-admit }; the robust kernel goes after round rounds-2.
// One 512-thread block per frame; every thread owns edges t, t+512, ... (level, last chi2 and outlier flag live in registers).
// ------------------------------------------------------------------------------------------------
constexpr int PO_EPT = 8;                                  // edges per thread -> at most 4096 edges per frame

struct PoseOnlyArgs {
    double* poses; const double* pts3d; const double* obs; const int32_t* counts; int n_fixed, cap;
    double fx, fy, cx, cy, chi2_th; int rounds, iters, pre;
    uint8_t* outlier; int32_t* n_inliers; int32_t* status;
};

// NT = threads per frame (pose_only_launch): 64 / 128 for large batches of frames, 256 for the one-frame call of the tracker (150 - 400
// matches: four waves pass the ~45 dependent Levenberg steps of a frame faster than eight that meet at every barrier, and faster than one
// or two that walk several edges each), 512 for frames of thousands of matches.
// The edge -> thread map (edge t + k NT) and with it the summation order depend on NT: results of different NT agree to rounding, the
// parity bars against the oracle (1e-8 relative on the pose, identical flags) hold for each.
template <int NT>
__global__ __launch_bounds__(NT) void k_pose_only(PoseOnlyArgs a) {
    constexpr int NW = NT / 64;
    __shared__ double s_red[NW];
    __shared__ double sT[12], sTb[12], sH[27], sx[6], s_part[NW][27];
    __shared__ double s_sc[4];          // [0] lambda [1] ni [2] ok
    const int f = blockIdx.x, t = threadIdx.x, wv = t >> 6, lane = t & 63;
    const int n = a.counts ? a.counts[f] : a.n_fixed;
    const double* P3 = a.pts3d + (size_t)f * a.cap * 3;
    const double* Z2 = a.obs + (size_t)f * a.cap * 2;
    double* pose = a.poses + (size_t)f * 7;
    if (n > PO_EPT * NT) { if (t == 0) { a.status[f] = MYSLAM_ERR_CAPACITY; a.n_inliers[f] = 0; } return; }
    if (t == 0) {
        double x = pose[0], y = pose[1], z = pose[2], q = pose[3];
        const double nn = sqrt(x * x + y * y + z * z + q * q);
        x /= nn; y /= nn; z /= nn; q /= nn;
        sT[0] = 1 - 2 * (y * y + z * z); sT[1] = 2 * (x * y - z * q);     sT[2] = 2 * (x * z + y * q);
        sT[3] = 2 * (x * y + z * q);     sT[4] = 1 - 2 * (x * x + z * z); sT[5] = 2 * (y * z - x * q);
        sT[6] = 2 * (x * z - y * q);     sT[7] = 2 * (y * z + x * q);     sT[8] = 1 - 2 * (x * x + y * y);
        sT[9] = pose[4]; sT[10] = pose[5]; sT[11] = pose[6];
    }
    __syncthreads();
    unsigned level = 0, outl = 0;                          // bit k <-> edge t + k*NT
    double echi[PO_EPT];
#pragma unroll
    for (int k = 0; k < PO_EPT; k++) echi[k] = 0.0;
    bool robust = true;
    // e = z - (K (T p)) / (K (T p)).z  (g2o_types.h:71-75)
    auto edge_err = [&](int i, double& e0, double& e1, double* pc) {
        const double px = P3[3 * i], py = P3[3 * i + 1], pz = P3[3 * i + 2];
        pc[0] = sT[0] * px + sT[1] * py + sT[2] * pz + sT[9];
        pc[1] = sT[3] * px + sT[4] * py + sT[5] * pz + sT[10];
        pc[2] = sT[6] * px + sT[7] * py + sT[8] * pz + sT[11];
        const double u = a.fx * pc[0] + 0.0 * pc[1] + a.cx * pc[2], v = 0.0 * pc[0] + a.fy * pc[1] + a.cy * pc[2];
        e0 = Z2[2 * i] - u / pc[2]; e1 = Z2[2 * i + 1] - v / pc[2];
    };
    // One- and two-wave blocks (the tracker's one-frame call) keep every edge's last evaluation — residual and camera-frame point — in
    // registers: the H pass of an iteration always follows an evaluation at exactly its state (the round's first, or the accepted trial's), so
    // it reads them back instead of evaluating again (the same doubles; two f64 divisions per edge and iteration less on a chain of ~45
    // dependent steps).  Eight-wave blocks of large batches keep their registers for occupancy and evaluate again.
#ifdef MYSLAM_POSE_ONLY_NO_KEEP                              // A/B builds only
    constexpr bool KEEP = false;
#else
    constexpr bool KEEP = NT <= 128;
#endif
    constexpr int NK = KEEP ? PO_EPT : 1;
    double ke0[NK], ke1[NK], kx[NK], ky[NK], kz[NK];
    auto active_chi2 = [&]() -> double {                   // computeActiveErrors + activeRobustChi2
        double s = 0;
#pragma unroll
        for (int k = 0; k < PO_EPT; k++) {
            const int i = t + k * NT;
            if (i < n && !((level >> k) & 1)) {
                double e0, e1, pc[3];
                edge_err(i, e0, e1, pc);
                if constexpr (KEEP) { ke0[k] = e0; ke1[k] = e1; kx[k] = pc[0]; ky[k] = pc[1]; kz[k] = pc[2]; }
                const double e2 = e0 * e0 + e1 * e1;
                echi[k] = e2;
                s += (!robust || e2 <= 1.0) ? e2 : 2 * sqrt(e2) - 1.0;
            }
        }
        return block_sum_n<NW>(s, s_red);
    };
    int cntOut = 0;
    for (int round = -a.pre; round < a.rounds; round++) {       // round < 0: the unclassified optimize() of LoopClosing::OptimizeCurrentPose
        double na = 0;
#pragma unroll
        for (int k = 0; k < PO_EPT; k++) na += (t + k * NT < n && !((level >> k) & 1)) ? 1.0 : 0.0;
        const int nact = (int)block_sum_n<NW>(na, s_red);
        if (nact > 0) {
            // g2o evaluates the active errors at the start of every iteration (computeActiveErrors + activeRobustChi2).  A new iteration only
            // follows an ACCEPTED trial, whose evaluation was made at exactly this state by exactly these operations: its chi2 is carried over
            // instead of being recomputed (the same double).  Only a round's first iteration evaluates — with the same function, so that every
            // sum keeps the order the oracle's has (accept / reject decisions of nearly converged iterations hang on the last bits; a first
            // attempt that let this evaluation ride in the H pass below, with another summation order, left the 1e-6 lock-step bar on one frame
            // in 200).  Round 5: one edge evaluation per iteration + one per trial instead of two + one — the tracker's one-frame call spends
            // its time in ~45 dependent Levenberg steps (frontend.cpp:176-276).
            double currentChi = active_chi2();
            bool fresh = true;                                // the kept evaluations (and currentChi) belong to the CURRENT state (block-uniform)
            for (int it = 0; it < a.iters; it++) {
                // a new iteration normally follows an accepted trial; the one exception (a NaN gain ratio leaves the trial loop with the state
                // restored and no exit condition met) evaluates again, as g2o does at the start of every iteration
                if (!fresh) { currentChi = active_chi2(); fresh = true; }
#ifdef MYSLAM_POSE_ONLY_RECOMPUTE_CHI                         // A/B builds only (tools/build_variants.sh): the round-4 form, one more evaluation per iteration
                if (it > 0) currentChi = active_chi2();
#endif
                // ---- H (upper triangle, 21) and b (6) ----
                double h[32];
#pragma unroll
                for (int u = 0; u < 32; u++) h[u] = 0.0;
#pragma unroll
                for (int k = 0; k < PO_EPT; k++) {
                    const int i = t + k * NT;
                    if (i < n && !((level >> k) & 1)) {
                        double e0, e1, pc[3];
                        if constexpr (KEEP) { e0 = ke0[k]; e1 = ke1[k]; pc[0] = kx[k]; pc[1] = ky[k]; pc[2] = kz[k]; }
                        else edge_err(i, e0, e1, pc);
                        const double X = pc[0], Y = pc[1], Zc = pc[2], Zinv = 1.0 / (Zc + 1e-18), Zinv2 = Zinv * Zinv;      // g2o_types.h:79-92
                        const double J[12] = {-a.fx * Zinv, 0, a.fx * X * Zinv2, a.fx * X * Y * Zinv2, -a.fx - a.fx * X * X * Zinv2, a.fx * Y * Zinv,
                                              0, -a.fy * Zinv, a.fy * Y * Zinv2, a.fy + a.fy * Y * Y * Zinv2, -a.fy * X * Y * Zinv2, -a.fy * X * Zinv};
                        const double e2 = e0 * e0 + e1 * e1;
                        const double w = (!robust || e2 <= 1.0) ? 1.0 : 1.0 / sqrt(e2);
                        int u = 0;
#pragma unroll
                        for (int r = 0; r < 6; r++) {
#pragma unroll
                            for (int c = r; c < 6; c++) h[u++] += w * (J[r] * J[c] + J[6 + r] * J[6 + c]);
                        }
#pragma unroll
                        for (int r = 0; r < 6; r++) h[21 + r] += -w * (J[r] * e0 + J[6 + r] * e1);
                    }
                }
#ifdef MYSLAM_POSE_ONLY_SUM_PER_VALUE                        // A/B builds only: 27 six-step wave sums (the form up to round 5)
#pragma unroll
                for (int u = 0; u < 27; u++) h[u] = wave_sum_lane63(h[u]);
                if (lane == 63) {
#pragma unroll
                    for (int u = 0; u < 27; u++) s_part[wv][u] = h[u];
                }
#else
                wave_sum32_halving(h, lane);
                if (!(lane & 1) && (lane >> 1) < 27) s_part[wv][lane >> 1] = h[0];
#endif
                __syncthreads();
                if (t < 27) { double s = 0; for (int w2 = 0; w2 < NW; w2++) s += s_part[w2][t]; sH[t] = s; }
                __syncthreads();
                if (it == 0 && t == 0) {
                    double mx = 0;
                    for (int r = 0; r < 6; r++) mx = fmax(mx, fabs(sH[r * 6 - r * (r - 1) / 2]));
                    s_sc[0] = 1e-5 * mx; s_sc[1] = 2.0;
                }
                __syncthreads();
                double rho = 0; int qmax = 0;
                do {
                    const double lambda = s_sc[0];
                    if (t < 12) sTb[t] = sT[t];
                    if (t == 0) {                              // (H + lambda I) x = b, dense Cholesky (LinearSolverDense) — the oracle's operations one for one:
                        double A[36], x[6];                    // with 1 / sqrt(pivot) and multiplications instead (measured: -0.03 ms per one-frame call) the
#pragma unroll                                                 // unconverged iterates of short rounds leave the 1e-8 bar (cond(H) ~ 1e8), so the divisions stay
                        for (int r = 0; r < 6; r++)
#pragma unroll
                            for (int c = 0; c < 6; c++) { const int rr = min(r, c), cc = max(r, c); A[r * 6 + c] = sH[rr * 6 - rr * (rr - 1) / 2 + (cc - rr)] + (r == c ? lambda : 0.0); }
                        bool ok = true;
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            double d = A[j * 6 + j];
#pragma unroll
                            for (int k2 = 0; k2 < j; k2++) d -= A[j * 6 + k2] * A[j * 6 + k2];
                            if (!(d > 0)) { ok = false; d = 1.0; }
                            A[j * 6 + j] = sqrt(d);
#pragma unroll
                            for (int i2 = j + 1; i2 < 6; i2++) {
                                double v = A[i2 * 6 + j];
#pragma unroll
                                for (int k2 = 0; k2 < j; k2++) v -= A[i2 * 6 + k2] * A[j * 6 + k2];
                                A[i2 * 6 + j] = v / A[j * 6 + j];
                            }
                        }
#pragma unroll
                        for (int i2 = 0; i2 < 6; i2++) { double v = sH[21 + i2]; for (int k2 = 0; k2 < i2; k2++) v -= A[i2 * 6 + k2] * x[k2]; x[i2] = v / A[i2 * 7]; }
#pragma unroll
                        for (int i2 = 5; i2 >= 0; i2--) { double v = x[i2]; for (int k2 = i2 + 1; k2 < 6; k2++) v -= A[k2 * 6 + i2] * x[k2]; x[i2] = v / A[i2 * 7]; }
#pragma unroll
                        for (int i2 = 0; i2 < 6; i2++) sx[i2] = x[i2];
                        s_sc[2] = ok ? 1.0 : 0.0;
                    }
                    __syncthreads();
                    const bool ok = s_sc[2] != 0.0;
                    if (ok && t == 0) pose_oplus(sT, sx);
                    __syncthreads();
                    const double tempChi = ok ? active_chi2() : 1e300;
                    double scale = 1e-3;
                    if (ok) for (int r = 0; r < 6; r++) scale += sx[r] * (lambda * sx[r] + sH[21 + r]);
                    rho = (currentChi - tempChi) / scale;
                    __syncthreads();
                    if (rho > 0 && isfinite(tempChi) && ok) {
                        if (t == 0) {
                            double alpha = 1. - pow(2 * rho - 1, 3);
                            alpha = fmin(alpha, 2. / 3.);
                            s_sc[0] = lambda * fmax(1. / 3., alpha); s_sc[1] = 2.0;
                        }
                        currentChi = tempChi; fresh = true;
                    } else {
                        if (t == 0) { s_sc[0] = lambda * s_sc[1]; s_sc[1] *= 2.0; }
                        if (t < 12) sT[t] = sTb[t];
                        fresh = false;
                    }
                    __syncthreads();
                    qmax++;
                } while (rho < 0 && qmax < 10 && isfinite(s_sc[0]));
                if (qmax == 10 || rho == 0 || !isfinite(s_sc[0])) break;
            }
        }
        if (round < 0) continue;
        // ---- classify every edge (frontend.cpp:229-242) ----
        double co = 0;
#pragma unroll
        for (int k = 0; k < PO_EPT; k++) {
            const int i = t + k * NT;
            if (i < n) {
                if ((outl >> k) & 1) { double e0, e1, pc[3]; edge_err(i, e0, e1, pc); echi[k] = e0 * e0 + e1 * e1; }
                if (echi[k] > a.chi2_th) { outl |= 1u << k; level |= 1u << k; co += 1.0; }
                else { outl &= ~(1u << k); level &= ~(1u << k); }
            }
        }
        cntOut = (int)block_sum_n<NW>(co, s_red);
        if (round == a.rounds - 2) robust = false;            // :244-246
    }
    // ---- write back ----
#pragma unroll
    for (int k = 0; k < PO_EPT; k++) { const int i = t + k * NT; if (i < n) a.outlier[(size_t)f * a.cap + i] = (outl >> k) & 1; }
    if (t == 0) {
        const double* R = sT;
        const double tr = R[0] + R[4] + R[8];
        double x, y, z, q;
        if (tr > 0) { const double s = sqrt(tr + 1.0) * 2; q = 0.25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s; }
        else if (R[0] > R[4] && R[0] > R[8]) { const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2; x = 0.25 * s; q = (R[7] - R[5]) / s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s; }
        else if (R[4] > R[8]) { const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2; y = 0.25 * s; q = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; z = (R[5] + R[7]) / s; }
        else { const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2; z = 0.25 * s; q = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; }
        pose[0] = x; pose[1] = y; pose[2] = z; pose[3] = q; pose[4] = R[9]; pose[5] = R[10]; pose[6] = R[11];
        a.n_inliers[f] = n - cntOut; a.status[f] = MYSLAM_OK;
    }
}

// nwaves == 0: the pose-list form (no per-wave copies of the pose blocks; chunk counts, pose totals / starts and the edge list instead)
static std::atomic<int> g_ba_build_pose_atomics{0};      // MYSLAM_BA_OPT_BUILD_POSE_ATOMICS
static size_t ba_lds(int maxP, int maxL, int maxE, int nwaves) {
    const size_t lists = nwaves ? 0 : (size_t)maxP * (((size_t)maxE + 63) / 64) + 2 * (size_t)maxP + 3 * (size_t)maxE;
    const size_t staged = nwaves ? 0 : 2 * (size_t)maxE + 3 * (size_t)maxL;
    return sizeof(double) * ((size_t)maxP * (12 + nwaves * 27) + staged) + sizeof(int) * (3 * (size_t)maxL + lists);
}

#ifndef MYSLAM_BA_WIDE_BELOW           // A/B builds (tools/build_variants.sh)
#define MYSLAM_BA_WIDE_BELOW 32
#endif
#ifndef MYSLAM_BA_LISTS_BELOW
#define MYSLAM_BA_LISTS_BELOW 32
#endif
constexpr int BA_WIDE_BELOW = MYSLAM_BA_WIDE_BELOW;      // fewer windows than this per call: 1024 threads per window (latency), else 256 (throughput)
constexpr int BA_LISTS_BELOW = MYSLAM_BA_LISTS_BELOW;    // fewer windows than this per call: the staged pose-list form when it fits LDS

static int ba_launch(const BaArgs& a, int nwin, hipStream_t s) {
    // a handful of windows (a live stream's key-frame): 1024 threads per window and, when the window fits LDS, the pose-list form.  Batches keep
    // 256 threads and the ds_add_f64 form: with two blocks per CU their time is the latency of a few resident waves either way, and the second
    // evaluation only adds to it (512 windows: 0.167 ms against 0.181, tools/ba_build_time.py).
    const bool wide_req = nwin < BA_WIDE_BELOW;
    const bool lists = wide_req && nwin < BA_LISTS_BELOW && !g_ba_build_pose_atomics.load() && ba_lds(a.maxP, a.maxL, a.maxE, 0) <= 150 * 1024;
    const bool wide = wide_req && (lists || ba_lds(a.maxP, a.maxL, a.maxE, 16) <= 150 * 1024);
    const size_t lds = ba_lds(a.maxP, a.maxL, a.maxE, lists ? 0 : wide ? 16 : 4);
    if (lds > 150 * 1024) return MYSLAM_ERR_CAPACITY;
    // (the limit is state of the function, not of the launch: always raised to what any plan may need — see launch_octree)
    static const bool raised = [] {
        bool ok = true;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build<1024, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
        ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(k_ba_build<1024, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
        return ok;
    }();
    (void)raised;
    ScopedProf sp(P_BA, s);
    if (lists) hipLaunchKernelGGL((k_ba_build<1024, true>), dim3(nwin), dim3(1024), lds, s, a);
    else if (wide) hipLaunchKernelGGL((k_ba_build<1024, false>), dim3(nwin), dim3(1024), lds, s, a);
    else hipLaunchKernelGGL((k_ba_build<256, false>), dim3(nwin), dim3(256), lds, s, a);
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
    HostCall hc;
    const int i_p = hc.in(poses, (size_t)nposes * 7), i_x = hc.in(points, (size_t)npts * 3), i_o = hc.in(obs, (size_t)nedges * 2);
    const int i_ep = hc.in(edge_pose, (size_t)nedges), i_el = hc.in(edge_pt, (size_t)nedges), i_f = hc.in(fixed_pt, fixed_pt ? (size_t)npts : 0);
    const int o_pp = hc.out(Hpp, (size_t)nposes * 36), o_ll = hc.out(Hll, (size_t)npts * 9), o_pl = hc.out(Hpl, (size_t)nedges * 18);
    const int o_bp = hc.out(bp, (size_t)nposes * 6), o_bl = hc.out(bl, (size_t)npts * 3), o_c = hc.out(chi2, (size_t)nedges);
    int rc = hc.upload();
    if (rc) return rc;
    BaArgs a{hc.dev<double>(i_p), hc.dev<double>(i_x), hc.dev<int32_t>(i_ep), hc.dev<int32_t>(i_el), hc.dev<double>(i_o),
             fixed_pt ? hc.dev<uint8_t>(i_f) : nullptr, nullptr, nposes, npts, nedges, nposes, npts, E,
             fx, fy, cx, cy, huber_delta, hc.dev<double>(o_pp), hc.dev<double>(o_ll), hc.dev<double>(o_pl), hc.dev<double>(o_bp),
             hc.dev<double>(o_bl), hc.dev<double>(o_c)};
    if ((rc = ba_launch(a, 1, hc.stream()))) return rc;
    return hc.download();
}

int myslam_ba_optimize_batch(double* d_poses, double* d_points, const int32_t* d_edge_pose, const int32_t* d_edge_pt, const double* d_obs,
                             const uint8_t* d_fixed, const int32_t* d_sizes, int nwin, int max_poses, int max_pts, int max_edges,
                             double fx, double fy, double cx, double cy, double huber_delta, int max_iters, double* d_scratch,
                             double* d_final_chi2, int32_t* d_iters, int32_t* d_status, void* hip_stream) {
    if (!d_poses || !d_points || !d_edge_pose || !d_edge_pt || !d_obs || !d_sizes || nwin < 1 || max_poses < 1 || max_pts < 1 ||
        max_edges < 1 || max_iters < 1 || !d_scratch || !d_final_chi2 || !d_iters || !d_status)
        return MYSLAM_ERR_INVALID;
    BaOptArgs a{d_poses, d_points, d_edge_pose, d_edge_pt, d_obs, d_fixed, d_sizes, 0, 0, 0, max_poses, max_pts, max_edges,
                fx, fy, cx, cy, huber_delta, max_iters, d_scratch, d_final_chi2, d_iters, d_status, 1, 0.0, nullptr, nullptr, nullptr, nullptr};
    return ba_opt_launch(a, nwin, (hipStream_t)hip_stream);
}

int myslam_ba_optimize(double* poses, int nposes, double* points, int npts, const int32_t* edge_pose, const int32_t* edge_pt,
                       const double* obs, int nedges, const uint8_t* fixed_pt, double fx, double fy, double cx, double cy,
                       double huber_delta, int max_iters, double* final_chi2, int* iters) {
    if (!poses || !points || nposes < 1 || npts < 1 || nedges < 1 || !edge_pose || !edge_pt || !obs || max_iters < 1) return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    HostCall hc;
    int32_t st[2] = {0, 0}; double chi = 0;
    const int i_p = hc.inout(poses, (size_t)nposes * 7), i_x = hc.inout(points, (size_t)npts * 3), i_o = hc.in(obs, (size_t)nedges * 2);
    const int i_ep = hc.in(edge_pose, (size_t)nedges), i_el = hc.in(edge_pt, (size_t)nedges), i_f = hc.in(fixed_pt, fixed_pt ? (size_t)npts : 0);
    const size_t wneed = std::max((size_t)nedges * 18, ba_opt_scratch_need(npts, nedges, true));
    const int o_st = hc.out(st, 2), o_chi = hc.out(&chi, 1), t_w = hc.tmp<double>(wneed);
    int rc = hc.upload();
    if (rc) return rc;
    int32_t* d_st = hc.dev<int32_t>(o_st);
    BaOptArgs a{hc.dev<double>(i_p), hc.dev<double>(i_x), hc.dev<int32_t>(i_ep), hc.dev<int32_t>(i_el), hc.dev<double>(i_o),
                fixed_pt ? hc.dev<uint8_t>(i_f) : nullptr, nullptr, nposes, npts, nedges, nposes, npts, nedges,
                fx, fy, cx, cy, huber_delta, max_iters, hc.dev<double>(t_w), hc.dev<double>(o_chi), d_st + 1, d_st, 1, 0.0, nullptr, nullptr, nullptr, nullptr, wneed};
    if ((rc = ba_opt_launch(a, 1, hc.stream()))) return rc;
    if ((rc = hc.download())) return rc;
    if (final_chi2) *final_chi2 = chi;
    if (iters) *iters = st[1];
    return st[0];
}

int myslam_ba_set_option(int option, int value) {
    if (option == MYSLAM_BA_OPT_BUILD_POSE_ATOMICS) { if (value < 0 || value > 1) return MYSLAM_ERR_INVALID; g_ba_build_pose_atomics.store(value); return MYSLAM_OK; }
    if (option == MYSLAM_BA_OPT_LANDMARKS_IN_HBM) { if (value < 0 || value > 1) return MYSLAM_ERR_INVALID; g_ba_landmarks_in_hbm.store(value); return MYSLAM_OK; }
    return MYSLAM_ERR_INVALID;
}

int myslam_ba_optimize_active_map_batch(double* d_poses, double* d_points, const int32_t* d_edge_pose, const int32_t* d_edge_pt,
                                        const double* d_obs, const uint8_t* d_fixed, const int32_t* d_sizes, int nwin, int max_poses,
                                        int max_pts, int max_edges, double fx, double fy, double cx, double cy, double huber_delta,
                                        double chi2_th, int max_rounds, int iters_per_round, double* d_scratch, double* d_edge_chi2,
                                        uint8_t* d_outlier, int32_t* d_rounds, int32_t* d_n_outliers, int32_t* d_status, void* hip_stream) {
    if (!d_poses || !d_points || !d_edge_pose || !d_edge_pt || !d_obs || !d_sizes || nwin < 1 || max_poses < 1 || max_pts < 1 ||
        max_edges < 1 || max_rounds < 1 || iters_per_round < 1 || !d_scratch || !d_edge_chi2 || !d_outlier || !d_rounds || !d_n_outliers || !d_status)
        return MYSLAM_ERR_INVALID;
    BaOptArgs a{d_poses, d_points, d_edge_pose, d_edge_pt, d_obs, d_fixed, d_sizes, 0, 0, 0, max_poses, max_pts, max_edges,
                fx, fy, cx, cy, huber_delta, iters_per_round, d_scratch, nullptr, nullptr, d_status,
                max_rounds, chi2_th, d_edge_chi2, d_outlier, d_rounds, d_n_outliers};
    return ba_opt_launch(a, nwin, (hipStream_t)hip_stream);
}

int myslam_ba_optimize_active_map(double* poses, int nposes, double* points, int npts, const int32_t* edge_pose, const int32_t* edge_pt,
                                  const double* obs, int nedges, const uint8_t* fixed_pt, double fx, double fy, double cx, double cy,
                                  double huber_delta, double chi2_th, int max_rounds, int iters_per_round,
                                  double* edge_chi2, uint8_t* outlier, int* rounds, int* n_outliers) {
    if (!poses || !points || nposes < 1 || npts < 1 || nedges < 1 || !edge_pose || !edge_pt || !obs || max_rounds < 1 || iters_per_round < 1 ||
        !edge_chi2 || !outlier)
        return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    HostCall hc;
    int32_t st[3] = {0, 0, 0};
    const int i_p = hc.inout(poses, (size_t)nposes * 7), i_x = hc.inout(points, (size_t)npts * 3), i_o = hc.in(obs, (size_t)nedges * 2);
    const int i_ep = hc.in(edge_pose, (size_t)nedges), i_el = hc.in(edge_pt, (size_t)nedges), i_f = hc.in(fixed_pt, fixed_pt ? (size_t)npts : 0);
    const int o_st = hc.out(st, 3), o_chi = hc.out(edge_chi2, (size_t)nedges), o_out = hc.out(outlier, (size_t)nedges);
    const size_t wneed = std::max((size_t)nedges * 18, ba_opt_scratch_need(npts, nedges, true));
    const int t_w = hc.tmp<double>(wneed);
    int rc = hc.upload();
    if (rc) return rc;
    int32_t* d_st = hc.dev<int32_t>(o_st);
    BaOptArgs a{hc.dev<double>(i_p), hc.dev<double>(i_x), hc.dev<int32_t>(i_ep), hc.dev<int32_t>(i_el), hc.dev<double>(i_o),
                fixed_pt ? hc.dev<uint8_t>(i_f) : nullptr, nullptr, nposes, npts, nedges, nposes, npts, nedges,
                fx, fy, cx, cy, huber_delta, iters_per_round, hc.dev<double>(t_w), nullptr, nullptr, d_st, max_rounds, chi2_th,
                hc.dev<double>(o_chi), hc.dev<uint8_t>(o_out), d_st + 1, d_st + 2, wneed};
    if ((rc = ba_opt_launch(a, 1, hc.stream()))) return rc;
    if ((rc = hc.download())) return rc;
    if (rounds) *rounds = st[1];
    if (n_outliers) *n_outliers = st[2];
    return st[0];
}

// Block size (measured on MI355X, tools/frontend_time.py + tools/latency_frontend.py; PO_EPT = 8 edges per thread is the kernel's limit).
// Many frames: small blocks, several frames per CU — 1024 frames x 150 / 500 / 1000 matches take 0.42 / 0.63 / 0.93 ms with 64 / 128 / 128
// threads against 1.46 / 1.56 / 1.84 ms with 512.  A few frames (the tracker's one-frame call): the latency of ~45 dependent Levenberg steps,
// 0.24 - 0.31 ms at 150 - 400 matches; 256 threads are best there (round 5, with the halving sums: 0.238 / 0.314 ms with 256 threads, 0.31 / 0.34
// with 128, 0.33 / 0.34 with 64, 0.345 at 400 matches with 512).
static void pose_only_launch(const PoseOnlyArgs& a, int batch, int max_edges, hipStream_t s) {
    int nt;
    if (batch >= 64) nt = max_edges <= 256 ? 64 : (max_edges <= 1024 ? 128 : (max_edges <= 2048 ? 256 : 512));
    else nt = max_edges <= 128 ? 128 : (max_edges <= 2048 ? 256 : 512);
    if (nt == 64) hipLaunchKernelGGL(k_pose_only<64>, dim3(batch), dim3(64), 0, s, a);
    else if (nt == 128) hipLaunchKernelGGL(k_pose_only<128>, dim3(batch), dim3(128), 0, s, a);
    else if (nt == 256) hipLaunchKernelGGL(k_pose_only<256>, dim3(batch), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_pose_only<512>, dim3(batch), dim3(512), 0, s, a);
}

int myslam_pose_only_optimize_batch(double* d_poses, const double* d_pts3d, const double* d_obs, const int32_t* d_counts, int batch, int cap,
                                    double fx, double fy, double cx, double cy, double chi2_th, int rounds, int iters, int pre_optimize,
                                    uint8_t* d_outlier, int32_t* d_n_inliers, int32_t* d_status, void* hip_stream) {
    if (!d_poses || !d_pts3d || !d_obs || !d_counts || batch < 1 || cap < 1 || rounds < 1 || iters < 1 || pre_optimize < 0 || !d_outlier ||
        !d_n_inliers || !d_status)
        return MYSLAM_ERR_INVALID;
    PoseOnlyArgs a{d_poses, d_pts3d, d_obs, d_counts, 0, cap, fx, fy, cx, cy, chi2_th, rounds, iters, pre_optimize, d_outlier, d_n_inliers, d_status};
    ScopedProf sp(P_BA, (hipStream_t)hip_stream);
    pose_only_launch(a, batch, cap, (hipStream_t)hip_stream);
    MYSLAM_HIP_CHECK(hipGetLastError());
    return MYSLAM_OK;
}

int myslam_pose_only_optimize(double* pose7, const double* pts3d, const double* obs, int n, double fx, double fy, double cx, double cy,
                              double chi2_th, int rounds, int iters, int pre_optimize, uint8_t* outlier, int* n_inliers) {
    if (!pose7 || n < 0 || (n > 0 && (!pts3d || !obs || !outlier)) || rounds < 1 || iters < 1 || pre_optimize < 0) return MYSLAM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return MYSLAM_ERR_HIP;
    const int m = n > 0 ? n : 1;
    HostCall hc;
    int32_t st[2] = {0, 0};
    const int i_p = hc.inout(pose7, 7), i_x = hc.in(pts3d, (size_t)3 * n), i_o = hc.in(obs, (size_t)2 * n);
    const int o_st = hc.out(st, 2), o_out = hc.out(outlier, (size_t)n);
    int rc = hc.upload();
    if (rc) return rc;
    int32_t* d_i = hc.dev<int32_t>(o_st);
    PoseOnlyArgs a{hc.dev<double>(i_p), hc.dev<double>(i_x), hc.dev<double>(i_o), nullptr, n, m, fx, fy, cx, cy, chi2_th, rounds, iters, pre_optimize,
                   hc.dev<uint8_t>(o_out), d_i, d_i + 1};
    pose_only_launch(a, 1, n, hc.stream());
    MYSLAM_HIP_CHECK(hipGetLastError());
    if ((rc = hc.download())) return rc;
    if (n_inliers) *n_inliers = st[0];
    return st[1];
}

// ------------------------------------------------------------------------------------------------
// Map -> flat arrays (host): the graph-build rules of Backend::OptimizeActiveMap, src/backend.cpp:139-206.  See myslam_hip.h.
// ------------------------------------------------------------------------------------------------
int myslam_ba_flatten_window(const uint64_t* active_kf_ids, int n_kf, const uint64_t* mp_ids, const uint8_t* mp_outlier,
                             const uint64_t* mp_first_observer_kf, int n_mp, const uint64_t* obs_mp_id, const uint64_t* obs_kf_id,
                             const float* obs_uv, const uint8_t* obs_feat_outlier, int n_obs, int32_t* pose_src, int32_t* pt_src,
                             int32_t* n_pts, int32_t* edge_pose, int32_t* edge_pt, double* edge_obs, int32_t* edge_src, int32_t* n_edges,
                             uint8_t* fixed_pt) {
    if (n_kf < 0 || n_mp < 0 || n_obs < 0 || !n_pts || !n_edges) return MYSLAM_ERR_INVALID;
    if ((n_kf && (!active_kf_ids || !pose_src)) || (n_mp && (!mp_ids || !mp_first_observer_kf || !pt_src || !fixed_pt)) ||
        (n_obs && (!obs_mp_id || !obs_kf_id || !obs_uv || !edge_pose || !edge_pt || !edge_obs || !edge_src)))
        return MYSLAM_ERR_INVALID;
    *n_pts = 0; *n_edges = 0;
    // pose slots: ascending key-frame id (g2o sorts its active vertices by id; ids are unique map keys)
    std::vector<int32_t> korder(n_kf);
    for (int i = 0; i < n_kf; i++) korder[i] = i;
    std::sort(korder.begin(), korder.end(), [&](int a, int b) { return active_kf_ids[a] < active_kf_ids[b]; });
    for (int i = 1; i < n_kf; i++)
        if (active_kf_ids[korder[i]] == active_kf_ids[korder[i - 1]]) return MYSLAM_ERR_INVALID;
    std::unordered_map<uint64_t, int32_t> kslot, mrow;
    kslot.reserve(n_kf * 2); mrow.reserve(n_mp * 2);
    for (int i = 0; i < n_kf; i++) { pose_src[i] = korder[i]; kslot[active_kf_ids[korder[i]]] = i; }
    for (int i = 0; i < n_mp; i++)
        if (!mrow.emplace(mp_ids[i], i).second) return MYSLAM_ERR_INVALID;
    // observations per map point, list order kept (a stable bucket pass over the rows)
    std::vector<int32_t> cnt(n_mp + 1, 0), rows(n_obs);
    std::vector<int32_t> orow(n_obs);
    for (int k = 0; k < n_obs; k++) {
        auto it = mrow.find(obs_mp_id[k]);
        if (it == mrow.end()) return MYSLAM_ERR_INVALID;                                   // an observation of a map point that is not active
        orow[k] = it->second; cnt[it->second + 1]++;
    }
    for (int i = 0; i < n_mp; i++) cnt[i + 1] += cnt[i];
    {
        std::vector<int32_t> fill(cnt.begin(), cnt.end() - 1);
        for (int k = 0; k < n_obs; k++) rows[fill[orow[k]]++] = k;
    }
    // landmark slots: ascending map-point id
    std::vector<int32_t> morder(n_mp);
    for (int i = 0; i < n_mp; i++) morder[i] = i;
    std::sort(morder.begin(), morder.end(), [&](int a, int b) { return mp_ids[a] < mp_ids[b]; });
    int L = 0, E = 0;
    for (int oi = 0; oi < n_mp; oi++) {
        const int i = morder[oi];
        if (mp_outlier && mp_outlier[i]) continue;                                         // :163
        const int e0 = E;
        for (int r = cnt[i]; r < cnt[i + 1]; r++) {
            const int k = rows[r];
            const auto ks = kslot.find(obs_kf_id[k]);
            if (ks == kslot.end()) return MYSLAM_ERR_INVALID;                              // :187 assert — behind the outlier skip of :163, as there
            if (obs_feat_outlier && obs_feat_outlier[k]) continue;                         // :189
            edge_pose[E] = ks->second; edge_pt[E] = L;
            edge_obs[2 * E] = (double)obs_uv[2 * k]; edge_obs[2 * E + 1] = (double)obs_uv[2 * k + 1];      // toVec2(feat->mkpPosition.pt), :194
            edge_src[E] = k;
            E++;
        }
        if (E == e0) continue;                                                             // no edge: g2o never activates the vertex
        pt_src[L] = i;
        fixed_pt[L] = kslot.find(mp_first_observer_kf[i]) == kslot.end() ? 1 : 0;          // :175-177
        L++;
    }
    *n_pts = L; *n_edges = E;
    return MYSLAM_OK;
}

}  // extern "C"
