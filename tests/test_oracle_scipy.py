"""Cross-implementation pins that need no OpenCV / g2o: the oracle's Levenberg optima against scipy.optimize.least_squares on the same
cost functions (the g2o-derived arithmetic is otherwise "parity unpinned").

  * local BA (Backend::OptimizeActiveMap, src/backend.cpp:126-232): EdgeProjection error (g2o_types.h:115-122), Huber kernel with
    delta = 5.991 on the squared edge error, pose update exp(delta) * T (g2o_types.h:32-37)
  * pose graph (LoopClosing::PoseGraphOptimization, src/loopclosing.cpp:537-610): EdgePoseGraph error
    log(M^-1 * T0 * T1^-1) (g2o_types.h:157-168), no robust kernel
"""
import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.sparse import lil_matrix


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])


def se3_exp(d):
    """Sophus SE3d::exp, tangent = (upsilon, omega)"""
    u, w = d[:3], d[3:]
    th = np.linalg.norm(w); W = hat(w)
    if th < 1e-10:
        R = np.eye(3) + W; V = np.eye(3) + 0.5 * W
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = V @ u
    return T


def se3_log(T):
    R, t = T[:3, :3], T[:3, 3]
    c = np.clip((np.trace(R) - 1) / 2, -1, 1); th = np.arccos(c)
    if th < 1e-10:
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    else:
        w = th / (2 * np.sin(th)) * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    W = hat(w); th = np.linalg.norm(w)
    if th < 1e-10:
        Vi = np.eye(3) - 0.5 * W
    else:
        Vi = np.eye(3) - 0.5 * W + (1 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * W @ W
    return np.concatenate([Vi @ t, w])


def T_of(p7):
    T = np.eye(4); T[:3, :3] = quat_R(p7[:4]); T[:3, 3] = p7[4:]
    return T


class BAProblem:
    """numpy restatement of the local-BA cost: sum over edges of rho_huber(|z - pi(T X)|^2), T <- exp(d) T, X <- X + dx (free landmarks)"""

    def __init__(self, poses, pts, ep, el, obs, fixed, K, delta=5.991):
        self.pts, self.ep, self.el, self.obs, self.K, self.delta = pts, ep, el, obs, K, delta
        self.P, self.E = len(poses), len(ep)
        self.T0 = np.stack([T_of(p) for p in poses])
        self.free = np.nonzero(np.asarray(fixed) == 0)[0]
        slot = -np.ones(len(pts), int); slot[self.free] = np.arange(len(self.free))
        self.n = 6 * self.P + 3 * len(self.free)
        self.S = lil_matrix((self.E, self.n), dtype=int)
        for k in range(self.E):
            self.S[k, 6 * ep[k]:6 * ep[k] + 6] = 1
            if slot[el[k]] >= 0:
                self.S[k, 6 * self.P + 3 * slot[el[k]]:6 * self.P + 3 * slot[el[k]] + 3] = 1

    def unpack(self, x):
        Ts = np.stack([se3_exp(x[6 * i:6 * i + 6]) @ self.T0[i] for i in range(self.P)])
        X = self.pts.copy(); X[self.free] += x[6 * self.P:].reshape(-1, 3)
        return Ts, X

    def edge_errors(self, x):                                  # EdgeProjection::computeError, g2o_types.h:115-122
        Ts, X = self.unpack(x)
        pc = np.einsum("eab,eb->ea", Ts[self.ep][:, :3, :3], X[self.el]) + Ts[self.ep][:, :3, 3]
        u = self.K[0] * pc[:, 0] / pc[:, 2] + self.K[2]; v = self.K[1] * pc[:, 1] / pc[:, 2] + self.K[3]
        return np.stack([self.obs[:, 0] - u, self.obs[:, 1] - v], 1)

    def edge_norms(self, x):
        return np.linalg.norm(self.edge_errors(x), axis=1)

    def robust_residuals(self, x):
        """2-vector per edge scaled so that its squared norm is rho(e^2): the sum of squares IS g2o's robustified cost (the kernel acts
        on the squared EDGE error, not per coordinate), and the Gauss-Newton model keeps both directions of every edge"""
        e = self.edge_errors(x); r = np.linalg.norm(e, axis=1); d = self.delta
        sc = np.where(r <= d, 1.0, np.sqrt(np.maximum(2 * d * r - d * d, 0)) / np.maximum(r, 1e-300))
        return (e * sc[:, None]).ravel()

    def cost(self, x):                                         # g2o's RobustKernelHuber on e^2
        r = self.edge_norms(x); d = self.delta
        return float(np.sum(np.where(r <= d, r * r, 2 * d * r - d * d)))

    def gradient(self, x, h=1e-6):                             # central differences of the robust cost, one column at a time
        g = np.zeros(self.n)
        for j in range(self.n):
            e = np.zeros(self.n); e[j] = h
            g[j] = (self.cost(x + e) - self.cost(x - e)) / (2 * h)
        return g

    def solve(self, dense=False, **kw):
        """dense: full finite-difference Jacobian + exact (SVD) trust-region steps — for small windows; else sparse differences + lsmr"""
        if dense:
            extra = dict(tr_solver="exact")
        else:
            S2 = lil_matrix((2 * self.E, self.n), dtype=int)
            S2[0::2] = self.S; S2[1::2] = self.S
            extra = dict(jac_sparsity=S2)
        return least_squares(self.robust_residuals, np.zeros(self.n), method="trf", ftol=1e-15, xtol=1e-15, gtol=1e-13, **extra, **kw)


def test_local_ba_optimum_is_stationary_for_an_independent_cost(oracle, synth):
    """BASELINE configs[3]'s window (10 key-frames x 300 landmarks, ~2950 edges, 3 % gross outliers, 10 % fixed landmarks; the camera
    moves ALONG its optical axis, so landmark depths near the axis are almost unobservable and generic solvers crawl): the point the
    oracle's Levenberg converges to is a stationary point of this file's numpy cost — same cost value, vanishing gradient, and scipy's
    trust-region solver started there finds nothing to improve."""
    poses, pts, ep, el, obs, fixed, K = synth.ba_problem()
    op, ox, chi, iters = oracle.ba_optimize(poses, pts, ep, el, obs, fixed, K, iters=100)
    assert iters < 100                                                                     # g2o's stop criterion fired: converged, not cut off
    start = BAProblem(poses, pts, ep, el, obs, fixed, K); at = BAProblem(op, ox, ep, el, obs, fixed, K)
    z = np.zeros(at.n)
    assert abs(at.cost(z) - chi) <= 1e-11 * chi, (at.cost(z), chi)                         # the oracle's robustified chi2 is this cost
    assert at.cost(z) < 0.7 * start.cost(z)                                                # (the start is away from the optimum; the floor is the outliers' Huber cost)
    g0, g1 = start.gradient(z), at.gradient(z)
    assert np.abs(g1).max() < 2e-7 * np.abs(g0).max(), (np.abs(g1).max(), np.abs(g0).max())
    sol = at.solve(max_nfev=60)
    assert 2 * sol.cost > chi * (1 - 1e-10) and np.abs(sol.x).max() < 1e-6, (2 * sol.cost, chi, np.abs(sol.x).max())
    assert np.array_equal(ox[fixed != 0], pts[fixed != 0])                                 # fixed landmarks do not move
    assert (at.edge_norms(z) ** 2 > 5.991).sum() >= int(0.02 * len(ep))                    # the gross outliers are still there to be flagged


def test_local_ba_optimum_matches_scipy(oracle):
    """A well-conditioned window (6 key-frames moving SIDEWAYS, 60 landmarks, 25 % fixed, 6 % of the observations 8-25 px off, i.e. in
    Huber's linear zone): oracle and scipy, both from the perturbed start, reach the same optimum — cost to 1e-9, poses and landmarks to 1e-6."""
    rng = np.random.default_rng(11)
    K = (718.856, 718.856, 607.1928, 185.2157)
    P, L = 6, 60
    poses = np.zeros((P, 7)); poses[:, 3] = 1; poses[:, 4] = -0.5 * np.arange(P)           # Tcw: camera i at x = 0.5 i, looking down +z
    X = np.stack([rng.uniform(-6, 8, L), rng.uniform(-2, 2, L), rng.uniform(6, 20, L)], 1)
    ep = np.tile(np.arange(P, dtype=np.int32), L); el = np.repeat(np.arange(L, dtype=np.int32), P)      # grouped by landmark
    pc = X[el] + poses[ep, 4:]
    obs = np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1) + rng.normal(0, 0.5, (P * L, 2))
    bad = rng.choice(P * L, int(0.06 * P * L), replace=False)
    obs[bad] += rng.uniform(8, 25, (len(bad), 2)) * rng.choice([-1, 1], (len(bad), 2))
    fixed = (rng.random(L) < 0.25).astype(np.uint8)
    pts = X + rng.normal(0, 0.05, X.shape) * (1 - fixed)[:, None]
    poses[:, 4:] += rng.normal(0, 0.02, (P, 3))
    pr = BAProblem(poses, pts, ep, el, obs, fixed, K)
    sol = pr.solve(dense=True, x_scale="jac", max_nfev=100)
    op, ox, chi, iters = oracle.ba_optimize(poses, pts, ep, el, obs, fixed, K, iters=100)
    assert iters < 100 and abs(2 * sol.cost - chi) <= 1e-9 * chi, (2 * sol.cost, chi, iters)
    assert (pr.edge_norms(sol.x) > 5.991).sum() >= len(bad) - 2                            # the Huber zone is populated
    Ts, Xs = pr.unpack(sol.x)
    assert np.abs(Ts - np.stack([T_of(p) for p in op])).max() < 1e-6 and np.abs(Xs - ox).max() < 1e-6


def test_pose_graph_optimum_matches_scipy(oracle, synth):
    """A 60-key-frame loop (chain + one loop edge, the reference's fixed set): the oracle's pose-graph optimum against scipy on
    e = log(M^-1 * T0 * T1^-1) with T <- exp(d) * T."""
    poses, fixed, e0, e1, meas, _ = synth.pose_graph(60, 1, seed=3)
    n = len(poses); T0 = np.stack([T_of(p) for p in poses]); Minv = np.stack([np.linalg.inv(T_of(m)) for m in meas])
    free = np.nonzero(fixed == 0)[0]; slot = -np.ones(n, int); slot[free] = np.arange(len(free))

    def unpack(x):
        Ts = T0.copy()
        for i, v in enumerate(free):
            Ts[v] = se3_exp(x[6 * i:6 * i + 6]) @ T0[v]
        return Ts

    def resid(x):
        Ts = unpack(x)
        return np.concatenate([se3_log(Minv[k] @ Ts[e0[k]] @ np.linalg.inv(Ts[e1[k]])) for k in range(len(e0))])

    S = lil_matrix((6 * len(e0), 6 * len(free)), dtype=int)
    for k in range(len(e0)):
        for v in (e0[k], e1[k]):
            if slot[v] >= 0:
                S[6 * k:6 * k + 6, 6 * slot[v]:6 * slot[v] + 6] = 1
    sol = least_squares(resid, np.zeros(6 * len(free)), jac_sparsity=S, method="trf", ftol=1e-15, xtol=1e-15, gtol=1e-14, max_nfev=60)
    # polish with exact (SVD) trust-region steps on the dense finite-difference Jacobian: the sparse path's lsmr crawls near the optimum
    x1 = sol.x
    resid1 = lambda d: resid(x1 + d)
    sol = least_squares(resid1, np.zeros_like(x1), tr_solver="exact", method="trf", ftol=1e-15, xtol=1e-15, gtol=1e-14, max_nfev=12)
    sol.x = x1 + sol.x
    op = poses.copy()
    for _ in range(3):
        op, chi, _ = oracle.pose_graph_optimize(op, fixed, e0, e1, meas, iters=20)
    Ts_o = np.stack([T_of(p) for p in op])
    r_o = np.concatenate([se3_log(Minv[k] @ Ts_o[e0[k]] @ np.linalg.inv(Ts_o[e1[k]])) for k in range(len(e0))])
    c_o, c_s = float(r_o @ r_o), float(2 * sol.cost)
    assert abs(c_o - chi) <= 1e-9 * max(c_o, 1e-12) + 1e-12, (c_o, chi)                    # the oracle's chi2 is this test's cost function
    assert abs(c_o - c_s) <= 1e-6 * c_s, (c_o, c_s)
    assert np.abs(unpack(sol.x) - Ts_o).max() < 5e-6
    assert np.abs(op[fixed != 0] - poses[fixed != 0]).max() < 1e-12          # fixed key-frames keep their pose (re-normalised quaternion)
