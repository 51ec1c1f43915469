#!/usr/bin/env python3
"""Regenerates the committed golden fixtures from the CPU oracle (inputs + expected outputs only — data).
The reference has no golden vectors and cannot be built, so these pin the oracle itself
against accidental change and give the GPU tests a second, file-based comparison point.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_package  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
from pyoracle import Oracle  # noqa: E402

synth = load_package().synth
o = Oracle()

img = synth.random_image(4242, 240, 320)
p = o.params(400)
kps, desc = o.detect_and_compute(p, img)
np.savez_compressed(os.path.join(HERE, "orb_small.npz"), image=img, nfeatures=400, kps=kps, desc=desc,
                    detect100=o.detect(o.params(100), img))

rng = np.random.default_rng(7)
q = rng.integers(0, 256, (64, 32), dtype=np.uint8); t = rng.integers(0, 256, (80, 32), dtype=np.uint8)
t[40] = t[2]; q[5] = t[2]
idx, dist = o.hamming_match(q, t)
np.savez_compressed(os.path.join(HERE, "hamming_small.npz"), q=q, t=t, idx=idx, dist=dist)

w = synth.calc_weights()
x = synth._rng(99).uniform(0, 1, (120, 160)).astype(np.float32)
np.savez_compressed(os.path.join(HERE, "calc_small.npz"), x=x, descr=o.calc_forward(w, x), weights_seed=0xCA1C)

poses, pts, ep, el, obs, fixed, K = synth.ba_problem(n_kf=4, n_mp=25)
Hpp, Hll, Hpl, bp, bl, chi2 = o.ba_build(poses, pts, ep, el, obs, fixed, K)
np.savez_compressed(os.path.join(HERE, "ba_small.npz"), poses=poses, pts=pts, ep=ep, el=el, obs=obs, fixed=fixed, K=np.array(K),
                    Hpp=Hpp, Hll=Hll, Hpl=Hpl, bp=bp, bl=bl, chi2=chi2)
# frontend operators: LK tracking of the ORB keypoints into a shifted copy, pose-only optimisation
img2 = np.roll(img, (1, -3), axis=(0, 1)).copy()
k100 = o.detect(o.params(100), img)
pts0 = np.stack([k100["x"], k100["y"]], 1).astype(np.float32)
lk_pts, lk_st, lk_err = o.lk_track(img, img2, pts0, pts0)
np.savez_compressed(os.path.join(HERE, "lk_small.npz"), prev=img, next=img2, pts=pts0, out=lk_pts, status=lk_st, err=lk_err)

rng = np.random.default_rng(17)
Kt = (718.856, 718.856, 607.1928, 185.2157)
P3 = np.stack([rng.uniform(-10, 10, 90), rng.uniform(-3, 3, 90), rng.uniform(5, 40, 90)], 1)
Tt = o.se3_exp(np.array([0.2, -0.05, 0.4, 0.01, -0.015, 0.02]))
x, y, z, w_ = Tt[:4]
Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w_), 2 * (x * z + y * w_)], [2 * (x * y + z * w_), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w_)],
               [2 * (x * z - y * w_), 2 * (y * z + x * w_), 1 - 2 * (x * x + y * y)]])
pc = P3 @ Rm.T + Tt[4:]
uv = np.stack([Kt[0] * pc[:, 0] / pc[:, 2] + Kt[2], Kt[1] * pc[:, 1] / pc[:, 2] + Kt[3]], 1) + rng.normal(0, 0.4, (90, 2))
uv[::9] += 30.0
T0 = np.array([0, 0, 0, 1, 0, 0, 0.0])
po_pose, po_out, po_inl = o.pose_only_optimize(T0, P3, uv, Kt)
np.savez_compressed(os.path.join(HERE, "pose_only_small.npz"), pose0=T0, pts3d=P3, obs=uv, K=np.array(Kt), pose=po_pose, outlier=po_out, inliers=po_inl)
# loop correction: a 40 key-frame pose graph with one loop, optimised for 3 and for 20 iterations
pg = synth.pose_graph(40, 1, seed=0x60)
pg3 = o.pose_graph_optimize(*pg[:5], iters=3)
pg20 = o.pose_graph_optimize(*pg[:5], iters=20)
np.savez_compressed(os.path.join(HERE, "pgo_small.npz"), poses=pg[0], fixed=pg[1], e0=pg[2], e1=pg[3], meas=pg[4],
                    poses3=pg3[0], chi3=pg3[1], poses20=pg20[0], chi20=pg20[1], its20=pg20[2])
# loop verification (rank 3): PnP-RANSAC on 90 matches with 30 % mismatches
pn = synth.pnp_problem(90, 0.3, 0.5, seed=0x77)
rc, pn_pose, pn_inl, pn_n = o.solve_pnp_ransac(pn[0], pn[1], pn[2])
assert rc == 0
np.savez_compressed(os.path.join(HERE, "pnp_small.npz"), pts3d=pn[0], pts2d=pn[1], K=np.array(pn[2]), pose=pn_pose, inlier=pn_inl, n_inliers=pn_n)
print("golden fixtures written to", HERE)
