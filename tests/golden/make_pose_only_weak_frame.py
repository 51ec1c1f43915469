#!/usr/bin/env python3
"""tests/golden/pose_only_weak_frame.npz: pose-only call 327 of the `one_way` drive (tests/kitti_layout.py VARIANTS; the lock-step run of
tests/test_gpu_runner_variants.py writes a problem that misses its tight bar to gpurun_out/pose_only_mismatch_<call>.npz) — a frame that keeps 39
matches, 34 of them inliers.  One Levenberg accept / reject decision of its four rounds hangs on the last bits of a sum: the ORACLE run on
observations that differ by one ulp lands on one of TWO poses, 2.7e-5 m apart in t_z, with identical flags.

    python tests/golden/make_pose_only_weak_frame.py gpurun_out/pose_only_mismatch_327.npz

stores the problem, the oracle's result, and the oracle's results on eight seeded one-ulp perturbations (the two branches)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402


def one_ulp_runs(o, pose, p3, obs, Kt, pre, n=8):
    rng = np.random.default_rng(0)
    return [o.pose_only_optimize(pose, p3, obs * (1 + rng.choice([-1.0, 1.0], size=obs.shape) * 2.2e-16), Kt, pre_optimize=pre) for _ in range(n)]


if __name__ == "__main__":
    d = np.load(sys.argv[1])
    o = pyoracle.Oracle()
    pose, p3, obs, Kt, pre = d["pose"], d["p3"], d["obs"], tuple(float(x) for x in d["Kt"]), int(d["pre"])
    rp, ro, ri = o.pose_only_optimize(pose, p3, obs, Kt, pre_optimize=pre)
    runs = one_ulp_runs(o, pose, p3, obs, Kt, pre)
    np.savez(os.path.join(ROOT, "tests", "golden", "pose_only_weak_frame.npz"), pose=pose, p3=p3, obs=obs, K=np.array(Kt), pre=pre, ref_pose=rp, ref_outlier=ro,
             ref_inliers=ri, ulp_poses=np.stack([q[0] for q in runs]), ulp_inliers=np.array([q[2] for q in runs]))
    print("matches", len(p3), "inliers", ri, "one-ulp runs: distance from the unperturbed result", [float(f"{np.abs(q[0] - rp).max():.2e}") for q in runs])
