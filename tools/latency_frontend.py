#!/usr/bin/env python3
"""Per-call latency of the two calls the tracker makes on EVERY frame (host pointers, B = 1): myslam_lk_track and
myslam_pose_only_optimize at the sizes the KITTI-layout sequence produces (150 - 400 points).   python tools/latency_frontend.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from __graft_entry__ import load_package
pkg = load_package(); api, synth = pkg.api, pkg.synth
K = synth.KITTI00; Kt = (K["fx"], K["fy"], K["cx"], K["cy"])
rng = np.random.default_rng(0)
out = {}
fr = synth.stereo_batch(2, stream_id=5)
a, b = np.ascontiguousarray(fr[0, 0]), np.ascontiguousarray(fr[1, 0])
lk = api.LKTracker()
for n in (150, 400):
    p0 = np.stack([rng.uniform(30, 1200, n), rng.uniform(30, 340, n)], 1).astype(np.float32)
    for _ in range(5): lk.track(a, b, p0, p0)
    t = []
    for _ in range(50):
        t0 = time.perf_counter(); lk.track(a, b, p0, p0); t.append(time.perf_counter() - t0)
    out[f"lk_track n={n}"] = float(np.median(t) * 1e3)
    # the tracker's steady state with the handle's image cache: prev is the last call's next (a hit), next was prefetched while "the pose was optimised"
    seq = [a, b]
    t = []
    for i in range(60):
        p_, n_ = seq[i & 1], seq[(i + 1) & 1]
        t0 = time.perf_counter(); lk.track_cached(p_, 2 * i + 1, n_, 2 * i + 3, p0, p0); t.append(time.perf_counter() - t0)
    out[f"lk_track_cached n={n} (prev hit)"] = float(np.median(t[10:]) * 1e3)
    t = []
    for i in range(60):
        p_, n_ = seq[i & 1], seq[(i + 1) & 1]
        lk.prefetch(n_, 1000 + 2 * i + 3); time.sleep(0.0005)
        t0 = time.perf_counter(); lk.track_cached(p_, 1000 + 2 * i + 1, n_, 1000 + 2 * i + 3, p0, p0); t.append(time.perf_counter() - t0)
    out[f"lk_track_cached n={n} (prev hit, next prefetched)"] = float(np.median(t[10:]) * 1e3)
    P = np.stack([rng.uniform(-20, 20, n), rng.uniform(-3, 3, n), rng.uniform(6, 40, n)], 1)
    pose = np.array([0, 0, 0, 1, 0.3, 0, 0.1])
    obs = np.stack([K["fx"] * P[:, 0] / P[:, 2] + K["cx"], K["fy"] * P[:, 1] / P[:, 2] + K["cy"]], 1) + rng.normal(0, 0.5, (n, 2))
    obs[: n // 20] += 30
    for _ in range(5): api.pose_only_optimize(pose, P, obs, Kt)
    t = []
    for _ in range(50):
        t0 = time.perf_counter(); api.pose_only_optimize(pose, P, obs, Kt); t.append(time.perf_counter() - t0)
    out[f"pose_only n={n}"] = float(np.median(t) * 1e3)
print(json.dumps(out))
