#!/usr/bin/env python3
"""Writes tests/golden/ba_chaotic_window.npz: the one local-BA window of the round-4 fuzz campaigns on which the HIP solve and the oracle
part ways (tools/gpu_fuzz_ba.py seed 5300, case 637: 4 key-frames x 25 landmarks, 96 edges after three duplicated ones, 30 % gross
outliers by construction, 62 of 96 edges flagged).  Inputs, the oracle's first three Levenberg iterates, its final answer, and — the point
of the fixture — the oracle AGAINST ITSELF with the observations moved by one ulp: the same ten-fold growth per iteration and the same
kind of final disagreement (61 against 62 outliers, poses 0.4 apart).  CPU only:   python tests/golden/make_ba_chaotic_window.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
from pyoracle import Oracle
synth = g.load_package().synth
o = Oracle()

rng = np.random.default_rng(5300)                     # replay of the fuzzer's case generator up to case 637
for it in range(638):
    n_kf = int(rng.integers(1, 11)); n_mp = int(rng.integers(3, 300))
    seed = int(rng.integers(1 << 30)); of = float(rng.choice([0.0, 0.03, 0.3]))
    poses, pts, ep, el, obs, fixed, K = synth.ba_problem(seed=seed, n_kf=n_kf, n_mp=n_mp, outlier_frac=of)
    mode = int(rng.integers(0, 5))
    if mode == 1 and len(ep) > 4:
        for _ in range(3):
            k = int(rng.integers(0, len(ep)))
            ep = np.insert(ep, k, ep[k]); el = np.insert(el, k, el[k]); obs = np.insert(obs, k, obs[k] + rng.normal(0, 1, 2), axis=0)
    if mode == 2:
        keep = np.ones(len(ep), bool)
        for l in rng.choice(n_mp, max(1, n_mp // 4), replace=False):
            idx = np.where(el == l)[0]
            keep[idx[int(rng.integers(0, 2)):]] = False
        ep, el, obs = ep[keep], el[keep], obs[keep]
    if mode == 3:
        fixed = np.ones_like(fixed)
assert (n_kf, n_mp, len(ep), mode) == (4, 25, 96, 1)
out = dict(poses=poses, pts=pts, ep=ep, el=el, obs=obs, fixed=fixed, K=np.array(K))
ref = o.ba_optimize_active_map(poses, pts, ep, el, obs, fixed, K)
out.update(ref_poses=ref[0], ref_pts=ref[1], ref_chi=ref[2], ref_out=ref[3], ref_rounds=ref[4], ref_nout=ref[5])
ITERS = (1, 2, 3, 5, 10)
for i in ITERS:
    a = o.ba_optimize(poses, pts, ep, el, obs, fixed, K, iters=i)
    out[f"it{i}_poses"] = a[0]; out[f"it{i}_pts"] = a[1]; out[f"it{i}_chi2"] = a[2]; out[f"it{i}_iters"] = a[3]
prng = np.random.default_rng(1)
spread = np.zeros((4, len(ITERS))); fin = []
for t in range(4):                                    # the oracle against itself: observations moved by one ulp
    obs2 = obs * (1.0 + prng.choice([-1, 1], obs.shape) * 2.0 ** -52)
    for j, i in enumerate(ITERS):
        b = o.ba_optimize(poses, pts, ep, el, obs2, fixed, K, iters=i)
        spread[t, j] = np.abs(b[0] - out[f"it{i}_poses"]).max()
    r2 = o.ba_optimize_active_map(poses, pts, ep, el, obs2, fixed, K)
    fin.append((r2[4], r2[5], np.abs(r2[0] - ref[0]).max(), int((r2[3] != ref[3]).sum())))
out.update(self_iters=np.array(ITERS), self_spread=spread, self_final=np.array(fin))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ba_chaotic_window.npz"), **out)
print("rounds", ref[4], "outliers", ref[5], "of", len(ep))
print("one-ulp self spread per iteration count", dict(zip(ITERS, spread.max(0))))
print("one-ulp self final (rounds, outliers, max pose difference, flags that differ)", fin)
