"""The oracle must keep reproducing the committed golden fixtures (CPU); the HIP path must too (GPU)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_reproduces_golden(oracle, synth):
    g = np.load(os.path.join(G, "orb_small.npz"))
    kps, desc = oracle.detect_and_compute(oracle.params(int(g["nfeatures"])), g["image"])
    assert kps.tobytes() == g["kps"].tobytes() and np.array_equal(desc, g["desc"])
    assert oracle.detect(oracle.params(100), g["image"]).tobytes() == g["detect100"].tobytes()
    assert np.array_equal(g["image"], synth.random_image(4242, 240, 320))          # the generator is part of the contract
    h = np.load(os.path.join(G, "hamming_small.npz"))
    idx, dist = oracle.hamming_match(h["q"], h["t"])
    assert np.array_equal(idx, h["idx"]) and np.array_equal(dist, h["dist"])
    c = np.load(os.path.join(G, "calc_small.npz"))
    assert np.abs(oracle.calc_forward(synth.calc_weights(int(c["weights_seed"])), c["x"]) - c["descr"]).max() < 1e-7
    b = np.load(os.path.join(G, "ba_small.npz"))
    got = oracle.ba_build(b["poses"], b["pts"], b["ep"], b["el"], b["obs"], b["fixed"], tuple(b["K"]))
    for a, name in zip(got, ["Hpp", "Hll", "Hpl", "bp", "bl", "chi2"]):
        assert np.allclose(a, b[name], rtol=1e-13, atol=1e-9), name
    k = np.load(os.path.join(G, "lk_small.npz"))
    out, st, err = oracle.lk_track(k["prev"], k["next"], k["pts"], k["pts"])
    assert np.array_equal(out, k["out"]) and np.array_equal(st, k["status"]) and np.array_equal(err, k["err"])
    q = np.load(os.path.join(G, "pose_only_small.npz"))
    pose, outl, inl = oracle.pose_only_optimize(q["pose0"], q["pts3d"], q["obs"], tuple(q["K"]))
    assert np.allclose(pose, q["pose"], rtol=1e-12, atol=1e-12) and np.array_equal(outl, q["outlier"]) and inl == int(q["inliers"])
    _check_pgo(oracle.pose_graph_optimize, synth, 1e-9)
    g = np.load(os.path.join(G, "pnp_small.npz"))
    rc, pose, inl, ni = oracle.solve_pnp_ransac(g["pts3d"], g["pts2d"], tuple(g["K"]))
    assert rc == 0 and np.array_equal(inl, g["inlier"]) and ni == int(g["n_inliers"]) and np.abs(pose - g["pose"]).max() < 1e-10


def _check_pgo(fn, synth, tol):
    """tol: the numeric-Jacobian noise floor of the operator at 40 key-frames is ~1e-6 (tests/test_gpu_pgo.py); the oracle on the
    same libm reproduces itself far below that."""
    p = np.load(os.path.join(G, "pgo_small.npz"))
    assert np.array_equal(p["poses"], synth.pose_graph(40, 1, seed=0x60)[0])
    for its, key in ((3, "3"), (20, "20")):
        poses, chi, done = fn(p["poses"], p["fixed"], p["e0"], p["e1"], p["meas"], iters=its)
        assert np.abs(poses - p["poses" + key]).max() < tol and abs(chi - float(p["chi" + key])) <= 1e3 * tol * float(p["chi" + key])
    assert done == int(p["its20"])


@pytest.mark.gpu
def test_hip_reproduces_golden(api, synth):
    g = np.load(os.path.join(G, "orb_small.npz"))
    kps = api.ORBextractor(100).Detect(g["image"])
    assert kps.tobytes() == g["detect100"].tobytes()
    h = np.load(os.path.join(G, "hamming_small.npz"))
    idx, dist = api.hamming_match(h["q"], h["t"])
    assert np.array_equal(idx, h["idx"]) and np.array_equal(dist, h["dist"])
    c = np.load(os.path.join(G, "calc_small.npz"))
    lcd = api.DeepLCD(synth.calc_weights(int(c["weights_seed"])))
    assert np.abs(lcd.debug_forward(c["x"], 4) - c["descr"]).max() < 2e-5
    b = np.load(os.path.join(G, "ba_small.npz"))
    got = api.ba_build(b["poses"], b["pts"], b["ep"], b["el"], b["obs"], b["fixed"], tuple(b["K"]))
    for a, name in zip(got, ["Hpp", "Hll", "Hpl", "bp", "bl", "chi2"]):
        assert np.allclose(a, b[name], rtol=1e-10, atol=1e-7), name
    k = np.load(os.path.join(G, "lk_small.npz"))
    out, st, err = api.LKTracker().track(k["prev"], k["next"], k["pts"], k["pts"])
    assert np.array_equal(out, k["out"]) and np.array_equal(st, k["status"]) and np.array_equal(err, k["err"])
    q = np.load(os.path.join(G, "pose_only_small.npz"))
    pose, outl, inl = api.pose_only_optimize(q["pose0"], q["pts3d"], q["obs"], tuple(q["K"]))
    assert np.allclose(pose, q["pose"], rtol=1e-8, atol=1e-9) and np.array_equal(outl, q["outlier"]) and inl == int(q["inliers"])
    _check_pgo(api.pose_graph_optimize, synth, 2e-5)
    g = np.load(os.path.join(G, "pnp_small.npz"))
    pose, inl, ni = api.solve_pnp_ransac(g["pts3d"], g["pts2d"], tuple(g["K"]))
    assert np.array_equal(inl, g["inlier"]) and ni == int(g["n_inliers"]) and np.abs(pose - g["pose"]).max() < 1e-9


def test_chaotic_window_fixture_is_the_oracle_and_is_chaotic(oracle):
    """tests/golden/ba_chaotic_window.npz (make_ba_chaotic_window.py): the fixture is what the oracle computes, every round of the solve
    fails on it, and the oracle run on one-ulp-different observations departs from itself ten-fold per Levenberg iteration — the property
    the GPU test (tests/test_gpu_ba.py::test_chaotic_window_is_pinned) and the BA fuzzer's acceptance rule rest on."""
    d = np.load(os.path.join(G, "ba_chaotic_window.npz"))
    args = (d["poses"], d["pts"], d["ep"], d["el"], d["obs"], d["fixed"], tuple(d["K"]))
    ref = oracle.ba_optimize_active_map(*args)
    assert ref[4] == int(d["ref_rounds"]) == 5 and ref[5] == int(d["ref_nout"]) and np.array_equal(ref[3], d["ref_out"])
    assert ref[5] > len(d["ep"]) // 2                                    # mostly outliers: no round can reach the 50 % inlier bar
    a = oracle.ba_optimize(*args, iters=3)
    assert np.abs(a[0] - d["it3_poses"]).max() < 1e-13 and a[3] == 3
    sp = d["self_spread"].max(0)                                         # one-ulp self spread after 1, 2, 3, 5, 10 iterations
    assert sp[0] < 1e-13 and sp[4] > 1e-8 and sp[4] > 1e5 * sp[0]        # rounding noise -> visible in ten iterations
    assert all(int(r) == 5 for r in d["self_final"][:, 0]) and d["self_final"][:, 2].min() > 1e-3      # finals 0.4 apart, all rounds fail every time


def test_weak_frame_fixture_is_the_oracle_and_has_two_branches(oracle):
    """tests/golden/pose_only_weak_frame.npz (make_pose_only_weak_frame.py): a 39-match frame of the `one_way` drive on which one Levenberg accept /
    reject decision of EstimateCurrentPose hangs on the last bits of a sum — the fixture is what the oracle computes, and the oracle run on
    observations one ulp away lands either on the same pose (1e-10) or 2.7e-5 m beside it, flags unchanged.  The lock-step rule of
    tests/oracle_backend.py::CheckedBackend.pose_only and tests/test_gpu_pose_only.py::test_weak_frame rest on this."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_pose_only_weak_frame", os.path.join(G, "make_pose_only_weak_frame.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    d = np.load(os.path.join(G, "pose_only_weak_frame.npz"))
    Kt, pre = tuple(float(x) for x in d["K"]), int(d["pre"])
    rp, ro, ri = oracle.pose_only_optimize(d["pose"], d["p3"], d["obs"], Kt, pre_optimize=pre)
    assert np.abs(rp - d["ref_pose"]).max() < 1e-12 and np.array_equal(ro, d["ref_outlier"]) and ri == int(d["ref_inliers"]) == 34 and len(d["p3"]) == 39
    runs = mk.one_ulp_runs(oracle, d["pose"], d["p3"], d["obs"], Kt, pre)
    dist = np.array([np.abs(q[0] - rp).max() for q in runs])
    assert all(np.array_equal(q[1], ro) and q[2] == ri for q in runs)
    assert (dist < 1e-7).any() and (dist > 2e-5).any() and ((dist < 1e-7) | ((dist > 2e-5) & (dist < 4e-5))).all(), dist       # two branches, nothing between
    assert np.abs(np.stack([q[0] for q in runs]) - d["ulp_poses"]).max() < 1e-9
