"""Composition parity (the stand-in for BASELINE configs[0], KITTI-00 first 200 pairs — neither the data set nor a reference build
exists here): a synthetic 200-frame stereo sequence with known camera poses is tracked, mapped and loop-closed through the whole
operator chain (tests/sequence_chain.py) twice — through the HIP library and through the CPU oracle — and the two logs must agree
entry by entry: key-points, LK tracks, outlier flags, descriptors, matches and consensus sets identically; landmarks and SE3 poses
within 1e-6; the pose graph within the bars of its operator test.  The estimated trajectory is also compared with the ground truth."""
import numpy as np
import pytest

import sequence_chain as sc

pytestmark = pytest.mark.gpu

N_FRAMES = 200
EXACT = {"detect", "lk_right", "lk_track", "loop_match"}


def _frames(synth, n):
    scene = synth.sequence_scene()
    C, yaw = synth.sequence_poses(n)
    return [synth.render_stereo(scene, C[t], yaw[t], t) for t in range(n)], C, yaw


def _ate(poses7, C, yaw, synth):
    """RMSE of the camera centres against the ground truth, both expressed in the frame of camera 0"""
    T0 = sc.T_of(synth.pose7_from_twc(C[0], yaw[0]))
    est = np.array([np.linalg.inv(sc.T_of(p))[:3, 3] for p in poses7])
    gt = np.array([np.linalg.inv(sc.T_of(synth.pose7_from_twc(C[t], yaw[t])) @ np.linalg.inv(T0))[:3, 3] for t in range(len(poses7))])
    return float(np.sqrt(np.mean(np.sum((est - gt) ** 2, axis=1)))), float(np.abs(est - gt).max())


def test_sequence_chain_hip_equals_oracle(api, oracle, synth, pkg):
    frames, C, yaw = _frames(synth, N_FRAMES)
    w = synth.calc_weights_handcrafted()         # a non-degenerate CALC-shaped model: the loop must be DETECTED by the rule, not forced
    K = synth.SEQ_K
    a = sc.Chain(sc.HipBackend(api, w), pkg.api, K, frames).run()
    b = sc.Chain(sc.OracleBackend(oracle, w), pkg.api, K, frames).run()
    assert len(a.log) == len(b.log) and len(a.kfs) == len(b.kfs) == (N_FRAMES - 1) // 6 + 1
    counts = {}
    for (ta, xa), (tb, xb) in zip(a.log, b.log):
        assert ta == tb and len(xa) == len(xb), (ta, tb)
        counts[ta] = counts.get(ta, 0) + 1
        for i, (u, v) in enumerate(zip(xa, xb)):
            assert u.shape == v.shape, (ta, counts[ta], i, u.shape, v.shape)
            if ta == "pgo" and i == 2:
                continue        # iterations done: at the rounding floor of chi2 Levenberg gives up at a noise-dependent iteration
            if ta in EXACT or u.dtype.kind in "biuV" or u.dtype.names:
                assert u.tobytes() == v.tobytes(), f"{ta} #{counts[ta]} output {i} differs"
            elif ta == "lcd":
                assert np.abs(u - v).max() < 2e-5, (ta, counts[ta], i)
            elif ta == "pgo":
                tol = 5e-4 if i == 0 else max(1e-9, 1e-3 * abs(float(v.ravel()[0])))
                assert np.abs(u - v).max() <= tol, (ta, i, np.abs(u - v).max())
            elif ta in ("correct_points", "local_fusion"):
                assert np.abs(u - v).max() < 5e-3, (ta, np.abs(u - v).max())
            else:                                                   # poses, landmarks, chi2 values
                assert np.allclose(u, v, rtol=1e-6, atol=1e-6), (ta, counts[ta], i, np.abs(u - v).max())
    assert counts["pose_only"] == N_FRAMES - 1 and counts["ba"] == len(a.kfs) - 1 and counts["lcd"] == len(a.kfs) and counts["local_fusion"] == 1
    assert a.n_loop_matches >= 10                                   # loopclosing.cpp:245: the loop is only closed with >= 10 3D-2D matches
    # DetectLoop by its own rule (src/loopclosing.cpp:124-161 + the database gate of :62): the only accepted candidate of the whole
    # sequence is (last key-frame, key-frame 0) — the camera is back at its start — and it is what closed the loop above
    assert a.detected == b.detected == [(len(a.kfs) - 1, 0)], (a.detected, b.detected)
    lcd = [x for t, x in a.log if t == "lcd"]
    best = [int(x[3][0]) for x in lcd]; cnt = [int(x[3][1]) for x in lcd]; score = [float(x[4][0]) for x in lcd]
    assert best[-1] == 0 and score[-1] >= 0.97 and cnt[-1] <= 3                     # the revisit: far above the 0.94 threshold
    gate = a.lcd_min_db
    assert max(score[gate + 1:-1]) < 0.94                                           # no other key-frame behind the gate is accepted ...
    assert any(0.92 < s_ < 0.94 for s_ in score[gate + 1:-1])                       # ... although some are "suspected" (> 0.92): both thresholds act
    assert np.median(score[8:gate]) < 0.90                                          # unrelated places score low (N(0, 1/fan_in) weights: 0.99 everywhere)
    # the scan's cut-off: the five youngest key-frames (ids spaced by 3, cur - id < 20) are never candidates, although they look most alike
    assert all(b_ // 3 <= i - 5 for i, b_ in enumerate(best) if score[i] > 0)
    for pa, pb in zip(a.poses, b.poses):
        assert np.allclose(pa, pb, rtol=1e-6, atol=1e-6)
    rmse, worst = _ate(a.poses, C, yaw, synth)
    rmse_o, _ = _ate(b.poses, C, yaw, synth)
    print(f"sequence: {N_FRAMES} frames, {len(a.kfs)} key-frames, {len(a.points)} landmarks, {a.n_loop_matches} loop matches; "
          f"ATE rmse {rmse:.4f} m (oracle chain {rmse_o:.4f} m), worst {worst:.4f} m over a {float(np.abs(C).max()):.1f} m excursion")
    # the reference's local BA optimises left-camera reprojections only and fixes no key-frame (backend.cpp:139-177): until the window
    # slides past the first key-frames the rigid gauge of every solve is free, and each of the first six solves moves the whole window
    # (key-frame 0 included) by 5-13 cm — a property of the reference algorithm that both chains share; the bar here is "tracking did
    # not break" + equality of the chains.  test_sequence_gauge_anchored shows what is left once that motion is taken out.
    assert rmse < 1.0 and abs(rmse - rmse_o) < 1e-5


def test_sequence_gauge_anchored(api, synth, pkg):
    """The same chain through the HIP library with the DIAGNOSTIC gauge anchor (sequence_chain.Chain(anchor_gauge=True): after every
    local BA the window is moved back rigidly so that its oldest key-frame keeps its pose): the trajectory error that remains is the
    composition's own — centimetres on a 13 m track —, which is what says that the operators compose into a working tracker."""
    frames, C, yaw = _frames(synth, N_FRAMES)
    a = sc.Chain(sc.HipBackend(api, synth.calc_weights_handcrafted()), pkg.api, synth.SEQ_K, frames, anchor_gauge=True).run()
    rmse, worst = _ate(a.poses, C, yaw, synth)
    print(f"sequence, gauge anchored: ATE rmse {rmse:.4f} m, worst {worst:.4f} m; {len(a.kfs)} key-frames, {a.n_loop_matches} loop matches")
    assert rmse < 0.12 and worst < 0.25 and a.n_loop_matches >= 10
