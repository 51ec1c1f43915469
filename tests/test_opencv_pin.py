"""Oracle vs the libraries the reference links (OpenCV 3.4.8, Caffe) — through fixtures that tools/dump_opencv_goldens.py writes on a
machine that has them.  This build environment has neither, so the fixtures are absent and every test here reports
    XFAIL  parity unpinned: ...
(an expected failure, not a skip: the gap stays visible in every test run).  With the fixtures present the tests compare for real
and a mismatch FAILS, naming the definition to change."""
import os

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")


def fixture(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        pytest.xfail(f"parity unpinned: {name} is not in tests/golden/ (run tools/dump_opencv_goldens.py where OpenCV / Caffe exist)")
    return np.load(path, allow_pickle=False)


def test_resize_inter_linear(oracle):
    f = fixture("opencv_resize.npz")
    cur = f["level0"]
    for l in range(1, 8):
        want = f[f"level{l}"]
        cur = oracle.resize(cur, want.shape[1], want.shape[0])
        assert np.array_equal(cur, want), f"cv::resize differs at pyramid level {l} (oracle/orb_oracle.cpp resize_linear, OpenCV {f['version']})"
    assert np.array_equal(oracle.resize(f["src"], 160, 120), f["small"]), "cv::resize to 160x120 differs (CALC input)"


def test_gaussian_blur(oracle):
    f = fixture("opencv_blur.npz")
    for i in range(4):
        got = oracle.blur7(f[f"src{i}"], 0)
        assert np.array_equal(got, f[f"out{i}"]), ("GaussianBlur 7x7 sigma=2 differs: set the taps OpenCV uses with orc_set_gauss_taps / "
                                                   f"myslam_orb_set_gauss_taps; getGaussianKernel(7, 2) * 256 = {f['kernel_sigma2'] * 256}")
    assert np.array_equal(oracle.blur7(f["lcd_src"], 1), f["lcd_out"]), f"GaussianBlur 7x7 sigma=0 differs; kernel * 256 = {f['kernel_sigma0'] * 256}"


def test_fast_score_and_nms(oracle):
    f = fixture("opencv_fast.npz")
    for i in range(12):
        roi = f[f"roi{i}"]
        for th in (20, 7):
            xs, ys, sc = oracle.fast_detect(roi, th)
            want = f[f"roi{i}_th{th}"]
            got = sorted(zip(ys.tolist(), xs.tolist(), sc.tolist()))
            ref = sorted(zip(want[:, 1].astype(int).tolist(), want[:, 0].astype(int).tolist(), want[:, 2].astype(int).tolist()))
            assert got == ref, f"cv::FAST differs on ROI {i} at threshold {th} (score definition / NMS, oracle/orb_oracle.cpp fast_scores)"
    lvl = f["level4"]
    for th in (20, 7):
        want = f[f"level4_nonms_th{th}"]
        m = oracle.fast_score_map(lvl, th)
        ys, xs = np.nonzero(m)
        assert sorted(zip(ys.tolist(), xs.tolist())) == sorted(zip(want[:, 1].astype(int).tolist(), want[:, 0].astype(int).tolist())), "FAST detections differ"


def test_fast_atan2(oracle):
    f = fixture("opencv_atan2.npz")
    got = np.array([oracle.fast_atan2(float(a), float(b)) for a, b in zip(f["y"], f["x"])], np.float32)
    assert np.array_equal(got.view(np.uint32), f["angle"].view(np.uint32)), "cv::fastAtan2 differs (polynomial / epsilon, oracle/orb_oracle.cpp fast_atan2)"


def test_bfmatcher_hamming(oracle):
    f = fixture("opencv_hamming.npz")
    idx, dist = oracle.hamming_match(f["query"], f["train"])
    assert np.array_equal(f["query_idx"], np.arange(len(idx))) and np.array_equal(idx, f["train_idx"]) and np.array_equal(dist, f["dist"].astype(np.int32))


def test_pyr_lk(oracle):
    f = fixture("opencv_lk.npz")
    for name in ("next", "right"):
        out, st, err = oracle.lk_track(f["prev"], f[name], f["pts"], f["pts"])
        assert np.array_equal(st, f[name + "_status"].astype(bool)), "calcOpticalFlowPyrLK status differs"
        ok = st
        # OpenCV accumulates its window sums in f32 in a build-dependent order; the restatement uses exact integer sums (DESIGN.md
        # section 5): positions agree to a small fraction of a pixel, not bitwise
        assert np.abs(out[ok] - f[name + "_pts"][ok]).max() < 0.05, "calcOpticalFlowPyrLK positions differ by more than the f32 accumulation noise"


def test_solve_pnp_ransac(oracle):
    f = fixture("opencv_pnp.npz")
    rc, pose, inl, n = oracle.solve_pnp_ransac(f["pts3d"], f["pts2d"], tuple(f["K"]))
    assert (rc == 0) == bool(f["ok"])
    want = np.zeros(len(inl), bool); want[f["inliers"].astype(int)] = True
    assert np.array_equal(inl, want), "solvePnPRansac consensus set differs (RNG stream / EPnP / bookkeeping, oracle/pnp_oracle.cpp)"


def test_calc_forward(oracle, pkg):
    f = fixture("caffe_calc.npz")
    proto, model = os.path.join(GOLD, "calc_model", "deploy.prototxt"), os.path.join(GOLD, "calc_model", "calc.caffemodel")
    if not (os.path.exists(proto) and os.path.exists(model)):
        pytest.xfail("parity unpinned: tests/golden/calc_model/{deploy.prototxt,calc.caffemodel} not present")
    L, w = pkg.api.calc_parse_caffe(proto, model)
    got = oracle.calc_forward_net(L, w, f["input"])
    want = f["output"] / np.linalg.norm(f["output"])
    assert np.abs(got - want).max() < 2e-5, "CALC forward differs from Caffe (layer list / LRN / pooling mode, oracle/calc_oracle.cpp)"
