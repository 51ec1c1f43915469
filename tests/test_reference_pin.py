"""Oracle vs the reference's OWN in-tree functions and g2o's Levenberg iterations — through fixtures that tools/dump_reference_goldens.cpp
writes on a machine where the reference is built (it drives the real libmyslam.so: ORBextractor::DetectAndCompute / Detect /
ScreenAndComputeKPsParams / CalcDescriptors, triangulation(), EdgeProjection, and the optimiser of Backend::OptimizeActiveMap with a
post-iteration hook).  This build environment cannot build the reference (OpenCV 3.4.8, Eigen, Sophus, g2o absent), so the fixtures are
absent and every comparison here reports
    XFAIL  parity unpinned: ...
(an expected failure, not a skip).  With tests/golden/reference/ref_*.npy present the tests compare for real and a mismatch FAILS.
The inputs are regenerated here by the same script the maintainer ran (tools/make_reference_inputs.py: deterministic)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

REF = os.path.join(ROOT, "tests", "golden", "reference")
KP_FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_reference_inputs
    d = str(tmp_path_factory.mktemp("reference_inputs"))
    make_reference_inputs.main(d)
    return d


def ref(name):
    path = os.path.join(REF, f"ref_{name}.npy")
    if not os.path.exists(path):
        pytest.xfail(f"parity unpinned: ref_{name}.npy is not in tests/golden/reference/ (build and run tools/dump_reference_goldens.cpp where the reference is built)")
    return np.load(path, allow_pickle=False)


def kps_rows(k):
    """the repo's key-point structs as the dump program's rows of 7 floats"""
    return np.stack([k[f].astype(np.float32) for f in KP_FIELDS], axis=1)


def test_pin_kit_files_are_numpy_compatible(inputs, tmp_path):
    """The C++ dump program reads and writes plain NPY files with its own 40-line reader / writer: round trip through it (compiled without
    OpenCV: -DMYSLAM_NPY_SELFTEST) for every dtype and rank the kit uses."""
    exe = str(tmp_path / "npy_selftest")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-DMYSLAM_NPY_SELFTEST", os.path.join(ROOT, "tools", "dump_reference_goldens.cpp"), "-o", exe])
    names = ["in_left.npy", "in_K.npy", "in_ba_poses.npy", "in_ba_edge_pose.npy", "in_ba_fixed.npy", "in_tri_points.npy", "in_tri_poses34.npy"]
    subprocess.check_call([exe] + [os.path.join(inputs, n) for n in names])
    for n in names:
        a = np.load(os.path.join(inputs, n)); b = np.load(os.path.join(inputs, n + ".copy.npy"))
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), n


def test_detect_and_compute(oracle, inputs):
    want_k, want_d = ref("dac_kps"), ref("dac_desc")
    L = np.load(os.path.join(inputs, "in_left.npy"))
    k, d = oracle.detect_and_compute(oracle.params(2000), L)
    assert kps_rows(k).tobytes() == want_k.tobytes(), "ORBextractor::DetectAndCompute key-points differ (src/ORBextractor.cpp:922-985)"
    assert np.array_equal(d, want_d), "ORBextractor::DetectAndCompute descriptors differ"


def test_detect_with_mask(oracle, inputs):
    want = ref("det_kps")
    L, M = np.load(os.path.join(inputs, "in_left.npy")), np.load(os.path.join(inputs, "in_mask.npy"))
    k = oracle.detect(oracle.params(300), L, mask=M)
    assert kps_rows(k).tobytes() == want.tobytes(), "ORBextractor::Detect differs (src/ORBextractor.cpp:989-1074)"


def test_screen_and_calc_descriptors(oracle, inputs):
    det, want_k, want_d = ref("det_kps"), ref("screen_kps"), ref("calc_desc")
    L = np.load(os.path.join(inputs, "in_left.npy"))
    from pyoracle import KP_DTYPE
    kin = np.zeros(len(det), KP_DTYPE)
    for i, f in enumerate(KP_FIELDS):
        kin[f] = det[:, i].astype(kin[f].dtype)
    p = oracle.params(300)
    ks = oracle.screen(p, L, kin)
    assert kps_rows(ks).tobytes() == want_k.tobytes(), "ScreenAndComputeKPsParams differs (src/ORBextractor.cpp:1083-1129)"
    assert np.array_equal(oracle.calc_descriptors(p, L, ks), want_d), "CalcDescriptors differs (src/ORBextractor.cpp:1180-1226)"


def test_triangulation(oracle, inputs):
    want_xyz, want_ok = ref("tri_xyz"), ref("tri_ok")
    P34, pts = np.load(os.path.join(inputs, "in_tri_poses34.npy")), np.load(os.path.join(inputs, "in_tri_points.npy"))
    for i in range(len(pts)):
        xyz, ratio = oracle.triangulate(P34.reshape(2, 12), pts[i])
        assert (ratio < 1e-2) == bool(want_ok[i]), f"triangulation(): accept / reject differs for case {i} (include/myslam/algorithm.h:28-31)"
        if want_ok[i]:
            assert np.allclose(xyz, want_xyz[i], rtol=1e-9, atol=1e-9), i


@pytest.mark.parametrize("w", ["ba", "ba_bad"])
def test_edge_projection(oracle, inputs, w):
    err, jxi, jxj = ref(f"{w}_edge_err"), ref(f"{w}_jxi"), ref(f"{w}_jxj")
    a = {n: np.load(os.path.join(inputs, f"in_{w}_{n}.npy")) for n in ("poses", "points", "edge_pose", "edge_point", "obs", "fixed")}
    K = np.load(os.path.join(inputs, "in_K.npy"))
    # the oracle exposes the blocks g2o forms from (e, Jxi, Jxj): H_pl = Jxi^T w Jxj per edge, b_l, chi2 = e^T e — rebuilt here from the dump
    Hpp, Hll, Hpl, bp, bl, chi2 = oracle.ba_build(a["poses"], a["points"], a["edge_pose"], a["edge_point"], a["obs"], np.zeros_like(a["fixed"]), K, delta=1e30)
    assert np.allclose(chi2, (err ** 2).sum(1), rtol=1e-11, atol=1e-12), "EdgeProjection::computeError differs (include/myslam/g2o_types.h:115-122)"
    want_hpl = np.einsum("kri,krj->kij", jxi, jxj)
    assert np.allclose(Hpl, want_hpl, rtol=1e-10, atol=1e-9 * np.abs(want_hpl).max()), "EdgeProjection::linearizeOplus differs (include/myslam/g2o_types.h:124-144)"


@pytest.mark.parametrize("w", ["ba", "ba_bad"])
def test_optimize_active_map_iterates(oracle, inputs, w):
    trace, poses, points, echi, rounds = ref(f"{w}_trace"), ref(f"{w}_poses"), ref(f"{w}_points"), ref(f"{w}_edge_chi2"), ref(f"{w}_rounds")
    a = {n: np.load(os.path.join(inputs, f"in_{w}_{n}.npy")) for n in ("poses", "points", "edge_pose", "edge_point", "obs", "fixed")}
    K = np.load(os.path.join(inputs, "in_K.npy"))
    (gp, gx, gchi, gout, gr, gn), tr = oracle.ba_optimize_active_map_traced(a["poses"], a["points"], a["edge_pose"], a["edge_point"], a["obs"], a["fixed"], K)
    assert gr == int(rounds[0]), "rounds that failed the inlier test differ (src/backend.cpp:212-232)"
    assert len(tr) == len(trace), f"Levenberg iterations: oracle {len(tr)}, g2o {len(trace)}"
    # per iteration: robust chi2 of the last trial, lambda, number of trials (g2o's post-iteration hook)
    assert np.array_equal(tr[:, 3], trace[:, 2]), "Levenberg trials per iteration differ"
    n_tight = min(len(tr), 10)                               # the first round to rounding; later rounds inherit the accumulated difference
    assert np.allclose(tr[:n_tight, 0], trace[:n_tight, 0], rtol=1e-9) and np.allclose(tr[:n_tight, 2], trace[:n_tight, 1], rtol=1e-7)
    if w == "ba":                                            # the well-posed window: final state (the chaotic one is pinned by its early iterates only)
        assert np.allclose(gp, poses, rtol=1e-7, atol=1e-8) and np.allclose(gx, points, rtol=1e-7, atol=1e-7)
        assert np.allclose(gchi, echi, rtol=1e-6, atol=1e-9)


def test_comparisons_execute_on_stand_in_fixtures(oracle, inputs, tmp_path, monkeypatch):
    """The comparisons above cannot run for real here; this runs their CODE on stand-in ref_*.npy files written from the oracle itself (and, for
    EdgeProjection, from the formulas of g2o_types.h:115-144 in numpy) so that a maintainer's first real run fails on a real difference, not
    on a shape or dtype slip in the test.  It pins nothing."""
    import test_reference_pin as T
    d = str(tmp_path)
    monkeypatch.setattr(T, "REF", d)
    L, M = np.load(os.path.join(inputs, "in_left.npy")), np.load(os.path.join(inputs, "in_mask.npy"))
    k, de = oracle.detect_and_compute(oracle.params(2000), L)
    np.save(os.path.join(d, "ref_dac_kps.npy"), kps_rows(k)); np.save(os.path.join(d, "ref_dac_desc.npy"), de)
    p = oracle.params(300)
    kd = oracle.detect(p, L, mask=M)
    np.save(os.path.join(d, "ref_det_kps.npy"), kps_rows(kd))
    ks = oracle.screen(p, L, kd)
    np.save(os.path.join(d, "ref_screen_kps.npy"), kps_rows(ks)); np.save(os.path.join(d, "ref_calc_desc.npy"), oracle.calc_descriptors(p, L, ks))
    P34, pts = np.load(os.path.join(inputs, "in_tri_poses34.npy")), np.load(os.path.join(inputs, "in_tri_points.npy"))
    tri = [oracle.triangulate(P34.reshape(2, 12), pts[i]) for i in range(len(pts))]
    np.save(os.path.join(d, "ref_tri_xyz.npy"), np.stack([t[0] for t in tri])); np.save(os.path.join(d, "ref_tri_ok.npy"), np.array([t[1] < 1e-2 for t in tri], np.uint8))
    assert 0 < int(np.sum([t[1] < 1e-2 for t in tri])) < len(tri)                  # the parallel-ray cases are rejected, the others accepted
    K = np.load(os.path.join(inputs, "in_K.npy"))
    for w in ("ba", "ba_bad"):
        a = {n: np.load(os.path.join(inputs, f"in_{w}_{n}.npy")) for n in ("poses", "points", "edge_pose", "edge_point", "obs", "fixed")}
        q = a["poses"][a["edge_pose"]]                                             # (qx qy qz qw tx ty tz), Tcw
        x, y, z, s = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * s), 2 * (x * z + y * s)], 1),
                      np.stack([2 * (x * y + z * s), 1 - 2 * (x * x + z * z), 2 * (y * z - x * s)], 1),
                      np.stack([2 * (x * z - y * s), 2 * (y * z + x * s), 1 - 2 * (x * x + y * y)], 1)], 1)
        pc = np.einsum("kij,kj->ki", R, a["points"][a["edge_point"]]) + q[:, 4:]
        X, Y, Zi = pc[:, 0], pc[:, 1], 1.0 / (pc[:, 2] + 1e-18)
        fx, fy, cx, cy = K
        err = a["obs"] - np.stack([fx * X * Zi + cx, fy * Y * Zi + cy], 1)
        o = np.zeros_like(X)
        jxi = np.stack([np.stack([-fx * Zi, o, fx * X * Zi ** 2, fx * X * Y * Zi ** 2, -fx - fx * X * X * Zi ** 2, fx * Y * Zi], 1),
                        np.stack([o, -fy * Zi, fy * Y * Zi ** 2, fy + fy * Y * Y * Zi ** 2, -fy * X * Y * Zi ** 2, -fy * X * Zi], 1)], 1)
        jxj = np.einsum("kri,kij->krj", jxi[:, :, :3], R)
        np.save(os.path.join(d, f"ref_{w}_edge_err.npy"), err); np.save(os.path.join(d, f"ref_{w}_jxi.npy"), jxi); np.save(os.path.join(d, f"ref_{w}_jxj.npy"), jxj)
        (gp, gx, gchi, gout, gr, gn), tr = oracle.ba_optimize_active_map_traced(a["poses"], a["points"], a["edge_pose"], a["edge_point"], a["obs"], a["fixed"], K)
        np.save(os.path.join(d, f"ref_{w}_trace.npy"), tr[:, [0, 2, 3]]); np.save(os.path.join(d, f"ref_{w}_poses.npy"), gp); np.save(os.path.join(d, f"ref_{w}_points.npy"), gx)
        np.save(os.path.join(d, f"ref_{w}_edge_chi2.npy"), gchi); np.save(os.path.join(d, f"ref_{w}_rounds.npy"), np.array([gr], np.int32))
        assert gr == (0 if w == "ba" else 5) and len(tr) >= 10
    T.test_detect_and_compute(oracle, inputs); T.test_detect_with_mask(oracle, inputs); T.test_screen_and_calc_descriptors(oracle, inputs)
    T.test_triangulation(oracle, inputs)
    for w in ("ba", "ba_bad"):
        T.test_edge_projection(oracle, inputs, w); T.test_optimize_active_map_iterates(oracle, inputs, w)
