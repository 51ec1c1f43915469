"""Host logic of the package's chain (<pkg>/chain.py — the reference's Frontend / Backend / LoopClosing / Map as one sequential schedule),
run on a CPU through the ORACLE back end (tests/oracle_backend.py): the key-frame rule, the map's bookkeeping, the helpers, the trajectory
writer, and that the committed KITTI-layout trajectory fixture is what its script produces.  No GPU: the HIP back end runs the same chain
in tests/test_gpu_sequence.py and tests/test_gpu_runner.py."""
import os

import numpy as np
import pytest

import kitti_layout
from conftest import ROOT
from oracle_backend import OracleBackend


@pytest.fixture(scope="module")
def short_run(pkg, synth, oracle):
    chain = pkg.chain
    scene = synth.sequence_scene(); C, yaw = synth.sequence_poses(200)
    frames = [synth.render_stereo(scene, C[t], yaw[t], t) for t in range(44)]
    cfg = {"numFeatures.trackingGood": 390}          # ~300 initial + 100 new features per key-frame: the rule inserts 8 key-frames within 44 frames
    c = chain.Chain(OracleBackend(oracle, synth.calc_weights_handcrafted(), cfg, chain), pkg.api, synth.SEQ_K, frames, cfg=cfg).run()
    return c, C, yaw


def test_keyframe_rule_and_map_bookkeeping(short_run, pkg):
    c, _, _ = short_run
    ninl = [int(x[2][0]) for t, x in c.log if t == "pose_only"]
    # Frontend::Track (frontend.cpp:97-120): TRACKING_BAD <=> trackingBad < inliers <= trackingGood -> DetectFeatures ... InsertKeyFrame
    assert c.kf_frames[0] == 0 and [i + 1 for i, v in enumerate(ninl) if 10 < v <= 390] == c.kf_frames[1:] and len(c.kf_frames) >= 8
    assert sorted(c.all_kfs) == list(range(len(c.kf_frames))) and len(c.active_kfs) <= 7
    for mp in c.all_mps.values():                   # every live map point is observed, its observations point back at it, active ones lie in the window
        assert mp.alive and (mp.obs or mp.outlier)
        assert all(f.mp is mp and f.kf is not None for f in mp.obs)
        assert all(f.kf.id in c.active_kfs for f in mp.active_obs)
    for mid, mp in c.active_mps.items():
        assert mid in c.all_mps and mp.active_obs
    kf1 = c.all_kfs[1]
    assert kf1.last_kf is c.all_kfs[0] and kf1.rel_to_last is not None and kf1.img is None          # LoopClosing released the image (bShowResult 0)
    assert len(c.db) == c.be.db_size() == len(c.all_kfs)                                             # no loop: every key-frame reaches the database
    assert c.stats["lk_init_from_projection"] > c.stats["lk_init_from_last"] > 0
    tags = [t for t, _ in c.log]
    i = tags.index("detect", 1)                      # second key-frame: DetectFeatures, FindFeaturesInRight, TriangulateNewPoints, then the back end, then the loop closer
    assert tags[i:i + 5] == ["detect", "lk_right", "triangulate", "ba", "lcd"]


def test_trajectory_writer_and_ate(short_run, pkg, synth, tmp_path):
    c, C, yaw = short_run
    c.save(str(tmp_path))
    lines = open(tmp_path / "trajectory.txt").read().strip().split("\n")
    assert len(lines) == len(c.all_kfs) and open(tmp_path / "loopEdges.txt").read() == ""
    r = np.array([[float(x) for x in l.split()] for l in lines])
    assert r[:, 0].tolist() == list(range(len(lines))) and np.allclose(r[:, 1], c.kf_frames)
    for row, k in zip(r, [c.all_kfs[i] for i in sorted(c.all_kfs)]):
        Twc = pkg.chain.T_inv(pkg.chain.T_of(k.pose))
        assert np.allclose(row[2:5], Twc[:3, 3], atol=1e-6)
    rmse, worst = kitti_layout.ate(pkg.chain, synth, c.poses, C, yaw)
    assert rmse < 0.3, (rmse, worst)


def test_helpers(pkg, oracle):
    chain = pkg.chain
    rng = np.random.default_rng(1)
    for _ in range(50):
        xi = np.concatenate([rng.normal(0, 2, 3), rng.normal(0, 1, 3) * rng.choice([1e-9, 0.3, 1.0, 3.0])])
        p7 = oracle.se3_exp(xi)
        assert abs(chain.se3_log_norm(chain.T_of(p7)) - np.linalg.norm(oracle.se3_log(p7))) < 1e-9
        T = chain.T_of(p7)
        assert np.allclose(chain.T_inv(T) @ T, np.eye(4), atol=1e-12) and np.allclose(chain.T_of(chain.p7_of(T)), T, atol=1e-12)
    K = chain.camera_from_config(kitti_layout.parse_yaml(kitti_layout.KITTI00_02_YAML))
    assert K["fx"] == float(np.float32(718.856)) and K["baseline"] == float(np.float32(np.float32(386.1448) / np.float32(718.856)))      # System::GetCamera works in float


def test_committed_trajectory_fixture_is_what_its_script_writes(pkg, synth, oracle):
    """tests/golden/kitti_layout_200_trajectory.txt: 200 rendered frames at 1241 x 376, the reference's YAML values, the oracle chain"""
    chain = pkg.chain
    cfg = kitti_layout.parse_yaml(kitti_layout.KITTI00_02_YAML)
    frames, C, yaw = kitti_layout.render(synth)
    c = chain.Chain(OracleBackend(oracle, synth.calc_weights_handcrafted(), cfg, chain), pkg.api, chain.camera_from_config(cfg), frames, cfg=cfg,
                    timestamps=[0.1 * t for t in range(len(frames))], log=False).run()
    gold = np.array([[float(x) for x in l.split()] for l in open(os.path.join(ROOT, "tests", "golden", "kitti_layout_200_trajectory.txt")).read().strip().split("\n")])
    kfs = [c.all_kfs[i] for i in sorted(c.all_kfs)]
    assert len(kfs) == len(gold) == 7 and [k.frame_id for k in kfs] == [0, 18, 36, 60, 151, 171, 184]
    for row, k in zip(gold, kfs):
        Twc = chain.T_inv(chain.T_of(k.pose))
        assert abs(row[1] - k.ts) < 1e-6 and np.allclose(row[2:5], Twc[:3, 3], atol=2e-6)
    rmse, worst = kitti_layout.ate(chain, synth, c.poses, C, yaw)
    assert rmse < 0.6                               # 0.445 m over a 120 m path


def test_compiled_host_helpers_match_chain_py(pkg, tmp_path):
    """host/myslam_system.hpp (the C++ twin of chain.py behind bin/run_kitti_stereo): its SE3 helpers against chain.py's numpy ones, and the
    compiled runner builds, links the library and reports usage / a missing config like the reference's main()."""
    import subprocess
    chain = pkg.chain
    exe = str(tmp_path / "system_se3_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "system_se3_test.cpp"), "-o", exe, "-I" + os.path.join(ROOT, "include")])
    rng = np.random.default_rng(5)
    cases = []
    for i in range(200):
        q = rng.normal(size=(2, 4)); t = rng.normal(size=(2, 3)) * rng.choice([1e-3, 1.0, 50.0])
        if i % 10 == 0:
            q[1] = q[0]; t[1] = t[0] + rng.normal(size=3) * 1e-12          # log of (nearly) the identity: the small-angle branch
        if i % 10 == 1:
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = np.pi - 1e-8 * (i % 3)      # relative rotation by (nearly) pi
            A = chain.T_of(np.concatenate([q[0] / np.linalg.norm(q[0]), t[0]]))
            Rel = np.eye(4); Rel[:3, :3] = chain.q_to_R(np.concatenate([np.sin(ang / 2) * ax, [np.cos(ang / 2)]]))
            p = chain.p7_of(chain.T_inv(Rel) @ A); q[1] = p[:4]; t[1] = p[4:]
        cases.append((np.concatenate([q[0], t[0]]), np.concatenate([q[1], t[1]])))
    text = "".join(" ".join(repr(float(x)) for x in a) + "\n" + " ".join(repr(float(x)) for x in b) + "\n" for a, b in cases)
    r = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=60)
    rows = np.array([[float(x) for x in l.split()] for l in r.stdout.strip().split("\n")])
    assert rows.shape == (len(cases), 8)
    for (a, b), row in zip(cases, rows):
        Cm = chain.mm(chain.T_of(a), chain.T_inv(chain.T_of(b)))
        # products, inverses and the quaternion extraction are written in one summation order on both sides: identical bits
        assert np.array_equal(row[:7], chain.p7_of(Cm)), (row, chain.p7_of(Cm))
        ref = chain.se3_log_norm(Cm)                 # acos / sin / cos come from two maths libraries: rounding, not bits
        assert abs(row[7] - ref) <= 1e-9 * max(1.0, ref), (row[7], ref)
    app = pkg._build.build_app()
    u = subprocess.run([app], capture_output=True, text=True)
    assert u.returncode == 1 and "path_to_config path_to_sequence" in u.stderr
    m = subprocess.run([app, str(tmp_path / "missing.yaml"), str(tmp_path)], capture_output=True, text=True)
    assert m.returncode == 1 and "does not exist" in m.stderr
