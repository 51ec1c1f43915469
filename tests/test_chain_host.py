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
    assert len(lines) == len(c.all_kfs) and open(tmp_path / "loop_edges.txt").read() == ""
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
