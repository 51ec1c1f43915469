"""The "fast" stand-in sequence (tests/kitti_layout.py VARIANTS: 200 frames at 1241 x 376, 2.5 m per frame along the wall and back) through the
package's chain with the ORACLE back end, on a CPU: the reference's key-frame rule must fire the way it does on KITTI-00 (the reference's sample
run holds 27 key-frames in the first 200 frames; this drive gives 30) — a regression pin of chain.py's host logic on a sequence where local BA
and DeepLCD run 30 times.  The GPU lock-step runs of this and the two longer variants are tests/test_gpu_runner_variants.py."""
import numpy as np

import kitti_layout
from oracle_backend import OracleBackend

FAST_KF_FRAMES = [0, 17, 20, 26, 32, 40, 46, 52, 54, 59, 65, 70, 78, 83, 91, 117, 122, 128, 137, 140, 145, 148, 152, 159, 165, 171, 177, 181, 186, 192]


def test_fast_variant_key_frames_by_the_references_rule(pkg, synth, oracle):
    chain = pkg.chain
    frames, C, yaw = kitti_layout.render_variant(synth, "fast")
    assert len(frames) == 200 and frames[0][0].shape == (376, 1241)
    cfg = kitti_layout.parse_yaml(kitti_layout.KITTI00_02_YAML)
    c = chain.Chain(OracleBackend(oracle, synth.calc_weights_handcrafted(), cfg, chain), pkg.api, chain.camera_from_config(cfg), frames, cfg=cfg,
                    timestamps=[0.1 * t for t in range(len(frames))]).run()
    counts = {}
    for t, _ in c.log:
        counts[t] = counts.get(t, 0) + 1
    ninl = [int(x[2][0]) for t, x in c.log if t == "pose_only"]
    # Frontend::Track (frontend.cpp:97-120): a key-frame exactly where trackingBad < inliers <= trackingGood (10 / 50 from the YAML); never LOST
    assert [i + 1 for i, v in enumerate(ninl) if 10 < v <= 50] == c.kf_frames[1:] and min(ninl) > 10
    assert c.kf_frames == FAST_KF_FRAMES, c.kf_frames
    assert counts["pose_only"] == 199 and counts["ba"] == counts["lcd"] == counts["detect"] == 30 and "detect_loop" not in counts
    rmse, worst = kitti_layout.ate(chain, synth, c.poses, C, yaw)
    rmse_al, rot = kitti_layout.ate_aligned(chain, synth, c.poses, C, yaw)
    path = float(np.sum(np.linalg.norm(np.diff(C, axis=0), axis=1)))
    print(f"fast variant, oracle chain: {len(c.all_kfs)} key-frames, min inliers {min(ninl)}, ATE {rmse:.3f} m anchored at frame 0, {rmse_al:.3f} m after rigid alignment "
          f"(rotation {rot:.2f} deg) over a {path:.0f} m path")
    assert rmse < 0.6 and rmse_al <= rmse + 1e-9


CORRIDOR_KF_FRAMES = [0, 17, 26, 34, 42, 51, 60, 69, 76, 84, 92, 100, 108, 116, 125, 133, 141, 149, 157, 165, 174, 183, 192, 199]


def test_corridor_variant_forward_motion(pkg, synth, oracle, tmp_path):
    """The forward drive through a corridor (tests/kitti_layout.py "corridor": 0.9 m per frame along the optical axis, features stream out of the
    vanishing point, grow and change pyramid level) through the oracle chain: 24 key-frames in 200 frames by the reference's rule (its own
    KITTI-00 run: 27), never LOST, drift below 1 % of the path."""
    chain = pkg.chain
    frames, C, yaw = kitti_layout.render_variant(synth, "corridor")
    cfg = kitti_layout.parse_yaml(kitti_layout.KITTI00_02_YAML)
    c = chain.Chain(OracleBackend(oracle, synth.calc_weights_handcrafted(), cfg, chain), pkg.api, chain.camera_from_config(cfg), frames, cfg=cfg,
                    timestamps=[0.1 * t for t in range(len(frames))]).run()
    ninl = [int(x[2][0]) for t, x in c.log if t == "pose_only"]
    assert [i + 1 for i, v in enumerate(ninl) if 10 < v <= 50] == c.kf_frames[1:] and min(ninl) > 10
    assert c.kf_frames == CORRIDOR_KF_FRAMES, c.kf_frames
    rmse, _ = kitti_layout.ate(chain, synth, c.poses, C, yaw)
    rmse_al, rot = kitti_layout.ate_aligned(chain, synth, c.poses, C, yaw)
    path = float(np.sum(np.linalg.norm(np.diff(C, axis=0), axis=1)))
    print(f"corridor variant, oracle chain: {len(c.all_kfs)} key-frames, min inliers {min(ninl)}, ATE {rmse:.3f} m anchored at frame 0, {rmse_al:.3f} m after rigid "
          f"alignment (rotation {rot:.2f} deg) over a {path:.0f} m path")
    assert rmse < 0.01 * path and rmse_al < rmse
    # the committed fixture is what this chain writes (tests/golden/make_kitti_layout_trajectory.py corridor): same text on a CPU
    import os
    c.save(str(tmp_path))
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_layout_corridor_trajectory.txt")
    assert open(tmp_path / "trajectory.txt").read() == open(gold).read()
