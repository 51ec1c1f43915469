"""tools/run_kitti_stereo.py — BASELINE configs[0]'s entry point — on rendered stereo sequences written in KITTI layout (times.txt,
image_0 / image_1/%06d.png, a config in the reference's YAML form): plumbing from files to trajectory.txt.
  * 200 frames at 1241 x 376 with the KITTI00-02 intrinsics and the reference's YAML values, key-frames by the reference's rule: the
    runner end to end, the HIP chain against the oracle chain entry by entry, the trajectory against the committed fixture;
  * the same 200 frames through the COMPILED runner (app/run_kitti_stereo.cpp over host/myslam_system.hpp, plain C++ above the C ABI):
    the same key-frames, frame poses and trajectory file as the Python chain;
  * 60 frames at 720 x 240 with a key-frame every 6th frame (`--kf-every`).
(The KITTI data itself is not available here: configs[0] proper stays untested.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

import kitti_layout
import png_files
from chain_compare import compare_runs
from conftest import ROOT
from oracle_backend import CheckedBackend, OracleBackend

pytestmark = pytest.mark.gpu


def _run(cfg, seq, n, out, *extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_kitti_stereo.py"), str(cfg), str(seq), "--frames", str(n), "--out", str(out), *extra],
                       capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def _rows(path):
    return np.array([[float(x) for x in l.split()] for l in open(path).read().strip().split("\n")])


@pytest.fixture(scope="module")
def kitti_seq(synth, tmp_path_factory):
    """the rendered 200-frame sequence in KITTI layout + the reference's YAML, written once for the tests of this module"""
    d = tmp_path_factory.mktemp("kitti")
    frames, C, yaw = kitti_layout.render(synth)
    seq = d / "sequences" / "00"
    ts = kitti_layout.write(str(seq), frames, png_files)
    cfg_path = d / "KITTI00-02.yaml"; cfg_path.write_text(kitti_layout.KITTI00_02_YAML)
    return dict(frames=frames, C=C, yaw=yaw, seq=seq, ts=ts, cfg_path=cfg_path)


def test_runner_200_frames_at_kitti_resolution(api, oracle, synth, pkg, tmp_path, kitti_seq):
    chain = pkg.chain
    frames, C, yaw, seq, ts, cfg_path = (kitti_seq[k] for k in ("frames", "C", "yaw", "seq", "ts", "cfg_path"))
    n = len(frames)
    assert n == 200 and frames[0][0].shape == (376, 1241)
    out = tmp_path / "result"
    stdout = _run(cfg_path, seq, n, out)
    assert "200 frames (1241x376)" in stdout and "frames/s" in stdout
    rows = _rows(out / "trajectory.txt")
    # the same sequence through the package's chain in this process: HIP back end and oracle back end
    cfg = kitti_layout.parse_yaml(kitti_layout.KITTI00_02_YAML)
    K = chain.camera_from_config(cfg)
    w = synth.calc_weights_handcrafted()
    # 1. lock-step: every operator call of the HIP chain repeated by the oracle on the same inputs (asserts inside); 2. the oracle chain free-running
    chk = CheckedBackend(chain.HipBackend(api, w, cfg), OracleBackend(oracle, w, cfg, chain))
    a = chain.Chain(chk, pkg.api, K, frames, cfg=cfg, timestamps=ts).run()
    b = chain.Chain(OracleBackend(oracle, w, cfg, chain), pkg.api, K, frames, cfg=cfg, timestamps=ts).run()
    assert len(a.all_kfs) == len(rows)
    rep = compare_runs(a, b)
    assert rep["same_key_frames"] and rep["tracks_within_0.03px"] >= 0.99 * rep["tracks"]
    counts = {}
    for t, _ in a.log:
        counts[t] = counts.get(t, 0) + 1
    assert counts["pose_only"] == n - 1 and counts["ba"] == counts["lcd"] == len(a.all_kfs)
    # key-frames by the reference's rule (frontend.cpp:97-120): exactly where the pose-only inlier count fell to trackingGood = 50 or below
    ninl = [int(x[2][0]) for t, x in a.log if t == "pose_only"]
    assert [i + 1 for i, v in enumerate(ninl) if 10 < v <= 50] == a.kf_frames[1:] and min(ninl) > 10
    assert a.stats["lk_init_from_projection"] > 10 * a.stats["lk_init_from_last"] > 0          # LK starts from the re-projection (frontend.cpp:136-147)
    assert "detect_loop" not in counts and chk.db_size() == len(a.all_kfs)                    # the gate of 50 key-frames never opens (as on KITTI-00's first 200 frames)
    # the runner's file == this process's HIP chain, and both == the committed fixture (written by the ORACLE chain on a CPU,
    # tests/golden/make_kitti_layout_trajectory.py) up to the 6 printed decimals
    a.save(str(tmp_path / "again"))
    assert open(out / "trajectory.txt").read() == open(tmp_path / "again" / "trajectory.txt").read()
    assert open(out / "loopEdges.txt").read() == ""
    gold = _rows(os.path.join(ROOT, "tests", "golden", "kitti_layout_200_trajectory.txt"))
    assert gold.shape == rows.shape and np.array_equal(gold[:, :2], rows[:, :2])
    dev_gold = float(np.abs(gold[:, 2:] - rows[:, 2:]).max())
    assert dev_gold <= 0.5, dev_gold          # two free runs (fixture: the oracle chain): the un-anchored BA gauge, see chain_compare.compare_runs
    rmse, worst = kitti_layout.ate(chain, synth, a.poses, C, yaw)
    rmse_o, _ = kitti_layout.ate(chain, synth, b.poses, C, yaw)
    path_len = float(np.sum(np.linalg.norm(np.diff(C, axis=0), axis=1)))
    print(f"configs[0] stand-in: {n} frames 1241x376, {len(a.all_kfs)} key-frames at frames {a.kf_frames} (the reference's rule, thresholds 50 / 10), "
          f"{len(a.all_mps)} map points; lock-step: {sum(chk.calls.values())} operator calls checked on identical inputs, largest deviations "
          f"{({k: float(f'{v:.2e}') for k, v in chk.dev.items()})}; free run HIP vs oracle chain: {rep}; {a.stats['lk_init_from_projection']} LK starts from a "
          f"re-projection; trajectory.txt vs the committed fixture: max deviation {dev_gold:.3e}; ATE rmse {rmse:.3f} m (oracle chain {rmse_o:.3f} m), worst {worst:.3f} m over a {path_len:.0f} m path; {stdout.strip().splitlines()[-1]}")
    assert rmse < 1.5 and rmse_o < 1.5 and abs(rmse - rmse_o) < 0.25


def test_compiled_runner_equals_the_python_chain(api, synth, pkg, tmp_path, kitti_seq):
    """bin/run_kitti_stereo (C++17 over the C ABI: host/myslam_system.hpp is Frontend / Backend / LoopClosing / Map, the operators are the
    facade's) on the 200-frame sequence: the same operator calls in the same order as chain.py, so the same key-frames, the same pose of
    every frame bit for bit and the same trajectory.txt."""
    import numpy as np
    chain = pkg.chain
    exe = pkg._build.build_app()
    frames, seq, ts, cfg_path = (kitti_seq[k] for k in ("frames", "seq", "ts", "cfg_path"))
    w = np.ascontiguousarray(synth.calc_weights_handcrafted(), np.float32).ravel()
    wfile = tmp_path / "handcrafted.calcw"
    with open(wfile, "wb") as f:
        f.write(b"CALCW1\0\0"); f.write(np.uint64(w.size).tobytes()); f.write(w.tobytes())
    out = tmp_path / "cpp"
    r = subprocess.run([exe, str(cfg_path), str(seq), "--frames", str(len(frames)), "--out", str(out), "--calc-weights", str(wfile), "--frame-poses"],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "200 frames (1241x376)" in r.stdout
    cfg = kitti_layout.parse_yaml(kitti_layout.KITTI00_02_YAML)
    a = chain.Chain(chain.HipBackend(api, w, cfg), pkg.api, chain.camera_from_config(cfg), frames, cfg=cfg, timestamps=ts, log=False).run()
    a.save(str(tmp_path / "py"))
    kf_frames = [int(x) for x in open(out / "key_frame_frames.txt").read().split()]
    assert kf_frames == a.kf_frames
    poses = _rows(out / "frame_poses_cw.txt")
    ref = np.stack(a.poses)
    assert poses.shape == ref.shape
    # both hosts write every sum in one order (chain.py's mm / mv, the header's loops) and call the same operators on the same bytes: the pose
    # of every frame is the same double (%.17g round-trips), the files are the same text
    assert np.array_equal(poses, ref), float(np.abs(poses - ref).max())
    assert open(out / "trajectory.txt").read() == open(tmp_path / "py" / "trajectory.txt").read()
    assert open(out / "loopEdges.txt").read() == ""
    print("compiled runner: " + " ".join(l for l in r.stdout.splitlines() if l.startswith("per tracked frame")))
    print(f"compiled runner: {r.stdout.strip().splitlines()[-1]}; key-frames at frames {kf_frames}; the pose of every frame and trajectory.txt "
          f"are bit-identical to the Python chain's")


def test_compiled_runner_closes_the_loop_like_the_python_chain(api, synth, pkg, tmp_path):
    """The sequence of tests/test_gpu_sequence.py (720 x 240, a key-frame every 6th frame, database gate 25, a loop back to key-frame 0 that
    goes through matching, PnP, pose refinement, LoopLocalFusion and the pose graph) through bin/run_kitti_stereo: the compiled host's loop
    closer makes the same calls as chain.py's — same loop edge, same poses, same files."""
    chain = pkg.chain
    exe = pkg._build.build_app()
    n = 200
    scene = synth.sequence_scene(); C, yaw = synth.sequence_poses(n)
    frames = [synth.render_stereo(scene, C[t], yaw[t], t) for t in range(n)]
    seq = tmp_path / "sequences" / "00"
    ts = kitti_layout.write(str(seq), frames, png_files)
    K = synth.SEQ_K
    yaml = ("%YAML:1.0\n" + "".join(f"Camera.{s}.{k}: {K[k]!r}\n" for s in ("left", "right") for k in ("fx", "fy", "cx", "cy")) + f"Camera.bf: {K['bf']!r}\n"
            "Map.activeMap.size: 7\nLCD.nDatabaseMinSize: 25\nLCD.similarityScoreThreshold.high: 0.94\nLCD.similarityScoreThreshold.low: 0.92\n")
    cfg_path = tmp_path / "cam.yaml"; cfg_path.write_text(yaml)
    w = np.ascontiguousarray(synth.calc_weights_handcrafted(), np.float32).ravel()
    wfile = tmp_path / "handcrafted.calcw"
    with open(wfile, "wb") as f:
        f.write(b"CALCW1\0\0"); f.write(np.uint64(w.size).tobytes()); f.write(w.tobytes())
    out = tmp_path / "cpp"
    r = subprocess.run([exe, str(cfg_path), str(seq), "--out", str(out), "--calc-weights", str(wfile), "--kf-every", "6", "--correct-threshold", "0", "--frame-poses"],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    cfg = kitti_layout.parse_yaml(yaml)
    a = chain.Chain(chain.HipBackend(api, w, cfg), pkg.api, chain.camera_from_config(cfg), frames, cfg=cfg, kf_every=6, correct_threshold=0.0,
                    timestamps=ts, log=False).run()
    a.save(str(tmp_path / "py"))
    assert [(x.id, y.id) for x, y in a.loops] == [(33, 0)] and "34 key-frames" in r.stdout and " 1 loops" in r.stdout
    poses = _rows(out / "frame_poses_cw.txt")
    assert np.array_equal(poses, np.stack(a.poses)), float(np.abs(poses - np.stack(a.poses)).max())
    for name in ("trajectory.txt", "loopEdges.txt"):
        assert open(out / name).read() == open(tmp_path / "py" / name).read(), name
    assert len(open(out / "loopEdges.txt").read().strip().split("\n")) == 2          # the current key-frame's line, then the loop key-frame's
    print(f"compiled runner, loop sequence: {r.stdout.strip().splitlines()[-1]}; every frame pose, trajectory.txt and loopEdges.txt bit-identical to chain.py's")


def test_runner_on_a_rendered_kitti_layout_sequence(api, synth, pkg, tmp_path):
    chain = pkg.chain
    n = 60
    scene = synth.sequence_scene(); C, yaw = synth.sequence_poses(200)
    seq = tmp_path / "sequences" / "00"
    kitti_layout.write(str(seq), [synth.render_stereo(scene, C[t], yaw[t], t) for t in range(n)], png_files)
    K = synth.SEQ_K
    cfg = tmp_path / "cam.yaml"
    cfg.write_text("%YAML:1.0\n# rendered 720 x 240 camera\n" + "".join(f"Camera.{s}.{k}: {K[k]}\n" for s in ("left", "right") for k in ("fx", "fy", "cx", "cy")) +
                   f"Camera.bf: {K['bf']}\nCamera.bNeedUndistortion: 0\nMap.activeMap.size: 7\nLCD.nDatabaseMinSize: 50\n"
                   "LCD.similarityScoreThreshold.high: 0.94\nLCD.similarityScoreThreshold.low: 0.92\n")
    out = tmp_path / "result"
    stdout = _run(cfg, seq, n, out, "--kf-every", "6")
    rows = _rows(out / "trajectory.txt")
    assert len(rows) == (n - 1) // 6 + 1                                     # a key-frame every 6th frame
    assert rows[:, 0].tolist() == list(range(len(rows))) and np.allclose(rows[:, 1], [0.1 * 6 * i for i in range(len(rows))], atol=1e-6)
    assert open(out / "loopEdges.txt").read() == ""                          # the gate of 50 key-frames never opens on 60 frames
    # the written camera centres (Twc translation) against the rendered path, both in the frame of camera 0
    T0 = chain.T_of(synth.pose7_from_twc(C[0], yaw[0]))
    gt = np.array([chain.T_inv(chain.T_of(synth.pose7_from_twc(C[6 * i], yaw[6 * i])) @ chain.T_inv(T0))[:3, 3] for i in range(len(rows))])
    err = np.linalg.norm(rows[:, 2:5] - gt, axis=1)
    assert err.max() < 0.6, err                                               # the reference's un-anchored BA gauge moves the first windows
    assert np.allclose(np.linalg.norm(rows[:, 5:9], axis=1), 1.0, atol=1e-5)
    assert "frames/s" in stdout
