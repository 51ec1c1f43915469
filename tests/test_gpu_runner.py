"""tools/run_kitti_stereo.py — BASELINE configs[0]'s entry point — on a rendered stereo sequence written in KITTI layout (times.txt,
image_0 / image_1/%06d.png, a config in the reference's YAML form): plumbing from files to trajectory.txt, checked against the known
camera path.  (The KITTI data itself is not available here: configs[0] proper stays untested.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

import png_files
import sequence_chain as sc
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_runner_on_a_rendered_kitti_layout_sequence(api, synth, tmp_path):
    n = 60
    scene = synth.sequence_scene(); C, yaw = synth.sequence_poses(200)
    seq = tmp_path / "sequences" / "00"; (seq / "image_0").mkdir(parents=True); (seq / "image_1").mkdir()
    for t in range(n):
        L, R = synth.render_stereo(scene, C[t], yaw[t], t)
        png_files.write_png_gray(str(seq / "image_0" / f"{t:06d}.png"), L, filters=True)
        png_files.write_png_gray(str(seq / "image_1" / f"{t:06d}.png"), R, filters=True)
    (seq / "times.txt").write_text("".join(f"{0.1 * t:.6e}\n" for t in range(n)))
    K = synth.SEQ_K
    cfg = tmp_path / "cam.yaml"
    cfg.write_text("%YAML:1.0\n# rendered 720 x 240 camera\n" + "".join(f"Camera.{s}.{k}: {K[k]}\n" for s in ("left", "right") for k in ("fx", "fy", "cx", "cy")) +
                   f"Camera.bf: {K['bf']}\nCamera.bNeedUndistortion: 0\nMap.activeMap.size: 7\nLCD.nDatabaseMinSize: 50\n"
                   "LCD.similarityScoreThreshold.high: 0.94\nLCD.similarityScoreThreshold.low: 0.92\n")
    out = tmp_path / "result"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_kitti_stereo.py"), str(cfg), str(seq), "--frames", str(n), "--out", str(out)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = open(out / "trajectory.txt").read().strip().split("\n")
    assert len(lines) == (n - 1) // 6 + 1                                     # a key-frame every 6th frame
    rows = np.array([[float(x) for x in l.split()] for l in lines])
    assert rows[:, 0].tolist() == list(range(len(lines))) and np.allclose(rows[:, 1], [0.1 * 6 * i for i in range(len(lines))], atol=1e-6)
    assert open(out / "loop_edges.txt").read() == ""                          # the gate of 50 key-frames never opens on 60 frames
    # the written camera centres (Twc translation) against the rendered path, both in the frame of camera 0
    T0 = sc.T_of(synth.pose7_from_twc(C[0], yaw[0]))
    gt = np.array([np.linalg.inv(sc.T_of(synth.pose7_from_twc(C[6 * i], yaw[6 * i])) @ np.linalg.inv(T0))[:3, 3] for i in range(len(lines))])
    err = np.linalg.norm(rows[:, 2:5] - gt, axis=1)
    assert err.max() < 0.6, err                                               # the reference's un-anchored BA gauge moves the first windows
    assert np.allclose(np.linalg.norm(rows[:, 5:9], axis=1), 1.0, atol=1e-5)
    assert "frames/s" in r.stdout
