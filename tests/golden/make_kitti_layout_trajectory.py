#!/usr/bin/env python3
"""Writes tests/golden/kitti_layout_200_trajectory.txt (and, with the argument `corridor`, tests/golden/kitti_layout_corridor_trajectory.txt: the
forward drive of tests/kitti_layout.py VARIANTS): the key-frame trajectory (System::SaveTrajectory format) of the rendered 200-frame
1241 x 376 KITTI-layout sequence (tests/kitti_layout.py) tracked by the package's chain through the CPU ORACLE back end with the
reference's KITTI00-02.yaml values.  tests/test_gpu_runner.py runs tools/run_kitti_stereo.py (the HIP library) on the same sequence and
compares.  Runs without a GPU:  python tests/golden/make_kitti_layout_trajectory.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import kitti_layout  # noqa: E402
from oracle_backend import OracleBackend  # noqa: E402
from pyoracle import Oracle  # noqa: E402

pkg = load_package(); synth, chain, api = pkg.synth, pkg.chain, pkg.api
cfg = kitti_layout.parse_yaml(kitti_layout.KITTI00_02_YAML)
variant = sys.argv[1] if len(sys.argv) > 1 else None
frames, C, yaw = kitti_layout.render_variant(synth, variant) if variant else kitti_layout.render(synth)
ts = [0.1 * t for t in range(len(frames))]
c = chain.Chain(OracleBackend(Oracle(), synth.calc_weights_handcrafted(), cfg, chain), api, chain.camera_from_config(cfg), frames, cfg=cfg,
                timestamps=ts, log=False).run()
out = os.path.join(HERE, "_traj_tmp")
c.save(out)
os.replace(os.path.join(out, "trajectory.txt"), os.path.join(HERE, f"kitti_layout_{variant or 200}_trajectory.txt"))
os.remove(os.path.join(out, "loopEdges.txt")); os.rmdir(out)
rmse, worst = kitti_layout.ate(chain, synth, c.poses, C, yaw)
print(f"{len(frames)} frames, {len(c.all_kfs)} key-frames at frames {c.kf_frames}, {len(c.all_mps)} map points, ATE rmse {rmse:.4f} m worst {worst:.4f} m")
