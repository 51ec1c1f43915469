#!/usr/bin/env python3
"""BASELINE configs[0]'s entry point: `run_kitti_stereo <config.yaml> <sequence dir>` of the reference (app/run_kitti_stereo.cpp) with the
per-frame dense path running through libmyslam_hip.so.

    python tools/run_kitti_stereo.py config/stereo/gray/KITTI00-02.yaml /data/kitti/sequences/00 [--frames 200] [--out result]

Reads <sequence>/times.txt and image_0 / image_1/%06d.png (the library's own PNG reader: no OpenCV), tracks every frame, inserts
key-frames, runs the local BA, the loop detector and — when DetectLoop accepts a candidate — the loop correction, and writes
<out>/trajectory.txt and <out>/loop_edges.txt in the reference's format (src/system.cpp:153-224).  The orchestration is the Python
restatement of Frontend / Backend / LoopClosing that the sequence test uses (tests/sequence_chain.py): a key-frame every
`--kf-every` frames instead of the reference's tracked-feature-count rule (src/frontend.cpp:118-124), loops closed after the last
frame.  The KITTI data set is not part of this repository; tests/test_gpu_runner.py runs this program on a rendered sequence in KITTI
layout.  The CALC model: `--calc-prototxt / --calc-model` (the reference's calc_model/ files) or, without them, the hand-built bank
`synth.calc_weights_handcrafted()`."""
import argparse
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402


def read_config(path):
    """the `key: value` lines of the reference's OpenCV-FileStorage YAML (config/stereo/gray/*.yaml)"""
    kv = {}
    for line in open(path):
        line = line.split("#", 1)[0].strip()
        m = re.match(r"^([A-Za-z_][\w.]*)\s*:\s*(.+)$", line)
        if m:
            kv[m.group(1)] = m.group(2).strip().strip('"')
    return kv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config"); ap.add_argument("sequence")
    ap.add_argument("--frames", type=int, default=0, help="process the first N frames only (configs[0]: 200)")
    ap.add_argument("--out", default="result")
    ap.add_argument("--kf-every", type=int, default=6)
    ap.add_argument("--calc-prototxt", default=""); ap.add_argument("--calc-model", default="")
    args = ap.parse_args()
    import torch  # noqa: F401  (the HIP runtime torch ships must be loaded before ours)
    pkg = load_package(); api, synth = pkg.api, pkg.synth
    import sequence_chain as sc
    cfg = read_config(args.config)
    # System::GetCamera (src/system.cpp:101-146): both cameras take the Camera.right.* intrinsics (reference quirk 8), baseline = bf / fx
    K = {"fx": float(cfg["Camera.right.fx"]), "fy": float(cfg["Camera.right.fy"]), "cx": float(cfg["Camera.right.cx"]),
         "cy": float(cfg["Camera.right.cy"]), "bf": float(cfg["Camera.bf"])}
    left, right, ts = api.load_images(args.sequence)
    n = len(left) if args.frames <= 0 else min(args.frames, len(left))
    assert n >= 2, f"{args.sequence}: times.txt lists {len(left)} frames"
    t0 = time.perf_counter()
    frames = [(api.read_png_gray(left[i]), api.read_png_gray(right[i])) for i in range(n)]
    t_read = time.perf_counter() - t0
    be = sc.HipBackend(api, synth.calc_weights_handcrafted())
    if args.calc_prototxt:
        be.lcd = api.DeepLCD.from_caffe(args.calc_prototxt, args.calc_model)
    window = int(cfg.get("Map.activeMap.size", 7))
    chain = sc.Chain(be, api, K, frames, kf_every=args.kf_every, window=window,
                     lcd_min_db=int(cfg.get("LCD.nDatabaseMinSize", 50)),
                     lcd_thr_high=float(cfg.get("LCD.similarityScoreThreshold.high", 0.94)),
                     lcd_thr_low=float(cfg.get("LCD.similarityScoreThreshold.low", 0.92)))
    t0 = time.perf_counter()
    chain.run()
    t_run = time.perf_counter() - t0
    os.makedirs(args.out, exist_ok=True)
    ids = np.arange(len(chain.kfs), dtype=np.uint64)
    kts = np.array([ts[k["frame"]] for k in chain.kfs]); poses = np.stack([k["pose"] for k in chain.kfs])
    api.save_trajectory(os.path.join(args.out, "trajectory.txt"), ids, kts, poses)
    cur = [c for c, _ in chain.detected]; loop = [l for _, l in chain.detected]
    api.save_loop_edges(os.path.join(args.out, "loop_edges.txt"), np.array(cur, np.uint64), kts[cur] if cur else np.zeros(0),
                        poses[cur] if cur else np.zeros((0, 7)), np.array(loop, np.uint64), kts[loop] if loop else np.zeros(0),
                        poses[loop] if loop else np.zeros((0, 7)))
    np.save(os.path.join(args.out, "frame_poses_cw.npy"), np.stack(chain.poses))
    print(f"{n} frames ({frames[0][0].shape[1]}x{frames[0][0].shape[0]}), {len(chain.kfs)} key-frames, {len(chain.points)} map points, "
          f"{len(chain.detected)} loops; read {t_read:.1f} s, tracked + mapped in {t_run:.1f} s = {n / t_run:.1f} frames/s "
          f"(Python orchestration, one call per operator and frame); wrote {args.out}/trajectory.txt, loop_edges.txt")


if __name__ == "__main__":
    main()
