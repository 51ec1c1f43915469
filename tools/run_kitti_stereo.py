#!/usr/bin/env python3
"""BASELINE configs[0]'s entry point: `run_kitti_stereo <config.yaml> <sequence dir>` of the reference (app/run_kitti_stereo.cpp) with the
per-frame dense path running through libmyslam_hip.so.

    python tools/run_kitti_stereo.py config/stereo/gray/KITTI00-02.yaml /data/kitti/sequences/00 [--frames 200] [--out result]

Reads <sequence>/times.txt and image_0 / image_1/%06d.png (the library's own PNG reader: no OpenCV), tracks every frame, inserts
key-frames by the reference's rule (pose-only inlier count against numFeatures.trackingGood / trackingBad of the YAML,
src/frontend.cpp:97-120), runs the local BA and the loop closer per key-frame, and writes <out>/trajectory.txt and <out>/loopEdges.txt in
the reference's format (src/system.cpp:153-224).  The orchestration is the package's chain.py (Frontend / Backend / LoopClosing / Map
restated as one sequential schedule; its header lists what a sequential program has to decide where the reference's threads race).
`--kf-every N` replaces the key-frame rule by "every N-th frame".  The KITTI data set is not part of this repository;
tests/test_gpu_runner.py runs this program on rendered sequences in KITTI layout.  The CALC model: `--calc-prototxt / --calc-model` (the
reference's calc_model/ files) or, without them, the hand-built bank `synth.calc_weights_handcrafted()`."""
import argparse
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def read_config(path):
    """the `key: value` lines of the reference's OpenCV-FileStorage YAML (config/stereo/gray/*.yaml; src/config.cpp)"""
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
    ap.add_argument("--kf-every", type=int, default=0, help="0 (default) = the reference's inlier-count rule; N > 0 = a key-frame every N-th frame")
    ap.add_argument("--frontend-wins-race", action="store_true",
                    help="the tracker reads the key-frame's image before DeepLCD blurred it in place (chain.py header)")
    ap.add_argument("--calc-prototxt", default=""); ap.add_argument("--calc-model", default="")
    args = ap.parse_args()
    import torch  # noqa: F401  (the HIP runtime torch ships must be loaded before ours)
    pkg = load_package(); api, synth, chain = pkg.api, pkg.synth, pkg.chain
    cfg = read_config(args.config)
    K = chain.camera_from_config(cfg)
    left, right, ts = api.load_images(args.sequence)
    n = len(left) if args.frames <= 0 else min(args.frames, len(left))
    assert n >= 2, f"{args.sequence}: times.txt lists {len(left)} frames"
    t_read = [0.0]

    def frame(i):           # cv::imread(..., IMREAD_GRAYSCALE) per step, as app/run_kitti_stereo.cpp:66-67
        t0 = time.perf_counter()
        fr = (api.read_png_gray(left[i]), api.read_png_gray(right[i]))
        t_read[0] += time.perf_counter() - t0
        return fr
    lcd = api.DeepLCD.from_caffe(args.calc_prototxt, args.calc_model) if args.calc_prototxt else None
    be = chain.HipBackend(api, None if lcd else synth.calc_weights_handcrafted(), cfg, lcd=lcd)
    sysm = chain.Chain(be, api, K, frame, cfg=cfg, kf_every=args.kf_every, lcd_blur_reaches_tracker=not args.frontend_wins_race,
                       timestamps=ts, log=False)
    t0 = time.perf_counter()
    done = 0
    for i in range(n):
        if not sysm.grab(i):
            print(f"System failed, now quited (frame {i}: tracking LOST)")          # app/run_kitti_stereo.cpp:83-86
            break
        done += 1
    t_run = time.perf_counter() - t0 - t_read[0]
    sysm.save(args.out)
    np.save(os.path.join(args.out, "frame_poses_cw.npy"), np.stack(sysm.poses))
    shape = sysm.last.R.shape
    print(f"{done} frames ({shape[1]}x{shape[0]}), {len(sysm.all_kfs)} key-frames, {len(sysm.all_mps)} map points, "
          f"{len(sysm.loops)} loops; read {t_read[0]:.1f} s, tracked + mapped in {t_run:.1f} s = {done / max(t_run, 1e-9):.1f} frames/s "
          f"(Python orchestration, one call per operator and frame); wrote {args.out}/trajectory.txt, loopEdges.txt")


if __name__ == "__main__":
    main()
