#!/usr/bin/env python3
"""Step 1 of the reference pin kit (see tools/dump_reference_goldens.cpp): writes the repo's synthetic inputs where a C++ program can read
them without numpy — `tests/golden/reference/in_*.npy` (plain NPY v1 files, little-endian, C order) and the two images as PGM for cv::imread.

    python tools/make_reference_inputs.py                       # here or on the maintainer's machine: the inputs are deterministic
    g++ ... tools/dump_reference_goldens.cpp ... -lmyslam ...   # on a machine that has the reference built (the header of that file)
    ./dump_reference_goldens tests/golden/reference             # writes ref_*.npy next to the inputs
    python -m pytest tests/test_reference_pin.py -q             # oracle vs the reference's own outputs: XFAIL until the ref_* files exist

Inputs: one synthetic 1241x376 stereo pair (synth.stereo_pair(0, 0)), a Detect() mask, 64 stereo correspondences for triangulation(), one
10 KF x 300 MP local-BA window with 3 % gross outliers and one with 60 % (all five rounds fail), the KITTI-00 intrinsics."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "reference")


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def main(OUT=OUT):
    synth = load_package().synth
    os.makedirs(OUT, exist_ok=True)
    L, R = synth.stereo_pair(0, 0)
    write_pgm(os.path.join(OUT, "in_left.pgm"), L); write_pgm(os.path.join(OUT, "in_right.pgm"), R)
    np.save(os.path.join(OUT, "in_left.npy"), L); np.save(os.path.join(OUT, "in_right.npy"), R)
    # Frontend::DetectFeatures' mask (frontend.cpp:283-289): 255 everywhere, 0 in 20x20 boxes around tracked features
    rng = np.random.default_rng(2026)
    mask = np.full(L.shape, 255, np.uint8)
    for _ in range(150):
        x, y = int(rng.integers(20, L.shape[1] - 20)), int(rng.integers(20, L.shape[0] - 20))
        mask[y - 10:y + 10, x - 10:x + 10] = 0
    write_pgm(os.path.join(OUT, "in_mask.pgm"), mask); np.save(os.path.join(OUT, "in_mask.npy"), mask)
    # triangulation(): 64 points seen by the rig's two cameras (Tcw left = I, right = (-baseline, 0, 0)), normalised-plane coordinates
    K = synth.KITTI00
    pw = np.stack([rng.uniform(-8, 8, 64), rng.uniform(-2, 2, 64), rng.uniform(4, 60, 64)], axis=1)
    b = K["bf"] / K["fx"] if "bf" in K else 0.537
    poses34 = np.zeros((2, 3, 4)); poses34[:, :, :3] = np.eye(3); poses34[1, 0, 3] = -b
    pn = np.zeros((64, 2, 3))
    for c in range(2):
        pc = pw @ poses34[c, :, :3].T + poses34[c, :, 3]
        pn[:, c, 0] = pc[:, 0] / pc[:, 2] + rng.normal(0, 2e-4, 64); pn[:, c, 1] = pc[:, 1] / pc[:, 2] + rng.normal(0, 2e-4, 64); pn[:, c, 2] = 1.0
    pn[60:, 1, 1] += np.array([0.05, 0.1, 0.2, 0.4])                # four rays that miss each other (epipolar mismatch): sigma3 / sigma2 decides, some rejected
    np.save(os.path.join(OUT, "in_tri_poses34.npy"), poses34); np.save(os.path.join(OUT, "in_tri_points.npy"), pn)
    np.save(os.path.join(OUT, "in_K.npy"), np.array([K["fx"], K["fy"], K["cx"], K["cy"]], np.float64))
    for name, frac, seed in (("ba", 0.03, 0xBA), ("ba_bad", 0.6, 5)):
        poses, pts, ep, el, obs, fixed, Kt = synth.ba_problem(seed=seed, outlier_frac=frac)
        np.save(os.path.join(OUT, f"in_{name}_poses.npy"), poses); np.save(os.path.join(OUT, f"in_{name}_points.npy"), pts)
        np.save(os.path.join(OUT, f"in_{name}_edge_pose.npy"), ep.astype(np.int32)); np.save(os.path.join(OUT, f"in_{name}_edge_point.npy"), el.astype(np.int32))
        np.save(os.path.join(OUT, f"in_{name}_obs.npy"), obs); np.save(os.path.join(OUT, f"in_{name}_fixed.npy"), fixed.astype(np.uint8))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
