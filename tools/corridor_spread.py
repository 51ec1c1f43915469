#!/usr/bin/env python3
"""Free run of the HIP chain on the corridor drive with whatever library is in-tree: key-frame frames, the largest distance of a key-frame's
camera centre from the committed fixture's path (tests/golden/kitti_layout_corridor_trajectory.txt, the ORACLE chain's run) and the ATE against
ground truth.  Used with tools/ab_*.sh over library variants whose pose-only sums run in different orders (all inside the per-call parity bars)
to see how far two free runs of a 200-frame drive spread — the basis of the bound in tests/test_gpu_runner_variants.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import kitti_layout  # noqa: E402

pkg = load_package()
synth, chain, api = pkg.synth, pkg.chain, pkg.api
name = sys.argv[1] if len(sys.argv) > 1 else "corridor"
frames, C, yaw = kitti_layout.render_variant(synth, name)
cfg = kitti_layout.parse_yaml(kitti_layout.KITTI00_02_YAML)
w = synth.calc_weights_handcrafted()
a = chain.Chain(chain.HipBackend(api, w, cfg), api, chain.camera_from_config(cfg), frames, cfg=cfg, timestamps=[0.1 * t for t in range(len(frames))], log=False).run()
rmse, worst = kitti_layout.ate(chain, synth, a.poses, C, yaw)
out = {"sequence": name, "key_frames": len(a.kf_frames), "kf_frames": a.kf_frames, "ate_rmse_m": round(rmse, 4), "ate_worst_m": round(worst, 4)}
fix = os.path.join(ROOT, "tests", "golden", f"kitti_layout_{name}_trajectory.txt")
if os.path.exists(fix):
    gold = np.array([[float(x) for x in l.split()] for l in open(fix)])
    dev = 0.0
    for k in a.all_kfs.values():
        c_k = chain.T_inv(chain.T_of(k.pose))[:3, 3]
        g = np.array([np.interp(k.ts, gold[:, 1], gold[:, 2 + i]) for i in range(3)])
        dev = max(dev, float(np.abs(c_k - g).max()))
    out["max_dev_from_fixture_m"] = round(dev, 4)
print(json.dumps(out))
