"""Times the device pose-graph optimisation against the oracle on the same graphs (GPU box)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: F401  (must precede the HIP library)
from __graft_entry__ import load_package
pkg = load_package(); api, synth = pkg.api, pkg.synth
import pyoracle
o = pyoracle.Oracle()
import os
sizes = [tuple(int(v) for v in a.split(':')) for a in os.environ.get('PGO_SIZES', '200:2,1500:6,4500:17,20000:40').split(',')]
for n, l in sizes:
    poses, fixed, e0, e1, meas, gt = synth.pose_graph(n, l, seed=n)
    api.pose_graph_optimize(poses, fixed, e0, e1, meas, iters=1)
    t = time.time(); gp, gchi, git = api.pose_graph_optimize(poses, fixed, e0, e1, meas); tg = time.time() - t
    t = time.time(); rp, rchi, rit = o.pose_graph_optimize(poses, fixed, e0, e1, meas); tc = time.time() - t
    print(f"n={n} loops={l}: gpu {tg*1e3:.1f} ms ({git} its, chi2 {gchi:.6g})  oracle {tc*1e3:.1f} ms ({rit} its, chi2 {rchi:.6g})  max|dt| {np.abs(gp[:,4:]-rp[:,4:]).max():.2e}", flush=True)
