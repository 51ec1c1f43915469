"""Times the device PnP-RANSAC against the oracle (GPU box)."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: F401
from __graft_entry__ import load_package
pkg = load_package(); api, synth = pkg.api, pkg.synth
import pyoracle
o = pyoracle.Oracle()
for n, frac in ((100, 0.3), (400, 0.5), (2000, 0.6)):
    pw, uv, K, pose, good = synth.pnp_problem(n, frac, 0.5, seed=n)
    api.solve_pnp_ransac(pw, uv, K)
    t = time.time()
    for _ in range(20): g = api.solve_pnp_ransac(pw, uv, K)
    tg = (time.time() - t) / 20
    t = time.time()
    for _ in range(20): r = o.solve_pnp_ransac(pw, uv, K)
    tc = (time.time() - t) / 20
    print(f"n={n} outliers={frac}: gpu {tg*1e3:.2f} ms  oracle {tc*1e3:.2f} ms  inliers {g[2]} / {r[3]}", flush=True)
