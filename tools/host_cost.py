#!/usr/bin/env python3
"""Host-side cost of enqueuing the step's *_batch calls (CPU time per call, device idle between rounds): where a small-batch step's
launch budget goes.  python tools/host_cost.py [--pairs 8] [--created-stream]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8); ap.add_argument("--created-stream", action="store_true"); ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    import torch
    pkg = load_package(); api, synth = pkg.api, pkg.synth
    dev = torch.device("cuda", 0)
    if a.created_stream:
        torch.cuda.set_stream(torch.cuda.Stream())
    st = torch.cuda.current_stream(); s = st.cuda_stream
    P, H, W = a.pairs, 376, 1241
    fr = synth.stereo_batch(P)
    imgs = torch.from_numpy(np.concatenate([fr[:, 0], fr[:, 1]])).to(dev)
    ext = api.ORBextractor(2000, stream=s); ext.set_option(ext.OPT_INTERNAL_STREAM, 0)
    cap = ext.max_keypoints()
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
    kps, desc, cnt, stat = z(2 * P * cap * 28, torch.uint8), z(2 * P * cap * 32, torch.uint8), z(2 * P, torch.int32), z(2 * P, torch.int32)
    midx, mdist, xyz, ok = z(P * cap, torch.int32), z(P * cap, torch.int32), z(P * cap * 3, torch.float64), z(P * cap, torch.uint8)
    lcd = api.DeepLCD(synth.calc_weights(), stream=s); descr = torch.zeros(P, 1064, device=dev)
    db = synth.lcd_database(10000); D = api.LoopDatabase(10000, stream=s); t_db = torch.from_numpy(db).to(dev)
    D.append_batch(np.arange(10000, dtype=np.uint64), t_db.data_ptr(), 10000)
    best, mx, dc = z(P, torch.int64), z(P, torch.float32), z(P, torch.int32)
    cur = np.full(P, 10020, np.uint64)
    ba_w, Kt = synth.ba_windows(P)
    maxP, maxL, maxE = ba_w[0].shape[1], ba_w[1].shape[1], ba_w[2].shape[1]
    b_in = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in ba_w]
    b_out = [torch.zeros(P, n, dtype=torch.float64, device=dev) for n in (maxP * 36, maxL * 9, maxE * 18, maxP * 6, maxL * 3, maxE)]
    K = synth.KITTI00
    ev = torch.cuda.Event(); ev.record(st)
    calls = {
        "orb_detect_and_compute_batch (2P images)": lambda: ext.detect_and_compute_batch(imgs.data_ptr(), 2 * P, H, W, W, H * W, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), stat.data_ptr(), cap),
        "hamming_match_batch": lambda: api.hamming_match_batch(desc.data_ptr(), cnt.data_ptr(), desc.data_ptr() + P * cap * 32, cnt.data_ptr() + 4 * P, P, cap, midx.data_ptr(), mdist.data_ptr(), s),
        "triangulate_stereo_batch": lambda: api.triangulate_stereo_batch(kps.data_ptr(), kps.data_ptr() + P * cap * 28, midx.data_ptr(), cnt.data_ptr(), P, cap, Kt, K["bf"] / K["fx"], xyz.data_ptr(), ok.data_ptr(), s),
        "lcd_describe_batch": lambda: lcd.describe_batch(imgs.data_ptr(), P, H, W, W, H * W, descr.data_ptr(), blur_in_place=False),
        "lcddb_query_batch": lambda: D.query_batch(descr.data_ptr(), cur, P, best.data_ptr(), mx.data_ptr(), dc.data_ptr()),
        "ba_build_batch": lambda: api.ba_build_batch(*[t.data_ptr() for t in b_in], P, maxP, maxL, maxE, Kt, 5.991, *[t.data_ptr() for t in b_out], s),
        "event record + wait (torch)": lambda: (ev.record(st), st.wait_event(ev)),
    }
    out = {"pairs": P, "stream": "created" if a.created_stream else "legacy NULL", "host_us_per_call": {}}
    for name, fn in calls.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for r in range(a.reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            if r % 8 == 7:
                torch.cuda.synchronize()            # the queue never runs full
        torch.cuda.synchronize()
        out["host_us_per_call"][name] = {"median": float(np.median(ts) * 1e6), "mean": float(np.mean(ts) * 1e6), "p90": float(np.percentile(ts, 90) * 1e6)}
    out["sum_median_us"] = sum(v["median"] for k, v in out["host_us_per_call"].items() if "event" not in k)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
