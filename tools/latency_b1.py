"""B = 1 drop-in latency: wall time per CALL of the host-pointer entry points in the shape the reference uses them — one
1241x376 frame / one key-frame per call (src/frontend.cpp:313-315, src/loopclosing.cpp:91-112,172, src/backend.cpp:208-232) —
beside the oracle's single-thread time for the same call on this host.  `python bench.py --workload latency` prints the table as
JSON; not the headline metric (that one is batched and device-resident)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _median_ms(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e3)


def run(api, synth, with_oracle=True, reps=30):
    o = None
    if with_oracle:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from pyoracle import Oracle
        o = Oracle()
    fr = synth.stereo_batch(2, stream_id=0)
    L, R, L1 = fr[0, 0], fr[0, 1], fr[1, 0]
    K = synth.KITTI00
    rows = {}

    def add(name, ref, gpu_fn, cpu_fn, cpu_reps=3):
        rec = {"reference_call": ref, "gpu_ms": _median_ms(gpu_fn, reps)}
        if o is not None and cpu_fn is not None:
            rec["oracle_1thread_ms"] = _median_ms(cpu_fn, cpu_reps, warm=1)
        rows[name] = rec

    det = api.ORBextractor(300)
    add("myslam_orb_detect (300 features)", "ORBextractor::Detect, src/frontend.cpp:313-315", lambda: det.Detect(L),
        (lambda: o.detect(o.params(300), L)) if o else None)
    ext = api.ORBextractor(2000)
    kl, dl = ext.DetectAndCompute(L); kr, dr = ext.DetectAndCompute(R)
    add("myslam_orb_detect_and_compute (2000 features)", "ORBextractor::DetectAndCompute, src/ORBextractor.cpp:922-985", lambda: ext.DetectAndCompute(L),
        (lambda: o.detect_and_compute(o.params(2000), L)) if o else None)
    w = synth.calc_weights()
    lcd = api.DeepLCD(w)
    add("myslam_lcd_calc_descr_original_img", "DeepLCD::calcDescrOriginalImg, src/loopclosing.cpp:91", lambda: lcd.calcDescrOriginalImg(L),
        (lambda: o.calc_forward(w, o.calc_preproc(L)[0])) if o else None)
    n_db = 10000
    db = synth.lcd_database(n_db); ids = np.arange(n_db, dtype=np.uint64)
    D = api.LoopDatabase(n_db)
    import ctypes as C
    hd = np.ascontiguousarray(db)
    # bulk append through the host-pointer entry point would be 10 000 calls: use one device copy (setup, not timed)
    import torch
    t_db = torch.from_numpy(hd).cuda()
    D.append_batch(ids, t_db.data_ptr(), n_db)
    q = db[1234].copy()
    add("myslam_lcddb_query (10 000 key-frames)", "LoopClosing::DetectLoop, src/loopclosing.cpp:124-161", lambda: D.query(q, n_db + 20),
        (lambda: o.lcddb_query(db, ids, q, n_db + 20)) if o else None)
    add("myslam_hamming_match (2000 x 2000)", "BFMatcher::match, src/loopclosing.cpp:172", lambda: api.hamming_match(dl, dr),
        (lambda: o.hamming_match(dl, dr)) if o else None)
    ba = synth.ba_problem(n_kf=7, n_mp=300)
    add("myslam_ba_optimize_active_map (7 KF x 300 MP)", "Backend::OptimizeActiveMap, src/backend.cpp:208-243", lambda: api.ba_optimize_active_map(*ba),
        (lambda: o.ba_optimize_active_map(*ba)) if o else None, cpu_reps=2)
    pts = np.stack([kl["x"], kl["y"]], 1).astype(np.float32)[:150]
    lk = api.LKTracker()
    add("myslam_lk_track (150 points)", "cv::calcOpticalFlowPyrLK, src/frontend.cpp:150-153", lambda: lk.track(L, L1, pts, pts),
        (lambda: o.lk_track(L, L1, pts, pts)) if o else None)
    xl, yl = kl["x"], kl["y"]
    idx, _ = api.hamming_match(dl, dr)
    add("myslam_triangulate_stereo (2000 matches)", "triangulation(), src/frontend.cpp:385-417",
        lambda: api.triangulate_stereo(xl, yl, kr["x"][idx], kr["y"][idx], K["fx"], K["fy"], K["cx"], K["cy"], K["bf"] / K["fx"]),
        (lambda: o.triangulate_stereo(xl, yl, kr["x"][idx], kr["y"][idx], K["fx"], K["fy"], K["cx"], K["cy"], K["bf"] / K["fx"])) if o else None)
    return {"metric": "per-call latency at B = 1 (host pointers, one 1241x376 frame / one key-frame per call)", "unit": "ms", "reps": reps,
            "higher_is_better": False, "latency_b1": rows}
