"""A whole batched step recorded into a HIP graph (myslam_graph_begin / _end, csrc/graph.hip) and replayed: both extractor handles with
their FAST gate, L/R Hamming match + triangulation, DeepLCD -> loop-DB scan -> BA block build on a side stream.  Every replay must equal
the eager step bit for bit; the loop database's row limits are fed through myslam_lcddb_update_query_limits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(api, synth, P=4, h=240, w=320, nf=500):
    import torch
    dev = torch.device("cuda")
    s_main, s_b, s_side = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    frames = np.stack([np.stack([synth.random_image(900 + i, h, w), np.roll(synth.random_image(900 + i, h, w), -5, axis=1)]) for i in range(P)])
    imgs = torch.from_numpy(np.concatenate([frames[:, 0], frames[:, 1]])).to(dev)
    A, B = api.ORBextractor(nf, stream=s_main.cuda_stream), api.ORBextractor(nf, stream=s_b.cuda_stream)
    cap = A.max_keypoints(h, w)
    z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
    o = {"kps": z(2 * P * cap * 28, torch.uint8), "desc": z(2 * P * cap * 32, torch.uint8), "cnt": z(2 * P, torch.int32), "stat": z(2 * P, torch.int32),
         "midx": z(P * cap, torch.int32), "mdist": z(P * cap, torch.int32), "xyz": z(P * cap * 3, torch.float64), "ok": z(P * cap, torch.uint8),
         "descr": torch.zeros(P, 1064, device=dev), "best": z(P, torch.int64), "max": z(P, torch.float32), "dbcnt": z(P, torch.int32)}
    lcd = api.DeepLCD(synth.calc_weights(), stream=s_side.cuda_stream)
    n_db = 300
    db = synth.lcd_database(n_db + 200)
    D = api.LoopDatabase(512, stream=s_side.cuda_stream)
    t_db = torch.from_numpy(db).to(dev)
    D.append_batch(np.arange(n_db, dtype=np.uint64), t_db.data_ptr(), n_db)
    ba_w, Kt = synth.ba_windows(P, seed0=77, n_kf=5, n_mp=60)
    maxP, maxL, maxE = ba_w[0].shape[1], ba_w[1].shape[1], ba_w[2].shape[1]
    b_in = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in ba_w]
    b_out = [torch.zeros(P, n, dtype=torch.float64, device=dev) for n in (maxP * 36, maxL * 9, maxE * 18, maxP * 6, maxL * 3, maxE)]
    K = synth.KITTI00
    ev_fa, ev_b = torch.cuda.Event(), torch.cuda.Event()
    ev_fa.record(s_main); ev_b.record(s_main); torch.cuda.synchronize()
    A.set_fast_event(ev_fa.cuda_event); B.set_fast_gate(ev_fa.cuda_event)          # B's FAST stage follows A's
    cur_ids = [np.full(P, n_db + 20, np.uint64)]

    def body():
        A.detect_and_compute_batch(imgs.data_ptr(), P, h, w, w, h * w, o["kps"].data_ptr(), o["desc"].data_ptr(), o["cnt"].data_ptr(), o["stat"].data_ptr(), cap)
        B.detect_and_compute_batch(imgs.data_ptr() + P * h * w, P, h, w, w, h * w, o["kps"].data_ptr() + P * cap * 28, o["desc"].data_ptr() + P * cap * 32,
                                   o["cnt"].data_ptr() + 4 * P, o["stat"].data_ptr() + 4 * P, cap)
        ev_b.record(s_b); s_main.wait_event(ev_b)
        api.hamming_match_batch(o["desc"].data_ptr(), o["cnt"].data_ptr(), o["desc"].data_ptr() + P * cap * 32, o["cnt"].data_ptr() + 4 * P, P, cap,
                                o["midx"].data_ptr(), o["mdist"].data_ptr(), s_main.cuda_stream)
        api.triangulate_stereo_batch(o["kps"].data_ptr(), o["kps"].data_ptr() + P * cap * 28, o["midx"].data_ptr(), o["cnt"].data_ptr(), P, cap,
                                     (K["fx"], K["fy"], K["cx"], K["cy"]), K["bf"] / K["fx"], o["xyz"].data_ptr(), o["ok"].data_ptr(), s_main.cuda_stream)
        lcd.describe_batch(imgs.data_ptr(), P, h, w, w, h * w, o["descr"].data_ptr(), blur_in_place=False)
        D.query_batch(o["descr"].data_ptr(), cur_ids[0], P, o["best"].data_ptr(), o["max"].data_ptr(), o["dbcnt"].data_ptr())
        api.ba_build_batch(*[t.data_ptr() for t in b_in], P, maxP, maxL, maxE, Kt, 5.991, *[t.data_ptr() for t in b_out], s_side.cuda_stream)

    def eager():
        s_b.wait_stream(s_main); s_side.wait_stream(s_main)
        body()
        s_main.wait_stream(s_b); s_main.wait_stream(s_side)
        torch.cuda.synchronize()

    outs = list(o.values()) + b_out
    snap = lambda: [t.clone() for t in outs]

    def clear():
        for t in outs:
            t.zero_()
    return dict(torch=torch, body=body, eager=eager, snap=snap, clear=clear, outs=outs, D=D, t_db=t_db, db=db, n_db=n_db, cur_ids=cur_ids, o=o,
                streams=(s_main, s_b, s_side), P=P,
                keep=(A, B, lcd, imgs, b_in, ev_fa, ev_b))          # (the handles hold the raw hipEvent_t: the torch events must outlive them)


def test_recorded_step_equals_the_eager_step_bit_for_bit(api, synth, oracle):
    c = _setup(api, synth)
    torch, (s_main, s_b, s_side) = c["torch"], c["streams"]
    c["eager"](); c["eager"]()                                    # lazy allocations; both FAST-statistics parities have run
    c["clear"](); c["eager"]()
    ref = c["snap"]()
    assert int(c["o"]["cnt"].min()) > 100 and int(c["o"]["stat"].abs().sum()) == 0
    graphs = [api.StepGraph.record(s_main.cuda_stream, [s_b.cuda_stream, s_side.cuda_stream], c["body"]) for _ in range(2)]
    assert all(g.node_count() >= 20 for g in graphs), [g.node_count() for g in graphs]      # (38 before the small-batch pyramid took three levels per launch)
    for k in range(6):                                            # the two recorded steps alternate (FAST statistics ping-pong)
        c["clear"]()
        graphs[k % 2].launch(s_main.cuda_stream)
        torch.cuda.synchronize()
        for i, (a, r) in enumerate(zip(c["outs"], ref)):
            assert torch.equal(a.view(torch.uint8), r.view(torch.uint8)), f"replay {k}: output {i} differs from the eager step"
    # the loop database grows INSIDE its allocation and the queries move on: the recorded scan covers the allocation and its copy node reads
    # the pinned row limits (and the row count) at every replay — the same graphs keep serving (round 5)
    D, P, n_db = c["D"], c["P"], c["n_db"]
    add = 100
    D.append_batch(np.arange(n_db, n_db + add, dtype=np.uint64), c["t_db"].data_ptr() + n_db * 1064 * 4, add)
    assert D.generation() == 0 and len(D) <= D.capacity()
    descr = c["o"]["descr"].cpu().numpy(); ids = np.arange(n_db + add, dtype=np.uint64)
    for k, new_ids in enumerate([np.array([n_db + add + 20, 150, 40, n_db + 5][:P], np.uint64), np.array([60, n_db + add + 20, 25, 330][:P], np.uint64)]):
        D.update_query_limits(new_ids)                            # same graphs, other cut-offs, more rows
        c["clear"](); graphs[k % 2].launch(s_main.cuda_stream); torch.cuda.synchronize()
        for q in range(P):
            rb, rm, rc = oracle.lcddb_query(c["db"][:n_db + add], ids, descr[q], int(new_ids[q]))
            assert int(c["o"]["best"][q]) == rb and abs(float(c["o"]["max"][q]) - rm) < 2e-5 and int(c["o"]["dbcnt"][q]) == rc, (k, q)
    c["cur_ids"][0] = new_ids
    c["clear"](); c["eager"]()                                    # the eager step agrees with the replay, every output bit for bit
    ref2 = c["snap"]()
    c["clear"](); graphs[0].launch(s_main.cuda_stream); torch.cuda.synchronize()
    for i, (a, r) in enumerate(zip(c["outs"], ref2)):
        assert torch.equal(a.view(torch.uint8), r.view(torch.uint8)), f"replay after appends: output {i}"
    # a database that moved (it outgrew its allocation) cannot serve a recorded step: the limits are refused and so is the launch —
    # and a caller that ignores both still reads valid memory (the old matrix stays allocated)
    D.reserve(4096)
    assert D.generation() == 1
    with pytest.raises(api.MyslamError) as e1:
        D.update_query_limits(new_ids)
    with pytest.raises(api.MyslamError) as e2:
        graphs[0].launch(s_main.cuda_stream)
    assert e1.value.code == -3 and e2.value.code == -3
    c["eager"](); c["eager"]()
    graphs = [api.StepGraph.record(s_main.cuda_stream, [s_b.cuda_stream, s_side.cuda_stream], c["body"]) for _ in range(2)]
    c["clear"](); graphs[0].launch(s_main.cuda_stream); torch.cuda.synchronize()
    for i, (a, r) in enumerate(zip(c["outs"], ref2)):
        assert torch.equal(a.view(torch.uint8), r.view(torch.uint8)), f"re-recorded step: output {i}"


def test_recording_rules(api, synth):
    c = _setup(api, synth, P=2)
    torch, (s_main, s_b, s_side) = c["torch"], c["streams"]
    c["eager"]()
    api.prof_enable(True)
    try:
        with pytest.raises(api.MyslamError):                      # profiling events are host bookkeeping: not replayable
            api.StepGraph.record(s_main.cuda_stream, [s_b.cuda_stream, s_side.cuda_stream], c["body"])
    finally:
        api.prof_enable(False)
    with pytest.raises(api.MyslamError):                          # a side stream must differ from the origin
        api.StepGraph.record(s_main.cuda_stream, [s_main.cuda_stream], lambda: None)
    c["eager"]()                                                  # the handles are still usable after the refused recordings
    assert int(c["o"]["stat"].abs().sum()) == 0
