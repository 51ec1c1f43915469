"""ONE loop database, several concurrent streams of queries (myslam_lcddb_query_ctx, csrc/lcddb.hip).

LoopClosing::_mvDatabase is a single std::map that every key-frame of the process goes into (reference include/myslam/loopclosing.h:120,
src/loopclosing.cpp:651-659).  With L cameras on one GPU the L loop-closing streams scan the SAME device matrix through L contexts;
appends go to the storage.  Every scan — eager or replayed from a recorded step, on the context's stream or on another — must equal the
oracle's ascending scan (oracle/calc_oracle.cpp lcddb_query = loopclosing.cpp:124-161) of the rows the database held when it was issued."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SCORE_ATOL = 2e-5


def _check(oracle, db, ids, n, qs, cur, best, mx, cnt, tag=""):
    for i in range(len(qs)):
        rb, rm, rc = oracle.lcddb_query(db[:n], ids[:n], qs[i], int(cur[i]))
        near = int((np.abs(db[:n] @ qs[i] - 0.92) < 1e-5).sum())
        assert int(best[i]) == rb and abs(float(mx[i]) - rm) < SCORE_ATOL and abs(int(cnt[i]) - rc) <= near, (tag, i, int(best[i]), rb, float(mx[i]), rm, int(cnt[i]), rc)


@pytest.mark.parametrize("nq", [1, 3, 40])
def test_two_contexts_on_two_streams_with_appends_in_between(api, oracle, synth, nq):
    """Two contexts on two streams query while a third party appends: each answer = the oracle's scan of the rows present at ITS call.
    nq = 1 / 3: the GEMV kernel (the live-stream case, one key-frame per call); nq = 40: the matrix-core kernel."""
    import torch
    n0, step, rounds = 500, 37, 6
    total = n0 + step * rounds
    db = synth.lcd_database(total, seed=11); ids = np.arange(total, dtype=np.uint64) * 2 + 3
    D = api.LoopDatabase(total + 64)
    t_db = torch.from_numpy(db).cuda()
    D.append_batch(ids[:n0], t_db.data_ptr(), n0)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ca, cb = D.context(sa.cuda_stream), D.context(sb.cuda_stream)
    rng = np.random.default_rng(nq)
    outs = []
    n = n0
    for r in range(rounds):
        for ctx, st in ((ca, sa), (cb, sb)):
            qs = db[rng.integers(0, n, nq)] * 0.96 + 0.04 * synth.lcd_database(nq, seed=100 + 7 * r + (st is sb))
            qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32)
            cur = np.where(rng.random(nq) < 0.5, ids[n - 1] + 20, ids[rng.integers(n // 2, n, nq)]).astype(np.uint64)
            with torch.cuda.stream(st):
                d_q = torch.from_numpy(qs).cuda()
                o = (torch.zeros(nq, dtype=torch.int64, device="cuda"), torch.zeros(nq, device="cuda"), torch.zeros(nq, dtype=torch.int32, device="cuda"))
            st.synchronize()
            ctx.query_batch(d_q.data_ptr(), cur, nq, *[t.data_ptr() for t in o])
            outs.append((n, qs, cur, o, d_q))
        # no synchronisation with the scans in flight: appends only write rows behind every issued limit
        D.append_batch(ids[n:n + step], t_db.data_ptr() + n * 1064 * 4, step)
        n += step
    torch.cuda.synchronize()
    assert len(D) == total and D.generation() == 0
    for k, (n_at, qs, cur, o, _) in enumerate(outs):
        _check(oracle, db, ids, n_at, qs, cur, o[0].cpu().numpy(), o[1].cpu().numpy(), o[2].cpu().numpy(), tag=f"call {k}")


def test_contexts_survive_growth_and_share_one_matrix(api, oracle, synth):
    """The database outgrows its allocation while contexts exist: the move waits for their streams, bumps the generation, and every
    context scans the new matrix afterwards.  16 contexts cost 16 small scratch buffers, not 16 copies of the matrix."""
    import torch
    n0, n1 = 200, 3000
    db = synth.lcd_database(n1, seed=5); ids = np.arange(n1, dtype=np.uint64)
    t_db = torch.from_numpy(db).cuda()
    free0 = torch.cuda.mem_get_info()[0]
    D = api.LoopDatabase(256)
    D.append_batch(ids[:n0], t_db.data_ptr(), n0)
    streams = [torch.cuda.Stream() for _ in range(16)]
    ctxs = [D.context(s.cuda_stream) for s in streams]
    nq = 2
    bufs = []
    for i, (c, s) in enumerate(zip(ctxs, streams)):
        qs = db[[i, n0 - 1 - i]].copy()
        d_q = torch.from_numpy(qs).cuda()
        o = (torch.zeros(nq, dtype=torch.int64, device="cuda"), torch.zeros(nq, device="cuda"), torch.zeros(nq, dtype=torch.int32, device="cuda"))
        torch.cuda.synchronize()
        c.query_batch(d_q.data_ptr(), np.full(nq, n0 + 20, np.uint64), nq, *[t.data_ptr() for t in o])
        bufs.append((qs, d_q, o))
    D.append_batch(ids[n0:], t_db.data_ptr() + n0 * 1064 * 4, n1 - n0)      # grows (moves) with 16 scans possibly in flight
    assert D.generation() >= 1 and len(D) == n1 and D.capacity() >= n1
    torch.cuda.synchronize()
    for qs, _, o in bufs:
        _check(oracle, db, ids, n0, qs, np.full(nq, n0 + 20), *[t.cpu().numpy() for t in o], tag="before the move")
    for i, (c, s) in enumerate(zip(ctxs, streams)):
        qs, d_q, o = bufs[i]
        c.query_batch(d_q.data_ptr(), np.full(nq, n1 + 20, np.uint64), nq, *[t.data_ptr() for t in o])
    torch.cuda.synchronize()
    for qs, _, o in bufs:
        _check(oracle, db, ids, n1, qs, np.full(nq, n1 + 20), *[t.cpu().numpy() for t in o], tag="after the move")
    used = free0 - torch.cuda.mem_get_info()[0]
    assert used < 3 * D.capacity() * 1064 * 4 + (64 << 20), f"{used >> 20} MiB in use for one {D.capacity()}-row matrix and 16 contexts"
    del ctxs


def test_recorded_scan_replayed_on_another_stream_while_limits_change(api, oracle, synth):
    """The advisor's scenario (round 4): a step recorded through a context is REPLAYED ON A STREAM THAT IS NOT THE CONTEXT'S, and the host
    rewrites the row limits / appends / grows the database right behind the launch with no device synchronisation of its own.  The
    context waits for the replay itself (an event behind every launch, on the launch stream); a moved matrix makes the launch fail
    instead of replaying against freed memory."""
    import torch
    n0, nq = 2000, 4
    db = synth.lcd_database(n0 + 600, seed=21); ids = np.arange(n0 + 600, dtype=np.uint64)
    t_db = torch.from_numpy(db).cuda()
    D = api.LoopDatabase(n0 + 300)
    D.append_batch(ids[:n0], t_db.data_ptr(), n0)
    s_rec, s_play = torch.cuda.Stream(), torch.cuda.Stream()
    ctx = D.context(s_rec.cuda_stream)
    qs = db[[5, 700, 1500, 1999]].copy()
    d_q = torch.from_numpy(qs).cuda()
    o = (torch.zeros(nq, dtype=torch.int64, device="cuda"), torch.zeros(nq, device="cuda"), torch.zeros(nq, dtype=torch.int32, device="cuda"))
    cur = [np.full(nq, n0 + 20, np.uint64)]
    body = lambda: ctx.query_batch(d_q.data_ptr(), cur[0], nq, *[t.data_ptr() for t in o])
    torch.cuda.synchronize()
    body(); torch.cuda.synchronize()                               # lazy allocations outside the capture
    g = api.StepGraph.record(s_rec.cuda_stream, [], body)
    n = n0
    rng = np.random.default_rng(3)
    for k in range(40):
        g.launch(s_play.cuda_stream)                               # NOT the context's stream
        # straight behind the launch, no torch synchronisation: new limits (must wait for the replay), sometimes an append first
        n_at, cur_at = n, cur[0].copy()
        if k % 3 == 2 and n + 10 <= n0 + 300:
            D.append_batch(ids[n:n + 10], t_db.data_ptr() + n * 1064 * 4, 10); n += 10
        new = np.where(rng.random(nq) < 0.5, n + 20, rng.integers(30, n, nq)).astype(np.uint64)
        ctx.update_query_limits(new)                               # returns only when the replay above has finished reading the old limits
        got = [t.cpu().numpy() for t in o]                         # the replay is complete here (update waited for it)
        _check(oracle, db, ids, n_at, qs, cur_at, *got, tag=f"replay {k}")
        cur[0] = new
    assert D.generation() == 0
    # growth moves the matrix: launch refused (CAPACITY), limits refused, eager queries keep working on the new matrix
    g.launch(s_play.cuda_stream)
    D.append_batch(ids[n:n0 + 600], t_db.data_ptr() + n * 1064 * 4, n0 + 600 - n)      # waits for that replay, then moves
    assert D.generation() == 1
    with pytest.raises(api.MyslamError) as e:
        g.launch(s_play.cuda_stream)
    assert e.value.code == -3
    with pytest.raises(api.MyslamError):
        ctx.update_query_limits(cur[0])
    cur[0] = np.full(nq, n0 + 600 + 20, np.uint64)
    body(); torch.cuda.synchronize()
    _check(oracle, db, ids, n0 + 600, qs, cur[0], *[t.cpu().numpy() for t in o], tag="eager after the move")
    g2 = api.StepGraph.record(s_rec.cuda_stream, [], body)
    for t in o:
        t.zero_()
    torch.cuda.synchronize()
    g2.launch(s_play.cuda_stream); torch.cuda.synchronize()
    _check(oracle, db, ids, n0 + 600, qs, cur[0], *[t.cpu().numpy() for t in o], tag="re-recorded")
    # a context that is destroyed invalidates the steps that captured it
    del ctx
    import gc; gc.collect()
    with pytest.raises(api.MyslamError):
        g2.launch(s_play.cuda_stream)


def test_sharded_scan_through_a_context_reports_the_break_after_appends(api, oracle, synth):
    """myslam_lcddb_ctx_query_batch_sharded: bit 31 of a record's count = "this shard's scan stopped at the cur - id < 20 break"; it is
    derived from the shard's row count AT THE CALL (kept in device memory beside the limits, so recorded steps see appends)."""
    import torch
    n = 640
    db = synth.lcd_database(n + 64, seed=9); ids = np.arange(n + 64, dtype=np.uint64)
    t_db = torch.from_numpy(db).cuda()
    D = api.LoopDatabase(n + 64)
    D.append_batch(ids[:n], t_db.data_ptr(), n)
    st = torch.cuda.Stream(); ctx = D.context(st.cuda_stream)
    nq = 3
    qs = db[[3, 300, 600]].copy(); d_q = torch.from_numpy(qs).cuda()
    d_cand = torch.zeros(nq * 16, dtype=torch.uint8, device="cuda")
    cur = np.array([n + 20, 310, n + 5], np.uint64)                # no break / break at row 291 / break at row n - 14
    torch.cuda.synchronize()
    ctx.query_batch_sharded(d_q.data_ptr(), cur, nq, d_cand.data_ptr()); torch.cuda.synchronize()
    rec = d_cand.cpu().numpy().view(api.CAND_DTYPE)
    assert [(int(c) >> 31) & 1 for c in rec["cnt"]] == [0, 1, 1]
    for i in range(nq):
        rb, rm, rc = oracle.lcddb_query(db[:n], ids[:n], qs[i], int(cur[i]))
        assert int(rec["best_id"][i]) == rb and abs(float(rec["max_score"][i]) - rm) < SCORE_ATOL and (int(rec["cnt"][i]) & 0x7fffffff) == rc
    D.append_batch(ids[n:n + 32], t_db.data_ptr() + n * 1064 * 4, 32)
    ctx.query_batch_sharded(d_q.data_ptr(), np.array([n + 32 + 20, n + 32 + 20, n + 25], np.uint64), nq, d_cand.data_ptr()); torch.cuda.synchronize()
    rec = d_cand.cpu().numpy().view(api.CAND_DTYPE)
    assert [(int(c) >> 31) & 1 for c in rec["cnt"]] == [0, 0, 1]


def test_scratch_reallocation_invalidates_recorded_steps(api, oracle, synth):
    """Advisor, round 5: a recorded scan names the context's OWN scratch by address (pinned row limits, partial results).  An eager call with
    more queries on the same context frees and reallocates that scratch while the matrix generation stays what it was — the old step must be
    refused (MYSLAM_ERR_CAPACITY: record it again), not replayed against freed memory."""
    import torch
    n0 = 900
    db = synth.lcd_database(n0, seed=31); ids = np.arange(n0, dtype=np.uint64)
    t_db = torch.from_numpy(db).cuda()
    D = api.LoopDatabase(n0 + 100)
    D.append_batch(ids, t_db.data_ptr(), n0)
    st = torch.cuda.Stream(); ctx = D.context(st.cuda_stream)

    def bufs(nq):
        return (torch.zeros(nq, dtype=torch.int64, device="cuda"), torch.zeros(nq, device="cuda"), torch.zeros(nq, dtype=torch.int32, device="cuda"))
    q2 = db[[4, 444]].copy(); d_q2 = torch.from_numpy(q2).cuda(); o2 = bufs(2)
    cur2 = np.full(2, n0 + 20, np.uint64)
    body2 = lambda: ctx.query_batch(d_q2.data_ptr(), cur2, 2, *[t.data_ptr() for t in o2])
    torch.cuda.synchronize()
    body2(); torch.cuda.synchronize()
    g = api.StepGraph.record(st.cuda_stream, [], body2)
    g.launch(st.cuda_stream); torch.cuda.synchronize()
    _check(oracle, db, ids, n0, q2, cur2, *[t.cpu().numpy() for t in o2], tag="recorded, 2 queries")
    # an eager call with 9 queries: row-limit staging (nq + 1 ints) and partial results are reallocated
    q9 = db[np.arange(9) * 97].copy(); d_q9 = torch.from_numpy(q9).cuda(); o9 = bufs(9)
    cur9 = np.full(9, n0 + 20, np.uint64)
    g.launch(st.cuda_stream)                                       # a replay in flight while the scratch is replaced: the call waits for it
    ctx.query_batch(d_q9.data_ptr(), cur9, 9, *[t.data_ptr() for t in o9]); torch.cuda.synchronize()
    _check(oracle, db, ids, n0, q9, cur9, *[t.cpu().numpy() for t in o9], tag="eager, 9 queries")
    assert D.generation() == 0
    with pytest.raises(api.MyslamError) as e:
        g.launch(st.cuda_stream)
    assert e.value.code == -3                                      # MYSLAM_ERR_CAPACITY
    with pytest.raises(api.MyslamError):
        ctx.update_query_limits(cur2)                              # no recorded query to feed any more
    # recorded again (now against the larger scratch), it replays; 40 record / destroy cycles leave no growing list of events to wait on
    for k in range(40):
        g2 = api.StepGraph.record(st.cuda_stream, [], body2)
        for t in o2:
            t.zero_()
        g2.launch(st.cuda_stream); torch.cuda.synchronize()
        if k % 13 == 0:
            _check(oracle, db, ids, n0, q2, cur2, *[t.cpu().numpy() for t in o2], tag=f"re-recorded {k}")
        del g2


def test_growth_is_refused_while_a_step_is_being_recorded(api, synth):
    """db_reserve synchronises every context's stream; a stream that is capturing must not be synchronised (the capture would be invalidated).  While
    a recording that scans through a context is open, an append that has to MOVE the matrix returns MYSLAM_ERR_UNSUPPORTED (before it synchronises
    anything); after myslam_graph_end the growth goes through and the step (recorded against the old matrix) is refused."""
    import torch
    n0 = 256
    db = synth.lcd_database(n0 + 600, seed=41); ids = np.arange(n0 + 600, dtype=np.uint64)
    t_db = torch.from_numpy(db).cuda()
    D = api.LoopDatabase(n0 + 32)
    D.append_batch(ids[:n0], t_db.data_ptr(), n0)
    st = torch.cuda.Stream(); ctx = D.context(st.cuda_stream)
    d_q = torch.from_numpy(db[[1, 2]].copy()).cuda()
    o = (torch.zeros(2, dtype=torch.int64, device="cuda"), torch.zeros(2, device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda"))
    cur = np.full(2, n0 + 700, np.uint64)
    seen = {}

    def body():
        ctx.query_batch(d_q.data_ptr(), cur, 2, *[t.data_ptr() for t in o])
        try:
            D.append_batch(ids[n0 + 8:n0 + 600], t_db.data_ptr() + (n0 + 8) * 1064 * 4, 592)    # must move the matrix
            seen["code"] = 0
        except api.MyslamError as e:
            seen["code"] = e.code
    D.append_batch(ids[n0:n0 + 8], t_db.data_ptr() + n0 * 1064 * 4, 8)
    torch.cuda.synchronize()
    ctx.query_batch(d_q.data_ptr(), cur, 2, *[t.data_ptr() for t in o]); torch.cuda.synchronize()
    g = api.StepGraph.record(st.cuda_stream, [], body)
    assert seen["code"] == -4, seen                                # MYSLAM_ERR_UNSUPPORTED
    assert len(D) == n0 + 8 and D.generation() == 0
    g.launch(st.cuda_stream); torch.cuda.synchronize()
    D.append_batch(ids[n0 + 8:n0 + 600], t_db.data_ptr() + (n0 + 8) * 1064 * 4, 592)    # recording closed: the matrix moves
    assert D.generation() == 1 and len(D) == n0 + 600
    with pytest.raises(api.MyslamError):
        g.launch(st.cuda_stream)


def test_asynchronous_appends_inside_a_stream_of_scans(api, oracle, synth):
    """myslam_lcddb_append_batch_async (round 6): AddToDatabase inside a pipelined step — the rows are copied on the caller's stream, the call does not wait, and scans issued
    afterwards ON THAT STREAM see them (their row limits are computed from the ids, which are updated at once).  60 rounds of scan + append with no host synchronisation in between,
    every scan checked against the oracle on the rows present at its call; beyond the allocation the call refuses (no growth without a synchronisation)."""
    import torch
    n0, step, rounds, nq = 300, 23, 60, 5
    total = n0 + step * rounds
    db = synth.lcd_database(total, seed=51); ids = np.arange(total, dtype=np.uint64) * 3 + 1
    t_db = torch.from_numpy(db).cuda()
    st = torch.cuda.Stream()
    D = api.LoopDatabase(total, stream=st.cuda_stream)
    D.append_batch(ids[:n0], t_db.data_ptr(), n0)
    rng = np.random.default_rng(9)
    outs = []
    n = n0
    torch.cuda.synchronize()
    for r in range(rounds):
        qs = db[rng.integers(0, n, nq)] * 0.97 + 0.03 * synth.lcd_database(nq, seed=900 + r)
        qs = (qs / np.linalg.norm(qs, axis=1, keepdims=True)).astype(np.float32); qs[0] = db[n - 1]                  # the row appended LAST round must be found
        cur = np.where(rng.random(nq) < 0.6, ids[n - 1] + 25, ids[rng.integers(n // 2, n, nq)]).astype(np.uint64); cur[0] = ids[n - 1] + 25
        with torch.cuda.stream(st):
            d_q = torch.from_numpy(qs).cuda()
            o = (torch.zeros(nq, dtype=torch.int64, device="cuda"), torch.zeros(nq, device="cuda"), torch.zeros(nq, dtype=torch.int32, device="cuda"))
        D.query_batch(d_q.data_ptr(), cur, nq, *[t.data_ptr() for t in o])
        outs.append((n, qs, cur, o, d_q))
        D.append_batch_async(ids[n:n + step], t_db.data_ptr() + n * 1064 * 4, step, st.cuda_stream)
        n += step
        assert len(D) == n
    torch.cuda.synchronize()
    for k, (n_at, qs, cur, o, _) in enumerate(outs):
        _check(oracle, db, ids, n_at, qs, cur, o[0].cpu().numpy(), o[1].cpu().numpy(), o[2].cpu().numpy(), tag=f"round {k}")
        assert int(o[0][0]) == int(ids[n_at - 1])
    room = D.capacity() - len(D)                                   # (the allocation is rounded up to whole blocks of rows)
    over = np.arange(room + 1, dtype=np.uint64) + ids[-1] + 5
    t_over = torch.zeros(room + 1, 1064, device="cuda")
    with pytest.raises(api.MyslamError) as e:
        D.append_batch_async(over, t_over.data_ptr(), room + 1, st.cuda_stream)
    assert e.value.code == -3 and D.generation() == 0 and len(D) == total
