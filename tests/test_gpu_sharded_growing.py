"""The growing sharded loop database on the device (round 6): N shards = N `myslam_lcddb` handles on the one GPU of the box, ownership by arrival, the
library's own scan behind `myslam_lcddb_query_batch_owned` (both kernels: GEMV for a few queries, matrix cores from 32) and the device merge — against ONE
scan of the whole map by the oracle (reference src/loopclosing.cpp:124-161, 651-659).  The collective around it is covered over gloo by
tests/test_sharded_growing.py; `GrowingShardedDatabase` is the same class there and here (world 1 per shard object would hide the interleaving, so the
N shards are driven directly)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SCORE_ATOL = 2e-5


@pytest.mark.parametrize("N,P", [(2, 1), (8, 4), (4, 40)])
def test_n_shards_on_one_gpu_grow_and_answer_like_one_map(api, oracle, synth, N, P):
    import torch
    steps = 120 if P < 40 else 50
    shards = [api.LoopDatabase(32) for _ in range(N)]            # small first allocation: every shard grows (moves) several times on the way
    rng = np.random.default_rng(N * 100 + P)
    ref_ids, ref_db = [], []
    total, nq_seen, above, broke, reached = 0, 0, 0, 0, 0
    for step in range(steps):
        nq = P
        base = step * (P + 3) + 60 * (step // 10)                             # every tenth step the ids jump by 60: windows that hold no id exist
        ids = (base + np.sort(rng.choice(P + 3, nq, replace=False))).astype(np.uint64)
        d = synth.lcd_database(nq, seed=9000 + step)
        if step % 9 == 8 and len(ref_ids) > 50:
            d[0] = ref_db[len(ref_ids) // 3]                                   # an exact copy of an old row joins the map ...
        if step % 9 == 2 and len(ref_ids) > 50:
            d[nq - 1] = ref_db[len(ref_ids) // 3]                              # ... and is asked for later: the lowest id of the equal rows must win
        cur = ids.copy()
        if step % 5 == 4 and step > 25:                                       # a query from the PAST: rows above cur are reached when nothing sits in its window
            g = 1 + (step // 5) % (step // 10 - 1)                                # a jump of the past: ids ... e | 60 free ids | ...
            e = (10 * g - 1) * (P + 3) + 60 * (g - 1) + P + 2                    # the largest id the step before the jump could have used
            cur[0] = np.uint64(e + 30)                                            # the window lies in the free range: the scan goes on above cur ...
            if step % 10 == 4:
                d[0] = ref_db[-1 - step % 7]                                      # ... and finds its best row there (a copy of a recent key-frame)
        if step % 5 == 3 and step > 25:                                       # a query from the past whose window holds ids: the scan breaks there
            g = 1 + (step // 5) % (step // 10 - 1)
            cur[0] = np.uint64((10 * g - 1) * (P + 3) + 60 * (g - 1) + P + 2 + 3)
        d_q = torch.from_numpy(d).cuda()
        d_gath = torch.zeros(N, nq * 32, dtype=torch.uint8, device="cuda")
        for s, D in enumerate(shards):
            D.query_batch_owned(d_q.data_ptr(), cur, nq, d_gath[s].data_ptr())
        d_best = torch.zeros(nq, dtype=torch.int64, device="cuda"); d_mx = torch.zeros(nq, device="cuda"); d_cnt = torch.zeros(nq, dtype=torch.int32, device="cuda")
        api.lcd_merge_owned_candidates_device(d_gath.data_ptr(), N, nq, d_best.data_ptr(), d_mx.data_ptr(), d_cnt.data_ptr())
        torch.cuda.synchronize()
        best, mx, cnt = d_best.cpu().numpy().view(np.uint64), d_mx.cpu().numpy(), d_cnt.cpu().numpy()
        hb, hm, hc = api.lcd_merge_owned_candidates(d_gath.cpu().numpy().view(api.OWNED_DTYPE).reshape(N, nq))       # the host form of the same merge
        assert np.array_equal(hb, best) and np.array_equal(hm.view(np.uint32), mx.view(np.uint32)) and np.array_equal(hc, cnt)
        if ref_ids:
            R_ids, R_db = np.array(ref_ids, np.uint64), np.stack(ref_db)
            for i in range(nq):
                rb, rm, rc = oracle.lcddb_query(R_db, R_ids, d[i], int(cur[i]))
                near = int((np.abs(R_db @ d[i] - 0.92) < 1e-5).sum())
                assert int(best[i]) == rb and abs(float(mx[i]) - rm) < SCORE_ATOL and abs(int(cnt[i]) - rc) <= near, (step, i, int(cur[i]), int(best[i]), rb, float(mx[i]), rm, int(cnt[i]), rc)
                in_window = bool(((R_ids <= cur[i]) & (R_ids + np.uint64(19) >= cur[i])).any())
                nq_seen += 1; above += rb > int(cur[i]); broke += in_window; reached += (not in_window) and bool((R_ids > cur[i]).any())
        # AddToDatabase after DetectLoop; the j-th key-frame of the step goes to shard (total + j) mod N
        for j in range(nq):
            shards[(total + j) % N].append_batch(ids[j:j + 1], d_q.data_ptr() + j * 1064 * 4, 1)
            ref_ids.append(int(ids[j])); ref_db.append(d[j].copy())
        total += nq
        rows = [len(D) for D in shards]
        assert max(rows) - min(rows) <= 1 and sum(rows) == total
    # both ends of the rule are exercised: scans that end at the break, scans that go on above cur (and some of those find their best row there)
    assert nq_seen > steps * P // 2 and broke > nq_seen // 4 and reached >= 3 and above >= 1, (nq_seen, broke, reached, above)
    assert all(D.generation() >= 1 for D in shards)


def test_growing_sharded_database_class_on_the_device(api, oracle, synth):
    """sharded_db.GrowingShardedDatabase with a HipShard at world 1 (the class the multi-rank job runs per rank; its collectives are skipped at world 1):
    append / query interleaved through the class, device tensors end to end."""
    import torch
    from conftest import load_package
    pkg = load_package()
    G = pkg.sharded_db.GrowingShardedDatabase(pkg.sharded_db.HipShard(64), 1, 0, via_cpu=False)
    ref_ids, ref_db = [], []
    P = 3
    for step in range(60):
        ids = (step * 5 + np.arange(P)).astype(np.uint64)
        d = synth.lcd_database(P, seed=300 + step)
        if step > 30 and step % 7 == 0:
            d[1] = ref_db[10]
        nv = 1 + step % P
        best, mx, cnt = G.step(ids, torch.from_numpy(d).cuda(), nv)
        if ref_ids:
            for i in range(nv):
                rb, rm, rc = oracle.lcddb_query(np.stack(ref_db), np.array(ref_ids, np.uint64), d[i], int(ids[i]))
                assert int(best[i]) == rb and abs(float(mx[i]) - rm) < SCORE_ATOL and int(cnt[i]) == rc, (step, i)
        for i in range(nv):
            ref_ids.append(int(ids[i])); ref_db.append(d[i].copy())
        assert G.shard.rows() == len(ref_ids) == G.total
