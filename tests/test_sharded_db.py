"""N>1 path on CPU: world_size-2 and world_size-8 gloo runs of the sharded loop-database exchange.
Per-shard scans come from the oracle here (the HIP scan itself is covered by tests/test_gpu_lcd.py); what is under test is the
record layout, the collective and the library's reduce rule (myslam_lcd_merge_candidates, a plain C++ host entry point of the
product library): results must equal ONE scan over the whole database."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_package


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _problem(pkg, n_total, nq, world):
    db = pkg.synth.lcd_database(n_total)
    db[n_total // 2 + 3] = db[5]                                   # an exact duplicate in another shard: lowest id must win
    ids = np.arange(n_total, dtype=np.uint64) * 2
    rng = np.random.default_rng(1)
    q = db[rng.integers(0, n_total, nq)] * 0.9 + 0.1 * pkg.synth.lcd_database(nq, seed=3)
    q[0] = db[5]
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    cur = rng.integers(30, 2 * n_total + 40, nq).astype(np.uint64)
    per = n_total // world
    if world >= 4:
        # query 1: its best match lives in shard 5, but the scan breaks inside shard 3 (an id within 20 of cur): shards 4.. are dead
        q[1] = db[5 * per + 7]
        cur[1] = ids[3 * per + per // 2] + 9
        # query 2: breaks exactly at the first row of shard 3 (shard 3 contributes nothing, shards 0-2 everything)
        cur[2] = ids[3 * per] + 19
    return db, ids, q, cur, per


def _worker(rank, world, port, n_total, nq, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyoracle import Oracle
    pkg = load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    db, ids, q, cur, per = _problem(pkg, n_total, nq, world)
    lo, hi = rank * per, (rank + 1) * per if rank < world - 1 else n_total
    pkg.sharded_db.check_shard_order(int(ids[lo]), int(ids[hi - 1]), world, via_cpu=True)
    best = np.zeros(nq, np.uint64); mx = np.zeros(nq, np.float32); cnt = np.zeros(nq, np.int32)
    for i in range(nq):
        # the cut-off (stop at the first id with cur - id < 20) is a property of the id, so it applies per shard
        best[i], mx[i], cnt[i] = o.lcddb_query(db[lo:hi], ids[lo:hi], q[i], int(cur[i]))
    rec = pkg.sharded_db.pack_candidates(best, mx, cnt, pkg.sharded_db.shard_breaks(ids[lo:hi], cur))
    t_best = torch.zeros(nq, dtype=torch.int64); t_mx = torch.zeros(nq); t_cnt = torch.zeros(nq, dtype=torch.int32)
    pkg.sharded_db.exchange_and_merge(torch.from_numpy(rec.view(np.uint8).copy()), world, t_best, t_mx, t_cnt, via_cpu=True)
    ref = [o.lcddb_query(db, ids, q[i], int(cur[i])) for i in range(nq)]
    np.save(os.path.join(out_dir, f"r{rank}.npy"),
            np.array([[int(t_best[i]), float(t_mx[i]), int(t_cnt[i])] + list(ref[i]) for i in range(nq)]))
    dist.destroy_process_group()


def _check(tmp_path, world):
    rs = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    for r in rs[1:]:
        assert np.array_equal(rs[0], r)                            # every rank ends with the same answer
    r0 = rs[0]
    assert np.array_equal(r0[:, 0], r0[:, 3]) and np.allclose(r0[:, 1], r0[:, 4], atol=1e-6) and np.array_equal(r0[:, 2], r0[:, 5])
    assert r0[0, 0] == 10                                          # duplicate rows at ids 10 and 2*(n/2+3): the lower id wins
    return r0


def test_two_rank_sharded_scan_equals_single_scan(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, 400, 24, str(tmp_path)), nprocs=world, join=True)
    _check(tmp_path, world)


def test_eight_rank_sharded_scan_with_a_break_inside_shard_3(tmp_path):
    """configs[4]'s topology: 8 shards; one query whose global best lies behind the shard in which the scan breaks."""
    world, port, n_total = 8, _free_port(), 800
    mp.spawn(_worker, args=(world, port, n_total, 24, str(tmp_path)), nprocs=world, join=True)
    r0 = _check(tmp_path, world)
    per = n_total // world
    assert r0[1, 0] < 2 * (3 * per + per // 2)                     # query 1 never saw shard 5's perfect match: the scan broke in shard 3
    assert r0[1, 1] < 0.999


def test_merge_rule_unit():
    pkg = load_package()
    api = pkg.api
    g = np.zeros((3, 4), api.CAND_DTYPE)
    g["max_score"] = [[0.5, 0.0, 0.9, 0.2], [0.5, 0.0, 0.95, 0.3], [0.3, 0.0, 0.95, 0.9]]
    g["best_id"] = [[7, 0, 11, 2 ** 63 + 5], [107, 0, 150, 2 ** 63 + 9], [250, 0, 201, 3]]
    g["cnt"] = [[1, 0, 2, 0], [0, 0, 1, 1], [3, 0, 1, 5]]
    best, mx, cnt = api.lcd_merge_candidates(g)
    assert best.tolist() == [7, 0, 150, 3] and cnt.tolist() == [4, 0, 4, 6] and np.allclose(mx, [0.5, 0.0, 0.95, 0.9])
    # shard 0 broke for q2, shard 1 for q0 and q3: later shards are ignored; ids >= 2^63 survive untouched (no float round trip)
    g["cnt"] = pkg.sharded_db.pack_candidates(g["best_id"].ravel(), g["max_score"].ravel(), g["cnt"].ravel(),
                                              np.array([[0, 0, 1, 0], [1, 0, 0, 1], [0, 0, 0, 0]], bool).ravel())["cnt"].reshape(3, 4)
    best, mx, cnt = api.lcd_merge_candidates(g)
    assert best.tolist() == [7, 0, 11, 2 ** 63 + 9] and cnt.tolist() == [1, 0, 2, 1]


def test_shard_breaks():
    pkg = load_package()
    ids = np.array([5, 9, 40, 41, 100], np.uint64)
    got = pkg.sharded_db.shard_breaks(ids, np.array([3, 5, 24, 25, 58, 59, 60, 61, 200, 10], np.uint64))
    assert got.tolist() == [False, True, True, True, True, True, True, False, False, True]
