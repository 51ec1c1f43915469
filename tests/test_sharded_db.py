"""N>1 path on CPU: world_size-2 gloo run of the sharded loop-database exchange.
Per-shard scans come from the oracle here (the HIP scan itself is covered by tests/test_gpu_lcd.py);
what is under test is the collective + reduce rule: results must equal ONE scan over the whole database."""
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


def _worker(rank, world, port, n_total, nq, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyoracle import Oracle
    pkg = load_package()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    db = pkg.synth.lcd_database(n_total)
    db[n_total // 2 + 3] = db[5]                                   # an exact duplicate in the other shard: lowest id must win
    ids = np.arange(n_total, dtype=np.uint64) * 2
    rng = np.random.default_rng(1)
    q = db[rng.integers(0, n_total, nq)] * 0.9 + 0.1 * pkg.synth.lcd_database(nq, seed=3)
    q[0] = db[5]
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    cur = rng.integers(30, 2 * n_total + 40, nq).astype(np.uint64)
    per = n_total // world
    lo, hi = rank * per, (rank + 1) * per if rank < world - 1 else n_total
    best = torch.zeros(nq, dtype=torch.int64); mx = torch.zeros(nq); cnt = torch.zeros(nq, dtype=torch.int32)
    for i in range(nq):
        # the cut-off (stop at the first id with cur - id < 20) is a property of the id, so it applies per shard
        b, m, c = o.lcddb_query(db[lo:hi], ids[lo:hi], q[i], int(cur[i]))
        best[i], mx[i], cnt[i] = b, m, c
    broke = torch.from_numpy(pkg.sharded_db.shard_breaks(ids[lo:hi], cur))
    pkg.sharded_db.merge_candidates(best, mx, cnt, world, broke)
    ref = [o.lcddb_query(db, ids, q[i], int(cur[i])) for i in range(nq)]
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([[int(best[i]), float(mx[i]), int(cnt[i])] + list(ref[i]) for i in range(nq)]))
    dist.destroy_process_group()


def test_two_rank_sharded_scan_equals_single_scan(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, 400, 24, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "r0.npy"); r1 = np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1)                                  # every rank ends with the same answer
    assert np.array_equal(r0[:, 0], r0[:, 3]) and np.allclose(r0[:, 1], r0[:, 4], atol=1e-6) and np.array_equal(r0[:, 2], r0[:, 5])
    assert r0[0, 0] == 10                                          # duplicate rows at ids 10 and 2*(203): the lower id wins


def test_merge_rule_unit():
    pkg = load_package()
    s = torch.tensor([[0.5, 0.0, 0.9], [0.5, 0.0, 0.95], [0.3, 0.0, 0.95]])
    i = torch.tensor([[7, 0, 11], [107, 0, 150], [250, 0, 201]])
    c = torch.tensor([[1, 0, 2], [0, 0, 1], [3, 0, 1]])
    mx, best, cnt = pkg.sharded_db.merge_shard_triples(s, i, c)
    assert best.tolist() == [7, 0, 150] and cnt.tolist() == [4, 0, 4] and torch.allclose(mx, torch.tensor([0.5, 0.0, 0.95]))
    broke = torch.tensor([[False, False, True], [True, False, False], [False, False, False]])
    mx, best, cnt = pkg.sharded_db.merge_shard_triples(s, i, c, broke)     # shard 0 broke for q2, shard 1 for q0
    assert best.tolist() == [7, 0, 11] and cnt.tolist() == [1, 0, 2]


def test_shard_breaks():
    pkg = load_package()
    ids = np.array([5, 9, 40, 41, 100], np.uint64)
    got = pkg.sharded_db.shard_breaks(ids, np.array([3, 5, 24, 25, 58, 59, 60, 61, 200, 10], np.uint64))
    assert got.tolist() == [False, True, True, True, True, True, True, False, False, True]
