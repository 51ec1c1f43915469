"""Multi-GPU loop database: id-range shards + one all-gather of per-shard candidates.

The reference scans ONE std::map in ascending id order keeping the first maximum (strict '>', maxScore
initialised to 0) and counting scores above the low threshold (src/loopclosing.cpp:124-161).  With the
database split into contiguous id ranges (shard r on rank r) every rank scores all queries against its
shard (`myslam_lcddb_query_batch_sharded`), the 16-byte per-shard records (`myslam_lcd_candidate`:
bestId u64, maxScore f32, count + "my scan hit the break" flag) are all-gathered as raw bytes (RCCL over
xGMI on GPUs, gloo in the CPU tests) and reduced identically on every rank by the library
(`myslam_lcd_merge_candidates[_device]`, csrc/lcddb.hip):
    maxScore = first maximum in shard order (== lowest id, == first in the reference's scan order);
    count    = sum of counts;
    the scan's early exit ("break at the first id with cur - id < 20", :133) ends the WHOLE scan, so shards
    behind the first shard whose own scan hit the break do not contribute.
This module is only the collective plumbing (torch.distributed); ids never leave their integer type.
The payload is 16 bytes per query per rank — latency bound, so it is issued once per batch of frames.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import api


def shard_breaks(ids_sorted, cur_ids):
    """broke[q] = the ascending scan of this shard's ids stops early for query q, i.e. some id has
    (cur - id) mod 2^64 < 20.  ids_sorted: ascending uint64 array (host).  The library computes the same flag
    itself inside myslam_lcddb_query_batch_sharded; this is the host-side statement of the rule for callers
    that build the records on their own."""
    ids = np.asarray(ids_sorted, np.uint64)
    cur = np.asarray(cur_ids, np.uint64)
    out = np.zeros(len(cur), bool)
    for i, c in enumerate(cur.tolist()):
        ranges = [(c - 19, c)] if c >= 19 else [(0, c), (2 ** 64 - (19 - c), 2 ** 64 - 1)]
        for a, b in ranges:
            j = int(np.searchsorted(ids, np.uint64(a), side="left"))
            if j < len(ids) and int(ids[j]) <= b:
                out[i] = True
    return out


def pack_candidates(best_id, max_score, count, broke=None):
    """numpy arrays -> api.CAND_DTYPE records (host); count bit 31 carries the break flag"""
    n = len(best_id)
    rec = np.zeros(n, api.CAND_DTYPE)
    rec["best_id"] = np.asarray(best_id, np.uint64)
    rec["max_score"] = np.asarray(max_score, np.float32)
    cnt = np.asarray(count, np.int64) & 0x7fffffff
    if broke is not None:
        cnt = cnt | (np.asarray(broke, bool).astype(np.int64) << 31)
    rec["cnt"] = cnt.astype(np.uint32).view(np.int32)
    return rec


def check_shard_order(first_id, last_id, world, via_cpu=False, device=None, group=None):
    """Every rank's id range [first_id, last_id] must lie before the next rank's (rank order == id order is what the merge assumes).
    Ids travel bit-cast to int64 and are compared as unsigned."""
    mine = torch.from_numpy(np.array([first_id, last_id], np.uint64).view(np.int64).copy())
    if not via_cpu:
        mine = mine.to(device)
    allr = torch.empty(2 * world, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(allr, mine, group=group)
    r = allr.cpu().numpy().view(np.uint64).reshape(world, 2)
    for k in range(world):
        assert r[k, 0] <= r[k, 1], f"shard {k}: empty or reversed id range {r[k]}"
        assert k == 0 or r[k - 1, 1] < r[k, 0], f"shard id ranges must ascend with rank: {r[k - 1]} !< {r[k]}"
    return r


def exchange_and_merge(cand, world, best_id, max_score, count, via_cpu=False, group=None):
    """cand: uint8 tensor of nq * 16 bytes (this shard's myslam_lcd_candidate records, from query_batch_sharded).
    All-gathers the records of every shard and reduces them into best_id (int64 storage of the u64 ids) / max_score (f32) /
    count (i32), all of length nq, on every rank.  via_cpu stages through host memory (gloo process groups): the reduce then runs
    in the library's host entry point; otherwise everything stays on the device, ordered on torch's current stream."""
    nq = best_id.shape[0]
    assert cand.dtype == torch.uint8 and cand.numel() == nq * 16
    if via_cpu:
        mine = cand.cpu()
        gathered = torch.empty(world * nq * 16, dtype=torch.uint8)
        dist.all_gather_into_tensor(gathered, mine, group=group)
        b, m, c = api.lcd_merge_candidates(gathered.numpy().view(api.CAND_DTYPE).reshape(world, nq))
        best_id.copy_(torch.from_numpy(b.view(np.int64))); max_score.copy_(torch.from_numpy(m)); count.copy_(torch.from_numpy(c))
    else:
        gathered = torch.empty(world * nq * 16, dtype=torch.uint8, device=cand.device)
        dist.all_gather_into_tensor(gathered, cand, group=group)
        api.lcd_merge_candidates_device(gathered.data_ptr(), world, nq, best_id.data_ptr(), max_score.data_ptr(), count.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
    return best_id, max_score, count


# ---- a sharded database that GROWS (round 6) -----------------------------------------------------------------------------------------------------
# The reference appends every key-frame to its one std::map (LoopClosing::AddToDatabase, src/loopclosing.cpp:651-659).  Under N ranks:
#   * OWNERSHIP BY ARRIVAL: the k-th key-frame the JOB has appended goes to rank k mod N.  Every rank sees the same all-gathered list of new key-frames
#     per step (sorted by id), so every rank computes the same owners with no extra collective, and the shards' row counts never differ by more than 1
#     whatever the id pattern (ids with gaps, streams that produce key-frames at different rates).  Ids only have to grow from step to step (they do:
#     KeyFrame ids come from one counter, src/keyframe.cpp:14), which keeps each shard's own ids ascending.
#   * a shard's ids now INTERLEAVE with the others': it answers with a 32-byte `myslam_lcd_owned_candidate` (both parts of the reference's scan + the break
#     flag, include/myslam_hip.h) and the merge is `myslam_lcd_merge_owned_candidates[_device]` — bit-identical to ONE ascending scan of the whole map.
# Per step and rank: all-gather of the new key-frames (id, valid flag, descriptor) -> every shard scores every query -> all-gather of the records ->
# merge -> append what this rank owns.  Queries of a step see the database as it was before the step's appends (DetectLoop runs before AddToDatabase,
# src/loopclosing.cpp:83-121).
class HipShard:
    """this rank's rows in a myslam_lcddb (device); queries / rows are torch device tensors"""

    def __init__(self, capacity=256, stream=0, device=None):
        self.db = api.LoopDatabase(capacity, stream=stream)
        self.device = device

    def rows(self):
        return len(self.db)

    def records(self, q_all, cur_ids):
        nq = len(cur_ids)
        rec = torch.empty(nq * 32, dtype=torch.uint8, device=q_all.device)
        self.db.query_batch_owned(q_all.data_ptr(), cur_ids, nq, rec.data_ptr())
        return rec

    def append(self, ids, rows):
        if len(ids):
            rows = rows.contiguous()
            self.db.append_batch(np.ascontiguousarray(ids, np.uint64), rows.data_ptr(), len(ids))


def owned_records_host(ids_sorted, scan, q, cur_ids):
    """The host statement of what myslam_lcddb_query_batch_owned returns, for shards whose scan is not the library's (the CPU tests use the oracle):
    scan(lo, hi, query, cur) -> (best_id, max_score, cnt) of rows [lo, hi) of this shard.  Returns api.OWNED_DTYPE records."""
    ids = np.asarray(ids_sorted, np.uint64)
    rec = np.zeros(len(cur_ids), api.OWNED_DTYPE)
    for i, c in enumerate(np.asarray(cur_ids, np.uint64).tolist()):
        lo = c - 19 if c >= 19 else 0
        p = int(np.searchsorted(ids, np.uint64(lo), side="left"))
        broke = p < len(ids) and int(ids[p]) <= c
        sb = int(np.searchsorted(ids, np.uint64(c), side="right"))
        se = len(ids)
        if c < 19:
            se = max(sb, int(np.searchsorted(ids, np.uint64(2 ** 64 - 1 - (18 - c)), side="left")))
        if p > 0:
            b, m, n = scan(0, p, q[i], c)
            rec["pre_best_id"][i], rec["pre_max_score"][i], rec["pre_cnt"][i] = b, m, n
        if broke:
            rec["pre_cnt"][i] = np.array(int(rec["pre_cnt"][i]) | (1 << 31), np.uint32).view(np.int32)
        if se > sb:
            b, m, n = scan(sb, se, q[i], c)
            rec["suf_best_id"][i], rec["suf_max_score"][i], rec["suf_cnt"][i] = b, m, n
    return rec


class GrowingShardedDatabase:
    """One rank's view of a loop database that `world` ranks grow together (see the block comment above).  `shard` = HipShard or any object with
    rows() / records(q_all, cur_ids) / append(ids, rows); via_cpu: the collectives run on host tensors (gloo), else on device tensors (RCCL)."""

    def __init__(self, shard, world, rank, via_cpu=False, group=None):
        self.shard, self.world, self.rank, self.via_cpu, self.group = shard, world, rank, via_cpu, group
        self.total = 0                     # key-frames the JOB has appended so far (identical on every rank)

    def owner(self, k):
        return k % self.world

    def step(self, ids, descr, nvalid, thr_low=0.92):
        """ids: uint64 [P] (host), descr: [P, 1064] f32 (host array when via_cpu, else device tensor), nvalid <= P of them are this rank's new key-frames
        of the step (the P slots keep the collectives' shapes fixed).  Returns (best_id u64, max_score f32, cnt i32) host arrays for the nvalid own queries."""
        P, W = len(ids), self.world
        meta = np.zeros(P + 1, np.int64); meta[0] = nvalid; meta[1:] = np.asarray(ids, np.uint64).view(np.int64)
        t_meta = torch.from_numpy(meta)
        t_descr = torch.from_numpy(np.ascontiguousarray(descr, np.float32)) if self.via_cpu else descr
        if not self.via_cpu:
            t_meta = t_meta.to(t_descr.device)
        all_meta = torch.empty(W * (P + 1), dtype=torch.int64, device=t_meta.device)
        all_descr = torch.empty(W * P, 1064, dtype=torch.float32, device=t_descr.device)
        if W > 1:
            dist.all_gather_into_tensor(all_meta, t_meta, group=self.group)
            dist.all_gather_into_tensor(all_descr, t_descr.contiguous(), group=self.group)
        else:
            all_meta.copy_(t_meta); all_descr.copy_(t_descr)
        am = all_meta.cpu().numpy().reshape(W, P + 1)
        nv = am[:, 0].astype(int)
        all_ids = am[:, 1:].copy().view(np.uint64)
        # every shard scores every slot (invalid slots carry id 0 and are ignored afterwards: fixed shapes, no second collective)
        cur = all_ids.reshape(-1).copy()
        NQ = W * P
        rec = self.shard.records(all_descr if not self.via_cpu else all_descr.numpy(), cur)
        rec_t = rec if isinstance(rec, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1).copy())
        gathered = torch.empty(W * NQ * 32, dtype=torch.uint8, device=rec_t.device)
        if W > 1:
            dist.all_gather_into_tensor(gathered, rec_t, group=self.group)
        else:
            gathered.copy_(rec_t)
        if self.via_cpu:
            best, mx, cnt = api.lcd_merge_owned_candidates(gathered.numpy().view(api.OWNED_DTYPE).reshape(W, NQ))
        else:
            d_best = torch.empty(NQ, dtype=torch.int64, device=gathered.device); d_mx = torch.empty(NQ, device=gathered.device)
            d_cnt = torch.empty(NQ, dtype=torch.int32, device=gathered.device)
            api.lcd_merge_owned_candidates_device(gathered.data_ptr(), W, NQ, d_best.data_ptr(), d_mx.data_ptr(), d_cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
            best, mx, cnt = d_best.cpu().numpy().view(np.uint64), d_mx.cpu().numpy(), d_cnt.cpu().numpy()
        # appends: the step's valid key-frames in id order; the j-th goes to rank (total + j) mod world
        slots = [(int(all_ids[r, j]), r * P + j) for r in range(W) for j in range(nv[r])]
        slots.sort()
        assert all(a[0] < b[0] for a, b in zip(slots, slots[1:])), "key-frame ids of one step must be distinct"
        mine = [(kid, s) for j, (kid, s) in enumerate(slots) if self.owner(self.total + j) == self.rank]
        if mine:
            idx = [s for _, s in mine]
            rows = all_descr[torch.tensor(idx, device=all_descr.device)] if not self.via_cpu else all_descr.numpy()[idx]
            self.shard.append(np.array([k for k, _ in mine], np.uint64), rows)
        self.total += len(slots)
        o = self.rank * P
        return best[o:o + nvalid], mx[o:o + nvalid], cnt[o:o + nvalid]
