"""Multi-GPU loop database: id-range shards + one all-gather of per-shard candidates.

The reference scans ONE std::map in ascending id order keeping the first maximum (strict '>', maxScore
initialised to 0) and counting scores above the low threshold (src/loopclosing.cpp:124-161).  With the
database split into contiguous id ranges (shard r on rank r) every rank scores all queries against its
shard; the per-shard triples (maxScore, bestId, count) are all-gathered (RCCL over xGMI on GPUs, gloo in
the CPU tests) and reduced identically on every rank:
    maxScore = max over shards;  bestId = lowest id among the shards attaining it (== first in scan order);
    count    = sum of counts.
The scan's early exit ("break at the first id with cur - id < 20", :133) ends the WHOLE scan, so a shard
whose id range lies behind the breaking id must not contribute: every shard also reports whether its own
scan hit the break, and shards after the first breaking shard are ignored by the reduce.
The payload is 32 bytes per query per rank — latency bound, so it is issued once per batch of frames.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_breaks(ids_sorted, cur_ids):
    """broke[q] = the ascending scan of this shard's ids stops early for query q, i.e. some id has
    (cur - id) mod 2^64 < 20.  ids_sorted: ascending uint64 array (host)."""
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


def merge_shard_triples(scores, ids, counts, broke=None):
    """scores [S, Q] f32/f64, ids [S, Q] int64, counts [S, Q] int, broke [S, Q] bool (shard order = id order)
    -> (maxScore [Q], bestId [Q], count [Q])."""
    if broke is not None:
        b = broke.to(torch.int64)
        dead = (torch.cumsum(b, dim=0) - b) > 0                     # a shard before this one already ended the scan
        scores = torch.where(dead, torch.zeros_like(scores), scores)
        counts = torch.where(dead, torch.zeros_like(counts), counts)
    mx = scores.max(dim=0).values
    big = torch.iinfo(torch.int64).max
    cand = torch.where(scores == mx.unsqueeze(0), ids, torch.full_like(ids, big))
    best = cand.min(dim=0).values
    best = torch.where(mx > 0, best, torch.zeros_like(best))       # bestId stays 0 when nothing scored above 0
    return mx, best, counts.sum(dim=0)


def merge_candidates(best_id, max_score, count, world, broke=None, group=None, via_cpu=False):
    """In-place merge across ranks of per-shard results for the same Q queries (Q = len(best_id)).
    broke: optional bool tensor [Q] from shard_breaks() (None = this shard never hits the early exit).
    via_cpu: stage the 32-byte-per-query payload through host memory (gloo process groups)."""
    q = best_id.shape[0]
    bk = torch.zeros(q, dtype=torch.float64, device=best_id.device) if broke is None else broke.to(torch.float64)
    packed = torch.stack([max_score.to(torch.float64), best_id.to(torch.float64), count.to(torch.float64), bk], dim=1).contiguous()
    if via_cpu:
        packed = packed.cpu()
    gathered = torch.empty((world, q, 4), dtype=torch.float64, device=packed.device)
    dist.all_gather_into_tensor(gathered.view(world * q, 4), packed, group=group)
    if via_cpu:
        gathered = gathered.to(best_id.device)
    mx, best, cnt = merge_shard_triples(gathered[:, :, 0], gathered[:, :, 1].to(torch.int64), gathered[:, :, 2].to(torch.int64),
                                        gathered[:, :, 3] > 0)
    max_score.copy_(mx.to(max_score.dtype)); best_id.copy_(best.to(best_id.dtype)); count.copy_(cnt.to(count.dtype))
    return best_id, max_score, count
