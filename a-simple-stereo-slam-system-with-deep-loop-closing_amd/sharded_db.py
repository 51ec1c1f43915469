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
