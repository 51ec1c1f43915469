"""bench/passes_multirank.py — multi-rank runs of bench.py: what the loop-database exchange costs, stage by stage, on an otherwise idle chip (after the timed
region).  The two all-gathers (queries: P x 4 256 B per rank in, N P x 4 256 B out; candidates: N P x 16 B per rank) and the N x larger scan are the
only work a rank does for the others: scaling efficiency below 1 is these numbers (DESIGN.md section 4 holds the predicted values).  The same four
stages in C++ + librccl: app/sharded_db_rccl.cpp."""
import time

import numpy as np
import torch
import torch.distributed as dist


def db_exchange(api, D, world, via_cpu, dev, side_stream, d_allq, d_descr, d_cand, d_best, d_max, d_dbcnt, cur_ids, NQ, P, n_db_local):
    reps = 20
    acc = np.zeros(4)
    with torch.cuda.stream(side_stream):
        for it in range(-2, reps):
            if via_cpu:      # gloo stages through the host: wall clock around synchronised stages
                torch.cuda.synchronize(); t_ = [time.perf_counter()]
                h_all = torch.empty(d_allq.shape, dtype=d_allq.dtype); dist.all_gather_into_tensor(h_all, d_descr.cpu()); d_allq.copy_(h_all)
                torch.cuda.synchronize(); t_.append(time.perf_counter())
                D.query_batch_sharded(d_allq.data_ptr(), cur_ids, NQ, d_cand.data_ptr())
                torch.cuda.synchronize(); t_.append(time.perf_counter())
                mine = d_cand.cpu(); gathered = torch.empty(world * NQ * 16, dtype=torch.uint8); dist.all_gather_into_tensor(gathered, mine)
                t_.append(time.perf_counter())
                b_, m_, c_ = api.lcd_merge_candidates(gathered.numpy().view(api.CAND_DTYPE).reshape(world, NQ))
                t_.append(time.perf_counter())
                ms = [(t_[k + 1] - t_[k]) * 1e3 for k in range(4)]
            else:            # RCCL: HIP events on the stream the collectives and the two library calls are ordered on
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                ev[0].record(side_stream)
                dist.all_gather_into_tensor(d_allq, d_descr); ev[1].record(side_stream)
                D.query_batch_sharded(d_allq.data_ptr(), cur_ids, NQ, d_cand.data_ptr()); ev[2].record(side_stream)
                gathered = torch.empty(world * NQ * 16, dtype=torch.uint8, device=dev)
                dist.all_gather_into_tensor(gathered, d_cand); ev[3].record(side_stream)
                api.lcd_merge_candidates_device(gathered.data_ptr(), world, NQ, d_best.data_ptr(), d_max.data_ptr(), d_dbcnt.data_ptr(), side_stream.cuda_stream)
                ev[4].record(side_stream)
                side_stream.synchronize()
                ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(4)]
            if it >= 0:
                acc += np.array(ms)
    acc /= reps
    mine_t = torch.tensor(acc, dtype=torch.float64, device="cpu" if via_cpu else dev)
    all_t = torch.empty(world * 4, dtype=torch.float64, device=mine_t.device)
    dist.all_gather_into_tensor(all_t, mine_t)
    all_t = all_t.cpu().numpy().reshape(world, 4)
    collective = {"collective_ms_per_step": float((all_t[:, 0] + all_t[:, 2]).max()), "shard_scan_ms_per_step": float(all_t[:, 1].max()),
                  "merge_ms_per_step": float(all_t[:, 3].max()),
                  "allgather_queries_ms": [float(v) for v in all_t[:, 0]], "allgather_candidates_ms": [float(v) for v in all_t[:, 2]],
                  "shard_scan_ms": [float(v) for v in all_t[:, 1]],
                  "allgather_queries_bytes_per_rank": int(P * 1064 * 4), "allgather_candidates_bytes_per_rank": int(NQ * 16),
                  "shard_rows": int(n_db_local), "queries_scanned_per_rank": int(NQ), "reps": reps,
                  "timing": "wall clock around synchronised stages (gloo stages through host memory)" if via_cpu else "HIP events on the side stream",
                  "note": "measured after the timed region on an otherwise idle chip: the cost of the loop-database exchange alone; inside a step it runs on the side "
                          "stream under the extractor"}
    return collective
