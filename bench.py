#!/usr/bin/env python3
"""bench.py — stereo frames/s of the per-frame dense path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch (default 512) of synthetic 1241x376 stereo pairs resident in HBM:
  ORB DetectAndCompute (2000 features) on left+right -> L/R 256-bit Hamming match -> stereo triangulation
  -> DeepLCD descriptor of the left image -> cosine scan of the key-frame database -> local-BA block build
  (one 10 KF x 300 landmark window per frame).           [BASELINE.json configs[3]; --workload orb_match = configs[1]]

One process per GPU (launched by torch.distributed.run for --gpus N > 1).  Frames shard across ranks
(stream r on rank r, no data-path collective); the loop database is sharded by key-frame id range and the only
exchange is an all-gather of the query descriptors and of the per-shard (score, id, count) candidates (RCCL).

Prints ONE JSON line (rank 0).  PyTorch is used for device memory, streams and torch.distributed only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

H, W = 376, 1241
# SQ_INSTS_VALU per image (wave-level instructions), rocprofv3 --pmc, profiles/r01_pmc_insts_orb_match_p64_v10.txt
VALU_INSTS_PER_IMAGE = {"fast_cells": 238618128 / 128, "describe": 56505600 / 128, "octree": 13590512 / 128}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
PYR_PX = 1444097               # sum of the 8 level areas
# algorithmic bytes per IMAGE of each ORB kernel (SURVEY.md §8(d) accounting)
ALGO_BYTES = {
    "resize": 1407767 + 977481,            # read levels 0-6, write levels 1-7
    "fast_cells": PYR_PX + 4 * 20000,      # read every level once + candidate list
    "blur7": 2 * PYR_PX,                   # read + write every level
    "describe": 2000 * (749 + 512 + 60),   # IC patch + BRIEF samples + outputs
    "octree": 2 * 4 * 56000,               # candidates in, selected out (latency bound in practice)
}


def pmc_traffic(kernel, imgs_per_launch):
    """HBM bytes per launch of `kernel` from the committed PMC run (separate rocprofv3 --pmc passes, tools/gpu_traffic.sh)."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json"))
    if not files:
        return None

    def version(f):                                            # ..._v<N>.json: the highest build number is the current one
        m = re.search(r"_v(\d+)\.json$", f)
        return int(m.group(1)) if m else -1
    d = json.load(open(max(files, key=version)))
    tag = {"fast_cells": "k_fast", "octree": "k_octree", "blur7": "k_blur7", "resize": "k_resize", "describe": "k_describe"}.get(kernel)
    for name, v in d["kernels"].items():
        if tag and name.startswith(tag):
            kb = v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)
            # the PMC run counted `launches` launches of this kernel over 3 steps (tools/gpu_traffic.sh: --steps 2 --warmup 1) of
            # 2 * pairs_per_step images each: bytes per image, times the images one launch of THIS run covers
            per_image = kb * 1024.0 * v.get("launches", 3) / (d.get("steps_total", 3) * 2.0 * d["pairs_per_step"])
            return per_image * imgs_per_launch
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=512, help="stereo pairs per step per GPU")
    ap.add_argument("--workload", default="full", choices=["full", "orb_match", "orb_match_lcd", "full_solve"])
    ap.add_argument("--db", type=int, default=0, help="key-frame database size (default 10000, or 6250 per GPU when sharded)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=32)
    ap.add_argument("--streams", type=int, default=2, choices=[1, 2],
                    help="2 = the DeepLCD / loop-DB / BA chain runs on a second HIP stream beside ORB + match + triangulation")
    ap.add_argument("--orb-split", type=int, default=0, choices=[0, 1, 2, 3, 4, 8],
                    help="S > 1 = the 2P images go through S extractor handles on S streams (S equal groups): the latency-bound oct-tree / "
                         "describe launches of one group run under the VALU-bound FAST launch of the other (2 is ~3 % faster than 1; "
                         "concurrent launches stretch each other, so per-launch durations are longer than when a kernel runs alone). "
                         "0 (default) = 2 with --streams 2, 1 with --streams 1")
    ap.add_argument("--pipeline", type=int, default=-1, choices=[-1, 0, 1, 2, 3],
                    help="1 = the left and the right images go through two extractor handles that take turns (myslam_orb_set_fast_event): "
                         "one handle's VALU-bound FAST stage runs under the other's latency-bound oct-tree / descriptor stages, match + "
                         "triangulation follow on a third stream, outputs are double-buffered and consecutive steps overlap (every step's "
                         "work is complete at the closing barrier).  0 = every step is joined before the next starts.  "
                         "-1 (default) = 1 with --streams 2, else 0")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region: run one joined, un-gated step and check that it reproduces the pipeline's last outputs bit for bit")
    ap.add_argument("--side-delay-ms", type=float, default=0.0, help="experiment: start the side chain this long after the step begins (spin kernel)")
    ap.add_argument("--no-join", action="store_true", help="do not join the side stream at the end of every step (streaming across steps)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = debugging aid: several ranks share GPU 0 and the collectives go through host memory")
    args = ap.parse_args()
    if args.pipeline < 0:
        args.pipeline = 1 if (args.streams == 2 and args.orb_split != 1) else 0
    if args.orb_split == 0:
        args.orb_split = 2 if (args.streams == 2 or args.pipeline) else 1
    if args.pipeline:
        assert args.orb_split >= 2, "--pipeline runs the images on two or more extractor handles (--orb-split >= 2)"
    return args


def cpu_baseline(synth, workload, n_pairs, db_np, gpu_frames):
    """The oracle (a plain C++ port of the reference arithmetic, oracle/) timed on this host over a bounded sample of the same
    frames and stages: frame-parallel over all host cores (std::thread pool, one frame per task, oracle/bench_oracle.cpp), plus
    a single-thread figure (the reference runs every stage single-threaded inside its std::thread)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from pyoracle import Oracle
    o = Oracle()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_pairs = min(len(gpu_frames), max(n_pairs, 2 * cores))            # the same frames the GPU processed
    stages = {"orb_match": 1, "orb_match_lcd": 2, "full": 3, "full_solve": 4}[workload]
    ids = np.arange(len(db_np), dtype=np.uint64)
    args = (synth.KITTI00, synth.calc_weights(), db_np, ids, synth.ba_problem())
    n1 = min(6, n_pairs)
    dt1 = o.bench_frames(gpu_frames[:n1], *args, stages=stages, threads=1)
    dt = o.bench_frames(gpu_frames[:n_pairs], *args, stages=stages, threads=cores)
    return {"value": n_pairs / dt, "unit": "stereo frames/s", "cores": cores, "kind": "port",
            "value_1thread": n1 / dt1,
            "sample": f"{n_pairs} of the GPU run's synthetic 1241x376 stereo pairs, same stages, oracle frame-parallel on {cores} "
                      f"host threads in {dt:.1f} s (single thread: {n1} pairs in {dt1:.1f} s)"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    via_cpu = args.backend == "gloo"
    if via_cpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if via_cpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    pkg = load_package()
    api, synth = pkg.api, pkg.synth
    assert api.device_count() >= 1
    main_stream = torch.cuda.current_stream()
    stream = main_stream.cuda_stream
    # the LCD -> DB -> BA chain only reads the input images: it runs beside the ORB chain on its own stream (MFMA conv2 under the
    # VALU-bound FAST kernel, the latency-bound small kernels under each other) and is joined at the end of every step
    side_stream = torch.cuda.Stream() if args.streams == 2 else main_stream
    stream2 = side_stream.cuda_stream
    P = args.pairs
    K = synth.KITTI00
    Kt = (K["fx"], K["fy"], K["cx"], K["cy"])

    # ---- inputs resident in HBM before the timed region ----
    frames = synth.stereo_batch(P, stream_id=rank)                      # [P, 2, H, W]
    imgs = np.concatenate([frames[:, 0], frames[:, 1]], axis=0)         # all left images, then all right images
    d_imgs = torch.from_numpy(imgs).to(dev)
    ext = api.ORBextractor(2000, stream=stream)
    cap = ext.max_keypoints()
    S = args.orb_split
    assert (2 * P) % S == 0
    orb_streams = [torch.cuda.Stream() for _ in range(S - 1)]
    orb_exts = [api.ORBextractor(2000, stream=st.cuda_stream) for st in orb_streams]
    NB = 2 if args.pipeline else 1          # pipeline: extractor outputs are double-buffered (step k+1 extracts while step k is matched)
    d_kps_b = [torch.zeros(2 * P * cap * 28, dtype=torch.uint8, device=dev) for _ in range(NB)]
    d_desc_b = [torch.zeros(2 * P * cap * 32, dtype=torch.uint8, device=dev) for _ in range(NB)]
    d_cnt_b = [torch.zeros(2 * P, dtype=torch.int32, device=dev) for _ in range(NB)]
    d_stat_b = [torch.zeros(2 * P, dtype=torch.int32, device=dev) for _ in range(NB)]
    d_kps, d_desc, d_cnt, d_stat = d_kps_b[0], d_desc_b[0], d_cnt_b[0], d_stat_b[0]
    d_midx = torch.zeros(P * cap, dtype=torch.int32, device=dev)
    d_mdist = torch.zeros(P * cap, dtype=torch.int32, device=dev)
    d_xyz = torch.zeros(P * cap * 3, dtype=torch.float64, device=dev)
    d_ok = torch.zeros(P * cap, dtype=torch.uint8, device=dev)
    use_lcd = args.workload != "orb_match"
    use_ba = args.workload in ("full", "full_solve")
    use_solve = args.workload == "full_solve"
    n_db_local = args.db or (10000 if world == 1 else 6250)
    db_np = None
    if use_lcd:
        lcd = api.DeepLCD(synth.calc_weights(), stream=stream2)
        d_descr = torch.zeros(P, 1064, device=dev)
        db_np = synth.lcd_database(n_db_local, seed=0xDB + rank)
        D = api.LoopDatabase(n_db_local, stream=stream2)
        ids = np.arange(rank * n_db_local, (rank + 1) * n_db_local, dtype=np.uint64)     # contiguous id range per shard
        t_db = torch.from_numpy(db_np).to(dev)
        D.append_batch(ids, t_db.data_ptr(), n_db_local)
        NQ = P * world
        cur_ids = np.full(NQ, world * n_db_local + 20, np.uint64)
        d_allq = torch.zeros(NQ, 1064, device=dev)
        d_best = torch.zeros(NQ, dtype=torch.int64, device=dev)
        d_max = torch.zeros(NQ, device=dev); d_dbcnt = torch.zeros(NQ, dtype=torch.int32, device=dev)
    if use_ba:
        poses, pts, ep, el, obs, fixed, _ = synth.ba_problem()
        maxP, maxL, maxE = len(poses), len(pts), len(ep)
        rep = lambda a: torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (P,) + a.shape))).to(dev)
        b_in = [rep(poses), rep(pts), rep(ep), rep(el), rep(obs), rep(fixed),
                torch.tensor([[maxP, maxL, maxE]] * P, dtype=torch.int32, device=dev)]
        b_out = [torch.zeros(P, n, dtype=torch.float64, device=dev) for n in (maxP * 36, maxL * 9, maxE * 18, maxP * 6, maxL * 3, maxE)]
        # the solve updates poses/points in place: every step starts from the pristine window
        s_poses, s_pts = b_in[0].clone(), b_in[1].clone()
        s_echi = torch.zeros(P, maxE, dtype=torch.float64, device=dev); s_out = torch.zeros(P, maxE, dtype=torch.uint8, device=dev)
        s_rd = torch.zeros(P, dtype=torch.int32, device=dev); s_no = torch.zeros(P, dtype=torch.int32, device=dev)
        s_st = torch.zeros(P, dtype=torch.int32, device=dev)

        def solve():        # Backend::OptimizeActiveMap solve stage: rounds of optimize(10) + outlier flags (backend.cpp:208-243)
            s_poses.copy_(b_in[0]); s_pts.copy_(b_in[1])
            api.ba_optimize_active_map_batch(s_poses.data_ptr(), s_pts.data_ptr(), *[t.data_ptr() for t in b_in[2:]], P, maxP, maxL, maxE, Kt,
                                             5.991, 5.991, 5, 10, b_out[2].data_ptr(), s_echi.data_ptr(), s_out.data_ptr(), s_rd.data_ptr(),
                                             s_no.data_ptr(), s_st.data_ptr(), stream2)

    def side_chain():
        if args.side_delay_ms > 0 and side_stream is not main_stream:
            torch.cuda._sleep(int(args.side_delay_ms * 1e-3 * 2.0e9))
        if use_lcd:
            lcd.describe_batch(d_imgs.data_ptr(), P, H, W, W, H * W, d_descr.data_ptr(), blur_in_place=False)
            if world > 1:       # every shard scores every rank's queries; candidates are merged after an all-gather
                if via_cpu:
                    h_all = torch.empty(d_allq.shape, dtype=d_allq.dtype)
                    dist.all_gather_into_tensor(h_all, d_descr.cpu())
                    d_allq.copy_(h_all)
                else:
                    dist.all_gather_into_tensor(d_allq, d_descr)
                D.query_batch(d_allq.data_ptr(), cur_ids, NQ, d_best.data_ptr(), d_max.data_ptr(), d_dbcnt.data_ptr())
                pkg.sharded_db.merge_candidates(d_best, d_max, d_dbcnt, world, via_cpu=via_cpu)
            else:
                D.query_batch(d_descr.data_ptr(), cur_ids, NQ, d_best.data_ptr(), d_max.data_ptr(), d_dbcnt.data_ptr())
        if use_ba:
            api.ba_build_batch(*[t.data_ptr() for t in b_in], P, maxP, maxL, maxE, Kt, 5.991, *[t.data_ptr() for t in b_out], stream2)
            if use_solve:
                solve()

    if args.pipeline:
        # two extractor handles take turns: handle A (left images, main stream) and handle B (right images, its own stream) each wait
        # for the other's FAST stage, so a FAST launch never runs beside the other FAST launch but under the other handle's oct-tree /
        # descriptor launches; match + triangulation of step k run on a third stream once both handles are done with step k, while the
        # handles already extract step k+1 into the other output buffer
        exts = [ext] + orb_exts
        sX, sM = [main_stream] + orb_streams, torch.cuda.Stream()
        G = 2 * P // S
        ev_fast = [torch.cuda.Event() for _ in range(S)]
        ev_done = [[torch.cuda.Event() for _ in range(S)] for _ in range(NB)]
        ev_match = [torch.cuda.Event() for _ in range(NB)]
        ev_start = torch.cuda.Event()
        for e in ev_fast + [x for pr in ev_done for x in pr] + ev_match + [ev_start]:
            e.record(main_stream)                       # creates the hipEvent_t behind the torch event
        for i, e in enumerate(exts):
            e.set_fast_event(ev_fast[i].cuda_event)
            if args.pipeline == 1:      # gate inside the call: only the FAST stages take turns (ring), the pyramids are not held back
                e.set_fast_gate(ev_fast[(i - 1) % S].cuda_event)
        step_no = [0]

        def step():
            p = step_no[0] % NB
            step_no[0] += 1
            kps, desc, cnt, stat = d_kps_b[p], d_desc_b[p], d_cnt_b[p], d_stat_b[p]
            for i, e in enumerate(exts):
                if args.pipeline == 2:
                    sX[i].wait_event(ev_fast[(i - 1) % S])      # the previous handle's FAST (whole call held back)
                sX[i].wait_event(ev_match[p])                   # buffer p was last read by the match of step k - 2
                if i == 0:
                    ev_start.record(sX[0])
                o = i * G
                e.detect_and_compute_batch(d_imgs.data_ptr() + o * H * W, G, H, W, W, H * W, kps.data_ptr() + o * cap * 28,
                                           desc.data_ptr() + o * cap * 32, cnt.data_ptr() + 4 * o, stat.data_ptr() + 4 * o, cap)
                ev_done[p][i].record(sX[i])
            for i in range(S):
                sM.wait_event(ev_done[p][i])
            api.hamming_match_batch(desc.data_ptr(), cnt.data_ptr(), desc.data_ptr() + P * cap * 32, cnt.data_ptr() + 4 * P,
                                    P, cap, d_midx.data_ptr(), d_mdist.data_ptr(), sM.cuda_stream)
            api.triangulate_stereo_batch(kps.data_ptr(), kps.data_ptr() + P * cap * 28, d_midx.data_ptr(), cnt.data_ptr(), P, cap,
                                         Kt, K["bf"] / K["fx"], d_xyz.data_ptr(), d_ok.data_ptr(), sM.cuda_stream)
            ev_match[p].record(sM)
            side_stream.wait_event(ev_start)            # the LCD / DB / BA chain of step k starts with step k
            with torch.cuda.stream(side_stream):
                side_chain()

    def step_joined():
        if S == 1:
            ext.detect_and_compute_batch(d_imgs.data_ptr(), 2 * P, H, W, W, H * W, d_kps.data_ptr(), d_desc.data_ptr(),
                                         d_cnt.data_ptr(), d_stat.data_ptr(), cap)
        else:               # S equal groups of images, group 0 on the main stream
            G = 2 * P // S
            for st in orb_streams:
                st.wait_stream(main_stream)
            for gi, e in enumerate([ext] + orb_exts):
                o = gi * G
                e.detect_and_compute_batch(d_imgs.data_ptr() + o * H * W, G, H, W, W, H * W, d_kps.data_ptr() + o * cap * 28,
                                           d_desc.data_ptr() + o * cap * 32, d_cnt.data_ptr() + 4 * o, d_stat.data_ptr() + 4 * o, cap)
            for st in orb_streams:
                main_stream.wait_stream(st)
        api.hamming_match_batch(d_desc.data_ptr(), d_cnt.data_ptr(), d_desc.data_ptr() + P * cap * 32, d_cnt.data_ptr() + 4 * P,
                                P, cap, d_midx.data_ptr(), d_mdist.data_ptr(), stream)
        api.triangulate_stereo_batch(d_kps.data_ptr(), d_kps.data_ptr() + P * cap * 28, d_midx.data_ptr(), d_cnt.data_ptr(), P, cap,
                                     Kt, K["bf"] / K["fx"], d_xyz.data_ptr(), d_ok.data_ptr(), stream)
        with torch.cuda.stream(side_stream):
            side_chain()
        if side_stream is not main_stream and not args.no_join:
            main_stream.wait_stream(side_stream)            # a step is complete when both chains are

    if not args.pipeline:
        step = step_joined

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    assert all(int(t.abs().sum()) == 0 for t in d_stat_b), "ORB capacity overflow"
    n_kp = d_cnt.float().mean().item()

    api.prof_reset(); api.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    api.prof_enable(False)
    prof = api.prof_read()
    if args.verify and args.pipeline:
        # the overlapped schedule must not change a single output: one plain step (handles un-gated, joined) against the last pipelined one
        p_last = (step_no[0] - 1) % NB
        outs = lambda p: [d_kps_b[p], d_desc_b[p], d_cnt_b[p], d_stat_b[p], d_midx, d_mdist, d_xyz, d_ok] + \
            ([d_descr, d_best, d_max, d_dbcnt] if use_lcd else []) + (list(b_out) if use_ba else [])
        ref = [t.clone() for t in outs(p_last)]
        for e in exts:
            e.set_fast_event(0); e.set_fast_gate(0)
        for t in outs(0):
            t.zero_()
        step_joined()
        torch.cuda.synchronize()
        n_exact = len(ref) - (len(b_out) if use_ba else 0)        # the BA blocks are f64 atomic sums: equal up to the order of the additions
        for i, (a, r) in enumerate(zip(outs(0), ref)):
            if i < n_exact:
                assert torch.equal(a.view(torch.uint8), r.view(torch.uint8)), f"pipeline output {i} differs from the joined step"
            else:
                assert torch.allclose(a, r, rtol=1e-10, atol=1e-10 * float(r.abs().max())), f"pipeline BA output {i} differs from the joined step"
        for i, e in enumerate(exts):
            e.set_fast_event(ev_fast[i].cuda_event)
            if args.pipeline == 1:
                e.set_fast_gate(ev_fast[(i - 1) % S].cuda_event)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if via_cpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    solve_ms = None
    if use_ba and not use_solve:        # the "g2o solve" half of configs[3], timed on its own (not part of `value`)
        with torch.cuda.stream(side_stream):
            solve(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                solve()
            torch.cuda.synchronize()
        solve_ms = (time.perf_counter() - t1) / args.steps * 1e3
        assert int(s_st.abs().sum()) == 0

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = world * P * args.steps / dt
        # dominant kernel and its roofline position (HIP events on the launch stream, over the timed region)
        busy = {k: v for k, v in prof.items() if v[1] > 0}
        # the dominant kernel of the critical (ORB) stream; with --streams 2 the side chain's kernels run underneath it and their
        # event-timed durations are stretched by the sharing, so they are not candidates
        chain = [k for k in busy if k in ("resize", "fast_cells", "octree", "blur7", "describe", "hamming_match", "triangulate")]
        # event-timed durations of overlapped launches say how long a kernel was resident, not how much of the chip it used (the
        # latency-bound oct-tree runs under FAST for as long as FAST takes): the dominant kernel is the one with the largest
        # instruction volume (PMC, VALU_INSTS_PER_IMAGE) among those that ran, by duration only if none of them is in that table
        dom = max(chain or busy, key=lambda k: (VALU_INSTS_PER_IMAGE.get(k, 0), busy[k][0]))
        dom_ms, dom_n = busy[dom]
        per_launch_ms = dom_ms / dom_n
        imgs_per_launch = 2 * P
        launches_per_step = dom_n / args.steps
        if dom in ALGO_BYTES:
            algo = ALGO_BYTES[dom] * imgs_per_launch / launches_per_step        # bytes per launch
            achieved = algo / (per_launch_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(dom, imgs_per_launch / launches_per_step), "avg_launch_ms": per_launch_ms,
                    "algorithmic_bytes_per_launch": algo,
                    "note": "packed-integer VALU bound in practice (see roofline_valu and DESIGN.md section 6); avg_launch_ms is the event-timed duration of one launch (the left and the right images go through two extractor handles on two streams unless --orb-split 1, so a launch covers half of the step's images), which shares the chip with the other handle's launches, the Gaussian-pyramid launches of the extractors' internal streams and the LCD / DB / BA chain (3.28 ms for all 1024 images when it runs alone: --streams 1 with MYSLAM_ORB_AUX=0); traffic = FETCH_SIZE+WRITE_SIZE of a separate rocprofv3 --pmc run (profiles/), uncorrected"}
        else:
            roof = {"bound": "hbm", "kernel": dom, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                    "traffic": None, "avg_launch_ms": per_launch_ms}
        out = {
            "metric": "stereo frames/sec (ORB+match+LCD+BA-build) @1241x376",
            "value": value, "unit": "stereo frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (ORB, Hamming), f32 (CALC, DB scan), f64 (triangulation, BA)", "data": "synthetic",
            "config": {"workload": {"full": "configs[3]: ORB extract L+R (2000 feats) + L/R Hamming match + triangulation + DeepLCD descriptor + "
                                            f"{n_db_local * world}-KF cosine DB scan + local-BA (10 KF x 300 MP) block build per frame",
                                    "full_solve": "configs[3] incl. solve: as 'full' + the Backend::OptimizeActiveMap solve stage (rounds of Levenberg-Marquardt "
                                                  "optimize(10) with Schur + Cholesky, outlier flags) of the 10 KF x 300 MP window per frame",
                                    "orb_match": "configs[1]: ORB extract L+R (2000 feats) + L/R Hamming match + triangulation",
                                    "orb_match_lcd": f"configs[2]: configs[1] + DeepLCD descriptor + {n_db_local * world}-KF cosine DB scan"}[args.workload],
                       "pairs_per_step_per_gpu": P, "image": "1241x376 u8", "keypoints_per_image": n_kp,
                       "hip_streams": args.streams, "orb_extractor_handles": S,
                       "parallelism": f"frame-sharded x{world}" + (", id-range sharded DB + all-gather of candidates" if world > 1 else "")},
            "roofline": roof,
            "roofline_mfma": None if "calc_conv2" not in busy else {
                "bound": "mfma", "kernel": "calc_conv2", "peak": 2500.0, "unit": "TFLOP/s (bf16, dense)",
                "achieved": 6 * 2 * 176160768 * P / (busy["calc_conv2"][0] / busy["calc_conv2"][1] * 1e-3) / 1e12,
                "frac": 6 * 2 * 176160768 * P / (busy["calc_conv2"][0] / busy["calc_conv2"][1] * 1e-3) / 1e12 / 2500.0,
                "effective_f32_tflops": 2 * 176160768 * P / (busy["calc_conv2"][0] / busy["calc_conv2"][1] * 1e-3) / 1e12,
                "avg_launch_ms": busy["calc_conv2"][0] / busy["calc_conv2"][1],
                "note": "CALC conv2 as an implicit GEMM on the bf16 matrix cores with f32 accuracy: every f32 operand is split exactly into three "
                        "bf16 pieces and the six largest partial products are accumulated in f32 (6 bf16 MFMA flops per f32 flop; error against an "
                        "f64 reference 1.5e-6, the same as the f32-input MFMA it replaces); with --streams 2 it shares the CUs with the ORB "
                        "kernels, so this duration is stretched (0.93 ms when it runs alone, --streams 1)"},
            # the issue-rate view of the same dominant kernel: wave-level VALU instructions per image from a separate
            # `rocprofv3 --pmc SQ_INSTS_VALU` run (profiles/r01_pmc_insts_orb_match_p64_v10.txt), x 64 lanes, against 256 CUs x 4 SIMDs x
            # 16 lanes per cycle at 2.4 GHz
            "roofline_valu": None if dom not in VALU_INSTS_PER_IMAGE else {
                "bound": "valu", "kernel": dom, "unit": "Tlane-op/s", "peak": 39.3,
                "achieved": VALU_INSTS_PER_IMAGE[dom] * imgs_per_launch / launches_per_step * 64 / (per_launch_ms * 1e-3) / 1e12,
                "frac": VALU_INSTS_PER_IMAGE[dom] * imgs_per_launch / launches_per_step * 64 / (per_launch_ms * 1e-3) / 1e12 / 39.3,
                "valu_wave_insts_per_image": VALU_INSTS_PER_IMAGE[dom]},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in busy.items()},
            "ba_solve_ms_per_step": solve_ms,     # OptimizeActiveMap solve stage for the same windows, outside the timed region
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(synth, args.workload, args.cpu_pairs, db_np if db_np is not None else synth.lcd_database(16), frames)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
