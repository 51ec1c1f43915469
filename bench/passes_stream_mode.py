"""bench/passes_stream_mode.py — the live-stream operating points of bench.py's line (`stream_mode`): 1 / 2 / 4 pairs per step replayed on lanes that share ONE
loop database, each point measured by a child process of the run (its own GPU_MAX_HW_QUEUES = 24: the setting belongs to the process)."""
import json
import os
import subprocess
import sys

from .runtime import BENCH_PY


def stream_mode_sweep(args):
    stream_mode = None
    pts = []
    for spec in args.stream_mode.split(","):
        pp, ll = [int(v) for v in spec.lower().split("x")]
        cmd = [sys.executable, BENCH_PY, "--pairs", str(pp), "--lanes", str(ll), "--graph", "1", "--steps", str(int(os.environ.get("MYSLAM_SM_STEPS", "0")) or max(400, 1600 // pp)), "--warmup", "2",
               "--workload", args.workload, "--no-extra-passes", "--no-cpu-baseline", "--parity-frames", str(min(2, pp)), "--frame-latency", "--stream-mode", "",
               "--scene-rects", str(args.scene_rects)]
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, GPU_MAX_HW_QUEUES="24"), timeout=600)
        try:
            cd = json.loads(r.stdout.strip().splitlines()[-1])
            pts.append({"pairs_per_step": pp, "lanes": ll, "value": cd["value"], "ms_per_step": cd["ms_per_step"], "frame_latency_ms": cd["frame_latency"],
                        "host_launch_ms_per_step": cd["host_launch_ms_per_step"], "graph_nodes": cd["graph_nodes"], "repeats_ms_per_step": cd.get("repeats_ms_per_step"), "parity_ok": (cd["parity_sample"] or {}).get("ok")})
        except Exception as e:               # a failed point is reported, not hidden
            pts.append({"pairs_per_step": pp, "lanes": ll, "error": f"{type(e).__name__}: {e}", "rc": r.returncode, "stderr_tail": r.stderr[-300:]})
    if pts:
        head = dict(pts[0])
        stream_mode = dict(head, unit="stereo frames/s", sweep=pts,
                           note="recorded steps (HIP graph replay) on L lanes, step k on lane k mod L; every lane has its own extractor / DeepLCD handles and a QUERY CONTEXT "
                                "of the ONE shared loop database (myslam_lcddb_query_ctx); child processes of this run, GPU_MAX_HW_QUEUES=24.  A step of a few frames is bound by the NUMBER "
                                "of its dependent launches (~4.6 us per graph node chip-wide, whatever the node does: profiles/r05_node_count_probe.json), so the small-batch forms "
                                "of the library fold a step's launches (graph_nodes); batching frames of several cameras into one step amortises them")
    return stream_mode
