#!/usr/bin/env python3
"""Where do the idle vector-issue slots of a step sit?  (task 1b)

    python tools/valu_timeline.py --tag v61 [--slice-us 250] [--pairs 512]

1. `rocprofv3 --kernel-trace` (no counters: counters serialise the dispatches) around the PIPELINED bench step (four streams, 512 pairs): start / end
   timestamp, queue and grid of every dispatch.
2. the per-kernel instruction volume from the committed counter summary (profiles/r<NN>_pmc_*.json: SQ_INSTS_VALU per image / per pair, collected in a
   one-stream run of the same kernels) — a dispatch of kernel k over u units issues u x insts(k) wave-instructions.
3. every dispatch's instructions are spread uniformly over its residency [start, end]; a slice's demand = sum over the dispatches that overlap it.  Issue
   capacity of a slice = 1024 SIMDs x clock x slice / 4 cycles per wave64 instruction.
Writes gpurun_out/r06_valu_timeline_<tag>.json: per 0.25 ms slice of three steady-state steps the modelled issue utilisation, the kernels resident
in the slice (by stream role) and the instruction share of each; plus per step the residency intervals of every dispatch (what runs beside what) and
the summary the DESIGN text quotes: utilisation while FAST is resident with / without a co-runner, hand-off gaps between the two handles' FAST launches.
The uniform-spread assumption is the model's limit: a latency-bound kernel issues in bursts.  It is exact for the totals and for which kernels are resident."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORB = ("k_ingest", "k_resize", "k_fast", "k_octree", "k_blur7", "k_describe")
CLOCK_GHZ = 2.4
SIMDS = 1024


def short(name):
    return name.split("(")[0].replace("void ", "").replace("myslam_hip::", "").strip()


def newest_pmc():
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")):
        m = re.search(r"r(\d+)_pmc_.*?(\d+)\.json$", f)
        if not m:
            continue
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if "kernels" in d and (best is None or (int(m.group(1)), int(m.group(2))) > best[0]):
            best = ((int(m.group(1)), int(m.group(2))), d, os.path.relpath(f, ROOT))
    return (best[1], best[2]) if best else (None, None)


def trace(pairs, steps, outdir, extra):
    if os.path.isdir(outdir):
        shutil.rmtree(outdir)
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", outdir, "-o", "t", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--pairs", str(pairs), "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline", "--no-extra-passes", "--parity-frames", "0", "--graph", "0"] + extra
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    files = glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True)
    if r.returncode != 0 or not files:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
        raise SystemExit(f"rocprofv3 kernel trace failed (rc {r.returncode})")
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    rows = list(csv.DictReader(open(files[0])))
    shutil.rmtree(outdir, ignore_errors=True)
    return rows, line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--pairs", type=int, default=512)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--slice-us", type=float, default=250.0)
    ap.add_argument("--bench-arg", action="append", default=[])
    ap.add_argument("--from-csv", default="", help="debug: analyse an existing kernel_trace.csv instead of running rocprofv3")
    args = ap.parse_args()
    pmc, pmc_path = newest_pmc()
    if not pmc:
        raise SystemExit("no profiles/r*_pmc_*.json")
    insts = {}              # kernel prefix -> (wave-instructions per unit, unit)
    for k, rec in pmc["kernels"].items():
        u = rec["unit"]
        insts[k] = (rec.get("valu_wave_insts_per_" + u, 0.0) / max(1, rec.get("launches_per_step", 1)), u)
    if args.from_csv:
        rows, line = list(csv.DictReader(open(args.from_csv))), None
    else:
        rows, line = trace(args.pairs, args.steps, os.path.join(ROOT, "gpurun_out", "valu_tl_tmp"), args.bench_arg)
    P = args.pairs
    disp = []
    for r in rows:
        n = short(r["Kernel_Name"])
        if not n.startswith("k_"):
            continue
        key = next((k for k in insts if n.startswith(k.split("<")[0])), None)
        per_unit, unit = insts.get(key, (0.0, "pair"))
        # units of a dispatch: the pipelined step runs P images per extractor call (two handles) and P pairs per side-chain call
        units = P
        disp.append({"k": n, "q": r.get("Queue_Id", "?"), "a": int(r["Start_Timestamp"]), "b": int(r["End_Timestamp"]), "insts": per_unit * units})
    disp.sort(key=lambda d: d["a"])
    # steady state: steps are delimited by the FAST launches (2 per step, alternating handles): take FAST launches 2 x (warmup + 4) ... + 6
    fast = [d for d in disp if d["k"].startswith("k_fast")]
    if len(fast) < 2 * 14:
        raise SystemExit(f"only {len(fast)} FAST dispatches in the trace")
    i0 = 2 * 9
    t0, t1 = fast[i0]["a"], fast[i0 + 6]["a"]               # three whole steps
    win = [d for d in disp if d["b"] > t0 and d["a"] < t1]
    sl = args.slice_us * 1e3
    nsl = int((t1 - t0 + sl - 1) // sl)
    cap = SIMDS * CLOCK_GHZ * 1e9 * (sl * 1e-9) / 4.0       # wave-instructions a slice can issue
    slices = []
    for s in range(nsl):
        a, b = t0 + s * sl, min(t1, t0 + (s + 1) * sl)
        dem = collections.defaultdict(float)
        for d in win:
            ov = min(b, d["b"]) - max(a, d["a"])
            if ov > 0 and d["b"] > d["a"]:
                dem[d["k"].split("<")[0]] += d["insts"] * ov / (d["b"] - d["a"])
        tot = sum(dem.values())
        slices.append({"t_ms": round((a - t0) / 1e6, 3), "util": round(tot / (cap * (b - a) / sl), 4),
                       "share": {k: round(v / tot, 3) for k, v in sorted(dem.items(), key=lambda kv: -kv[1]) if tot and v / tot >= 0.02}})
    # FAST residency and what the hand-off between the two handles costs
    fw = fast[i0:i0 + 7]
    handoff = [(fw[i + 1]["a"] - fw[i]["b"]) / 1e3 for i in range(6)]                       # us between one handle's FAST ending and the other's starting
    fast_res = sum(d["b"] - d["a"] for d in fw[:6]) / 1e6
    total_insts = sum(d["insts"] * (min(t1, d["b"]) - max(t0, d["a"])) / max(1, d["b"] - d["a"]) for d in win)
    step_ms = (t1 - t0) / 3e6
    by_k = collections.defaultdict(lambda: [0.0, 0, 0.0])
    for d in win:
        if d["a"] >= t0 and d["a"] < t1:
            e = by_k[d["k"].split("<")[0]]
            e[0] += (d["b"] - d["a"]) / 1e6; e[1] += 1; e[2] += d["insts"]
    util_sorted = sorted(s["util"] for s in slices)
    out = {
        "build": args.tag, "pairs_per_step": P, "slice_us": args.slice_us, "clock_ghz_assumed": CLOCK_GHZ, "simds": SIMDS,
        "counter_source": pmc_path, "bench_line_ms_per_step_under_trace": None if not line else line.get("ms_per_step"),
        "window": {"steps": 3, "ms": round((t1 - t0) / 1e6, 3), "step_ms": round(step_ms, 3),
                   "valu_wave_insts": total_insts, "issue_utilisation": round(total_insts / (SIMDS * CLOCK_GHZ * 1e9 * (t1 - t0) * 1e-9 / 4.0), 4)},
        "fast": {"launches": 6, "resident_ms_total": round(fast_res, 3), "resident_fraction_of_window": round(fast_res / ((t1 - t0) / 1e6), 3),
                 "handoff_gap_us": [round(h, 1) for h in handoff],
                 "note": "handoff_gap_us = time between the end of one handle's FAST dispatch and the start of the other handle's (negative = overlap)"},
        "per_kernel_in_window": {k: {"resident_ms": round(v[0], 3), "dispatches": v[1], "valu_wave_insts": v[2]} for k, v in sorted(by_k.items(), key=lambda kv: -kv[1][2])},
        "utilisation_quantiles": {"min": util_sorted[0], "p10": util_sorted[len(util_sorted) // 10], "median": util_sorted[len(util_sorted) // 2],
                                  "p90": util_sorted[(9 * len(util_sorted)) // 10], "max": util_sorted[-1]},
        "slices": slices,
        "dispatches_step0": [{"k": d["k"].split("<")[0], "q": d["q"], "start_ms": round((d["a"] - t0) / 1e6, 3), "end_ms": round((d["b"] - t0) / 1e6, 3)}
                             for d in win if d["a"] < t0 + (t1 - t0) / 3 and d["b"] > t0],
        "method": "kernel trace (no counters) of the pipelined step; per-dispatch VALU volume = units x SQ_INSTS_VALU per unit of the committed one-stream counter "
                  "summary, spread uniformly over the dispatch's residency; capacity = 1024 SIMDs x clock / 4 cycles per wave64 instruction",
    }
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"r06_valu_timeline_{args.tag}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path)
    print(json.dumps({k: out[k] for k in ("window", "fast", "utilisation_quantiles", "per_kernel_in_window")}, indent=1))
    for s in slices[:int(step_ms * 1e3 / args.slice_us) + 2]:
        print(f"{s['t_ms']:7.3f} ms  util {s['util']:.2f}  " + " ".join(f"{k}:{v:.2f}" for k, v in s["share"].items()))


if __name__ == "__main__":
    main()
