#!/usr/bin/env python3
"""Re-collects the hardware counters behind bench.py's roofline objects for the CURRENT build and writes ONE summary,
gpurun_out/r<round>_pmc_<tag>.json (copy it to profiles/ and commit it: bench.py reads the newest profiles/r*_pmc_*.json).

    python tools/pmc_collect.py --round 2 --tag v21 [--pairs 64] [--workload full]

Three separate rocprofv3 passes of the same bench command (kernel trace + counters only, as MI355X_MICROARCH.md prescribes:
FETCH_SIZE and WRITE_SIZE do not fit into one pass): SQ instruction counts, FETCH_SIZE, WRITE_SIZE.  The bench runs joined on one
stream (--streams 1) so that every ORB launch covers all 2P images and nothing overlaps, and with MYSLAM_ORB_OPT_COPY_INPUT 1 so that the
calibration kernel below sees every image (by default the extractor reads level 0 in place and copies only the last image of a batch).

FETCH_SIZE calibration: on gfx950 the counter tallies 128-byte requests at 64 bytes.  k_ingest reads every byte of the 1241x376
input exactly once (466 616 B per image, known from the algorithm), so fetch_scale = known bytes / counted bytes of k_ingest is
measured in this very run and applied to every kernel's FETCH_SIZE; WRITE_SIZE is checked the same way against the bytes k_ingest
provably writes (16-byte stores covering ceil(1241 / 16) * 16 bytes per row).
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 376, 1241
PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU_MFMA_MOPS_I8"],
          ["FETCH_SIZE"], ["WRITE_SIZE"]]
STEPS_TOTAL = 6          # bench --warmup 2 --steps 2, preceded by the two untimed steps of the run's own checks (bench.py, build v68)
ORB_KERNELS = ("k_ingest", "k_resize", "k_fast", "k_octree", "k_blur7", "k_describe")      # unit = image; everything else: unit = stereo pair


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("myslam_hip::", "").strip()
    return n


def run_pass(counters, pairs, workload, outdir, scene_rects, extra=()):
    if os.path.isdir(outdir):
        shutil.rmtree(outdir)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", outdir, "-o", "a", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "2", "--pairs", str(pairs), "--workload", workload,
           "--streams", "1", "--orb-internal-stream", "0", "--orb-copy-input", "1", "--no-cpu-baseline", "--no-extra-passes", "--no-repeats", "--graph", "0", "--parity-frames", "0",
           "--scene-rects", str(scene_rects)] + list(extra)
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
        raise SystemExit(f"rocprofv3 pass {counters} failed (rc {r.returncode})")
    # per kernel: dispatch id -> {counter: value}; only the LAST step's dispatches are summed (the first launches of a run carry one-off
    # behaviour: e.g. the FAST kernel takes its two-phase path until its statistics say "dense")
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for row in csv.DictReader(open(files[0])):
        k = short(row["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        per[k][int(row.get("Dispatch_Id", row.get("Correlation_Id", "0")))][row["Counter_Name"]] += float(row["Counter_Value"])
    shutil.rmtree(outdir, ignore_errors=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = {}
    for k, d in per.items():
        ids = sorted(d)
        per_step = max(1, len(ids) // STEPS_TOTAL)
        for i in ids[-per_step:]:
            for c, v in d[i].items():
                agg[k][c] += v
        nd[k] = per_step
    return agg, nd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, default=2)
    ap.add_argument("--tag", required=True)
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--workload", default="full")
    ap.add_argument("--scene-rects", type=int, default=6000)
    ap.add_argument("--bench-arg", action="append", default=[], help="extra argument passed to bench.py (repeatable), e.g. --bench-arg=--fast-mode --bench-arg=1")
    ap.add_argument("--sq-only", action="store_true", help="instruction counters only (one pass, no traffic calibration)")
    args = ap.parse_args()
    P, steps_total = args.pairs, 1          # run_pass() keeps the last step only
    out_root = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_root, exist_ok=True)
    kernels = collections.defaultdict(dict)
    for counters in (PASSES[:1] if args.sq_only else PASSES):
        try:
            agg, nd = run_pass(counters, P, args.workload, os.path.join(out_root, "pmc_tmp"), args.scene_rects, args.bench_arg)
        except SystemExit:
            if len(counters) > 5:          # an optional counter this rocprofv3 does not know: retry with the basic SQ set
                agg, nd = run_pass(counters[:5], P, args.workload, os.path.join(out_root, "pmc_tmp"), args.scene_rects, args.bench_arg)
            else:
                raise
        for k, v in agg.items():
            kernels[k]["dispatches"] = nd[k]
            for c, x in v.items():
                kernels[k][c + "_total"] = x
    # per-unit figures
    for k, rec in kernels.items():
        unit = "image" if k.startswith(ORB_KERNELS) else "pair"
        nunits = steps_total * (2 * P if unit == "image" else P)
        rec["unit"] = unit
        rec["launches_per_step"] = rec["dispatches"]
        if "SQ_INSTS_VALU_total" in rec:
            rec["valu_wave_insts_per_" + unit] = rec["SQ_INSTS_VALU_total"] / nunits
        if "FETCH_SIZE_total" in rec:
            rec["fetch_bytes_per_" + unit + "_raw"] = rec["FETCH_SIZE_total"] * 1024.0 / nunits
        if "WRITE_SIZE_total" in rec:
            rec["write_bytes_per_" + unit] = rec["WRITE_SIZE_total"] * 1024.0 / nunits
    if args.sq_only:
        for k, rec in sorted(kernels.items()):
            u = rec["unit"]
            print(f"{k:28s} VALU/{u} {rec.get('valu_wave_insts_per_' + u, 0):12.0f}  SALU {rec.get('SQ_INSTS_SALU_total', 0) / (2 * P if u == 'image' else P):12.0f}  "
                  f"LDS {rec.get('SQ_INSTS_LDS_total', 0) / (2 * P if u == 'image' else P):10.0f}  busy cycles {rec.get('SQ_BUSY_CYCLES_total', 0):14.0f}")
        return
    ing = kernels.get("k_ingest")
    if not ing or "fetch_bytes_per_image_raw" not in ing:
        raise SystemExit("k_ingest missing from the FETCH_SIZE pass: cannot calibrate")
    known_r = float(H * W)
    known_w = float(((W + 15) // 16) * 16 * H)
    fetch_scale = known_r / ing["fetch_bytes_per_image_raw"]
    write_check = ing["write_bytes_per_image"] / known_w
    for k, rec in kernels.items():
        u = rec["unit"]
        if "fetch_bytes_per_" + u + "_raw" in rec:
            rec["fetch_bytes_per_" + u + "_corrected"] = rec["fetch_bytes_per_" + u + "_raw"] * fetch_scale
    pyr_px = 1444097
    cross = {}
    for k, rec in kernels.items():
        if k.startswith(("k_blur7", "k_resize", "k_fast")) and "fetch_bytes_per_image_corrected" in rec:
            cross[k] = {"corrected_fetch_over_pyramid_bytes": rec["fetch_bytes_per_image_corrected"] / pyr_px}
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    out = {
        "build": args.tag, "round": args.round, "build_id": load_package().api.build_id(),      # digest of the library's sources (myslam_hip_version)
        "command": f"rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 2 --warmup 2 --pairs {P} --workload {args.workload} --streams 1 "
                   f"--orb-internal-stream 0 --orb-copy-input 1 --no-cpu-baseline --no-extra-passes --scene-rects {args.scene_rects}   (one pass per counter set; counters of the LAST "
                   f"step; COPY_INPUT 1 so that the calibration kernel k_ingest sees every image — by default it copies only the last image of a batch)",
        "counter_sets": PASSES, "pairs_per_step": P, "steps_total": steps_total,
        "calibration": {"kernel": "k_ingest", "known_read_bytes_per_image": known_r, "counted_read_bytes_per_image": ing["fetch_bytes_per_image_raw"],
                        "fetch_scale": fetch_scale, "known_write_bytes_per_image": known_w, "counted_write_over_known": write_check,
                        "cross_check": cross,
                        "note": "FETCH_SIZE x fetch_scale = bytes fetched through the L2's memory side (Infinity-Cache hits included); WRITE_SIZE needs "
                                "no correction when counted_write_over_known is 1.00"},
        "kernels": {k: kernels[k] for k in sorted(kernels)},
    }
    path = os.path.join(out_root, f"r{args.round:02d}_pmc_{args.tag}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path)
    print(json.dumps(out["calibration"], indent=1))
    for k, rec in out["kernels"].items():
        u = rec["unit"]
        print(f"{k:28s} {u:5s} launches/step {rec['launches_per_step']:5.1f}  VALU/{u} {rec.get('valu_wave_insts_per_' + u, 0):12.0f}  "
              f"fetch/{u} {rec.get('fetch_bytes_per_' + u + '_corrected', 0):12.0f}  write/{u} {rec.get('write_bytes_per_' + u, 0):12.0f}")


if __name__ == "__main__":
    main()
