#!/usr/bin/env python3
"""Reads the block trace of a profiling build (bench.py --block-trace FILE with a library built with -DMYSLAM_BLOCK_TRACE) and answers, with MEASURED
block residencies on the 32 CUs of XCD 0, the question the counter-volume model of tools/valu_timeline.py can only approximate: where do the idle
vector-issue slots of a step sit?

    python tools/block_trace_report.py gpurun_out/bt.npy [--pmc profiles/r05_pmc_v60.json] [--slice-us 100] > profiles/r06_valu_timeline.json

Per record: start (100 MHz clock), duration, kernel, CU (se / sh / cu of HW_ID), block id.  Per slice of one steady-state step and per kernel:
  started        blocks (describe: work items) that began in the slice
  resident       mean number of blocks resident PER CU (sum of overlaps / slice / CUs seen)
  valu_share     the kernel's instructions issued in the slice, as a fraction of the slice's issue capacity on those CUs — every block's instruction
                 count (counter summary: SQ_INSTS_VALU per image / blocks per image) spread over ITS OWN measured residency
and `util` = the sum.  Plus per kernel the block lifetime distribution inside / outside the phases in which another kernel is resident."""
import argparse
import collections
import json
import os
import sys

import numpy as np

KID = {0: "k_fast_strip", 1: "k_resize_strip", 2: "k_octree", 3: "k_blur7_strip", 4: "k_describe2"}
PMC_KEY = {0: "k_fast_strip", 1: "k_resize_strip", 2: "k_octree", 3: "k_blur7_strip", 4: "k_describe2"}
CLOCK_GHZ = 2.4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("npy")
    ap.add_argument("--pmc", default="")
    ap.add_argument("--slice-us", type=float, default=100.0)
    ap.add_argument("--images-per-launch", type=int, default=512)
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = np.load(args.npy)
    t0 = r[:, 0].astype(np.int64)                       # 10 ns ticks
    w = r[:, 1]
    dt = (w & np.uint64(0xffffff)).astype(np.int64)
    kid = ((w >> np.uint64(24)) & np.uint64(0xf)).astype(np.int64)
    cu = ((w >> np.uint64(32)) & np.uint64(0xff)).astype(np.int64)
    order = np.argsort(t0)
    t0, dt, kid, cu = t0[order], dt[order], kid[order], cu[order]
    ncu = len(np.unique(cu))
    base = t0[0]
    t0 = (t0 - base) * 10                               # ns
    dt = dt * 10
    # instruction volume per block: counter summary per image / blocks of one image (measured here: records of a launch x 8 XCDs / images)
    pmc_path = args.pmc
    if not pmc_path:
        import glob, re
        cands = [f for f in glob.glob(os.path.join(root, "profiles", "r*_pmc_*.json")) if re.search(r"r(\d+)_pmc_v\d+\.json$", f)]
        pmc_path = max(cands, key=lambda f: tuple(int(x) for x in re.search(r"r(\d+)_pmc_v(\d+)\.json$", f).groups()))
    pmc = json.load(open(pmc_path))["kernels"]
    per_img = {}
    for k, name in PMC_KEY.items():
        rec = next((v for kk, v in pmc.items() if kk.startswith(name)), None)
        per_img[k] = rec["valu_wave_insts_per_image"] if rec else 0.0
    # FAST launches: within a launch the block ids of the started blocks grow from 0 (per XCD in steps of 8); a new launch begins where they fall back
    bid = ((w[order] >> np.uint64(40)) & np.uint64(0xffffff)).astype(np.int64)
    fi = np.where(kid == 0)[0]
    launches = []
    cur_start, hi = None, 0
    for i in fi:
        if cur_start is None:
            cur_start, hi, last = int(t0[i]), int(bid[i]), int(t0[i])
            continue
        if bid[i] < hi // 4 and hi > 5000:
            launches.append((cur_start, last)); cur_start, hi = int(t0[i]), int(bid[i])
        hi = max(hi, int(bid[i])); last = int(t0[i])
    launches.append((cur_start, last))
    launches = [(a, b) for a, b in launches if b - a > 300000]
    if len(launches) < 5:
        raise SystemExit(f"only {len(launches)} FAST launches in the trace")
    # one steady-state step = FAST launches 2 and 3 (two handles): from the start of launch 2 to the start of launch 4
    wa, wb = launches[2][0], (launches[4][0] if len(launches) > 4 else launches[3][1])
    step_ms = (wb - wa) / 1e6
    blocks_per_img = {}
    for k in KID:
        n_in = int(((kid == k) & (t0 >= wa) & (t0 < wb)).sum())
        blocks_per_img[k] = n_in * (256.0 / ncu) / (2 * args.images_per_launch) if n_in else 0.0           # x (256 / traced CUs), two launches of images_per_launch images per step
    inst_per_block = {k: (per_img[k] / blocks_per_img[k] if blocks_per_img[k] else 0.0) for k in KID}
    sl = args.slice_us * 1e3
    nsl = int(np.ceil((wb - wa) / sl))
    cap = ncu * 4 * CLOCK_GHZ * sl / 4.0                 # wave-instructions the traced CUs can issue per slice (4 SIMDs, 4 cycles each)
    slices = []
    sel = (t0 + dt > wa) & (t0 < wb)
    T0, DT, K = t0[sel], dt[sel], kid[sel]
    for s in range(nsl):
        a, b = wa + s * sl, min(wb, wa + (s + 1) * sl)
        ov = np.minimum(T0 + DT, b) - np.maximum(T0, a)
        live = ov > 0
        row = {"t_ms": round((a - wa) / 1e6, 3), "util": 0.0, "kernels": {}}
        for k, name in KID.items():
            m = live & (K == k)
            if not m.any():
                continue
            res = float(ov[m].sum()) / (b - a) / ncu
            insts = float((ov[m] / np.maximum(DT[m], 1)).sum()) * inst_per_block[k]
            started = int(((K == k) & (T0 >= a) & (T0 < b)).sum())
            share = insts / (cap * (b - a) / sl)
            row["kernels"][name] = {"started": started, "resident_per_cu": round(res, 2), "valu_share": round(share, 3)}
            row["util"] += share
        row["util"] = round(row["util"], 3)
        slices.append(row)
    # FAST block lifetimes by co-runner: classify every FAST block by the kernel (other than FAST) with the largest residency during its life — coarse: by slice
    dom = []
    for row in slices:
        others = {k: v["resident_per_cu"] for k, v in row["kernels"].items() if k != "k_fast_strip"}
        dom.append(max(others, key=others.get) if others and max(others.values()) > 0.05 else "alone")
    life = collections.defaultdict(list)
    fsel = (K == 0) & (T0 >= wa) & (T0 < wb)
    for a_, d_ in zip(T0[fsel], DT[fsel]):
        life[dom[min(nsl - 1, int((a_ - wa) // sl))]].append(d_ / 1e3)
    fast_life = {k: {"blocks": len(v), "median_us": round(float(np.median(v)), 2), "p90_us": round(float(np.percentile(v, 90)), 2)} for k, v in life.items()}
    # FAST progress rate by co-runner (blocks started per ms on the traced CUs) and its resident blocks per CU
    rate = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for row, d in zip(slices, dom):
        f = row["kernels"].get("k_fast_strip")
        e = rate[d]
        e[0] += 1
        if f:
            e[1] += f["started"]; e[2] += f["resident_per_cu"]
        e[3] += row["util"]
    fast_rate = {k: {"slices": v[0], "fast_blocks_started_per_ms": round(v[1] / (v[0] * args.slice_us / 1e3), 1), "fast_resident_per_cu": round(v[2] / v[0], 2),
                     "issue_util": round(v[3] / v[0], 3)} for k, v in rate.items()}
    # where a FAST block's life goes (thread 0's clock at the barriers that end its phases): tile staged | scored | NMS + record list | filter + append
    phases = None
    if r.shape[1] >= 3:
        mk = r[order][:, 2]
        fm = (kid == 0) & (t0 >= wa) & (t0 < wb)
        m0 = ((mk >> np.uint64(0)) & np.uint64(0xffff)).astype(np.float64)[fm] / 100.0
        m1 = ((mk >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64)[fm] / 100.0
        m2 = ((mk >> np.uint64(32)) & np.uint64(0xffff)).astype(np.float64)[fm] / 100.0
        life_us = dt[fm] / 1e3
        ok = (m0 > 0) & (m1 >= m0) & (m2 >= m1) & (life_us >= m2)
        m3 = ((mk >> np.uint64(48)) & np.uint64(0xffff)).astype(np.float64)[fm] / 100.0
        seg = {"decode_before_the_tile_loads": m3[ok], "decode_and_stage_tile": m0[ok], "score": (m1 - m0)[ok], "nms_and_record_list": (m2 - m1)[ok], "filter_and_append": (life_us - m2)[ok], "whole_block": life_us[ok]}
        phases = {k: {"median_us": round(float(np.median(v)), 2), "mean_us": round(float(v.mean()), 2), "p90_us": round(float(np.percentile(v, 90)), 2)} for k, v in seg.items()}
        phases["blocks"] = int(ok.sum())
    util = np.array([row["util"] for row in slices])
    out = {
        "source": os.path.basename(args.npy), "counter_source": os.path.relpath(pmc_path, root), "cus_traced": ncu, "xcd": 0, "slice_us": args.slice_us,
        "clock_ghz_assumed": CLOCK_GHZ, "step_ms": round(step_ms, 3), "records": int(len(r)),
        "blocks_per_image": {KID[k]: round(v, 2) for k, v in blocks_per_img.items()},
        "valu_wave_insts_per_block": {KID[k]: round(v, 1) for k, v in inst_per_block.items()},
        "step_issue_util_orb_kernels": round(float(util.mean()), 3),
        "fast_block_phases_us": phases,
        "fast_block_lifetime_by_co_runner": fast_life,
        "fast_rate_by_co_runner": fast_rate,
        "slices": slices,
        "method": "every block of the five ORB kernels that ran on XCD 0 records its start and duration (s_memrealtime) and CU; a block's instruction volume "
                  "(SQ_INSTS_VALU per image of the counter summary / measured blocks per image) is spread over its own residency; capacity = CUs traced x 4 SIMDs "
                  "x clock / 4 cycles.  The side chain's kernels (DeepLCD, DB scan, matcher, BA) are not traced: `util` is the ORB kernels' share only",
    }
    json.dump(out, sys.stdout, indent=1)
    sys.stderr.write(json.dumps({k: out[k] for k in ("step_ms", "cus_traced", "blocks_per_image", "valu_wave_insts_per_block", "step_issue_util_orb_kernels", "fast_block_phases_us",
                                                     "fast_block_lifetime_by_co_runner", "fast_rate_by_co_runner")}, indent=1) + "\n")
    for row, d in zip(slices, dom):
        sys.stderr.write(f"{row['t_ms']:6.2f} util {row['util']:.2f} [{d:14s}] " +
                         " ".join(f"{k[2:8]}:{v['resident_per_cu']:.1f}/{v['valu_share']:.2f}" for k, v in row["kernels"].items()) + "\n")


if __name__ == "__main__":
    main()
