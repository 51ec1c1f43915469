#!/usr/bin/env python3
"""phase marks of the traced kernels' blocks (trace build, bench.py --block-trace): python tools/block_trace_phases.py bt.npy  -> per kernel the medians of the marked phases"""
import json, sys
import numpy as np
r = np.load(sys.argv[1])
w = r[:, 1]; kid = ((w >> np.uint64(24)) & np.uint64(0xf)).astype(int); dt = (w & np.uint64(0xffffff)).astype(float) / 100.0
mk = r[:, 2]
names = {0: ("k_fast_strip", ["decode_and_stage_tile", "score", "nms_and_record_list", "filter_and_append"], [0, 1, 2]),
         4: ("k_describe2 (per 64-key-point item)", ["slots_and_patch_addresses", "A_moments_mfma", "B_angle", "C_brief_and_store"], [0, 1, 2])}
out = {}
for k, (name, labels, marks) in names.items():
    f = kid == k
    if not f.any():
        continue
    m = [((mk >> np.uint64(16 * i)) & np.uint64(0xffff)).astype(float)[f] / 100.0 for i in marks]
    life = dt[f]
    ok = (m[0] > 0) & (m[1] >= m[0]) & (m[2] >= m[1]) & (life >= m[2])
    segs = [m[0], m[1] - m[0], m[2] - m[1], life - m[2]]
    out[name] = {lab: round(float(np.median(v[ok])), 2) for lab, v in zip(labels, segs)}
    out[name]["whole_us_median"] = round(float(np.median(life[ok])), 2); out[name]["blocks"] = int(ok.sum())
for k in (1, 2, 3):
    if (kid == k).any():
        out[{1: "k_resize_strip", 2: "k_octree", 3: "k_blur7_strip"}[k]] = {"whole_us_median": round(float(np.median(dt[kid == k])), 2), "blocks": int((kid == k).sum())}
print(json.dumps(out, indent=1))
