#!/usr/bin/env python3
"""k_octree's blocks in a block trace (trace build with the oct-tree's phase marks): per level the medians of candidates, the two candidate passes, the node-list
simulation, the best-key pass, the order sort, and the whole block.   python tools/block_trace_octree.py bt.npy"""
import json, sys
import numpy as np
r = np.load(sys.argv[1])
w = r[:, 1]; kid = ((w >> np.uint64(24)) & np.uint64(0xf)).astype(int); dt = (w & np.uint64(0xffffff)).astype(float) / 100.0
f = kid == 2
mk = r[f, 2]; aux = r[f, 3]; life = dt[f]
m = [((mk >> np.uint64(16 * i)) & np.uint64(0xffff)).astype(float) / 100.0 for i in range(3)]
lvl = (aux & np.uint64(0xff)).astype(int); n = ((aux >> np.uint64(8)) & np.uint64(0xffffff)).astype(int)
out = {}
for l in sorted(set(lvl)):
    g = (lvl == l) & (m[0] > 0)
    if not g.any(): continue
    med = lambda v: round(float(np.median(v[g])), 1)
    out[int(l)] = {"blocks": int(g.sum()), "candidates": int(np.median(n[g])), "A_B_candidate_passes_us": med(m[0]), "C_node_list_us": med(m[1] - m[0]), "D_best_key_us": med(m[2] - m[1]),
                   "E_order_us": med(life - m[2]), "whole_us": med(life), "whole_p90_us": round(float(np.percentile(life[g], 90)), 1)}
print(json.dumps(out, indent=1))
