#!/bin/bash
# the same block trace with ONE extractor handle on one stream and no side chain: what a FAST block's phases take when nothing shares its CU
TAG=${1:-x}
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so; cp tools/build/ab/libbt.so $P/libmyslam_hip.so
timeout 600 python bench.py --no-cpu-baseline --no-extra-passes --parity-frames 0 --steps 10 --workload orb_match --streams 1 --block-trace gpurun_out/bt_alone_$TAG.npy > gpurun_out/bt_alone_bench_$TAG.json 2> gpurun_out/bt_alone_bench_$TAG.err
echo "bench rc=$?"; tail -1 gpurun_out/bt_alone_bench_$TAG.err
cp /tmp/orig_lib.so $P/libmyslam_hip.so
python tools/block_trace_phases.py gpurun_out/bt_alone_$TAG.npy; python - <<PY
import numpy as np, json
r = np.load("gpurun_out/bt_alone_$TAG.npy")
w = r[:, 1]; kid = ((w >> np.uint64(24)) & np.uint64(0xf)).astype(int); dt = (w & np.uint64(0xffffff)).astype(float) / 100.0
mk = r[:, 2]; f = kid == 0
m = [((mk >> np.uint64(16 * i)) & np.uint64(0xffff)).astype(float)[f] / 100.0 for i in range(4)]
life = dt[f]; ok = (m[0] > 0) & (m[1] >= m[0]) & (m[2] >= m[1]) & (life >= m[2])
seg = {"decode_before_the_tile_loads": m[3][ok], "decode_and_stage_tile": m[0][ok], "score": (m[1] - m[0])[ok], "nms_and_record_list": (m[2] - m[1])[ok], "filter_and_append": (life - m[2])[ok], "whole_block": life[ok]}
out = {k: {"median_us": round(float(np.median(v)), 2), "mean_us": round(float(v.mean()), 2), "p90_us": round(float(np.percentile(v, 90)), 2)} for k, v in seg.items()}
out["blocks"] = int(ok.sum())
for k in (1, 2, 3, 4):
    if (kid == k).any(): out["kernel_%d_block_median_us" % k] = round(float(np.median(dt[kid == k])), 2)
json.dump(out, open("gpurun_out/r06_fast_block_phases_alone_$TAG.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
