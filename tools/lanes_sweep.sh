#!/bin/bash
# Live-stream operating points against the number of lanes: frames/s and the loaded frame latency (median / p90) of recorded one-pair steps, same box.
#   tools/lanes_sweep.sh <tag> [pairs=1]   -> gpurun_out/r06_lanes_sweep_<tag>.json
TAG=${1:-x}; PAIRS=${2:-1}
for L in 2 4 6 8 12 16; do
  GPU_MAX_HW_QUEUES=24 timeout 300 python bench.py --pairs $PAIRS --lanes $L --graph 1 --steps 3200 --warmup 2 --no-extra-passes --no-cpu-baseline --parity-frames 0 --frame-latency --stream-mode "" > gpurun_out/lanes_${TAG}_$L.json 2> gpurun_out/lanes_${TAG}_$L.err; echo "L=$L rc=$?"
done
python - <<PY
import json
out = {"build": "$TAG", "pairs_per_step": $PAIRS, "points": []}
for L in (2, 4, 6, 8, 12, 16):
    try:
        d = json.load(open(f"gpurun_out/lanes_${TAG}_{L}.json"))
    except Exception as e:
        print(L, "failed", e); continue
    fl = d["frame_latency"]
    p = {"lanes": L, "frames_per_s": d["value"], "repeats_ms_per_step": d["repeats_ms_per_step"], "loaded_median_ms": fl["loaded_median_ms"], "loaded_p90_ms": fl["loaded_p90_ms"],
         "one_lane_alone_ms": fl["one_lane_alone_ms"], "little_law_ms": L * $PAIRS / d["value"] * 1e3, "graph_nodes": d["graph_nodes"]}
    out["points"].append(p); print(p)
out["note"] = ("recorded one-stream steps replayed on L lanes (step k on lane k mod L), all lanes kept busy; little_law_ms = lanes x pairs / throughput: the latency a "
               "closed system of L requests in flight must show at that throughput, whatever the node-level cause")
json.dump(out, open("gpurun_out/r06_lanes_sweep_$TAG.json", "w"), indent=1)
PY
