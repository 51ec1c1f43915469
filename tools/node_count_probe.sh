#!/bin/bash
# What bounds the one-pair operating point under load: the number of graph nodes per step or the work in them?  The recorded 1-pair step on 16 lanes with parts
# of the side chain left out (--side-skip: ba = 1 node with 13 % of the chain's kernel time, db = 3 nodes with 7 %, lcd = 6 nodes with 21 %).
#   tools/node_count_probe.sh [pairs] [lanes]   -> gpurun_out/node_count_probe.json
PP=${1:-1}; LL=${2:-16}
export GPU_MAX_HW_QUEUES=24
echo "[" > gpurun_out/node_count_probe.json; first=1
for skip in "" ba db lcd,db lcd,db,ba; do
  for rep in 1 2; do
    timeout 300 python bench.py --pairs $PP --lanes $LL --graph 1 --steps 1600 --warmup 2 --no-extra-passes --no-cpu-baseline --parity-frames 0 --stream-mode "" --side-skip "$skip" > gpurun_out/ncp.json 2> gpurun_out/ncp.err || { echo "skip=$skip failed"; tail -3 gpurun_out/ncp.err; continue; }
    [ $first = 1 ] || echo "," >> gpurun_out/node_count_probe.json; first=0
    python - "$skip" >> gpurun_out/node_count_probe.json <<'PY'
import json, sys
d = json.load(open("gpurun_out/ncp.json"))
print(json.dumps({"side_skip": sys.argv[1], "graph_nodes": d.get("graph_nodes"), "ms_per_step": round(d["ms_per_step"], 5), "value": round(d["value"])}))
PY
  done
done
echo "]" >> gpurun_out/node_count_probe.json
cat gpurun_out/node_count_probe.json
