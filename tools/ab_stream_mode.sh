#!/bin/bash
# A/B of library builds at the live-stream operating point: tools/ab_stream_mode.sh <dir with lib*.so> [pairs] [lanes]
# (the recorded step on L lanes, two alternating runs per build; GPU_MAX_HW_QUEUES=24 as bench.py's stream_mode children)
D=$1; PP=${2:-1}; LL=${3:-16}
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
export GPU_MAX_HW_QUEUES=24
cp $P/libmyslam_hip.so /tmp/orig_lib.so
for rep in 1 2; do
for f in $D/lib*.so; do
  n=$(basename $f .so)
  cp $f $P/libmyslam_hip.so
  timeout 300 python bench.py --pairs $PP --lanes $LL --graph 1 --steps 1600 --warmup 2 --no-extra-passes --no-cpu-baseline --parity-frames 1 --stream-mode "" > gpurun_out/absm_${n}_$rep.json 2> gpurun_out/absm_${n}_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/absm_${n}_$rep.json"))
    print("$n", $rep, "nodes", d.get("graph_nodes"), "us/step", round(d["ms_per_step"] * 1000, 1), "frames/s", round(d["value"]), "parity", (d.get("parity_sample") or {}).get("ok"))
except Exception as e:
    print("$n failed", e)
PY
done; done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
