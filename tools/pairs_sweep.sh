#!/bin/bash
# Batch-size curve of the bench line: stereo pairs per step in {8, 32, 128, 512} (resident and streamed values).
#   tools/pairs_sweep.sh <tag>      -> gpurun_out/r03_pairs_sweep_<tag>.json (copy to profiles/)
TAG=${1:-x}; mkdir -p gpurun_out
echo "[" > gpurun_out/r03_pairs_sweep_$TAG.json
first=1
for P in 8 32 128 512; do
  timeout 600 python bench.py --pairs $P --steps $((P >= 128 ? 50 : 200)) --warmup 5 --no-cpu-baseline > gpurun_out/sweep_$P.json 2> gpurun_out/sweep_$P.err || { echo "pairs $P failed"; tail -3 gpurun_out/sweep_$P.err; continue; }
  [ $first = 1 ] || echo "," >> gpurun_out/r03_pairs_sweep_$TAG.json; first=0
  python - >> gpurun_out/r03_pairs_sweep_$TAG.json <<PY
import json
d = json.load(open("gpurun_out/sweep_$P.json"))
print(json.dumps({"pairs_per_step": $P, "value": d["value"], "ms_per_step": d["ms_per_step"], "streamed_value": d["streamed"]["value"],
                  "streamed_ms_per_step": d["streamed"]["ms_per_step"], "h2d_GBps": d["streamed"]["h2d_GBps"],
                  "cadence6_value": (d.get("full_solve_cadence6") or {}).get("value"), "steps": d["steps"]}))
PY
done
echo "]" >> gpurun_out/r03_pairs_sweep_$TAG.json
cat gpurun_out/r03_pairs_sweep_$TAG.json
