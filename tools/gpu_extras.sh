#!/bin/bash
# Secondary bench lines of a round: the sparse stream (the two-phase FAST path's regime), the other workloads, the joined one-stream
# run whose profiled pass gives clean per-kernel times.   tools/gpu_extras.sh <tag>
TAG=${1:-x}; mkdir -p gpurun_out
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench_${name}_$TAG.json 2> gpurun_out/bench_${name}_$TAG.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_${name}_$TAG.json"))
    print("$name: fps", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in ((d.get("profiled_pass") or {}).get("kernel_ms_per_step") or {}).items()})
except Exception as e:
    print("$name failed:", e)
PY
}
run sparse300 --scene-rects 300
run sparse1000 --scene-rects 1000
run sparse300_dense --scene-rects 300 --fast-mode 1
run orb_match --workload orb_match
run orb_match_lcd --workload orb_match_lcd
run full_solve --workload full_solve
run 1stream --streams 1 --orb-internal-stream 0
run joined --pipeline 0
