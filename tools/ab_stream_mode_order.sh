#!/bin/bash
# Are bench.py's live-stream points lower than the stand-alone command's because the children run beside the parent's idle GPU context (its hardware queues stay
# mapped), or because their timed region is short (400 steps)?  Same box: the sweep BEFORE the parent's context exists / after the other passes, 400 / 1600 steps.
for rep in 1 2; do
for mode in early late; do
for st in 0 1600; do
  fl=""; [ $mode = late ] && fl="--stream-mode-late"
  MYSLAM_SM_STEPS=$st python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --steps 20 $fl > gpurun_out/smo_${mode}_${st}_$rep.json 2> gpurun_out/smo_${mode}_${st}_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/smo_${mode}_${st}_$rep.json"))
    print("$mode", "steps", $st or "default", "rep", $rep, "step", round(d["ms_per_step"], 3), [(p["pairs_per_step"], p["lanes"], round(p["value"]), round(p["frame_latency_ms"]["loaded_median_ms"], 2)) for p in d["stream_mode"]["sweep"]])
except Exception as e:
    print("$mode $st failed", e)
PY
done; done; done
