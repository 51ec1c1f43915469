#!/bin/bash
# The driver's command (--steps 20 --warmup 5) with the warm-up steps straight before the timed region (build v68) / with the run's checks and the clock probe between
# them (up to v67).  Same box, alternating; every line: the timed region and the two identical regions after it, the clock readings around them.
for rep in 1 2 3 4; do
for mode in new gap; do
  fl=""; [ $mode = gap ] && fl="--gap-before-timed"
  python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --no-extra-passes --steps 20 --warmup 5 $fl > gpurun_out/wg_${mode}_$rep.json 2> gpurun_out/wg_${mode}_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/wg_${mode}_$rep.json"))
    print("$mode", $rep, [round(x, 3) for x in d["repeats_ms_per_step"]], [round(x) for x in d["clock_mhz"]])
except Exception as e:
    print("$mode failed", e)
PY
done; done
