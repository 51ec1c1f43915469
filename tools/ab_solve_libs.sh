#!/bin/bash
# same-box A/B of the cadence passes (the local-BA solve beside the step) over library builds: tools/ab_solve_libs.sh <dir with lib*.so>
D=$1; shift
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_fallbacks.py -x -q 2>&1 | tail -3
for rep in 1 2; do for f in $D/lib*.so; do
  n=$(basename $f .so); cp $f $P/libmyslam_hip.so
  timeout 600 python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --stream-mode "" "$@" > gpurun_out/abs_${n}_$rep.json 2> gpurun_out/abs_${n}_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/abs_${n}_$rep.json"))
    print("$n rep=$rep step", round(d["ms_per_step"], 3), "cad6", round(d["full_solve_cadence6"]["ms_per_step"], 3), "every", round(d["full_solve_every_frame"]["ms_per_step"], 3), "solve_all", round(d["ba_solve_all_windows_ms"], 3), d["roofline_ba_optimize"]["frac"])
except Exception as e:
    print("$n failed", e)
PY
done; done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
