#!/bin/bash
# Same-box A/B of library builds by what the extractor's kernels take ALONE (pass 6 of bench.py) and by the pipelined step:
#   tools/ab_alone.sh <tag> name1 name2 ...      (tools/build/ab/lib<name>.so; "base" = the in-tree library)
TAG=$1; shift
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = base ]; then cp /tmp/orig_lib.so $P/libmyslam_hip.so; else cp tools/build/ab/lib$n.so $P/libmyslam_hip.so; fi
  python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --stream-mode "" --steps 60 > gpurun_out/aba_${TAG}_${n}_$rep.json 2> gpurun_out/aba_${TAG}_${n}_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/aba_${TAG}_${n}_$rep.json"))
    print("$n", $rep, "step", round(d["ms_per_step"], 3), "cad6", round(d["full_solve_cadence6"]["ms_per_step"], 3), "alone", {k[2:]: round(v, 3) for k, v in d["extractor_alone"]["kernel_ms_per_call"].items()},
          "piped", {k[2:]: round(v, 3) for k, v in d["profiled_pass"]["kernel_ms_per_step"].items() if k in ("k_fast_strip", "k_describe2", "k_conv2_f16x3")})
except Exception as e:
    print("$n failed", e)
PY
done; done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
