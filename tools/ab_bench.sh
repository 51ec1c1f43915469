#!/bin/bash
# A/B timing of library builds on ONE box: tools/ab_bench.sh <dir with lib*.so> [bench args...]
# Every lib*.so in the directory is copied over the in-tree library in turn and benched with the same arguments.
D=$1; shift
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
for rep in 1 2; do
for f in $D/lib*.so; do
  n=$(basename $f .so)
  cp $f $P/libmyslam_hip.so
  python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_${n}_$rep.json 2> gpurun_out/ab_${n}_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/ab_${n}_$rep.json"))
    print("$n", $rep, round(d["value"]), round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in ((d.get("profiled_pass") or {}).get("kernel_ms_per_step") or {}).items() if k in ("k_fast_strip", "k_octree", "k_describe2", "k_resize_strip", "k_blur7_strip", "k_conv1_f16x3_pool_lrn", "k_conv2_f16x3", "k_ba_build", "k_triangulate", "k_hamming_fp4")})
except Exception as e:
    print("$n failed", e)
PY
done; done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
