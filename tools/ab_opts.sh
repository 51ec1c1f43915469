#!/bin/bash
# Same-box A/B of bench OPTIONS (not builds): tools/ab_opts.sh <tag> "<args A>" "<args B>" ...   (two rounds, alternating)
TAG=$1; shift
for rep in 1 2; do
i=0
for a in "$@"; do
  i=$((i+1))
  python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --stream-mode "" --steps 60 $a > gpurun_out/abo_${TAG}_${i}_$rep.json 2> gpurun_out/abo_${TAG}_${i}_$rep.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/abo_${TAG}_${i}_$rep.json"))
    print("[$a]", $rep, "step", [round(v, 3) for v in d["repeats_ms_per_step"]], "cad6", round(d["full_solve_cadence6"]["ms_per_step"], 3), "alone", {k[2:]: round(v, 3) for k, v in d["extractor_alone"]["kernel_ms_per_call"].items()},
          "piped", {k[2:]: round(v, 3) for k, v in d["profiled_pass"]["kernel_ms_per_step"].items() if k in ("k_fast_strip", "k_describe2", "k_conv2_f16x3", "k_db_scan_bf16x6")})
except Exception as e:
    print("[$a] failed", e)
PY
done; done
