# same-box A/B of the cadence passes: per-landmark solve state in LDS (133 KB per window) against the HBM scratch form (81 KB)
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_fallbacks.py -x -q 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
  timeout 600 python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --stream-mode "" --solve-lm-hbm $v > gpurun_out/ab_solve_${v}_$rep.json 2> gpurun_out/ab_solve_${v}_$rep.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_solve_${v}_$rep.json"))
print("lm_hbm=$v rep=$rep step", round(d["ms_per_step"], 3), "cad6", round(d["full_solve_cadence6"]["ms_per_step"], 3), "every", round(d["full_solve_every_frame"]["ms_per_step"], 3), "solve_all", round(d["ba_solve_all_windows_ms"], 3), d["roofline_ba_optimize"]["frac"])
PY
done; done
