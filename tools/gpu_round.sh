#!/bin/bash
# One GPU session: [tests] + counter summary (tools/pmc_collect.py) + bench + rocprofv3 kernel stats of the same command + B=1 latency.
#   tools/gpu_round.sh <tag> [tests|notests] [extra pytest args...]
# Everything lands in gpurun_out/; copy what should be judged into profiles/ (r<NN>_*).
TAG=${1:-x}; MODE=${2:-tests}; shift 2
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
if [ "$MODE" = "tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_$TAG.log
  tail -5 gpurun_out/pytest_$TAG.log
fi
# machine peaks first: bench.py normalises its roofline objects against the newest profiles/r*_peaks.json
timeout 600 python tools/peaks.py --round 3 > gpurun_out/peaks_$TAG.log 2>&1; echo "peaks rc=$?"; head -1 gpurun_out/peaks_$TAG.log
[ -f gpurun_out/r03_peaks.json ] && cp gpurun_out/r03_peaks.json profiles/r03_peaks.json
# counters next: bench.py reads the newest profiles/r*_pmc_*.json, so the bench line of this session is priced with this build's counts
timeout 1200 python tools/pmc_collect.py --round 3 --tag $TAG > gpurun_out/pmc_$TAG.log 2>&1; echo "pmc rc=$?"; tail -24 gpurun_out/pmc_$TAG.log
[ -f gpurun_out/r03_pmc_$TAG.json ] && cp gpurun_out/r03_pmc_$TAG.json profiles/r03_pmc_$TAG.json
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_$TAG.err
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_$TAG.json"))
    print("fps", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "profiled", d["profiled_pass"] and round(d["profiled_pass"]["ms_per_step"], 3),
          "cadence6", d["full_solve_cadence6"] and round(d["full_solve_cadence6"]["value"]),
          "streamed", d.get("streamed") and (round(d["streamed"]["value"]), round(d["streamed"]["h2d_GBps"], 1)))
    print({k: round(v, 3) for k, v in (d["profiled_pass"] or {}).get("kernel_ms_per_step", {}).items()})
    print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "valu", d["roofline_valu"]["frac"], d["roofline_valu"]["source"], "mfma", d["roofline_mfma"] and d["roofline_mfma"]["frac"], "cpu", d["cpu_baseline"] and (round(d["cpu_baseline"]["value"], 1), round(d["cpu_baseline"]["value_1thread"], 2)))
except Exception as e:
    print("bench json unreadable:", e)
PY
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o orb -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err )
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$TAG.csv && head -24 "$f"
rm -rf gpurun_out/prof_$TAG
timeout 600 python bench.py --workload latency > gpurun_out/latency_$TAG.json 2> gpurun_out/latency_$TAG.err; echo "latency rc=$?"; tail -2 gpurun_out/latency_$TAG.err
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/latency_$TAG.json"))
    for k, v in d["latency_b1"].items(): print(f"{k:50s} gpu {v['gpu_ms']:8.3f} ms   oracle {v.get('oracle_1thread_ms', float('nan')):9.3f} ms")
except Exception as e:
    print("latency json unreadable:", e)
PY
