TAG=${TAG:-v43}
[ -z "$SKIP_PYTEST" ] && { timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$TAG.log; }
timeout 900 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_$TAG.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
print("fps", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "host", round(d["host_launch_ms_per_step"],3), "streamed", d.get("streamed") and round(d["streamed"]["value"]), "cad6", d["full_solve_cadence6"] and round(d["full_solve_cadence6"]["value"]), "every", d["full_solve_every_frame"] and round(d["full_solve_every_frame"]["value"]))
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), "alone", d["roofline"].get("alone",{}).get("frac"), "valu", d["roofline_valu"] and (round(d["roofline_valu"]["frac"],3), d["roofline_valu"]["frac_alone"]), "mfma", d["roofline_mfma"] and round(d["roofline_mfma"]["frac"],4))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["value_1thread"], d["cpu_baseline"]["cores"], "parity", d["parity_sample"]["ok"], "lanes", d["step_graph"] and round(d["step_graph"]["value"]))
PY
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o orb -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --parity-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.err )
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$TAG.csv && head -14 "$f"
rm -rf gpurun_out/prof_$TAG
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1_$TAG -o orb -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --parity-frames 0 --streams 1 --no-extra-passes --graph 0 > $GRAFT_REPO_ROOT/gpurun_out/prof1_$TAG.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof1_$TAG.err )
f=$(find gpurun_out/prof1_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_1stream_$TAG.csv && head -8 "$f"
rm -rf gpurun_out/prof1_$TAG
timeout 600 python bench.py --workload latency > gpurun_out/latency_$TAG.json 2> gpurun_out/latency_$TAG.err; echo "latency rc=$?"
# the timed region alone under rocprofv3 (warm-up + K pipelined steps, no other pass): its per-kernel averages are the per-launch durations the
# bench line's roofline objects quote from the HIP events of pass 2
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/proft_$TAG -o orb -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --parity-frames 0 --no-extra-passes --graph 0 --stream-input 0 > $GRAFT_REPO_ROOT/gpurun_out/proft_$TAG.json 2> $GRAFT_REPO_ROOT/gpurun_out/proft_$TAG.err )
f=$(find gpurun_out/proft_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_timed_$TAG.csv && head -4 "$f" | cut -c1-40,180-320
rm -rf gpurun_out/proft_$TAG
