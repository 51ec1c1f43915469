#!/bin/bash
# bench + rocprofv3 kernel trace of the same command; summaries are copied to profiles/ by hand
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${1:-r1}
P=${2:-256}
timeout 1200 python bench.py --steps 10 --warmup 2 --pairs $P > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
tail -2 gpurun_out/bench_$R.err
cat gpurun_out/bench_$R.json
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$R -o orb -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --pairs $P --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$R.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_$R.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$R -name "*stats*" | head; 
f=$(find gpurun_out/prof_$R -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f"
rm -f $(find gpurun_out/prof_$R -name "*kernel_trace.csv")
