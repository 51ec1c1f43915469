#!/bin/bash
# rocprofv3 kernel stats of the pose-graph optimisation at PGO_SIZES (default 1500:6) -> gpurun_out/prof_pgo/pgo_kernel_stats.csv
REPO=$PWD
export PGO_SIZES=${PGO_SIZES:-1500:6}
mkdir -p gpurun_out/prof_pgo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pgo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pgo -o pgo -- python $REPO/tools/pgo_time.py > $REPO/gpurun_out/prof_pgo/run.log 2>&1
cat $REPO/gpurun_out/prof_pgo/run.log | grep "n=" 
f=$(find /tmp/prof_pgo -name "*kernel_stats.csv" | head -1)
cp $f $REPO/gpurun_out/prof_pgo/pgo_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:14]:
    print("%-60s calls %6s  avg %10.1f us  total %9.2f ms  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
