#!/bin/bash
# PMC pass (own run, kernel-trace only): SQ instruction mix and wait breakdown per kernel
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${1:-pmc1}
cd /tmp
rocprofv3 --kernel-trace --pmc ${PMC:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS} --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$R -o a -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --pairs 64 --workload ${WL:-orb_match} --streams 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/$R.err
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/$R/**/*counter_collection.csv", recursive=True)
print(f)
first = None
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    if first is None: first = row["Counter_Name"]
    k = row["Kernel_Name"].split("(")[0][-40:]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == first: n[k] += 1
for k, v in agg.items():
    d = n[k] or 1
    if "hip::" not in k: continue
    print(k, "launches", d, {c: round(x / d) for c, x in v.items()})
PY
rm -rf gpurun_out/$R
