mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "0 300" "1 300" "0 6000" "1 6000"; do set -- $cfg
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pm -o a -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --pairs 64 --workload orb_match --streams 1 --orb-internal-stream 0 --no-cpu-baseline --no-extra-passes --fast-mode $1 --scene-rects $2 > /dev/null 2>&1 )
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pm/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(float); n = 0
for row in csv.DictReader(open(f)):
    if "k_fast_strip" in row["Kernel_Name"]:
        agg[row["Counter_Name"]] += float(row["Counter_Value"])
print("mode $1 rects $2", {k: round(v / 3 / 128) for k, v in agg.items()})
PY
rm -rf gpurun_out/pm
done
