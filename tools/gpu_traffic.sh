#!/bin/bash
# HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in separate PMC passes (MI355X_MICROARCH.md §HBM), kernel-trace only
mkdir -p gpurun_out; export TMPDIR=/tmp
P=${1:-256}
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr_$C -o a -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --pairs $P --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/tr_$C.err
  cd $GRAFT_REPO_ROOT
done
python - <<PY
import csv, glob, collections, json
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/tr_{C}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        if "hip::" not in row["Kernel_Name"]: continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("myslam_hip::", "")
        agg[k] += float(row["Counter_Value"]); n[k] += 1
    for k in agg: out.setdefault(k, {})[C + "_KB_per_launch"] = agg[k] / n[k]; out[k]["launches"] = n[k]
print(json.dumps({"pairs_per_step": $P, "steps_total": 3, "note": "raw rocprofv3 counter values (KB) averaged per launch; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md)", "kernels": out}, indent=1))
PY
rm -rf gpurun_out/tr_FETCH_SIZE gpurun_out/tr_WRITE_SIZE
