#!/bin/bash
# Kernel timeline of ONE recorded 1-pair step on one lane: per-kernel duration and the gap to the previous kernel of the chain.
#   tools/lane_trace.sh <tag> [pairs] [lanes]
TAG=${1:-x}; PAIRS=${2:-1}; LANES=${3:-1}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lane_$TAG -o t -- python $R/bench.py --pairs $PAIRS --lanes $LANES --graph 1 --steps 200 --no-extra-passes --no-cpu-baseline --parity-frames 0 > $R/gpurun_out/lane_$TAG.json 2> $R/gpurun_out/lane_$TAG.err )
f=$(find $R/gpurun_out/lane_$TAG -name "*kernel_trace.csv" | head -1)
python - "$f" "$R/gpurun_out/lane_trace_$TAG.json" <<'PY'
import csv, sys, json, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-200 * 30:]                   # the tail: replayed steps only
# one step = from k_ingest to the next k_ingest
steps, cur = [], []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("myslam_hip::", "")
    if (n.startswith("k_ingest") or n.startswith("k_pyr_head")) and cur:
        steps.append(cur); cur = []
    cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
steps = [s for s in steps[5:-1] if len(s) == len(steps[len(steps) // 2])]
agg = collections.OrderedDict()
for s in steps:
    for i, (n, a, b) in enumerate(s):
        key = f"{i:02d} {n}"
        d = agg.setdefault(key, [0.0, 0.0])
        d[0] += (b - a) / 1e3; d[1] += ((a - s[i - 1][2]) / 1e3 if i else 0.0)
out = {"steps": len(steps), "nodes": len(agg), "step_us": sum(s[-1][2] - s[0][1] for s in steps) / len(steps) / 1e3,
       "kernels": {k: {"dur_us": round(v[0] / len(steps), 2), "gap_before_us": round(v[1] / len(steps), 2)} for k, v in agg.items()}}
out["sum_dur_us"] = round(sum(v["dur_us"] for v in out["kernels"].values()), 1); out["sum_gap_us"] = round(sum(v["gap_before_us"] for v in out["kernels"].values()), 1)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $R/gpurun_out/lane_$TAG
