#!/bin/bash
# per-kernel durations of ONE-frame calls (rocprofv3 kernel trace of a short loop of myslam_orb_detect_and_compute)
cd "$(dirname "$0")/.."; R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cat > /tmp/lat1.py <<PY
import sys, os, numpy as np
sys.path.insert(0, "$R")
from __graft_entry__ import load_package
import torch
pkg = load_package(); api, synth = pkg.api, pkg.synth
L = synth.stereo_pair(0, 0)[0]
e = api.ORBextractor(2000)
for _ in range(60): e.DetectAndCompute(L)
PY
( cd /tmp && rm -rf /tmp/latk && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/latk -o a -- python /tmp/lat1.py > /dev/null 2>&1 )
f=$(find /tmp/latk -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/latency_kernel_stats.csv && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    n = r["Name"].split("(")[0].replace("void ", "").replace("myslam_hip::", "")
    if not n.startswith("k_"): continue
    per_call = float(r["TotalDurationNs"]) / 60 / 1e3
    tot += per_call
    print(f"{n:28s} launches/call {int(r['Calls']) / 60:5.1f}  us/call {per_call:7.1f}  avg us {float(r['AverageNs']) / 1e3:6.1f}")
print("sum of kernel time per call: %.1f us" % tot)
PY
