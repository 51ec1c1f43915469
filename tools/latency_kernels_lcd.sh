#!/bin/bash
# per-kernel durations of ONE-frame DeepLCD calls (rocprofv3 kernel trace of a short loop of myslam_lcd_calc_descr_original_img)
cd "$(dirname "$0")/.."; R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
cat > /tmp/lat2.py <<PY
import sys, os, numpy as np
sys.path.insert(0, "$R")
from __graft_entry__ import load_package
import torch
pkg = load_package(); api, synth = pkg.api, pkg.synth
L = synth.stereo_pair(0, 0)[0]
lcd = api.DeepLCD(synth.calc_weights())
for _ in range(60): lcd.calcDescrOriginalImg(L)
PY
( cd /tmp && rm -rf /tmp/latk2 && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/latk2 -o a -- python /tmp/lat2.py > /tmp/lat2.out 2>&1 ) || tail -5 /tmp/lat2.out
f=$(find /tmp/latk2 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/latency_kernel_stats_lcd.csv && python - "$f" <<'PY'
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
