#!/bin/bash
# per library variant (in-tree = "new"): the tracker's per-frame call times and the free-run spread on the corridor drive
D=$1
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
for f in /tmp/orig_lib.so $D/lib*.so; do
  n=$(basename $f .so); cp $f $P/libmyslam_hip.so
  echo "$n spread: $(python tools/corridor_spread.py 2>/dev/null | tail -1)"
  for rep in 1 2; do
  python tools/latency_frontend.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$n rep $rep pose_only 150/400:', round(d['pose_only n=150'], 4), round(d['pose_only n=400'], 4))"
  done
done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
