#!/bin/bash
# same-box A/B of the tracker's two per-frame calls (tools/latency_frontend.py) and of the compiled runner's rate:
#   tools/ab_frontend.sh <dir with lib*.so variants>     (the in-tree library is "new", every lib*.so of the directory a variant)
D=$1
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
for rep in 1 2 3; do
  for f in /tmp/orig_lib.so $D/lib*.so; do
    n=$(basename $f .so); cp $f $P/libmyslam_hip.so
    echo "$n rep $rep: $(python tools/latency_frontend.py 2>/dev/null | tail -1)"
  done
done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
