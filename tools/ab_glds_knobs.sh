#!/bin/bash
# the descriptor kernel with 3 windows per wave in flight (in-tree) against 0 (register-staged) and 2, and its blocks-per-CU knob re-measured: same box, 3 rounds
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
run() { # name lib extra-args
  cp $2 $P/libmyslam_hip.so
  python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --stream-mode "" --no-extra-passes --steps 60 $3 > gpurun_out/gk_$1_$rep.json 2> gpurun_out/gk_$1_$rep.err
  python -c "
import json
d = json.load(open('gpurun_out/gk_$1_$rep.json')); print('$1', $rep, [round(x, 3) for x in d['repeats_ms_per_step']])"
}
for rep in 1 2 3; do
  run glds0 tools/build/ab/libglds0.so ""
  run glds2 tools/build/ab/libglds2.so ""
  run glds3 /tmp/orig_lib.so ""
  run glds3_side1 /tmp/orig_lib.so "--side-blocks-per-cu 1"
  run glds3_side3 /tmp/orig_lib.so "--side-blocks-per-cu 3"
  run glds3_side4 /tmp/orig_lib.so "--side-blocks-per-cu 4"
  run glds2_side3 tools/build/ab/libglds2.so "--side-blocks-per-cu 3"
done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
