#!/bin/bash
# descriptor kernel, direct-to-LDS form: blocks per CU of its limited grid swept further (2 / 3 windows per wave in flight); same box, 2 rounds
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
run() { # name lib extra-args
  cp $2 $P/libmyslam_hip.so
  python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --stream-mode "" --no-extra-passes --steps 60 $3 > gpurun_out/gk2_$1_$rep.json 2> gpurun_out/gk2_$1_$rep.err
  python -c "
import json
d = json.load(open('gpurun_out/gk2_$1_$rep.json')); print('$1', $rep, [round(x, 3) for x in d['repeats_ms_per_step']])"
}
for rep in 1 2; do
  run glds0_side2 tools/build/ab/libglds0.so ""
  for sb in 4 5 6 8 0; do
    run glds3_side$sb /tmp/orig_lib.so "--side-blocks-per-cu $sb"
    run glds2_side$sb tools/build/ab/libglds2.so "--side-blocks-per-cu $sb"
  done
done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
