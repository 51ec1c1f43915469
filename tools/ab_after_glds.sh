#!/bin/bash
# after the descriptor kernel's direct-to-LDS form (shorter blocks, unlimited grid): the whole GPU suite on the new build, then FAST's blocks per CU re-measured
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_glds.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_glds.log
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
run() { # name lib extra-args
  cp $2 $P/libmyslam_hip.so
  python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --stream-mode "" --no-extra-passes --steps 60 $3 > gpurun_out/ag_$1_$rep.json 2> gpurun_out/ag_$1_$rep.err
  python -c "
import json
d = json.load(open('gpurun_out/ag_$1_$rep.json')); print('$1', $rep, [round(x, 3) for x in d['repeats_ms_per_step']])"
}
for rep in 1 2 3; do
  run base /tmp/orig_lib.so ""
  run fast5 tools/build/ab/libfast5.so ""
  run fast7 tools/build/ab/libfast7.so ""
  run fast7w6 tools/build/ab/libfast7w6.so ""
done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
