#!/bin/bash
# k_describe2 phase C with direct-to-LDS loads, 2 / 3 / 4 windows per wave in flight (MYSLAM_KD_GLDS): correctness of each variant library (the extractor's
# GPU tests against the oracle), then the same-box A/B (tools/ab_alone.sh).
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib0.so
for n in "$@"; do
  cp tools/build/ab/lib$n.so $P/libmyslam_hip.so
  timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_fallbacks.py -x -q > gpurun_out/glds_pytest_$n.log 2>&1; echo "$n pytest rc=$?"; tail -2 gpurun_out/glds_pytest_$n.log
done
cp /tmp/orig_lib0.so $P/libmyslam_hip.so
bash tools/ab_alone.sh glds base "$@"
