#!/bin/bash
# grid FAST in blocks of NS consecutive strips with the next strip's tile prefetched by LDS-DMA (MYSLAM_FAST_NS): correctness of each variant library (the extractor's GPU
# tests against the oracle, batch sizes that take the multi-strip launch included), then the same-box A/B
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib0.so
for n in "$@"; do
  cp tools/build/ab/lib$n.so $P/libmyslam_hip.so
  timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_fallbacks.py -x -q > gpurun_out/ms_pytest_$n.log 2>&1; echo "$n pytest rc=$?"; tail -2 gpurun_out/ms_pytest_$n.log
  python bench.py --no-cpu-baseline --parity-frames 8 --stream-input 0 --stream-mode "" --no-extra-passes --steps 10 2>gpurun_out/ms_par_$n.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n parity', d['parity_sample']['ok'], d['parity_sample'].get('mismatches'))"
done
cp /tmp/orig_lib0.so $P/libmyslam_hip.so
bash tools/ab_alone.sh ms base "$@"
