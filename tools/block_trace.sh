#!/bin/bash
# Block trace of the pipelined step (profiling build: tools/build_variants.sh orb_kernels.hip bt:-DMYSLAM_BLOCK_TRACE built beforehand).
#   tools/block_trace.sh <tag> [bench args...]   -> gpurun_out/bt_<tag>.npy, gpurun_out/r06_valu_timeline_<tag>.json
TAG=${1:-x}; shift
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
cp tools/build/ab/libbt.so $P/libmyslam_hip.so
timeout 600 python bench.py --no-cpu-baseline --no-extra-passes --parity-frames 0 --steps 20 --block-trace gpurun_out/bt_$TAG.npy "$@" > gpurun_out/bt_bench_$TAG.json 2> gpurun_out/bt_bench_$TAG.err
echo "bench rc=$?"; tail -2 gpurun_out/bt_bench_$TAG.err
cp /tmp/orig_lib.so $P/libmyslam_hip.so
python tools/block_trace_report.py gpurun_out/bt_$TAG.npy > gpurun_out/r06_valu_timeline_$TAG.json 2> gpurun_out/bt_report_$TAG.log; echo "report rc=$?"
head -60 gpurun_out/bt_report_$TAG.log; python -c "
import json; d=json.load(open('gpurun_out/bt_bench_$TAG.json')); print('traced build: fps', round(d['value']), 'ms', round(d['ms_per_step'],3))"
