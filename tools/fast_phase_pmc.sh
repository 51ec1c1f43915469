#!/bin/bash
# Dynamic instruction split of k_fast_strip (dense path): one library per MYSLAM_FAST_PHASE (built here by
#   tools/build_variants.sh orb_kernels.hip ph1:-DMYSLAM_FAST_PHASE=1 ... ph5:-DMYSLAM_FAST_PHASE=5 ph0:-DMYSLAM_FAST_PHASE=0),
# each run under rocprofv3 --pmc (SQ counters, one-stream bench, dense path forced); tools/fast_phase_report.py differences the counts.
#   tools/fast_phase_pmc.sh <tag> [dir]
TAG=${1:-x}; D=${2:-tools/build/ab}
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
for f in $D/libph*.so; do
  n=$(basename $f .so); n=${n#lib}
  cp $f $P/libmyslam_hip.so
  timeout 600 python tools/pmc_collect.py --round 5 --tag $n --sq-only --pairs 64 --workload orb_match --bench-arg=--fast-mode --bench-arg=1 > gpurun_out/fastphase_${n}_$TAG.log 2>&1
  echo "$n rc=$?"; grep k_fast_strip gpurun_out/fastphase_${n}_$TAG.log
done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
