#!/bin/bash
# developer aid: per-phase cycle counts of the oct-tree kernel (image 0, every level)
export TMPDIR=/tmp
MYSLAM_EXTRA_FLAGS=-DMYSLAM_OCT_TIMING python a-simple-stereo-slam-system-with-deep-loop-closing_amd/build.py --force > /dev/null 2>&1
python bench.py --steps 1 --warmup 0 --pairs 16 --workload orb_match --no-cpu-baseline 2>&1 | grep "^oct" | tail -8
