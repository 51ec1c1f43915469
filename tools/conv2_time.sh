#!/bin/bash
# times k_conv2_patch_bf16x6 with parts of its loop compiled out (CV2_MODE, calc.hip): where does a K step's time go?
cd "$(dirname "$0")/.."
PKG=a-simple-stereo-slam-system-with-deep-loop-closing_amd
for m in 0 1 2 3 4; do
  touch $PKG/csrc/calc.hip
  MYSLAM_HIPCC_EXTRA="-DCV2_MODE=$m" python $PKG/build.py > /dev/null 2>&1 || { echo "mode $m: build failed"; continue; }
  echo -n "mode $m: "; timeout 300 python tools/conv2_time.py 2>/dev/null | tail -1
done
touch $PKG/csrc/calc.hip; python $PKG/build.py > /dev/null 2>&1
