#!/bin/bash
# conv2 tilings alone and beside FAST (FAST occupancy capped by dynamic-LDS padding so that the small tiling can co-reside)
for v in 0 1 2; do
  MYSLAM_CONV2_V=$v timeout 300 python -m pytest tests/test_gpu_lcd.py -x -q 2>&1 | tail -1
  MYSLAM_CONV2_V=$v python bench.py --streams 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conv2 v$v alone', round(d['kernel_ms_per_step']['calc_conv2'],3), 'step', round(d['ms_per_step'],3))"
  for pad in 0 4100; do
    MYSLAM_CONV2_V=$v MYSLAM_FAST_LDS_PAD=$pad python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('conv2 v$v pad $pad: fps', round(d['value']), 'step', round(d['ms_per_step'],3), 'fast', round(k['fast_cells'],2), 'conv2', round(k['calc_conv2'],2))"
  done
done
