#!/bin/bash
# two-stream overlap vs FAST occupancy (dynamic LDS padding caps its resident blocks per CU)
for pad in 0 4096 8192 14000 22000 34000; do
  MYSLAM_FAST_LDS_PAD=$pad python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad $pad', round(d['value']), round(d['ms_per_step'],3), 'fast', round(d['kernel_ms_per_step']['fast_cells'],3), 'conv2', round(d['kernel_ms_per_step']['calc_conv2'],3))"
done
