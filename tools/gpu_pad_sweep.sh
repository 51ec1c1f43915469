#!/bin/bash
# pipelined bench vs FAST occupancy (dynamic LDS padding caps its resident blocks per CU)
for rep in 1 2; do
for pad in ${PADS:-0 8192 11000 14000 18000 22000}; do
  MYSLAM_FAST_LDS_PAD=$pad python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad $pad', round(d['value']), round(d['ms_per_step'],3), 'fast', round(d['roofline']['avg_launch_ms'],3), 'valu frac', round(d['roofline_valu']['frac'],3))"
done; done
