#!/bin/bash
# internal blur stream: off / after FAST / after the pyramid; one and two bench streams
for m in 0 1 2; do for st in 1 2; do
  MYSLAM_ORB_AUX=$m python bench.py --streams $st --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('aux $m streams $st', round(d['value']), round(d['ms_per_step'],3))"
done; done
