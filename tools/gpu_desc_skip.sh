export TMPDIR=/tmp MYSLAM_ORB_AUX=0
for st in 1 2 0; do
  if [ $st = 0 ]; then MYSLAM_EXTRA_FLAGS= python a-simple-stereo-slam-system-with-deep-loop-closing_amd/build.py --force > /dev/null 2>&1
  else MYSLAM_EXTRA_FLAGS=-DMYSLAM_DESC_SKIP=$st python a-simple-stereo-slam-system-with-deep-loop-closing_amd/build.py --force > /dev/null 2>&1; fi
  echo -n "skip=$st "; python bench.py --steps 5 --warmup 1 --pairs 512 --workload orb_match --streams 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['kernel_ms_per_step']['describe'],3))"
done
