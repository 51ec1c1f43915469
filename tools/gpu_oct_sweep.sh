#!/bin/bash
# developer aid: oct-tree block size sweep
export TMPDIR=/tmp
for ot in 1024 512 256; do
  sed -i "s/^constexpr int OT = [0-9]*;/constexpr int OT = $ot;/" a-simple-stereo-slam-system-with-deep-loop-closing_amd/csrc/orb_kernels.hip
  python a-simple-stereo-slam-system-with-deep-loop-closing_amd/build.py > /dev/null 2>&1
  echo -n "OT=$ot "; python bench.py --steps 5 --warmup 1 --pairs 256 --workload orb_match --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['kernel_ms_per_step']['octree'],3))"
done
timeout 900 python -m pytest tests/test_gpu_orb.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
