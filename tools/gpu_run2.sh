#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/pytest.log
for f in tests/test_gpu_match_tri.py tests/test_gpu_ba.py tests/test_gpu_lcd.py tests/test_gpu_orb.py tests/test_golden.py; do
  echo "=== $f" >> gpurun_out/pytest.log
  timeout 900 python -m pytest $f -m gpu -q --tb=short --timeout 300 -p no:cacheprovider 2>&1 | tail -80 >> gpurun_out/pytest.log
done
for P in 64 256; do
  echo "=== bench pairs=$P" >> gpurun_out/pytest.log
  timeout 900 python bench.py --steps 5 --warmup 2 --pairs $P --no-cpu-baseline > gpurun_out/bench_p$P.json 2> gpurun_out/bench_p$P.err
  tail -3 gpurun_out/bench_p$P.err >> gpurun_out/pytest.log
  python -c "
import json;d=json.load(open('gpurun_out/bench_p$P.json'));print('value',d['value'],'ms/step',d['ms_per_step']);print(json.dumps(d['kernel_ms_per_step']))" >> gpurun_out/pytest.log
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o orb -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --pairs 64 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_r1 | head -20 >> gpurun_out/pytest.log
grep -E "^===|passed|failed|error|value|resize" gpurun_out/pytest.log | tail -40
