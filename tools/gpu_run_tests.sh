#!/bin/bash
# run every GPU test file in its own process (a fault in one must not hide the others), then a short bench
mkdir -p gpurun_out
export TMPDIR=/tmp
for f in tests/test_gpu_match_tri.py tests/test_gpu_ba.py tests/test_gpu_lcd.py tests/test_gpu_orb.py tests/test_golden.py; do
  echo "=== $f" >> gpurun_out/pytest.log
  timeout 900 python -m pytest $f -m gpu -q --tb=short --timeout 300 -p no:cacheprovider 2>&1 | tail -120 >> gpurun_out/pytest.log
done
echo "=== smoke" >> gpurun_out/pytest.log
timeout 300 python __graft_entry__.py smoke >> gpurun_out/pytest.log 2>&1
echo "=== bench" >> gpurun_out/pytest.log
timeout 900 python bench.py --steps 3 --warmup 1 --pairs 16 --cpu-pairs 2 > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err
tail -5 gpurun_out/bench_small.err >> gpurun_out/pytest.log
cat gpurun_out/bench_small.json >> gpurun_out/pytest.log
grep -E "^===|passed|failed|error" gpurun_out/pytest.log | tail -40
