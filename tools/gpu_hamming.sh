#!/bin/bash
# Hamming kernel check: parity tests, fuzz, per-kernel time from the one-stream bench
timeout 600 python -m pytest tests/test_gpu_match_tri.py -x -q 2>&1 | tail -3
timeout 300 python tools/gpu_fuzz_misc.py ${SEED:-3} 40 2>&1 | tail -3
python bench.py --streams 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'hamming', d['kernel_ms_per_step']['hamming_match'])"
