#!/bin/bash
# ORB parity tests + one-stream per-kernel times + default bench value
timeout 900 python -m pytest tests/test_gpu_orb.py -x -q 2>&1 | tail -3
python bench.py --streams 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['value']), round(d['ms_per_step'],3))"
