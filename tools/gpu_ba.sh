#!/bin/bash
# BA parity + full_solve bench
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_facade.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider 2>&1 | tail -${TAILN:-12}
timeout 600 python bench.py --steps 10 --warmup 2 --pairs ${PAIRS:-256} --workload full_solve --streams 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',round(d['value'],1),'ms/step',round(d['ms_per_step'],3))
print({k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
