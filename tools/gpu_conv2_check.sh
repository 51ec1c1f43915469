#!/bin/bash
# conv2 variant check: LCD parity tests, conv2 time alone, default bench
timeout 300 python -m pytest tests/test_gpu_lcd.py -x -q 2>&1 | tail -2
python bench.py --streams 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one stream: conv2', round(d['kernel_ms_per_step']['calc_conv2'],3), 'step', round(d['ms_per_step'],3))"
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('default: fps', round(d['value']), 'step', round(d['ms_per_step'],3), 'fast', round(k['fast_cells'],2), 'conv2', round(k['calc_conv2'],2))"; done
