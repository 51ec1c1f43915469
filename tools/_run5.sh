mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_orb.py tests/test_gpu_fallbacks.py tests/test_gpu_facade.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python tools/gpu_fuzz_orb.py 7 300 2>&1 | tail -3
for r in 6000 300; do
python bench.py --no-cpu-baseline --steps 20 --scene-rects $r 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pipelined rects $r fps', round(d['value']), 'ms', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['profiled_pass']['kernel_ms_per_step'].items() if k in ('k_fast_strip','k_octree','k_describe2')})"
python bench.py --no-cpu-baseline --steps 10 --scene-rects $r --streams 1 --orb-internal-stream 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('1-stream rects $r fps', round(d['value']), {k: round(v,3) for k,v in d['profiled_pass']['kernel_ms_per_step'].items()})"
done
