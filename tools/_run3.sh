mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_v22.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_v22.log
for r in 300 1000 2000 3500 6000; do for m in 0 1; do
python bench.py --no-cpu-baseline --steps 10 --pairs 256 --workload orb_match --fast-mode $m --scene-rects $r --streams 1 --orb-internal-stream 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rects $r mode $m fast', round(d['profiled_pass']['kernel_ms_per_step']['k_fast_strip'],3), 'fps', round(d['value']))"
done; done
