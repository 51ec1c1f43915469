mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_v22.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_v22.log
for m in -1 0 1; do for r in 6000 300; do
python bench.py --no-cpu-baseline --steps 20 --fast-mode $m --scene-rects $r > gpurun_out/b_${m}_${r}.json 2> gpurun_out/b_${m}_${r}.err
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/b_${m}_${r}.json"))
    k = d["profiled_pass"]["kernel_ms_per_step"]
    print("mode $m rects $r: fps", round(d["value"]), "ms", round(d["ms_per_step"], 3), "fast", round(k.get("k_fast_strip", 0), 3), "kp", round(d["config"]["keypoints_per_image"]))
except Exception as e: print("mode $m rects $r failed", e)
PY
done; done
for m in 0 1; do python bench.py --no-cpu-baseline --steps 20 --fast-mode $m --streams 1 --orb-internal-stream 0 --no-extra-passes > /dev/null 2>&1; python bench.py --no-cpu-baseline --steps 10 --fast-mode $m --streams 1 --orb-internal-stream 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('1-stream mode $m', round(d['value']), {k: round(v,3) for k,v in d['profiled_pass']['kernel_ms_per_step'].items()})"; done
