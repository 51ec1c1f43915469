# samples the shader clock and power while bench.py's timed region runs: tools/clock_probe.sh [bench args]
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_probe.log 2>&1 &
SMI=$!
timeout 600 python bench.py --no-cpu-baseline --stream-input 0 --parity-frames 0 --no-extra-passes --graph 0 --steps 1200 "$@" > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err
kill $SMI 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/clock_bench.json')); print('fps', round(d['value']), 'ms', round(d['ms_per_step'],3))"
sort gpurun_out/clock_probe.log | uniq -c | sort -rn | head -12
