# A/B of bench.py launch configurations on one box: tools/ab_streams.sh (results in gpurun_out/ab_*.json)
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --stream-input 0 --parity-frames 0 --no-extra-passes --graph 0 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', round(d['value']), round(d['ms_per_step'],3), 'host', round(d['host_launch_ms_per_step'],3))" || tail -3 gpurun_out/ab_$tag.err; }
for rep in 1 2; do
for q in 5 4 3 2 6; do GPU_MAX_HW_QUEUES=$q run q$q; done
GPU_MAX_HW_QUEUES=4 run q4_lcd2 --lcd-split 2
done
