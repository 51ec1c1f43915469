# A/B of bench.py launch configurations on one box: tools/ab_streams.sh (results in gpurun_out/ab_*.json)
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --stream-input 0 --parity-frames 0 --no-extra-passes --graph 0 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', round(d['value']), round(d['ms_per_step'],3), 'host', round(d['host_launch_ms_per_step'],3))" || tail -3 gpurun_out/ab_$tag.err; }
for rep in 1 2; do
run s50_w3 --steps 50 --warmup 3
run s50_w30 --steps 50 --warmup 30
run s200_w3 --steps 200 --warmup 3
run s200_w30 --steps 200 --warmup 30
run s600_w30 --steps 600 --warmup 30
run s1200_w30 --steps 1200 --warmup 30
done
