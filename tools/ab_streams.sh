# A/B of bench.py's streamed pass on one box: tools/ab_streams.sh
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --parity-frames 0 --graph 0 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab_$tag.json')); s=d['streamed']; print('$tag', round(d['value']), 'streamed', round(s['value']), round(s['h2d_GBps'],1), round(s['h2d_copy_ms_avg'],2))" || tail -3 gpurun_out/ab_$tag.err; }
for rep in 1 2; do
run base
GPU_MAX_HW_QUEUES=6 run split2 --stream-split 2
GPU_MAX_HW_QUEUES=8 run split4 --stream-split 4
GPU_MAX_HW_QUEUES=6 run split4_q6 --stream-split 4
done
