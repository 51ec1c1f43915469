run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --stream-input 0 --parity-frames 0 --no-extra-passes --graph 0 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', round(d['value']), round(d['ms_per_step'],3))" || tail -3 gpurun_out/ab_$tag.err; }
for rep in 1 2; do
for p in 512 256 384 192 768 1024; do run p$p --pairs $p --steps $((102400 / p)); done
done
