# pairs-per-step curve of the bench line (resident input): tools/pairs_sweep_r4.sh
for P in 8 16 32 64 128 256; do
  timeout 600 python bench.py --pairs $P --steps $((P >= 128 ? 100 : 400)) --warmup 5 --no-cpu-baseline --stream-input 0 > gpurun_out/sweep_$P.json 2> gpurun_out/sweep_$P.err || { echo "pairs $P failed"; tail -3 gpurun_out/sweep_$P.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/sweep_$P.json')); o=d.get('step_eager') or d.get('step_graph') or {}; print($P, round(d['value']), round(d['ms_per_step'],3), d['launch_mode'][:40], 'other', round(o.get('value',0)), 'parity', d['parity_sample']['ok'])"
done
