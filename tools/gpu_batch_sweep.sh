for p in 256 512 1024; do
  s=$(date +%s.%N)
  python bench.py --steps 6 --warmup 2 --pairs $p --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pairs', d['config']['pairs_per_step_per_gpu'], 'value',round(d['value'],1),'ms/step',round(d['ms_per_step'],3))"
  e=$(date +%s.%N); echo "wall $(echo "$e - $s" | bc) s"
done
