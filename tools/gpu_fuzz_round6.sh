# closing fuzz campaign of round 6 on the current build: the campaigns of rounds 4 / 5 (tools/gpu_fuzz_round4.sh) + round 6's new paths.   tools/gpu_fuzz_round6.sh [seed base]
S=${1:-6000}
bash tools/gpu_fuzz_round4.sh $S 1
for k in 1 2 3 4; do timeout 900 python tools/gpu_fuzz_round6.py $((S + k)) 80 4 > gpurun_out/fuzz_r6_$((S + k)).log 2>&1; echo "round6 rc=$?"; tail -1 gpurun_out/fuzz_r6_$((S + k)).log; done
python -c "
import glob, re
tot = 0; bad = 0
for f in sorted(glob.glob('gpurun_out/fuzz_*_$S.log') + glob.glob('gpurun_out/fuzz_r6_*.log')):
    t = open(f).read()
    for m in re.finditer(r'(\d+) mismatches', t): bad += int(m.group(1))
    print(f.split('/')[-1], '|', t.strip().splitlines()[-1][:160])
print('TOTAL mismatches', bad)
" | tee gpurun_out/r06_fuzz_final.log
