#!/bin/bash
# schedule knobs re-measured after the descriptor kernel's direct-to-LDS form (build v69): extractor handles per step, DeepLCD chain split, internal blur stream; same box, 2 rounds
run() { # name args
  python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --stream-mode "" --no-extra-passes --steps 60 $2 > gpurun_out/sk_$1_$rep.json 2> gpurun_out/sk_$1_$rep.err
  python -c "
import json
try:
    d = json.load(open('gpurun_out/sk_$1_$rep.json')); print('$1', $rep, [round(x, 3) for x in d['repeats_ms_per_step']])
except Exception as e: print('$1 failed', e)"
}
for rep in 1 2; do
  run base ""
  run split3 "--orb-split 3 --pairs 510"
  run split4 "--orb-split 4"
  run lcd2 "--lcd-split 2"
  run lcd4 "--lcd-split 4"
  run internal1 "--orb-internal-stream 1"
  run split4_lcd2 "--orb-split 4 --lcd-split 2"
done
