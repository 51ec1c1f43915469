#!/bin/bash
# the whole bench line (every pass) for the descriptor-kernel candidates; same box, 2 rounds
P=a-simple-stereo-slam-system-with-deep-loop-closing_amd
cp $P/libmyslam_hip.so /tmp/orig_lib.so
run() { # name lib extra-args
  cp $2 $P/libmyslam_hip.so
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 $3 > gpurun_out/gf_$1_$rep.json 2> gpurun_out/gf_$1_$rep.err
  python -c "
import json
d = json.load(open('gpurun_out/gf_$1_$rep.json'))
print('$1', $rep, [round(x, 3) for x in d['repeats_ms_per_step']], 'cad6', round(d['full_solve_cadence6']['ms_per_step'], 3), 'every', round(d['full_solve_every_frame']['ms_per_step'], 3), 'streamed', round(d['streamed']['value']), 'sparse', round(d['kitti_like_scene']['value']), 'grow', round(d['db_grow']['ms_per_step'], 3), 'lanes', round(d['step_graph']['value']), 'sm', [round(p['value']) for p in d['stream_mode']['sweep']], 'parity', d['parity_sample']['ok'])"
}
for rep in 1 2; do
  run glds0_side2 tools/build/ab/libglds0.so ""
  run glds2_side0 tools/build/ab/libglds2.so "--side-blocks-per-cu 0"
  run glds3_side4 /tmp/orig_lib.so "--side-blocks-per-cu 4"
  run glds3_side0 /tmp/orig_lib.so "--side-blocks-per-cu 0"
done
cp /tmp/orig_lib.so $P/libmyslam_hip.so
