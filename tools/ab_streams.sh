# A/B of bench.py launch configurations on one box: tools/ab_streams.sh (results in gpurun_out/ab_*.json)
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline --stream-input 0 "$@" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python -c "
import json; d=json.load(open('gpurun_out/ab_$tag.json')); o=d.get('step_eager') or d.get('step_graph') or {}; l=d.get('step_lanes_eager') or {}; print('$tag', d['launch_mode'][:20], round(d['value']), round(d['ms_per_step'],3), 'host', round(d['host_launch_ms_per_step'],3), '| other', round(o.get('value',0)), round(o.get('ms_per_step',0),3), round(o.get('host_launch_ms_per_step',0),3), '| lanes eager', round(l.get('value',0)), round(l.get('host_launch_ms_per_step',0),3), 'parity', (d.get('parity_sample') or {}).get('ok'))" || tail -3 gpurun_out/ab_$tag.err; }
GPU_MAX_HW_QUEUES=32 run q32_l16_8 --pairs 8 --steps 400 --lanes 16
GPU_MAX_HW_QUEUES=32 run q32_l24_8 --pairs 8 --steps 400 --lanes 24
run def_8 --pairs 8 --steps 400
run def_16 --pairs 16 --steps 400
run def_32 --pairs 32 --steps 200
run def_64 --pairs 64 --steps 100
run def_128 --pairs 128 --steps 50
run def_512
GPU_MAX_HW_QUEUES=8 run q8_512
GPU_MAX_HW_QUEUES=8 run q8_l8_512 --lanes 8
GPU_MAX_HW_QUEUES=8 run q8_l2_512 --lanes 2
