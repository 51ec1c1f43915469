# One GPU session of round 6: the whole GPU suite, the counter summary of the CURRENT build (it carries the library's build digest: bench.py's roofline.traffic_stale
# turns false), the bench line, rocprofv3 kernel stats of the timed region / the one-stream run / the whole command, B = 1 latencies.   TAG=v64 bash tools/gpu_round6.sh
TAG=${TAG:-v72}
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$TAG.log
timeout 1200 python tools/pmc_collect.py --round 6 --tag $TAG > gpurun_out/pmc_$TAG.log 2>&1; echo "pmc rc=$?"; tail -12 gpurun_out/pmc_$TAG.log
[ -f gpurun_out/r06_pmc_$TAG.json ] && cp gpurun_out/r06_pmc_$TAG.json profiles/r06_pmc_$TAG.json      # so that the bench line below normalises against THIS build's counters
SKIP_PYTEST=1 TAG=$TAG bash tools/gpu_round4.sh
# kernel arguments in device memory (a runtime setting of the application, like GPU_MAX_HW_QUEUES): does the step move?
for v in 0 1; do HIP_FORCE_DEV_KERNARG=$v python bench.py --no-cpu-baseline --parity-frames 0 --stream-input 0 --stream-mode "" --no-extra-passes --steps 100 > gpurun_out/kernarg_${v}_$TAG.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/kernarg_${v}_$TAG.json')); print('HIP_FORCE_DEV_KERNARG=$v', [round(x,3) for x in d['repeats_ms_per_step']])"; done
