# One GPU session of round 5: the whole GPU suite, the counter summary of the CURRENT build (bench.py reads the newest profiles/r*_pmc_*.json), the
# bench line, rocprofv3 kernel stats of the timed region / the one-stream run / the whole command, B = 1 latencies.   TAG=v55 bash tools/gpu_round5.sh
TAG=${TAG:-v55}
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_$TAG.log
timeout 1200 python tools/pmc_collect.py --round 5 --tag $TAG > gpurun_out/pmc_$TAG.log 2>&1; echo "pmc rc=$?"; tail -12 gpurun_out/pmc_$TAG.log
[ -f gpurun_out/r05_pmc_$TAG.json ] && cp gpurun_out/r05_pmc_$TAG.json profiles/r05_pmc_$TAG.json      # so that the bench line below normalises against THIS build's counters
SKIP_PYTEST=1 TAG=$TAG bash tools/gpu_round4.sh
