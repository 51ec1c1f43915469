mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sequence.py tests/test_gpu_orb.py -m gpu -x -q -s 2>&1 | tail -30
