#!/bin/bash
# per-kernel times with and without -amdgpu-mfma-vgpr-form on orb_kernels.hip (flag effect on the non-MFMA kernels)
run() { MYSLAM_ORB_AUX=0 python bench.py --streams 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$1', round(d['ms_per_step'],3), 'fast', round(k['fast_cells'],3), 'describe', round(k['describe'],3), 'octree', round(k['octree'],3), 'resize', round(k['resize'],3), 'blur', round(k['blur7'],3))"; }
run with-flag; run with-flag
sed -i 's/"orb_kernels.hip": EXACT + \["-mllvm", "-amdgpu-mfma-vgpr-form"\],/"orb_kernels.hip": EXACT,/' a-simple-stereo-slam-system-with-deep-loop-closing_amd/build.py
python a-simple-stereo-slam-system-with-deep-loop-closing_amd/build.py > /dev/null 2>&1
run no-flag; run no-flag
