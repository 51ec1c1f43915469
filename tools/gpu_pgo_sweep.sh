#!/bin/bash
# run-length sweep of the pose-graph solver (developer tuning)
for r in 16 32 64 128 256; do echo "run $r"; MYSLAM_PGO_RUN=$r PGO_SIZES=${PGO_SIZES:-200:2,1500:6,20000:40} python tools/pgo_time.py 2>&1 | grep "n="; done
