#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu --timeout 180 --timeout-method thread -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
for W in 4 5 6 8; do echo "SLAM_PL_WAVES=$W"; SLAM_PL_WAVES=$W timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep particle_lanes; done
timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep beam_lanes
