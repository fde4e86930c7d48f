#!/bin/bash
cd $GRAFT_REPO_ROOT
for S in 18 8 4 2 1; do echo "SLAM_PL_SEG=$S"; SLAM_PL_SEG=$S timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep particle_lanes; done
