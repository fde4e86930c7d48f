#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in "" tools/exp/libslam_lat2.so; do echo "LIB=$L"; KB_NOCHECK=1 SLAM_HIP_LIB=$L timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep -E "particle_lanes|rror" | tr '\n' ' ' | sed 's/particle_lanes//g; s/G pb\/s//g; s/ms  ->  *[0-9.]*//g'; echo; done
