#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu --timeout 180 --timeout-method thread -q -x 2>&1 | tail -2
for S in 6 9 12 18 24 30; do echo "SEG=$S"; SLAM_PL_SEG=$S timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep -E "particle_lanes|rror" | tr '\n' ' ' | sed 's/particle_lanes//g; s/G pb\/s//g; s/ms  ->  *[0-9.]*//g'; echo; done
