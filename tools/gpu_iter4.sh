#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu --timeout 300 --timeout-method thread -q -x 2>&1 | tail -2
for W in 6 5 4; do echo "WAVES=$W"; SLAM_PL_WAVES=$W timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep -E "particle_lanes|rror" | tr '\n' ' ' | sed 's/particle_lanes//g; s/G pb\/s//g; s/ms  ->  *[0-9.]*//g'; echo; done
