#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu --timeout 180 --timeout-method thread -q -x 2>&1 | tail -2
for W in 4 6 8; do for R in 0 24 32 48; do echo "WAVES=$W REGEN=$R"; SLAM_PL_WAVES=$W SLAM_PL_REGEN=$R timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep -E "particle_lanes|Error|error" | tr '\n' ' ' | sed 's/particle_lanes//g; s/G pb\/s//g'; echo; done; done
