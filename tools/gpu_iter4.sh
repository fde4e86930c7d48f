#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu --timeout 180 --timeout-method thread -q -x 2>&1 | tail -2
for W in 4 6 8; do for O in 1 3; do echo "WAVES=$W ORDER=$O"; SLAM_PL_WAVES=$W SLAM_PL_ORDER=$O timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep -E "particle_lanes|Error|error" | tr '\n' ' ' | sed 's/particle_lanes//g; s/G pb\/s//g'; echo; done; done
