#!/bin/bash
cd $GRAFT_REPO_ROOT
# kind 3; tile shift field = shift+1; heading bits
for M in 3 $((3+(1<<4)+(8<<8))) $((3+(2<<4)+(10<<8))) $((3+(3<<4)+(10<<8))) $((3+(2<<4)+(12<<8))) $((3+(3<<4)+(12<<8))) $((3+(4<<4)+(12<<8))) $((3+(2<<4)+(6<<8))) $((3+(3<<4)+(14<<8))); do echo "ORDER=$M"; SLAM_PL_ORDER=$M timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep -E "particle_lanes|rror" | tr '\n' ' ' | sed 's/particle_lanes//g; s/G pb\/s//g; s/ms  ->  *[0-9.]*//g'; echo; done
