#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u bench.py 2>&1 | tail -1 > gpurun_out/bench_now.json; cat gpurun_out/bench_now.json
SB_CPU=0 timeout -k 10 300 python -u tools/sbench.py 2>&1 | tail -3
