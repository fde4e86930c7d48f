#!/bin/bash
# kernel trace of the 125k grid-SLAM entry under env switches: prints per-kernel averages
# usage: bash tools/s2_prof.sh <tag> [ENV=VAL ...]
TAG=$1; shift
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/s2prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BENCH_SECONDARY=${SEC:-slam125k}
for kv in "$@"; do export "$kv"; done
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o b -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
cd $REPO && python3 - $OUT <<'PY'
import glob, sys
sys.path.insert(0, "profiles")
import summarize
for d in glob.glob(sys.argv[1] + "/trace/*_results.db"):
    print("\n".join(l for l in summarize.kernel_stats(d).split("\n")[:14]))
PY
