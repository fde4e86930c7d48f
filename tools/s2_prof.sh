#!/bin/bash
# kernel trace of grid-SLAM bench entries under env switches: prints per-kernel averages
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
    print("\n".join(l for l in summarize.kernel_stats(d).split("\n")[:16]))
PY
rm -rf $OUT/trace
# PMC=1: one SQ counter pass as well (the S2 kernels' rows only)
if [ -n "$PMC" ]; then
  cd /tmp
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc -o b -- python3 $REPO/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
  cd $REPO && python3 - $OUT <<'PY'
import glob, sys
sys.path.insert(0, "profiles")
import summarize
for d in glob.glob(sys.argv[1] + "/pmc/*_results.db"):
    print("\n".join(l for l in summarize.pmc_stats(d).split("\n") if "k_weights_S2" in l or l.startswith("kernel")))
PY
  rm -rf $OUT/pmc
fi
