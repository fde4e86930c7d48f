#!/bin/bash
# rocprofv3 kernel trace (+ SQ counters in a run of their own) of tools/sbench.py; run through gpurun from the repo root.
# Usage: SB_N=.. SB_ONLY=.. bash tools/prof_sbench.sh <tag>
set -u
TAG=${1:-sb}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SB_CPU=0
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o sb -- python3 $REPO/tools/sbench.py > "$OUT/trace.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d "$OUT/pmc_sq" -o sb -- python3 $REPO/tools/sbench.py > "$OUT/pmc_sq.log" 2>&1 || exit 2
cd $REPO && python3 - "$OUT" <<'PY'
import glob, sys
sys.path.insert(0, "profiles")
import summarize
out = sys.argv[1]
for d in glob.glob(out + "/trace/*_results.db"):
    print("\n".join(summarize.kernel_stats(d).split("\n")[:12]))
for d in glob.glob(out + "/pmc_sq/*_results.db"):
    print("\n".join(l for l in summarize.pmc_stats(d).split("\n") if "weights_S" in l))
PY
