#!/bin/bash
# second SQ counter set (stall breakdown) for tools/sbench.py; usage like prof_sbench.sh
set -u
TAG=${1:-sb}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SB_CPU=0
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU -d "$OUT/pmc_sq2" -o sb -- python3 $REPO/tools/sbench.py > "$OUT/pmc_sq2.log" 2>&1 || exit 2
cd $REPO && python3 - "$OUT" <<'PY'
import glob, sys
sys.path.insert(0, "profiles")
import summarize
for d in glob.glob(sys.argv[1] + "/pmc_sq2/*_results.db"):
    print("\n".join(l for l in summarize.pmc_stats(d).split("\n") if "weights_S" in l))
PY
