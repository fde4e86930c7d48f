#!/bin/bash
# SQ counters + kernel trace of the headline bench (k_rays_pl); usage: bash tools/prof_headline.sh TAG
set -u
TAG=${1:-hl}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o hl -- python3 $REPO/bench.py --no-secondary --no-cpu-baseline > "$OUT/trace.log" 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d "$OUT/pmc_sq" -o hl -- python3 $REPO/bench.py --no-secondary --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d "$OUT/pmc_l2" -o hl -- python3 $REPO/bench.py --no-secondary --no-cpu-baseline > "$OUT/pmc_l2.log" 2>&1 || exit 4
cd $REPO && python3 - "$OUT" <<'PY'
import glob, sys
sys.path.insert(0, "profiles")
import summarize
for sub in ("pmc_sq", "pmc_l2"):
    for d in glob.glob(sys.argv[1] + f"/{sub}/*_results.db"):
        print("\n".join(l for l in summarize.pmc_stats(d).split("\n") if "k_rays_pl" in l or "k_sum_lw" in l))
for d in glob.glob(sys.argv[1] + "/trace/*_results.db"):
    print("\n".join(summarize.timed_region(d).split("\n")))
PY
