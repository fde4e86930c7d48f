#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   kernel-trace + stats, then PMC passes in runs of their own (never combined with tracing).
# Usage: bash profiles/run_profile.sh <tag>      -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# counter passes: the headline configuration plus the 125 000-particle grid-SLAM shard (k_weights_S2) and the 16384^2 shard
export BENCH_SECONDARY=${BENCH_SECONDARY_PMC:-slam125k,loc16384}
BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
# trace pass 1: the headline alone, bench.py defaults (5 warm-up + 20 timed steps, then 20 more steps with every kernel group
# bracketed by events for kernel_ms_per_step): the same command the bench line comes from, minus the secondary entries
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- python3 $REPO/bench.py --no-cpu-baseline --no-secondary > "$OUT/trace.log" 2>&1 || exit 1
# trace pass 2: the secondary entries (grid-SLAM 1k and 125k shard, 8192^2 and 16384^2 localization shards)
BENCH_SECONDARY=slam1k,slam125k,loc8192,loc16384 timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace_sec" -o bench -- python3 $REPO/bench.py --no-cpu-baseline > "$OUT/trace_sec.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_l2" -o bench -- $BENCH > "$OUT/pmc_l2.log" 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d "$OUT/pmc_grbm" -o bench -- $BENCH > "$OUT/pmc_grbm.log" 2>&1 || exit 5
# summaries on the box (the databases of the counter passes exceed what gpurun copies back), then the databases go
cd "$REPO" && python3 profiles/summarize.py "$TAG" > "$OUT/summary.log" 2>&1 || { tail -20 "$OUT/summary.log"; exit 6; }
mkdir -p "$REPO/gpurun_out/prof_${TAG}_out"
cp profiles/rocprof_${TAG}*.txt profiles/pmc_weights_kernel.json "$REPO/gpurun_out/prof_${TAG}_out/"
grep '^{' "$OUT/trace.log" > "$REPO/gpurun_out/prof_${TAG}_out/bench_line_trace_pass1.json" || true
grep '^{' "$OUT/trace_sec.log" > "$REPO/gpurun_out/prof_${TAG}_out/bench_line_trace_pass2.json" || true
rm -rf "$OUT"/trace "$OUT"/trace_sec "$OUT"/pmc_fetch "$OUT"/pmc_l2 "$OUT"/pmc_sq "$OUT"/pmc_grbm
ls -l "$REPO/gpurun_out/prof_${TAG}_out"
