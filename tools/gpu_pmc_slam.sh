#!/bin/bash
# PMC passes over the SLAM-mode weighting kernel (tools/sbench.py, tiled canvases)
cd /tmp; export TMPDIR=/tmp
export SB_CPU=0 SB_ONLY=tiled SB_N=${SB_N:-8192} SB_C=2464 SB_ROWS=2354 SB_POOL=2000000
O=$GRAFT_REPO_ROOT/gpurun_out/prof_slam
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $O/p1 -o s -- python3 $GRAFT_REPO_ROOT/tools/sbench.py > $O/p1.log 2>&1 || { echo pass1 failed; tail -5 $O/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU -d $O/p2 -o s -- python3 $GRAFT_REPO_ROOT/tools/sbench.py > $O/p2.log 2>&1 || { echo pass2 failed; tail -5 $O/p2.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/p3 -o s -- python3 $GRAFT_REPO_ROOT/tools/sbench.py > $O/p3.log 2>&1 || { echo pass3 failed; tail -5 $O/p3.log; exit 1; }
for d in p1 p2 p3; do
python3 - $(find $O/$d -name '*.db' | head -1) <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for r in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_weights_S%' group by kernel_name, counter_name"):
    print("%-36s %-24s %4d %16.0f" % (r[0][:36], r[1], r[2], r[3]))
PY
done
