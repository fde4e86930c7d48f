#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_kb5
rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/trace -o kb -- python3 $GRAFT_REPO_ROOT/tools/kbench.py > $O/trace.log 2>&1; echo trace rc=$?
python3 - <<PY
import sqlite3,glob
db=glob.glob("$O/trace/*.db")[0]
cur=sqlite3.connect(db).cursor()
rows=cur.execute("select name,count(*),avg(end-start),sum(end-start) from kernels group by name order by 4 desc").fetchall()
for r in rows[:14]: print("%-80s %5d %10.0f %12d"%(r[0][:80],r[1],r[2],r[3]))
PY
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $O/pmc -o kb -- python3 $GRAFT_REPO_ROOT/tools/kbench.py > $O/pmc.log 2>&1; echo pmc rc=$?
python3 $GRAFT_REPO_ROOT/tools/show_pmc.py $(find $O/pmc -name '*.db' | head -1) | grep rays
