#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_b6
rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/trace -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1; echo trace rc=$?
python3 - <<PY
import sqlite3,glob
db=glob.glob("$O/trace/*.db")[0]
cur=sqlite3.connect(db).cursor()
rows=cur.execute("select name,count(*),avg(end-start),sum(end-start) from kernels group by name order by 4 desc").fetchall()
for r in rows[:25]: print("%-90s %5d %10.0f %12d"%(r[0][:90],r[1],r[2],r[3]))
PY
