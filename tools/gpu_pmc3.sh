#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_kb3
rm -rf $O
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $O/pmc -o kb -- python3 $GRAFT_REPO_ROOT/tools/kbench.py > $O.log 2>&1; echo pmc rc=$?
python3 $GRAFT_REPO_ROOT/tools/show_pmc.py $(find $O -name '*.db' | head -1)
