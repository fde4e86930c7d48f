#!/bin/bash
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_kb2
timeout -s KILL 200 rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM -d $GRAFT_REPO_ROOT/gpurun_out/prof_kb2/pmc -o kb -- python3 $GRAFT_REPO_ROOT/tools/kbench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_kb2.log 2>&1; echo pmc rc=$?
