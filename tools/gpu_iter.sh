#!/bin/bash
# one GPU iteration: parity tests, kernel microbench, SQ counters of the microbench
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu --timeout 180 --timeout-method thread -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -u tools/kbench.py 2>&1 | grep -v amdgpu.ids
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_kb
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD -d $GRAFT_REPO_ROOT/gpurun_out/prof_kb/pmc_sq -o kb -- python3 $GRAFT_REPO_ROOT/tools/kbench.py > $GRAFT_REPO_ROOT/gpurun_out/prof_kb_sq.log 2>&1; echo pmc rc=$?
