#!/bin/bash
# A/B of the grid-SLAM kernels: the bench's grid-SLAM entries under library variants / env switches
# usage: bash tools/s2_ab.sh name[:ENV=VAL[:ENV=VAL]] ...
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  IFS=: read -ra parts <<< "$spec"
  echo "== ${parts[0]}"
  env "${parts[@]:1}" BENCH_SECONDARY=${SEC:-slam1k,slam125k} timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 2 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        for e in d.get('secondary', []):
            print('   %.3f ms/step  kernel %.3f ms  groups %s  %s' % (e['ms_per_step'], e['roofline']['kernel_avg_ms'], {k: round(v, 3) for k, v in e['kernel_ms_per_step'].items()}, e['workload'][:40]))
" || exit 1
done
