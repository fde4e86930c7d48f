#!/bin/bash
# usage: bash tools/defer_ab.sh name[:ENV=VAL...] ...
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  IFS=: read -ra parts <<< "$spec"
  echo "== ${parts[0]}"
  env "${parts[@]:1}" BENCH_SECONDARY=${SEC:-slam125k} timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 2 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        for e in d.get('secondary', []):
            print('   %.3f ms/step  %s' % (e['ms_per_step'], e['workload'][:40]))
" || exit 1
done
