#!/bin/bash
# A/B on the bench's localization shards; usage: bash tools/lp_ab.sh name[:ENV=VAL...] ...
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  IFS=: read -ra parts <<< "$spec"
  echo "== ${parts[0]}"
  env "${parts[@]:1}" BENCH_SECONDARY=${SEC:-loc16384} timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 2 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln)
        print('   headline %.3f ms/step, weights kernel %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_avg_ms']))
        for e in d.get('secondary', []):
            print('   %.3f ms/step  kernel %.3f ms  %.1f G pb/s  %s' % (e['ms_per_step'], e['roofline']['kernel_avg_ms'], e['value'] / 1e9, e['workload'][:60]))
" || exit 1
done
