#!/bin/bash
# kernel trace of the headline bench only: the timed region's kernels, per-step launch list; usage: bash tools/trace_headline.sh TAG [ENV=VAL ...]
TAG=${1:-hl}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o hl -- python3 $REPO/bench.py --no-secondary --no-cpu-baseline > "$OUT/trace.log" 2>&1 || { tail -5 $OUT/trace.log; exit 2; }
grep '^{' $OUT/trace.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('ms_per_step', d['ms_per_step'], 'host_enqueue', d.get('host_enqueue_ms_per_step'), 'kernel_avg_ms', d['roofline'].get('kernel_avg_ms'))
"
cd $REPO && python3 - "$OUT" <<'PY'
import glob, sys, sqlite3
sys.path.insert(0, "profiles")
import summarize
for d in glob.glob(sys.argv[1] + "/trace/*_results.db"):
    print(summarize.timed_region(d))
    cur = sqlite3.connect(d).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    # one step in the middle of the timed region: from one k_predict to the next
    pred = [i for i, r in enumerate(rows) if "k_predict" in r[0]]
    a, b = pred[len(pred) // 2 - 10], pred[len(pred) // 2 - 9]
    t0 = rows[a][1]
    print("one step, launch by launch (start us, duration us, gap before us):")
    prev = None
    for n, s, e in rows[a:b]:
        print("  %8.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, n[:100]))
        prev = e
    print("  launches in the step:", b - a, " step period us:", (rows[b][1] - t0) / 1e3)
PY
rm -rf $OUT/trace
