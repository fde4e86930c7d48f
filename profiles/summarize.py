"""Summarises rocprofv3 rocpd databases (gpurun_out/prof_<tag>/*) into text files under profiles/.
Usage: python profiles/summarize.py <tag>"""
import glob
import os
import sqlite3
import sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["%-90s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for n, c, t, a, mn, mx in rows:
        out.append("%-90s %8d %14d %12.0f %12d %12d %6.2f%%" % (n[:90], c, t, a, mn, mx, 100.0 * t / tot))
    return "\n".join(out)


def pmc_stats(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = cur.execute(f"select {kcol}, counter_name, count(*), avg(value), sum(value) from counters_collection "
                       f"group by {kcol}, counter_name order by {kcol}, counter_name").fetchall()
    out = ["%-70s %-24s %8s %18s %20s" % ("kernel", "counter", "launches", "avg_per_launch", "sum")]
    for k, c, n, a, s in rows:
        out.append("%-70s %-24s %8d %18.1f %20.1f" % (k[:70], c, n, a, s))
    return "\n".join(out)


def main(tag):
    base = os.path.join("gpurun_out", f"prof_{tag}")
    lines = [f"# rocprofv3 summary, tag {tag} (command: see profiles/run_profile.sh)"]
    for d in sorted(glob.glob(os.path.join(base, "*", "*_results.db"))):
        name = os.path.basename(os.path.dirname(d))
        lines.append(f"\n## {name}")
        try:
            lines.append(kernel_stats(d) if name == "trace" else pmc_stats(d))
        except Exception as e:  # noqa: BLE001
            lines.append(f"(could not read: {e})")
    text = "\n".join(lines) + "\n"
    open(os.path.join("profiles", f"rocprof_{tag}.txt"), "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
