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


def timed_region(db, steps=20):
    """What bench.py's HIP events bracket: per kernel the average over the LAST `steps` launches (the
    timed steps; rocprofv3's own averages include the warm-up steps, whose wide cloud makes them slower),
    and the weighting pass as one span, from the start of k_order_keys to the end of k_sum_lw."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    by = {}
    for n, a, b in rows:
        by.setdefault(n, []).append((a, b))
    out = ["%-90s %8s %12s" % (f"kernel (last {steps} steps)", "launches", "avg_ns")]
    per_step = {n: len(v) // max(len(by.get(next((k for k in by if "k_sum_lw" in k), n), [])), 1) for n, v in by.items()}
    for n, v in sorted(by.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
        k = max(per_step.get(n, 1), 1) * steps
        if len(v) < k:
            continue
        tail = v[-k:]
        out.append("%-90s %8d %12.0f" % (n[:90], len(tail), sum(b - a for a, b in tail) / len(tail)))
    keys = [v for n, v in by.items() if "k_order_keys" in n]
    sums = [v for n, v in by.items() if "k_sum_lw" in n]
    if keys and sums and len(keys[0]) == len(sums[0]) >= steps:
        spans = [e[1] - s[0] for s, e in zip(keys[0][-steps:], sums[0][-steps:])]
        out.append("weighting pass, start of k_order_keys to end of k_sum_lw (= roofline.kernel_avg_ms of bench.py): "
                   "avg %.0f ns over %d steps, min %d, max %d" % (sum(spans) / len(spans), len(spans), min(spans), max(spans)))
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


def weights_traffic(base):
    """HBM bytes per launch of the weighting pass from the FETCH_SIZE / WRITE_SIZE passes (KB per
    dispatch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
    import json

    def avg(db, counter, like):
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        kcol = "kernel_name" if "kernel_name" in cols else "name"
        r = cur.execute(f"select avg(value) from counters_collection where counter_name=? and {kcol} like ?",
                        (counter, like)).fetchone()
        return float(r[0]) if r and r[0] is not None else None

    f = glob.glob(os.path.join(base, "pmc_fetch", "*_results.db"))
    w = glob.glob(os.path.join(base, "pmc_l2", "*_results.db"))
    if not f or not w:
        return None
    out = {"source": "profiles/rocprof_%s.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                     "bench.py --steps 10 --warmup 3 --no-cpu-baseline, BENCH_SECONDARY=slam125k)" % os.path.basename(base).replace("prof_", ""),
           "units": "FETCH_SIZE and WRITE_SIZE are KB per dispatch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes "
                    "for gfx950 (it tallies 128-B requests at 64 B); byte-gather reads of k_rays_pl are an uncalibrated "
                    "access width",
           "kernels": {}}
    total = 0.0
    for k in ("k_rays_pl", "k_sum_lw"):
        fk, wk = avg(f[0], "FETCH_SIZE", f"%{k}%"), avg(w[0], "WRITE_SIZE", f"%{k}%")
        if fk is None or wk is None:
            return None
        b = 2.0 * fk * 1024.0 + wk * 1024.0
        out["kernels"][k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes": b}
        total += b
    out["hbm_bytes_per_launch"] = total
    # the grid-SLAM walk kernel of the secondary entry (not part of the headline figure)
    fk, wk = avg(f[0], "FETCH_SIZE", "%k_weights_S2%"), avg(w[0], "WRITE_SIZE", "%k_weights_S2%")
    if fk is not None and wk is not None:
        out["kernels"]["k_weights_S2 (125 000-particle shard)"] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                                                                   "hbm_bytes": 2.0 * fk * 1024.0 + wk * 1024.0}
    # bench.py quotes hbm_bytes_per_launch only while the kernel sources are the ones measured here
    import hashlib
    h = hashlib.sha256()
    d = os.path.join("slam_amd", "csrc")
    for name in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, name), "rb").read())
    out["source_fingerprint"] = h.hexdigest()[:16]
    json.dump(out, open(os.path.join("profiles", "pmc_weights_kernel.json"), "w"), indent=1)
    return out


def main(tag):
    base = os.path.join("gpurun_out", f"prof_{tag}")
    lines = [f"# rocprofv3 summary, tag {tag} (command: see profiles/run_profile.sh)"]
    for d in sorted(glob.glob(os.path.join(base, "*", "*_results.db"))):
        name = os.path.basename(os.path.dirname(d))
        lines.append(f"\n## {name}")
        try:
            lines.append(kernel_stats(d) if name == "trace" else pmc_stats(d))
            if name == "trace":
                lines.append("\n## timed region of the trace run (bench.py defaults: 5 warm-up + 20 timed steps)")
                lines.append(timed_region(d))
        except Exception as e:  # noqa: BLE001
            lines.append(f"(could not read: {e})")
    text = "\n".join(lines) + "\n"
    open(os.path.join("profiles", f"rocprof_{tag}.txt"), "w").write(text)
    print(text)
    print(weights_traffic(base))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
