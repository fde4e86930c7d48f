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


def timed_region(db, steps=20, after=23):
    """What bench.py's timed region holds: per kernel the average over the `steps` launches per step-kernel that come
    before the last `after` steps (bench.py runs the timed steps, then 20 more with every group bracketed by events and 3
    for the host's own cost; rocprofv3's own averages also include the warm-up steps, whose wide cloud makes them slower), and the
    weighting pass as one span, from the start of k_order_keys to the end of k_sum_lw."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    by = {}
    for n, a, b in rows:
        by.setdefault(n, []).append((a, b))
    out = ["%-90s %8s %12s" % (f"kernel (the {steps} timed steps)", "launches", "avg_ns")]
    n_steps = len(next((v for k, v in by.items() if "k_sum_lw" in k), [])) or 1
    for n, v in sorted(by.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
        per = len(v) // n_steps
        if per < 1 or len(v) < per * (steps + after):
            continue
        sel = v[len(v) - per * (steps + after):len(v) - per * after]
        out.append("%-90s %8d %12.0f" % (n[:90], len(sel), sum(b - a for a, b in sel) / len(sel)))
    keys = [v for n, v in by.items() if "k_order_keys" in n]
    sums = [v for n, v in by.items() if "k_sum_lw" in n]
    if keys and sums and len(keys[0]) == len(sums[0]) >= steps + after:
        k0, s0 = keys[0][-(steps + after):-after], sums[0][-(steps + after):-after]
        spans = [e[1] - s[0] for s, e in zip(k0, s0)]
        out.append("weighting pass, start of k_order_keys to end of k_sum_lw (= roofline.kernel_avg_ms of bench.py): "
                   "avg %.0f ns over %d steps, min %d, max %d" % (sum(spans) / len(spans), len(spans), min(spans), max(spans)))
        first, last = keys[0][-(steps + after)][0], sums[0][-after - 1][1]
        busy = sum(b - a for n, v in by.items() for a, b in v if a >= first and b <= last)
        out.append("timed region on the device: %.1f us per step from the first k_order_keys to the last k_sum_lw, kernels busy %.1f us "
                   "per step" % ((last - first) / steps / 1e3, busy / steps / 1e3))
    return "\n".join(out)


def slam_summary(base, tag):
    """profiles/rocprof_<tag>_slam.txt: the grid-SLAM kernels of the 125 000-particle entry (the launches of
    k_weights_S2 with the largest grid, and every tile kernel between the first and the last of them)."""
    lines = [f"# rocprofv3 summary of the grid-SLAM kernels, tag {tag} (profiles/run_profile.sh: trace pass 2 = bench.py --no-cpu-baseline with all",
             "# secondary entries; counter passes = bench.py --steps 5 --warmup 2 --no-cpu-baseline with BENCH_SECONDARY=slam125k).",
             "# 125 000 particles x 90 beams, 2354x2464 canvases as copy-on-write tiles, maze scans (bench.py slam_entry)."]
    t = glob.glob(os.path.join(base, "trace_sec", "*_results.db"))
    if t:
        cur = sqlite3.connect(t[0]).cursor()
        big = cur.execute("select max(grid_x) from kernels where name like '%k_weights_S2%'").fetchone()[0]
        w = cur.execute("select min(start), max(end), count(*) from kernels where name like '%k_weights_S2%' and grid_x=?", (big,)).fetchone()
        lines.append(f"\n## kernel trace: the {w[2]} launches of k_weights_S2* with grid {big} threads and the tile kernels around them")
        lines.append("%-84s %8s %12s %12s %12s" % ("kernel", "launches", "avg_ns", "min_ns", "max_ns"))
        rows = cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels where start>=? and end<=? "
                           "and (name like '%k_weights_S2%' or name like '%k_tile_%' or name like '%k_tab_gather%' or name like '%k_set_sectors%' "
                           "or name like '%k_defer_split%' or name like '%k_mark_sources%' or name like '%k_list_sources%') "
                           "and (name not like '%k_weights_S2%' or grid_x=?) group by name order by 3 desc", (w[0] - 40_000_000, w[1] + 5_000_000, big)).fetchall()
        for n, c, a, mn, mx in rows:
            lines.append("%-84s %8d %12.0f %12d %12d" % (n[:84], c, a, mn, mx))
    lines.append("\n## counters of the grid-SLAM kernels per launch, by kernel (every launch has the grid of 125 000 wavefronts; the replay and")
    lines.append("## the edge pass of k_weights_S2<true, false, false> leave most of them at once), separate passes")
    vals = {}
    for sub in ("pmc_sq", "pmc_fetch", "pmc_l2", "pmc_grbm"):
        for d in glob.glob(os.path.join(base, sub, "*_results.db")):
            cur = sqlite3.connect(d).cursor()
            for kn, c, n, a, mx in cur.execute("select kernel_name, counter_name, count(*), avg(value), max(value) from counters_collection "
                                               "where kernel_name like '%k_weights_S2%' group by kernel_name, counter_name order by kernel_name, counter_name").fetchall():
                short = kn.split("(")[0].replace("void slamk::", "").replace("slamk::", "")
                lines.append("%-40s %-20s launches %4d  avg per launch %18.1f  max %18.1f" % (short, c, n, a, mx))
                if "k_weights_S2F" in kn:
                    vals[c] = a
    if "SQ_WAVES" in vals and "SQ_INSTS_VALU" in vals:
        rays = vals["SQ_WAVES"] * 90.0
        lines.append("   -> k_weights_S2F: %.0f VALU and %.0f SALU wavefront instructions per ray (%.2f M rays per launch); round 2's k_weights_S2<tiled>: 392 + 398"
                     % (vals["SQ_INSTS_VALU"] / rays, vals.get("SQ_INSTS_SALU", 0.0) / rays, rays / 1e6))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        lines.append("   -> k_weights_S2F: HBM traffic per launch %.2f GB read (FETCH_SIZE KB doubled for gfx950) + %.2f GB written"
                     % (2 * vals["FETCH_SIZE"] * 1024 / 1e9, vals["WRITE_SIZE"] * 1024 / 1e9))
    text = "\n".join(lines) + "\n"
    open(os.path.join("profiles", f"rocprof_{tag}_slam.txt"), "w").write(text)
    return text


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
                     "bench.py --steps 5 --warmup 2 --no-cpu-baseline, BENCH_SECONDARY=slam125k)" % os.path.basename(base).replace("prof_", ""),
           "units": "FETCH_SIZE and WRITE_SIZE are KB per dispatch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes "
                    "for gfx950 (it tallies 128-B requests at 64 B); byte-gather reads of k_rays_pl are an uncalibrated "
                    "access width",
           "kernels": {}}
    total = 0.0
    for k in ("k_rays_pl", "k_sum_lw"):
        like = "%k_rays_pl<5, false, false>%" if k == "k_rays_pl" else f"%{k}%"  # (not the slab-less variant of the large shards)
        fk, wk = avg(f[0], "FETCH_SIZE", like), avg(w[0], "WRITE_SIZE", like)
        if fk is None or wk is None:
            return None
        b = 2.0 * fk * 1024.0 + wk * 1024.0
        out["kernels"][k] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes": b}
        total += b
    out["hbm_bytes_per_launch"] = total
    # the grid-SLAM kernels of the secondary entry (not part of the headline figure).  Since round 3 an update weighs the
    # particles with k_weights_S2F and leaves the canvases to the resample, which runs the ordinary kernel
    # k_weights_S2<true, false, false> on the selected particles (the "replay"; one more launch of it per update serves the
    # particles too close to row 0 / column 0, usually none).  Per deferred update: the mean over the k_weights_S2F launches,
    # and the sum over the other kernel's launches without its full passes (the first update or two of a filter run it on
    # every particle: values above 30 % of the largest), divided by the number of k_weights_S2F launches.
    def per_update(db, counter):
        cur = sqlite3.connect(db).cursor()
        f = [r[0] for r in cur.execute("select value from counters_collection where counter_name=? and kernel_name like '%k_weights_S2F%'", (counter,))]
        o = [r[0] for r in cur.execute("select value from counters_collection where counter_name=? and kernel_name like '%k_weights_S2<true, false, false>%'", (counter,))]
        if not f:
            return None, None
        part = [v for v in o if v <= 0.3 * max(o)] if o else []
        return sum(f) / len(f), sum(part) / len(f)
    (ff, fr), (wf, wr) = per_update(f[0], "FETCH_SIZE"), per_update(w[0], "WRITE_SIZE")
    if ff is not None and wf is not None:
        out["kernels"]["k_weights_S2F (125 000-particle shard)"] = {"FETCH_SIZE_KB": ff, "WRITE_SIZE_KB": wf, "hbm_bytes": 2.0 * ff * 1024.0 + wf * 1024.0}
        out["kernels"]["k_weights_S2 on the survivors (125 000-particle shard, per deferred update)"] = {
            "FETCH_SIZE_KB": fr, "WRITE_SIZE_KB": wr, "hbm_bytes": 2.0 * fr * 1024.0 + wr * 1024.0}
    # the weighting kernel of the 16384^2 shard (12.5 M particles; its sparse cloud gets the wave-per-particle kernel
    # k_weights_L<0> from the second pass on): the launches with the largest grid
    def avg_big(db, counter, like):
        cur = sqlite3.connect(db).cursor()
        big = cur.execute("select max(grid_size) from counters_collection where kernel_name like ?", (like,)).fetchone()[0]
        if not big:
            return None, None
        r = cur.execute("select avg(value) from counters_collection where counter_name=? and kernel_name like ? and grid_size=?",
                        (counter, like, big)).fetchone()
        return (float(r[0]) if r and r[0] is not None else None), big
    (fk, big), (wk, _) = avg_big(f[0], "FETCH_SIZE", "%k_weights_L<0,%"), avg_big(w[0], "WRITE_SIZE", "%k_weights_L<0,%")
    if fk is not None and wk is not None and big and big > 100_000_000:
        out["kernels"]["weighting kernel of the 16384^2 shard (k_weights_L<0>, 12.5 M particles)"] = {
            "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes": 2.0 * fk * 1024.0 + wk * 1024.0}
    # instruction issue (round 3): wavefront-level VALU / SALU instruction counts and wavefronts per launch of the same kernels,
    # from the SQ pass; bench.py turns them into an issue-slot fraction with the kernel's measured duration
    sq = glob.glob(os.path.join(base, "pmc_sq", "*_results.db"))
    if sq:
        def sq_of(like, big_only=False):
            cur = sqlite3.connect(sq[0]).cursor()
            cond, args = "kernel_name like ?", [like]
            if big_only:
                big = cur.execute("select max(grid_size) from counters_collection where kernel_name like ?", (like,)).fetchone()[0]
                cond, args = cond + " and grid_size=?", args + [big]
            return {c: float(a) for c, a in cur.execute(f"select counter_name, avg(value) from counters_collection where {cond} group by counter_name", args)}
        SQ_KEYS = ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")
        for name, like, big in (("k_rays_pl", "%k_rays_pl<5, false, false>%", False), ("k_sum_lw", "%k_sum_lw%", False),
                                ("weighting kernel of the 16384^2 shard (k_weights_L<0>, 12.5 M particles)", "%k_weights_L<0,%", True)):
            v = sq_of(like, big)
            if v and name in out["kernels"]:
                out["kernels"][name]["sq"] = {k: v[k] for k in SQ_KEYS if k in v}
        nf, nr = "k_weights_S2F (125 000-particle shard)", "k_weights_S2 on the survivors (125 000-particle shard, per deferred update)"
        if nf in out["kernels"]:
            sf, sr = {}, {}
            for k in SQ_KEYS:
                a, b = per_update(sq[0], k)
                if a is not None:
                    sf[k], sr[k] = a, b
            out["kernels"][nf]["sq"], out["kernels"][nr]["sq"] = sf, sr
    # the kernels' own durations in the trace runs (the timed steps of the headline run; the largest-grid launches of the
    # secondary run), so that bench.py can state how busy each kernel keeps the vector ALUs by itself, beside the figure for
    # the whole pass that it measures live
    def trace_ns(db, like, steps=20, after=23, big_only=False):
        cur = sqlite3.connect(db).cursor()
        v = cur.execute("select end - start, grid_size from kernels where name like ? order by start", (like,)).fetchall() \
            if "grid_size" in [r[1] for r in cur.execute("pragma table_info(kernels)")] else \
            [(r[0], 0) for r in cur.execute("select end - start from kernels where name like ? order by start", (like,))]
        if not v:
            return None
        if big_only:
            big = max(g for _, g in v)
            d = [t for t, g in v if g == big]
        else:
            d = [t for t, _ in v]
            if len(d) >= steps + after:
                d = d[-(steps + after):-after]
        return sum(d) / len(d)
    tr = glob.glob(os.path.join(base, "trace", "*_results.db"))
    ts = glob.glob(os.path.join(base, "trace_sec", "*_results.db"))
    if tr:
        for name, like in (("k_rays_pl", "%k_rays_pl<5, false, false>%"), ("k_sum_lw", "%k_sum_lw%")):
            t = trace_ns(tr[0], like)
            if t and name in out["kernels"]:
                out["kernels"][name]["trace_avg_ns"] = t
    if ts:
        for name, like in (("weighting kernel of the 16384^2 shard (k_weights_L<0>, 12.5 M particles)", "%k_weights_L<0,%"),
                           ("k_weights_S2F (125 000-particle shard)", "%k_weights_S2F%")):
            t = trace_ns(ts[0], like, big_only=True)
            if t and name in out["kernels"]:
                out["kernels"][name]["trace_avg_ns"] = t
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
            lines.append(kernel_stats(d) if name.startswith("trace") else pmc_stats(d))
            if name == "trace":
                lines.append("\n## timed region of the trace run (bench.py defaults: 5 warm-up + 20 timed steps, then 20 steps bracketed by events and 3 more)")
                lines.append(timed_region(d))
        except Exception as e:  # noqa: BLE001
            lines.append(f"(could not read: {e})")
    text = "\n".join(lines) + "\n"
    open(os.path.join("profiles", f"rocprof_{tag}.txt"), "w").write(text)
    print(text)
    print(weights_traffic(base))
    print(slam_summary(base, tag))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
