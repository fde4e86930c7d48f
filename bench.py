"""bench.py — particle-filter step throughput on MI355X (BASELINE.json metric).

One "step" = MCL::predict + MCL::update (weights from raycast + beam likelihood, running sum,
multinomial selection, sort/gather) over the resident particle set, as in the reference's timed
loop (benchmark/mcl.cpp:58-64), with its odometry and alphas.  Unlike that loop, which re-applies
one stale scan while the robot turns 2.5 rad per step, the scan is refreshed from the (noise-free)
true pose every step as apps/grid_slam.cpp:142-168 does, so the filter tracks instead of collapsing.

Workload at N = 1: BASELINE.json configs[1] — localization on assets/floor_plan.png (1297 x 599),
100k particles x 360 beams, max range 500, sigma 5, scanner offset {0,30,0}.  N > 1 is weak scaling:
100k particles per GPU, one process per GPU, global resampling through one RCCL all-gather per step.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ODOM = [2.5, 0.02, 0.02]            # benchmark/mcl.cpp:60
ALPHAS = [5e-4, 5e-4, 1e-2, 1e-2]   # benchmark/mcl.cpp:61
OFFSET = [0.0, 30.0, 0.0]           # apps/grid_slam.cpp:99
STDDEV, MAX_DIST = 5.0, 500.0       # apps/grid_slam.cpp:88-89
TRUE_POSE = [520.0, 280.0, 0.4]
SEED = 0x5EED0001
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def load_free_grid(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", "maps.npz"))
    sh = z[name + "_shape"]
    return np.unpackbits(z[name + "_bits"])[: int(sh[0]) * int(sh[1])].reshape(int(sh[0]), int(sh[1]))


def true_trajectory(n_steps):
    """Noise-free robot motion, as benchmark/mcl.cpp:62 moves real_position (default alphas = 0):
    pose k is where the robot is when scan k is taken, i.e. after k+1 odometry steps."""
    x, y, th = TRUE_POSE
    out = []
    for _ in range(n_steps):
        r1, t, r2 = ODOM
        x, y = x + t * np.cos(th + r1), y + t * np.sin(th + r1)
        th = np.fmod(th + r1 + r2, 2 * np.pi)
        if th > np.pi:
            th -= 2 * np.pi
        out.append((x, y, th))
    return out


def synth_scans(sa, free, device, n_beams, poses):
    """apps/grid_slam.cpp:144-166 with the device FakeLidar: one scan per robot pose ->
    list of (angles, dists).  The scan is refreshed every step, as the app does (the reference's
    own benchmark re-applies one stale scan while the robot turns 2.5 rad per step)."""
    from slam_amd import capi

    lib = capi.load_library()
    sensors = [sa.MCL.sensor_position(sa.Pose(*p), sa.Pose(*OFFSET)) for p in poses]
    sp = np.array([[s.x, s.y, s.theta] for s in sensors])
    hits = np.empty((len(poses), n_beams, 2))
    m01 = np.ascontiguousarray(free, dtype=np.uint8)
    capi.check(lib.slam_fake_lidar_scan(device, capi.ptr(m01), m01.shape[0], m01.shape[1], capi.ptr(sp), len(poses), 0.0,
                                        2 * np.pi, MAX_DIST, n_beams, capi.ptr(hits)))
    step = 2 * np.pi / n_beams
    ang = np.arange(n_beams) * step - np.pi
    scans = []
    for k in range(len(poses)):
        d = np.sqrt((hits[k, :, 1] - sp[k, 1]) ** 2 + (hits[k, :, 0] - sp[k, 0]) ** 2)
        scans.append((ang, np.where(hits[k, :, 0] != -1, d, MAX_DIST)))
    return scans


def synth_scan(sa, free, device, n_beams):
    return synth_scans(sa, free, device, n_beams, [TRUE_POSE])[0]


def cpu_baseline(free, ang, dist, n_beams, budget_s=12.0):
    """The CPU oracle (port of the reference's update loop, threaded like slam/mcl.cpp:120-137) timed on
    a bounded sample of the same workload.  Checker code, called from here only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc

    cores = min(16, len(os.sched_getaffinity(0)))
    g = (free * 255).astype(np.uint8)
    rng = np.random.default_rng(0)

    def run(n):
        x = TRUE_POSE[0] + 25 * rng.standard_normal(n)
        y = TRUE_POSE[1] + 25 * rng.standard_normal(n)
        th = TRUE_POSE[2] + 0.2 * rng.standard_normal(n)
        z = orc.philox_normals(SEED, 0, 0, n)
        t0 = time.perf_counter()
        x, y, th = orc.predict(x, y, th, ODOM, ALPHAS, z)
        w, _ = orc.update_weights(x, y, th, g, ang, dist, STDDEV, MAX_DIST, OFFSET, n_threads=cores, fast=True)
        idx, _, _ = orc.resample(w, orc.philox_uniforms(SEED, 0, 0, n), orc.ORDER_REFERENCE)
        orc.sort_order(w[idx], 1)
        return time.perf_counter() - t0

    n0 = 64 * cores
    t = run(n0)
    n = int(max(n0, min(400000, n0 * budget_s / max(t, 1e-3))))
    t = run(n)
    return {"value": n * n_beams / t, "unit": "particle-beams/s", "cores": cores, "kind": "port",
            "sample": f"{n} particles x {n_beams} beams of the same workload, one full step (predict+update), "
                      f"{t:.2f} s, oracle built with the reference's flags (-Ofast), {cores} threads in the "
                      f"reference's slicing"}


def source_fingerprint():
    """Hash of the kernel sources: a recorded counter value is only quoted while the code it was taken from is unchanged."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "slam_amd", "csrc")
    for name in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


HBM_MEASURED_GBS = 6290.0   # MI355X_MICROARCH.md: float4 copy, 79 % of the spec
N_SIMD, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs; peak engine clock


def physical_roofline(kernel_names, kernel_ms, f64_key):
    """What the kernel does to the machine, from the committed counter passes (profiles/pmc_weights_kernel.json, taken by
    profiles/run_profile.sh with rocprofv3 --pmc in runs of their own) and the kernel's duration measured live in THIS run.
    Quoted only while the kernel sources are the ones the counters were taken from (fingerprint), else None.
      hbm: measured bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) / duration, as a fraction of 8 TB/s and of 6.29 TB/s;
      valu: wavefront VALU instructions per launch x 4 cycles (a 16-lane SIMD issues one wavefront instruction per 4 cycles), and
            weighed with the static share of 64-bit instructions at their measured rates (profiles/valu_f64_share.json,
            profiles/ubench_r03.txt), over the cycles of 1024 SIMDs in that duration."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_weights_kernel.json")))
        fp = source_fingerprint()
        if rec.get("source_fingerprint") != fp:
            return None
        ks = [rec["kernels"][k] for k in kernel_names]
        traffic = sum(k["hbm_bytes"] for k in ks)
        out = {"hbm_bytes_per_launch": traffic, "hbm_gbs": traffic / (kernel_ms * 1e-3) / 1e9}
        out["hbm_measured_frac"] = out["hbm_gbs"] / HBM_PEAK_GBS
        out["hbm_frac_of_measured_copy_rate"] = out["hbm_gbs"] / HBM_MEASURED_GBS
        if all("sq" in k for k in ks):
            valu = sum(k["sq"]["SQ_INSTS_VALU"] for k in ks)
            salu = sum(k["sq"]["SQ_INSTS_SALU"] for k in ks)
            share = None
            try:
                sh = json.load(open(os.path.join(ROOT, "profiles", "valu_f64_share.json")))
                if sh.get("source_fingerprint") == fp:
                    share = sh["kernels"][f64_key]["f64_share"]
            except Exception:
                share = None
            slots = N_SIMD * CLOCK_HZ * kernel_ms * 1e-3
            v = {"wave_valu_insts_per_launch": valu, "wave_salu_insts_per_launch": salu, "simd_cycles_available": slots,
                 "f64_share_static": share, "valu_insts_per_cycle_per_simd": valu / slots}
            # a wavefront VALU instruction holds its 16-lane SIMD for 4 cycles; measured on this chip (profiles/ubench_r03.txt, 4 wavefronts
            # per SIMD, 2.4 GHz nominal): f64 add/fma/mul/cvt 4.6-4.9 cycles, 32-bit integer and float 3.2-3.5 (NOT two slots per f64
            # instruction, as rounds 1-2 assumed)
            v["valu_busy_frac"] = valu * 4.0 / slots
            if share is not None:
                v["valu_busy_frac_measured_rates"] = valu * (share * 4.9 + (1.0 - share) * 3.4) / slots
            v["note"] = ("valu_busy_frac = VALU wavefront instructions x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time): the share of the kernel's "
                         "duration in which the vector ALUs are issuing; ..._measured_rates weighs the static 64-bit share with 4.9 cycles and the rest "
                         "with 3.4 (profiles/ubench_r03.txt); SALU and memory instructions issue beside them; the clock under load may be below "
                         "the nominal 2.4 GHz, which makes both fractions lower bounds")
            out["valu"] = v
            # each kernel by itself, from the committed profile alone (its instruction count over its own duration in the trace run
            # of the same sources): the live figure above divides by the time of the whole group, launch-bound helpers included
            per = {}
            for name, k in zip(kernel_names, ks):
                if "trace_avg_ns" in k and "sq" in k:
                    per[name] = {"avg_ms_in_the_committed_trace": k["trace_avg_ns"] * 1e-6,
                                 "valu_busy_frac": k["sq"]["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * CLOCK_HZ * k["trace_avg_ns"] * 1e-9)}
            if per:
                v["per_kernel_from_the_committed_profile"] = per
        return out
    except Exception:
        return None


def algorithmic_frac(achieved):
    """SURVEY.md §8d's figure as a fraction of the 8 TB/s roofline; above 1 it is no fraction of a roofline (the kernel does not
    read those bytes: the clearance search skips them) and is reported as a ratio of its own."""
    f = achieved / HBM_PEAK_GBS
    return (f, None) if f <= 1.0 else (None, f)


def slam_scans(sa, free, device, n_beams, true0, n_steps):
    """apps/grid_slam.cpp:144-166: one scan per robot pose while the robot drives straight on (UP key: 2.5 per frame)."""
    poses, (x, y, th) = [], true0
    for _ in range(n_steps):
        x, y = x + 2.5 * np.cos(th), y + 2.5 * np.sin(th)
        poses.append((x, y, th))
    return synth_scans(sa, free, device, n_beams, poses)


def slam_cpu_baseline(scans, rows, cols, n_beams, n=None):
    """CPU oracle in SLAM mode (walk + occupancy update, threaded like slam/mcl.cpp:120-137; resample with the deep
    copy of slam/mcl.cpp:147-155) on a bounded number of particles: the mean of its last three steps (>= 1 s of work)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc

    cores = min(16, len(os.sched_getaffinity(0)))
    n = n or max(cores, min(4096, int(4.0e9 // (rows * cols))))  # canvases of the sample within ~4 GB (twice that during the copy)
    maps = np.full((n, rows, cols), 128, np.uint8)
    xs, ys, ths = np.full(n, cols / 2.0), np.full(n, rows / 2.0), np.full(n, np.pi / 2)
    odom, alphas = [0.0, 2.5, 0.0], [1e-3] * 4
    steps = min(5, len(scans))
    times = []
    for k in range(steps):
        xs, ys, ths = orc.predict(xs, ys, ths, odom, alphas, orc.philox_normals(1, k, 0, n))
        t0 = time.perf_counter()
        w, _ = orc.update_weights(xs, ys, ths, maps, *scans[k], STDDEV, MAX_DIST, OFFSET, slam_mode=True, n_threads=cores, fast=True)
        idx, _, _ = orc.resample(w, orc.philox_uniforms(1, k, 0, n))
        src = idx[orc.sort_order(w[idx], 1)]
        maps = maps[src]
        xs, ys, ths = xs[src], ys[src], ths[src]
        times.append(time.perf_counter() - t0)
    last = times[-3:]
    dt = sum(last) / len(last)
    return {"value": n * n_beams / dt, "unit": "particle-beams/s", "cores": cores, "kind": "port",
            "sample": f"{n} particles x {n_beams} beams with {rows}x{cols} canvases, mean of steps {steps - len(last) + 1}..{steps} of the same scans "
                      f"(update + resample with deep copy), {dt:.2f} s per step, {sum(last):.2f} s in all, oracle built -Ofast, {cores} threads"}


def slam_entry(sa, capi, device, label, n, rows, cols, free, true0, pool, steps, warmup, with_cpu, reps=1):
    """Grid-SLAM mode (exactly slam::MCL: per-particle canvases updated along every ray), 90 beams, the app's
    sensor (apps/grid_slam.cpp:88-99) and motion noise (:132); canvases as copy-on-write tiles."""
    import torch

    B = 90
    extra = 3
    scans = slam_scans(sa, free, device, B, true0, warmup + steps + extra)
    odom, alphas = [0.0, 2.5, 0.0], [1e-3] * 4
    # a short entry (steps x < 1 ms) is timed on `reps` filters, each from the same start over the same scans, and the median
    # taken: on some fresh boxes the first pass of a process whose GPU sat idle through the CPU baseline ran at 2-3 ms per step
    # (host and device alike) and the next at 0.54
    runs = []
    for rep in range(reps):
        f = sa.Filter(n, rows, cols, mode=capi.MODE_SLAM, device=device, seed=SEED, map_storage=capi.MAPS_TILED, tile_pool_tiles=pool,
                      async_launch=True)
        k = 0
        f.collect_stats(False)  # the cell counters are instrumentation: off in the timed steps, on where the bytes are counted
        for _ in range(warmup):
            f.predict(odom, alphas); f.update(*scans[k], STDDEV, MAX_DIST, OFFSET); k += 1
        f.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            f.predict(odom, alphas); f.update(*scans[k], STDDEV, MAX_DIST, OFFSET); k += 1
        host_ms = (time.perf_counter() - t0) / steps * 1e3  # time to ENQUEUE a step
        f.synchronize()
        runs.append(((time.perf_counter() - t0) / steps * 1e3, host_ms))
        if rep + 1 < reps:
            f.close()
    ms_step, host_ms = sorted(runs)[len(runs) // 2]
    # kernel time of the update as the timed steps run it (since round 3: weights from k_weights_S2F, canvases of the selected
    # particles drawn inside the resample), every kernel group bracketed by events, over the next steps
    f.profile(True)
    f.kernel_ms()
    extra_a = extra // 2 + 1
    for _ in range(extra_a):
        f.predict(odom, alphas); f.update(*scans[k], STDDEV, MAX_DIST, OFFSET); k += 1
    f.synchronize()
    kms = f.kernel_ms()
    f.profile(False)
    defer = f.defer_stats()
    # algorithmic bytes (SURVEY.md §8d, S mode): what the reference's update reads and writes, counted by the ordinary pass
    # (every canvas drawn) on the cloud as it is now, one pass at a time (the counters of a pass are read after it)
    f.collect_stats(True)
    alg = 0.0
    reads = writes = 0.0
    extra_b = extra - extra_a if extra - extra_a > 0 else 1
    for _ in range(extra_b):
        f.predict(odom, alphas); f.update_weights(*scans[k % len(scans)], STDDEV, MAX_DIST, OFFSET); k += 1
        st = f.stats()
        reads += st["cells_read"]; writes += st["cells_written"]
        alg += st["cells_read"] + st["cells_written"] + 40.0 * n + 16.0 * B   # SURVEY.md §8d, S mode
        f.resample()
    f.synchronize()
    tiles = f.tile_stats()
    f.close()
    torch.cuda.synchronize()
    # the update's kernels: group "weights" (k_weights_S2F, or the ordinary kernel while an update does not defer) plus group
    # "clone" (the ordinary kernel on the selected particles, tile plans and copies, the tile census of the resample)
    w_ms = (kms["weights"][0] + kms["clone"][0]) / extra_a
    achieved = alg / extra_b / (w_ms * 1e-3) / 1e9
    NF, NR = "k_weights_S2F (125 000-particle shard)", "k_weights_S2 on the survivors (125 000-particle shard, per deferred update)"
    traffic = None
    if n == 125_000:  # the counter passes of profiles/run_profile.sh cover this entry; quoted while the sources are unchanged
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_weights_kernel.json")))
            if rec.get("source_fingerprint") == source_fingerprint():
                traffic = rec["kernels"][NF]["hbm_bytes"] + rec["kernels"][NR]["hbm_bytes"]
        except Exception:
            traffic = None
    out = {
        "workload": label, "value": n * B / (ms_step * 1e-3), "unit": "particle-beams/s", "ms_per_step": ms_step, "steps": steps,
        "warmup": warmup, "dtype": "f64 positions / u8 cells",
        "instrumentation": "cell counters (slam_mcl_get_stats) off in the timed steps and in the steps kernel_avg_ms comes from, on in the "
                           "ordinary passes the bytes are counted by",
        "roofline": {"bound": "valu" if defer["deferred"] else "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "physical": physical_roofline([NF, NR], w_ms, "k_weights_S2F") if n == 125_000 else None,
                     "kernel": "the update's kernels: k_weights_S2F (walk + beam likelihood of every particle, canvases untouched) + "
                               "k_weights_S2<tiled> on the particles the resample selects (tile copies, walk, occupancy update) + tile census",
                     "kernel_avg_ms": w_ms, "kernel_groups_ms": {"weights": kms["weights"][0] / extra_a, "clone": kms["clone"][0] / extra_a},
                     "deferred_updates": defer,
                     "algorithmic_bytes_per_launch": alg / extra_b, "cells_read_per_ray": reads / extra_b / (n * B),
                     "cells_written_per_ray": writes / extra_b / (n * B),
                     "note": "bytes = cells the reference's update reads + writes for EVERY particle (counted by the ordinary pass) + 40 B per "
                             "particle (SURVEY.md §8d); the deferred update reads them all but writes only into the canvases that survive "
                             "the resample (2-9 % of the particles here), so its HBM traffic (traffic, physical.hbm_*) is far below the "
                             "round-2 kernel's 58 GB per launch; what binds k_weights_S2F is instruction issue (physical.valu; DESIGN.md §6c)"},
        "kernel_ms_per_step": {g: v[0] / extra_a for g, v in kms.items()},
        "host_enqueue_ms_per_step": host_ms,
        "ms_per_step_runs": [r[0] for r in runs], "ms_per_step_is": "the only pass" if reps == 1 else f"median of {reps} passes of {steps} steps, each on a new filter from the same start",
        "tiles": tiles,
    }
    if with_cpu:
        out["cpu_baseline"] = slam_cpu_baseline(scans, rows, cols, B)
    return out


def localization_shard_entry(sa, capi, device, label, size, n, B, zmax, uniform, steps):
    """One GPU's share of BASELINE configs[2] / configs[4]: localization on the synthetic grids of slam_amd/synth.py,
    log-domain weights (the linear ones underflow beyond ~434 beams), sigma 0.5."""
    import torch

    from slam_amd import synth

    g = synth.make_grid(size)
    free01 = (g >= 128).astype(np.uint8)
    true = synth.open_pose(g)  # (until late in round 2 the 16384^2 sensor stood inside a wall: every beam measured 0.5 cells)
    sensor = sa.MCL.sensor_position(sa.Pose(*true), sa.Pose(*OFFSET))
    gm = sa.GridMap(free01, device=device)
    hits = gm.fake_lidar_scan([sensor.x, sensor.y, sensor.theta], 0.0, 2 * np.pi, zmax, B)[0]
    gm.close()
    ang = np.arange(B) * (2 * np.pi / B) - np.pi
    dist = np.where(hits[:, 0] != -1, np.hypot(hits[:, 1] - sensor.y, hits[:, 0] - sensor.x), zmax)
    f = sa.Filter(n, size, size, device=device, seed=SEED, sort=capi.SORT_NONE, weight_domain=capi.WEIGHT_LOG, async_launch=True,
                  raycaster=int(os.environ.get("BENCH_LOC_RAYCASTER", capi.RAYCAST_FAST)))  # experiments: 2 = wave per particle
    f.set_map(g)
    if uniform:
        f.seed_uniform_free()
    else:
        f.seed_gaussian(true, [25.0, 25.0, 0.2])
    # one weighting pass per step on the seeded cloud (a resample would collapse the uniform cloud: the regime
    # on record here is the wide one); the cells of the reference's walk from a subset on the cell-by-cell kernel
    f.predict(ODOM, ALPHAS)
    f.update_weights(ang, dist, 0.5, zmax, OFFSET)
    f.synchronize()
    f.profile(True)
    f.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(steps):
        f.predict(ODOM, ALPHAS)
        f.update_weights(ang, dist, 0.5, zmax, OFFSET)
    f.synchronize()
    ms_step = (time.perf_counter() - t0) / steps * 1e3
    kms = f.kernel_ms()
    x, y, th = f.poses()
    f.close()
    sub = np.random.default_rng(0).choice(n, 4096, replace=False)
    wk = sa.Filter(4096, size, size, device=device, raycaster=capi.RAYCAST_WALK, sort=capi.SORT_NONE, weight_domain=capi.WEIGHT_LOG)
    wk.set_map(g)
    wk.set_poses(x[sub], y[sub], th[sub])
    wk.update_weights(ang, dist, 0.5, zmax, OFFSET)
    sbar = wk.stats()["cells_read"] / (4096.0 * B)
    wk.close()
    torch.cuda.synchronize()
    w_ms = kms["weights"][0] / max(kms["weights"][1], 1)
    alg = n * B * sbar + 40.0 * n + 16.0 * B
    achieved = alg / (w_ms * 1e-3) / 1e9
    traffic = None
    if size == 16384:  # the counter passes of profiles/run_profile.sh cover this entry; quoted while the sources are unchanged
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_weights_kernel.json")))
            if rec.get("source_fingerprint") == source_fingerprint():
                traffic = rec["kernels"]["weighting kernel of the 16384^2 shard (k_weights_L<0>, 12.5 M particles)"]["hbm_bytes"]
        except Exception:
            traffic = None
    return {
        "workload": label, "value": n * B / (ms_step * 1e-3), "unit": "particle-beams/s", "ms_per_step": ms_step, "steps": steps,
        "dtype": "f64", "step": "predict + weighting pass (no resample: the seeded cloud stays wide)",
        "scan": {"range_mean": float(dist.mean()), "range_median": float(np.median(dist)), "beams_at_max_dist": int((dist >= zmax).sum()),
                 "note": "the cost of a pass on a wide cloud depends on the scan: a particle's search ends at its hit or at z + 4 sigma; "
                         "scans from open places of the same grid are up to 5x slower (DESIGN.md, tools/uniform_time.py)"},
        "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algorithmic_frac(achieved)[0],
                     "algorithmic_over_hbm_peak": algorithmic_frac(achieved)[1],
                     "traffic": traffic,
                     "physical": physical_roofline(["weighting kernel of the 16384^2 shard (k_weights_L<0>, 12.5 M particles)"], w_ms, "k_weights_L<0>") if size == 16384 else None,
                     "kernel": "weighting pass (k_order_keys + order sort + k_rays_pl, or k_weights_L<0> on a sparse cloud)", "kernel_avg_ms": w_ms,
                     "algorithmic_bytes_per_launch": alg, "cells_per_ray_reference_walk": sbar,
                     "note": "achieved = ALGORITHMIC, reference-equivalent bytes (cells the reference's walk reads, from 4096 particles on the "
                             "cell-by-cell kernel) over the kernel time, as SURVEY.md §8d defines it; the clearance search reads a few bytes per "
                             "ray instead, so the figure can exceed the HBM peak (then frac is null and the ratio is algorithmic_over_hbm_peak). "
                             "What binds is f64 instruction issue: physical.valu"},
    }


def floor_plan_wide_entry(sa, capi, device, n, B, steps):
    """The headline map and scan with the cloud the headline STARTS from: Gaussian, sigma 25 cells / 0.2 rad around the true
    pose, never resampled (predict + weighting pass per step).  The headline's own cloud has collapsed onto the pose after
    its warm-up resamples (config.cloud): this entry is the same kernel chain on rays that do not share their cells."""
    import torch

    free = load_free_grid("floor_plan")
    grid = (free * 255).astype(np.uint8)
    ang, dist = synth_scan(sa, free, device, B)
    f = sa.Filter(n, *grid.shape, device=device, seed=SEED, sort=capi.SORT_NONE, async_launch=True)
    f.set_map(grid)
    f.seed_gaussian(TRUE_POSE, [25.0, 25.0, 0.2])
    f.update_weights(ang, dist, STDDEV, MAX_DIST, OFFSET)
    f.synchronize()
    f.profile(2)
    f.kernel_ms()
    t0 = time.perf_counter()
    for _ in range(steps):
        f.update_weights(ang, dist, STDDEV, MAX_DIST, OFFSET)
    f.synchronize()
    ms_step = (time.perf_counter() - t0) / steps * 1e3
    kms = f.kernel_ms()
    x, y, th = f.poses()
    f.close()
    wk = sa.Filter(n, *grid.shape, device=device, raycaster=capi.RAYCAST_WALK, sort=capi.SORT_NONE)
    wk.set_map(grid)
    wk.set_poses(x, y, th)
    wk.update_weights(ang, dist, STDDEV, MAX_DIST, OFFSET)
    sbar = wk.stats()["cells_read"] / float(n * B)
    wk.close()
    torch.cuda.synchronize()
    w_ms = kms["weights"][0] / max(kms["weights"][1], 1)
    alg = n * B * sbar + 40.0 * n + 16.0 * B
    achieved = alg / (w_ms * 1e-3) / 1e9
    return {
        "workload": f"BASELINE configs[1] map and scan, the cloud before any resample: {n} particles x {B} beams, Gaussian sigma 25 cells / 0.2 rad "
                    "around the true pose, one weighting pass per step (no resample: the cloud stays wide)",
        "value": n * B / (ms_step * 1e-3), "unit": "particle-beams/s", "ms_per_step": ms_step, "steps": steps, "dtype": "f64",
        "cloud": {"std_x": float(np.std(x)), "std_y": float(np.std(y)), "std_theta": float(np.std(th))},
        "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algorithmic_frac(achieved)[0],
                     "algorithmic_over_hbm_peak": algorithmic_frac(achieved)[1], "traffic": None,
                     "kernel": "weighting pass = k_order_keys + order sort + k_rays_pl + k_sum_lw", "kernel_avg_ms": w_ms,
                     "algorithmic_bytes_per_launch": alg, "cells_per_ray_reference_walk": sbar,
                     "note": "algorithmic, reference-equivalent bytes over the kernel time (SURVEY.md §8d); binds on f64 instruction issue and, "
                             "unlike the headline's collapsed cloud, on the spread of trip counts inside a wavefront"},
    }


def secondary_entries(sa, capi, device, with_cpu):
    """Configurations next to the headline one (BASELINE configs[0], [3], [2], [4] at one GPU's share)."""
    out = []
    want = os.environ.get("BENCH_SECONDARY", "wide100k,slam1k,slam125k,loc8192,loc16384").split(",")
    if "wide100k" in want:
        out.append(floor_plan_wide_entry(sa, capi, device, 100_000, 360, 10))
    if "slam1k" in want:
        out.append(slam_entry(sa, capi, device,
                              "BASELINE configs[0] shape on the GPU: grid-SLAM (slam::MCL), assets/map.png scans, 1000 particles x 90 beams, "
                              "1000x1000 canvases (tiled), full step = predict + update (walk + map update, select, sort, clone)",
                              1000, 1000, 1000, load_free_grid("map"), (400.0, 400.0, np.pi / 90), 0, 20, 5, with_cpu, reps=3))
    if "slam125k" in want:
        out.append(slam_entry(sa, capi, device,
                              "BASELINE configs[3], one GPU's share of 8: grid-SLAM, assets/maze.png scans (robot at (1944.5, 2188.5), mean range "
                              "~140 cells), 125000 particles x 90 beams, 2354x2464 canvases as copy-on-write tiles (64 GB pool)",
                              125_000, 2354, 2464, load_free_grid("maze"), (1944.5, 2188.5, np.pi / 2), 16_000_000, 8, 3, False))
    if "loc8192" in want:
        out.append(localization_shard_entry(sa, capi, device, "BASELINE configs[2], one GPU's share of 8: 1.25M particles x 1080 beams, synthetic "
                                            "8192^2 grid, Gaussian cloud, max_dist 500, sigma 0.5, log-domain weights", 8192, 1_250_000, 1080,
                                            500.0, False, 3))
    if "loc16384" in want:
        out.append(localization_shard_entry(sa, capi, device, "BASELINE configs[4], one GPU's share of 8: 12.5M uniformly seeded particles x 720 "
                                            "long-range beams (max_dist 4000), synthetic 16384^2 grid, sigma 0.5, log-domain weights",
                                            16384, 12_500_000, 720, 4000.0, True, 2))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=100_000, help="particles per GPU")
    ap.add_argument("--beams", type=int, default=360)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline configuration only")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import slam_amd as sa
    from slam_amd import capi
    from slam_amd.dist import HipShard, ShardedFilter

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_SHARDED=1: rehearse the multi-GPU code path (RCCL init, staged all-gather, sharded
    # resample) with whatever world size the launcher gives, one rank included
    force_sharded = os.environ.get("BENCH_FORCE_SHARDED", "0") == "1"
    if world > 1 or (force_sharded and "RANK" in os.environ):
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    stream = torch.cuda.current_stream().cuda_stream

    free = load_free_grid("floor_plan")
    grid = (free * 255).astype(np.uint8)
    rows, cols = grid.shape
    n_loc, B = args.particles, args.beams
    scans = synth_scans(sa, free, local_rank, B, true_trajectory(args.warmup + args.steps))
    ang, dscan = scans[0]
    step_no = [0]

    sharded = world > 1 or force_sharded
    if not sharded:
        f = sa.Filter(n_loc, rows, cols, device=local_rank, seed=SEED, stream=stream, async_launch=True)
        f.set_map(grid)
        step_predict = lambda: f.predict(ODOM, ALPHAS)
        upd = lambda a, d: f.update(a, d, STDDEV, MAX_DIST, OFFSET)
    else:
        shard = HipShard(n_loc, rows, cols, rank, world, local_rank, SEED, stream=stream, async_launch=True)
        shard.set_map(grid)
        sf = ShardedFilter(shard, world=world, always_collective=force_sharded and dist.is_initialized(),
                           split_exchange=os.environ.get("BENCH_SPLIT_EXCHANGE", "0") == "1")
        f = shard.f
        step_predict = lambda: sf.predict(ODOM, ALPHAS)
        upd = lambda a, d: sf.update(a, d, STDDEV, MAX_DIST, OFFSET)

    def step_update():
        a, d = scans[step_no[0]]
        step_no[0] += 1
        upd(a, d)
    f.seed_gaussian(TRUE_POSE, [25.0, 25.0, 0.2])  # Gaussian cloud around the true pose

    def barrier():
        if world > 1:
            dist.barrier()

    def measure_sbar():
        """Mean cells the reference's walk reads per ray for the CURRENT poses (cell-by-cell kernel)."""
        x, y, th = f.poses()
        wk = sa.Filter(n_loc, rows, cols, device=local_rank, raycaster=capi.RAYCAST_WALK, sort=capi.SORT_NONE)
        wk.set_map(grid)
        wk.set_poses(x, y, th)
        a_, d_ = scans[min(step_no[0], len(scans) - 1)]
        wk.update_weights(a_, d_, STDDEV, MAX_DIST, OFFSET)
        s = wk.stats()["cells_read"] / float(n_loc * B)
        wk.close()
        return s

    for _ in range(args.warmup):
        step_predict()
        step_update()
    torch.cuda.synchronize()
    sbar0 = measure_sbar() if rank == 0 else 0.0
    # HIP events over the timed region for the dominant kernel only (level 2: a start event on the first and a stop
    # event on the last kernel of the weighting pass, carried by those dispatches themselves).  An event pair around
    # EVERY group (level 1) puts ten event records per step into the stream, ~10 us of idle GPU each: that breakdown
    # is taken in a pass of its own after the timed region (kernel_ms_per_step).
    f.profile(2)
    f.kernel_ms()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_predict()
        step_update()
    host_elapsed = time.perf_counter() - t0   # time to ENQUEUE the steps (diagnostic: is the loop host-bound?)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kms = f.kernel_ms()
    # the per-group breakdown: the same steps once more (the scans repeat), every group bracketed by events
    f.profile(1)
    step_no[0] = max(0, len(scans) - args.steps)
    for _ in range(args.steps):
        step_predict()
        step_update()
    torch.cuda.synchronize()
    barrier()
    groups = f.kernel_ms()
    f.profile(False)
    # what a step costs the HOST: three steps enqueued on an idle GPU (fewer than the four the scan ring lets the host run
    # ahead, so that nothing here waits for the device; over the 20 timed steps the host is throttled to the GPU's pace by
    # that ring, and host_enqueue_ms_per_step only says that it kept up)
    step_no[0] = max(0, len(scans) - 3)
    torch.cuda.synchronize()
    th0 = time.perf_counter()
    for _ in range(3):
        step_predict()
        step_update()
    host_own = (time.perf_counter() - th0) / 3 * 1e3
    torch.cuda.synchronize()
    barrier()

    if rank == 0:
        # how collapsed the timed cloud is: ancestors of the last resample and the spread of the poses
        last_idx = f.last_indices()
        cx, cy, cth = f.poses()
        cloud_stats = {"unique_ancestors_last_resample": int(np.unique(last_idx).size), "particles": int(n_loc),
                       "std_x_cells": float(np.std(cx)), "std_y_cells": float(np.std(cy)), "std_theta_rad": float(np.std(cth)),
                       "note": "after the warm-up and timed resamples (sigma 5, 360 beams) the cloud has converged onto the pose: the rays of "
                               "a wavefront share their cells, the best case for the ray kernel; secondary[0] is the same chain on the "
                               "un-resampled sigma-25 cloud"}
        sbar1 = measure_sbar()
        sbar = 0.5 * (sbar0 + sbar1)
        ms_step = elapsed / args.steps * 1e3
        total_pb = float(world) * n_loc * B
        w_ms, w_n = kms["weights"]
        k_avg_ms = w_ms / max(w_n, 1)
        # per particle-beam S_bar x 1 B (cells the reference reads) + 40/B B (pose in, weight out)
        alg_bytes = n_loc * B * sbar + 40.0 * n_loc + 16.0 * B
        achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9
        traffic = None
        # HBM bytes per launch from the rocprofv3 PMC passes of profiles/run_profile.sh: quoted only while the
        # kernel sources are the ones the counters were taken from (the file records their fingerprint)
        pmc = os.path.join(ROOT, "profiles", "pmc_weights_kernel.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("source_fingerprint") == source_fingerprint():
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "particle-beam raycasts+likelihoods/sec",
            "value": total_pb / (ms_step * 1e-3),
            "unit": "particle-beams/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[1]: MCL localization, floor_plan 1297x599, {n_loc} particles x {B} beams per GPU, "
                            f"max_dist {MAX_DIST:g}, sigma {STDDEV:g}, full step = predict + update (weights, running sum, "
                            f"multinomial select, {'' if sharded else 'sort, '}gather)",
                "particles_per_gpu": n_loc, "beams": B, "map": "floor_plan", "noise": "philox4x32-10", "cloud": cloud_stats,
                "resampling": "multinomial (reference rule)" + (", global over all ranks via one all-gather" if sharded else ""),
            },
            "roofline": {
                "bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": algorithmic_frac(achieved)[0], "algorithmic_over_hbm_peak": algorithmic_frac(achieved)[1], "traffic": traffic,
                "physical": physical_roofline(["k_rays_pl", "k_sum_lw"], k_avg_ms, "k_rays_pl"),
                "note": "achieved/frac are ALGORITHMIC, reference-equivalent bytes (cells the reference's walk would read, SURVEY.md §8d) "
                        "over the kernel time; the clearance search reads far fewer bytes and the binding limit is VALU issue (DESIGN.md §6)",
                "kernel": "weighting pass = k_order_keys + pose-coherent order (sample sort, 4 launches) + k_rays_pl + k_sum_lw (raycast + beam likelihood)", "kernel_avg_ms": k_avg_ms,
                "algorithmic_bytes_per_launch": alg_bytes, "cells_per_ray_reference_walk": sbar,
                "kernel_particle_beams_per_s": n_loc * B / (k_avg_ms * 1e-3),
            },
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in groups.items()},
            "kernel_ms_per_step_note": "from a second pass of the same steps after the timed region, every group bracketed by HIP events "
                                       "(the timed region itself carries only the weighting pass's two events)",
            "host_enqueue_ms_per_step": host_elapsed / args.steps * 1e3,
            "host_cost_ms_per_step": host_own,
            "host_note": "host_enqueue_ms_per_step: wall time of the enqueue loop of the timed region (the scan ring lets the host run at most "
                         "4-8 steps ahead, so it tracks the GPU); host_cost_ms_per_step: the host's own time to enqueue a step (3 steps on an idle GPU)",
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(free, ang, dscan, B)
        if world == 1 and not sharded and not args.no_secondary:
            f.close()
            out["secondary"] = secondary_entries(sa, capi, local_rank, not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
