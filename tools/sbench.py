"""SLAM-mode (per-particle canvases) step timing, BASELINE configs[0]-like: N particles x 90 beams, 1000x1000 canvas."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import slam_amd as sa
from slam_amd import capi
N = int(os.environ.get("SB_N", 1000)); B = 90; C = int(os.environ.get("SB_C", 1000))
free = bench.load_free_grid("map")
poses = []
x, y, th = 400.0, 400.0, np.pi / 90
odom = [0.0, 2.5, 0.0]  # apps/grid_slam.cpp:48-50 (UP key)
alphas = [1e-3] * 4     # apps/grid_slam.cpp:132
for _ in range(12):
    x, y = x + 2.5 * np.cos(th), y + 2.5 * np.sin(th)
    poses.append((x, y, th))
old = bench.OFFSET
sens = [sa.MCL.sensor_position(sa.Pose(*p), sa.Pose(*bench.OFFSET)) for p in poses]
sp = np.array([[s.x, s.y, s.theta] for s in sens]); hits = np.empty((len(poses), B, 2))
m01 = np.ascontiguousarray(free, dtype=np.uint8)
capi.check(capi.load_library().slam_fake_lidar_scan(0, capi.ptr(m01), m01.shape[0], m01.shape[1], capi.ptr(sp), len(poses), 0.0, 2 * np.pi, 500.0, B, capi.ptr(hits)))
ang = np.arange(B) * (2 * np.pi / B) - np.pi
scans = [(ang, np.where(hits[k, :, 0] != -1, np.hypot(hits[k, :, 1] - sp[k, 1], hits[k, :, 0] - sp[k, 0]), 500.0)) for k in range(len(poses))]
CR = int(os.environ.get("SB_ROWS", C))
variants = [("wave", capi.RAYCAST_FAST_BEAM_LANES, capi.MAPS_DENSE), ("dense2", capi.RAYCAST_TWO_PHASE, capi.MAPS_DENSE),
            ("tiled_old", capi.RAYCAST_FAST_BEAM_LANES, capi.MAPS_TILED),
            ("tiled", capi.RAYCAST_FAST, capi.MAPS_TILED)]  # dense is asked for explicitly: tiles + the two-phase kernel are the default
if os.environ.get("SB_THREAD", "0") == "1":
    variants.append(("thread", capi.RAYCAST_FAST_PARTICLE_LANES, capi.MAPS_DENSE))
if os.environ.get("SB_ONLY"):
    variants = [v for v in variants if v[0] in os.environ["SB_ONLY"].split(",")]
if os.environ.get("SB_ONLY") == "shard":
    # one tiled shard driven through slam_amd.dist (world 1): select, sparse pack (tables + distinct tiles), import
    import torch
    from slam_amd.dist import HipShard, ShardedFilter
    sf = ShardedFilter(HipShard(N, CR, C, 0, 1, 0, 1234, mode=capi.MODE_SLAM, map_storage=capi.MAPS_TILED,
                                tile_pool_tiles=int(os.environ.get("SB_POOL", 0))), world=1)
    sf.b.f.profile(True)
    for k in range(len(scans)):
        sf.predict(odom, alphas)
        sf.update(*scans[k], 5.0, 500.0, bench.OFFSET)
        if k == 3:
            sf.b.f.kernel_ms()
            torch.cuda.synchronize(); t0 = time.perf_counter()
    sf.b.f.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (len(scans) - 4) * 1e3
    km = sf.b.f.kernel_ms()
    print(f"shard   N={N} canvas={CR}x{C}: {dt:8.3f} ms/step wall;", {k: round(v[0] / max(v[1], 1), 3) for k, v in km.items()},
          sf.b.f.tile_stats(), flush=True)
    sys.exit(0)
for name, rc, storage in variants:
    f = sa.Filter(N, CR, C, mode=capi.MODE_SLAM, raycaster=rc, map_storage=storage,
                  tile_pool_tiles=int(os.environ.get("SB_POOL", 0)) if storage == capi.MAPS_TILED else 0)
    f.profile(True)
    for k in range(len(scans)):
        f.predict(odom, alphas)
        f.update(*scans[k], 5.0, 500.0, bench.OFFSET)
        if k == 3:
            f.kernel_ms()
            t0 = time.perf_counter()
    f.synchronize()
    dt = (time.perf_counter() - t0) / (len(scans) - 4) * 1e3
    km = f.kernel_ms(); st = f.stats()
    if storage == capi.MAPS_TILED:  # one more step in two halves: tiles copied by one weighting pass
        f.predict(odom, alphas); f.update_weights(*scans[-1], 5.0, 500.0, bench.OFFSET)
        print("   tiles copied by one weighting pass:", f.tile_stats()["allocated_since"], flush=True)
        f.resample(); f.kernel_ms()
    print(f"{name:7s} N={N} canvas={CR}x{C}: {dt:8.3f} ms/step wall;", {k: round(v[0] / max(v[1], 1), 3) for k, v in km.items()},
          f"cells_read/ray={st['cells_read'] / (N * B):.1f} written/ray={st['cells_written'] / (N * B):.1f}",
          f.tile_stats() if storage == capi.MAPS_TILED else "", flush=True)
    f.close()
if os.environ.get("SB_CPU", "1") == "1":
    import _oracle as orc
    n = min(N, 256)
    maps = np.full((n, C, C), 128, np.uint8); xs = np.full(n, C / 2.0); ys = np.full(n, C / 2.0); ths = np.full(n, np.pi / 2)
    cores = min(16, len(os.sched_getaffinity(0)))
    for k in range(6):
        xs, ys, ths = orc.predict(xs, ys, ths, odom, alphas, orc.philox_normals(1, k, 0, n))
        t0 = time.perf_counter()
        w, _ = orc.update_weights(xs, ys, ths, maps, *scans[k], 5.0, 500.0, bench.OFFSET, slam_mode=True, n_threads=cores, fast=True)
        t1 = time.perf_counter()
        idx, _, _ = orc.resample(w, orc.philox_uniforms(1, k, 0, n))
        src = idx[orc.sort_order(w[idx], 1)]
        maps = maps[src]; xs, ys, ths = xs[src], ys[src], ths[src]
        t2 = time.perf_counter()
    print(f"cpu oracle (-Ofast, {cores} threads) n={n}: weights {1e3 * (t1 - t0):.2f} ms, resample+clone {1e3 * (t2 - t1):.2f} ms -> per 1000 particles {1e3 * (t2 - t0) * 1000 / n:.1f} ms")
