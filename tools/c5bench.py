"""Global-localization regime (BASELINE configs[4]-like): particles uniform over a 16384^2 synthetic grid,
zmax 4000.  Times one weighting pass.  Usage: python tools/c5bench.py [N] [B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slam_amd as sa
from slam_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 72
size, zmax = int(os.environ.get("C5_SIZE", 16384)), float(os.environ.get("C5_ZMAX", 4000.0))
g = synth.make_grid(size)
free01 = (g >= 128).astype(np.uint8)
fc = np.argwhere(g[size // 4:size // 4 + 64, size // 4:size // 4 + 2000] >= 128)
ci, cj = fc[len(fc) // 2]
true = [float(cj + size // 4) + 0.5, float(size - 1 - (ci + size // 4)) + 0.5, 0.7]
sp = sa.MCL.sensor_position(sa.Pose(*true), sa.Pose(0.0, 30.0, 0.0))
sensor = np.array([sp.x, sp.y, sp.theta])
hits = np.empty((B, 2))
lib = capi.load_library()
capi.check(lib.slam_fake_lidar_scan(0, capi.ptr(free01), size, size, capi.ptr(np.ascontiguousarray(sensor[None])), 1, 0.0, 2 * np.pi, zmax, B, capi.ptr(hits)))
# (angle, distance) per ray as benchmark/mcl.cpp:37-56 builds them from the hit points; a miss reads max_dist
ang = np.arange(B) * (2 * np.pi / B) - np.pi
dist = np.where(hits[:, 0] != -1, np.hypot(hits[:, 1] - sensor[1], hits[:, 0] - sensor[0]), zmax)
f = sa.Filter(N, size, size, seed=7, sort=capi.SORT_NONE, weight_domain=capi.WEIGHT_LOG)
f.set_map(g)
f.seed_uniform_free()
f.profile(True)
for it in range(3):
    f.update_weights(ang, dist, 0.5, zmax, [0.0, 30.0, 0.0])
    ms, n = f.kernel_ms()["weights"]
    print(f"pass {it}: {ms / n:.2f} ms = {N * B / (ms / n) / 1e6:.2f} G particle-beams/s", flush=True)
if os.environ.get("C5_WALK"):
    sub = np.random.default_rng(0).choice(N, 4096, replace=False)
    x, y, th = f.poses()
    w = sa.Filter(4096, size, size, raycaster=capi.RAYCAST_WALK, sort=capi.SORT_NONE, weight_domain=capi.WEIGHT_LOG)
    w.set_map(g); w.set_poses(x[sub], y[sub], th[sub]); w.update_weights(ang, dist, 0.5, zmax, [0.0, 30.0, 0.0])
    print("reference walk cells/ray:", w.stats()["cells_read"] / (4096 * B))
f.close()
