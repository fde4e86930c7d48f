"""Weighting pass on a uniformly seeded cloud (global localization, first steps): python tools/uniform_time.py [map] [N] [B] [sigma] [max_dist]
SLAM_PL_ORDER forces an order (3 = heading first, 1618 = place first); default = chosen from the cloud's extent."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import slam_amd as sa
from slam_amd import capi
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "floor_plan"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 360
SIGMA = float(sys.argv[4]) if len(sys.argv) > 4 else 5.0
ZMAX = float(sys.argv[5]) if len(sys.argv) > 5 else 500.0
if name.startswith("synth"):  # synth8192, synth16384: the benchmark's synthetic grids
    from slam_amd import synth
    g = synth.make_grid(int(name[5:]))
    free = (g >= 128).astype(np.uint8)
else:
    free = bench.load_free_grid(name)
    g = (free * 255).astype(np.uint8)
rows, cols = g.shape
sub = free[rows // 4:rows // 4 + 64]
cells = np.argwhere(sub > 0)
ci, cj = cells[len(cells) // 2]
ci += rows // 4
true = [float(cj) + 0.5, float(rows - 1 - ci) + 0.5, 0.7]
gm = sa.GridMap(free.astype(np.uint8))
hits = gm.fake_lidar_scan(true, 0.0, 2 * np.pi, ZMAX, B)[0]
gm.close()
ang = np.arange(B) * (2 * np.pi / B) - np.pi
dist = np.where(hits[:, 0] != -1, np.hypot(hits[:, 1] - true[1], hits[:, 0] - true[0]), ZMAX)
f = sa.Filter(N, rows, cols, seed=3, sort=capi.SORT_NONE, weight_domain=capi.WEIGHT_LOG, async_launch=True,
              raycaster=int(os.environ.get("UT_RAYCASTER", capi.RAYCAST_FAST)))  # 2 = one wavefront per particle, lanes over beams
f.set_map(g)
f.seed_uniform_free()
f.update_weights(ang, dist, SIGMA, ZMAX, [0.0, 0.0, 0.0])
f.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    f.update_weights(ang, dist, SIGMA, ZMAX, [0.0, 0.0, 0.0])
f.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"{name} {rows}x{cols}, {N} uniform particles x {B} beams: {dt * 1e3:.3f} ms per pass = {N * B / dt / 1e9:.1f} G particle-beams/s "
      f"(SLAM_PL_ORDER={os.environ.get('SLAM_PL_ORDER', 'auto')})", flush=True)
if os.environ.get("UT_TRACE"):
    f.profile(True); f.kernel_ms()
    for _ in range(5):
        f.update_weights(ang, dist, SIGMA, ZMAX, [0.0, 0.0, 0.0])
    f.synchronize()
    print({k: (round(v[0] / max(v[1], 1), 4), v[1]) for k, v in f.kernel_ms().items()})
    f.set_poses(*[np.full(N, v) for v in true])
    f.update_weights(ang, dist, SIGMA, ZMAX, [0.0, 0.0, 0.0]); f.kernel_ms()
    for _ in range(5):
        f.update_weights(ang, dist, SIGMA, ZMAX, [0.0, 0.0, 0.0])
    f.synchronize()
    print("all particles at the true pose:", {k: (round(v[0] / max(v[1], 1), 4), v[1]) for k, v in f.kernel_ms().items()})
f.close()
