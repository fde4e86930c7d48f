"""Randomised soak of SLAM mode: dense wave kernel vs dense thread kernel vs tiled canvases over
several filter steps on random canvases, scans and clouds (weights, selection and canvases must be
identical).  Not part of the test suite: python tools/soak_slam.py [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slam_amd as sa
from slam_amd import capi
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", 2)))
t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    rows, cols = int(rng.integers(20, 700)), int(rng.integers(20, 700))
    N = int(rng.choice([16, 64, 100, 300])); B = int(rng.choice([1, 13, 90, 200]))
    zmax = float(rng.choice([3.5, 40.0, 150.0, 500.0])); seed = int(rng.integers(1, 2 ** 31))
    T = ((rows + 63) // 64) * ((cols + 63) // 64)
    pool = 0 if rng.random() < 0.5 else int(1.5 * N * T) + 8
    mk = lambda **kw: sa.Filter(N, rows, cols, mode=capi.MODE_SLAM, seed=seed, **kw)
    fs = {"wave": mk(raycaster=capi.RAYCAST_FAST_BEAM_LANES, map_storage=capi.MAPS_DENSE),
          "thread": mk(raycaster=capi.RAYCAST_FAST_PARTICLE_LANES, map_storage=capi.MAPS_DENSE),
          "tiled": mk(map_storage=capi.MAPS_TILED, tile_pool_tiles=pool),
          "tiled_old": mk(raycaster=capi.RAYCAST_FAST_BEAM_LANES, map_storage=capi.MAPS_TILED, tile_pool_tiles=pool),
          "dense2": mk(raycaster=capi.RAYCAST_TWO_PHASE, map_storage=capi.MAPS_DENSE)}
    x = cols / 2 + rng.choice([1.0, 30.0, 400.0]) * rng.standard_normal(N); y = rows / 2 + rng.choice([1.0, 30.0]) * rng.standard_normal(N)
    th = rng.uniform(-3, 3, N)
    for f in fs.values():
        f.set_poses(x, y, th)
    off = [rng.uniform(-10, 10), rng.uniform(-10, 10), rng.uniform(-1, 1)]
    for step in range(int(rng.integers(1, 6))):
        ang = np.sort(rng.uniform(-np.pi, np.pi, B)); dist = rng.uniform(0, zmax, B); dist[rng.random(B) < 0.3] = zmax
        odom = [rng.uniform(-0.3, 0.3), rng.uniform(0, 8), rng.uniform(-0.3, 0.3)]
        for f in fs.values():
            f.predict(odom, [1e-3, 1e-3, 1e-2, 1e-2])
            if rng.random() < 0.0:
                pass
            f.update(ang, dist, 5.0, zmax, off)
        ref = fs["wave"]
        for name in ("thread", "tiled", "tiled_old", "dense2"):
            f = fs[name]
            ok = np.array_equal(f.last_indices(), ref.last_indices()) and np.array_equal(f.weights().view(np.uint64), ref.weights().view(np.uint64))
            for p in rng.choice(N, 4, replace=False):
                ok = ok and np.array_equal(f.map_of(int(p)), ref.map_of(int(p)))
            if not ok:
                print("MISMATCH", name, rows, cols, N, B, zmax, seed, pool, step, flush=True)
                sys.exit(1)
    for f in fs.values():
        f.close()
    cases += 1
    if cases % 10 == 0:
        print(f"{cases} cases, {time.time() - t0:.0f} s", flush=True)
print(f"slam soak ok: {cases} cases")
