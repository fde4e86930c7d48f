"""Randomised soak of SLAM mode: dense wave kernel vs dense thread kernel vs tiled canvases over
several filter steps on random canvases, scans and clouds (weights, selection and canvases must be
identical).  Not part of the test suite: python tools/soak_slam.py [seconds]
SOAK_EVEN=1 spaces the beams of a scan evenly (a real scanner's fan), which is what lets the tiled filter defer its
canvases: run it with SLAM_DEFER=2 to soak "weights now, canvases of the survivors later" against the dense kernels."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slam_amd as sa
from slam_amd import capi
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", 2)))
t0 = time.time(); cases = 0
n_cases = int(os.environ.get("SOAK_CASES", 0))  # > 0: this many cases, whatever the time
while (cases < n_cases) if n_cases else (time.time() - t0 < budget):
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
    history = []
    for step in range(int(rng.integers(1, 6))):
        ang = np.sort(rng.uniform(-np.pi, np.pi, B))
        if os.environ.get("SOAK_EVEN"):
            fan = rng.choice([np.pi / 2, np.pi, 2 * np.pi])
            ang = rng.uniform(-np.pi, np.pi) + fan * np.arange(B) / B + (rng.uniform(-1e-3, 1e-3, B) if rng.random() < 0.5 else 0.0)
        dist = rng.uniform(0, zmax, B); dist[rng.random(B) < 0.3] = zmax
        odom = [rng.uniform(-0.3, 0.3), rng.uniform(0, 8), rng.uniform(-0.3, 0.3)]
        history.append((ang, dist, odom))
        for f in fs.values():
            f.predict(odom, [1e-3, 1e-3, 1e-2, 1e-2])
            if rng.random() < 0.0:
                pass
            f.update(ang, dist, 5.0, zmax, off)
        ref = fs["wave"]
        for name in ("thread", "tiled", "tiled_old", "dense2"):
            f = fs[name]
            ok = np.array_equal(f.last_indices(), ref.last_indices()) and np.array_equal(f.weights().view(np.uint64), ref.weights().view(np.uint64))
            for p in (range(N) if os.environ.get("SOAK_ALL") and name == "tiled" else rng.choice(N, 4, replace=False)):
                ok = ok and np.array_equal(f.map_of(int(p)), ref.map_of(int(p)))
            if not ok:
                print("MISMATCH", name, rows, cols, N, B, zmax, seed, pool, step, flush=True)
                wa, wb = f.weights(), ref.weights()
                bad = np.flatnonzero(wa.view(np.uint64) != wb.view(np.uint64))
                print("  weights differ at", bad.size, "of", N, bad[:8], wa[bad[:4]], wb[bad[:4]])
                print("  indices differ at", int((f.last_indices() != ref.last_indices()).sum()))
                px, py, pth = ref.poses()
                for p in range(N):
                    ma, mb = f.map_of(p), ref.map_of(p)
                    if not np.array_equal(ma, mb):
                        ii, jj = np.nonzero(ma != mb)
                        print("  canvas", p, "pose", px[p], py[p], pth[p], "cells", ii.size, list(zip(ii[:6], jj[:6])), ma[ii[:6], jj[:6]], mb[ii[:6], jj[:6]])
                        break
                print("  angles", ang[:4], "off", off, "dist", dist[:6])
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez(os.path.join(ROOT, "gpurun_out", "soak_case.npz"), dims=[rows, cols, N, B, seed, pool], zmax=zmax, x=x, y=y, th=th, off=off,
                         ang=np.array([h[0] for h in history]), dist=np.array([h[1] for h in history]), odom=np.array([h[2] for h in history]))
                sys.exit(1)
    for f in fs.values():
        f.close()
    cases += 1
    if cases % 10 == 0:
        print(f"{cases} cases, {time.time() - t0:.0f} s", flush=True)
print(f"slam soak ok: {cases} cases")
