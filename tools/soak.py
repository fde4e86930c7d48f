"""Randomised soak of the localization weighting pass: the clearance-skipping kernels (both shapes)
against the cell-by-cell walk kernel on random maps, ranges, offsets, clouds.  Not part of the test
suite; run on a GPU box: python tools/soak.py [seconds]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slam_amd as sa
from slam_amd import capi, synth
import bench
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", 1)))
maps = {n: (bench.load_free_grid(n) * 255).astype(np.uint8) for n in ("floor_plan", "map", "maze")}
maps["synth2048"] = synth.make_grid(2048)
t0 = time.time(); cases = 0; rays = 0
while time.time() - t0 < budget:
    name = rng.choice(list(maps)); g = maps[name]
    if rng.random() < 0.3:   # random crop / random noise map
        r0, c0 = rng.integers(0, g.shape[0] // 2), rng.integers(0, g.shape[1] // 2)
        g = np.ascontiguousarray(g[r0:r0 + rng.integers(3, g.shape[0] - r0), c0:c0 + rng.integers(3, g.shape[1] - c0)])
    if rng.random() < 0.15:
        g = ((rng.random(g.shape) > rng.choice([0.001, 0.02, 0.3])) * 255).astype(np.uint8)
    rows, cols = g.shape
    N = int(rng.choice([100, 4096, 5000, 20000])); B = int(rng.choice([1, 7, 90, 360]))
    zmax = float(rng.choice([0.7, 5.5, 61.25, 500.0, 1800.0]))
    off = [rng.uniform(-40, 40), rng.uniform(-40, 40), rng.uniform(-3, 3)] if rng.random() < 0.7 else [0.0, 30.0, 0.0]
    cx, cy = rng.uniform(0, cols), rng.uniform(0, rows)
    spread = rng.choice([0.5, 5.0, 60.0, 2000.0])
    x = cx + spread * rng.standard_normal(N); y = cy + spread * rng.standard_normal(N)
    th = rng.uniform(-2 * np.pi, np.pi, N) if rng.random() < 0.5 else rng.uniform(-3, 3) + 0.05 * rng.standard_normal(N)
    ang = np.sort(rng.uniform(-np.pi, np.pi, B)); dist = rng.uniform(0, zmax, B); dist[rng.random(B) < 0.2] = zmax
    sd = float(rng.choice([0.5, 5.0]))
    res = {}
    for shape, rc in (("walk", capi.RAYCAST_WALK), ("beam", capi.RAYCAST_FAST_BEAM_LANES), ("plane", capi.RAYCAST_FAST_PARTICLE_LANES)):
        f = sa.Filter(N, rows, cols, raycaster=rc, sort=capi.SORT_NONE, weight_domain=capi.WEIGHT_LOG)
        f.set_map(g); f.set_poses(x, y, th); f.record_hits(True)
        f.update_weights(ang, dist, sd, zmax, off)
        res[shape] = (f.weights(), *f.hits())
        if shape != "walk":
            # without recorded hits the hit search stops beyond z + 4 sigma (ray_fast_L0's k_lim): same weights
            f.record_hits(False)
            f.update_weights(ang, dist, sd, zmax, off)
            res[shape + "/limit"] = f.weights()
        f.close()
    for shape in ("beam", "plane"):
        ok = np.array_equal(res[shape][2], res["walk"][2]) and np.array_equal(res[shape][1].view(np.uint64), res["walk"][1].view(np.uint64)) \
            and np.array_equal(res[shape][0].view(np.uint64), res["walk"][0].view(np.uint64))
        ok = ok and np.array_equal(res[shape + "/limit"].view(np.uint64), res["walk"][0].view(np.uint64))
        if not ok:
            print("MISMATCH", shape, name, g.shape, N, B, zmax, off, cx, cy, spread, flush=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "soak_fail.npz"), g=g, x=x, y=y, th=th, ang=ang, dist=dist, zmax=zmax, off=off, sd=sd)
            sys.exit(1)
    cases += 1; rays += N * B
    if cases % 20 == 0:
        print(f"{cases} cases, {rays / 1e6:.0f} M rays, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {cases} cases, {rays / 1e6:.0f} M rays")
