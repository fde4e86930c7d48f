"""Randomised soak of the sharded SLAM-mode exchange of tiled canvases (tables + distinct tiles,
slam_mcl_pack_tiles / slam_mcl_shard_import_tiles; the all-to-alls of slam_amd.dist emulated on one
device) against one dense handle: selection, weights and canvases must be identical over several
steps.  Not part of the test suite: python tools/soak_shard.py [seconds]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slam_amd as sa
from slam_amd import capi
from slam_amd.dist import HipShard
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", 3)))
t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    rows, cols = int(rng.integers(20, 700)), int(rng.integers(20, 700))
    S = int(rng.choice([1, 2, 4])); n_loc = int(rng.choice([5, 16, 75, 130])); N = S * n_loc
    B = int(rng.choice([1, 13, 90])); zmax = float(rng.choice([3.5, 40.0, 150.0, 500.0])); seed = int(rng.integers(1, 2 ** 31))
    single = sa.Filter(N, rows, cols, mode=capi.MODE_SLAM, seed=seed, sort=capi.SORT_NONE, map_storage=capi.MAPS_DENSE)
    backs = [HipShard(n_loc, rows, cols, r, S, 0, seed, mode=capi.MODE_SLAM, map_storage=capi.MAPS_TILED) for r in range(S)]
    x = cols / 2 + rng.choice([1.0, 30.0, 400.0]) * rng.standard_normal(N); y = rows / 2 + rng.choice([1.0, 30.0]) * rng.standard_normal(N)
    th = rng.uniform(-3, 3, N)
    single.set_poses(x, y, th)
    for r, b in enumerate(backs):
        sl = slice(r * n_loc, (r + 1) * n_loc)
        b.f.set_poses(x[sl], y[sl], th[sl])
    off = [rng.uniform(-10, 10), rng.uniform(-10, 10), rng.uniform(-1, 1)]
    for step in range(int(rng.integers(1, 6))):
        ang = np.sort(rng.uniform(-np.pi, np.pi, B)); dist = rng.uniform(0, zmax, B); dist[rng.random(B) < 0.3] = zmax
        odom = [rng.uniform(-0.3, 0.3), rng.uniform(0, 8), rng.uniform(-0.3, 0.3)]
        al = [1e-3, 1e-3, 1e-2, 1e-2]
        single.predict(odom, al); single.update(ang, dist, 5.0, zmax, off)
        for b in backs:
            b.predict(odom, al); b.update_weights(ang, dist, 5.0, zmax, off)
        gathered = torch.cat([b.packed_state() for b in backs]); torch.cuda.synchronize()
        idxs = [b.shard_select(gathered, S) for b in backs]
        got = []
        for r in range(S):
            want = torch.unique(idxs[r]); owner = torch.div(want, n_loc, rounding_mode="floor")
            tabs, tiles, base = [], [], 0
            for o in range(S):
                t, p = backs[o].pack_tiles(want[owner == o] - o * n_loc)
                tabs.append(torch.where(t > 0, t + base, t)); tiles.append(p); base += p.shape[0]
            got.append((torch.cat(tabs), torch.cat(tiles), torch.searchsorted(want, idxs[r])))
        for r, b in enumerate(backs):
            b.import_tiles(*got[r])
            assert b.f.tile_stats()["live"] <= got[r][1].shape[0]
        ok = np.array_equal(np.concatenate([b.f.last_indices() for b in backs]), single.last_indices())
        ok = ok and np.array_equal(np.concatenate([b.f.weights() for b in backs]).view(np.uint64), single.weights().view(np.uint64))
        for p in rng.choice(N, min(N, 6), replace=False):
            ok = ok and np.array_equal(backs[int(p) // n_loc].f.map_of(int(p) % n_loc), single.map_of(int(p)))
        if not ok:
            print("MISMATCH", rows, cols, S, n_loc, B, zmax, seed, step, flush=True)
            sys.exit(1)
    single.close()
    for b in backs:
        b.f.close()
    cases += 1
    if cases % 10 == 0:
        print(f"{cases} cases, {time.time() - t0:.0f} s", flush=True)
print(f"shard soak ok: {cases} cases")
