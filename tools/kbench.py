"""Times the weight-kernel variants on the bench workload (GPU box).  Usage: python tools/kbench.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import slam_amd as sa  # noqa: E402
from slam_amd import capi  # noqa: E402

N = int(os.environ.get("KB_N", 100000))
B = int(os.environ.get("KB_B", 360))
free = bench.load_free_grid("floor_plan")
grid = (free * 255).astype(np.uint8)
rows, cols = grid.shape
NS = int(os.environ.get("KB_STEPS", 6))
scans = bench.synth_scans(sa, free, 0, B, bench.true_trajectory(NS))
variants = {"beam_lanes": capi.RAYCAST_FAST_BEAM_LANES, "particle_lanes": capi.RAYCAST_FAST_PARTICLE_LANES}
ref = sa.Filter(N, rows, cols, seed=bench.SEED)
ref.set_map(grid)
ref.seed_gaussian(bench.TRUE_POSE, [25.0, 25.0, 0.2])
states = []
golden = {}
for step in range(NS):
    ref.predict(bench.ODOM, bench.ALPHAS)
    states.append((step, ref.poses()))
    ref.update(*scans[step], bench.STDDEV, bench.MAX_DIST, bench.OFFSET)
for name, rc in variants.items():
    f = sa.Filter(N, rows, cols, raycaster=rc, sort=capi.SORT_NONE)
    f.set_map(grid)
    f.profile(True)
    for step, (x, y, th) in states:
        f.set_poses(x, y, th)
        ang, dist = scans[step]
        f.update_weights(ang, dist, bench.STDDEV, bench.MAX_DIST, bench.OFFSET)
        f.kernel_ms()
        for _ in range(3):
            f.update_weights(ang, dist, bench.STDDEV, bench.MAX_DIST, bench.OFFSET)
        ms, n = f.kernel_ms()["weights"]
        w = f.weights()
        if name == "beam_lanes":
            golden[step] = w
        else:
            assert os.environ.get("KB_NOCHECK") or np.array_equal(w.view(np.uint64), golden[step].view(np.uint64)), "variants disagree"
        print(f"{name:15s} step {step}: {ms / n:8.3f} ms  -> {N * B / (ms / n) / 1e6:8.1f} G pb/s", flush=True)
    f.close()
