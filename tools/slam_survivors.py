"""Grid-SLAM at the bench's 125 000-particle shard: distinct ancestors per resample and family sizes (who copies tiles)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import slam_amd as sa
from slam_amd import capi
N = int(os.environ.get("SB_N", 125000)); B = 90
free = bench.load_free_grid("maze")
scans = bench.slam_scans(sa, free, 0, B, (1944.5, 2188.5, np.pi / 2), 8)
f = sa.Filter(N, 2354, 2464, mode=capi.MODE_SLAM, seed=bench.SEED, map_storage=capi.MAPS_TILED, tile_pool_tiles=16_000_000)
for k in range(8):
    f.predict([0.0, 2.5, 0.0], [1e-3] * 4)
    f.update_weights(*scans[k], 5.0, 500.0, bench.OFFSET)
    w = f.weights()
    ts = f.tile_stats()
    f.resample()
    idx = f.last_indices()
    u, c = np.unique(idx, return_counts=True)
    ess = w.sum() ** 2 / (w ** 2).sum()
    print(f"step {k}: tiles copied {ts['allocated_since']} ({ts['allocated_since'] / N:.1f} per particle), live {ts['live']}; ESS {ess:.0f}; distinct ancestors {u.size} "
          f"({100.0 * u.size / N:.1f} %), single children {int((c == 1).sum())}, largest family {c.max()}, weights max/median {w.max() / np.median(w):.3g}", flush=True)
