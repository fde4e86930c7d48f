"""Per-call host times of the configs[0]-shape grid-SLAM entry (bench.py slam_entry) in a fresh process: where does an
occasional 2-3 ms per step of host time come from?  python tools/slam1k_host.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import slam_amd as sa
from slam_amd import capi
import bench
torch.cuda.set_device(0)
free = bench.load_free_grid("map")
B = 90
scans = bench.slam_scans(sa, free, 0, B, (400.0, 400.0, np.pi / 90), 60)
for rep in range(2):
    f = sa.Filter(1000, 1000, 1000, mode=capi.MODE_SLAM, device=0, seed=bench.SEED, map_storage=capi.MAPS_TILED, tile_pool_tiles=0, async_launch=True)
    f.collect_stats(False)
    odom, alphas = [0.0, 2.5, 0.0], [1e-3] * 4
    k = 0
    for _ in range(5):
        f.predict(odom, alphas); f.update(*scans[k], bench.STDDEV, bench.MAX_DIST, bench.OFFSET); k += 1
    f.synchronize()
    tp, tu = [], []
    t0 = time.perf_counter()
    for _ in range(40):
        a = time.perf_counter(); f.predict(odom, alphas); b = time.perf_counter(); f.update(*scans[k], bench.STDDEV, bench.MAX_DIST, bench.OFFSET); c = time.perf_counter(); k += 1
        tp.append((b - a) * 1e3); tu.append((c - b) * 1e3)
    host = (time.perf_counter() - t0) / 40 * 1e3
    f.synchronize()
    tot = (time.perf_counter() - t0) / 40 * 1e3
    print(f"rep {rep}: host {host:.3f} ms/step, total {tot:.3f} ms/step")
    print("  predict ms:", " ".join(f"{v:.2f}" for v in tp))
    print("  update  ms:", " ".join(f"{v:.2f}" for v in tu))
    f.close()
