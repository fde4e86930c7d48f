"""Time of one fake-lidar scan through the resident map handle (slam_map_*: rays up, one launch, hits down):
python tools/scan_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import slam_amd as sa
import _oracle as orc

for name, pose in (("floor_plan", [520.0, 280.0, 0.4]), ("maze", [1944.5, 2188.5, 1.57])):
    free = orc.load_map(name)
    gm = sa.GridMap(free.astype(np.uint8))
    for B in (90, 360, 1080):
        gm.fake_lidar_scan(pose, 0.0, 2 * np.pi, 500.0, B)
        t0 = time.perf_counter()
        for _ in range(200):
            h = gm.fake_lidar_scan(pose, 0.0, 2 * np.pi, 500.0, B)[0]
        dt = (time.perf_counter() - t0) / 200
        print(f"{name} {B} rays: {dt * 1e6:.0f} us per scan, {np.count_nonzero(h[:, 0] != -1)} hits, checksum {float(np.sum(h)):.6f}", flush=True)
    gm.close()
