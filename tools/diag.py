import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import slam_amd as sa
N, B = 100000, 360
free = bench.load_free_grid("floor_plan"); grid = (free * 255).astype(np.uint8); rows, cols = grid.shape
ang, dist = bench.synth_scan(sa, free, 0, B)
scans = bench.synth_scans(sa, free, 0, B, bench.true_trajectory(8))
print("scan: hits", np.sum(dist < 500), "mean dist", dist.mean())
f = sa.Filter(N, rows, cols, seed=bench.SEED); f.set_map(grid); f.seed_gaussian(bench.TRUE_POSE, [25.0, 25.0, 0.2])
for step in range(8):
    f.predict(bench.ODOM, bench.ALPHAS)
    x, y, th = f.poses()
    ang, dist = scans[step]
    f.update_weights(ang, dist, bench.STDDEV, bench.MAX_DIST, bench.OFFSET)
    w = f.weights()
    f.resample()
    idx = f.last_indices()
    print(f"step {step}: mean=({x.mean():.1f},{y.mean():.1f},{th.mean():.3f}) std=({x.std():.2f},{y.std():.2f},{th.std():.3f}) zero_w={np.mean(w==0):.3f} max_w={w.max():.3e} log10(sum)={np.log10(max(w.sum(),1e-320)):.1f} unique_src={np.unique(idx).size}")
