"""Time of Filter.set_map (raw map upload + clearance planes) on the synthetic grids: python tools/setmap_time.py [sizes...]
SLAM_CONE_BUILD=0 builds the planes on the host (clearance.h) and uploads them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import slam_amd as sa
from slam_amd import synth

for size in [int(a) for a in sys.argv[1:]] or [1024, 4096, 8192, 16384]:
    g = synth.make_grid(size)
    f = sa.Filter(64, size, size)
    f.set_map(g)  # first call: allocations
    t0 = time.perf_counter()
    f.set_map(g)
    dt = time.perf_counter() - t0
    print(f"{size}x{size}: set_map {dt * 1e3:.1f} ms (SLAM_CONE_BUILD={os.environ.get('SLAM_CONE_BUILD', '1')})", flush=True)
    f.close()
