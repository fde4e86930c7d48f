"""Replays a case saved by tools/soak_slam.py (gpurun_out/soak_case.npz) on the dense wave kernel and on tiled canvases,
with the hits of every ray recorded, and prints the first ray whose result differs.  python tools/defer_repro.py [case.npz]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_amd as sa
from slam_amd import capi
c = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "soak_case.npz"))
rows, cols, N, B, seed, pool = (int(v) for v in c["dims"]); zmax = float(c["zmax"])
mk = lambda **kw: sa.Filter(N, rows, cols, mode=capi.MODE_SLAM, seed=seed, **kw)
ref = mk(raycaster=capi.RAYCAST_FAST_BEAM_LANES, map_storage=capi.MAPS_DENSE)
til = mk(map_storage=capi.MAPS_TILED, tile_pool_tiles=pool)
for f in (ref, til):
    f.set_poses(c["x"], c["y"], c["th"]); f.record_hits(True)
print("case", rows, cols, N, B, zmax, "off", c["off"])
for step in range(len(c["ang"])):
    ang, dist, odom = c["ang"][step], c["dist"][step], c["odom"][step]
    for f in (ref, til):
        f.predict(list(odom), [1e-3, 1e-3, 1e-2, 1e-2])
    pre = ref.poses()
    for f in (ref, til):
        f.update(ang, dist, 5.0, zmax, list(c["off"]))
    ha, hb = ref.hits(), til.hits()
    same = all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(ha, hb))
    pa, pb = ref.prefix(), til.prefix()
    print("step", step, "hits equal", same, "prefix equal", np.array_equal(pa[0], pb[0]), "indices equal", np.array_equal(ref.last_indices(), til.last_indices()))
    if not same:
        xy_a, cell_a = np.asarray(ha[0]).reshape(N, B, 2), np.asarray(ha[1]).reshape(N, B)
        xy_b, cell_b = np.asarray(hb[0]).reshape(N, B, 2), np.asarray(hb[1]).reshape(N, B)
        pp, bb = np.nonzero(cell_a != cell_b)
        print("  rays that differ:", pp.size, "particles", np.unique(pp)[:10])
        for p, b in list(zip(pp, bb))[:6]:
            print("  particle", p, "pose", pre[0][p], pre[1][p], pre[2][p], "beam", b, "angle", ang[b], "z", dist[b],
                  "dense hit", xy_a[p, b], cell_a[p, b], divmod(int(cell_a[p, b]), cols), "tiled hit", xy_b[p, b], cell_b[p, b], divmod(int(cell_b[p, b]), cols))
        break
