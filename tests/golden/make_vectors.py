"""Generates tests/golden/oracle_vectors.npz: frozen outputs of the CPU oracle (oracle/oracle.cpp).

The reference has no golden vectors and cannot be built here, so these are NOT reference outputs:
they freeze the oracle (a statement-by-statement restatement of the reference) so that any later
drift of the oracle, of detmath.h or of the HIP kernels shows up as a test failure.  `det_*` arrays
were produced with the deterministic math of slam_amd/csrc/detmath.h and are compared bit for bit;
`libm_*` arrays with glibc (the reference's own calls) and are compared with a tolerance because
glibc's last bit depends on the CPU it dispatches for.
Run from the repo root:  python tests/golden/make_vectors.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as orc  # noqa: E402

ODOM, ALPHAS, OFFSET = [2.5, 0.02, 0.02], [5e-4, 5e-4, 1e-2, 1e-2], [0.0, 30.0, 0.0]


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def build():
    rng = np.random.default_rng(20261111)
    out = {}
    # G7 update tables
    out["g7_free"], out["g7_occ"] = orc.update_tables()
    # G8 sensor position / offset truncation
    offs = np.array([[0, 30, 0], [0.5, 0.5, 0], [3, 4.9, 0.2], [-12.5, 7.25, -1.0], [0, 0, 0.3]], float)
    frames = np.array([[400, 400, np.pi], [10.5, 20.25, 0.3], [0, 0, -2.0], [999, 1, 1.5], [5, 5, 0]], float)
    out["g8_off"], out["g8_frame"] = offs, frames
    out["g8_dtheta"] = np.array([orc.scanner_offset(o) for o in offs])
    out["g8_det_sensor"] = np.array([orc.sensor_position(f, o, orc.DET) for f, o in zip(frames, offs)])
    # G1 motion
    N = 256
    x, y, th = rng.uniform(0, 1000, N), rng.uniform(0, 1000, N), rng.uniform(-2 * np.pi, np.pi, N)
    z = orc.reference_normals(3 * N, tick0=1, tick_stride=1)
    out["g1_in"] = np.stack([x, y, th])
    out["g1_z"] = z
    out["g1_det_out"] = np.stack(orc.predict(x, y, th, ODOM, ALPHAS, z, orc.DET))
    out["g1_libm_out"] = np.stack(orc.predict(x, y, th, ODOM, ALPHAS, z, orc.LIBM))
    # G2 single rays on a small canvas with walls, previously drawn cells and edge cases
    g = np.full((96, 128), 128, np.uint8)
    g[20:70, 90] = 0
    g[70, 30:91] = 0
    g[5:15, 5:15] = 255
    g[40:60, 40:60] = 60
    rays = []
    for k in range(160):
        rays.append([rng.uniform(-4, 132), rng.uniform(-4, 100), rng.uniform(-4, 4), rng.uniform(2, 90)])
    rays += [[-0.5, 50.5, 0.0, 30.0], [64.0, 48.0, np.pi / 4, 200.0], [64.5, 48.5, np.pi / 2, 90.0], [500.0, 500.0, 1.0, 10.0],
             [91.5, 40.5, np.pi, 60.0], [10.5, 85.5, 0.0, 40.0]]
    rays = np.array(rays)
    zmax = 90.0
    gd = g.copy()
    hits, cells, stats = [], [], []
    for (px, py, t, zz) in rays:
        h, c, st = orc.raycast_mapping(gd, px, py, t, min(zz, zmax) ** 2, zmax ** 2, math=orc.DET)  # canvas evolves ray by ray
        hits.append(h), cells.append(c), stats.append([st["cells_read"], st["cells_written"], st["steps"]])
    out["g2_canvas"], out["g2_rays"], out["g2_zmax"] = g, rays, np.float64(zmax)
    out["g2_det_hits"], out["g2_det_cells"], out["g2_det_stats"] = np.array(hits), np.array(cells, np.int32), np.array(stats)
    out["g2_det_canvas_after"] = gd
    # G6 FakeLidar scans on the three asset grids
    for name, pose in (("map", [400.0, 400.0, np.pi / 90]), ("floor_plan", [520.0, 280.0, 0.4]), ("maze", [1200.0, 1100.0, -1.0])):
        free = orc.load_map(name)
        sensor = orc.sensor_position(pose, OFFSET, orc.DET)
        out[f"g6_{name}_sensor"] = sensor
        out[f"g6_{name}_det_hits"] = orc.fake_lidar_scan(free.astype(np.int32), sensor, 0.0, 2 * np.pi, 500.0, 90, orc.DET)
    # G3 SLAM-mode update: N = 16, B = 90, 200 x 200 canvas, two passes
    free = orc.load_map("map")
    ang, dist = orc.make_scan(free, [400.0, 400.0, np.pi / 90], OFFSET, 90, math=orc.DET)
    out["scan_ang"], out["scan_dist"] = ang, dist
    N = 16
    x, y, th = 100 + 4 * rng.standard_normal(N), 100 + 4 * rng.standard_normal(N), np.pi / 2 + 0.3 * rng.standard_normal(N)
    maps = np.full((N, 200, 200), 128, np.uint8)
    out["g3_in"] = np.stack([x, y, th])
    for p in range(2):
        w, _ = orc.update_weights(x, y, th, maps, ang, dist, 5.0, 500.0, OFFSET, slam_mode=True, math=orc.DET)
        out[f"g3_det_w{p}"] = w
        out[f"g3_det_crc{p}"] = np.array([crc(m) for m in maps])
    # GL localization weights on floor_plan
    freef = orc.load_map("floor_plan")
    gf = (freef * 255).astype(np.uint8)
    angl, distl = orc.make_scan(freef, [520.0, 280.0, 0.4], OFFSET, 120, math=orc.DET)
    N = 96
    x, y, th = 520 + 15 * rng.standard_normal(N), 280 + 15 * rng.standard_normal(N), 0.4 + 0.15 * rng.standard_normal(N)
    x[:4], y[:4] = [-10.0, 1300.0, 600.0, 3.5], [50.0, 300.0, 610.0, 2.5]
    out["gl_in"], out["gl_ang"], out["gl_dist"] = np.stack([x, y, th]), angl, distl
    w, hxy, hc, st = orc.update_weights(x, y, th, gf, angl, distl, 5.0, 500.0, OFFSET, want_hits=True, math=orc.DET)
    out["gl_det_w"], out["gl_det_hit_xy"], out["gl_det_hit_cell"] = w, hxy, hc
    out["gl_det_stats"] = np.array([st["cells_read"], st["cells_written"], st["steps"]])
    wl, _, hcl, _ = orc.update_weights(x, y, th, gf, angl, distl, 5.0, 500.0, OFFSET, want_hits=True, math=orc.LIBM)
    out["gl_libm_w"], out["gl_libm_hit_cell"] = wl, hcl
    # G4 resampling
    cases = {"generic": rng.uniform(0, 1, 300) ** 5, "zeros": np.zeros(33), "single": np.eye(1, 77, 30).ravel() * 3.5,
             "repeats": np.repeat([0.0, 0.5, 0.0, 0.25, 0.0], 13), "one": np.array([0.7]),
             "big": rng.uniform(0, 1, 5000) * (rng.uniform(size=5000) > 0.3)}
    for name, w in cases.items():
        u = orc.reference_uniforms(w.size, tick=w.size + 5)
        out[f"g4_{name}_w"], out[f"g4_{name}_u"] = w, u
        for order, tag in ((orc.ORDER_REFERENCE, "ref"), (orc.ORDER_BLOCKED, "blk")):
            idx, P, s = orc.resample(w, u, order)
            out[f"g4_{name}_{tag}_idx"], out[f"g4_{name}_{tag}_prefix"] = idx, P
        out[f"g4_{name}_perm"] = orc.sort_order(w[out[f"g4_{name}_blk_idx"]], 0)
    # G5 end-to-end SLAM loop, benchmark/mcl.cpp sequence at N = 25, canvas 240 x 180, 5 steps
    N, rows, cols = 25, 180, 240
    st_ = [np.full(N, cols / 2.0), np.full(N, rows / 2.0), np.full(N, np.pi / 2)]
    maps = np.full((N, rows, cols), 128, np.uint8)
    sensor = orc.sensor_position([400.0, 400.0, np.pi], OFFSET, orc.DET)
    hits = orc.fake_lidar_scan(orc.load_map("floor_plan").astype(np.int32), sensor, 0.0, np.pi, 500.0, 90, orc.DET)
    a5, d5 = orc.hits_to_scan(hits, sensor, 0.0, np.pi, 500.0)
    out["g5_ang"], out["g5_dist"] = a5, d5
    for step in range(5):
        z = orc.reference_normals(3 * N, tick0=50 * step + 1, tick_stride=1)
        u = orc.reference_uniforms(N, tick=900 + step)
        x, y, th = orc.predict(*st_, ODOM, ALPHAS, z, orc.DET)
        w, _ = orc.update_weights(x, y, th, maps, a5, d5, 5.0, 500.0, OFFSET, slam_mode=True, math=orc.DET)
        idx, _, _ = orc.resample(w, u, orc.ORDER_BLOCKED)
        src = idx[orc.sort_order(w[idx], 0)]
        maps[:] = maps[src]
        st_ = [x[src], y[src], th[src]]
        out[f"g5_det_idx{step}"] = idx
        out[f"g5_det_state{step}"] = np.stack(st_ + [w[src]])
        out[f"g5_det_crc{step}"] = np.array([crc(m) for m in maps])
    return out


if __name__ == "__main__":
    v = build()
    path = os.path.join(HERE, "oracle_vectors.npz")
    np.savez_compressed(path, **v)
    print(path, os.path.getsize(path), "bytes,", len(v), "arrays")
