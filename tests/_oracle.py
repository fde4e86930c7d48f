"""ctypes wrapper of oracle/_build/liboracle.so — the CPU restatement of the reference, used by the
tests (and by bench.py's cpu_baseline leg) only as the checker.  Never imported by slam_amd."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIBM, DET = 0, 1
ORDER_REFERENCE, ORDER_BLOCKED = 0, 1

_P = ctypes.c_void_p
_libs = {}


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def lib(fast=False):
    name = "liboracle_ofast.so" if fast else "liboracle.so"
    if name not in _libs:
        path = os.path.join(ORACLE_DIR, "_build", name)
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_update_weights.restype = ctypes.c_int
        L.orc_raycast_mapping.restype = ctypes.c_int
        L.orc_fake_lidar_scan.restype = ctypes.c_int
        L.orc_raycast.restype = ctypes.c_int
        L.orc_measurement_model_beam.restype = ctypes.c_int
        _libs[name] = L
    return _libs[name]


def _p(a):
    return a.ctypes.data_as(_P) if a is not None else None


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def update_tables():
    fr = np.empty(256, np.uint8)
    oc = np.empty(256, np.uint8)
    lib().orc_update_tables(_p(fr), _p(oc))
    return fr, oc


def raycast_mapping(grid, px, py, theta, z2, zmax2, step=1.0, write=True, math=LIBM):
    """One ray; `grid` (uint8, C-contiguous) is modified in place when write."""
    assert grid.dtype == np.uint8 and grid.flags.c_contiguous
    hit = np.empty(2)
    cell = ctypes.c_int()
    st = (ctypes.c_long * 3)()
    rc = lib().orc_raycast_mapping(_p(grid), grid.shape[0], grid.shape[1], ctypes.c_double(px), ctypes.c_double(py),
                                   ctypes.c_double(theta), ctypes.c_double(z2), ctypes.c_double(zmax2),
                                   ctypes.c_double(step), int(write), math, _p(hit), ctypes.byref(cell), st)
    assert rc == 0
    return hit, cell.value, {"cells_read": st[0], "cells_written": st[1], "steps": st[2]}


def measurement_model_beam(grid, px, py, theta, d2, stddev, max2, write=True, math=LIBM):
    w = ctypes.c_double()
    rc = lib().orc_measurement_model_beam(_p(grid), grid.shape[0], grid.shape[1], ctypes.c_double(px),
                                          ctypes.c_double(py), ctypes.c_double(theta), ctypes.c_double(d2),
                                          ctypes.c_double(stddev), ctypes.c_double(max2), int(write), math,
                                          ctypes.byref(w))
    assert rc == 0
    return w.value


def update_weights(x, y, th, grids, angles, dists, stddev, max_dist, offset=(0, 0, 0), slam_mode=False, math=LIBM,
                   n_threads=1, want_hits=False, fast=False, log_domain=False):
    """The weighting half of MCL::update.  slam_mode: grids is (N, rows, cols), updated in place.
    Otherwise grids is the shared (rows, cols) map."""
    x, y, th = _f(x), _f(y), _f(th)
    N = x.size
    angles, dists, offset = _f(angles), _f(dists), _f(offset)
    B = angles.size
    assert grids.dtype == np.uint8 and grids.flags.c_contiguous
    rows, cols = grids.shape[-2:]
    if slam_mode:
        assert grids.shape[0] == N
    w = np.empty(N)
    hit_xy = np.empty((N, B, 2)) if want_hits else None
    hit_cell = np.empty((N, B), np.int32) if want_hits else None
    st = (ctypes.c_long * 3)()
    rc = lib(fast).orc_update_weights(int(slam_mode), ctypes.c_long(N), _p(x), _p(y), _p(th), _p(grids), rows, cols, B,
                                      _p(angles), _p(dists), ctypes.c_double(stddev), ctypes.c_double(max_dist),
                                      _p(offset), math | (2 if log_domain else 0), n_threads, _p(w), _p(hit_xy),
                                      _p(hit_cell), st)
    assert rc == 0
    stats = {"cells_read": st[0], "cells_written": st[1], "steps": st[2]}
    return (w, hit_xy, hit_cell, stats) if want_hits else (w, stats)


def log_domain_weights(lw, math=LIBM):
    """exp(lw - max lw): what the selection runs on in the log-domain mode."""
    lw = _f(lw)
    out = np.empty_like(lw)
    lib().orc_log_domain_weights(ctypes.c_long(lw.size), _p(lw), math, _p(out))
    return out


def predict(x, y, th, odom, alphas, z, math=LIBM):
    x, y, th = _f(x).copy(), _f(y).copy(), _f(th).copy()
    odom, alphas, z = _f(odom), _f(alphas), _f(z)
    assert z.size == 3 * x.size
    lib().orc_predict(ctypes.c_long(x.size), _p(x), _p(y), _p(th), _p(odom), _p(alphas), _p(z), math)
    return x, y, th


def predict_velocity(x, y, th, vel, dt, alphas, z, math=LIBM):
    x, y, th = _f(x).copy(), _f(y).copy(), _f(th).copy()
    vel, alphas, z = _f(vel), _f(alphas), _f(z)
    assert z.size == 3 * x.size and vel.size == 2 and alphas.size == 6
    lib().orc_predict_velocity(ctypes.c_long(x.size), _p(x), _p(y), _p(th), _p(vel), ctypes.c_double(dt), _p(alphas), _p(z), math)
    return x, y, th


def reference_triangular(n, tick0=1, tick_stride=1, stddev=1.0):
    out = np.empty(n)
    lib().orc_reference_triangular(ctypes.c_long(n), ctypes.c_long(tick0), ctypes.c_long(tick_stride),
                                   ctypes.c_double(stddev), _p(out))
    return out


def pdf_triangular(stddev, x):
    x = _f(x)
    out = np.empty_like(x)
    lib().orc_pdf_triangular(ctypes.c_long(x.size), ctypes.c_double(stddev), _p(x), _p(out))
    return out


def reference_normals(n, tick0=1, tick_stride=1, stddev=1.0):
    out = np.empty(n)
    lib().orc_reference_normals(ctypes.c_long(n), ctypes.c_long(tick0), ctypes.c_long(tick_stride),
                                ctypes.c_double(stddev), _p(out))
    return out


def reference_uniforms(n, tick=1):
    out = np.empty(n)
    lib().orc_reference_uniforms(ctypes.c_long(n), ctypes.c_long(tick), _p(out))
    return out


def philox_normals(seed, step, first, n):
    z = np.empty(3 * n)
    lib().orc_philox_normals(ctypes.c_uint64(seed), ctypes.c_uint64(step), ctypes.c_long(first), ctypes.c_long(n), _p(z))
    return z


def philox_uniforms(seed, step, first, n):
    u = np.empty(n)
    lib().orc_philox_uniforms(ctypes.c_uint64(seed), ctypes.c_uint64(step), ctypes.c_long(first), ctypes.c_long(n), _p(u))
    return u


def seed_uniform_free(seed, first, n, grid):
    g = np.ascontiguousarray(grid, dtype=np.uint8)
    x, y, th = np.empty(n), np.empty(n), np.empty(n)
    lib().orc_seed_uniform_free(ctypes.c_uint64(seed), ctypes.c_long(first), ctypes.c_long(n), _p(g), g.shape[0], g.shape[1],
                                _p(x), _p(y), _p(th))
    return x, y, th


def seed_gaussian(seed, first, n, mean, sigma):
    x, y, th = np.empty(n), np.empty(n), np.empty(n)
    lib().orc_seed_gaussian(ctypes.c_uint64(seed), ctypes.c_long(first), ctypes.c_long(n), _p(_f(mean)), _p(_f(sigma)),
                            _p(x), _p(y), _p(th))
    return x, y, th


def resample(w, canon, order=ORDER_REFERENCE, resampler=0):
    w, canon = _f(w), _f(canon)
    N = w.size
    P = np.empty(N)
    idx = np.empty(N, np.int32)
    s = ctypes.c_double()
    lib().orc_resample(ctypes.c_long(N), _p(w), _p(canon), order, resampler, _p(P), _p(idx), ctypes.byref(s))
    return idx, P, s.value


def sort_order(w_sel, mode=0):
    w_sel = _f(w_sel)
    perm = np.empty(w_sel.size, np.int32)
    lib().orc_sort_order(ctypes.c_long(w_sel.size), _p(w_sel), mode, _p(perm))
    return perm


def sensor_position(frame, offset, math=LIBM):
    out = np.empty(3)
    lib().orc_sensor_position(_p(_f(frame)), _p(_f(offset)), math, _p(out))
    return out


def scanner_offset(offset):
    out = np.empty(2)
    lib().orc_scanner_offset(_p(_f(offset)), _p(out))
    return out


def average_pose(x, y, th, math=LIBM):
    x, y, th = _f(x), _f(y), _f(th)
    out = np.empty(3)
    lib().orc_average_pose(ctypes.c_long(x.size), _p(x), _p(y), _p(th), math, _p(out))
    return out


def fake_lidar_scan(grid_i32, pose, start, stop, max_dist, n_rays, math=LIBM):
    g = np.ascontiguousarray(grid_i32, dtype=np.int32)
    hits = np.empty((n_rays, 2))
    rc = lib().orc_fake_lidar_scan(_p(g), g.shape[0], g.shape[1], _p(_f(pose)), ctypes.c_double(start),
                                   ctypes.c_double(stop), ctypes.c_double(max_dist), n_rays, math, _p(hits))
    assert rc == 0
    return hits


def hits_to_scan(hits, sensor, start, stop, max_dist):
    hits = _f(hits)
    n = hits.shape[0]
    a, d = np.empty(n), np.empty(n)
    lib().orc_hits_to_scan(n, _p(hits), _p(_f(sensor)), ctypes.c_double(start), ctypes.c_double(stop),
                           ctypes.c_double(max_dist), _p(a), _p(d))
    return a, d


def raycast(kind, grid, px, py, theta, max2, step=0.5, math=LIBM):
    """kind 0: float64 map (<0.5), 1: uint8 (<128), 2: int32 (==0)."""
    dt = {0: np.float64, 1: np.uint8, 2: np.int32}[kind]
    g = np.ascontiguousarray(grid, dtype=dt)
    hit = np.empty(2)
    rc = lib().orc_raycast(kind, _p(g), g.shape[0], g.shape[1], ctypes.c_double(px), ctypes.c_double(py),
                           ctypes.c_double(theta), ctypes.c_double(max2), ctypes.c_double(step), math, _p(hit))
    assert rc == 0
    return hit


def detmath(fn, x):
    x = _f(x)
    out = np.empty_like(x)
    lib().orc_detmath(fn, ctypes.c_long(x.size), _p(x), _p(out))
    return out


def libm(fn, x):
    x = _f(x)
    out = np.empty_like(x)
    lib().orc_libm(fn, ctypes.c_long(x.size), _p(x), _p(out))
    return out


# ---- scenario helpers shared by tests, smoke() and bench.py ----------------------------------------
def load_map(name):
    """Binary grid (1 = free, 0 = wall) from tests/golden/maps.npz."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "maps.npz"))
    sh = z[name + "_shape"]
    return np.unpackbits(z[name + "_bits"])[: int(sh[0]) * int(sh[1])].reshape(int(sh[0]), int(sh[1]))


def make_scan(free01, true_pose, offset, n_rays, start=0.0, stop=2 * np.pi, max_dist=500.0, math=LIBM):
    """The callers' scan synthesis (apps/grid_slam.cpp:144-166): sensor pose -> FakeLidar -> (angle, dist)."""
    sensor = sensor_position(true_pose, offset, math)
    hits = fake_lidar_scan(free01.astype(np.int32), sensor, start, stop, max_dist, n_rays, math)
    return hits_to_scan(hits, sensor, start, stop, max_dist)
