"""Grid SLAM on tiled canvases: "weights now, canvases of the survivors later" (TiledDefer / k_weights_S2F).
An update that is resampled right away may leave every canvas as it is, weigh the particles against private LDS windows,
and draw the scan only into the canvases the resample selects.  Nothing observable may depend on it: weights, selection
and canvases are compared with the dense two-phase kernel (itself pinned to the oracle in test_gpu_parity.py), bit for bit
(slam/mcl.cpp:102-144 draws every canvas, slam/mcl.cpp:147-203 then keeps the selected ones)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def sa():
    import slam_amd

    return slam_amd


def room_scan(B, step):
    ang = -np.pi + 2 * np.pi * np.arange(B) / B
    dist = 38.0 + 9.0 * np.sin(3 * ang + 0.2 * step) + 4.0 * np.cos(7 * ang)
    dist[::11] = 75.0  # no return: free space up to max_dist
    return ang, dist


@pytest.mark.parametrize("corner", ["inside", "top_left"])
def test_deferred_updates_change_nothing(sa, corner):
    """8192 particles (the smallest filter that defers by itself), default settings.  "top_left" puts the cloud next to row 0
    and column 0, whose cells are two units wide (the casts of slam/raycast.cpp:171 truncate toward zero): part of the cloud
    must then take the ordinary pass inside a deferred update (slam_defer_safe)."""
    capi = sa.capi
    N, B, rows, cols, zmax = 8192, 36, 160, 200, 75.0
    rng = np.random.default_rng(11)
    cx, cy = (100.0, 80.0) if corner == "inside" else (30.0, rows - 30.0)
    x, y, th = cx + 12.0 * rng.standard_normal(N), cy + 12.0 * rng.standard_normal(N), rng.uniform(-3, 3, N)
    dense = sa.Filter(N, rows, cols, mode=capi.MODE_SLAM, seed=5, raycaster=capi.RAYCAST_TWO_PHASE, map_storage=capi.MAPS_DENSE)
    tiled = sa.Filter(N, rows, cols, mode=capi.MODE_SLAM, seed=5, map_storage=capi.MAPS_TILED)
    for f in (dense, tiled):
        f.set_poses(x, y, th)
    edge_seen = 0
    for step in range(7):
        ang, dist = room_scan(B, step)
        odom = [0.02, 1.5, -0.01]
        for f in (dense, tiled):
            f.predict(odom, [1e-3, 1e-3, 1e-2, 1e-2])
            f.update(ang, dist, 5.0, zmax, [3.0, 1.0, 0.1])
        assert np.array_equal(tiled.prefix()[0].view(np.uint64), dense.prefix()[0].view(np.uint64)), step
        assert np.array_equal(tiled.last_indices(), dense.last_indices()), step
        assert np.array_equal(tiled.weights().view(np.uint64), dense.weights().view(np.uint64)), step
        for p in rng.choice(N, 24, replace=False):
            assert np.array_equal(tiled.map_of(int(p)), dense.map_of(int(p))), (step, p)
        edge_seen = max(edge_seen, tiled.defer_stats()["edge_particles"])
    st = tiled.defer_stats()
    assert st["updates"] == 7 and st["deferred"] >= 3, st  # (the first update cannot defer: no resample has been seen yet)
    if corner == "top_left":
        assert 0 < edge_seen < N, edge_seen
    else:
        assert edge_seen == 0
    for p in range(0, N, 97):
        assert np.array_equal(tiled.map_of(p), dense.map_of(p)), p
    dense.close()
    tiled.close()


def test_forced_deferral_soak_in_a_child_process():
    """SLAM_DEFER=2 defers every update that can be (any particle count); tools/soak_slam.py with evenly spaced beams then
    runs 400 random small cases (canvas sizes, clouds partly off the canvas, 1-200 beams, 1-5 steps) on five kernel shapes
    and compares every canvas.  The library reads SLAM_DEFER once, hence the child process."""
    env = dict(os.environ, SLAM_DEFER="2", SOAK_EVEN="1", SOAK_ALL="1", SOAK_CASES="400")
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "soak_slam.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "slam soak ok: 400 cases" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
