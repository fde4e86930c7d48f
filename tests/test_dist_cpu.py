"""The N > 1 path on CPU: slam_amd.dist.ShardedFilter over `gloo`, world_size 2 (and 1), with the
per-shard compute supplied by the oracle.  Checks the collective logic and that the sharded result
is identical to the unsharded oracle filter (same global Philox streams, canonical blocked order)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
ODOM, ALPHAS, OFFSET = [2.5, 0.02, 0.02], [5e-4, 5e-4, 1e-2, 1e-2], [0.0, 30.0, 0.0]


def run_world(tmp_path, world, n_total, steps, mode="loc", wire="canvases"):
    out = tmp_path / f"w{world}{mode}"
    out.mkdir()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world + os.getpid() % 200),
           os.path.join(HERE, "_dist_worker.py"), str(out), str(n_total), str(steps), mode, wire]
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    parts = [np.load(out / f"rank{r}.npz") for r in range(world)]
    keys = ("x", "y", "th", "w", "idx", "avg", "best") + (("maps", "best_map") if mode == "slam" else ())
    return {k: np.concatenate([p[k] for p in parts]) for k in keys}


def check_read_out(got, ref, world, n_total, slam=False):
    """ShardedFilter.summarize() / best() on every rank against the unsharded oracle: slam::average_pose over all
    particles (slam/util.cpp:66-85, apps/grid_slam.cpp:119) and get_particles().front() (apps/grid_slam.cpp:112: the
    heaviest particle, ties to the lowest slot, which is where slam/mcl.cpp:236-238's sort puts it on one handle)."""
    want = orc.average_pose(ref["x"], ref["y"], ref["th"], math=orc.DET)
    assert got["avg"].shape == (world, 3) and got["best"].shape == (world, 6)
    for r in range(world):
        assert np.array_equal(got["avg"][r].view(np.uint64), got["avg"][0].view(np.uint64))     # the same bits on every rank
        assert np.array_equal(got["best"][r].view(np.uint64), got["best"][0].view(np.uint64))
    assert np.allclose(got["avg"][0], want, rtol=1e-12, atol=0)    # a different summation order (blocks, then ranks)
    k = int(np.flatnonzero(ref["w"] == ref["w"].max())[0])
    bw, slot, rank, bx, by, bth = got["best"][0]
    assert bw == ref["w"][k] and int(slot) == k and int(rank) == k // (n_total // world)
    assert (bx, by, bth) == (ref["x"][k], ref["y"][k], ref["th"][k])
    if slam:
        for r in range(world):
            assert np.array_equal(got["best_map"][r], ref["maps"][k])


def reference_run(n_total, steps):
    free = orc.load_map("floor_plan")
    grid = (free * 255).astype(np.uint8)
    rng = np.random.default_rng(5)
    x, y, th = 520 + 20 * rng.standard_normal(n_total), 280 + 20 * rng.standard_normal(n_total), 0.4 + 0.2 * rng.standard_normal(n_total)
    ang, dscan = orc.make_scan(free, [520.0, 280.0, 0.4], OFFSET, 45)
    idx = None
    for s in range(steps):
        x, y, th = orc.predict(x, y, th, ODOM, ALPHAS, orc.philox_normals(0xD157, s, 0, n_total), math=orc.DET)
        w, _ = orc.update_weights(x, y, th, grid, ang, dscan, 5.0, 500.0, OFFSET, math=orc.DET)
        idx, _, _ = orc.resample(w, orc.philox_uniforms(0xD157, s, 0, n_total), orc.ORDER_BLOCKED)
        x, y, th, w = x[idx], y[idx], th[idx], w[idx]
    return {"x": x, "y": y, "th": th, "w": w, "idx": idx}


@pytest.mark.parametrize("world,wire", [(1, "whole"), (2, "whole"), (4, "whole"), (2, "split")])
def test_sharded_filter_gloo_matches_unsharded(tmp_path, world, wire):
    """wire = "split": poses gathered after predict, weights after the weighting pass (two all-gathers)."""
    n_total, steps = 500, 3   # 250 per rank: not a multiple of the 64-chunks of the blocked running sum
    got = run_world(tmp_path, world, n_total, steps, wire=wire)
    ref = reference_run(n_total, steps)
    assert np.array_equal(got["idx"], ref["idx"])
    for k in ("x", "y", "th", "w"):
        assert np.array_equal(got[k].view(np.uint64), ref[k].view(np.uint64)), k
    check_read_out(got, ref, world, n_total)


def reference_run_slam(n_total, steps):
    import _dist_worker as dw

    rng = np.random.default_rng(6)
    x = dw.SLAM_POSE[0] + 2 * rng.standard_normal(n_total)
    y = dw.SLAM_POSE[1] + 2 * rng.standard_normal(n_total)
    th = dw.SLAM_POSE[2] + 0.1 * rng.standard_normal(n_total)
    maps = np.full((n_total, dw.SLAM_CANVAS, dw.SLAM_CANVAS), 128, np.uint8)
    ang, dscan = orc.make_scan(dw.slam_room(), dw.SLAM_POSE, dw.SLAM_OFFSET, dw.SLAM_BEAMS, max_dist=dw.SLAM_ZMAX)
    idx = None
    for s in range(steps):
        x, y, th = orc.predict(x, y, th, [0.0, 0.5, 0.0], ALPHAS, orc.philox_normals(0x51A4, s, 0, n_total), math=orc.DET)
        w, _ = orc.update_weights(x, y, th, maps, ang, dscan, 2.0, dw.SLAM_ZMAX, dw.SLAM_OFFSET, slam_mode=True, math=orc.DET)
        idx, _, _ = orc.resample(w, orc.philox_uniforms(0x51A4, s, 0, n_total), orc.ORDER_BLOCKED)
        x, y, th, w, maps = x[idx], y[idx], th[idx], w[idx], maps[idx].copy()
    return {"x": x, "y": y, "th": th, "w": w, "idx": idx, "maps": maps}


@pytest.mark.parametrize("world,wire", [(1, "canvases"), (2, "canvases"), (4, "canvases"), (1, "tiles"), (2, "tiles"), (4, "tiles")])
def test_sharded_slam_mode_gloo_matches_unsharded(tmp_path, world, wire):
    """SLAM mode: the canvases of the selected particles cross ranks (all-to-all) and every rank ends
    up with exactly the canvases the unsharded filter holds in those slots.  "tiles": the sparse wire
    form of tiled canvases (tables + distinct tiles, four all-to-alls)."""
    n_total, steps = 128, 3
    got = run_world(tmp_path, world, n_total, steps, mode="slam", wire=wire)
    ref = reference_run_slam(n_total, steps)
    assert np.array_equal(got["idx"], ref["idx"])
    for k in ("x", "y", "th", "w"):
        assert np.array_equal(got[k].view(np.uint64), ref[k].view(np.uint64)), k
    assert np.array_equal(got["maps"], ref["maps"])
    check_read_out(got, ref, world, n_total, slam=True)
    assert len(np.unique(ref["maps"].reshape(n_total, -1), axis=0)) > 1  # the canvases do differ
