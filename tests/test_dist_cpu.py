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


def run_world(tmp_path, world, n_total, steps):
    out = tmp_path / f"w{world}"
    out.mkdir()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world + os.getpid() % 200),
           os.path.join(HERE, "_dist_worker.py"), str(out), str(n_total), str(steps)]
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    parts = [np.load(out / f"rank{r}.npz") for r in range(world)]
    return {k: np.concatenate([p[k] for p in parts]) for k in ("x", "y", "th", "w", "idx")}


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


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_filter_gloo_matches_unsharded(tmp_path, world):
    n_total, steps = 512, 3
    got = run_world(tmp_path, world, n_total, steps)
    ref = reference_run(n_total, steps)
    assert np.array_equal(got["idx"], ref["idx"])
    for k in ("x", "y", "th", "w"):
        assert np.array_equal(got[k].view(np.uint64), ref[k].view(np.uint64)), k
