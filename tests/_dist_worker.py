"""Worker of tests/test_dist_cpu.py: runs slam_amd.dist.ShardedFilter over `gloo` with a CPU shard
backend built from the oracle (TEST ONLY — the product backend is HipShard).  Each rank writes its
final shard state to <out>/rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import _oracle as orc  # noqa: E402
from slam_amd.dist import ShardedFilter  # noqa: E402

ODOM, ALPHAS, OFFSET = [2.5, 0.02, 0.02], [5e-4, 5e-4, 1e-2, 1e-2], [0.0, 30.0, 0.0]


class OracleShard:
    """Same interface as slam_amd.dist.HipShard, computed by the CPU oracle (detmath mode)."""

    def __init__(self, n_local, rank, world, seed, grid, x, y, th):
        self.n, self.rank, self.world, self.seed, self.grid = n_local, rank, world, seed, grid
        self.first = rank * n_local
        self.x, self.y, self.th = x.copy(), y.copy(), th.copy()
        self.w = np.full(n_local, 1.0 / (world * n_local))
        self.pstep = self.rstep = 0
        self.last_idx = None

    def predict(self, odom, alphas):
        z = orc.philox_normals(self.seed, self.pstep, self.first, self.n)
        self.x, self.y, self.th = orc.predict(self.x, self.y, self.th, odom, alphas, z, math=orc.DET)
        self.pstep += 1

    def update_weights(self, angles, dists, stddev, max_dist, offset):
        if self.slam_mode:
            self.w, _ = orc.update_weights(self.x, self.y, self.th, self.maps, angles, dists, stddev, max_dist, offset,
                                           slam_mode=True, math=orc.DET)
            return
        self.w, _ = orc.update_weights(self.x, self.y, self.th, self.grid, angles, dists, stddev, max_dist, offset,
                                       math=orc.DET)

    def packed_state(self):
        return torch.from_numpy(np.concatenate([self.x, self.y, self.th, self.w]))

    def new_gather_buffer(self, world):
        return torch.empty(world * 4 * self.n, dtype=torch.float64)

    # read-out across ranks (HipShard.summary / pose_of / map_of): slam/util.cpp:66-85 and the front of slam/mcl.cpp:236-238
    def summary(self):
        c, s = orc.detmath(1, self.th), orc.detmath(0, self.th)
        k = int(np.flatnonzero(self.w == self.w.max())[0])
        return torch.tensor([self.x.sum(), self.y.sum(), c.sum(), s.sum(), float(self.n), self.w[k], float(self.first + k), float(k)],
                            dtype=torch.float64)

    def pose_of(self, k):
        return float(self.x[k]), float(self.y[k]), float(self.th[k])

    def map_of(self, k):
        return torch.from_numpy(self.maps[k].copy())

    def canvas_shape(self):
        return self.maps.shape[1:]

    # SLAM mode (per-particle canvases): same three pieces as HipShard
    slam_mode = False

    def enable_slam(self, rows, cols):
        self.slam_mode = True
        self.n_local = self.n
        self.maps = np.full((self.n, rows, cols), 128, np.uint8)  # slam/mcl.cpp:30

    def shard_select(self, gathered, world):
        self.shard_resample(gathered, world)
        return torch.from_numpy(self.last_idx.astype(np.int64))

    def pack_maps(self, local_idx):
        return torch.from_numpy(self.maps[local_idx.numpy()].reshape(local_idx.numel(), -1).copy())

    def clone_maps(self, pool, pos):
        self.maps = pool.numpy()[pos.numpy()].reshape((self.n,) + self.maps.shape[1:]).copy()

    # the sparse wire form of tiled canvases (slam_amd.dist, `tiled_wire`), emulated on dense canvases:
    # 64x64 tiles, blank (all 128) = entry 0, equal tiles of one request travel once
    tiled_wire = False

    def _as_tiles(self, canv):
        c, rows, cols = canv.shape
        tr, tc = (rows + 63) // 64, (cols + 63) // 64
        pad = np.full((c, tr * 64, tc * 64), 128, np.uint8)
        pad[:, :rows, :cols] = canv
        return pad.reshape(c, tr, 64, tc, 64).transpose(0, 1, 3, 2, 4).reshape(c, tr * tc, 4096)

    def pack_tiles(self, local_idx):
        t = self._as_tiles(self.maps[local_idx.numpy()])
        tabs = np.zeros(t.shape[:2], np.int32)
        full = ~(t == 128).all(axis=2)
        if full.any():
            uniq, inv = np.unique(t[full], axis=0, return_inverse=True)
            tabs[full] = inv.reshape(-1) + 1
        else:
            uniq = np.zeros((0, 4096), np.uint8)
        return torch.from_numpy(tabs), torch.from_numpy(np.ascontiguousarray(uniq))

    def import_tiles(self, tabs, tiles, pos):
        rows, cols = self.maps.shape[1:]
        tr, tc = (rows + 63) // 64, (cols + 63) // 64
        payload = np.concatenate([np.full((1, 4096), 128, np.uint8), tiles.numpy()])
        t = payload[tabs.numpy()[pos.numpy()]]                      # (n, T, 4096)
        canv = t.reshape(self.n, tr, tc, 64, 64).transpose(0, 1, 3, 2, 4).reshape(self.n, tr * 64, tc * 64)
        self.maps = np.ascontiguousarray(canv[:, :rows, :cols])

    def shard_resample_split(self, gathered_poses, gathered_weights, world):
        g = torch.cat([gathered_poses.reshape(world, 3, self.n), gathered_weights.reshape(world, 1, self.n)], dim=1)
        self.shard_resample(g.reshape(-1), world)

    def shard_resample(self, gathered, world):
        g = gathered.numpy().reshape(world, 4, self.n)
        allx, ally, allth, allw = (g[:, q, :].reshape(-1) for q in range(4))
        N = world * self.n
        u = np.zeros(N)
        u[self.first:self.first + self.n] = orc.philox_uniforms(self.seed, self.rstep, self.first, self.n)
        idx, _, _ = orc.resample(allw, u, orc.ORDER_BLOCKED)
        idx = idx[self.first:self.first + self.n]
        self.x, self.y, self.th, self.w = allx[idx].copy(), ally[idx].copy(), allth[idx].copy(), allw[idx].copy()
        self.last_idx = idx
        self.rstep += 1


SLAM_CANVAS, SLAM_POSE, SLAM_OFFSET, SLAM_BEAMS, SLAM_ZMAX = 96, [48.0, 44.0, 0.3], [0.0, 6.0, 0.0], 24, 40.0


def slam_room():
    """Ground truth of the SLAM-mode case: a walled room with a pillar (1 = free)."""
    free = np.ones((SLAM_CANVAS, SLAM_CANVAS), np.int32)
    free[:3, :] = free[-3:, :] = free[:, :3] = free[:, -3:] = 0
    free[60:70, 20:30] = 0
    return free


def main_slam(out, n_total, steps):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_loc = n_total // world
    rng = np.random.default_rng(6)
    x = SLAM_POSE[0] + 2 * rng.standard_normal(n_total)
    y = SLAM_POSE[1] + 2 * rng.standard_normal(n_total)
    th = SLAM_POSE[2] + 0.1 * rng.standard_normal(n_total)
    sl = slice(rank * n_loc, (rank + 1) * n_loc)
    shard = OracleShard(n_loc, rank, world, 0x51A4, None, x[sl], y[sl], th[sl])
    shard.enable_slam(SLAM_CANVAS, SLAM_CANVAS)
    shard.tiled_wire = len(sys.argv) > 5 and sys.argv[5] == "tiles"
    sf = ShardedFilter(shard, world=world)
    ang, dscan = orc.make_scan(slam_room(), SLAM_POSE, SLAM_OFFSET, SLAM_BEAMS, max_dist=SLAM_ZMAX)
    for _ in range(steps):
        sf.predict([0.0, 0.5, 0.0], ALPHAS)
        sf.update(ang, dscan, 2.0, SLAM_ZMAX, SLAM_OFFSET)
    avg, _ = sf.summarize()
    best = sf.best(with_map=True)
    np.savez(os.path.join(out, f"rank{rank}.npz"), x=shard.x, y=shard.y, th=shard.th, w=shard.w, idx=shard.last_idx,
             maps=shard.maps, avg=np.array(avg)[None], best=np.array([[best["weight"], best["global_slot"], best["rank"], best["x"],
                                                                      best["y"], best["theta"]]]), best_map=best["map"][None])
    dist.barrier()
    dist.destroy_process_group()


def main():
    out, n_total, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4 and sys.argv[4] == "slam":
        return main_slam(out, n_total, steps)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_loc = n_total // world
    free = orc.load_map("floor_plan")
    grid = (free * 255).astype(np.uint8)
    rng = np.random.default_rng(5)
    x, y, th = 520 + 20 * rng.standard_normal(n_total), 280 + 20 * rng.standard_normal(n_total), 0.4 + 0.2 * rng.standard_normal(n_total)
    sl = slice(rank * n_loc, (rank + 1) * n_loc)
    shard = OracleShard(n_loc, rank, world, 0xD157, grid, x[sl], y[sl], th[sl])
    sf = ShardedFilter(shard, world=world, split_exchange=len(sys.argv) > 5 and sys.argv[5] == "split")
    ang, dscan = orc.make_scan(free, [520.0, 280.0, 0.4], OFFSET, 45)
    for _ in range(steps):
        sf.predict(ODOM, ALPHAS)
        sf.update(ang, dscan, 5.0, 500.0, OFFSET)
    avg, _ = sf.summarize()
    best = sf.best()
    np.savez(os.path.join(out, f"rank{rank}.npz"), x=shard.x, y=shard.y, th=shard.th, w=shard.w, idx=shard.last_idx,
             avg=np.array(avg)[None], best=np.array([[best["weight"], best["global_slot"], best["rank"], best["x"], best["y"],
                                                      best["theta"]]]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
