"""One-process-per-GPU sharded filter: particles are split into contiguous shards,
one per rank; predict and weighting are local; resampling is global and needs ONE exchange step, an
all-gather of the packed per-shard state (32 B per particle) over RCCL/xGMI (`nccl` backend).
`split_exchange=True` splits it: the poses (24 B) are final after predict, so their all-gather is
issued on a side stream and can run under the weighting pass, and only the weights (8 B) are
gathered on the critical path (off by default: whether the side stream really overlaps depends on
how the runtime maps streams to hardware queues).  Every
rank then rebuilds the running sum of all weights in the canonical blocked order, draws its own
output slots and copies the selected particles out of the gathered buffer.  The result is
bit-identical for any number of ranks (tests/test_dist_cpu.py, tests/test_gpu_parity.py).
In SLAM mode (per-particle canvases) the selected canvases are exchanged as well: one all-to-all of
whole canvases, each distinct (source, destination rank) pair travelling once.  Shards that hold
copy-on-write tiled canvases exchange tables and distinct tiles instead (`tiled_wire`).

The per-shard compute is behind a small backend interface so that the collective logic can be
exercised with `gloo` on CPU tensors in the tests; the product backend is HipShard (libslam_hip.so).
"""
import torch
import torch.distributed as dist

from . import capi
from .mcl import Filter


class HipShard:
    """Shard backend on one MI355X through the C ABI."""

    def __init__(self, n_local, rows, cols, rank, world, device, seed, stream=None, mode=capi.MODE_LOCALIZATION, **kw):
        if mode == capi.MODE_SLAM:
            kw.setdefault("map_storage", capi.MAPS_DENSE)
        self.f = Filter(n_local, rows, cols, mode=mode, device=device, seed=seed,
                        sort=capi.SORT_NONE, prefix_order=capi.PREFIX_BLOCKED, global_first=rank * n_local,
                        global_n=world * n_local, stream=stream, **kw)
        self.n_local = n_local
        self.device = torch.device(f"cuda:{device}")
        self.slam_mode = mode == capi.MODE_SLAM
        self.map_stride = int(self.f.device_maps().map_stride) if self.slam_mode else 0
        # tiled canvases travel as tables + distinct tiles (SLAM_MAPS_TILED; include/slam_hip.h steps 2', 3')
        self.tiled_wire = self.slam_mode and kw.get("map_storage") == capi.MAPS_TILED
        self.tiles_per_canvas = ((rows + 63) // 64) * ((cols + 63) // 64)

    def set_map(self, grid):
        self.f.set_map(grid)

    def canvas_shape(self):
        return (self.f.rows, self.f.cols)

    def predict(self, odom, alphas):
        self.f.predict(odom, alphas)

    def update_weights(self, angles, dists, stddev, max_dist, offset):
        self.f.update_weights(angles, dists, stddev, max_dist, offset)

    def packed_state(self):
        return self.f.state_tensor().reshape(-1)

    def new_gather_buffer(self, world):
        return torch.empty(world * 4 * self.n_local, dtype=torch.float64, device=self.device)

    def shard_resample(self, gathered, world):
        self.f.shard_resample(gathered.data_ptr(), world)

    def shard_resample_split(self, gathered_poses, gathered_weights, world):
        self.f.shard_resample_split(gathered_poses.data_ptr(), gathered_weights.data_ptr(), world)

    # what the caller reads every frame (apps/grid_slam.cpp:112-119), this shard's part
    def summary(self):
        return torch.from_numpy(self.f.shard_summary())

    def pose_of(self, local_slot):
        x, y, th = self.f.poses(int(local_slot), 1)
        return float(x[0]), float(y[0]), float(th[0])

    def map_of(self, local_slot):
        return torch.from_numpy(self.f.map_of(int(local_slot)))

    # SLAM mode: the three local pieces around the canvas exchange
    def shard_select(self, gathered, world):
        self.f.shard_select(gathered.data_ptr(), world)
        return self.f.index_tensor().to(torch.int64)

    def pack_maps(self, local_idx):
        idx32 = local_idx.to(torch.int32).contiguous()
        out = torch.empty((idx32.numel(), self.map_stride), dtype=torch.uint8, device=self.device)
        self.f.pack_maps(idx32.data_ptr(), idx32.numel(), out.data_ptr())
        return out

    def clone_maps(self, pool, pos):
        pos32 = pos.to(torch.int32).contiguous()
        self.f.shard_clone_maps(pool.contiguous().data_ptr(), pool.shape[0], pos32.data_ptr())

    def pack_tiles(self, local_idx):
        """(tables (count, tiles_per_canvas) int32, tiles (n_tiles, 4096) uint8) of the requested canvases."""
        idx32 = local_idx.to(torch.int32).contiguous()
        n_tiles = self.f.pack_tiles_count(idx32.data_ptr(), idx32.numel())
        tabs = torch.empty((idx32.numel(), self.tiles_per_canvas), dtype=torch.int32, device=self.device)
        tiles = torch.empty((n_tiles, capi.TILE_BYTES), dtype=torch.uint8, device=self.device)
        self.f.pack_tiles(idx32.data_ptr(), idx32.numel(), tabs.data_ptr(), tiles.data_ptr(), n_tiles)
        return tabs, tiles

    def import_tiles(self, tabs, tiles, pos):
        pos32 = pos.to(torch.int32).contiguous()
        self.f.shard_import_tiles(tabs.contiguous().data_ptr(), tabs.shape[0], tiles.contiguous().data_ptr(), tiles.shape[0],
                                  pos32.data_ptr())


class ShardedFilter:
    """The filter step over `world` ranks.  `backend` is the local shard (HipShard in production)."""

    def __init__(self, backend, world=None, group=None, always_collective=False, split_exchange=False, stage_exchange=False):
        self.b = backend
        self.stage_exchange = stage_exchange
        self.group = group
        self.always_collective = always_collective  # issue the all-gather even for one rank (rehearsal)
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self._gathered = backend.new_gather_buffer(self.world)
        self._stage = None
        # split exchange (localization mode): poses gathered early on a side stream, weights late
        self._split = split_exchange and (self.world > 1 or always_collective) and not getattr(backend, "slam_mode", False) \
            and hasattr(backend, "shard_resample_split")
        self._gp = self._gw = self._stage_p = self._stage_w = self._comm = self._poses_done = None

    def predict(self, odom, alphas):
        self.b.predict(odom, alphas)
        if self._split:
            self._gather_poses()

    def _gather_poses(self):
        """All-gather of [x | y | theta] (final once predict has run), off the critical path."""
        local = self.b.packed_state()
        n3 = 3 * (local.numel() // 4)
        if self._gp is None:
            self._stage_p, self._stage_w = torch.empty_like(local[:n3]), torch.empty_like(local[n3:])
            self._gp = torch.empty(self.world * n3, dtype=local.dtype, device=local.device)
            self._gw = torch.empty(self.world * (local.numel() - n3), dtype=local.dtype, device=local.device)
            if local.is_cuda:
                self._comm = torch.cuda.Stream(device=local.device)
        if self._comm is None:
            dist.all_gather_into_tensor(self._gp, local[:n3], group=self.group)
            return
        # the side stream waits for predict (and, through it, for the previous resample, the last
        # reader of _gp); the resample of this step waits for _poses_done
        self._comm.wait_event(torch.cuda.current_stream(local.device).record_event())
        with torch.cuda.stream(self._comm):
            dist.all_gather_into_tensor(self._gp, local[:n3], group=self.group)
            self._poses_done = self._comm.record_event()

    # ---- read-out across ranks: slam::average_pose and get_particles().front() ------------------------------------
    def summarize(self):
        """One small all-gather (64 B per rank): every rank's {sum x, sum y, sum cos, sum sin, n, best weight, best
        global slot, best local slot}, combined in rank order by the library's host arithmetic, so every rank holds the
        same bits.  Returns (Pose-like (x, y, theta) mean of ALL particles: slam/util.cpp:66-85 at apps/grid_slam.cpp:119;
        best = dict(weight, global_slot, rank, local_slot): the particle a sorted unsharded filter holds at the front,
        slam/mcl.cpp:236-238 with ties to the lowest slot)."""
        import ctypes

        import numpy as np

        mine = self.b.summary().to(torch.float64).reshape(-1)
        if self.world > 1 or self.always_collective:
            dev = self._gathered.device
            allq = torch.empty(self.world * mine.numel(), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(allq, mine.to(dev), group=self.group)
            allq = allq.cpu()
        else:
            allq = mine
        q = np.ascontiguousarray(allq.numpy(), dtype=np.float64)
        lib = capi.load_library()
        avg = np.empty(3)
        bw, slot, rank, loc = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int64()
        capi.check(lib.slam_shard_summary_combine(capi.ptr(q), self.world, capi.ptr(avg), ctypes.byref(bw), ctypes.byref(slot),
                                                  ctypes.byref(rank), ctypes.byref(loc)), lib)
        return (float(avg[0]), float(avg[1]), float(avg[2])), {"weight": bw.value, "global_slot": slot.value, "rank": rank.value,
                                                               "local_slot": loc.value}

    def average_pose(self):
        return self.summarize()[0]

    def best(self, with_map=False):
        """get_particles().front() of the whole filter: pose, weight and (SLAM mode, with_map) canvas of the heaviest
        particle, served by the rank that owns it (one broadcast of 24 B, plus rows x cols bytes for the canvas)."""
        _, best = self.summarize()
        me = dist.get_rank(self.group) if (dist.is_initialized() and self.world > 1) else 0
        dev = self._gathered.device
        pose = torch.zeros(3, dtype=torch.float64)
        if me == best["rank"]:
            pose = torch.tensor(self.b.pose_of(best["local_slot"]), dtype=torch.float64)
        canvas = None
        if self.world > 1:
            src = dist.get_global_rank(self.group, best["rank"]) if self.group is not None else best["rank"]
            pose = pose.to(dev)
            dist.broadcast(pose, src=src, group=self.group)
            pose = pose.cpu()
        if with_map and getattr(self.b, "slam_mode", False):
            if me == best["rank"]:
                canvas = self.b.map_of(best["local_slot"]).to(dev)
            if self.world > 1:
                if canvas is None:
                    canvas = torch.empty(self.b.canvas_shape(), dtype=torch.uint8, device=dev)
                dist.broadcast(canvas, src=src, group=self.group)
            canvas = canvas.cpu().numpy()
        best.update(x=float(pose[0]), y=float(pose[1]), theta=float(pose[2]), map=canvas)
        return best

    def update(self, angles, dists, stddev, max_dist, offset=(0.0, 0.0, 0.0)):
        self.b.update_weights(angles, dists, stddev, max_dist, offset)
        local = self.b.packed_state()
        if self._split:
            dist.all_gather_into_tensor(self._gw, local[self._stage_p.numel():], group=self.group)
            if self._poses_done is not None:
                torch.cuda.current_stream(local.device).wait_event(self._poses_done)
            self.b.shard_resample_split(self._gp, self._gw, self.world)
            return
        if self.world > 1 or self.always_collective:
            # the one exchange step of the path: all-gather of [x | y | theta | w] of every shard, straight
            # out of the library's resident buffer (a zero-copy view: RCCL reads plain device memory, no
            # staging copy; `stage_exchange` keeps the old detour through a torch-allocated tensor)
            if self.stage_exchange:
                if self._stage is None:
                    self._stage = torch.empty_like(local)
                self._stage.copy_(local)
                local = self._stage
            dist.all_gather_into_tensor(self._gathered, local, group=self.group)
        else:
            self._gathered.copy_(local)
        if not getattr(self.b, "slam_mode", False):
            self.b.shard_resample(self._gathered, self.world)
            return
        # SLAM mode (per-particle canvases): the selected canvases live with their owners.
        #   want  = the distinct global sources of my slots, sorted -> grouped by owner
        #   each owner learns which of its canvases every rank wants (two small all-to-alls),
        #   packs them once per requester and the canvases travel in ONE all-to-all;
        #   my slot k then clones row searchsorted(want, idx[k]) of what arrived.
        idx = self.b.shard_select(self._gathered, self.world)
        want = torch.unique(idx)
        n_loc = self.b.n_local
        owner = torch.div(want, n_loc, rounding_mode="floor")
        n_want = torch.bincount(owner, minlength=self.world)
        local_ids = want - owner * n_loc
        tiled = getattr(self.b, "tiled_wire", False)
        if self.world > 1:
            n_req = torch.empty_like(n_want)
            dist.all_to_all_single(n_req, n_want, group=self.group)
            want_split, req_split = n_want.tolist(), n_req.tolist()
            req = torch.empty(sum(req_split), dtype=local_ids.dtype, device=local_ids.device)
            dist.all_to_all_single(req, local_ids, req_split, want_split, group=self.group)
            if tiled:
                # per requester: tables of its canvases + every distinct tile they name, once
                packs = [self.b.pack_tiles(part) for part in req.split(req_split)]
                n_send = torch.tensor([t.shape[0] for _, t in packs], dtype=n_want.dtype, device=n_want.device)
                n_recv = torch.empty_like(n_send)
                dist.all_to_all_single(n_recv, n_send, group=self.group)
                send_tabs, send_tiles = torch.cat([a for a, _ in packs]), torch.cat([t for _, t in packs])
                tabs = torch.empty((want.numel(), send_tabs.shape[1]), dtype=send_tabs.dtype, device=send_tabs.device)
                dist.all_to_all_single(tabs, send_tabs, want_split, req_split, group=self.group)
                tiles = torch.empty((int(n_recv.sum()), send_tiles.shape[1]), dtype=torch.uint8, device=send_tiles.device)
                dist.all_to_all_single(tiles, send_tiles, n_recv.tolist(), n_send.tolist(), group=self.group)
                # table entries count from 1 within their owner's payload: shift by the owner's first tile
                base = torch.repeat_interleave(torch.cumsum(n_recv, 0) - n_recv, n_want).to(tabs.dtype)
                tabs = torch.where(tabs > 0, tabs + base[:, None], tabs)
                self.b.import_tiles(tabs, tiles, torch.searchsorted(want, idx))
                return
            send = self.b.pack_maps(req)
            pool = torch.empty((want.numel(), send.shape[1]), dtype=torch.uint8, device=send.device)
            dist.all_to_all_single(pool, send, want_split, req_split, group=self.group)
        elif tiled:
            tabs, tiles = self.b.pack_tiles(local_ids)
            self.b.import_tiles(tabs, tiles, torch.searchsorted(want, idx))
            return
        else:
            pool = self.b.pack_maps(local_ids)
        self.b.clone_maps(pool, torch.searchsorted(want, idx))
