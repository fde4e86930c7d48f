"""One-process-per-GPU sharded filter: particles are split into contiguous shards,
one per rank; predict and weighting are local; resampling is global and needs ONE exchange step, an
all-gather of the packed per-shard state (32 B per particle) over RCCL/xGMI (`nccl` backend).  Every
rank then rebuilds the running sum of all weights in the canonical blocked order, draws its own
output slots and copies the selected particles out of the gathered buffer.  The result is
bit-identical for any number of ranks (tests/test_dist_cpu.py, tests/test_gpu_parity.py).

The per-shard compute is behind a small backend interface so that the collective logic can be
exercised with `gloo` on CPU tensors in the tests; the product backend is HipShard (libslam_hip.so).
"""
import torch
import torch.distributed as dist

from . import capi
from .mcl import Filter


class HipShard:
    """Shard backend on one MI355X through the C ABI."""

    def __init__(self, n_local, rows, cols, rank, world, device, seed, stream=None, **kw):
        self.f = Filter(n_local, rows, cols, mode=capi.MODE_LOCALIZATION, device=device, seed=seed,
                        sort=capi.SORT_NONE, prefix_order=capi.PREFIX_BLOCKED, global_first=rank * n_local,
                        global_n=world * n_local, stream=stream, **kw)
        self.n_local = n_local
        self.device = torch.device(f"cuda:{device}")

    def set_map(self, grid):
        self.f.set_map(grid)

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


class ShardedFilter:
    """The filter step over `world` ranks.  `backend` is the local shard (HipShard in production)."""

    def __init__(self, backend, world=None, group=None):
        self.b = backend
        self.group = group
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self._gathered = backend.new_gather_buffer(self.world)
        self._stage = None

    def predict(self, odom, alphas):
        self.b.predict(odom, alphas)

    def update(self, angles, dists, stddev, max_dist, offset=(0.0, 0.0, 0.0)):
        self.b.update_weights(angles, dists, stddev, max_dist, offset)
        local = self.b.packed_state()
        if self.world > 1:
            # the one exchange step of the path: all-gather of [x | y | theta | w] of every shard.
            # The library's buffer is viewed zero-copy; it is staged through a torch-allocated
            # tensor (32 B/particle) so that RCCL only ever sees allocator-owned memory.
            if self._stage is None:
                self._stage = torch.empty_like(local)
            self._stage.copy_(local)
            dist.all_gather_into_tensor(self._gathered, self._stage, group=self.group)
        else:
            self._gathered.copy_(local)
        self.b.shard_resample(self._gathered, self.world)
