"""Host-side Python mirror of the reference's filter interface (slam/mcl.h:12-46) over the C ABI.

`Filter` is a thin handle wrapper (one call per C entry point).  `MCL` keeps the reference's names,
argument meaning and defaults so that tests read like the reference's callers
(benchmark/mcl.cpp:25-64, apps/grid_slam.cpp:100-168).  All compute runs in libslam_hip.so on the
GPU; nothing here falls back to the CPU.
"""
import ctypes
import math
from dataclasses import dataclass

import numpy as np

from . import capi
from .capi import check, f64, ptr


@dataclass
class Pose:  # slam/pose.h:7-12
    x: float
    y: float
    theta: float


@dataclass
class Odometry:  # slam/pose.h:19-24
    rotation_1: float
    translation: float
    rotation_2: float


class Filter:
    """One shard of particles resident on one GPU (handle of include/slam_hip.h)."""

    def __init__(self, n_particles, rows, cols, mode=capi.MODE_LOCALIZATION, device=0, seed=0x5EED0001,
                 noise=capi.NOISE_PHILOX, prefix_order=capi.PREFIX_BLOCKED, resampler=capi.RESAMPLE_MULTINOMIAL,
                 sort=capi.SORT_FULL, raycaster=capi.RAYCAST_FAST, global_first=0, global_n=0, stream=None,
                 async_launch=False, weight_domain=capi.WEIGHT_LINEAR, map_storage=capi.MAPS_AUTO, tile_pool_tiles=0):
        self.lib = capi.load_library()
        p = capi.Params()
        p.struct_size = ctypes.sizeof(capi.Params)
        p.mode = mode
        p.n_particles = n_particles
        p.rows, p.cols = rows, cols
        p.device = device
        p.noise = noise
        p.seed = seed
        p.prefix_order = prefix_order
        p.resampler = resampler
        p.sort = sort
        p.raycaster = raycaster
        p.global_first = global_first
        p.global_n = global_n
        p.stream = stream
        p.async_launch = 1 if async_launch else 0
        p.weight_domain = weight_domain
        p.map_storage = map_storage
        p.tile_pool_tiles = tile_pool_tiles
        self.params = p
        self.n = int(n_particles)
        self.global_n = int(global_n) if global_n else int(n_particles)
        self.global_first = int(global_first) if global_n else 0
        self.rows, self.cols = int(rows), int(cols)
        self.mode = mode
        self._h = ctypes.c_void_p()
        check(self.lib.slam_mcl_create(ctypes.byref(p), ctypes.byref(self._h)), self.lib)
        self._last_B = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.slam_mcl_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ---------------------------------------------------------------------------------
    def set_map(self, grid_u8):
        g = np.ascontiguousarray(grid_u8, dtype=np.uint8)
        check(self.lib.slam_mcl_set_map_u8(self._h, ptr(g), g.shape[0], g.shape[1]), self.lib)

    def clearance(self):
        """The clearance planes built by set_map: array (n_planes, plane_stride) of uint8 (slam_mcl_get_clearance)."""
        import ctypes

        n, stride = ctypes.c_int32(0), ctypes.c_int64(0)
        check(self.lib.slam_mcl_get_clearance(self._h, None, 0, ctypes.byref(n), ctypes.byref(stride)), self.lib)
        out = np.empty((n.value, stride.value), dtype=np.uint8)
        check(self.lib.slam_mcl_get_clearance(self._h, ptr(out), out.size, ctypes.byref(n), ctypes.byref(stride)), self.lib)
        return out

    def set_poses(self, x, y, theta, first=0):
        x, y, theta = f64(x), f64(y), f64(theta)
        check(self.lib.slam_mcl_set_poses(self._h, ptr(x), ptr(y), ptr(theta), first, x.size), self.lib)

    def set_weights(self, w, first=0):
        w = f64(w)
        check(self.lib.slam_mcl_set_weights(self._h, ptr(w), first, w.size), self.lib)

    def set_maps(self, maps_u8, first=0):
        m = np.ascontiguousarray(maps_u8, dtype=np.uint8)
        check(self.lib.slam_mcl_set_maps_u8(self._h, ptr(m), first, m.shape[0]), self.lib)

    def seed_gaussian(self, mean, sigma):
        m, s = f64(mean), f64(sigma)
        check(self.lib.slam_mcl_seed_gaussian(self._h, ptr(m), ptr(s)), self.lib)

    def seed_uniform_free(self):
        check(self.lib.slam_mcl_seed_uniform_free(self._h), self.lib)

    def set_noise(self, unit_normals=None, canonical_uniforms=None):
        z = f64(unit_normals) if unit_normals is not None else None
        u = f64(canonical_uniforms) if canonical_uniforms is not None else None
        check(self.lib.slam_mcl_set_noise(self._h, ptr(z), ptr(u)), self.lib)

    # -- the step ------------------------------------------------------------------------------
    def predict(self, odom, alphas):
        o, a = f64(odom), f64(alphas)
        check(self.lib.slam_mcl_predict(self._h, ptr(o), ptr(a)), self.lib)

    def predict_velocity(self, vel, dt, alphas):
        """slam/motion.cpp:34-56 for every particle; vel = (v, w), six alphas."""
        v = np.ascontiguousarray(vel, dtype=np.float64)
        a = np.ascontiguousarray(alphas, dtype=np.float64)
        assert v.size == 2 and a.size == 6
        check(self.lib.slam_mcl_predict_velocity(self._h, ptr(v), float(dt), ptr(a)), self.lib)

    def _scan(self, angles, dists, offset):
        a, d, o = f64(angles), f64(dists), f64(offset)
        self._last_B = a.size
        return a, d, o

    def update(self, angles, dists, stddev, max_dist, scanner_offset=(0.0, 0.0, 0.0)):
        a, d, o = self._scan(angles, dists, scanner_offset)
        check(self.lib.slam_mcl_update(self._h, ptr(a), ptr(d), a.size, stddev, max_dist, ptr(o)), self.lib)

    def update_weights(self, angles, dists, stddev, max_dist, scanner_offset=(0.0, 0.0, 0.0)):
        a, d, o = self._scan(angles, dists, scanner_offset)
        check(self.lib.slam_mcl_update_weights(self._h, ptr(a), ptr(d), a.size, stddev, max_dist, ptr(o)), self.lib)

    def resample(self):
        check(self.lib.slam_mcl_resample(self._h), self.lib)

    def shard_resample(self, gathered_state_ptr, n_shards):
        """gathered_state_ptr: device address of the all-gathered packed states ([n_shards][4][n])."""
        check(self.lib.slam_mcl_shard_resample(self._h, ctypes.c_void_p(int(gathered_state_ptr)), n_shards), self.lib)

    def shard_resample_split(self, gathered_poses_ptr, gathered_weights_ptr, n_shards):
        """As shard_resample with poses ([n_shards][3][n]) and weights ([n_shards][n]) gathered separately."""
        check(self.lib.slam_mcl_shard_resample_split(self._h, ctypes.c_void_p(int(gathered_poses_ptr)),
                                                     ctypes.c_void_p(int(gathered_weights_ptr)), n_shards), self.lib)

    def shard_select(self, gathered_state_ptr, n_shards):
        """SLAM-mode shards, step 1: poses and weights resampled, canvases pending (see include/slam_hip.h)."""
        check(self.lib.slam_mcl_shard_select(self._h, ctypes.c_void_p(int(gathered_state_ptr)), n_shards), self.lib)

    def pack_maps(self, local_idx_ptr, count, out_ptr):
        check(self.lib.slam_mcl_pack_maps(self._h, ctypes.c_void_p(int(local_idx_ptr)), count,
                                          ctypes.c_void_p(int(out_ptr))), self.lib)

    def shard_clone_maps(self, pool_ptr, pool_count, pos_ptr):
        check(self.lib.slam_mcl_shard_clone_maps(self._h, ctypes.c_void_p(int(pool_ptr)), pool_count,
                                                 ctypes.c_void_p(int(pos_ptr))), self.lib)

    def pack_tiles_count(self, local_idx_ptr, count):
        """Tiled canvases, sparse exchange: how many distinct tiles the `count` requested canvases name."""
        n = ctypes.c_int64(0)
        check(self.lib.slam_mcl_pack_tiles_count(self._h, ctypes.c_void_p(int(local_idx_ptr)), count, ctypes.byref(n)), self.lib)
        return n.value

    def pack_tiles(self, local_idx_ptr, count, tabs_ptr, tiles_ptr, n_tiles):
        check(self.lib.slam_mcl_pack_tiles(self._h, ctypes.c_void_p(int(local_idx_ptr)), count, ctypes.c_void_p(int(tabs_ptr)),
                                           ctypes.c_void_p(int(tiles_ptr)), n_tiles), self.lib)

    def shard_import_tiles(self, tabs_ptr, n_tabs, tiles_ptr, n_tiles, pos_ptr):
        check(self.lib.slam_mcl_shard_import_tiles(self._h, ctypes.c_void_p(int(tabs_ptr)), n_tabs,
                                                   ctypes.c_void_p(int(tiles_ptr)), n_tiles, ctypes.c_void_p(int(pos_ptr))), self.lib)

    def device_maps(self):
        dm = capi.DeviceMaps()
        check(self.lib.slam_mcl_device_maps_get(self._h, ctypes.byref(dm)), self.lib)
        return dm

    def index_tensor(self):
        """Zero-copy torch view (n,) int32 of the global source indices of the last select."""
        import torch

        dm = self.device_maps()

        class _Arr:
            __cuda_array_interface__ = {"shape": (self.n,), "typestr": "<i4", "data": (int(dm.idx_dev), False),
                                        "version": 2, "strides": None}

        return torch.as_tensor(_Arr(), device=f"cuda:{self.params.device}")

    def state_tensor(self):
        """Zero-copy torch view (4, n) float64 of the packed resident state [x | y | theta | weight].
        The buffer changes after every resample; call again after each update."""
        import torch

        st = self.device_state()

        class _Arr:
            __cuda_array_interface__ = {"shape": (4, self.n), "typestr": "<f8", "data": (int(st.state_dev), False),
                                        "version": 2, "strides": None}

        return torch.as_tensor(_Arr(), device=f"cuda:{self.params.device}")

    def synchronize(self):
        check(self.lib.slam_mcl_synchronize(self._h), self.lib)

    # -- read-back -----------------------------------------------------------------------------
    def poses(self, first=0, count=None):
        count = self.n - first if count is None else count
        x, y, t = (np.empty(count) for _ in range(3))
        check(self.lib.slam_mcl_get_poses(self._h, ptr(x), ptr(y), ptr(t), first, count), self.lib)
        return x, y, t

    def weights(self, first=0, count=None):
        count = self.n - first if count is None else count
        w = np.empty(count)
        check(self.lib.slam_mcl_get_weights(self._h, ptr(w), first, count), self.lib)
        return w

    def map_of(self, particle):
        m = np.empty((self.rows, self.cols), dtype=np.uint8)
        check(self.lib.slam_mcl_get_map_u8(self._h, particle, ptr(m)), self.lib)
        return m

    def last_indices(self):
        idx = np.empty(self.n, dtype=np.int32)
        check(self.lib.slam_mcl_get_last_indices(self._h, ptr(idx)), self.lib)
        return idx

    def last_order(self):
        perm = np.empty(self.n, dtype=np.int32)
        check(self.lib.slam_mcl_get_last_order(self._h, ptr(perm)), self.lib)
        return perm

    def prefix(self):
        P = np.empty(self.global_n)
        s = np.empty(1)
        check(self.lib.slam_mcl_get_prefix(self._h, ptr(P), ptr(s)), self.lib)
        return P, float(s[0])

    def collect_stats(self, enable=True):
        """Counters of stats() on (default) or off: they are instrumentation and cost the grid-SLAM kernel about 5 %."""
        check(self.lib.slam_mcl_collect_stats(self._h, 1 if enable else 0), self.lib)

    def record_hits(self, enable=True):
        check(self.lib.slam_mcl_record_hits(self._h, 1 if enable else 0), self.lib)

    def hits(self):
        B = self._last_B
        xy = np.empty((self.n, B, 2))
        cell = np.empty((self.n, B), dtype=np.int32)
        check(self.lib.slam_mcl_get_hits(self._h, ptr(xy), ptr(cell)), self.lib)
        return xy, cell

    def average_pose(self):
        out = np.empty(3)
        check(self.lib.slam_mcl_average_pose(self._h, ptr(out)), self.lib)
        return Pose(*out)

    def shard_summary(self):
        """This shard's part of average_pose and its heaviest particle (slam_mcl_shard_summary): 8 doubles
        {sum x, sum y, sum cos, sum sin, n, best weight, best global slot, best local slot}."""
        out = np.empty(capi.SHARD_SUMMARY_DOUBLES)
        check(self.lib.slam_mcl_shard_summary(self._h, ptr(out)), self.lib)
        return out

    def stats(self):
        out = np.zeros(3, dtype=np.int64)
        check(self.lib.slam_mcl_get_stats(self._h, ptr(out)), self.lib)
        return {"cells_read": int(out[0]), "cells_written": int(out[1]), "steps": int(out[2])}

    def tile_stats(self):
        out = np.zeros(3, dtype=np.int64)
        check(self.lib.slam_mcl_get_tile_stats(self._h, ptr(out)), self.lib)
        return {"pool": int(out[0]), "live": int(out[1]), "allocated_since": int(out[2])}

    def defer_stats(self):
        """Tiled canvases: how many updates left their canvases for the resample that followed (include/slam_hip.h)."""
        out = np.zeros(3, dtype=np.int64)
        check(self.lib.slam_mcl_get_defer_stats(self._h, ptr(out)), self.lib)
        return {"updates": int(out[0]), "deferred": int(out[1]), "edge_particles": int(out[2])}

    def profile(self, enable=True):
        """True / 1: an event pair around every kernel group; 2: the weighting pass only, timed on its kernels' own
        dispatch timestamps where the path allows (no event records between the kernels); False / 0: off."""
        check(self.lib.slam_mcl_profile(self._h, 2 if enable == 2 else (1 if enable else 0)), self.lib)

    def kernel_ms(self, reset=True):
        ms = np.zeros(5)
        n = np.zeros(5, dtype=np.int64)
        check(self.lib.slam_mcl_get_kernel_ms(self._h, ptr(ms), ptr(n), 1 if reset else 0), self.lib)
        names = ("predict", "weights", "resample", "sort", "clone")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(names)}

    def device_state(self):
        st = capi.DeviceState()
        check(self.lib.slam_mcl_device_state_get(self._h, ctypes.byref(st)), self.lib)
        return st


class GridMap:
    """One uint8 grid resident on the GPU (slam_map_* of include/slam_hip.h): FakeLidar scans, batched
    read-only raycasts (slam/raycast.cpp:7-141, slam/fake_lidar.cpp:15-33) and raycast_mapping on one
    canvas (slam/raycast.cpp:143-223) without re-uploading the grid per call."""

    def __init__(self, grid_u8, device=0):
        self.lib = capi.load_library()
        g = np.ascontiguousarray(grid_u8, dtype=np.uint8)
        self.rows, self.cols = g.shape
        self._h = ctypes.c_void_p()
        check(self.lib.slam_map_create(device, ptr(g), self.rows, self.cols, ctypes.byref(self._h)), self.lib)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.slam_map_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, grid_u8):
        g = np.ascontiguousarray(grid_u8, dtype=np.uint8)
        assert g.shape == (self.rows, self.cols)
        check(self.lib.slam_map_upload(self._h, ptr(g)), self.lib)

    def download(self):
        g = np.empty((self.rows, self.cols), dtype=np.uint8)
        check(self.lib.slam_map_download(self._h, ptr(g)), self.lib)
        return g

    def fake_lidar_scan(self, poses3, start, stop, max_dist, n_rays):
        """FakeLidar::scan for every pose (x, y, theta): hits (n_poses, n_rays, 2), (-1, -1) = miss; grid 0 = wall."""
        p = np.ascontiguousarray(poses3, dtype=np.float64).reshape(-1, 3)
        hits = np.empty((p.shape[0], n_rays, 2))
        check(self.lib.slam_map_fake_lidar_scan(self._h, ptr(p), p.shape[0], start, stop, max_dist, n_rays, ptr(hits)), self.lib)
        return hits

    def raycast(self, kind, poses3, max_distance_squared, step_size=0.5):
        """kind 1: raycast<unsigned char> (hit = cell < 128); kind 2: raycast<int> on a 0/1 grid (hit = 0)."""
        p = np.ascontiguousarray(poses3, dtype=np.float64).reshape(-1, 3)
        hits = np.empty((p.shape[0], 2))
        check(self.lib.slam_map_raycast_batch(self._h, kind, ptr(p), p.shape[0], max_distance_squared, step_size, ptr(hits)), self.lib)
        return hits

    def raycast_mapping(self, poses3, z_squared, z_max_squared, step_size=1.0):
        """slam::raycast_mapping for the rays in order on the resident canvas (which is updated): hits (n, 2)."""
        p = np.ascontiguousarray(poses3, dtype=np.float64).reshape(-1, 3)
        z2 = np.ascontiguousarray(z_squared, dtype=np.float64).reshape(-1)
        assert z2.size == p.shape[0]
        hits = np.empty((p.shape[0], 2))
        check(self.lib.slam_map_raycast_mapping(self._h, ptr(p), ptr(z2), p.shape[0], z_max_squared, step_size, ptr(hits)), self.lib)
        return hits


@dataclass
class Particle:  # slam/pose.h:32-37 (map is materialised on request)
    pose: Pose
    weight: float
    map: object = None


class MCL:
    """slam::MCL (slam/mcl.h:12-46): grid-SLAM filter with one canvas per particle."""

    def __init__(self, n_particles, canvas_size=(1000, 1000), **kw):
        width, height = canvas_size  # cv::Size is (width, height)
        self._canvas = (int(width), int(height))
        self._f = Filter(n_particles, rows=int(height), cols=int(width), mode=capi.MODE_SLAM, **kw)

    @property
    def filter(self):
        return self._f

    def starting_pose(self):  # slam/mcl.cpp:22-25
        return Pose(self._canvas[0] / 2.0, self._canvas[1] / 2.0, math.pi / 2)

    def predict(self, odom, alphas):  # slam/mcl.cpp:41-47
        o = (odom.rotation_1, odom.translation, odom.rotation_2) if isinstance(odom, Odometry) else odom
        self._f.predict(o, alphas)

    def update(self, scan, stddev, max_dist, scanner_offset=Pose(0, 0, 0), n_threads=-1):
        """scan: sequence of (angle, dist) (slam/mcl.cpp:102-106).  n_threads is accepted and ignored."""
        s = np.asarray(scan, dtype=np.float64).reshape(-1, 2)
        off = (scanner_offset.x, scanner_offset.y, scanner_offset.theta) if isinstance(scanner_offset, Pose) else scanner_offset
        self._f.update(s[:, 0], s[:, 1], stddev, max_dist, off)

    def get_particles(self, with_maps=False):
        """Poses and weights of all particles; the map of particle 0 (the highest weight after an
        update, which is all apps/grid_slam.cpp:112 reads) unless with_maps asks for every map."""
        x, y, t = self._f.poses()
        w = self._f.weights()
        out = [Particle(Pose(x[k], y[k], t[k]), w[k]) for k in range(self._f.n)]
        for k in range(self._f.n if with_maps else 1):
            out[k].map = self._f.map_of(k)
        return out

    @staticmethod
    def sensor_position(frame_position, scanner_offset):  # slam/mcl.cpp:88-100 (host scalar math)
        d = math.sqrt(float(int((0 - scanner_offset.x) ** 2 + (0 - scanner_offset.y) ** 2)))
        theta = math.atan2(scanner_offset.y, scanner_offset.x)
        dx = math.cos(frame_position.theta + theta) * d
        dy = math.sin(frame_position.theta + theta) * d
        return Pose(frame_position.x + dx, frame_position.y + dy, frame_position.theta + scanner_offset.theta)
