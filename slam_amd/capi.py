"""ctypes binding of libslam_hip.so (include/slam_hip.h).  No CPU fallback: if the HIP library is
missing or fails to load, importing the product path raises."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLAM_HIP_LIB") or os.path.join(_HERE, "lib", "libslam_hip.so")  # override: experiments only

MODE_LOCALIZATION, MODE_SLAM = 0, 1
NOISE_PHILOX, NOISE_INJECTED = 0, 1
PREFIX_BLOCKED, PREFIX_REFERENCE = 0, 1
RESAMPLE_MULTINOMIAL, RESAMPLE_SYSTEMATIC = 0, 1
SORT_FULL, SORT_NONE = 0, 1
WEIGHT_LINEAR, WEIGHT_LOG = 0, 1
TILE_SIDE, TILE_BYTES = 64, 4096  # SLAM_TILE_SIDE, SLAM_TILE_BYTES of include/slam_hip.h
SHARD_SUMMARY_DOUBLES = 8         # SLAM_SHARD_SUMMARY_DOUBLES
MAPS_AUTO, MAPS_DENSE, MAPS_TILED = 0, 1, 2
RAYCAST_FAST, RAYCAST_WALK, RAYCAST_FAST_BEAM_LANES, RAYCAST_FAST_PARTICLE_LANES, RAYCAST_TWO_PHASE, RAYCAST_TWO_PHASE_LITERAL = 0, 1, 2, 3, 4, 5
RAYCAST_FAST_PARTICLE_LANES_DIRECT = 6


class SlamError(RuntimeError):
    pass


class Params(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("n_particles", ctypes.c_int64),
        ("rows", ctypes.c_int32),
        ("cols", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("noise", ctypes.c_int32),
        ("seed", ctypes.c_uint64),
        ("prefix_order", ctypes.c_int32),
        ("resampler", ctypes.c_int32),
        ("sort", ctypes.c_int32),
        ("raycaster", ctypes.c_int32),
        ("global_first", ctypes.c_int64),
        ("global_n", ctypes.c_int64),
        ("stream", ctypes.c_void_p),
        ("async_launch", ctypes.c_int32),
        ("weight_domain", ctypes.c_int32),
        ("map_storage", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("tile_pool_tiles", ctypes.c_int64),
    ]


class DeviceMaps(ctypes.Structure):
    _fields_ = [
        ("maps_dev", ctypes.c_void_p),
        ("idx_dev", ctypes.c_void_p),
        ("map_stride", ctypes.c_int64),
        ("n_particles", ctypes.c_int64),
    ]


class DeviceState(ctypes.Structure):
    _fields_ = [
        ("state_dev", ctypes.c_void_p),
        ("prefix_dev", ctypes.c_void_p),
        ("n_particles", ctypes.c_int64),
        ("global_n", ctypes.c_int64),
    ]


_P = ctypes.c_void_p
_I32, _I64, _D = ctypes.c_int32, ctypes.c_int64, ctypes.c_double

# name -> (restype, argtypes); every symbol include/slam_hip.h declares
SYMBOLS = {
    "slam_last_error": (ctypes.c_char_p, []),
    "slam_version": (ctypes.c_int, []),
    "slam_mcl_create": (ctypes.c_int, [ctypes.POINTER(Params), ctypes.POINTER(_P)]),
    "slam_mcl_destroy": (None, [_P]),
    "slam_mcl_set_map_u8": (ctypes.c_int, [_P, _P, _I32, _I32]),
    "slam_mcl_get_clearance": (ctypes.c_int, [_P, _P, _I64, _P, _P]),
    "slam_mcl_set_poses": (ctypes.c_int, [_P, _P, _P, _P, _I64, _I64]),
    "slam_mcl_seed_gaussian": (ctypes.c_int, [_P, _P, _P]),
    "slam_mcl_seed_uniform_free": (ctypes.c_int, [_P]),
    "slam_mcl_set_weights": (ctypes.c_int, [_P, _P, _I64, _I64]),
    "slam_mcl_set_maps_u8": (ctypes.c_int, [_P, _P, _I64, _I64]),
    "slam_mcl_set_noise": (ctypes.c_int, [_P, _P, _P]),
    "slam_mcl_predict": (ctypes.c_int, [_P, _P, _P]),
    "slam_mcl_update": (ctypes.c_int, [_P, _P, _P, _I32, _D, _D, _P]),
    "slam_mcl_predict_velocity": (ctypes.c_int, [_P, _P, _D, _P]),
    "slam_mcl_update_weights": (ctypes.c_int, [_P, _P, _P, _I32, _D, _D, _P]),
    "slam_mcl_resample": (ctypes.c_int, [_P]),
    "slam_mcl_shard_resample": (ctypes.c_int, [_P, _P, _I32]),
    "slam_mcl_shard_select": (ctypes.c_int, [_P, _P, _I32]),
    "slam_mcl_shard_resample_split": (ctypes.c_int, [_P, _P, _P, _I32]),
    "slam_mcl_pack_maps": (ctypes.c_int, [_P, _P, _I64, _P]),
    "slam_mcl_shard_clone_maps": (ctypes.c_int, [_P, _P, _I64, _P]),
    "slam_mcl_pack_tiles_count": (ctypes.c_int, [_P, _P, _I64, _P]),
    "slam_mcl_pack_tiles": (ctypes.c_int, [_P, _P, _I64, _P, _P, _I64]),
    "slam_mcl_shard_import_tiles": (ctypes.c_int, [_P, _P, _I64, _P, _I64, _P]),
    "slam_mcl_shift_tables": (ctypes.c_int, [_P, _P, _I64, _I32]),
    "slam_mcl_device_maps_get": (ctypes.c_int, [_P, _P]),
    "slam_mcl_get_tile_stats": (ctypes.c_int, [_P, _P]),
    "slam_mcl_get_defer_stats": (ctypes.c_int, [_P, _P]),
    "slam_mcl_get_poses": (ctypes.c_int, [_P, _P, _P, _P, _I64, _I64]),
    "slam_mcl_get_weights": (ctypes.c_int, [_P, _P, _I64, _I64]),
    "slam_mcl_get_map_u8": (ctypes.c_int, [_P, _I64, _P]),
    "slam_mcl_get_last_indices": (ctypes.c_int, [_P, _P]),
    "slam_mcl_get_last_order": (ctypes.c_int, [_P, _P]),
    "slam_mcl_get_prefix": (ctypes.c_int, [_P, _P, _P]),
    "slam_mcl_get_hits": (ctypes.c_int, [_P, _P, _P]),
    "slam_mcl_record_hits": (ctypes.c_int, [_P, _I32]),
    "slam_mcl_collect_stats": (ctypes.c_int, [_P, _I32]),
    "slam_mcl_average_pose": (ctypes.c_int, [_P, _P]),
    "slam_mcl_shard_summary": (ctypes.c_int, [_P, _P]),
    "slam_shard_summary_combine": (ctypes.c_int, [_P, _I32, _P, _P, _P, _P, _P]),
    "slam_mcl_get_stats": (ctypes.c_int, [_P, _P]),
    "slam_mcl_profile": (ctypes.c_int, [_P, _I32]),
    "slam_mcl_get_kernel_ms": (ctypes.c_int, [_P, _P, _P, _I32]),
    "slam_mcl_device_state_get": (ctypes.c_int, [_P, ctypes.POINTER(DeviceState)]),
    "slam_mcl_synchronize": (ctypes.c_int, [_P]),
    "slam_fake_lidar_scan": (ctypes.c_int, [_I32, _P, _I32, _I32, _P, _I32, _D, _D, _D, _I32, _P]),
    "slam_raycast_u8": (ctypes.c_int, [_I32, _P, _I32, _I32, _P, _I64, _D, _D, _P]),
    "slam_raycast_batch": (ctypes.c_int, [_I32, _I32, _P, _I32, _I32, _P, _I64, _D, _D, _P]),
    "slam_map_create": (ctypes.c_int, [_I32, _P, _I32, _I32, ctypes.POINTER(_P)]),
    "slam_map_destroy": (None, [_P]),
    "slam_map_upload": (ctypes.c_int, [_P, _P]),
    "slam_map_download": (ctypes.c_int, [_P, _P]),
    "slam_map_fake_lidar_scan": (ctypes.c_int, [_P, _P, _I32, _D, _D, _D, _I32, _P]),
    "slam_map_raycast_batch": (ctypes.c_int, [_P, _I32, _P, _I64, _D, _D, _P]),
    "slam_map_raycast_mapping": (ctypes.c_int, [_P, _P, _P, _I64, _D, _D, _P]),
    "slam_static_init": (ctypes.c_int, [_I32]),
    "slam_resample_select_static": (ctypes.c_int, [_I32, _P, _P, _I64, _I32, _P, ctypes.c_uint64, ctypes.c_uint64, _P]),
}
SELECT_STATIC_MAX = 262144  # SLAM_SELECT_STATIC_MAX of include/slam_hip.h

_lib = None


def load_library(path=None):
    """Loads libslam_hip.so and binds every declared symbol.  Raises SlamError if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise SlamError(
            f"{p} not found: build it with `python -m slam_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback."
        )
    # PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one process
    # leave the second one without GPUs.  Importing torch first makes this library bind to the
    # runtime torch already loaded (same SONAME), so torch streams/RCCL and our kernels share it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load_library()
        raise SlamError(f"slam_hip error {rc}: {lib.slam_last_error().decode()}")


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
