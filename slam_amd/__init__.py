"""slam_amd — MI355X (gfx950) implementation of the MCL / grid-SLAM particle-filter step of
ferasboulala/slam, reached through the C ABI in include/slam_hip.h (libslam_hip.so)."""
from . import capi
from .capi import SlamError, load_library
from .mcl import MCL, Filter, GridMap, Odometry, Particle, Pose

__all__ = ["capi", "SlamError", "load_library", "MCL", "Filter", "GridMap", "Odometry", "Particle", "Pose"]
