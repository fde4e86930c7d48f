"""CPU study for k_weights_L<0> on the 16384^2 shard: lock-step iterations of a 64-beam group in which at least one lane
runs the re-synchronisation block of the jump loop (tools/slowpath_sim.cpp).  Counting only."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc
from slam_amd import synth
hh = ctypes.CDLL(os.path.join(ROOT, "tests", "host_harness", "libray_harness.so"))
sp = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libslowpath.so"))
hh.hh_cone_plane_stride.restype = ctypes.c_ulong
size = int(os.environ.get("SIZE", 16384)); B = int(os.environ.get("B", 720)); zmax = float(os.environ.get("ZMAX", 4000)); NP = int(os.environ.get("NP", 600))
g = synth.make_grid(size)
stride = hh.hh_cone_plane_stride(size, size)
planes = np.empty(8 * stride, np.uint8)
hh.hh_build_cone_planes(g.ctypes.data_as(ctypes.c_void_p), size, size, planes.ctypes.data_as(ctypes.c_void_p))
true = synth.open_pose(g)
off = [0.0, 30.0, 0.0]
sensor = orc.sensor_position(true, off, orc.DET)
free01 = (g >= 128).astype(np.int32)
hits = orc.fake_lidar_scan(free01, sensor, 0.0, 2 * np.pi, zmax, B, orc.DET)
ang, dist = orc.hits_to_scan(hits, sensor, 0.0, 2 * np.pi, zmax)
x, y, th = orc.seed_uniform_free(1, 0, NP, g)
so = orc.scanner_offset(off)
G = (B + 63) // 64
reads = np.empty((NP, B), np.int32); ev = np.empty((NP, B), np.int32); it = np.empty((NP, G))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
sp.sp_pass(P(planes), size, size, ctypes.c_long(NP), P(x), P(y), P(th), B, P(ang), P(dist), ctypes.c_double(zmax), ctypes.c_double(0.5),
           ctypes.c_double(so[0]), ctypes.c_double(so[1]), ctypes.c_double(off[2]), P(reads), P(it), P(ev))
trips = np.maximum(reads - 1, 0)
pad = (-B) % 64
lock = np.pad(trips, ((0, 0), (0, pad))).reshape(NP, -1, 64).max(axis=2)
print("lookup iterations per group: %.1f; of which with a re-synchronisation in some lane: %.2f (%.0f %%)" % (lock.mean(), it.mean(), 100 * it.mean() / lock.mean()))
print("re-synchronisations per ray: %.3f; rays with any: %.1f %%; groups with any: %.1f %%" % (ev.mean(), 100 * (ev > 0).mean(), 100 * (it > 0).mean()))
